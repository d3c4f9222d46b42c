import sys, json
d = json.loads(sys.stdin.read())
print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"])
