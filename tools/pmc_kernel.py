"""Summarise rocprofv3 --pmc csv output: per kernel-name mean of each counter.
usage: python tools/pmc_kernel.py <dir-with-*_counter_collection.csv> [name-substring]"""
import csv
import glob
import sys
from collections import defaultdict


def main():
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            if pat not in name:
                continue
            key = name.split("(")[0][-70:]
            acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in acc.items():
        print(k)
        for c, v in sorted(d.items()):
            print("   %-28s %16.0f   (n=%d)" % (c, sum(v) / len(v), len(v)))


if __name__ == "__main__":
    main()
