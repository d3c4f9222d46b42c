"""Which torch streams share a hardware queue?  A long spin on stream a, a trivial kernel on stream b right behind it (host order):
if b's kernel ends only after a's spin, the two streams are multiplexed on one queue.   python tools/r06/queue_probe.py [n_streams]"""
import sys
import torch

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
streams = [torch.cuda.default_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(n)]
x = torch.zeros(64, device=dev)
torch.cuda.synchronize()
SPIN = 400000                                        # cycles of torch.cuda._sleep (~200 us)


def shares(a, b):
    ea0, ea1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    eb = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(a):
        ea0.record()
        torch.cuda._sleep(SPIN)
        ea1.record()
    with torch.cuda.stream(b):
        x.add_(1.0)
        eb.record()
    torch.cuda.synchronize()
    spin = ea0.elapsed_time(ea1)
    return ea0.elapsed_time(eb) > 0.7 * spin, spin


for _ in range(2):                                   # warm both paths
    shares(streams[1], streams[2])
print("spin ms", round(shares(streams[1], streams[2])[1], 3))
print("     " + " ".join("%2d" % j for j in range(len(streams))) + "   (0 = the default stream; X = b waits for a)")
for i, a in enumerate(streams):
    row = []
    for j, b in enumerate(streams):
        row.append(" ." if i == j else (" X" if shares(a, b)[0] else " -"))
    print("%2d   " % i + " ".join(row))
