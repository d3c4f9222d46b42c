"""round 5 probe: what ProcessGroupNCCL's flight recorder shows about eager collectives and when the watchdog retires them
(parallel._watchdog_idle relies on `retired`).  Run: python tools/r05/fr_probe.py  (1 rank, nccl backend)."""
import os, sys, time, pickle
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multiagentperception_amd import parallel  # sets TORCH_NCCL_TRACE_BUFFER_SIZE default
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29631")
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
x = torch.ones(1 << 20, device="cuda")
t0 = time.monotonic()
w = dist.all_reduce(x, async_op=True)
w.wait(); torch.cuda.synchronize()
for k in range(40):
    act = parallel._fr_entries(True); al = parallel._fr_entries(False)
    print("%.3f s: all %d active %d  %s" % (time.monotonic() - t0, len(al), len(act),
          [(e.get("profiling_name"), e.get("state"), e.get("retired")) for e in al][-3:]))
    if k == 0 and al:
        print("keys:", sorted(al[0].keys()))
    if al and all(e.get("retired") for e in al):
        break
    time.sleep(0.02)
for i in range(5):
    dist.all_reduce(x)
torch.cuda.synchronize()
print("idle wait: %.3f s" % parallel._watchdog_idle())
print("after:", [(e.get("state"), e.get("retired")) for e in parallel._fr_entries(False)])
dist.destroy_process_group()
