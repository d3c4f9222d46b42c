"""round 5: reproduce the hipGraphLaunch out-of-bounds read (profiles/r05_capture_crash.txt ADDENDUM 2) without the test suite.
Mechanism under test: hipGraphLaunch skips the exec's streams that share the launch stream's hardware queue, unchecked; new streams go to
the hardware queue with the fewest users, so after a BURST of exec destructions (the audition drops 3 of 4 candidates; tests drop models)
the next execs' streams pile onto one queue, and a launch stream that sits on that queue collides with all of them.
python tools/r05/hipgraph_oob_repro.py [iterations] [branches]      (pure torch; env GPU_MAX_HW_QUEUES etc. are what is being varied)"""
import gc, sys, time
import torch

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
branches = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
x = torch.zeros(1 << 16, device=dev)
users = [torch.cuda.Stream(dev) for _ in range(8)]          # torch's pool streams: created early, spread over the queues
sides = [torch.cuda.Stream(dev) for _ in range(branches)]


def make_graph():
    g = torch.cuda.CUDAGraph()
    gc.disable()
    try:
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            main = torch.cuda.current_stream()
            y = x * 2
            outs = []
            for s in sides[:branches - 1]:
                s.wait_stream(main)
                with torch.cuda.stream(s):
                    outs.append(y + 1)
            z = y * 3
            for s in sides[:branches - 1]:
                main.wait_stream(s)
            for o in outs:
                z = z + o
    finally:
        gc.enable()
    return g, z


# warm-up (allocator, kernels)
for s in sides:
    with torch.cuda.stream(s):
        (x * 2 + 1)
torch.cuda.synchronize()
t0 = time.time()
alive = []
launches = 0
for it in range(iters):
    cands = [make_graph() for _ in range(4)]
    keep = cands[it % 4]
    del cands                                               # burst: 3 execs (and their streams) destroyed
    alive.append(keep)
    if len(alive) > 6:                                      # models going out of scope: another burst now and then
        del alive[:4]
    for g, z in alive[-3:]:
        for u in users[:3 + it % 5]:
            with torch.cuda.stream(u):
                g.replay()
                launches += 1
    if it % 50 == 49:
        torch.cuda.synchronize()
        print("iteration %d: %d graph launches, %.1f s" % (it + 1, launches, time.time() - t0), flush=True)
torch.cuda.synchronize()
print("SURVIVED %d iterations, %d launches" % (iters, launches))
