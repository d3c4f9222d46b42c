"""round 5 probe for GPUTEST_r04's SIGSEGV: what happens when Python's cyclic GC frees an OLD torch.cuda.CUDAGraph (garbage kept alive by a
reference cycle, e.g. a pytest.raises traceback that holds a model + engine + graphs) while ANOTHER graph is being captured.
Scenarios run in subprocesses; prints each one's return code.   python tools/r05/gc_in_capture.py [scenario]"""
import gc, os, subprocess, sys, faulthandler
faulthandler.enable()


def make_graph(torch, n=3):
    x = torch.zeros(1 << 20, device="cuda")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            y = x * 2 + 1
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        y = x
        for _ in range(n):
            y = y * 2 + 1
    return g, x, y


def scenario(name):
    import torch
    if name == "gc_old_graph_in_capture":
        g, x, y = make_graph(torch)
        g.replay(); torch.cuda.synchronize()
        cyc = [g, x, y]; cyc.append(cyc)
        del g, x, y, cyc                                # garbage: only the cyclic GC can free it
        x2 = torch.zeros(1 << 20, device="cuda")
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, capture_error_mode="thread_local"):
            z = x2 * 3
            gc.collect()                                # the old graph's destructor runs HERE, inside the capture
            z = z + 1
        g2.replay(); torch.cuda.synchronize()
        print("ok", float(z[0]))
    elif name == "gc_old_graph_in_flight_in_capture":
        g, x, y = make_graph(torch, n=200)
        cyc = [g, x, y]; cyc.append(cyc)
        for _ in range(20):
            g.replay()
        del g, x, y, cyc
        x2 = torch.zeros(1 << 20, device="cuda")
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, capture_error_mode="thread_local"):
            z = x2 * 3
            gc.collect()
            z = z + 1
        g2.replay(); torch.cuda.synchronize()
        print("ok", float(z[0]))
    elif name == "del_old_graph_in_capture_global_mode":
        g, x, y = make_graph(torch)
        g.replay(); torch.cuda.synchronize()
        cyc = [g, x, y]; cyc.append(cyc)
        del g, x, y, cyc
        x2 = torch.zeros(1 << 20, device="cuda")
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            z = x2 * 3
            gc.collect()
            z = z + 1
        g2.replay(); torch.cuda.synchronize()
        print("ok", float(z[0]))
    elif name == "gc_plain_tensors_in_capture":
        t = [torch.zeros(1 << 22, device="cuda") for _ in range(8)]
        cyc = [t]; cyc.append(cyc)
        del t, cyc
        x2 = torch.zeros(1 << 20, device="cuda")
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, capture_error_mode="thread_local"):
            z = x2 * 3
            gc.collect()
            z = z + 1
        g2.replay(); torch.cuda.synchronize()
        print("ok", float(z[0]))
    elif name == "gc_old_graph_between_replays":
        g, x, y = make_graph(torch, n=200)
        g2, x2, y2 = make_graph(torch, n=200)
        cyc = [g, x, y]; cyc.append(cyc)
        for _ in range(50):
            g.replay(); g2.replay()
        del g, x, y, cyc
        gc.collect()
        for _ in range(50):
            g2.replay()
        torch.cuda.synchronize()
        print("ok", float(y2[0]))
    else:
        raise SystemExit("unknown scenario")


ALL = ["gc_plain_tensors_in_capture", "gc_old_graph_between_replays", "gc_old_graph_in_capture", "gc_old_graph_in_flight_in_capture",
       "del_old_graph_in_capture_global_mode"]
if __name__ == "__main__":
    if len(sys.argv) > 1:
        scenario(sys.argv[1])
    else:
        for s in ALL:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), s], capture_output=True, text=True, timeout=300)
            tail = (r.stdout + r.stderr).strip().splitlines()
            print("=== %s: rc=%d" % (s, r.returncode))
            for ln in tail[-25:]:
                print("    " + ln)
