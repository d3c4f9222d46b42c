"""Tensor-level wrappers over the C ABI (include/w2c_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every op below
hands raw device pointers + sizes to libw2c_hip.so on torch's CURRENT stream of
the tensor's device (so the path is re-entrant under DataParallel replicas and
capturable in a graph).  There is no fallback: CPU tensors raise.
"""
import threading

import torch

from . import _native
from ._native import W2CError, check

BF16 = torch.bfloat16


def _need_gpu(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise W2CError("w2c ops run only on an MI355X device tensor (got %s); there is no CPU fallback" % t.device)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise W2CError("tensors on different devices: %s vs %s" % (dev, t.device))
        if not t.is_contiguous():
            raise W2CError("non-contiguous tensor passed to a w2c op")
    return dev


def _stream(dev):
    rec = getattr(_tls, "recorder", None)
    if rec is not None:
        rec.dirty = True                            # this capture window holds at least one kernel node (ops.record_program)
    return torch.cuda.current_stream(dev).cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


_tls = threading.local()          # per-thread (DataParallel runs one thread per replica): the opt-in conv timer, the lanes recorder


class _Arena:
    """Outputs of an eagerly replayed region of a recorded program (Recorder.eager_static): the region's kernels are issued by the host at
    every replay, but what they write is read by CAPTURED graphs, so their output tensors must keep their addresses -- the first run
    allocates them (mode "record"), every later run hands the same tensors out again in allocation order (mode "replay")."""

    def __init__(self):
        self.tensors, self.mode, self.i = [], "record", 0

    def take(self, shape, dtype, device):
        if self.mode == "record":
            t = torch.empty(shape, dtype=dtype, device=device)
            self.tensors.append(t)
            return t
        if self.i >= len(self.tensors):
            raise W2CError("eager region of a recorded program allocates more outputs at replay than when it was recorded")
        t = self.tensors[self.i]
        self.i += 1
        if tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            raise W2CError("eager region of a recorded program: output %d changed shape / dtype between recording and replay" % (self.i - 1))
        return t


def _empty(shape, dtype, device):
    """torch.empty for a wrapper's OUTPUT tensor (through the active arena, if a recorded program is replaying an eager region)"""
    ar = getattr(_tls, "arena", None)
    if ar is not None:
        return ar.take(tuple(shape), dtype, device)
    return torch.empty(shape, dtype=dtype, device=device)


class SlotRef:
    """Stand-in for a caller-owned tensor whose ADDRESS a kernel reads from a device-resident pointer slot when it starts
    (include/w2c_hip.h "indirect operands": the argument is the slot's address with bit 0 set).  Carries the shape / dtype / device the
    wrappers check; `like` is a tensor (or SlotRef) of the geometry every later target will have."""
    is_cuda = True

    def __init__(self, slots, index, like):
        if slots.dtype != torch.int64 or not slots.is_cuda or index < 0 or index >= slots.numel():
            raise W2CError("SlotRef: slots must be an int64 device tensor and index inside it")
        self.slots, self.index = slots, index
        self.shape, self.dtype, self.device = tuple(like.shape), like.dtype, slots.device

    def data_ptr(self):
        return (self.slots.data_ptr() + 8 * self.index) | 1

    def dim(self):
        return len(self.shape)

    def numel(self):
        n = 1
        for d in self.shape:
            n *= d
        return n

    def is_contiguous(self):
        return True


# The collector is process-global and capture windows may be open on several threads (thread-local capture mode, DataParallel replicas):
# it is switched off when the FIRST window opens and back on (if it was on) when the LAST one closes (two threads toggling it
# independently re-enabled it inside the other's window).
_gc_lock = threading.Lock()
_gc_depth = 0
_gc_was_on = False


def _gc_hold():
    import gc
    global _gc_depth, _gc_was_on
    with _gc_lock:
        if _gc_depth == 0:
            _gc_was_on = gc.isenabled()
            gc.disable()
        _gc_depth += 1


def _gc_release():
    import gc
    global _gc_depth
    with _gc_lock:
        _gc_depth -= 1
        if _gc_depth == 0 and _gc_was_on:
            gc.enable()


class capture:
    """`with ops.capture() as graph:` -- capture the launches of the block into a fresh torch.cuda.CUDAGraph (thread-local error mode) with
    Python's cyclic garbage collector held off for the length of the window.

    Why (round 5, the crash of GPUTEST_r04; tools/r05/gc_in_capture.py reproduces it in 12 lines): objects kept alive only by a reference
    cycle -- a model + engine + its graphs inside a pytest.raises traceback, a closure -- are freed whenever the cyclic collector happens to
    run, i.e. at an arbitrary allocation.  If that is inside a capture window and the garbage holds an OLD CUDAGraph, its destructor
    (at::cuda::CUDAGraph::~CUDAGraph: hipGraphExecDestroy, pool release, and on ROCm a hipDeviceSynchronize) runs in the middle of the
    capture: the runtime refuses the synchronize while a stream is capturing, the check throws inside a destructor and the process dies
    (native frame: at::cuda::CUDAGraph::~CUDAGraph -> c10::hip::c10_hip_check_implementation -> std::terminate;
    profiles/r05_capture_crash.txt).  PyTorch used to run gc.collect() before every capture; 2.10 does so only on request.  So: collect
    BEFORE the window (old graphs die outside it), disable the collector INSIDE it (nothing is finalised between begin and end capture),
    restore it after.  Every capture of this package goes through here."""

    def __init__(self, graph=None, thread_local=True):
        self.graph = graph if graph is not None else torch.cuda.CUDAGraph()
        self._ctx = torch.cuda.graph(self.graph, capture_error_mode="thread_local") if thread_local else torch.cuda.graph(self.graph)

    def __enter__(self):
        import gc
        gc.collect()
        torch.cuda.synchronize()
        _gc_hold()
        try:
            self._ctx.__enter__()
        except BaseException:
            _gc_release()
            raise
        return self.graph

    def __exit__(self, et, ev, tb):
        try:
            return self._ctx.__exit__(et, ev, tb)
        finally:
            _gc_release()


# ---------------------------------------------------------------------------------------------------------------------------------
# Lanes: how this package expresses concurrency on one device (round 6).
#
# A forward has at most a handful of independent launch chains (the value trunk || the policy trunk and its tail; the five encoders of
# the single-request models).  Engine code never touches torch streams directly: it asks for the device's lanes and says
#     with L.on(1, after=(0,)): ...      run this block on lane 1, behind everything lane 0 has been given so far
#     tok = L.mark()                     a point in the current lane's order
#     L.wait(tok)                        the current lane goes on only behind that point
#     L.join(1)                          the current lane goes on only behind everything lane 1 has been given
#     L.eager(fn)                        something that must be issued by the host every time (an RCCL collective): never captured
#     L.eager_static(fn)                 a few launches kept out of the graphs (issued by the host at every replay) whose OUTPUTS captured
#                                        graphs read: they keep their addresses (ops._Arena)
# Lane 0 is the caller's stream, lanes 1.. are long-lived side streams.  Two executors implement the same five verbs:
#   * EagerLanes  -- stream / event calls, the launches go out as the Python code runs (model.use_hip_graph = False);
#   * a recorder  -- record_program(): the SAME Python code is run once under capture and cut, at exactly those verbs, into a list of
#     SINGLE-BRANCH HIP graphs (one linear chain of kernel nodes each) plus the event edges between them; Program.replay() launches
#     them on the lanes' streams, edges as hipEventRecord / hipStreamWaitEvent between the launches.
# Why not one graph with parallel branches (rounds 3-5): which streams the runtime runs the branches of a multi-branch exec on is the
# runtime's choice per instantiation -- hipGraphLaunch of the HIP runtime bundled with torch 2.10 + rocm7.0 reads out of bounds while
# choosing (profiles/r05_capture_crash.txt), and a bad draw puts both trunk chains on one hardware pipe for the exec's whole life
# (1.7x the step time; rounds 4-5 auditioned four instantiations per shape and kept the fastest, under GPU_MAX_HW_QUEUES=16).  A
# single-branch exec has max_streams == 1: the stream-selection loop has nothing to choose, and the lanes are streams this package
# created once -- no audition, no environment variable, no burst of exec-stream destructions.
_MAX_LANES = 4


def _new_side_stream(dev):
    # default priority.  (Measured, round 6: the value lane or the policy lane on a high-priority stream -- the device offers two levels --
    # makes the forward 4-6 % slower, 1.087 -> 1.127 / 1.151 ms on the same box; round 3 saw the same under graph capture.)
    return torch.cuda.Stream(device=dev)


def _lane_stream(dev, k):
    """long-lived side stream `k` of this thread on `dev`: lanes 1.. of the eager executor AND of every Program this thread records (Program.__init__ says why
    they are shared), the recorder's capture streams, the warm-up stream"""
    st = _tls.__dict__.setdefault("lane_streams", {})
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device(), k)
    s = st.get(key)
    if s is None:
        s = st[key] = _new_side_stream(dev)
    return s


class EagerLanes:
    """The verbs above as plain stream / event calls.  Stateless apart from lane 0 = the stream that was current when it was made, so a
    nested helper may ask ops.lanes(dev) again: mark / wait / join always act on torch's CURRENT stream."""
    recording = False

    def __init__(self, dev):
        self.dev = dev
        self.main = torch.cuda.current_stream(dev)

    def stream(self, k):
        return self.main if k == 0 else _lane_stream(self.dev, k)

    def on(self, k, after=()):
        s = self.stream(k)
        for j in after:
            sj = self.stream(j)
            if sj != s:
                s.wait_stream(sj)
        return torch.cuda.stream(s)

    def mark(self):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        return ev

    def wait(self, tok):
        torch.cuda.current_stream(self.dev).wait_event(tok)

    def join(self, k):
        cur, s = torch.cuda.current_stream(self.dev), self.stream(k)
        if s != cur:
            cur.wait_stream(s)

    def eager(self, fn):
        return fn()

    def eager_static(self, fn):
        return fn()


class _LaneCtx:
    def __init__(self, rec, k, after):
        self.rec, self.k, self.after = rec, k, tuple(after)

    def __enter__(self):
        r = self.rec
        self.prev = r._lane
        r._close()
        for j in self.after:
            if j != self.k:
                r.prog.append(("sync", self.k, j, torch.cuda.Event()))
        r._open(self.k)

    def __exit__(self, et, ev, tb):
        r = self.rec
        r._close(abort=et is not None)
        if et is None:
            r._open(self.prev)
        return False


class _CapStreams:
    def __init__(self, dev):
        self.dev = dev

    def __getitem__(self, lane):
        # lane 0 is captured on one extra stream ("cap0", also the warm-up stream), a side lane on that lane's own long-lived stream: the
        # package makes TWO streams per thread and device in all.  The runtime spreads streams over four hardware queues in creation order;
        # every further stream of ours is one more chance that a caller's stream shares a queue with the value lane, and then the two
        # chains of a forward run one behind the other (bench.py's forwards_in_flight, two engines on two caller streams: 1.10 ms per
        # forward with three capture streams made ahead of the callers', 0.92 with the callers' streams made first)
        return _lane_stream(self.dev, "cap0" if lane == 0 else lane)


class _Recorder:
    """record_program()'s executor: one capture window open at any time, on the current lane; every verb closes it, notes the edge and
    opens the next one.  Each window becomes its own CUDAGraph with its own private memory pool (blocks freed inside a window are
    reused inside that window only -- windows on different lanes run concurrently at replay); tensors handed from one window to a
    later one stay alive because the Python code holds them until the recording ends and the Program keeps the result."""
    recording = True

    def __init__(self, dev):
        self.dev = dev
        self.prog = []
        self.cap = _CapStreams(dev)                 # capture-time streams: long-lived, per thread, made when a lane is first used (every
                                                    # stream ever made keeps its share of one of the runtime's four hardware queues)
        self._lane, self._g, self._sctx = 0, None, None
        self.dirty = False
        self.n_events = 0

    def _open(self, lane):
        if lane >= _MAX_LANES:
            raise W2CError("lane %d: at most %d lanes" % (lane, _MAX_LANES))
        self._lane = lane
        self._sctx = torch.cuda.stream(self.cap[lane])
        self._sctx.__enter__()
        self._g = torch.cuda.CUDAGraph()
        self.dirty = False
        try:
            self._g.capture_begin(capture_error_mode="thread_local")
        except BaseException:
            self._sctx.__exit__(None, None, None)
            self._g = self._sctx = None
            raise

    def _close(self, abort=False):
        if self._g is None:
            return
        g, self._g = self._g, None
        import warnings
        empty = False
        try:
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                g.capture_end()
            empty = any("empty" in str(w.message).lower() for w in caught)
        finally:
            self._sctx.__exit__(None, None, None)
            self._sctx = None
        if not abort and (self.dirty or not empty):
            self.prog.append(("graph", self._lane, g))

    # ---- the five verbs ----
    def on(self, k, after=()):
        return _LaneCtx(self, k, after)

    def mark(self):
        lane = self._lane
        self._close()
        ev = torch.cuda.Event()
        self.prog.append(("record", lane, ev))
        self._open(lane)
        return ev

    def wait(self, tok):
        lane = self._lane
        self._close()
        self.prog.append(("wait", lane, tok))
        self._open(lane)

    def join(self, k):
        lane = self._lane
        if k == lane:
            return
        self._close()
        self.prog.append(("sync", lane, k, torch.cuda.Event()))
        self._open(lane)

    def eager(self, fn):
        lane = self._lane
        self._close()
        self.prog.append(("call", lane, fn))
        with torch.cuda.stream(self.cap[lane]):
            res = fn()                              # (runs on unwritten buffers here: captured work has not executed)
        self._open(lane)
        return res

    def eager_static(self, fn):
        """fn's launches are issued by the host at every replay (no graph, hence no graph-launch boundary in front of and behind them:
        ~8-10 us each on this runtime, see the measurements in engine.CommEngine._record), but its outputs feed captured graphs: they
        come from an arena that hands the SAME tensors out at every replay.  Returns fn's result (those tensors)."""
        lane = self._lane
        self._close()
        arena = _Arena()
        box = {}

        def run():
            prev = getattr(_tls, "arena", None)
            _tls.arena = arena
            arena.i = 0
            try:
                box["res"] = fn()
            finally:
                _tls.arena = prev
            if arena.mode == "replay" and arena.i != len(arena.tensors):
                raise W2CError("eager region of a recorded program allocated fewer outputs at replay than when it was recorded")

        with torch.cuda.stream(self.cap[lane]):
            run()
        arena.mode = "replay"
        res = box["res"]
        self.prog.append(("call", lane, run))
        self._open(lane)
        return res


def streams_share_queue(a, b, spin_cycles=300000):
    """Do torch streams a and b of one device run on the same hardware queue?  (The runtime multiplexes streams on four; two streams of
    one queue run their work one behind the other -- a caller's stream on the value lane's queue turns the forward's two chains into
    one.)  Probe: a ~150 us spin on one stream, a trivial kernel on the other issued right behind it; if that kernel ends only after the
    spin, they share.  Both directions.  Synchronises the device; never call it inside a capture.  False when torch has no spin kernel."""
    spin = getattr(torch.cuda, "_sleep", None)
    if spin is None or a == b:
        return a == b
    x = torch.zeros(64, device=a.device)
    torch.cuda.synchronize(a.device)

    def held(p, q):
        e0, e1, eq = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        with torch.cuda.stream(p):
            e0.record()
            spin(int(spin_cycles))
            e1.record()
        with torch.cuda.stream(q):
            x.add_(1.0)
            eq.record()
        torch.cuda.synchronize(a.device)
        return e0.elapsed_time(eq) > 0.7 * e0.elapsed_time(e1)
    held(a, b)                                      # (first use of a stream: not timed)
    return held(a, b) or held(b, a)


def caller_streams(dev, n, tries=12):
    """n new streams for a loop that keeps n engines in flight: mutually on different hardware queues, and none on the queue of this
    thread's value lane (lane 1), as far as `tries` candidates allow (the rest are taken as they come).  The rejected candidates stay
    alive with the returned list's first element (a destroyed stream would hand its queue slot to the next one made)."""
    dev = torch.device(dev)
    lane = _lane_stream(dev, 1)
    chosen, rejected = [], []
    for _ in range(tries):
        if len(chosen) == n:
            break
        c = torch.cuda.Stream(device=dev)
        if any(streams_share_queue(c, o) for o in [lane] + chosen):
            rejected.append(c)
        else:
            chosen.append(c)
    while len(chosen) < n:
        chosen.append(rejected.pop() if rejected else torch.cuda.Stream(device=dev))
    if chosen:
        chosen[0]._w2c_keepalive = rejected
    return chosen


def lanes(dev):
    """the lanes executor in effect on this thread: the recorder inside record_program(), else stream / event calls"""
    rec = getattr(_tls, "recorder", None)
    if rec is not None:
        if dev.index is not None and rec.dev.index is not None and dev.index != rec.dev.index:
            raise W2CError("record_program: launches on %s inside a recording for %s" % (dev, rec.dev))
        return rec
    return EagerLanes(dev)


class Program:
    """A recorded forward: single-branch graphs + the edges between them (see the block comment above).  replay() issues it on the
    caller's current stream (lane 0) and the recording thread's lane streams; `result` is what the recorded function returned (tensors in
    the graphs' private pools: static across replays)."""

    def __init__(self, dev, prog, result):
        self.dev, self.prog, self.result = dev, prog, result
        n_side = max([st[1] for st in prog] + [st[2] for st in prog if st[0] == "sync"] + [0])
        # the side lanes are the recording thread's long-lived lane streams, SHARED by every Program it records (and by its eager forwards):
        # a process that keeps several forwards in flight (one engine per caller stream, bench.py's forwards_in_flight) then runs on
        # callers + 1 streams instead of 2 per engine -- beyond the runtime's four hardware queues streams share a pipe and one engine's
        # chains queue behind another's (3 engines in flight, ms per forward, same box: own side streams 1.097 / 1.102, shared 0.992 /
        # 0.993; one forward at a time 0.986 either way).  Sharing a lane orders the value chains of consecutive forwards one behind the
        # other, which they are anyway; no cycle can form (a lane's work waits for caller-stream events only, never the reverse before
        # its own join)
        self.side = [_lane_stream(dev, k + 1) for k in range(n_side)]
        self.n_graphs = sum(1 for st in prog if st[0] == "graph")
        self.n_calls = sum(1 for st in prog if st[0] == "call")

    def replay(self):
        s0 = torch.cuda.current_stream(self.dev)
        S = [s0] + self.side
        for st in self.prog:
            kind = st[0]
            if kind == "graph":
                if st[1] == 0:
                    st[2].replay()
                else:
                    with torch.cuda.stream(S[st[1]]):
                        st[2].replay()
            elif kind == "sync":                    # lane st[1] goes on behind everything lane st[2] has been given
                st[3].record(S[st[2]])
                S[st[1]].wait_event(st[3])
            elif kind == "record":
                st[2].record(S[st[1]])
            elif kind == "wait":
                S[st[1]].wait_event(st[2])
            else:                                   # "call": issued by the host at every replay (collectives)
                with torch.cuda.stream(S[st[1]]):
                    st[2]()
        return self.result


def record_program(dev, fn, warmup=2, before_warmup=None):
    """Run fn() `warmup` times eagerly on a side stream (function attributes, lazily built plans, the allocator), then once more under
    the recorder -> Program.  fn must be idempotent and may use the lanes verbs; every tensor it allocates lives in the Program."""
    import gc
    if getattr(_tls, "recorder", None) is not None:
        raise W2CError("record_program inside a recording")
    if warmup > 0:
        side = _lane_stream(dev, "cap0")              # (nothing is being captured during the warm-up forwards)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            if before_warmup is not None:
                before_warmup()
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream(dev).wait_stream(side)
    # collect BEFORE the windows and hold the collector off INSIDE them: a CUDAGraph finalised inside a capture window kills the process
    # (see ops.capture)
    gc.collect()
    torch.cuda.synchronize(dev)
    _gc_hold()
    rec = _Recorder(dev)
    _tls.recorder = rec
    ok = False
    try:
        rec._open(0)
        res = fn()
        rec._close()
        ok = True
    finally:
        _tls.recorder = None
        if not ok:
            try:
                rec._close(abort=True)
            except Exception:                       # noqa: BLE001
                pass
        _gc_release()
    return Program(dev, rec.prog, res)


def set_slots(slots, tensors):
    """w2c_set_slots: slots[i] = address of tensors[i] (None -> 0), in stream order on the current stream."""
    import ctypes
    dev = slots.device
    n = len(tensors)
    vals = (ctypes.c_void_p * n)(*[(_p(t) or None) for t in tensors])
    with torch.cuda.device(dev):
        check(_native.lib().w2c_set_slots(_p(slots), n, ctypes.cast(vals, ctypes.c_void_p), _stream(dev)), "w2c_set_slots")


def copy_to_slot(src, dst):
    """w2c_copy_to_slot: src (contiguous device tensor, nbytes % 4 == 0) -> dst (tensor of the same byte size, or a SlotRef)."""
    dev = _need_gpu(src)
    nbytes = src.numel() * src.element_size()
    with torch.cuda.device(dev):
        check(_native.lib().w2c_copy_to_slot(_p(src), nbytes, _p(dst), _stream(dev)), "w2c_copy_to_slot")


class KernelTimer:
    """Opt-in per-launch timing of the dominant kernel (w2c_conv_igemm_bf16) with HIP events on
    the launch stream (torch.cuda.Event on torch's current stream == the stream we launch on).
    Used by bench.py's roofline attribution pass only; never active inside its timed region."""

    def __init__(self, spans=False, device=None, capacity=4096):
        self.records = []          # (start_event | span slot, end_event | None, flops, key, algorithmic bytes)
        self.spans = bool(spans)
        if self.spans:
            # span mode (bench.py under HIP-graph replay): every conv launch carries a pointer to its own slot of this buffer and its
            # workgroups record min(start) / max(end) of the 100 MHz wall clock there -- on every replay of the captured graph
            # (include/w2c_hip.h w2c_debug_conv_span).  reset() before a replay, read after a sync.
            self.buf = torch.empty((capacity, 2), dtype=torch.int64, device=device)
            self.reset()

    def reset(self):
        self.buf[:, 0] = torch.iinfo(torch.int64).max       # (stamps are far below 2^63: signed and unsigned order agree)
        self.buf[:, 1] = 0

    def begin(self, dev):
        if self.spans:
            idx = len(self.records)
            if idx >= self.buf.shape[0]:
                raise W2CError("KernelTimer: span buffer full")
            check(_native.lib().w2c_debug_conv_span(self.buf.data_ptr() + 16 * idx), "w2c_debug_conv_span")
            return idx
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record(torch.cuda.current_stream(dev))
        return ev0

    def end(self, token, dev, flops, key, nbytes):
        if self.spans:
            self.records.append((token, None, flops, key, nbytes))
            return
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record(torch.cuda.current_stream(dev))
        self.records.append((token, ev1, flops, key, nbytes))

    def _intervals(self):
        """[(start_ms, end_ms, record)] of the launches that ran since the last reset (span mode) / of all records (event mode)"""
        if self.spans:
            t = self.buf[:len(self.records)].cpu().numpy()
            live = [(int(t[i, 0]), int(t[i, 1]), r) for i, r in enumerate(self.records) if t[i, 1] > 0]
            if not live:
                return []
            t0 = min(a for a, _, _ in live)
            return [((a - t0) * 1e-5, (b - t0) * 1e-5, r) for a, b, r in live]           # 100 MHz ticks -> ms
        base = self.records[0][0]
        return [(base.elapsed_time(r[0]), base.elapsed_time(r[1]), r) for r in self.records]

    def algorithmic_bytes(self):
        """sum over the measured launches of input + output (+ residual) activations + weights, each once."""
        return float(sum(r[4] for _, _, r in self._intervals()))

    def summary(self):
        """-> total_ms, total_flops, launches, per-shape {key: [ms, flops, count]} (call after a sync)."""
        total_ms, total_fl, per, n = 0.0, 0.0, {}, 0
        for a, b, r in self._intervals():
            ms, fl, key = b - a, r[2], r[3]
            total_ms += ms
            total_fl += fl
            n += 1
            e = per.setdefault(key, [0.0, 0.0, 0])
            e[0] += ms
            e[1] += fl
            e[2] += 1
        return total_ms, total_fl, n, per

    def busy_ms(self):
        """Length of the UNION of the measured launches' [start, end] intervals (ms; call after a sync).  The two trunks run as two
        concurrent launch chains on two streams (engine.TrunkPlan.after_stem): launches that share the chip each take longer than
        alone, and the sum of their durations counts that wall time twice.  The union is the time the chip spent in the family."""
        spans = sorted((a, b) for a, b, _ in self._intervals())
        if not spans:
            return 0.0
        busy, lo, hi = 0.0, spans[0][0], spans[0][1]
        for a, b in spans[1:]:
            if a > hi:
                busy += hi - lo
                lo, hi = a, b
            else:
                hi = max(hi, b)
        return busy + (hi - lo)




def set_conv_timer(timer):
    _tls.conv_timer = timer


_zero_pages = {}


def zero_page(dev):
    key = (dev.type, dev.index)
    z = _zero_pages.get(key)
    if z is None:
        z = torch.zeros(256, dtype=torch.uint8, device=dev)
        _zero_pages[key] = z
    return z


def stem_conv7x7_bn_relu(x, n_agents, w_packed, scale, shift, out=None):
    """x f32 [B,3N,H,W] -> bf16 NHWC [N*B, H/2, W/2, Cout] (agent-major)."""
    dev = _need_gpu(x, w_packed, scale, shift, out)
    B, c3n, H, W = x.shape
    if c3n != 3 * n_agents or x.dtype != torch.float32:
        raise W2CError("stem: expected f32 [B, 3*%d, H, W], got %s %s" % (n_agents, tuple(x.shape), x.dtype))
    cout = scale.numel()
    if out is None:
        out = _empty((n_agents * B, H // 2, W // 2, cout), dtype=BF16, device=dev)
    with torch.cuda.device(dev):
        check(_native.lib().w2c_stem_conv7x7_bn_relu(_p(x), B, n_agents, H, W, _p(w_packed), _p(scale), _p(shift),
                                                     cout, _p(out), _stream(dev)), "w2c_stem_conv7x7_bn_relu")
    return out


def stem_conv7x7_bn_relu_maxpool(x, n_agents, w_packed, scale, shift, out=None):
    """x f32 [B,3N,H,W] -> bf16 NHWC [N*B, H/4, W/4, Cout]: stem + maxpool 3x3/2 fused."""
    dev = _need_gpu(x, w_packed, scale, shift, out)
    B, c3n, H, W = x.shape
    if c3n != 3 * n_agents or x.dtype != torch.float32:
        raise W2CError("stem: expected f32 [B, 3*%d, H, W], got %s %s" % (n_agents, tuple(x.shape), x.dtype))
    cout = scale.numel()
    if out is None:
        out = _empty((n_agents * B, H // 4, W // 4, cout), dtype=BF16, device=dev)
    with torch.cuda.device(dev):
        check(_native.lib().w2c_stem_conv7x7_bn_relu_maxpool(_p(x), B, n_agents, H, W, _p(w_packed), _p(scale),
                                                             _p(shift), cout, _p(out), _stream(dev)),
              "w2c_stem_conv7x7_bn_relu_maxpool")
    return out


FRAME_MEAN_BGR = (103.939, 116.779, 123.68)          # airsimLoader.mean (airsim_loader.py)


def stem_u8_conv7x7_bn_relu_maxpool(frames, w_packed, scale, shift, mean_bgr=FRAME_MEAN_BGR, out=None):
    """frames u8 RGB [B,N,H,W,3] -> bf16 NHWC [N*B, H/4, W/4, Cout]; loader transform fused (see the C header)."""
    dev = _need_gpu(frames, w_packed, scale, shift, out)
    if frames.dtype != torch.uint8 or frames.dim() != 5 or frames.shape[4] != 3:
        raise W2CError("stem_u8: expected u8 [B, N, H, W, 3], got %s %s" % (tuple(frames.shape), frames.dtype))
    B, N, H, W, _ = frames.shape
    cout = scale.numel()
    if out is None:
        out = _empty((N * B, H // 4, W // 4, cout), dtype=BF16, device=dev)
    with torch.cuda.device(dev):
        check(_native.lib().w2c_stem_u8_conv7x7_bn_relu_maxpool(_p(frames), float(mean_bgr[0]), float(mean_bgr[1]),
                                                                float(mean_bgr[2]), B, N, H, W, _p(w_packed), _p(scale),
                                                                _p(shift), cout, _p(out), _stream(dev)),
              "w2c_stem_u8_conv7x7_bn_relu_maxpool")
    return out


def maxpool3x3s2(x, out=None):
    dev = _need_gpu(x, out)
    M, H, W, C = x.shape
    if out is None:
        out = _empty((M, H // 2, W // 2, C), dtype=BF16, device=dev)
    with torch.cuda.device(dev):
        check(_native.lib().w2c_maxpool3x3s2(_p(x), M, H, W, C, _p(out), _stream(dev)), "w2c_maxpool3x3s2")
    return out


_splitk_ws = {}
_SPLITK_WS_MAX = 16            # (device, stream) entries kept; the least recently used one goes first


def splitk_workspace(dev, nbytes):
    """Split-K scratch, one per (device, stream): launches that share a workspace are ordered by the stream they run on,
    so concurrent streams / DataParallel replica threads / a graph under capture never share one.  Contents are
    irrelevant to the kernel (every partial tile is written before it is read).  The cache is BOUNDED (a process that keeps
    creating streams would otherwise pin 16 MiB per stream forever): least recently used entries are dropped -- the caching
    allocator keeps a dropped block alive until the launches already queued on its stream have run, and a captured graph holds
    its own reference to the block it was captured with."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), _stream(dev))
    ws = _splitk_ws.pop(key, None)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 16 << 20), dtype=torch.uint8, device=dev)
    _splitk_ws[key] = ws                                   # (re)insert as most recently used
    while len(_splitk_ws) > _SPLITK_WS_MAX:
        _splitk_ws.pop(next(iter(_splitk_ws)))
    return ws


def pack_wfrag(w_packed, cin):
    """[G, Cout, 9*cin] bf16 weights (K = (tap, cin), the layout every other conv kernel reads) -> the MFMA A-fragment order of
    the weights-to-registers 3x3 kernels (csrc/conv_wreg.inl): [G][Cout/32][chunk*9 + tap][k slice 0..3][half][channel % 32][8],
    i.e. one contiguous 1 KB block per (32 channels, K-step, 16-deep k slice) in lane order.  Same bytes, permuted."""
    G, cout, K = w_packed.shape
    if K != 9 * cin or cin % 64 or cout % 32:
        raise W2CError("pack_wfrag: needs a 3x3 conv with cin % 64 == 0 and cout % 32 == 0")
    nch = cin // 64
    v = w_packed.reshape(G, cout // 32, 32, 9, nch, 4, 2, 8)          # g, nb, c32, tap, cc, kk, half, e
    return v.permute(0, 1, 4, 3, 5, 6, 2, 7).contiguous().reshape(G, cout, K)


_MAX_X_BYTES = (1 << 31) - 1       # the conv kernels address x through a 32-bit buffer descriptor (include/w2c_hip.h)


def conv_igemm(x, x_ch_off, cin, w_packed, cout, ksize, stride, groups, scale, shift, residual=None, relu=True,
               out=None, out_f32=False, out_cstride=None, variant=None, ksplit=None, out_groups=None, _gstride=0,
               out_ch_off=0):
    """x: bf16 NHWC [M,H,W,xcs]; the conv reads channels [x_ch_off + g*cin, ...).  Returns/accepts
    out NHWC [M,Ho,Wo,out_cstride] (bf16, or f32 when out_f32).  ksplit: None = one workgroup per tile;
    0 = split-K chosen by the library for tail layers; n = forced n-way split.
    Batches whose activation tensor reaches 2 GiB are run as several launches over slices of M (results are
    independent of the image count, so the slicing is invisible).
    out_groups: one NHWC [M,Ho,Wo,cs] tensor PER GROUP (same shape, dtype and cs >= cout) instead of one side-by-side
    tensor -- e.g. the value trunk's squeezer writing V straight into its slot of an all-gather buffer; returns them."""
    if out_groups is not None:
        if out is not None or residual is not None or len(out_groups) != groups:
            raise W2CError("conv: out_groups excludes out/residual and needs one tensor per group")
        g0 = out_groups[0]
        _need_gpu(*out_groups)
        esz = g0.element_size()
        gstride = 0
        for i, t in enumerate(out_groups):
            if t.shape != g0.shape or t.dtype != g0.dtype or g0.shape[3] < cout:
                raise W2CError("conv: out_groups tensors must share shape/dtype with >= cout channels")
            d = t.data_ptr() - g0.data_ptr()
            if i == 1:
                gstride = d // esz
            if d != i * gstride * esz or d % 16:
                raise W2CError("conv: out_groups tensors must be evenly spaced, 16-byte aligned")
        if gstride == 0 and groups > 1:
            raise W2CError("conv: out_groups tensors alias")
        conv_igemm(x, x_ch_off, cin, w_packed, cout, ksize, stride, groups, scale, shift, relu=relu, out=g0,
                   out_f32=out_f32, out_cstride=g0.shape[3], variant=variant, ksplit=ksplit, out_groups=None,
                   _gstride=gstride if groups > 1 else 0)
        return list(out_groups)
    dev = _need_gpu(x, w_packed, scale, shift, residual, out)
    M, H, W, xcs = x.shape
    pad = 1 if ksize == 3 else 0
    Ho = (H + 2 * pad - ksize) // stride + 1
    Wo = (W + 2 * pad - ksize) // stride + 1
    if out_cstride is None:
        out_cstride = out.shape[3] if (out is not None and out.dim() == 4) else groups * cout
    if out_ch_off:                                   # write channels [out_ch_off, +groups*cout) of a wider `out` tensor
        if out is None or residual is not None or out_ch_off < 0 or out_ch_off % 8 or out_ch_off + groups * cout > out.shape[3]:
            raise W2CError("conv: out_ch_off needs an `out` tensor wide enough, no residual, a multiple of 8")
        out_cstride = out.shape[3]
    if out_cstride < (groups if not _gstride else 1) * cout:
        raise W2CError("conv: out_cstride %d too small for %d x %d output channels" % (out_cstride, groups, cout))
    if x_ch_off < 0 or x_ch_off + groups * cin > xcs:
        raise W2CError("conv: channels [%d, %d) outside the tensor's %d channels" % (x_ch_off, x_ch_off + groups * cin, xcs))
    if out is None:
        out = _empty((M, Ho, Wo, out_cstride), dtype=torch.float32 if out_f32 else BF16, device=dev)
    if tuple(out.shape) != (M, Ho, Wo, out_cstride):
        raise W2CError("conv: bad out shape %s, want %s" % (tuple(out.shape), (M, Ho, Wo, out_cstride)))
    if residual is not None and tuple(residual.shape) != tuple(out.shape):
        raise W2CError("conv: residual geometry must equal the output geometry")
    per_img = max(H * W * xcs * 2, Ho * Wo * out_cstride * 2)
    if M * per_img > _MAX_X_BYTES:
        step = max(1, _MAX_X_BYTES // per_img)
        for lo in range(0, M, step):
            hi = min(M, lo + step)
            conv_igemm(x[lo:hi], x_ch_off, cin, w_packed, cout, ksize, stride, groups, scale, shift,
                       residual=None if residual is None else residual[lo:hi], relu=relu, out=out[lo:hi],
                       out_f32=out_f32, out_cstride=out_cstride, variant=variant, ksplit=ksplit, _gstride=_gstride,
                       out_ch_off=out_ch_off)
        return out
    xptr = x.data_ptr() + 2 * x_ch_off
    optr = out.data_ptr() + (4 if out_f32 else 2) * out_ch_off
    timer = getattr(_tls, "conv_timer", None)
    tok = timer.begin(dev) if timer is not None else None
    with torch.cuda.device(dev):
        ws_bytes = 0
        if ksplit is not None and variant is None:
            ws_bytes = _native.lib().w2c_conv_splitk_workspace_bytes(M, H, W, cin, cout, ksize, stride, groups, int(ksplit))
            if ws_bytes < 0:
                raise W2CError("conv split-K: unsupported shape")
        if ws_bytes > 0:
            ws = splitk_workspace(dev, ws_bytes)
            check(_native.lib().w2c_conv_igemm_bf16_splitk(xptr, M, H, W, cin, xcs, _p(w_packed), cout, ksize, stride,
                                                           groups, _p(scale), _p(shift), _p(residual),
                                                           1 if relu else 0, optr, out_cstride,
                                                           1 if out_f32 else 0, _p(zero_page(dev)), int(ksplit),
                                                           _p(ws), ws.numel(), int(_gstride), _stream(dev)),
                  "w2c_conv_igemm_bf16_splitk")
        elif variant is None:
            check(_native.lib().w2c_conv_igemm_bf16(xptr, M, H, W, cin, xcs, _p(w_packed), cout, ksize, stride, groups,
                                                    _p(scale), _p(shift), _p(residual), 1 if relu else 0,
                                                    optr, out_cstride, 1 if out_f32 else 0,
                                                    _p(zero_page(dev)), int(_gstride), _stream(dev)), "w2c_conv_igemm_bf16")
        else:
            check(_native.lib().w2c_conv_igemm_bf16_variant(xptr, M, H, W, cin, xcs, _p(w_packed), cout, ksize, stride,
                                                            groups, _p(scale), _p(shift), _p(residual),
                                                            1 if relu else 0, optr, out_cstride,
                                                            1 if out_f32 else 0, _p(zero_page(dev)), int(variant),
                                                            int(_gstride), _stream(dev)), "w2c_conv_igemm_bf16_variant(%d)" % variant)
    if timer is not None:
        flops = 2.0 * M * Ho * Wo * cout * (ksize * ksize * cin) * groups
        nbytes = (M * H * W * cin * groups * 2 + M * Ho * Wo * cout * groups * (4 if out_f32 else 2)
                  + (M * Ho * Wo * cout * groups * 2 if residual is not None else 0) + groups * cout * ksize * ksize * cin * 2)
        timer.end(tok, dev, flops, (M * Ho * Wo, cin, cout, ksize, stride, groups), nbytes)
    return out


def pack_wfrag_device(w_packed, cin):
    """w2c_pack_wfrag_bf16: the library's own (device-side) form of pack_wfrag."""
    dev = _need_gpu(w_packed)
    G, cout, K = w_packed.shape
    if K != 9 * cin:
        raise W2CError("pack_wfrag: needs 3x3 weights [G, Cout, 9*cin]")
    out = torch.empty_like(w_packed)
    with torch.cuda.device(dev):
        check(_native.lib().w2c_pack_wfrag_bf16(_p(w_packed), _p(out), G, cout, cin, _stream(dev)), "w2c_pack_wfrag_bf16")
    return out


def conv3x3_wreg_supported(H, W, cin, cout):
    return bool(_native.lib().w2c_conv3x3_wreg_supported(int(H), int(W), int(cin), int(cout)))


def conv3x3_wreg_f32(x, x_ch_off, cin, wfrag, cout, scale, shift, relu=False, out=None, out_ch_off=0):
    """w2c_conv3x3_wreg_f32out: the weights-to-registers kernel (default form) writing an f32 NHWC tensor, one group, no residual
    (the decoder's first conv on the value maps, engine.DecoderPlan.value_maps).  out: f32 [M,H,W,>=out_ch_off+cout] or None."""
    dev = _need_gpu(x, wfrag, scale, shift, out)
    M, H, W, xcs = x.shape
    if out is None:
        out = _empty((M, H, W, cout), dtype=torch.float32, device=dev)
    ocs = out.shape[3]
    if tuple(out.shape[:3]) != (M, H, W) or out.dtype != torch.float32 or out_ch_off % 4 or out_ch_off + cout > ocs:
        raise W2CError("conv3x3_wreg_f32: out must be f32 [M,H,W,>= out_ch_off + cout]")
    if x_ch_off < 0 or x_ch_off + cin > xcs:
        raise W2CError("conv3x3_wreg_f32: channel window outside the tensor")
    per_img = max(H * W * xcs * 2, H * W * ocs * 4)
    if M * per_img > _MAX_X_BYTES:
        step = max(1, _MAX_X_BYTES // per_img)
        for lo in range(0, M, step):
            conv3x3_wreg_f32(x[lo:lo + step], x_ch_off, cin, wfrag, cout, scale, shift, relu, out[lo:lo + step], out_ch_off)
        return out
    timer = getattr(_tls, "conv_timer", None)
    tok = timer.begin(dev) if timer is not None else None
    with torch.cuda.device(dev):
        check(_native.lib().w2c_conv3x3_wreg_f32out(x.data_ptr() + 2 * x_ch_off, M, H, W, cin, xcs, _p(wfrag), cout, 1, _p(scale), _p(shift),
                                                    1 if relu else 0, out.data_ptr() + 4 * out_ch_off, ocs, 0, _stream(dev)),
              "w2c_conv3x3_wreg_f32out")
    if timer is not None:
        timer.end(tok, dev, 2.0 * M * H * W * cout * 9 * cin, (M * H * W, cin, cout, 3, 1, 1),
                  M * H * W * cin * 2 + M * H * W * cout * 4 + cout * 9 * cin * 2)
    return out


def conv3x3_wreg(x, x_ch_off, cin, wfrag, cout, groups, scale, shift, residual=None, relu=True, out=None, out_cstride=None,
                 form=0, out_ch_off=0, _gstride=0, out_groups=None):
    """w2c_conv3x3_wreg_bf16: 3x3 / stride 1 / pad 1, bf16 NHWC in and out; `wfrag` from pack_wfrag_device.  Same tensor
    conventions as conv_igemm (x_ch_off, out_ch_off, groups side by side unless _gstride; out_groups = one evenly spaced
    tensor per group, e.g. the squeezer writing V into its slot of the all-gather buffer)."""
    if out_groups is not None:
        if out is not None or residual is not None or len(out_groups) != groups:
            raise W2CError("conv3x3_wreg: out_groups excludes out/residual and needs one tensor per group")
        g0 = out_groups[0]
        _need_gpu(*out_groups)
        gstride = 0
        for i, t in enumerate(out_groups):
            if t.shape != g0.shape or t.dtype != BF16 or g0.shape[3] < cout:
                raise W2CError("conv3x3_wreg: out_groups tensors must share shape, be bf16 and have >= cout channels")
            d = t.data_ptr() - g0.data_ptr()
            if i == 1:
                gstride = d // 2
            if d != i * gstride * 2 or d % 16:
                raise W2CError("conv3x3_wreg: out_groups tensors must be evenly spaced, 16-byte aligned")
        if gstride == 0 and groups > 1:
            raise W2CError("conv3x3_wreg: out_groups tensors alias")
        conv3x3_wreg(x, x_ch_off, cin, wfrag, cout, groups, scale, shift, relu=relu, out=g0, out_cstride=g0.shape[3], form=form,
                     _gstride=gstride if groups > 1 else 0)
        return list(out_groups)
    dev = _need_gpu(x, wfrag, scale, shift, residual, out)
    M, H, W, xcs = x.shape
    if out_cstride is None:
        out_cstride = out.shape[3] if out is not None else groups * cout
    if out is None:
        out = _empty((M, H, W, out_cstride), dtype=BF16, device=dev)
    if tuple(out.shape) != (M, H, W, out_cstride) or out.dtype != BF16:
        raise W2CError("conv3x3_wreg: bad out tensor")
    if residual is not None and (tuple(residual.shape) != tuple(out.shape) or out_ch_off):
        raise W2CError("conv3x3_wreg: residual geometry must equal the output geometry")
    if x_ch_off < 0 or x_ch_off + groups * cin > xcs or out_ch_off % 8 or out_ch_off + (1 if _gstride else groups) * cout > out_cstride:
        raise W2CError("conv3x3_wreg: channel window outside the tensor")
    per_img = max(H * W * xcs * 2, H * W * out_cstride * 2)
    if M * per_img > _MAX_X_BYTES:
        step = max(1, _MAX_X_BYTES // per_img)
        for lo in range(0, M, step):
            hi = min(M, lo + step)
            conv3x3_wreg(x[lo:hi], x_ch_off, cin, wfrag, cout, groups, scale, shift, None if residual is None else residual[lo:hi],
                         relu, out[lo:hi], out_cstride, form, out_ch_off, _gstride)
        return out
    timer = getattr(_tls, "conv_timer", None)
    tok = timer.begin(dev) if timer is not None else None
    with torch.cuda.device(dev):
        check(_native.lib().w2c_conv3x3_wreg_bf16(x.data_ptr() + 2 * x_ch_off, M, H, W, cin, xcs, _p(wfrag), cout, groups,
                                                  _p(scale), _p(shift), _p(residual), 1 if relu else 0,
                                                  out.data_ptr() + 2 * out_ch_off, out_cstride, int(_gstride), int(form), _stream(dev)),
              "w2c_conv3x3_wreg_bf16")
    if timer is not None:
        flops = 2.0 * M * H * W * cout * 9 * cin * groups
        nbytes = (M * H * W * cin * groups * 2 + M * H * W * cout * groups * 2
                  + (M * H * W * cout * groups * 2 if residual is not None else 0) + groups * cout * 9 * cin * 2)
        timer.end(tok, dev, flops, (M * H * W, cin, cout, 3, 1, groups), nbytes)
    return out


FP8 = torch.float8_e4m3fn
FP8_MAX = 448.0


def conv_fp8(x, x_ch_off, cin, w_packed, cout, ksize, stride, groups, scale, shift, residual=None, relu=True,
             out_bf16=True, out_fp8_scale=None, out_groups=None, variant=-1, out=None, out_ch_off=0):
    """The conv with fp8 (e4m3) operands when x is a uint8 tensor [M,H,W,xcs] of e4m3 bytes (w_packed uint8
    [G,Cout,k*k*cin]), or with bf16 operands when x is bf16 -- either way able to emit a bf16 result, an fp8 result
    (e4m3(result / out_fp8_scale), uint8 [M,Ho,Wo,G*cout]) or both.  Returns (y_bf16 | None | list (out_groups),
    y_fp8 | None).  Scales: the caller folds weight / input scales into `scale` (include/w2c_hip.h)."""
    dev = _need_gpu(x, w_packed, scale, shift, residual)
    f8 = x.dtype == torch.uint8
    if not f8 and x.dtype != BF16:
        raise W2CError("conv_fp8: x must be uint8 (e4m3 bytes) or bf16")
    if w_packed.dtype != (torch.uint8 if f8 else BF16):
        raise W2CError("conv_fp8: weights must have the operand type of x")
    M, H, W, xcs = x.shape
    pad = 1 if ksize == 3 else 0
    Ho = (H + 2 * pad - ksize) // stride + 1
    Wo = (W + 2 * pad - ksize) // stride + 1
    if x_ch_off < 0 or x_ch_off + groups * cin > xcs:
        raise W2CError("conv_fp8: channels [%d, %d) outside the tensor's %d channels" % (x_ch_off, x_ch_off + groups * cin, xcs))
    y, gstride, ycs = None, 0, groups * cout
    if out_groups is not None:
        g0 = out_groups[0]
        _need_gpu(*out_groups)
        if len(out_groups) != groups or any(t.shape != g0.shape or t.dtype != BF16 for t in out_groups) or \
                tuple(g0.shape[:3]) != (M, Ho, Wo) or g0.shape[3] < cout:
            raise W2CError("conv_fp8: bad out_groups")
        if groups > 1:
            gstride = (out_groups[1].data_ptr() - g0.data_ptr()) // 2
            if any(t.data_ptr() - g0.data_ptr() != 2 * i * gstride for i, t in enumerate(out_groups)) or gstride % 8 or gstride == 0:
                raise W2CError("conv_fp8: out_groups tensors must be evenly spaced, 16-byte aligned")
        y, ycs = g0, g0.shape[3]
    elif out is not None:                       # write channels [out_ch_off, out_ch_off + groups*cout) of a wider bf16 tensor
        _need_gpu(out)
        if out.dtype != BF16 or tuple(out.shape[:3]) != (M, Ho, Wo) or out_ch_off < 0 or out_ch_off % 8 or \
                out_ch_off + groups * cout > out.shape[3] or residual is not None:
            raise W2CError("conv_fp8: bad out / out_ch_off")
        y, ycs = out, out.shape[3]
    elif out_bf16:
        y = torch.empty((M, Ho, Wo, groups * cout), dtype=BF16, device=dev)
    y8 = None
    if out_fp8_scale is not None:
        y8 = torch.empty((M, Ho, Wo, groups * cout), dtype=torch.uint8, device=dev)
    if y is None and y8 is None:
        raise W2CError("conv_fp8: no output requested")
    if residual is not None and (residual.dtype != BF16 or tuple(residual.shape) != (M, Ho, Wo, groups * cout)
                                 or out_groups is not None):
        raise W2CError("conv_fp8: residual must be bf16 [M,Ho,Wo,groups*cout] (and excludes out_groups)")
    timer = getattr(_tls, "conv_timer", None)
    tok = timer.begin(dev) if timer is not None else None
    es = 1 if f8 else 2
    with torch.cuda.device(dev):
        check(_native.lib().w2c_conv_igemm_fp8(x.data_ptr() + es * x_ch_off, 1 if f8 else 0, M, H, W, cin, xcs, _p(w_packed),
                                               cout, ksize, stride, groups, _p(scale), _p(shift), _p(residual),
                                               1 if relu else 0, (_p(y) + 2 * out_ch_off) if y is not None else 0, ycs,
                                               int(gstride), _p(y8), groups * cout,
                                               float(out_fp8_scale) if out_fp8_scale is not None else 1.0,
                                               _p(zero_page(dev)), int(variant), _stream(dev)), "w2c_conv_igemm_fp8")
    if timer is not None:
        flops = 2.0 * M * Ho * Wo * cout * (ksize * ksize * cin) * groups
        nbytes = (M * H * W * cin * groups * es + M * Ho * Wo * cout * groups * ((2 if y is not None else 0) + (1 if y8 is not None else 0))
                  + (M * Ho * Wo * cout * groups * 2 if residual is not None else 0) + groups * cout * ksize * ksize * cin * es)
        timer.end(tok, dev, flops, (M * Ho * Wo, cin, cout, ksize, stride, groups, "fp8" if f8 else "bf16>fp8"), nbytes)
    return (list(out_groups) if out_groups is not None else y), y8


def conv_s2_block(x, x_ch_off, cin, w3, scale3, shift3, w1, scale1, shift1, cout, groups, t_bf16=True, t_fp8_scale=None,
                  variant=-1):
    """Front of a stride-2 BasicBlock in one launch (include/w2c_hip.h w2c_conv_s2_block): returns (t bf16 | None,
    t fp8 (uint8) | None, idt bf16).  x bf16 or uint8 (e4m3) NHWC; w3 [G,Cout,9*cin], w1 [G,Cout,cin] in x's operand type."""
    dev = _need_gpu(x, w3, scale3, shift3, w1, scale1, shift1)
    f8 = x.dtype == torch.uint8
    want = torch.uint8 if f8 else BF16
    if (not f8 and x.dtype != BF16) or w3.dtype != want or w1.dtype != want:
        raise W2CError("conv_s2_block: x / w3 / w1 must share one operand type (bf16 or uint8 e4m3)")
    M, H, W, xcs = x.shape
    if x_ch_off < 0 or x_ch_off + groups * cin > xcs:
        raise W2CError("conv_s2_block: channels outside the tensor")
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    t16 = _empty((M, Ho, Wo, groups * cout), dtype=BF16, device=dev) if t_bf16 else None
    t8 = _empty((M, Ho, Wo, groups * cout), dtype=torch.uint8, device=dev) if t_fp8_scale is not None else None
    if t16 is None and t8 is None:
        raise W2CError("conv_s2_block: no conv1 output requested")
    idt = _empty((M, Ho, Wo, groups * cout), dtype=BF16, device=dev)
    es = 1 if f8 else 2
    timer = getattr(_tls, "conv_timer", None)
    tok = timer.begin(dev) if timer is not None else None
    with torch.cuda.device(dev):
        check(_native.lib().w2c_conv_s2_block(x.data_ptr() + es * x_ch_off, 1 if f8 else 0, M, H, W, cin, xcs, _p(w3), _p(scale3),
                                              _p(shift3), _p(w1), _p(scale1), _p(shift1), cout, groups, _p(t16), groups * cout,
                                              _p(t8), groups * cout, float(t_fp8_scale) if t_fp8_scale is not None else 1.0,
                                              _p(idt), groups * cout, _p(zero_page(dev)), int(variant), _stream(dev)),
              "w2c_conv_s2_block")
    if timer is not None:
        flops = 2.0 * M * Ho * Wo * cout * (10 * cin) * groups
        nbytes = (M * H * W * cin * groups * es + M * Ho * Wo * cout * groups * ((2 if t16 is not None else 0) + (1 if t8 is not None else 0) + 2)
                  + groups * cout * 10 * cin * es)
        timer.end(tok, dev, flops, (M * Ho * Wo, cin, cout, "3+1", 2, groups), nbytes)
    return t16, t8, idt


def pack_w1frag(w1, cin):
    """[G, Cout, cin] bf16 weights of a 1x1 conv -> MFMA A-fragment order with one tap (include/w2c_hip.h, w2c_conv_s2_block_wreg):
    [G][Cout/32][cin/64][k slice 0..3][half][channel % 32][8].  Same bytes, permuted (plan-build time)."""
    G, cout, K = w1.shape
    if K != cin or cin % 64 or cout % 32:
        raise W2CError("pack_w1frag: needs a 1x1 conv with cin % 64 == 0 and cout % 32 == 0")
    v = w1.reshape(G, cout // 32, 32, cin // 64, 4, 2, 8)              # g, nb, c32, cc, kk, half, e
    return v.permute(0, 1, 3, 4, 5, 2, 6).contiguous().reshape(G, cout, K)


def conv_s2_block_wreg_supported(H, W, cin, cout):
    return bool(_native.lib().w2c_conv_s2_block_wreg_supported(int(H), int(W), int(cin), int(cout)))


def conv_s2_block_wreg(x, x_ch_off, cin, w3frag, scale3, shift3, w1frag, scale1, shift1, cout, groups, form=0):
    """Front of a stride-2 BasicBlock on the weights-to-registers kernel (include/w2c_hip.h w2c_conv_s2_block_wreg): returns
    (t bf16, idt bf16).  x bf16 NHWC; w3frag from pack_wfrag_device, w1frag from pack_w1frag."""
    dev = _need_gpu(x, w3frag, scale3, shift3, w1frag, scale1, shift1)
    if x.dtype != BF16 or w3frag.dtype != BF16 or w1frag.dtype != BF16:
        raise W2CError("conv_s2_block_wreg: bf16 operands")
    M, H, W, xcs = x.shape
    if x_ch_off < 0 or x_ch_off + groups * cin > xcs:
        raise W2CError("conv_s2_block_wreg: channels outside the tensor")
    if tuple(w3frag.shape) != (groups, cout, 9 * cin) or tuple(w1frag.shape) != (groups, cout, cin):
        raise W2CError("conv_s2_block_wreg: weight shapes")
    Ho, Wo = H // 2, W // 2
    t16 = _empty((M, Ho, Wo, groups * cout), dtype=BF16, device=dev)
    idt = _empty((M, Ho, Wo, groups * cout), dtype=BF16, device=dev)
    timer = getattr(_tls, "conv_timer", None)
    tok = timer.begin(dev) if timer is not None else None
    with torch.cuda.device(dev):
        check(_native.lib().w2c_conv_s2_block_wreg(x.data_ptr() + 2 * x_ch_off, M, H, W, cin, xcs, _p(w3frag), _p(scale3), _p(shift3),
                                                   _p(w1frag), _p(scale1), _p(shift1), cout, groups, _p(t16), groups * cout,
                                                   _p(idt), groups * cout, int(form), _stream(dev)),
              "w2c_conv_s2_block_wreg")
    if timer is not None:
        flops = 2.0 * M * Ho * Wo * cout * (10 * cin) * groups
        nbytes = M * H * W * cin * groups * 2 + M * Ho * Wo * cout * groups * 4 + groups * cout * 10 * cin * 2
        timer.end(tok, dev, flops, (M * Ho * Wo, cin, cout, "3+1", 2, groups), nbytes)
    return t16, idt


def conv_s2_front_c64_supported(H, W, cin, cout):
    return bool(_native.lib().w2c_conv_s2_front_c64_supported(int(H), int(W), int(cin), int(cout)))


def conv_s2_front_c64(x, x_ch_off, w3frag, scale3, shift3, w1frag, scale1, shift1, groups, slabs=False):
    """Front of the first stride-2 BasicBlock (64 -> 128 per group) on the persistent weights-stationary kernel (include/w2c_hip.h
    w2c_conv_s2_front_c64): returns (t, idt), bf16.  slabs=False: [M, H/2, W/2, groups*128], groups side by side (w2c_conv_s2_block's
    layout); slabs=True: [groups, M, H/2, W/2, 128], one compact tensor per group (each trunk's chain goes on with a one-group tensor)."""
    dev = _need_gpu(x, w3frag, scale3, shift3, w1frag, scale1, shift1)
    if x.dtype != BF16 or w3frag.dtype != BF16 or w1frag.dtype != BF16:
        raise W2CError("conv_s2_front_c64: bf16 operands")
    M, H, W, xcs = x.shape
    if x_ch_off < 0 or x_ch_off + groups * 64 > xcs:
        raise W2CError("conv_s2_front_c64: channels outside the tensor")
    if tuple(w3frag.shape) != (groups, 128, 9 * 64) or tuple(w1frag.shape) != (groups, 128, 64):
        raise W2CError("conv_s2_front_c64: weight shapes")
    Ho, Wo = H // 2, W // 2
    if slabs:
        shape, cs, gs = (groups, M, Ho, Wo, 128), 128, M * Ho * Wo * 128
    else:
        shape, cs, gs = (M, Ho, Wo, groups * 128), groups * 128, 128
    t16 = _empty(shape, dtype=BF16, device=dev)
    idt = _empty(shape, dtype=BF16, device=dev)
    timer = getattr(_tls, "conv_timer", None)
    tok = timer.begin(dev) if timer is not None else None
    with torch.cuda.device(dev):
        check(_native.lib().w2c_conv_s2_front_c64(x.data_ptr() + 2 * x_ch_off, M, H, W, xcs, _p(w3frag), _p(scale3), _p(shift3),
                                                  _p(w1frag), _p(scale1), _p(shift1), groups, _p(t16), cs, gs, _p(idt), cs, gs,
                                                  _stream(dev)),
              "w2c_conv_s2_front_c64")
    if timer is not None:
        flops = 2.0 * M * Ho * Wo * 128 * (10 * 64) * groups
        nbytes = M * H * W * 64 * groups * 2 + M * Ho * Wo * 128 * groups * 4 + groups * 128 * 10 * 64 * 2
        timer.end(tok, dev, flops, (M * Ho * Wo, 64, 128, "3+1", 2, groups), nbytes)
    return t16, idt


_wgrad_ws = {}


def stem_conv7x7_train(x_nhwc3, w_packed, out=None):
    """training forward of the 7x7/2 stem: x bf16 NHWC [M,H,W,3] -> raw conv bf16 NHWC [M,H/2,W/2,64]."""
    dev = _need_gpu(x_nhwc3, w_packed, out)
    M, H, W, c = x_nhwc3.shape
    if c != 3 or x_nhwc3.dtype != BF16 or not x_nhwc3.is_contiguous() or w_packed.dtype != BF16 or w_packed.numel() != 64 * 224:
        raise W2CError("stem_conv7x7_train: contiguous bf16 [M,H,W,3] frames and [64,224] packed weights expected")
    if out is None:
        out = torch.empty((M, H // 2, W // 2, 64), dtype=BF16, device=dev)
    with torch.cuda.device(dev):
        check(_native.lib().w2c_stem_conv7x7_train_bf16(_p(x_nhwc3), M, H, W, _p(w_packed), _p(out), _stream(dev)),
              "w2c_stem_conv7x7_train_bf16")
    return out


def stem_wgrad(x_nhwc3, dy):
    """dW f32 [64,3,7,7] of the stem conv from x (bf16 NHWC [M,H,W,3]) and dy (bf16 NHWC [M,H/2,W/2,>=64])."""
    dev = _need_gpu(x_nhwc3, dy)
    M, H, W, c = x_nhwc3.shape
    if (c != 3 or x_nhwc3.dtype != BF16 or dy.dtype != BF16 or not x_nhwc3.is_contiguous() or not dy.is_contiguous()
            or tuple(dy.shape[:3]) != (M, H // 2, W // 2)):
        raise W2CError("stem_wgrad: bf16 [M,H,W,3] frames and bf16 [M,H/2,W/2,C] gradient expected")
    lib = _native.lib()
    need = lib.w2c_stem_wgrad_workspace_bytes(M, H, W)
    if need < 0:
        raise W2CError("stem_wgrad: H % 16 == 0 and W % 32 == 0 required")
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), _stream(dev))
    ws = _wgrad_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(int(need), 32 << 20), dtype=torch.uint8, device=dev)
        _wgrad_ws[key] = ws
    dw = torch.empty((64, 3, 7, 7), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(lib.w2c_stem_wgrad_bf16(_p(x_nhwc3), M, H, W, _p(dy), dy.shape[3], _p(dw), _p(ws), ws.numel(), _stream(dev)),
              "w2c_stem_wgrad_bf16")
    return dw


def pack_conv_weights(weight, mode):
    """nn.Conv2d weight f32 [Cout,Cin,k,k] -> packed bf16 operand: mode 0 [1, Cout, k*k*Cin] (forward), mode 1
    [1, Cin, k*k*Cout] (flipped + transposed: the input-gradient conv).  include/w2c_hip.h w2c_pack_conv_weights_bf16."""
    dev = _need_gpu(weight)
    co, ci, k, k2 = weight.shape
    if weight.dtype != torch.float32 or not weight.is_contiguous() or k != k2:
        raise W2CError("pack_conv_weights: contiguous f32 [Cout,Cin,k,k] expected")
    out = torch.empty((1, co if mode == 0 else ci, k * k * (ci if mode == 0 else co)), dtype=BF16, device=dev)
    with torch.cuda.device(dev):
        check(_native.lib().w2c_pack_conv_weights_bf16(_p(weight), co, ci, k, mode, _p(out), _stream(dev)), "w2c_pack_conv_weights_bf16")
    return out


def pack_conv_weights_both(weight):
    """-> (forward operand [1,Cout,k*k*Cin], input-gradient operand [1,Cin,k*k*Cout]) from one read of the parameter."""
    dev = _need_gpu(weight)
    co, ci, k, k2 = weight.shape
    if weight.dtype != torch.float32 or not weight.is_contiguous() or k != k2:
        raise W2CError("pack_conv_weights: contiguous f32 [Cout,Cin,k,k] expected")
    fwd = torch.empty((1, co, k * k * ci), dtype=BF16, device=dev)
    dg = torch.empty((1, ci, k * k * co), dtype=BF16, device=dev)
    with torch.cuda.device(dev):
        check(_native.lib().w2c_pack_conv_weights_bf16_both(_p(weight), co, ci, k, _p(fwd), _p(dg), _stream(dev)),
              "w2c_pack_conv_weights_bf16_both")
    return fwd, dg


def conv_wgrad(x, x_ch_off, cin, dy, cout, ksize, stride, groups, oihw=False, _limit=None):
    """dW f32 [G, cout, ksize*ksize*cin] of the conv that maps x (bf16 NHWC, channels [x_ch_off, +G*cin)) to an output whose
    gradient is dy (bf16 NHWC [M,Ho,Wo,G*cout]).  include/w2c_hip.h w2c_conv_wgrad_bf16."""
    dev = _need_gpu(x, dy)
    if x.dtype != BF16 or dy.dtype != BF16:
        raise W2CError("conv_wgrad: bf16 tensors expected")
    M, H, W, xcs = x.shape
    pad = 1 if ksize == 3 else 0
    Ho, Wo = (H + 2 * pad - ksize) // stride + 1, (W + 2 * pad - ksize) // stride + 1
    if tuple(dy.shape[:3]) != (M, Ho, Wo) or dy.shape[3] < groups * cout or x_ch_off < 0 or x_ch_off + groups * cin > xcs:
        raise W2CError("conv_wgrad: geometry mismatch")
    # the kernel addresses x and dy through 31-bit buffer descriptors (like the forward, whose wrapper slices the batch): cut the
    # batch into slices that fit and add the slices' dW in slice order (fixed order: deterministic)
    lim = int(_limit) if _limit else (1 << 31) - 1
    per_img = max(H * W * xcs * 2, Ho * Wo * dy.shape[3] * 2)
    if M * per_img > lim:
        step = max(1, lim // per_img)
        if step >= M or per_img > lim:
            raise W2CError("conv_wgrad: one image exceeds the kernels' 2 GiB addressing range")
        total = None
        for lo in range(0, M, step):
            part = conv_wgrad(x[lo:lo + step], x_ch_off, cin, dy[lo:lo + step], cout, ksize, stride, groups, oihw=oihw, _limit=_limit)
            total = part if total is None else total.add_(part)
        return total
    need = _native.lib().w2c_conv_wgrad_workspace_bytes(M, H, W, cin, cout, ksize, stride, groups)
    if need < 0:
        raise W2CError("conv_wgrad: unsupported shape (Cin, Cout multiples of 64; 3x3 or 1x1; stride 1 or 2)")
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), _stream(dev))
    ws = _wgrad_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(int(need), 32 << 20), dtype=torch.uint8, device=dev)
        _wgrad_ws[key] = ws
    if oihw:                    # nn.Conv2d's parameter layout (groups = 1): what autograd returns for `weight`
        if groups != 1:
            raise W2CError("conv_wgrad: the OIHW form is for groups = 1")
        dw = torch.empty((cout, cin, ksize, ksize), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(_native.lib().w2c_conv_wgrad_bf16_oihw(x.data_ptr() + 2 * x_ch_off, M, H, W, cin, xcs, _p(dy), cout, dy.shape[3],
                                                         ksize, stride, _p(dw), _p(ws), ws.numel(), _stream(dev)),
                  "w2c_conv_wgrad_bf16_oihw")
        return dw
    dw = torch.empty((groups, cout, ksize * ksize * cin), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_native.lib().w2c_conv_wgrad_bf16(x.data_ptr() + 2 * x_ch_off, M, H, W, cin, xcs, _p(dy), cout, dy.shape[3], ksize,
                                                stride, groups, _p(dw), _p(ws), ws.numel(), _stream(dev)), "w2c_conv_wgrad_bf16")
    return dw


def zero_insert2(dy, H, W):
    """dy bf16 NHWC [M,Ho,Wo,C] -> [M,H,W,C] with dy at the even positions, zeros elsewhere (stride-2 dgrad helper)."""
    dev = _need_gpu(dy)
    M, Ho, Wo, C = dy.shape
    u = torch.empty((M, H, W, C), dtype=BF16, device=dev)
    with torch.cuda.device(dev):
        check(_native.lib().w2c_zero_insert2_bf16(_p(dy), M, Ho, Wo, C, _p(u), H, W, _stream(dev)), "w2c_zero_insert2_bf16")
    return u


_bn_ws = {}


def _bn_workspace(dev, P, C):
    need = _native.lib().w2c_bn_workspace_bytes(P, C)
    if need < 0:
        raise W2CError("bn: C must be a multiple of 8 and <= 2048")
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), _stream(dev))
    ws = _bn_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(int(need), 1024 * 2 * 2048 * 4), dtype=torch.uint8, device=dev)
        _bn_ws[key] = ws
    return ws


def bn_train_forward(x, gamma, beta, running_mean, running_var, momentum, eps, residual=None, relu=True, out=None,
                     num_batches_tracked=None):
    """x dense bf16 NHWC [..., C] -> (y bf16 same shape, mean f32 [C], rstd f32 [C]); running stats updated in place
    (None to skip).  include/w2c_hip.h w2c_bn_train_forward."""
    dev = _need_gpu(x, gamma, beta, running_mean, running_var, residual)
    C = x.shape[-1]
    P = x.numel() // C
    if x.dtype != BF16 or (residual is not None and (residual.dtype != BF16 or residual.shape != x.shape)):
        raise W2CError("bn: bf16 NHWC tensors expected")
    y = torch.empty_like(x) if out is None else out
    if y.shape != x.shape or y.dtype != BF16 or not y.is_contiguous():
        raise W2CError("bn: bad out tensor")
    stats = torch.empty((4, C), dtype=torch.float32, device=dev)          # mean | rstd | a | b
    ws = _bn_workspace(dev, P, C)
    with torch.cuda.device(dev):
        if num_batches_tracked is not None and (num_batches_tracked.dtype != torch.int64 or num_batches_tracked.device != dev):
            raise W2CError("bn: num_batches_tracked must be an int64 tensor on the activations' device")
        check(_native.lib().w2c_bn_train_forward(_p(x), P, C, _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                                                 _p(num_batches_tracked), float(momentum), float(eps), _p(residual), 1 if relu else 0, _p(y),
                                                 stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(), _p(ws),
                                                 ws.numel(), _stream(dev)), "w2c_bn_train_forward")
    return y, stats[0], stats[1]


def bn_train_backward(dy, y, x, gamma, mean, rstd, want_dres=False):
    """-> (dx bf16, dres bf16 | None, dgamma f32 [C], dbeta f32 [C]); y = the forward output when ReLU was applied, else None."""
    dev = _need_gpu(dy, y, x, gamma, mean, rstd)
    C = x.shape[-1]
    P = x.numel() // C
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    g = torch.empty((5, C), dtype=torch.float32, device=dev)              # dgamma | dbeta | k1 | k2 | k3
    ws = _bn_workspace(dev, P, C)
    with torch.cuda.device(dev):
        check(_native.lib().w2c_bn_train_backward(_p(dy), _p(y), _p(x), P, C, _p(gamma), _p(mean), _p(rstd), _p(dx), _p(dres),
                                                  g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), _p(ws), ws.numel(),
                                                  _stream(dev)), "w2c_bn_train_backward")
    return dx, dres, g[0], g[1]


def _allreduce_sums(sums_and_count, group):
    """SUM over the ranks of [2C sums | pixel count] (f64), in place"""
    import torch.distributed as dist
    dist.all_reduce(sums_and_count, op=dist.ReduceOp.SUM, group=group)
    return sums_and_count


def bn_train_forward_sync(x, gamma, beta, running_mean, running_var, momentum, eps, group, residual=None, relu=True, out=None,
                          num_batches_tracked=None):
    """bn_train_forward with the statistics taken over EVERY rank's pixels (agent-sharded training: the reference's train-mode
    BatchNorm runs over the agent-concatenated batch, agent.py:1108-1111): local sums (w2c_bn_train_sums) -> all-reduce of the 2C sums +
    the pixel count -> finalize + apply from the global sums (w2c_bn_train_forward_sums).  -> (y, mean, rstd, P_total)."""
    dev = _need_gpu(x, gamma, beta, running_mean, running_var, residual)
    C = x.shape[-1]
    P = x.numel() // C
    if x.dtype != BF16 or (residual is not None and (residual.dtype != BF16 or residual.shape != x.shape)):
        raise W2CError("bn: bf16 NHWC tensors expected")
    y = torch.empty_like(x) if out is None else out
    stats = torch.empty((4, C), dtype=torch.float32, device=dev)
    sums = torch.empty(2 * C + 1, dtype=torch.float64, device=dev)
    ws = _bn_workspace(dev, P, C)
    with torch.cuda.device(dev):
        check(_native.lib().w2c_bn_train_sums(0, _p(x), 0, 0, 0, 0, P, C, _p(sums), _p(ws), ws.numel(), _stream(dev)), "w2c_bn_train_sums")
    sums[2 * C] = float(P)
    _allreduce_sums(sums, group)
    p_total = float(sums[2 * C].item())            # (one host read per layer: the agent-sharded training step is not graph-captured)
    with torch.cuda.device(dev):
        check(_native.lib().w2c_bn_train_forward_sums(_p(x), P, C, _p(sums), p_total, _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                                                      _p(num_batches_tracked), float(momentum), float(eps), _p(residual),
                                                      1 if relu else 0, _p(y), stats[0].data_ptr(), stats[1].data_ptr(),
                                                      stats[2].data_ptr(), _stream(dev)), "w2c_bn_train_forward_sums")
    return y, stats[0], stats[1], p_total


def bn_train_backward_sync(dy, y, x, gamma, mean, rstd, p_total, group, want_dres=False):
    """bn_train_backward with the two backward sums added over the ranks; dgamma / dbeta are THIS rank's contributions."""
    dev = _need_gpu(dy, y, x, gamma, mean, rstd)
    C = x.shape[-1]
    P = x.numel() // C
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    g = torch.empty((5, C), dtype=torch.float32, device=dev)
    loc = torch.empty(2 * C, dtype=torch.float64, device=dev)
    ws = _bn_workspace(dev, P, C)
    with torch.cuda.device(dev):
        check(_native.lib().w2c_bn_train_sums(1, _p(x), _p(dy), _p(y), _p(mean), _p(rstd), P, C, _p(loc), _p(ws), ws.numel(), _stream(dev)),
              "w2c_bn_train_sums")
    glob = _allreduce_sums(loc.clone(), group)
    with torch.cuda.device(dev):
        check(_native.lib().w2c_bn_train_backward_sums(_p(dy), _p(y), _p(x), P, C, _p(gamma), _p(mean), _p(rstd), _p(loc), _p(glob),
                                                       float(p_total), _p(dx), _p(dres), g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(),
                                                       _stream(dev)), "w2c_bn_train_backward_sums")
    return dx, dres, g[0], g[1]


def linear(x, w, b, relu, x_stride=None, rows=None, k=None):
    """y[M,O] = act(x[M,K] W^T + b); x bf16 or f32 (2-D view given by rows/k/x_stride)."""
    dev = _need_gpu(x, w, b)
    O, K = w.shape
    if rows is None:
        rows = x.shape[0]
    if x_stride is None:
        x_stride = K
    if rows < 1 or x_stride < K or x.numel() < (rows - 1) * x_stride + K:
        raise W2CError("linear: %d rows of stride %d (K=%d) do not fit the %d-element input" % (rows, x_stride, K, x.numel()))
    y = torch.empty((rows, O), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_native.lib().w2c_linear_f32(_p(x), 1 if x.dtype == BF16 else 0, x_stride, rows, K, _p(w), _p(b), O,
                                           1 if relu else 0, _p(y), _stream(dev)), "w2c_linear_f32")
    return y


def head_tail(h0, col_off, k1, w1t, b1, w2t, b2):
    """out[M,O] = W2 relu(W1 h0[:, col_off:col_off+k1] + b1) + b2 (weights K-major)."""
    dev = _need_gpu(h0, w1t, b1, w2t, b2)
    M, stride = h0.shape
    H1 = w1t.shape[1]
    O = w2t.shape[1]
    if col_off < 0 or col_off + k1 > stride or w1t.shape[0] != k1 or w2t.shape[0] != H1:
        raise W2CError("head_tail: columns [%d, %d) / weights %s, %s do not fit h0 %s" %
                       (col_off, col_off + k1, tuple(w1t.shape), tuple(w2t.shape), tuple(h0.shape)))
    out = torch.empty((M, O), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_native.lib().w2c_head_tail_f32(h0.data_ptr() + 4 * col_off, stride, M, k1, _p(w1t), _p(b1), H1, _p(w2t),
                                              _p(b2), O, _p(out), _stream(dev)), "w2c_head_tail_f32")
    return out


def head_tail2(h0, k1, tail_a, tail_b, out_a=None, out_b=None):
    """Both heads' tails in one launch: tail_x = (col_off, w1t, b1, w2t, b2) -> (out_a [M,O_a], out_b [M,O_b]);
    out_a / out_b may be preallocated (e.g. a rank's rows of the key all-gather buffer)."""
    ca, w1a, b1a, w2a, b2a = tail_a
    cb, w1b, b1b, w2b, b2b = tail_b
    dev = _need_gpu(h0, w1a, b1a, w2a, b2a, w1b, b1b, w2b, b2b)
    M, stride = h0.shape
    H1 = w1a.shape[1]
    if w1b.shape[1] != H1:
        raise W2CError("head_tail2: both heads must share the hidden width")
    if min(ca, cb) < 0 or max(ca, cb) + k1 > stride or w1a.shape[0] != k1 or w1b.shape[0] != k1:
        raise W2CError("head_tail2: column ranges / weights do not fit h0 %s" % (tuple(h0.shape),))
    if out_a is None:
        out_a = _empty((M, w2a.shape[1]), torch.float32, dev)
    if out_b is None:
        out_b = _empty((M, w2b.shape[1]), torch.float32, dev)
    for o, w2 in ((out_a, w2a), (out_b, w2b)):
        if tuple(o.shape) != (M, w2.shape[1]) or o.dtype != torch.float32 or not o.is_contiguous() or o.device != dev:
            raise W2CError("head_tail2: bad preallocated output %s" % (tuple(o.shape),))
    with torch.cuda.device(dev):
        check(_native.lib().w2c_head_tail2_f32(_p(h0), stride, M, k1, H1, ca, _p(w1a), _p(b1a), _p(w2a), _p(b2a), w2a.shape[1],
                                               _p(out_a), cb, _p(w1b), _p(b1b), _p(w2b), _p(b2b), w2b.shape[1], _p(out_b),
                                               _stream(dev)), "w2c_head_tail2_f32")
    return out_a, out_b


def pack_fc0_frag(w0):
    """[O, K] f32 fc.0 weights (heads stacked along O) -> the fragment order of w2c_head_fc0_mfma_f32 (include/w2c_hip.h):
    [O/32][K/8][half][o % 32][4].  Same values, permuted; host or device tensor."""
    O, K = w0.shape
    if O % 32 or K % 8:
        raise W2CError("pack_fc0_frag: O % 32 == 0 and K % 8 == 0 required")
    return w0.reshape(O // 32, 32, K // 8, 2, 4).permute(0, 2, 3, 1, 4).contiguous().reshape(O, K)


HEAD_FC0_KSPLIT = 16


def head_fc0_supported(M, K, O, ksplit=HEAD_FC0_KSPLIT):
    # geometry only (never the row count M): a rank's shard and the unsharded batch must take the same kernel
    return O % 32 == 0 and K % (ksplit * 256) == 0


def head_fc0_mfma(x, x_stride, M, K, wfrag, O, ksplit=HEAD_FC0_KSPLIT, part=None):
    """fc.0 of the heads on the f32 matrix pipe: x bf16 rows [M][x_stride] -> split-K partials f32 [ksplit, M, O] (no bias, no ReLU)."""
    dev = _need_gpu(x, wfrag, part)
    if x.dtype != BF16 or wfrag.dtype != torch.float32:
        raise W2CError("head_fc0_mfma: x must be bf16, wfrag f32")
    if part is None:
        part = _empty((ksplit, M, O), torch.float32, dev)
    with torch.cuda.device(dev):
        check(_native.lib().w2c_head_fc0_mfma_f32(_p(x), x_stride, M, K, _p(wfrag), O, ksplit, _p(part), _stream(dev)),
              "w2c_head_fc0_mfma_f32")
    return part


def head_tail2_parts(part, b0, k1, tail_a, tail_b, out_a=None, out_b=None):
    """head_tail2 on split-K partials of fc.0 (part f32 [P, M, O_total]): h0 = relu(sum_p part[p] + b0) in the same launch."""
    ca, w1a, b1a, w2a, b2a = tail_a
    cb, w1b, b1b, w2b, b2b = tail_b
    dev = _need_gpu(part, b0, w1a, b1a, w2a, b2a, w1b, b1b, w2b, b2b)
    P, M, stride = part.shape
    H1 = w1a.shape[1]
    if w1b.shape[1] != H1 or min(ca, cb) < 0 or max(ca, cb) + k1 > stride or w1a.shape[0] != k1 or w1b.shape[0] != k1:
        raise W2CError("head_tail2_parts: column ranges / weights do not fit %s" % (tuple(part.shape),))
    if out_a is None:
        out_a = _empty((M, w2a.shape[1]), torch.float32, dev)
    if out_b is None:
        out_b = _empty((M, w2b.shape[1]), torch.float32, dev)
    for o, w2 in ((out_a, w2a), (out_b, w2b)):
        if tuple(o.shape) != (M, w2.shape[1]) or o.dtype != torch.float32 or not o.is_contiguous() or o.device != dev:
            raise W2CError("head_tail2_parts: bad preallocated output %s" % (tuple(o.shape),))
    with torch.cuda.device(dev):
        check(_native.lib().w2c_head_tail2p_f32(_p(part), P, M * stride, _p(b0), stride, M, k1, H1, ca, _p(w1a), _p(b1a), _p(w2a),
                                                _p(b2a), w2a.shape[1], _p(out_a), cb, _p(w1b), _p(b1b), _p(w2b), _p(b2b),
                                                w2b.shape[1], _p(out_b), _stream(dev)), "w2c_head_tail2p_f32")
    return out_a, out_b


MODE_IDS = {"softmax": 0, "argmax_test": 1, "activated": 2}


def comm_graph(query, key, wq, bq, B, N, who, mode, thres=0.2, tie_bias=0.001, q_lo=0, q_n=None):
    """-> prob [B,N,q_n] f32, coef [B,N,q_n] f32, action [B,q_n] i64, nnz_offdiag [B] i32."""
    dev = _need_gpu(query, key, wq, bq)
    Dk, Dq = wq.shape
    if q_n is None:
        q_n = N - q_lo
    prob = torch.empty((B, N, q_n), dtype=torch.float32, device=dev)
    coef = torch.empty((B, N, q_n), dtype=torch.float32, device=dev)
    action = torch.empty((B, q_n), dtype=torch.int64, device=dev)
    nnz = torch.empty((B,), dtype=torch.int32, device=dev)
    ws = torch.empty((N * B, Dq + 1), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_native.lib().w2c_comm_graph(_p(query), _p(key), _p(wq), _p(bq), B, N, Dq, Dk, 1 if who else 0,
                                           MODE_IDS[mode], float(thres), float(tie_bias), q_lo, q_n, _p(ws),
                                           _p(prob), _p(coef), _p(action), _p(nnz), _stream(dev)), "w2c_comm_graph")
    return prob, coef, action, nnz


def comm_graph_projected(query, tproj, B, N, who, mode, thres=0.2, tie_bias=0.001, q_lo=0, q_n=None):
    """Like comm_graph, from projected keys tproj [N*B, Dq+1] (see w2c_comm_graph_projected)."""
    dev = _need_gpu(query, tproj)
    Dq = tproj.shape[1] - 1
    if q_n is None:
        q_n = N - q_lo
    prob = torch.empty((B, N, q_n), dtype=torch.float32, device=dev)
    coef = torch.empty((B, N, q_n), dtype=torch.float32, device=dev)
    action = torch.empty((B, q_n), dtype=torch.int64, device=dev)
    nnz = torch.empty((B,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(_native.lib().w2c_comm_graph_projected(_p(query), _p(tproj), B, N, Dq, 1 if who else 0, MODE_IDS[mode],
                                                     float(thres), float(tie_bias), q_lo, q_n, _p(prob), _p(coef),
                                                     _p(action), _p(nnz), _stream(dev)), "w2c_comm_graph_projected")
    return prob, coef, action, nnz


def graph_outputs(dev, B, N, q_n):
    """prob f32 [B,N,q_n] | action i64 [B,q_n] | nnz i32 [B] carved out of ONE buffer, so a caller that must hand out fresh
    copies (the HIP-graph path: its outputs are static buffers) copies once instead of three times.  -> (pack, prob, action, nnz)"""
    n_prob, n_act = B * N * q_n * 4, B * q_n * 8
    off_act = (n_prob + 7) // 8 * 8
    off_nnz = off_act + n_act
    pack = torch.empty(off_nnz + B * 4, dtype=torch.uint8, device=dev)
    return (pack,) + carve_graph_outputs(pack, B, N, q_n)


def carve_graph_outputs(pack, B, N, q_n):
    n_prob, n_act = B * N * q_n * 4, B * q_n * 8
    off_act = (n_prob + 7) // 8 * 8
    off_nnz = off_act + n_act
    prob = pack[:n_prob].view(torch.float32).view(B, N, q_n)
    action = pack[off_act:off_act + n_act].view(torch.int64).view(B, q_n)
    nnz = pack[off_nnz:off_nnz + B * 4].view(torch.int32)
    return prob, action, nnz


def comm_graph_fuse(query, tproj, v, v_ch, B, N, who, mode, thres=0.2, tie_bias=0.001, q_lo=0, q_n=None, append_own=False):
    """comm_graph_projected + fuse_values in one launch (w2c_comm_graph_fuse).  -> fused bf16 [q_n*B,h,w,C|2C], prob, coef,
    action, nnz, pack (prob / action / nnz are views of `pack`, see graph_outputs)."""
    dev = _need_gpu(query, tproj, v)
    Dq = tproj.shape[1] - 1
    if q_n is None:
        q_n = N - q_lo
    pack, prob, action, nnz = graph_outputs(dev, B, N, q_n)
    coef = torch.empty((B, N, q_n), dtype=torch.float32, device=dev)
    _, h, w, vcs = v.shape
    ocs = 2 * v_ch if append_own else v_ch
    out = torch.empty((q_n * B, h, w, ocs), dtype=BF16, device=dev)
    with torch.cuda.device(dev):
        check(_native.lib().w2c_comm_graph_fuse(_p(query), _p(tproj), B, N, Dq, 1 if who else 0, MODE_IDS[mode], float(thres),
                                                float(tie_bias), q_lo, q_n, _p(prob), _p(coef), _p(action), _p(nnz), _p(v), vcs,
                                                h * w, v_ch, 1 if append_own else 0, _p(out), ocs, _stream(dev)),
              "w2c_comm_graph_fuse")
    return out, prob, coef, action, nnz, pack


def pack_offsets(B, N, q_n):
    """byte offsets of action / nnz inside the packed graph outputs (graph_outputs)"""
    n_prob, n_act = B * N * q_n * 4, B * q_n * 8
    off_act = (n_prob + 7) // 8 * 8
    return off_act, off_act + n_act


def comm_graph_fuse_u(query, tproj, u, C, bias, B, N, who, mode, thres=0.2, tie_bias=0.001, q_lo=0, q_n=None, own_off=-1, pack2=None,
                      u_own=None):
    """w2c_comm_graph_fuse_u: graph + fusion of the U maps (decoder conv0 of every agent's value map, no bias, f32 NHWC
    [N*B,h,w,ucs]) + bias + ReLU -> y bf16 [q_n*B,h,w,C] = relu(conv0(fused map)), prob, coef, action, nnz, pack.
    MIMOcomWho's own term: u_own = f32 [q_n*B,h,w,own_cs] rows of the LOCAL queries (channels [0,C)), or own_off >= 0 = the own map sits
    in channels [own_off, own_off+C) of u's own rows (the one-GPU layout [U | U_own]).
    pack2: optional tensor / SlotRef receiving a second copy of the packed prob | action | nnz (the caller-owned outputs of a
    captured forward)."""
    dev = _need_gpu(query, tproj, u, bias)
    Dq = tproj.shape[1] - 1
    if q_n is None:
        q_n = N - q_lo
    if u.dtype != torch.float32 or bias.dtype != torch.float32 or bias.numel() != C:
        raise W2CError("comm_graph_fuse_u: u and bias must be f32, bias [C]")
    pack, prob, action, nnz = graph_outputs(dev, B, N, q_n)
    coef = torch.empty((B, N, q_n), dtype=torch.float32, device=dev)
    _, h, w, ucs = u.shape
    own_ptr, own_cs = 0, 0
    if u_own is not None:
        if (u_own.dtype != torch.float32 or u_own.device != dev or u_own.dim() != 4 or u_own.shape[0] != q_n * B or tuple(u_own.shape[1:3]) != (h, w)
                or u_own.shape[3] < C or not u_own.is_contiguous()):
            raise W2CError("comm_graph_fuse_u: u_own must be contiguous f32 [q_n*B,h,w,>=C] on u's device")
        own_ptr, own_cs = u_own.data_ptr(), u_own.shape[3]
    elif own_off >= 0:
        if own_off + C > ucs:
            raise W2CError("comm_graph_fuse_u: own_off + C exceeds u's channel stride")
        own_ptr, own_cs = u.data_ptr() + 4 * (q_lo * B * h * w * ucs + own_off), ucs
    out = torch.empty((q_n * B, h, w, C), dtype=BF16, device=dev)
    with torch.cuda.device(dev):
        check(_native.lib().w2c_comm_graph_fuse_u(_p(query), _p(tproj), B, N, Dq, 1 if who else 0, MODE_IDS[mode], float(thres),
                                                  float(tie_bias), q_lo, q_n, _p(prob), _p(coef), _p(action), _p(nnz), _p(u), ucs,
                                                  h * w, C, own_ptr or None, own_cs, _p(bias), _p(out), C, _p(pack2), *pack_offsets(B, N, q_n),
                                                  _stream(dev)),
              "w2c_comm_graph_fuse_u")
    return out, prob, coef, action, nnz, pack


def fuse_values(v, v_ch, coef, B, N, q_lo, q_n, append_own=False, out=None):
    """v: bf16 NHWC [N*B,h,w,vcs] (first v_ch channels are the value map) -> [q_n*B,h,w,C or 2C]."""
    dev = _need_gpu(v, coef, out)
    _, h, w, vcs = v.shape
    ocs = 2 * v_ch if append_own else v_ch
    if out is None:
        out = torch.empty((q_n * B, h, w, ocs), dtype=BF16, device=dev)
    with torch.cuda.device(dev):
        check(_native.lib().w2c_fuse_values(_p(v), vcs, _p(coef), B, N, q_lo, q_n, h * w, v_ch, 1 if append_own else 0,
                                            _p(out), ocs, _stream(dev)), "w2c_fuse_values")
    return out


def upsample_bilinear32(low, n_classes, out=None):
    """low f32 NHWC [M,h,w,lcs] -> f32 NCHW [M,n_classes,32h,32w]."""
    dev = _need_gpu(low, out)
    M, h, w, lcs = low.shape
    if out is None:
        out = torch.empty((M, n_classes, 32 * h, 32 * w), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_native.lib().w2c_upsample_bilinear32(_p(low), M, h, w, lcs, n_classes, _p(out), _stream(dev)),
              "w2c_upsample_bilinear32")
    return out


def upsample_bilinear32_backward(gout):
    """gout f32 NCHW [M,C,32h,32w] -> f32 NCHW [M,C,h,w]: the adjoint of upsample_bilinear32."""
    dev = _need_gpu(gout)
    M, C, H, W = gout.shape
    if gout.dtype != torch.float32 or H % 32 or W % 32:
        raise W2CError("upsample backward: f32 [M,C,32h,32w] expected")
    glow = torch.empty((M, C, H // 32, W // 32), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_native.lib().w2c_upsample_bilinear32_backward(_p(gout), M, H // 32, W // 32, C, _p(glow), _stream(dev)),
              "w2c_upsample_bilinear32_backward")
    return glow


def cross_entropy2d_forward(logits, target, weight=None, size_average=True, ignore_index=250, per_pixel=False):
    """logits f32 NCHW [N,C,H,W] (contiguous), target int64 [N,H,W], weight f32 [C] or None ->
    (out3 f32 [3] = {loss, denominator, dropped out-of-range targets}, lse f32 [N,H,W], loss_px f32 [N,H,W] or None)."""
    dev = _need_gpu(logits)
    N, C, H, W = logits.shape
    if (logits.dtype != torch.float32 or not logits.is_contiguous() or target.dtype != torch.int64 or not target.is_contiguous()
            or tuple(target.shape) != (N, H, W) or target.device != dev):
        raise W2CError("cross_entropy2d: contiguous f32 NCHW logits and int64 [N,H,W] target on one device expected")
    if weight is not None and (weight.dtype != torch.float32 or weight.numel() != C or weight.device != dev or not weight.is_contiguous()):
        raise W2CError("cross_entropy2d: weight must be f32 [C] on the logits' device")
    lib = _native.lib()
    lse = torch.empty((N, H, W), dtype=torch.float32, device=dev)
    loss_px = torch.empty((N, H, W), dtype=torch.float32, device=dev) if per_pixel else None
    out3 = torch.empty(3, dtype=torch.float32, device=dev)
    nbytes = lib.w2c_cross_entropy2d_workspace_bytes(N * H * W)
    ws = torch.empty(nbytes // 8, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        check(lib.w2c_cross_entropy2d_forward(_p(logits), _p(target), _p(weight) if weight is not None else None, N, C, H * W,
                                              ignore_index, 1 if size_average else 0, _p(lse),
                                              _p(loss_px) if per_pixel else None, _p(out3), _p(ws), nbytes, _stream(dev)),
              "w2c_cross_entropy2d_forward")
    return out3, lse, loss_px


def cross_entropy2d_backward(logits, target, weight, lse, denom=None, gout=None, gpx=None, ignore_index=250):
    """-> d loss / d logits, f32 NCHW; denom / gout: 1-element f32 device tensors (or None), gpx: f32 [N,H,W] or None."""
    dev = _need_gpu(logits)
    N, C, H, W = logits.shape
    for t in (denom, gout, gpx):
        if t is not None and (t.dtype != torch.float32 or t.device != dev or not t.is_contiguous()):
            raise W2CError("cross_entropy2d backward: f32 contiguous scale tensors on the logits' device expected")
    if gpx is not None and gpx.numel() != N * H * W:
        raise W2CError("cross_entropy2d backward: gpx must have one entry per pixel")
    d = torch.empty_like(logits)
    with torch.cuda.device(dev):
        check(_native.lib().w2c_cross_entropy2d_backward(_p(logits), _p(target), _p(weight) if weight is not None else None,
                                                         _p(lse), N, C, H * W, ignore_index,
                                                         _p(denom) if denom is not None else None,
                                                         _p(gout) if gout is not None else None,
                                                         _p(gpx) if gpx is not None else None, _p(d), _stream(dev)),
              "w2c_cross_entropy2d_backward")
    return d


def upsample32_argmax(low, n_classes, out=None):
    """low f32 NHWC [M,h,w,lcs] -> u8 labels [M,32h,32w] = argmax_c of the bilinear x32 upsample (out: tensor or SlotRef)."""
    dev = _need_gpu(low)
    M, h, w, lcs = low.shape
    if out is None:
        out = torch.empty((M, 32 * h, 32 * w), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(_native.lib().w2c_upsample32_argmax(_p(low), M, h, w, lcs, n_classes, _p(out), _stream(dev)),
              "w2c_upsample32_argmax")
    return out


def _gt_kind(gt):
    if gt.dtype == torch.uint8:
        return 0
    if gt.dtype == torch.int64:
        return 1
    raise W2CError("ground-truth labels must be uint8 or int64, got %s" % gt.dtype)


CONFUSION_WS_PARTIALS = 32


def confusion_workspace(dev, n_classes):
    """zeroed int64 workspace of w2c_upsample32_argmax_confusion's two-level flush (include/w2c_hip.h); the kernel leaves it zeroed.
    One per stream: the engines keep their own."""
    stride = (n_classes * n_classes + 15) // 16 * 16
    return torch.zeros(CONFUSION_WS_PARTIALS * stride + 16, dtype=torch.int64, device=dev)


def upsample32_argmax_confusion(low, n_classes, gt, hist, want_labels=False, out=None, ws=None):
    """K9 + class argmax + confusion matrix (metrics.py:99-108) in one launch.  gt: u8 or i64 [M,32h,32w];
    hist: int64 [n*n] accumulated in place.  Returns the u8 label map when want_labels, else None.  (gt / hist / out may be SlotRefs.)
    ws: confusion_workspace() of the calling stream, or None (direct flush)."""
    dev = _need_gpu(low, gt, hist)
    M, h, w, lcs = low.shape
    if tuple(gt.shape) != (M, 32 * h, 32 * w):
        raise W2CError("confusion: labels %s do not match the %s prediction map" % (tuple(gt.shape), (M, 32 * h, 32 * w)))
    if hist.dtype != torch.int64 or hist.numel() != n_classes * n_classes:
        raise W2CError("confusion: hist must be int64 [%d]" % (n_classes * n_classes))
    if ws is not None and (ws.dtype != torch.int64 or ws.numel() < CONFUSION_WS_PARTIALS * ((n_classes * n_classes + 15) // 16 * 16) + 16):
        raise W2CError("confusion: workspace too small (ops.confusion_workspace)")
    if out is None:
        out = torch.empty((M, 32 * h, 32 * w), dtype=torch.uint8, device=dev) if want_labels else None
    with torch.cuda.device(dev):
        check(_native.lib().w2c_upsample32_argmax_confusion(_p(low), M, h, w, lcs, n_classes, _p(gt), _gt_kind(gt), _p(out),
                                                            _p(hist), _p(ws), CONFUSION_WS_PARTIALS if ws is not None else 0,
                                                            _stream(dev)), "w2c_upsample32_argmax_confusion")
    return out


def confusion_matrix(gt, pred, n_classes, hist):
    """hist[n*gt + pred] += 1 on the device (gt u8/i64, pred u8, same shape); hist int64 [n*n]."""
    dev = _need_gpu(gt, pred, hist)
    if gt.shape != pred.shape or pred.dtype != torch.uint8 or hist.dtype != torch.int64 or hist.numel() != n_classes ** 2:
        raise W2CError("confusion_matrix: bad arguments")
    with torch.cuda.device(dev):
        check(_native.lib().w2c_confusion_matrix(_p(gt), _gt_kind(gt), _p(pred), gt.numel(), n_classes, _p(hist),
                                                 _stream(dev)), "w2c_confusion_matrix")
    return hist


def nchw_f32_to_nhwc_bf16(x, cstride=None):
    dev = _need_gpu(x)
    M, C, H, W = x.shape
    cstride = cstride or C
    y = torch.zeros((M, H, W, cstride), dtype=BF16, device=dev)
    with torch.cuda.device(dev):
        check(_native.lib().w2c_nchw_f32_to_nhwc_bf16(_p(x), M, C, H, W, _p(y), cstride, _stream(dev)),
              "w2c_nchw_f32_to_nhwc_bf16")
    return y


def nhwc_bf16_to_nchw_f32(x, channels=None):
    dev = _need_gpu(x)
    M, H, W, cs = x.shape
    C = channels or cs
    y = torch.empty((M, C, H, W), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_native.lib().w2c_nhwc_bf16_to_nchw_f32(_p(x), cs, M, C, H, W, _p(y), _stream(dev)),
              "w2c_nhwc_bf16_to_nchw_f32")
    return y
