"""multiagentperception_amd -- MI355X (gfx950) native When2com forward path.

Drop-in for the reference's ``ptsemseg.models`` boundary (``get_model`` ->
``MIMOcom`` / ``MIMOcomWho`` / ``Single_agent``); the hot path runs in
hand-written HIP kernels behind the C ABI of ``include/w2c_hip.h``.
"""
__version__ = "0.1.0"

import os as _os

# ProcessGroupNCCL's flight recorder is how parallel._watchdog_idle() SEES the watchdog's list before it captures RCCL collectives into a
# HIP graph; the recorder is sized when the process group is created, so the variable has to be in the environment before
# torch.distributed.init_process_group() -- importing this package first is enough.  (Unset and imported too late: the sharded step
# falls back to the 3-segment form with eager collectives and says so in a warning.)
_os.environ.setdefault("TORCH_FR_BUFFER_SIZE", "2000")

# hipGraphLaunch of the HIP runtime bundled with torch 2.10 + rocm7.0 picks the streams of a graph's parallel branches from the streams the
# exec created at instantiate, SKIPPING those that share the launch stream's hardware queue -- without a bounds check.  New streams go to the
# hardware queue with the fewest users; with the default 4 hardware queues a burst of exec destructions (dropping models; the graph
# audition) makes the next execs' streams pile onto one queue, a launch stream on that queue collides with all of them, the loop reads past
# the vector: SIGSEGV in CUDAGraph.replay() (GPUTEST_r04; profiles/r05_capture_crash.txt: native frames, disassembly, and
# tools/r05/hipgraph_oob_repro.py -- pure torch, dies within 750 launches at 4 queues, 720 000 launches without a fault at 16, same speed).
# The runtime reads the variable when it initialises (the first HIP call of the process), so: import this package before touching the GPU.
if "GPU_MAX_HW_QUEUES" not in _os.environ:
    _os.environ["GPU_MAX_HW_QUEUES"] = "16"
    try:
        import torch as _torch
        if _torch.cuda.is_initialized():
            import warnings as _warnings
            _warnings.warn("multiagentperception_amd: the HIP runtime was initialised before this package was imported, so GPU_MAX_HW_QUEUES=16 "
                           "(the mitigation of a hipGraphLaunch out-of-bounds read, see profiles/r05_capture_crash.txt) is NOT in effect; "
                           "export GPU_MAX_HW_QUEUES=16 or import the package first")
    except Exception:                                   # noqa: BLE001
        pass
