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
