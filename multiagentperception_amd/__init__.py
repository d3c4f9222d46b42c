"""multiagentperception_amd -- MI355X (gfx950) native When2com forward path.

Drop-in for the reference's ``ptsemseg.models`` boundary (``get_model`` ->
``MIMOcom`` / ``MIMOcomWho`` / ``Single_agent``); the hot path runs in
hand-written HIP kernels behind the C ABI of ``include/w2c_hip.h``.
"""
__version__ = "0.1.0"

