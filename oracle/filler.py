"""TEST INFRASTRUCTURE -- the deterministic weight filler / synthetic frame generator the golden fixtures were made with.
The implementation lives in multiagentperception_amd/synth.py (bench.py needs the same generator for its measured path and
must not import oracle/ there); this module re-exports it under the name the oracle, the fixture generators and the tests use."""
from multiagentperception_amd.synth import *  # noqa: F401,F403
from multiagentperception_amd.synth import (_gain, _splitmix64, _synthetic_u8_bgr, apply_to_module, canonical_name,  # noqa: F401
                                            fill_array, fill_state_dict, synthetic_frames, synthetic_frames_u8,
                                            synthetic_labels, uniform_pm1)
