"""seq2seq_vc_amd -- MI355X (gfx950) native hot path of unilight/seq2seq-vc.

Sub-packages mirror the reference's name-lookup boundary (`seq2seq_vc.models`, `.losses`,
`.trainers`, `.collaters`, `.schedulers`; reference bin/vc_train.py:304-352,397-443) and sit on
hand-written HIP kernels reached through the C ABI in include/s2svc_hip.h.
"""
__version__ = "0.1.0"
