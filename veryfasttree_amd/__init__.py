"""veryfasttree_amd — MI355X-native backend for VeryFastTree's profile-operations hot path.

Only what the path needs lives here: csrc/ (HIP kernels + the C ABI of include/vft_hip.h), host/ (the C++
`HipOperations<Precision>` slot the reference plugs backends into), backend.py (ctypes binding) and synth.py
(synthetic alignments for tests and bench).
"""
from .backend import HipProfileOps, VftError, load_library  # noqa: F401
