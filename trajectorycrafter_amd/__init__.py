"""trajectorycrafter_amd — MI355X-native TrajectoryCrafter denoising hot path.

`trajectorycrafter_amd.models.*` mirrors the reference's `models/*` modules (same class names,
signatures, config and state-dict keys); `trajectorycrafter_amd.ops` is the torch-tensor front end
of the C ABI in include/tcx_hip.h (libtcx_hip.so, hand-written HIP for gfx950).
"""
__version__ = "0.1.0"
