"""trajectorycrafter_amd — MI355X-native TrajectoryCrafter denoising hot path.

`trajectorycrafter_amd.models.*` mirrors the reference's `models/*` modules (same class names,
signatures, config and state-dict keys); `trajectorycrafter_amd.ops` is the torch-tensor front end
of the C ABI in include/tcx_hip.h (libtcx_hip.so, hand-written HIP for gfx950).
"""
import os as _os

# dmabuf IPC (see dp.py): must be in the environment before the HSA runtime starts, whoever launched this process
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

__version__ = "0.1.0"
