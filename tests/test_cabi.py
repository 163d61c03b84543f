"""The C-ABI library loads on a CPU-only host and exports every symbol include/tcx_hip.h declares;
argument validation works without touching a GPU (no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "tcx_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tcx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from trajectorycrafter_amd import _lib, build
    build.build(verbose=False)
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/tcx_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(names)
    assert lib.tcx_version() == 1


def test_argument_validation_without_gpu():
    from trajectorycrafter_amd import _lib
    lib = _lib.load()
    # null pointers / bad head dim are rejected before any launch
    rc = lib.tcx_attn_fwd(None, None, None, None, 1, 1, 8, 8, 64, *([64] * 12), 1.0, 0, None, 0, None)
    assert rc == -4 and b"null" in lib.tcx_last_error_string()
    buf = ctypes.create_string_buffer(4096)
    p = (ctypes.addressof(buf) + 15) & ~15
    rc = lib.tcx_attn_fwd(p, p, p, p, 1, 1, 8, 8, 32, *([64] * 12), 1.0, 0, None, 0, None)
    assert rc == -1 and b"head dim" in lib.tcx_last_error_string()
    rc = lib.tcx_attn_fwd(p + 2, p, p, p, 1, 1, 8, 8, 64, *([64] * 12), 1.0, 0, None, 0, None)
    assert rc == -3
    rc = lib.tcx_layernorm_modulate(p, p, 1, 4, 12, 48, 48, None, None, None, None, None, None, 0, 0, 1e-5, None)
    assert rc == -1                       # C % 8 != 0
    rc = lib.tcx_conv3d_cl(p, None, p, None, None, p, 1, 2, 4, 4, 12, 8, 3, 3, 3, 2, 0, 1, 1, 1, 4, 4, None, None)
    assert rc == -1                       # Cin % 8 != 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from trajectorycrafter_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _lib.load()
    except _lib.TcxError as e:
        assert "only compute path" in str(e)
    else:
        raise AssertionError("expected TcxError")


def test_ops_reject_cpu_tensors():
    import torch
    from trajectorycrafter_amd import ops
    x = torch.zeros(1, 4, 1, 64, dtype=torch.bfloat16)
    try:
        ops.attn_fwd(x, x, x, 1.0)
    except ops.TcxError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("expected TcxError")
