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


def test_header_is_plain_c(tmp_path):
    """include/tcx_hip.h compiles as C99 on its own (no C++, no HIP, no torch types in the boundary)."""
    import subprocess
    c = tmp_path / "hdr.c"
    c.write_text('#include "tcx_hip.h"\nint main(void) { return (int)sizeof(&tcx_attn_fwd) * 0 + TCX_OK; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(c)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def _build_c_host(tmp_path):
    import subprocess
    from trajectorycrafter_amd import build
    build.build(verbose=False)
    exe = str(tmp_path / "cabi_smoke")
    libdir = os.path.join(ROOT, "trajectorycrafter_amd")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cabi_smoke.cpp"),
                        "-L", libdir, "-ltcx_hip", f"-Wl,-rpath,{libdir}", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def test_c_host_program_links(tmp_path):
    """tests/cabi_smoke.cpp — a host program with no Python and no torch — compiles against the header and links against
    libtcx_hip.so (hipcc cross-compiles on a CPU-only host); it is run on the GPU by test_c_host_program below."""
    assert os.path.exists(_build_c_host(tmp_path))


import pytest  # noqa: E402


@pytest.mark.gpu
def test_c_host_program(tmp_path):
    """The C ABI from a foreign host: hipMalloc'd buffers, tcx_gemm_bf16 (ragged M, K-tail, bias + GELU) and tcx_layernorm_modulate
    checked against host loops, a rejected call read back through the return code and tcx_last_error_string()."""
    import subprocess
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = subprocess.run([_build_c_host(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "cabi_smoke ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    print(r.stdout)
