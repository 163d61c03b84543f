"""The VAE's LDS-DMA implicit-GEMM convolution (`csrc/conv_mfma.hip`, taken by `tcx_conv3d_cl` for Cin % 64 == 0, Cout >= 128)
vs the oracle (`oracle/vae.py`, `oracle/diffusers_restated.py`) on identical seeded inputs.  GPU only.

Every gather feature is exercised on BOTH tile shapes (Cout >= 256: 256 x 256; Cout = 128..255: 512 x 128): causal context from
the cache and from the replicated first frame (reference autoencoder_magvit.py:138-157), spatial zero padding (:159-160), the
nearest x2 upsample + temporal frame map of CogVideoXUpsample3D, the stride-2 (0,1,0,1)-padded downsample conv, 1x1x1 shortcuts,
the fused residual, ragged M (last tile partly empty), ragged Cout, batch > 1.
Tolerance: one bf16 ulp of the oracle's value + atol 2e-3 (fp32 accumulation order is the only difference), as for the
register-staged kernel in test_kernels_gpu.py; plus agreement with that kernel (TCX_CONV_GENERIC=1 in a child process).
"""
import math
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import diffusers_restated as dr
from oracle import vae as ovae
from oracle.prec import Prec
from tests.test_kernels_gpu import assert_bf16_close, bf, dev, from_cl, to_cl, w_cl

BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from trajectorycrafter_amd import ops as _ops
    return _ops


@pytest.mark.parametrize("Cin,Cout,N,T,H,W", [(128, 128, 1, 3, 20, 30), (256, 128, 1, 2, 17, 23), (128, 256, 1, 3, 16, 18),
                                             (512, 512, 1, 2, 9, 11), (256, 256, 2, 2, 10, 12), (128, 192, 1, 1, 13, 40),
                                             (192, 320, 1, 2, 8, 9), (64, 128, 1, 2, 12, 12)])
def test_causal_conv3d_first_chunk_cache_and_residual(ops, Cin, Cout, N, T, H, W):
    g = torch.Generator().manual_seed(Cin + Cout + T)
    p = Prec("bf16")
    w = bf(torch.randn(Cout, Cin, 3, 3, 3, generator=g) / math.sqrt(27 * Cin))
    b = bf(torch.randn(Cout, generator=g) * 0.1)
    sd = {"c.conv.weight": w.float(), "c.conv.bias": b.float()}
    x1, x2 = bf(torch.randn(N, Cin, T, H, W, generator=g)), bf(torch.randn(N, Cin, 2, H, W, generator=g))
    res = bf(torch.randn(N, Cout, 2, H, W, generator=g))
    cache = {}
    r1 = ovae.causal_conv3d(p, sd, "c.", x1.float(), cache)
    r2 = ovae.causal_conv3d(p, sd, "c.", x2.float(), cache, res=res.float())
    dw, db = dev(w_cl(w)), dev(b)
    d1, d2 = dev(to_cl(x1)), dev(to_cl(x2))
    y1 = ops.conv3d_cl(d1, dw, db)                                         # first chunk: frame 0 replicated as context
    c = torch.cat([d1[:, :1], d1[:, :1], d1], 1)[:, -2:].contiguous()
    y2 = ops.conv3d_cl(d2, dw, db, cache=c, res=dev(to_cl(res)))            # later chunk: cached context + fused shortcut add
    assert_bf16_close(from_cl(y1), r1, atol=2e-3)
    assert_bf16_close(from_cl(y2), r2, atol=2e-3)


@pytest.mark.parametrize("C,T,compress", [(128, 3, True), (256, 2, True), (256, 3, False), (512, 1, True), (128, 4, True)])
def test_upsample_conv(ops, C, T, compress):
    from trajectorycrafter_amd.models.autoencoder_magvit import upsample_t_map
    g = torch.Generator().manual_seed(C + T)
    H, W = 9, 14
    x = bf(torch.randn(1, C, T, H, W, generator=g))
    w = bf(torch.randn(C, C, 3, 3, generator=g) / math.sqrt(9 * C))
    b = bf(torch.randn(C, generator=g) * 0.1)
    ref = dr.upsample3d(Prec("bf16"), {"conv.weight": w.float(), "conv.bias": b.float()}, "", x.float(), compress)
    tm = torch.tensor(upsample_t_map(T, compress), dtype=torch.int32, device="cuda")
    y = ops.conv3d_cl(dev(to_cl(x)), dev(w.permute(0, 2, 3, 1).reshape(C, 1, 3, 3, C).contiguous()), dev(b), ups=1, t_map=tm)
    assert_bf16_close(from_cl(y), ref, atol=2e-3)


@pytest.mark.parametrize("C,T,compress,H,W", [(128, 5, True, 22, 26), (256, 4, True, 10, 12), (256, 3, False, 11, 13), (128, 1, True, 31, 17)])
def test_downsample_conv(ops, C, T, compress, H, W):
    g = torch.Generator().manual_seed(C + T + 40)
    x = bf(torch.randn(1, C, T, H, W, generator=g))
    w = bf(torch.randn(C, C, 3, 3, generator=g) / math.sqrt(9 * C))
    b = bf(torch.randn(C, generator=g) * 0.1)
    ref = dr.downsample3d(Prec("bf16"), {"conv.weight": w.float(), "conv.bias": b.float()}, "", x.float(), compress)
    xcl = dev(to_cl(x))
    if compress and T > 1:
        xcl = ops.avgpool_t(xcl)
    y = ops.conv3d_cl(xcl, dev(w.permute(0, 2, 3, 1).reshape(C, 1, 3, 3, C).contiguous()), dev(b), stride=2, pad=(0, 0),
                      out_hw=((H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1))
    assert_bf16_close(from_cl(y), ref, atol=2e-3)


@pytest.mark.parametrize("Cin,Cout", [(256, 128), (512, 256), (128, 512)])
def test_pointwise_shortcut_conv(ops, Cin, Cout):
    """CogVideoXResnetBlock3D.conv_shortcut (1x1x1, :316-318): K = Cin, one tap, two (or more) K-tiles."""
    g = torch.Generator().manual_seed(Cin)
    x = bf(torch.randn(2, 3, 7, 9, Cin, generator=g))
    w = bf(torch.randn(Cout, Cin, generator=g) / math.sqrt(Cin))
    b = bf(torch.randn(Cout, generator=g))
    res = bf(torch.randn(2, 3, 7, 9, Cout, generator=g))
    y = ops.conv3d_cl(dev(x), dev(w.reshape(Cout, 1, 1, 1, Cin)), dev(b), res=dev(res))
    assert_bf16_close(y, F.linear(x.float(), w.float(), b.float()) + res.float(), atol=2e-3)
    y0 = ops.conv3d_cl(dev(x), dev(w.reshape(Cout, 1, 1, 1, Cin)), None)
    assert_bf16_close(y0, F.linear(x.float(), w.float()), atol=2e-3)


@pytest.mark.parametrize("Cin,Cout,N,T,H,W", [(128, 3, 1, 3, 17, 23), (64, 3, 2, 2, 9, 10), (128, 4, 1, 1, 12, 31), (256, 2, 1, 2, 8, 8)])
def test_narrow_output_conv(ops, Cin, Cout, N, T, H, W):
    """Cout <= 4 (the decoder's conv_out, reference autoencoder_magvit.py:913,953): the dot-product kernel of csrc/conv.hip
    (v_dot2c_f32_bf16, weights in LDS) — first chunk, cached chunk, ragged pixel count, batch 2."""
    g = torch.Generator().manual_seed(Cin + Cout + T)
    p = Prec("bf16")
    w = bf(torch.randn(Cout, Cin, 3, 3, 3, generator=g) / math.sqrt(27 * Cin))
    b = bf(torch.randn(Cout, generator=g) * 0.1)
    sd = {"c.conv.weight": w.float(), "c.conv.bias": b.float()}
    x1, x2 = bf(torch.randn(N, Cin, T, H, W, generator=g)), bf(torch.randn(N, Cin, 2, H, W, generator=g))
    cache = {}
    r1 = ovae.causal_conv3d(p, sd, "c.", x1.float(), cache)
    r2 = ovae.causal_conv3d(p, sd, "c.", x2.float(), cache)
    dw, db = dev(w_cl(w)), dev(b)
    d1, d2 = dev(to_cl(x1)), dev(to_cl(x2))
    y1 = ops.conv3d_cl(d1, dw, db)
    c = torch.cat([d1[:, :1], d1[:, :1], d1], 1)[:, -2:].contiguous()
    y2 = ops.conv3d_cl(d2, dw, db, cache=c)
    assert_bf16_close(from_cl(y1), r1, atol=2e-3)
    assert_bf16_close(from_cl(y2), r2, atol=2e-3)


def test_padding_reads_zeros_not_memory(ops):
    """Border taps must contribute exactly zero: with x = 1 everywhere and w = 1, y counts the in-range taps (27 inside, 18 on an
    edge, 12 in a corner; the causal context is the replicated first frame so time never truncates)."""
    C = 128
    x = torch.ones(1, 2, 6, 7, C, dtype=BF, device="cuda")
    w = torch.ones(C, 3, 3, 3, C, dtype=BF, device="cuda") / C
    y = ops.conv3d_cl(x, w, None).float()
    cnt = F.conv2d(torch.ones(1, 1, 6, 7), torch.ones(1, 1, 3, 3), padding=1)[0, 0] * 3
    assert torch.equal(y[0, 0, :, :, 0].cpu(), cnt) and torch.equal(y[0, 1, :, :, 5].cpu(), cnt)


def test_agrees_with_register_staged_kernel(ops):
    """Same call on the generic kernel (child process with TCX_CONV_GENERIC=1): the two kernels differ in fp32 summation
    order only -> >= 99 % of the outputs bit-identical, none further than 1 ulp + 2e-3."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, torch
sys.path.insert(0, %r)
from trajectorycrafter_amd import ops
g = torch.Generator().manual_seed(3)
x = torch.randn(1, 3, 33, 47, 256, generator=g).to(torch.bfloat16).cuda()
c = torch.randn(1, 2, 33, 47, 256, generator=g).to(torch.bfloat16).cuda()
w = (torch.randn(128, 3, 3, 3, 256, generator=g) / 83).to(torch.bfloat16).cuda()
b = torch.randn(128, generator=g).to(torch.bfloat16).cuda()
torch.save(ops.conv3d_cl(x, w, b, cache=c).cpu(), sys.argv[1])
''' % root
    outs = []
    for tag, env in (("mfma", {}), ("generic", {"TCX_CONV_GENERIC": "1"})):
        path = f"/tmp/tcx_conv_{tag}_{os.getpid()}.pt"
        subprocess.run([sys.executable, "-c", code, path], check=True, env=dict(os.environ, **env), timeout=300)
        outs.append(torch.load(path, weights_only=True).float())
        os.remove(path)
    a, b = outs
    assert float((a == b).float().mean()) > 0.99
    assert bool(((a - b).abs() <= b.abs() * 2.0 ** -7 + 2e-3).all())
