"""VAE timing on the GPU box: full-size decode (13 latent frames -> 49 x 480 x 720), the two encodes that precede a clip when the
conditioning comes from pixels (49-frame masked render + 10-frame reference), and the dominant conv shapes in isolation
(TFLOP/s = 2 * taps * Cin * Cout * positions / time).  TCX_CONV_GENERIC=1 times the register-staged kernel instead.
usage: python tools/vae_bench.py [decode] [encode] [shapes]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajectorycrafter_amd import init_weights as iw, ops
from trajectorycrafter_amd.models.autoencoder_magvit import AutoencoderKLCogVideoX

BF = torch.bfloat16
dev = torch.device("cuda:0")
what = sys.argv[1:] or ["decode", "encode", "shapes"]
tag = "generic" if os.environ.get("TCX_CONV_GENERIC") == "1" else "mfma"


def timed(fn, reps=3, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


if "shapes" in what:
    g = torch.Generator(device=dev).manual_seed(0)
    for name, Cin, Cout, T, H, W, kT, ups, res in (("up3 resnet 128->128 3x3x3", 128, 128, 8, 480, 720, 3, 0, True),
                                                  ("up3 resnet0 256->128 3x3x3", 256, 128, 8, 480, 720, 3, 0, False),
                                                  ("up3 shortcut 256->128 1x1x1", 256, 128, 8, 480, 720, 1, 0, False),
                                                  ("up2 resnet 256->256 3x3x3", 256, 256, 8, 240, 360, 3, 0, True),
                                                  ("up2 upsample 256->256 3x3 (x2)", 256, 256, 8, 240, 360, 1, 1, False),
                                                  ("up1 resnet 256->256 3x3x3", 256, 256, 4, 120, 180, 3, 0, True),
                                                  ("up1 resnet0 512->256 3x3x3", 512, 256, 4, 120, 180, 3, 0, False),
                                                  ("up0 resnet 512->512 3x3x3", 512, 512, 2, 60, 90, 3, 0, True),
                                                  ("conv_out 128->3 3x3x3", 128, 3, 8, 480, 720, 3, 0, False)):
        k = 3
        x = torch.randn(1, T, H, W, Cin, device=dev, dtype=BF, generator=g)
        w = torch.randn(Cout, kT, k if kT == 3 or ups else 1, k if kT == 3 or ups else 1, Cin, device=dev, dtype=BF, generator=g) / (27 * Cin) ** 0.5
        b = torch.randn(Cout, device=dev, dtype=BF, generator=g)
        cache = torch.randn(1, 2, H, W, Cin, device=dev, dtype=BF, generator=g) if kT == 3 else None
        Ho, Wo = H << ups, W << ups
        r = torch.randn(1, T, Ho, Wo, Cout, device=dev, dtype=BF, generator=g) if res else None
        tm = torch.arange(T, dtype=torch.int32, device=dev) if ups else None
        dt = timed(lambda: ops.conv3d_cl(x, w, b, cache=cache, res=r, ups=ups, t_map=tm), reps=5, warm=2)
        taps = w.shape[1] * w.shape[2] * w.shape[3]
        flop = 2.0 * taps * Cin * Cout * T * Ho * Wo
        print(f"[{tag}] {name:34s} {dt * 1e3:8.3f} ms  {flop / dt / 1e12:7.1f} TFLOP/s", flush=True)
        del x, w, b, cache, r

if "gn" in what:
    g = torch.Generator(device=dev).manual_seed(0)
    for name, C, T, H, W in (("up3 128ch 8x480x720", 128, 8, 480, 720), ("up2 256ch 8x240x360", 256, 8, 240, 360),
                             ("up1 512ch 4x120x180", 512, 4, 120, 180), ("enc 128ch 9x480x720", 128, 9, 480, 720)):
        x = torch.randn(1, T, H, W, C, device=dev, dtype=BF, generator=g)
        gw, gb = torch.randn(C, device=dev, dtype=BF, generator=g), torch.randn(C, device=dev, dtype=BF, generator=g)
        ytab = torch.randn(1, 2, 60, 90, C, device=dev, dtype=BF, generator=g)
        btab = torch.randn(1, 2, 60, 90, C, device=dev, dtype=BF, generator=g)
        tm = torch.tensor([(j * 2) // T for j in range(T)], dtype=torch.int32, device=dev)
        nbytes = x.numel() * 2
        ts = timed(lambda: ops.groupnorm_stats(x, 32, 1e-6), reps=10, warm=2)
        st = ops.groupnorm_stats(x, 32, 1e-6)
        ta = timed(lambda: ops.groupnorm_apply(x, st, gw, gb, 32, ytab, btab, tm, True), reps=10, warm=2)
        tp = timed(lambda: ops.groupnorm_apply(x, st, gw, gb, 32, silu=True), reps=10, warm=2)
        print(f"[gn] {name:22s} stats {ts * 1e6:7.1f} us ({nbytes / ts / 1e12:4.2f} TB/s)  apply+spatialnorm+silu {ta * 1e6:7.1f} us "
              f"({2 * nbytes / ta / 1e12:4.2f} TB/s)  apply+silu {tp * 1e6:7.1f} us ({2 * nbytes / tp / 1e12:4.2f} TB/s)", flush=True)
        del x

vae = None
if "decode" in what or "encode" in what:
    with torch.device("meta"):
        vae = AutoencoderKLCogVideoX()
    vae.load_state_dict(iw.random_state_dict(iw.vae_param_shapes(dict(vae.config)), seed=1, dtype=BF, device=dev), strict=True, assign=True)
    vae.eval()
if "decode" in what:
    z = torch.randn(1, 16, 13, 60, 90, device=dev, dtype=BF)
    dt = timed(lambda: vae.decode_to_frames(z, 1 / 1.15258426), reps=3)
    print(f"[{tag}] decode 13 latent frames -> 49 x 480 x 720: {dt * 1e3:.1f} ms  ({3.150e14 / dt / 1e12:.0f} TFLOP/s on 3.150e14 FLOP)", flush=True)
if "encode" in what:
    x49 = torch.rand(1, 3, 49, 480, 720, device=dev, dtype=BF) * 2 - 1
    x10 = x49[:, :, :10].contiguous()
    dt49 = timed(lambda: vae.encode(x49), reps=2)
    dt10 = timed(lambda: vae.encode(x10), reps=2)
    print(f"[{tag}] encode 49 frames: {dt49 * 1e3:.1f} ms; encode 10 reference frames: {dt10 * 1e3:.1f} ms", flush=True)
