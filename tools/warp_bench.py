"""Point-cloud render at the reference's render size (49 x 576 x 1024, inference.py:41-42): time per clip and the
algorithmic-byte / atomic rates.  Usage: python tools/warp_bench.py [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trajectorycrafter_amd.models.utils import Warper

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
b, h, w = 49, 576, 1024
g = torch.Generator().manual_seed(0)
dev = torch.device("cuda:0")
frame = (torch.rand(b, 3, h, w, generator=g) * 2 - 1).to(dev)
yy = torch.linspace(0, 1, h)[None, None, :, None]
depth = (2.0 + yy + 0.2 * torch.rand(b, 1, h, w, generator=g)).to(dev)          # ground-plane-like depth + texture
k = torch.tensor([[0.7 * w, 0, w / 2], [0, 0.7 * w, h / 2], [0, 0, 1]])[None].repeat(b, 1, 1)
t1 = torch.eye(4)[None].repeat(b, 1, 1)
t2 = t1.clone()
ang = torch.linspace(0, 0.3, b)
t2[:, 0, 0], t2[:, 0, 2], t2[:, 2, 0], t2[:, 2, 2] = ang.cos(), ang.sin(), -ang.sin(), ang.cos()
t2[:, 0, 3] = torch.linspace(0, 0.5, b)
wp = Warper(device="cuda:0")
for _ in range(2):
    out = wp.forward_warp(frame, None, depth, t1, t2, k, None, False, twice=False, per_frame=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    out = wp.forward_warp(frame, None, depth, t1, t2, k, None, False, twice=False, per_frame=True)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
px = b * h * w
# algorithmic bytes / pixel: project r 4 (depth) w 12 (flow, tdepth); splat r 24 (flow, tdepth, rgb), 20 float atomics;
# memset 20 (padded acc); resolve r 20 w 20  -> 100 B + 20 atomics
print(f"forward_warp {b}x{h}x{w}: {ms:.3f} ms/clip  {px / ms / 1e6:.2f} Gpx/s  {px * 100 / ms / 1e6:.0f} GB/s algorithmic  "
      f"{px * 20 / ms / 1e6:.1f} G atomics/s  coverage {float(out[1].mean()):.3f}")
