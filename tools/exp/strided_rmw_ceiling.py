"""Is the access pattern of tcx_qk_layernorm_rope (in-place read-modify-write of the q and k thirds of the fused QKV rows: 12 KB touched
of every 18 KB) inherently slower than a contiguous copy?  torch elementwise kernels as the probe."""
import torch
dev = torch.device("cuda:0")
B, S, H, D = 2, 17776, 48, 64
qkv = torch.randn(B, S, 3 * H * D, device=dev, dtype=torch.bfloat16)
qk = qkv[:, :, : 2 * H * D]                       # the strided view the kernel works on (q | k of every row)
cont = torch.empty(B, S, 2 * H * D, device=dev, dtype=torch.bfloat16)
src = torch.randn_like(cont)
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
nbytes = 2 * cont.numel() * 2
for name, fn in (("contiguous out-of-place  add(src, 0, out=cont)", lambda: torch.add(src, 0, out=cont)),
                 ("contiguous in-place      cont.add_(0)", lambda: cont.add_(0)),
                 ("strided in-place         qk.add_(0)   (the kernel's pattern)", lambda: qk.add_(0)),
                 ("strided -> contiguous    add(qk, 0, out=cont)", lambda: torch.add(qk, 0, out=cont))):
    us = t(fn)
    print(f"{name}: {us:.1f} us = {nbytes / us / 1e6:.2f} TB/s")
