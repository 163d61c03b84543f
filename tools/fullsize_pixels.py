"""Full-size end-to-end call from pixels (reference signature): video/mask/reference -> frames, 480x720, 49 frames,
2 denoise steps, reduced layer count.  Robustness / memory check of the HIP VAE encoder at full resolution."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
BF = torch.bfloat16
args = bench.parse()
args.layers = 4
pipe, _ = bench.build_models(args, torch.device("cuda"))
g = torch.Generator().manual_seed(0)
video = torch.rand(1, 3, 49, 480, 720, generator=g)
mask = (torch.rand(1, 1, 49, 60, 90, generator=g) < 0.3).float().repeat_interleave(8, 3).repeat_interleave(8, 4) * 255
mask[:, :, 0] = 0
ref = video[:, :, :10].clone()
pe = torch.randn(1, 226, 4096, generator=g).to(BF)
torch.cuda.synchronize(); t = time.time()
out = pipe(prompt=None, prompt_embeds=pe, negative_prompt_embeds=pe, height=480, width=720, num_frames=49,
           num_inference_steps=2, guidance_scale=6.0, video=video, mask_video=mask, reference=ref,
           generator=torch.Generator().manual_seed(43)).videos
torch.cuda.synchronize()
print("frames", tuple(out.shape), out.dtype, out.device, float(out.min()), float(out.max()), "finite", bool(torch.isfinite(out).all()),
      "wall %.2f s" % (time.time() - t), "peak mem GB %.1f" % (torch.cuda.max_memory_allocated() / 2**30))
t = time.time()
post = pipe.vae.encode((video * 2 - 1).to("cuda", BF)).latent_dist
torch.cuda.synchronize()
print("encode 49f 480x720: %.3f s, mean shape" % (time.time() - t), tuple(post.mean.shape))
