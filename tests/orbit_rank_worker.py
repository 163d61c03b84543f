"""One rank of the configs[3] rehearsal (tests/test_orbits_gpu.py starts W of these; also runs alone with WORLD_SIZE unset).

    python tests/orbit_rank_worker.py OUT_DIR [n_variants]

Builds the tiny pipeline from the committed fixtures on cuda:0 (all ranks share the single GPU of the test box:
TCX_BENCH_SINGLE_DEVICE=1, gloo rendezvous), a seeded synthetic clip (frames + depths), runs `driver.run_orbits` — per variant:
poses -> point-cloud render -> VAE encodes -> 2 CFG/DDIM steps -> decode; then the single all-gather — and writes what this rank
holds to OUT_DIR/rank{r}.safetensors."""
import ast
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests.conftest import load_golden                                                       # noqa: E402

BF = torch.bfloat16


def build(dev):
    from trajectorycrafter_amd.models.autoencoder_magvit import AutoencoderKLCogVideoX
    from trajectorycrafter_amd.models.crosstransformer3d import CrossTransformer3DModel
    from trajectorycrafter_amd.models.pipeline_trajectorycrafter import TrajCrafter_Pipeline
    from trajectorycrafter_amd.models.utils import Warper
    tt, mt = load_golden("transformer_tiny.safetensors")
    tv, mv = load_golden("vae_tiny.safetensors")
    tr = CrossTransformer3DModel(**ast.literal_eval(mt["config"]))
    tr.load_state_dict({k[2:]: v for k, v in tt.items() if k.startswith("w.")}, strict=True)
    vae = AutoencoderKLCogVideoX(**ast.literal_eval(mv["config"]))
    vae.load_state_dict({k[2:]: v for k, v in tv.items() if k.startswith("w.")}, strict=True)
    return TrajCrafter_Pipeline(None, None, vae.to(dev, BF).eval(), tr.to(dev, BF).eval()), Warper(device=str(dev))


def clip(dev, n_frames=9, H=64, W=96):
    g = torch.Generator().manual_seed(2024)
    frames = torch.rand(n_frames, 3, H, W, generator=g) * 2 - 1
    depths = 1.5 + torch.rand(n_frames, 1, H, W, generator=g)
    depths[:, :, 20:40, 30:60] = 0.9                                                        # a near slab: occlusions and holes
    K = torch.tensor([[50.0, 0, W / 2], [0, 50.0, H / 2], [0, 0, 1]]).repeat(n_frames, 1, 1)
    pe = torch.randn(1, 226, 32, generator=g).to(dev, BF)
    ne = torch.randn(1, 226, 32, generator=g).to(dev, BF)
    return frames.to(dev), depths.to(dev), K.to(dev), pe, ne


def main():
    out_dir, n_var = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4
    from safetensors.torch import save_file
    from trajectorycrafter_amd import dp
    from trajectorycrafter_amd.driver import ORBIT_VARIANTS, run_orbits
    rank, world, _ = dp.init_distributed(os.environ.get("TCX_DIST_BACKEND", "gloo"))
    dev = torch.device("cuda:0")
    pipe, warper = build(dev)
    frames, depths, K, pe, ne = clip(dev)
    kw = dict(variants=ORBIT_VARIANTS[:n_var], radius=0.6, K=K, sample_size=(32, 48), prompt_embeds=pe, negative_prompt_embeds=ne,
              num_inference_steps=2, seed=43, mask=True)
    out = run_orbits(pipe, warper, frames, depths, **kw)
    mine = run_orbits(pipe, warper, frames, depths, gather=False, **kw)
    save_file({"gathered": out.cpu(), "mine": mine.cpu()}, os.path.join(out_dir, f"rank{rank}.safetensors"))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
