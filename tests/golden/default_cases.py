"""The default-width fixture cases: configurations, seeds and input tensors, shared by the generator
(tests/golden/make_golden.py default, build container) and the tests (CPU + GPU box).

Weights (0.31 B transformer / 0.22 B VAE parameters) and the wide inputs are NOT stored in the fixture: both sides
regenerate them from `trajectorycrafter_amd.init_weights.hashed_state_dict` / `hashed_normal`, an integer-hash
stream that is bit-identical on every host; the fixture stores their sha256 digest and the reference's OUTPUTS.
"""
from __future__ import annotations

import torch

from trajectorycrafter_amd import init_weights as iw

# 5B geometry, 2 of the 42 layers (crosstransformer3d.py:459-491 defaults + the checkpoint's cross-attention settings)
DEFAULT_TR2 = dict(num_attention_heads=48, attention_head_dim=64, in_channels=33, out_channels=16, num_layers=2,
                   text_embed_dim=4096, time_embed_dim=512, max_text_seq_length=226, sample_width=24, sample_height=16,
                   sample_frames=9, use_rotary_positional_embeddings=True, is_train_cross=True,
                   cross_attn_in_channels=16, cross_attn_interval=2, cross_attn_dim_head=128, cross_attn_num_heads=16)
DEFAULT_VAE = dict()                     # AutoencoderKLCogVideoX(): 128 / 256 / 256 / 512, 3 layers per block, 32 groups

TR_SEED, VAE_SEED, IN_SEED = 41, 42, 43
TR_LATENT = (2, 3, 16, 16, 24)           # [B, T, C, h, w] -> 3 x 8 x 12 = 288 video tokens + 226 text tokens
TR_REF_FRAMES = 2                        # 2 x 8 x 12 = 192 reference tokens
VAE_Z = (1, 16, 5, 4, 6)                 # -> 17 frames 32 x 48
VAE_VIDEO = (1, 3, 17, 32, 48)


def transformer_weights():
    return iw.hashed_state_dict(iw.transformer_param_shapes(DEFAULT_TR2), TR_SEED)


def vae_weights():
    return iw.hashed_state_dict(iw.vae_param_shapes(DEFAULT_VAE), VAE_SEED)


def bf16r(t):
    return t.to(torch.bfloat16).float()


def transformer_inputs():
    """bf16-representable inputs: the fp32 and the bf16 runs see the same numbers."""
    B, T, C, h, w = TR_LATENT
    n = lambda shape, stream: bf16r(iw.hashed_normal(shape, IN_SEED, stream))
    return dict(hidden_states=n((B, T, C, h, w), 0), encoder_hidden_states=n((B, 226, 4096), 1),
                inpaint_latents=n((B, T, C + 1, h, w), 2), cross_latents=n((B, TR_REF_FRAMES, C, h, w), 3),
                timestep=torch.tensor([781, 781]))


def vae_inputs():
    z = bf16r(iw.hashed_normal(VAE_Z, IN_SEED, 10))
    video = bf16r((iw.hashed_normal(VAE_VIDEO, IN_SEED, 11) * 0.4).clamp(-1.0, 1.0))   # in [-1, 1] like normalised frames; exact IEEE ops only
    return dict(z=z, video=video)


PIPE_GLOBAL_SEED = 177                   # the reference draws the reference-latent posterior sample from the GLOBAL rng (:886)


def pipeline_inputs():
    """9 frames 32x48 (latent [1,3,16,4,6]): video / reference in [0, 1], a blocky 0 / 255 mask (first frame unmasked), prompt
    embeddings at the 5B text width, seeded noise — all from the host-independent stream, bf16-representable where they are cast."""
    Fv, H, W = 9, 32, 48
    video = (iw.hashed_normal((1, 3, Fv, H, W), IN_SEED, 20) * 0.25 + 0.5).clamp(0.0, 1.0)
    blocks = (iw.hashed_normal((1, 1, Fv, H // 8, W // 8), IN_SEED, 21) > 0.5).float()
    mask = blocks.repeat_interleave(8, 3).repeat_interleave(8, 4) * 255.0
    mask[:, :, 0] = 0
    return dict(video=video, mask_video=mask, reference=video[:, :, :5].clone(),
                prompt_embeds=bf16r(iw.hashed_normal((1, 226, 4096), IN_SEED, 22)),
                negative_prompt_embeds=bf16r(iw.hashed_normal((1, 226, 4096), IN_SEED, 23)),
                latents0=bf16r(iw.hashed_normal((1, 3, 16, H // 8, W // 8), IN_SEED, 24)))
