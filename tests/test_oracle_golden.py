"""Oracle vs the committed golden fixtures (generated from the reference's own classes by
tests/golden/make_golden.py).  CPU only."""
import ast

import torch

from oracle import pipeline as opl
from oracle import transformer as otr
from oracle import vae as ovae
from oracle import diffusers_restated as dr
from oracle.prec import Prec


def _weights(t):
    return {k[2:]: v.float() for k, v in t.items() if k.startswith("w.")}


def _close(a, b, rtol=2e-4, atol=2e-5):
    torch.testing.assert_close(a.float(), b.float(), rtol=rtol, atol=atol)


def test_transformer_forward_matches_reference(golden):
    t, meta = golden("transformer_tiny.safetensors")
    cfg = ast.literal_eval(meta["config"])
    sd = _weights(t)
    taps = {}
    out = otr.transformer_forward(sd, cfg, t["hidden_states"], t["encoder_hidden_states"], t["timestep"],
                                  t["inpaint_latents"], t["cross_latents"], (t["rope_cos"], t["rope_sin"]),
                                  prec="fp32", taps=taps)
    _close(out, t["out_sample"])
    _close(taps["patch_embed"], t["tap_patch_embed"][:, 10:])


def test_transformer_components_match_reference(golden):
    t, meta = golden("transformer_tiny.safetensors")
    cfg = dict(otr.DEFAULT_CONFIG)
    cfg.update(ast.literal_eval(meta["config"]))
    sd = _weights(t)
    p = Prec("fp32")
    D = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    emb = dr.timestep_embedding(p, sd, "time_embedding.", dr.timesteps_proj(t["timestep"], D))
    pe = t["tap_patch_embed"]
    h, e = otr.cogvideox_block(p, sd, "transformer_blocks.0.", pe[:, 10:], pe[:, :10], emb,
                               (t["rope_cos"], t["rope_sin"]), cfg["num_attention_heads"], cfg["norm_eps"])
    _close(h, t["tap_block0_hidden"])
    _close(e, t["tap_block0_encoder"])
    ref_tok = otr.patch_embed_video(p, sd["ref_patch_embed.proj.weight"], sd["ref_patch_embed.proj.bias"],
                                    t["cross_latents"], 2)
    _close(ref_tok, t["tap_ref_tokens"])
    ca = otr.perceiver_cross_attention(p, sd, "perceiver_cross_attention.0.", t["tap_ref_tokens"],
                                       t["tap_block0_hidden"], cfg["cross_attn_num_heads"], cfg["cross_attn_dim_head"])
    _close(ca, t["tap_cross0"])


def test_transformer_sincos_branch_matches_reference(golden):
    """The non-rotary (2B) position-embedding branch (crosstransformer3d.py:752-784) run by the reference itself: at the configured
    sample size and at a smaller latent with fewer frames (trilinear resize + row cut).  The sincos table is the oracle's
    restatement of diffusers' (absent offline: that part is parity-unpinned, KAT in test_oracle_kat.py)."""
    t, meta = golden("transformer_sincos_tiny.safetensors")
    tw, _ = golden(meta["weights"])
    cfg = ast.literal_eval(meta["config"])
    assert cfg["use_rotary_positional_embeddings"] is False
    sd = _weights(tw)
    for tag in "ab":
        out = otr.transformer_forward(sd, cfg, t[f"hidden_states_{tag}"], t[f"encoder_hidden_states_{tag}"], t[f"timestep_{tag}"],
                                      t[f"inpaint_latents_{tag}"], t[f"cross_latents_{tag}"], None)
        _close(out, t[f"out_sample_{tag}"])
    rot = otr.transformer_forward(sd, dict(cfg, use_rotary_positional_embeddings=True), t["hidden_states_a"], t["encoder_hidden_states_a"],
                                  t["timestep_a"], t["inpaint_latents_a"], t["cross_latents_a"], None)
    assert not torch.allclose(rot, t["out_sample_a"], atol=1e-2)                   # the table matters


def test_vae_decode_encode_match_reference(golden):
    t, meta = golden("vae_tiny.safetensors")
    cfg = ast.literal_eval(meta["config"])
    sd = _weights(t)
    _close(ovae.vae_decode(sd, cfg, t["z"]), t["decoded"])
    _close(ovae.vae_decode(sd, cfg, t["z"][:, :, :1]), t["decoded_single_frame"])
    post = ovae.vae_encode(sd, cfg, t["video"])
    _close(post.mean, t["enc_mean"])
    _close(post.logvar, t["enc_logvar"])


def test_vae_tiled_decode_matches_reference(golden):
    """oracle.vae.vae_tiled_decode vs the reference's own enable_tiling() + decode (autoencoder_magvit.py:1303-1392): 3 x 3 ragged
    tiles, in-place seam blends, default and caller-set tile geometry.  fp32: bit-for-bit."""
    t, meta = golden("vae_tiled_tiny.safetensors")
    tv, _ = golden(meta["weights"])
    cfg = ast.literal_eval(meta["config"])
    sd = _weights(tv)
    d = ovae.vae_tiled_decode(sd, cfg, t["z"])
    assert d.shape == (1, 3, 17, 96, 80) and torch.equal(d, t["decoded_tiled"])
    d2 = ovae.vae_tiled_decode(sd, cfg, t["z"][:, :, :3], tile_sample_min_height=64, tile_sample_min_width=64,
                               tile_overlap_factor_height=0.25, tile_overlap_factor_width=0.25)
    assert torch.equal(d2, t["decoded_tiled_64"])
    assert not torch.allclose(d, ovae.vae_decode(sd, cfg, t["z"]), atol=1e-3)          # tiling changes the result (zero borders per tile)
    # a latent no larger than the tile falls through to the plain decode (:1222-1225)
    small = t["z"][:, :, :, :6, :5]
    assert torch.equal(ovae.vae_tiled_decode(sd, cfg, small), ovae.vae_decode(sd, cfg, small))


def test_pipeline_matches_reference(golden):
    tp, meta = golden("pipeline_tiny.safetensors")
    tt, mt = golden("transformer_tiny.safetensors")
    tv, mv = golden("vae_tiny.safetensors")
    tr_cfg, vae_cfg = ast.literal_eval(mt["config"]), ast.literal_eval(mv["config"])
    torch.manual_seed(int(meta["global_seed"]))      # the reference samples ref latents from the global RNG
    kw = dict(prompt_embeds=tp["prompt_embeds"], negative_prompt_embeds=tp["negative_prompt_embeds"],
              video=tp["video"], mask_video=tp["mask_video"], reference=tp["reference"], height=32, width=48,
              latents=tp["latents0"], num_inference_steps=2, guidance_scale=6.0, num_frames=9)
    lat = opl.pipeline_call(_weights(tt), tr_cfg, _weights(tv), vae_cfg, output_type="latent", **kw)
    _close(lat, tp["latents_out"], rtol=1e-3, atol=1e-4)
    torch.manual_seed(int(meta["global_seed"]))
    frames = opl.pipeline_call(_weights(tt), tr_cfg, _weights(tv), vae_cfg, **kw)
    _close(frames, tp["frames"], rtol=1e-3, atol=1e-4)
    assert frames.shape == (1, 3, 9, 32, 48) and float(frames.min()) >= 0 and float(frames.max()) <= 1


def test_forward_warp_matches_reference(golden):
    """oracle.warp vs the reference's own Warper.forward_warp.  fp32 sums in a different order:
    rtol 2e-5 (flow reaches 3.7e3 px for points pushed behind the camera) / atol 5e-5."""
    from oracle import warp
    t, _ = golden("warp_tiny.safetensors")
    warped, mask2, wdepth, flow = warp.forward_warp(t["frame"], None, t["depth"], t["t1"], t["t2"], t["K"])
    assert torch.equal(mask2, t["mask2"])
    _close(flow, t["flow"], rtol=2e-5, atol=5e-5)
    _close(warped, t["warped"], rtol=2e-5, atol=5e-5)
    _close(wdepth, t["warped_depth"], rtol=2e-5, atol=5e-5)
    assert 0.1 < float(mask2[1].mean()) < 0.3 and float(mask2[0].mean()) > 0.7     # the fixture has holes and occlusion
    # twice=True (reference :294-347): warp there, splat the flow, splat frame and depth back along the negated warped flow
    tw_frame, tw_mask, tw_depth, none = warp.forward_warp_twice(t["frame"], None, t["depth"], t["t1"], t["t2"], t["K"])
    assert none is None and torch.equal(tw_mask, t["twice_mask"])
    _close(tw_frame, t["twice_frame"], rtol=2e-5, atol=5e-5)
    _close(tw_depth, t["twice_depth"], rtol=2e-5, atol=5e-5)
    assert 0.5 < float(tw_mask[0].mean()) < 1.0                                      # back in the source view, with the occlusions as holes
