"""Oracle vs the REFERENCE's own classes at the widths the product dispatches on.  CPU only.

tests/golden/transformer_default.safetensors / vae_default.safetensors / pipeline_tiny_bf16.safetensors were written by
`tests/golden/make_golden.py default`: the reference's `CrossTransformer3DModel` (2 layers at the 5B geometry: 48 x 64 heads,
cross-attention 16 x 128, text 226 x 4096) and `AutoencoderKLCogVideoX()` (128 / 256 / 256 / 512) over plain-torch diffusers
stand-ins (tests/golden/diffusers_plain.py — no oracle code under the reference's classes), once in fp32 and once as
`.to(bfloat16)` EAGER on the CPU, i.e. the reference's own bf16 execution.  Weights and the wide inputs are regenerated here
from a host-independent integer-hash stream and checked against the stored sha256.

  * oracle fp32  == the reference's fp32 outputs (rtol 2e-4 as for the tiny fixtures; measured 1.4e-6 / bit-equal);
  * oracle `Prec("bf16_ref")` IS a faithful model of the reference's eager bf16 run: same error against fp32 (within 10 %) and
    no farther from the eager run than two independent bf16 executions are from each other.
"""
import ast
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import default_cases as dc                                   # noqa: E402

from oracle import pipeline as opl                          # noqa: E402
from oracle import transformer as otr                       # noqa: E402
from oracle import vae as ovae                              # noqa: E402
from trajectorycrafter_amd import init_weights as iw        # noqa: E402


def _close(a, b, rtol=2e-4, atol=2e-5):
    torch.testing.assert_close(a.float(), b.float(), rtol=rtol, atol=atol)


def _mean(a, b):
    return float((a.float() - b.float()).abs().mean())


def check_bf16_model(what, emulated, eager, exact, lo=0.9, hi=1.1, mutual=1.5):
    """`emulated` (oracle bf16_ref) models `eager` (the reference's real bf16 run): equal error against the fp32 result within
    [lo, hi] and mutual distance <= `mutual` x that error (two uncorrelated bf16 executions sit at about sqrt(2) x)."""
    e_em, e_eg, d = _mean(emulated, exact), _mean(eager, exact), _mean(emulated, eager)
    msg = f"{what}: mean|bf16_ref - fp32| {e_em:.4e}  mean|eager - fp32| {e_eg:.4e}  mean|bf16_ref - eager| {d:.4e}"
    print(msg)
    assert lo * e_eg <= e_em <= hi * e_eg, msg
    assert d <= mutual * e_eg, msg
    return e_em, e_eg, d


@pytest.fixture(scope="module")
def tr_case(golden):
    t, meta = golden("transformer_default.safetensors")
    sd = dc.transformer_weights()
    assert iw.state_dict_digest(sd) == meta["weights_digest"], "the hashed weight stream differs on this host"
    cfg = ast.literal_eval(meta["config"])
    assert cfg == dc.DEFAULT_TR2
    return t, sd, cfg, dc.transformer_inputs()


@pytest.fixture(scope="module")
def vae_case(golden):
    t, meta = golden("vae_default.safetensors")
    sd = dc.vae_weights()
    assert iw.state_dict_digest(sd) == meta["weights_digest"], "the hashed weight stream differs on this host"
    return t, sd, ast.literal_eval(meta["config"]), dc.vae_inputs()


def _forward(sd, cfg, x, t, prec, taps=None):
    return otr.transformer_forward(sd, cfg, x["hidden_states"], x["encoder_hidden_states"], x["timestep"], x["inpaint_latents"],
                                   x["cross_latents"], (t["rope_cos"], t["rope_sin"]), prec=prec, taps=taps)


def test_transformer_5b_geometry_fp32_matches_reference(tr_case):
    t, sd, cfg, x = tr_case
    taps = {}
    out = _forward(sd, cfg, x, t, "fp32", taps)
    _close(out, t["out_sample"])
    # the per-module taps the reference's own sub-modules produced (strided slices): block 0 and the dh = 128 cross-attention
    from oracle import diffusers_restated as dr
    from oracle.prec import Prec
    p = Prec("fp32")
    emb = dr.timestep_embedding(p, sd, "time_embedding.", dr.timesteps_proj(x["timestep"], 3072))
    _close(emb, t["tap_temb"])
    pe_v = otr.patch_embed_video(p, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"],
                                 torch.cat([x["hidden_states"], x["inpaint_latents"]], 2), 2)
    pe_t = p.linear(x["encoder_hidden_states"], sd["patch_embed.text_proj.weight"], sd["patch_embed.text_proj.bias"])
    h, e = otr.cogvideox_block(p, sd, "transformer_blocks.0.", pe_v, pe_t, emb, (t["rope_cos"], t["rope_sin"]), 48, 1e-5)
    _close(h[:, ::4, ::16], t["tap_block0_hidden"])
    _close(e[:, ::4, ::16], t["tap_block0_encoder"])
    ref_tok = otr.patch_embed_video(p, sd["ref_patch_embed.proj.weight"], sd["ref_patch_embed.proj.bias"], x["cross_latents"], 2)
    ca = otr.perceiver_cross_attention(p, sd, "perceiver_cross_attention.0.", ref_tok, h, 16, 128)
    _close(ca[:, ::3, ::8], t["tap_cross0"])


def test_transformer_bf16_ref_models_the_reference_eager_bf16(tr_case):
    t, sd, cfg, x = tr_case
    out = _forward(sd, cfg, x, t, "bf16_ref")
    e_em, e_eg, d = check_bf16_model("2-layer 5B-geometry transformer", out, t["out_sample_bf16_eager"], t["out_sample"])
    assert d <= 0.75 * e_eg                                   # two blocks deep the roundings still line up (measured 0.38 x)
    # the HIP path's rounding contract is the more accurate of the two (fewer rounding points), never the less accurate
    con = _forward(sd, cfg, x, t, "bf16")
    assert _mean(con, t["out_sample"]) <= 1.05 * e_eg


def test_vae_default_width_fp32_matches_reference(vae_case):
    t, sd, cfg, x = vae_case
    _close(ovae.vae_decode(sd, cfg, x["z"]), t["decoded"])
    _close(ovae.vae_decode(sd, cfg, x["z"][:, :, :1]), t["decoded_single_frame"])
    post = ovae.vae_encode(sd, cfg, x["video"])
    _close(post.mean, t["enc_mean"])
    _close(post.logvar, t["enc_logvar"])


def test_vae_bf16_ref_models_the_reference_eager_bf16(vae_case):
    t, sd, cfg, x = vae_case
    dec = ovae.vae_decode(sd, cfg, x["z"], prec="bf16_ref")
    check_bf16_model("default-width VAE decode (17 frames 32x48)", dec, t["decoded_bf16_eager"], t["decoded"])
    dec1 = ovae.vae_decode(sd, cfg, x["z"][:, :, :1], prec="bf16_ref")
    check_bf16_model("default-width VAE decode (T = 1)", dec1, t["decoded_single_frame_bf16_eager"], t["decoded_single_frame"], lo=0.85, hi=1.15)
    post = ovae.vae_encode(sd, cfg, x["video"], prec="bf16_ref")
    check_bf16_model("default-width VAE encode mean", post.mean, t["enc_mean_bf16_eager"], t["enc_mean"])


def test_tiny_pipeline_bf16_ref_models_the_reference_eager_bf16(golden):
    """The 2-step CFG + decode pipeline of pipeline_tiny.safetensors run by the reference in eager bf16 (DESIGN §4's table quotes
    these frames): the oracle's `bf16_ref` pipeline has the same error against the reference's fp32 frames."""
    tb, meta = golden("pipeline_tiny_bf16.safetensors")
    tp, _ = golden("pipeline_tiny.safetensors")
    tt, mt = golden("transformer_tiny.safetensors")
    tv, mv = golden("vae_tiny.safetensors")
    w = lambda t: {k[2:]: v.float() for k, v in t.items() if k.startswith("w.")}
    torch.manual_seed(int(meta["global_seed"]))
    kw = dict(prompt_embeds=tp["prompt_embeds"], negative_prompt_embeds=tp["negative_prompt_embeds"], video=tp["video"],
              mask_video=tp["mask_video"], reference=tp["reference"], height=32, width=48, latents=tp["latents0"],
              num_inference_steps=2, guidance_scale=6.0, num_frames=9)
    frames = opl.pipeline_call(w(tt), ast.literal_eval(mt["config"]), w(tv), ast.literal_eval(mv["config"]), prec="bf16_ref", **kw)
    check_bf16_model("tiny pipeline frames", frames, tb["frames_bf16_eager"], tp["frames"], lo=0.8, hi=1.25)


@pytest.fixture(scope="module")
def pipe_case(golden, tr_case, vae_case):
    t, meta = golden("pipeline_default.safetensors")
    _, tsd, tcfg, _ = tr_case
    _, vsd, vcfg, _ = vae_case
    assert meta["weights_digest_transformer"] == iw.state_dict_digest(tsd) and meta["weights_digest_vae"] == iw.state_dict_digest(vsd)
    return t, int(meta["global_seed"]), tsd, tcfg, vsd, vcfg, dc.pipeline_inputs()


def _pipe_kw(x):
    return dict(prompt_embeds=x["prompt_embeds"], negative_prompt_embeds=x["negative_prompt_embeds"], video=x["video"],
                mask_video=x["mask_video"], reference=x["reference"], height=32, width=48, latents=x["latents0"],
                num_inference_steps=2, guidance_scale=6.0, num_frames=9)


def test_pipeline_at_product_widths_fp32_matches_reference(pipe_case):
    """`TrajCrafter_Pipeline.__call__` of the reference — conditioning from pixels (two VAE encodes), 2 CFG / DDIM steps of the
    2-layer 5B-geometry transformer, decode through the default-width VAE — against the oracle's pipeline."""
    t, seed, tsd, tcfg, vsd, vcfg, x = pipe_case
    torch.manual_seed(seed)
    lat = opl.pipeline_call(tsd, tcfg, vsd, vcfg, output_type="latent", **_pipe_kw(x))
    _close(lat, t["latents_out"], rtol=1e-3, atol=1e-4)
    torch.manual_seed(seed)
    frames = opl.pipeline_call(tsd, tcfg, vsd, vcfg, **_pipe_kw(x))
    _close(frames, t["frames"], rtol=1e-3, atol=1e-4)
    assert frames.shape == (1, 3, 9, 32, 48) and float(frames.min()) >= 0 and float(frames.max()) <= 1


def test_pipeline_at_product_widths_bf16_ref_models_the_reference_eager_bf16(pipe_case):
    t, seed, tsd, tcfg, vsd, vcfg, x = pipe_case
    torch.manual_seed(seed)
    frames = opl.pipeline_call(tsd, tcfg, vsd, vcfg, prec="bf16_ref", **_pipe_kw(x))
    check_bf16_model("default-width pipeline frames", frames, t["frames_bf16_eager"], t["frames"], lo=0.8, hi=1.25)
