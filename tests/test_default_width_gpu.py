"""HIP path vs the REFERENCE's own runs at the widths the product dispatches on.  GPU only.

Fixtures (tests/golden/make_golden.py default; see tests/test_oracle_default.py for what they are and how the oracle is pinned by
them): the reference's 2-layer `CrossTransformer3DModel` at the 5B geometry and `AutoencoderKLCogVideoX()` at default widths,
fp32 and eager-bf16 outputs; weights / wide inputs regenerated from the host-independent stream and digest-checked.

Per case: (1) `_check_deep` — against the reference's fp32 output the HIP result is as accurate as the oracle's bf16 contract;
(2) it is as accurate as the reference's OWN bf16 execution (mean error <= 1.15 x) and no farther from it than the contract is;
(3) the launch really went where the product's launches go: fused-QKV 48-head self-attention (D = 64), `attn_fwd_kernel<128>`
for the cross-attention, `conv_mfma` tiles for every wide conv, `conv_narrow` for conv_out (recorded from the ops layer and
resolved through `tcx_conv3d_route`, the dispatch function itself).
"""
import ast
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import default_cases as dc                                   # noqa: E402

from oracle import transformer as otr                       # noqa: E402
from oracle import vae as ovae                              # noqa: E402
from tests.test_models_gpu import _check_deep, record_parity  # noqa: E402
from trajectorycrafter_amd import init_weights as iw        # noqa: E402

BF = torch.bfloat16


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.backends.cuda.matmul.allow_tf32 = False
    return torch.device("cuda:0")


def _mean(a, b):
    return float((a.float().cpu() - b.float().cpu()).abs().mean())


def _vs_reference_bf16(what, got, contract, eager, exact):
    """HIP against the reference's own eager bf16 output: as accurate against fp32 (<= 1.15 x) and as close to the reference's bf16
    run as the oracle's contract is (<= 1.25 x; both differ from it by their own and its rounding noise)."""
    e_hip, e_eager, d_hip, d_con = _mean(got, exact), _mean(eager, exact), _mean(got, eager), _mean(contract, eager)
    rec = {"test": what + " — vs the reference's eager bf16 run", "hip_vs_fp32_mean": e_hip, "ref_eager_bf16_vs_fp32_mean": e_eager,
           "hip_vs_ref_eager_bf16_mean": d_hip, "contract_vs_ref_eager_bf16_mean": d_con}
    print(rec)
    record_parity(rec)
    assert e_hip <= 1.15 * e_eager + 1e-5, rec
    assert d_hip <= 1.25 * d_con + 1e-5, rec


class _Spy:
    """Records the shapes the model hands to ops.attn_fwd / ops.conv3d_cl / ops.gemm_bf16 (the product's launch sites)."""

    def __init__(self, monkeypatch):
        from trajectorycrafter_amd import ops
        self.attn, self.conv, self.gemm = [], [], []
        a0, c0, g0 = ops.attn_fwd, ops.conv3d_cl, ops.gemm_bf16

        def attn(q, k, v, *a, **kw):
            self.attn.append((tuple(q.shape), tuple(k.shape), tuple(q.stride()), kw.get("bound_proven", False)))
            return a0(q, k, v, *a, **kw)

        def conv(x, w, bias, cache=None, res=None, ups=0, t_map=None, stride=1, pad=None, out_hw=None):
            self.conv.append(dict(Cin=x.shape[-1], Cout=w.shape[0], k=tuple(w.shape[1:4]), ups=ups, stride=stride, T=x.shape[1], H=x.shape[2],
                                  W=x.shape[3], t_map=t_map is not None, res=res is not None))
            return c0(x, w, bias, cache=cache, res=res, ups=ups, t_map=t_map, stride=stride, pad=pad, out_hw=out_hw)

        def gemm(x, w, *a, **kw):
            self.gemm.append((tuple(w.shape), x.shape[:-1].numel()))
            return g0(x, w, *a, **kw)

        monkeypatch.setattr(ops, "attn_fwd", attn)
        monkeypatch.setattr(ops, "conv3d_cl", conv)
        monkeypatch.setattr(ops, "gemm_bf16", gemm)


def test_transformer_5b_geometry_vs_reference_runs(golden, gpu, monkeypatch):
    from trajectorycrafter_amd.models.crosstransformer3d import CrossTransformer3DModel
    t, meta = golden("transformer_default.safetensors")
    cfg = ast.literal_eval(meta["config"])
    sd = dc.transformer_weights()
    assert iw.state_dict_digest(sd) == meta["weights_digest"], "the hashed weight stream differs on this host"
    x = dc.transformer_inputs()
    model = CrossTransformer3DModel(**cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to(gpu, BF).eval()
    spy = _Spy(monkeypatch)
    rot = (t["rope_cos"].to(gpu), t["rope_sin"].to(gpu))
    out = model(x["hidden_states"].to(gpu, BF), x["encoder_hidden_states"].to(gpu, BF), x["timestep"].to(gpu),
                inpaint_latents=x["inpaint_latents"].to(gpu, BF), cross_latents=x["cross_latents"].to(gpu, BF),
                image_rotary_emb=rot, return_dict=False)[0]
    assert out.dtype == BF and out.shape == t["out_sample"].shape
    # the oracle's bf16 contract, evaluated by torch on the device in fp32 (checker arithmetic; pinned on the CPU by test_oracle_default)
    sdg = {k: v.to(gpu) for k, v in sd.items()}
    con = otr.transformer_forward(sdg, cfg, *(x[n].to(gpu) for n in ("hidden_states", "encoder_hidden_states", "timestep",
                                                                       "inpaint_latents", "cross_latents")), rot, prec="bf16")
    what = "reference-run 2-layer 5B-geometry CrossTransformer3DModel.forward (514 tokens)"
    _check_deep(out, con, t["out_sample"], what)
    _vs_reference_bf16(what, out, con, t["out_sample_bf16_eager"], t["out_sample"])
    # dispatch: 2 self-attention launches on the fused-QKV layout (48 heads x 64, row stride 3 * 3072) + 1 cross-attention at D = 128
    self_attn = [a for a in spy.attn if a[0][-1] == 64]
    cross = [a for a in spy.attn if a[0][-1] == 128]
    assert len(self_attn) == 2 and len(cross) == 1, spy.attn
    assert self_attn[0][0] == (2, 514, 48, 64) and self_attn[0][2][1] == 3 * 3072
    assert cross[0][0] == (2, 288, 16, 128) and cross[0][1] == (2, 192, 16, 128)
    widths = {w for w, _ in spy.gemm}
    assert {(9216, 3072), (12288, 3072), (3072, 12288), (3072, 3072), (2048, 3072), (4096, 3072), (3072, 2048), (3072, 4096)} <= widths, widths
    assert any(rows <= 8 for _, rows in spy.gemm)                           # the skinny (M = batch) AdaLN / time-embedding kernel


def test_transformer_5b_geometry_submodules_vs_reference_taps(golden, gpu):
    """The reference's own sub-module outputs (strided taps): block 0 through the mirror's CogVideoXBlock.forward and the
    dh = 128 PerceiverCrossAttention on the reference-run block output."""
    from trajectorycrafter_amd.models.crosstransformer3d import CrossTransformer3DModel
    t, meta = golden("transformer_default.safetensors")
    cfg = ast.literal_eval(meta["config"])
    sd = dc.transformer_weights()
    x = dc.transformer_inputs()
    model = CrossTransformer3DModel(**cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to(gpu, BF).eval()
    rot = (t["rope_cos"].to(gpu), t["rope_sin"].to(gpu))
    emb = model.time_embedding(model.time_proj(x["timestep"].to(gpu)).to(BF))
    assert _mean(emb, t["tap_temb"]) <= 1.5 * _mean(t["tap_temb_bf16_eager"], t["tap_temb"]) + 1e-4
    pe = model.patch_embed(x["encoder_hidden_states"].to(gpu, BF), torch.cat([x["hidden_states"], x["inpaint_latents"]], 2).to(gpu, BF))
    h, e = model.transformer_blocks[0](pe[:, 226:].contiguous(), pe[:, :226].contiguous(), emb, rot)
    for name, got, tap in (("hidden", h[:, ::4, ::16], "tap_block0_hidden"), ("encoder", e[:, ::4, ::16], "tap_block0_encoder")):
        e_hip, e_eager = _mean(got, t[tap]), _mean(t[tap + "_bf16_eager"], t[tap])
        print(f"block 0 {name}: mean|hip - fp32| {e_hip:.4e}  mean|reference eager bf16 - fp32| {e_eager:.4e}")
        assert e_hip <= 1.15 * e_eager + 1e-5
    ref_tok = model.ref_patch_embed(x["cross_latents"].to(gpu, BF))
    ca = model.perceiver_cross_attention[0](ref_tok, h)
    e_hip, e_eager = _mean(ca[:, ::3, ::8], t["tap_cross0"]), _mean(t["tap_cross0_bf16_eager"], t["tap_cross0"])
    print(f"cross-attention 0 (16 x 128): mean|hip - fp32| {e_hip:.4e}  mean|reference eager bf16 - fp32| {e_eager:.4e}")
    assert e_hip <= 1.15 * e_eager + 1e-5                                  # its input h is the HIP block's, so this bounds both


def test_vae_default_width_vs_reference_runs(golden, gpu, monkeypatch):
    from trajectorycrafter_amd import ops
    from trajectorycrafter_amd.models.autoencoder_magvit import AutoencoderKLCogVideoX
    t, meta = golden("vae_default.safetensors")
    cfg = ast.literal_eval(meta["config"])
    sd = dc.vae_weights()
    assert iw.state_dict_digest(sd) == meta["weights_digest"], "the hashed weight stream differs on this host"
    x = dc.vae_inputs()
    vae = AutoencoderKLCogVideoX(**cfg)
    vae.load_state_dict(sd, strict=True)
    vae = vae.to(gpu, BF).eval()
    sdg = {k: v.to(gpu) for k, v in sd.items()}
    route = ops.conv3d_route
    spy = _Spy(monkeypatch)
    dec = vae.decode(x["z"].to(gpu, BF)).sample
    routes = [route(Cin=c["Cin"], Cout=c["Cout"], k=c["k"], ups=c["ups"], stride=c["stride"], T=c["T"], H=c["H"], W=c["W"],
                    t_map=c["t_map"], res=c["res"]) for c in spy.conv]
    wide = [r for r, c in zip(routes, spy.conv) if c["Cin"] % 64 == 0 and c["Cout"] >= 128]
    assert wide and all(r in (1, 2) for r in wide) and {1, 2} <= set(wide), routes     # conv_mfma<2,4> and <4,2>
    assert 3 in routes                                                                  # conv_narrow (conv_out 128 -> 3)
    assert all(r == 4 for r, c in zip(routes, spy.conv) if c["Cin"] == 16)             # conv_in + SpatialNorm's 1x1x1 tables: igemm
    con = ovae.vae_decode(sdg, cfg, x["z"].to(gpu), prec="bf16")
    what = "reference-run default-width VAE decode (17 frames 32x48)"
    _check_deep(dec, con, t["decoded"], what)
    _vs_reference_bf16(what, dec, con, t["decoded_bf16_eager"], t["decoded"])
    d1 = vae.decode(x["z"][:, :, :1].to(gpu, BF)).sample
    con1 = ovae.vae_decode(sdg, cfg, x["z"][:, :, :1].to(gpu), prec="bf16")
    _check_deep(d1, con1, t["decoded_single_frame"], "reference-run default-width VAE decode (T = 1)")
    # encode: 17 frames -> 5 + 4 + 4 + 4 frame chunks through the stride-2 gathers and the temporal average pool
    spy.conv.clear()
    post = vae.encode(x["video"].to(gpu, BF)).latent_dist
    routes = [route(Cin=c["Cin"], Cout=c["Cout"], k=c["k"], ups=c["ups"], stride=c["stride"], T=c["T"], H=c["H"], W=c["W"],
                    t_map=c["t_map"], res=c["res"]) for c in spy.conv]
    assert {1, 2} <= set(routes), routes
    pc = ovae.vae_encode(sdg, cfg, x["video"].to(gpu), prec="bf16")
    what = "reference-run default-width VAE encode mean (17 frames 32x48)"
    _check_deep(post.mean, pc.mean, t["enc_mean"], what)
    _vs_reference_bf16(what, post.mean, pc.mean, t["enc_mean_bf16_eager"], t["enc_mean"])
    _check_deep(post.logvar, pc.logvar, t["enc_logvar"], "reference-run default-width VAE encode logvar")


def test_pipeline_at_product_widths_vs_reference_runs(golden, gpu):
    """The whole path at the widths the product dispatches on: `TrajCrafter_Pipeline.__call__` of the reference (conditioning from
    pixels, 2 CFG / DDIM steps of the 2-layer 5B-geometry transformer, decode through the default-width VAE; fixture
    pipeline_default.safetensors: fp32 and eager-bf16 runs) against the HIP pipeline.  Like tests/test_pipeline_gpu.py the
    conditioning latents come from the oracle's VAE encoder (the reference samples the reference-frame posterior from the
    GLOBAL rng), handed over as `inpaint_latents=` / `ref_latents=`; the HIP encoder itself is checked in the VAE test above."""
    from oracle import pipeline as opl
    from trajectorycrafter_amd.models.autoencoder_magvit import AutoencoderKLCogVideoX
    from trajectorycrafter_amd.models.crosstransformer3d import CrossTransformer3DModel
    from trajectorycrafter_amd.models.pipeline_trajectorycrafter import TrajCrafter_Pipeline
    t, meta = golden("pipeline_default.safetensors")
    tcfg, vcfg = ast.literal_eval(meta["tr_config"]), ast.literal_eval(meta["vae_config"])
    tsd, vsd = dc.transformer_weights(), dc.vae_weights()
    assert iw.state_dict_digest(tsd) == meta["weights_digest_transformer"] and iw.state_dict_digest(vsd) == meta["weights_digest_vae"]
    x = dc.pipeline_inputs()
    tr = CrossTransformer3DModel(**tcfg)
    tr.load_state_dict(tsd, strict=True)
    vae = AutoencoderKLCogVideoX(**vcfg)
    vae.load_state_dict(vsd, strict=True)
    pipe = TrajCrafter_Pipeline(None, None, vae.to(gpu, BF).eval(), tr.to(gpu, BF).eval())
    torch.manual_seed(int(meta["global_seed"]))
    inpaint, ref = opl.build_conditioning(vsd, vcfg, x["video"], x["mask_video"], x["reference"], 32, 48, "fp32")
    kw = dict(prompt=None, height=32, width=48, num_frames=9, num_inference_steps=2, guidance_scale=6.0,
              prompt_embeds=x["prompt_embeds"].to(BF), negative_prompt_embeds=x["negative_prompt_embeds"].to(BF),
              latents=x["latents0"].to(BF), inpaint_latents=inpaint.to(BF), ref_latents=ref.to(BF))
    lat = pipe(output_type="latent", **kw).videos
    # the oracle's bf16 contract (244 tokens, 2 layers: seconds on the CPU; pinned against this fixture by tests/test_oracle_default.py)
    g = lambda a: a.to(BF).float()
    con_lat = opl.denoise(tsd, tcfg, g(x["latents0"]), g(x["prompt_embeds"]), g(x["negative_prompt_embeds"]), g(inpaint), g(ref),
                          32, 48, 2, 6.0, prec="bf16")
    what = "reference-run pipeline at product widths (2 CFG / DDIM steps, 2-layer 5B geometry)"
    _check_deep(lat, con_lat, t["latents_out"], what + ": latents")
    frames = pipe(**kw).videos
    assert frames.shape == (1, 3, 9, 32, 48) and float(frames.min()) >= 0 and float(frames.max()) <= 1
    con_frames = opl.decode_latents(vsd, vcfg, con_lat, prec="bf16")
    _check_deep(frames, con_frames, t["frames"], what + ": frames (+ default-width VAE decode)")
    _vs_reference_bf16(what + ": frames", frames, con_frames, t["frames_bf16_eager"], t["frames"])
    assert torch.equal(pipe(**kw).videos, frames)
