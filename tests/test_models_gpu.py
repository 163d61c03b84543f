"""Model-level parity on the GPU: HIP-backed mirror classes vs the CPU oracle under the bf16 rounding
contract, on the committed golden fixtures (weights + inputs generated from the reference).

Tolerance: the model runs 2 transformer blocks (or ~20 conv layers) in bf16; outputs are compared
with the oracle's bf16-contract output.  Both sides round at the same tensors, so the residual
difference is accumulation order inside GEMM / attention / conv: we allow 2 bf16 ulps + atol and
require the mean error to be far below one ulp.  Against the fp32 golden (the reference's own fp32
output) we check the looser bf16-vs-fp32 bound to show the contract tracks the reference.
"""
import ast

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import transformer as otr
from oracle import vae as ovae

BF = torch.bfloat16


def _weights(t):
    return {k[2:]: v for k, v in t.items() if k.startswith("w.")}


def _check(got, ref, ulps=2.0, atol=4e-3, mean_tol=2e-3):
    """single-op-depth closeness: within `ulps` bf16 ulps + atol for 99.9 % of the elements."""
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape
    assert torch.isfinite(got).all()
    err = (got - ref).abs()
    bound = ref.abs() * 2.0 ** -7 * ulps + atol
    frac_bad = float((err > bound).float().mean())
    assert frac_bad < 1e-3, f"{frac_bad:.2%} elements outside {ulps} ulp + {atol}; max err {float(err.max()):.4g}"
    assert float(err.mean()) < mean_tol * (float(ref.abs().mean()) + 1e-6) + 1e-4, float(err.mean())


def record_parity(rec: dict) -> None:
    """Append a measured-tolerance record to gpurun_out/parity.jsonl (merged back from the GPU box; tools/collect_parity.py turns
    it into profiles/<round>_parity.json, which DESIGN §4 quotes and which justifies the factors of `_check_deep`)."""
    import json
    import os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass


def _check_deep(got, contract, exact, what, record=True, mean_x=1.15, p999_x=1.3, hc_x=1.5):
    """Deep bf16 pipelines (many rounded layers) diverge element-wise even between two correct
    implementations, because a one-ulp flip early on is amplified downstream.  The criterion that
    still catches real bugs: measured against the reference's own fp32 output (`exact`, from the golden
    fixture), the HIP result must be as accurate as the oracle's bf16 rounding contract —
      mean|hip - exact| <= mean_x * mean|contract - exact|,  p99.9|hip - exact| <= p999_x * p99.9|contract - exact|,
    and HIP must sit as close to the contract as the contract sits to the exact result (hc_x).
    Factors (round 4): every record of profiles/r3_parity.json measures mean <= 1.033 x, p99.9 <= 1.030 x,
    hip-vs-contract <= 1.19 x, so the defaults are 1.15 / 1.3 / 1.5 (they were 1.5 / 2 / 2); a caller that needs more passes its
    factors explicitly and says which record justifies them.  tools/exp/parity_sensitivity.sh shows what these bounds catch."""
    got, contract, exact = got.float().cpu(), contract.float().cpu(), exact.float().cpu()
    assert got.shape == exact.shape and torch.isfinite(got).all()
    e_hip, e_con, e_hc = (got - exact).abs(), (contract - exact).abs(), (got - contract).abs()
    q = lambda t: float(torch.quantile(t.flatten()[:4_000_000], 0.999))
    msg = (f"{what}: mean|hip-exact| {float(e_hip.mean()):.3e}  mean|contract-exact| {float(e_con.mean()):.3e}  "
           f"mean|hip-contract| {float(e_hc.mean()):.3e}  p99.9 {q(e_hip):.3e} vs {q(e_con):.3e}  scale {float(exact.abs().mean()):.3e}")
    print(msg)
    inside = lambda a, b: float(((a - b).abs() <= 1e-3 * b.abs() + 1e-4).float().mean())
    rec = {"test": what, "scale": float(exact.abs().mean()),
           "hip_vs_fp32": {"max": float(e_hip.max()), "mean": float(e_hip.mean()), "p99.9": q(e_hip), "inside_rtol1e-3_atol1e-4": inside(got, exact)},
           "contract_vs_fp32": {"max": float(e_con.max()), "mean": float(e_con.mean()), "p99.9": q(e_con), "inside_rtol1e-3_atol1e-4": inside(contract, exact)},
           "hip_vs_contract": {"max": float(e_hc.max()), "mean": float(e_hc.mean()), "inside_rtol1e-3_atol1e-4": inside(got, contract)}}
    if record:
        record_parity(rec)
    if exact.numel() < 65536:                  # the 99.9th percentile of a small tensor is the tail of < 65 elements: r4 records reach 1.26 x
        p999_x = max(p999_x, 1.5 if exact.numel() >= 4096 else 2.0)      # below 4 k elements "p99.9" is the largest few errors (192-element case: 1.73 x)
    # sampling noise of the two mean ratios: |error| is roughly exponential (coefficient of variation ~ 1), so the ratio of two means over
    # n elements has a relative standard deviation of about sqrt(2 / n); four of those are granted on top of the factor.  Nothing for a
    # full-size tensor (n = 1e8: 6e-4), 0.03 at the default-width cases (n = 37 k), 0.18 for a 960-element latent — the round-4 deep
    # fuzz sweeps (5 500 cases) tripped the bare 1.15 twice, at 1.156 and 1.162, both on 960-element outputs
    slack = 4.0 * (2.0 / max(exact.numel(), 1)) ** 0.5
    mean_x, hc_x = mean_x + slack, hc_x + slack
    assert float(e_hip.mean()) <= mean_x * float(e_con.mean()) + 1e-5, msg
    assert q(e_hip) <= p999_x * q(e_con) + 1e-4, msg
    assert float(e_hc.mean()) <= hc_x * float(e_con.mean()) + 1e-5, msg
    return rec


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def test_transformer_forward_tiny(golden, gpu):
    from trajectorycrafter_amd.models.crosstransformer3d import CrossTransformer3DModel
    t, meta = golden("transformer_tiny.safetensors")
    cfg = ast.literal_eval(meta["config"])
    sd = _weights(t)
    model = CrossTransformer3DModel(**cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to(gpu, BF).eval()
    rot = (t["rope_cos"].to(gpu), t["rope_sin"].to(gpu))
    out = model(t["hidden_states"].to(gpu, BF), t["encoder_hidden_states"].to(gpu, BF), t["timestep"].to(gpu),
                inpaint_latents=t["inpaint_latents"].to(gpu, BF), cross_latents=t["cross_latents"].to(gpu, BF),
                image_rotary_emb=rot, return_dict=False)[0]
    assert out.dtype == BF and out.shape == t["out_sample"].shape
    ref = otr.transformer_forward({k: v.float() for k, v in sd.items()}, cfg, t["hidden_states"], t["encoder_hidden_states"],
                                  t["timestep"], t["inpaint_latents"], t["cross_latents"], (t["rope_cos"], t["rope_sin"]),
                                  prec="bf16")
    _check_deep(out, ref, t["out_sample"], "transformer tiny (2 blocks + cross-attention)")
    # opt-in reuse of the reference-token K / V (model.cache_cross_kv): bit-identical, reused while the SAME cross_latents tensor
    # comes back (the pipeline's `ref_input`), recomputed when it is modified in place or replaced
    cl = t["cross_latents"].to(gpu, BF)
    call = lambda c: model(t["hidden_states"].to(gpu, BF), t["encoder_hidden_states"].to(gpu, BF), t["timestep"].to(gpu),
                           inpaint_latents=t["inpaint_latents"].to(gpu, BF), cross_latents=c, image_rotary_emb=rot, return_dict=False)[0]
    model.cache_cross_kv = True
    try:
        a = call(cl)
        entry0 = model._cross_kv_cache
        b = call(cl)
        assert torch.equal(a, out) and torch.equal(b, out) and model._cross_kv_cache is entry0 and entry0[0] is cl
        cl.mul_(-1.0)                                           # in-place change -> version counter -> recomputed
        c = call(cl)
        assert model._cross_kv_cache is not entry0 and not torch.equal(c, out)
        # two clips of the same shape and different values back to back.  The second clip's tensor is a NEW
        # object; the caching allocator may well give it the first one's address (forced here by freeing it first) — the cache
        # must not serve the first clip's K / V for it
        model.clear_cross_kv_cache()
        clip1 = t["cross_latents"].to(gpu, BF)
        o1 = call(clip1)
        addr = clip1.data_ptr()
        held = model._cross_kv_cache[0]
        assert held is clip1
        del clip1, held
        model.clear_cross_kv_cache()                            # what the pipeline does at the start of every clip
        clip2 = (t["cross_latents"] * -0.5 + 0.25).to(gpu, BF)
        o2 = call(clip2)
        model.cache_cross_kv = False
        assert torch.equal(o1, out) and torch.equal(call(clip2), o2) and not torch.equal(o2, o1)
        # and WITHOUT the explicit clear: the entry holds clip A alive, so clip B cannot alias it; identity decides
        model.cache_cross_kv = True
        clip_a = t["cross_latents"].to(gpu, BF)
        oa = call(clip_a)
        clip_b = (t["cross_latents"] * -0.5 + 0.25).to(gpu, BF)
        assert clip_b.data_ptr() != clip_a.data_ptr()
        ob = call(clip_b)
        assert torch.equal(oa, out) and torch.equal(ob, o2)
        del addr
        model.cache_cross_kv = False
        assert torch.equal(call(cl), c)
    finally:
        model.cache_cross_kv = False
        model._cross_kv_cache = None


def test_transformer_sincos_branch_tiny(golden, gpu):
    """use_rotary_positional_embeddings=False (the 2B family, reference :752-784): the HIP forward against the reference's own run
    of that branch (fixture) and the oracle's bf16 contract — at the configured sample size and at a smaller latent with fewer
    frames (trilinear resize of the table + row cut)."""
    from trajectorycrafter_amd.models.crosstransformer3d import CrossTransformer3DModel
    t, meta = golden("transformer_sincos_tiny.safetensors")
    tw, _ = golden(meta["weights"])
    cfg = ast.literal_eval(meta["config"])
    sd = _weights(tw)
    model = CrossTransformer3DModel(**cfg)
    model.load_state_dict(sd, strict=True)
    assert torch.equal(model.pos_embedding, t["pos_embedding"])
    model = model.to(gpu, BF).eval()
    sdf = {k: v.float() for k, v in sd.items()}
    for tag in "ab":
        args = [t[f"{n}_{tag}"] for n in ("hidden_states", "encoder_hidden_states", "timestep", "inpaint_latents", "cross_latents")]
        out = model(args[0].to(gpu, BF), args[1].to(gpu, BF), args[2].to(gpu), inpaint_latents=args[3].to(gpu, BF),
                    cross_latents=args[4].to(gpu, BF), image_rotary_emb=None, return_dict=False)[0]
        ref = otr.transformer_forward(sdf, cfg, *args, None, prec="bf16")
        _check_deep(out, ref, t[f"out_sample_{tag}"], f"non-rotary transformer tiny ({tag})")


def test_transformer_block_and_cross_attention_signatures(golden, gpu):
    """The reference's per-module call signatures work on the mirror classes (CogVideoXBlock :224-230,
    PerceiverCrossAttention :376)."""
    from trajectorycrafter_amd.models.crosstransformer3d import CrossTransformer3DModel
    from oracle import diffusers_restated as dr
    from oracle.prec import Prec
    t, meta = golden("transformer_tiny.safetensors")
    cfg = ast.literal_eval(meta["config"])
    sd = _weights(t)
    model = CrossTransformer3DModel(**cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to(gpu, BF).eval()
    p = Prec("bf16")
    sdf = {k: v.float() for k, v in sd.items()}
    D = cfg["num_attention_heads"] * 64
    emb = dr.timestep_embedding(p, sdf, "time_embedding.", p.R(dr.timesteps_proj(t["timestep"], D)))
    pe = t["tap_patch_embed"].to(BF)
    h, e = model.transformer_blocks[0](pe[:, 10:].to(gpu), pe[:, :10].to(gpu), emb.to(gpu, BF),
                                       (t["rope_cos"].to(gpu), t["rope_sin"].to(gpu)))
    rh, re = otr.cogvideox_block(p, sdf, "transformer_blocks.0.", pe[:, 10:].float(), pe[:, :10].float(), emb,
                                 (t["rope_cos"], t["rope_sin"]), cfg["num_attention_heads"], 1e-5)
    _check(h, rh)
    _check(e, re)
    ref_tok, lat = t["tap_ref_tokens"].to(BF), t["tap_block0_hidden"].to(BF)
    ca = model.perceiver_cross_attention[0](ref_tok.to(gpu), lat.to(gpu))
    rca = otr.perceiver_cross_attention(p, sdf, "perceiver_cross_attention.0.", ref_tok.float(), lat.float(), 2, 64)
    # called without add_to_latents the projection is stored (one rounding); fused, it is not.  The kernel's bound-centred
    # softmax and the oracle's max-centred one round every probability independently (kernel tests: 3 * 2^-9 bound), on
    # top of five rounded stages: mean tolerance 3e-3 of the mean magnitude instead of the single-op 2e-3
    _check(ca, p.R(rca), mean_tol=3e-3)


def test_transformer_rejects_cpu_and_missing_conditioning(golden, gpu):
    from trajectorycrafter_amd.models.crosstransformer3d import CrossTransformer3DModel
    from trajectorycrafter_amd._lib import TcxError
    t, meta = golden("transformer_tiny.safetensors")
    cfg = ast.literal_eval(meta["config"])
    model = CrossTransformer3DModel(**cfg).to(gpu, BF)
    hs, enc = t["hidden_states"].to(gpu, BF), t["encoder_hidden_states"].to(gpu, BF)
    with pytest.raises(ValueError):
        model(hs, enc, t["timestep"].to(gpu), inpaint_latents=None, cross_latents=t["cross_latents"].to(gpu, BF))
    with pytest.raises(ValueError):
        model(hs, enc, t["timestep"].to(gpu), inpaint_latents=t["inpaint_latents"].to(gpu, BF), cross_latents=None)
    with pytest.raises(TcxError):
        model(hs.cpu(), enc.cpu(), t["timestep"], inpaint_latents=t["inpaint_latents"], cross_latents=t["cross_latents"])


def test_vae_decode_tiny(golden, gpu):
    from trajectorycrafter_amd.models.autoencoder_magvit import AutoencoderKLCogVideoX
    t, meta = golden("vae_tiny.safetensors")
    cfg = ast.literal_eval(meta["config"])
    sd = _weights(t)
    vae = AutoencoderKLCogVideoX(**cfg)
    vae.load_state_dict(sd, strict=True)
    vae = vae.to(gpu, BF).eval()
    sdf = {k: v.float() for k, v in sd.items()}
    z = t["z"].to(BF)
    dec = vae.decode(z.to(gpu)).sample
    ref = ovae.vae_decode(sdf, cfg, z.float(), prec="bf16")
    assert dec.shape == ref.shape == (1, 3, 17, 32, 48)
    _check_deep(dec, ref, t["decoded"], "vae decode tiny (13 resnets, 2 chunks)")
    # single latent frame (T == 1 path, reference :1227-1233)
    d1 = vae.decode(z[:, :, :1].to(gpu)).sample
    _check_deep(d1, ovae.vae_decode(sdf, cfg, z[:, :, :1].float(), prec="bf16"), t["decoded_single_frame"], "vae decode T=1")
    # decode is re-entrant after the cache is cleared: same result twice
    assert torch.equal(vae.decode(z.to(gpu)).sample, dec)
    # fused frames epilogue == (x/2+.5).clamp(0,1).float()
    fr = vae.decode_to_frames(z.to(gpu))
    assert torch.equal(fr, (dec / 2 + 0.5).clamp(0, 1).float())


def test_vae_tiled_decode_tiny(golden, gpu):
    """enable_tiling() + decode against the reference's own tiled decode (fixture) and the oracle under the bf16 contract:
    3 x 3 ragged tiles x 2 temporal chunks, seams blended in place by tcx_blend_ramp_bf16."""
    from trajectorycrafter_amd.models.autoencoder_magvit import AutoencoderKLCogVideoX
    t, meta = golden("vae_tiled_tiny.safetensors")
    tv, _ = golden(meta["weights"])
    cfg = ast.literal_eval(meta["config"])
    sd = _weights(tv)
    vae = AutoencoderKLCogVideoX(**cfg)
    vae.load_state_dict(sd, strict=True)
    vae = vae.to(gpu, BF).eval()
    sdf = {k: v.float() for k, v in sd.items()}
    z = t["z"].to(BF)
    plain = vae.decode(z.to(gpu)).sample
    vae.enable_tiling()
    assert (vae.tile_latent_min_height, vae.tile_latent_min_width) == (6, 5)
    dec = vae.decode(z.to(gpu)).sample
    assert dec.shape == (1, 3, 17, 96, 80) and not torch.equal(dec, plain)
    _check_deep(dec, ovae.vae_tiled_decode(sdf, cfg, z.float(), prec="bf16"), t["decoded_tiled"], "vae tiled decode tiny (9 tiles, 2 chunks)")
    assert torch.equal(vae.decode(z.to(gpu)).sample, dec)                          # re-entrant
    assert torch.equal(vae.decode_to_frames(z.to(gpu)), (dec / 2 + 0.5).clamp(0, 1).float())
    assert torch.equal(vae.decode_cl_bf16(z.to(gpu)).permute(0, 4, 1, 2, 3), dec)
    # caller-set geometry, one temporal chunk
    kw = dict(tile_sample_min_height=64, tile_sample_min_width=64, tile_overlap_factor_height=0.25, tile_overlap_factor_width=0.25)
    vae.enable_tiling(**kw)
    d2 = vae.decode(z[:, :, :3].to(gpu)).sample
    _check_deep(d2, ovae.vae_tiled_decode(sdf, cfg, z[:, :, :3].float(), prec="bf16", **kw), t["decoded_tiled_64"], "vae tiled decode, 64 px tiles")
    # a latent inside one tile takes the plain path (:1222-1225); a single frame fails like the reference's empty concat
    small = z[:, :, :, :8, :8].contiguous().to(gpu)
    tiled_small = vae.decode(small).sample
    vae.disable_tiling()
    assert torch.equal(tiled_small, vae.decode(small).sample)
    assert torch.equal(vae.decode(z.to(gpu)).sample, plain)
    vae.enable_tiling()
    with pytest.raises(ValueError, match="single frames"):
        vae.decode(z[:, :, :1].to(gpu))
    # slicing (:1274-1278) is the batched decode, bit for bit
    vae.disable_tiling()
    z2 = torch.cat([z, z.flip(4)], 0).to(gpu)
    batched = vae.decode(z2).sample
    vae.enable_slicing()
    assert torch.equal(vae.decode(z2).sample, batched)
    vae.disable_slicing()


def test_vae_submodule_and_blend_api(golden, gpu):
    """The reference's names for the pieces: `vae.decoder(z)` / `vae.encoder(x)` on NCTHW tensors (one chunk), `_decode`,
    `tiled_decode` regardless of `use_tiling`, `blend_v` / `blend_h` on [N,C,T,H,W] tensors (in place on b)."""
    from trajectorycrafter_amd.models.autoencoder_magvit import AutoencoderKLCogVideoX
    t, meta = golden("vae_tiled_tiny.safetensors")
    tv, _ = golden(meta["weights"])
    cfg = ast.literal_eval(meta["config"])
    vae = AutoencoderKLCogVideoX(**cfg)
    vae.load_state_dict(_weights(tv), strict=True)
    vae = vae.to(gpu, BF).eval()
    z = t["z"].to(gpu, BF)
    full = vae.decode(z).sample
    assert torch.equal(vae._decode(z).sample, full) and torch.equal(vae._decode(z, return_dict=False)[0], full)
    vae._clear_fake_context_parallel_cache()
    first = vae.decoder(z[:, :, :3].contiguous())                          # the first chunk of decode (frame 0 replicated)
    second = vae.decoder(z[:, :, 3:].contiguous())                         # the second one through the conv caches the first left
    vae._clear_fake_context_parallel_cache()
    assert torch.equal(torch.cat([first, second], dim=2), full)
    x = (full[:, :, :9].float().clamp(-1, 1)).to(BF)
    mom = vae.encoder(x[:, :, :5].contiguous())
    mom2 = vae.encoder(x[:, :, 5:9].contiguous())
    vae._clear_fake_context_parallel_cache()
    post = vae.encode(x).latent_dist
    assert torch.equal(torch.cat([mom, mom2], dim=2)[:, :16], post.mean)
    assert not vae.use_tiling
    tiled = vae.tiled_decode(z).sample
    vae.enable_tiling()
    assert torch.equal(tiled, vae.decode(z).sample)
    vae.disable_tiling()
    a, b = full[:, :, :, :20].clone(), full[:, :, :, 30:60].clone()
    want = b.clone()
    for y in range(8):
        want[:, :, :, y, :] = a[:, :, :, -8 + y, :] * (1 - y / 8) + want[:, :, :, y, :] * (y / 8)
    assert vae.blend_v(a, b, 8) is b and torch.equal(b, want)
    a, b = full[..., :12].clone(), full[..., 20:50].clone()
    want = b.clone()
    for xx in range(5):
        want[..., xx] = a[..., -5 + xx] * (1 - xx / 5) + want[..., xx] * (xx / 5)
    assert torch.equal(vae.blend_h(a, b, 5), want)


def test_blend_ramp_bit_exact(gpu):
    """tcx_blend_ramp_bf16 against the reference's eager bf16 loop (blend_v / blend_h, autoencoder_magvit.py:1282-1301) run by
    torch on the device: same three roundings per element -> identical bits; ragged extents (min with both tile sizes)."""
    from trajectorycrafter_amd import ops
    from trajectorycrafter_amd._lib import TcxError
    g = torch.Generator(device=gpu).manual_seed(3)
    for (Ha, Wa), (Hb, Wb), ext, dim in (((16, 12), (16, 12), 5, 2), ((16, 12), (7, 12), 9, 2), ((16, 12), (16, 5), 8, 3),
                                          ((3, 12), (16, 12), 8, 2), ((16, 12), (16, 9), 1, 3), ((40, 72), (40, 72), 40, 3)):
        a = torch.randn(2, 3, Ha, Wa, 3, device=gpu, generator=g).to(BF)
        b = torch.randn(2, 3, Hb, Wb, 3, device=gpu, generator=g).to(BF)
        want = b.clone()
        e = min(a.shape[dim], b.shape[dim], ext)
        for y in range(e):                                   # the reference's loop, on [N,T,H,W,C] instead of [N,C,T,H,W]
            if dim == 2:
                want[:, :, y] = a[:, :, -e + y] * (1 - y / e) + want[:, :, y] * (y / e)
            else:
                want[:, :, :, y] = a[:, :, :, -e + y] * (1 - y / e) + want[:, :, :, y] * (y / e)
        a0 = a.clone()
        got = ops.blend_ramp(a, b, ext, dim)
        assert got is b and torch.equal(b, want) and torch.equal(a, a0), (Ha, Wa, Hb, Wb, ext, dim)
    with pytest.raises(TcxError):
        ops.blend_ramp(a, b[:, :, :, :, :2].contiguous(), 4, 2)
    assert ops.blend_ramp(a, b, 0, 2) is b


def test_vae_encode_tiny(golden, gpu):
    """HIP VAE encoder (stride-2 conv gather, temporal avg-pool, GroupNorm+SiLU, 4-frame chunks with conv cache)."""
    from trajectorycrafter_amd.models.autoencoder_magvit import AutoencoderKLCogVideoX
    t, meta = golden("vae_tiny.safetensors")
    cfg = ast.literal_eval(meta["config"])
    sd = _weights(t)
    vae = AutoencoderKLCogVideoX(**cfg)
    vae.load_state_dict(sd, strict=True)
    vae = vae.to(gpu, BF).eval()
    sdf = {k: v.float() for k, v in sd.items()}
    video = t["video"].to(BF)
    post = vae.encode(video.to(gpu)).latent_dist
    ref = ovae.vae_encode(sdf, cfg, video.float(), prec="bf16")
    assert post.mean.shape == t["enc_mean"].shape == (1, 16, 3, 4, 6)
    _check_deep(post.mean, ref.mean, t["enc_mean"], "vae encode mean (9 frames, 2 chunks)")
    _check_deep(post.logvar, ref.logvar, t["enc_logvar"], "vae encode logvar")
    assert torch.equal(vae.encode(video.to(gpu))[0].mode(), post.mean)       # re-entrant, tuple access like the reference
    g = torch.Generator(device="cuda").manual_seed(1)
    smp = post.sample(g)
    assert smp.shape == post.mean.shape and torch.isfinite(smp.float()).all()
    # forward = encode -> mode / sample -> decode (:1394-1410)
    rt = vae(video.to(gpu))
    assert torch.equal(rt.sample, vae.decode(post.mode()).sample) and rt.sample.shape == video.shape
    rt2 = vae(video.to(gpu), sample_posterior=True, return_dict=False, generator=torch.Generator(device="cuda").manual_seed(1))
    assert isinstance(rt2, tuple) and torch.equal(rt2[0].sample, vae.decode(smp).sample)
    # single frame path (:1193-1197)
    p1 = vae.encode(video[:, :, :1].to(gpu)).latent_dist
    r1 = ovae.vae_encode(sdf, cfg, video[:, :, :1].float(), prec="bf16")
    e1 = ovae.vae_encode(sdf, cfg, video[:, :, :1].float(), prec="fp32")
    _check_deep(p1.mean, r1.mean, e1.mean, "vae encode single frame")


def test_vae_submodule_reference_signature_forwards(golden, gpu):
    """The reference's per-module call signatures on the VAE mirrors (VERDICT r3 missing 3; reference autoencoder_magvit.py
    :61-73, :135-163, :183-212, :320-355, :436-464, :524-548, :631-660): NCTHW in / out, same HIP kernels as the composed model.
    Each against the oracle's building block under the bf16 contract (single-op depth: `_check`), and composed back into the
    decoder / encoder bit for bit."""
    import torch.nn.functional as F
    from oracle.prec import Prec
    from oracle import diffusers_restated as dr
    from trajectorycrafter_amd.models.autoencoder_magvit import AutoencoderKLCogVideoX
    t, meta = golden("vae_tiny.safetensors")
    cfg = ast.literal_eval(meta["config"])
    sd = _weights(t)
    vae = AutoencoderKLCogVideoX(**cfg)
    vae.load_state_dict(sd, strict=True)
    vae = vae.to(gpu, BF).eval()
    sdf = {k: v.float() for k, v in sd.items()}
    p = Prec("bf16")
    g = torch.Generator().manual_seed(9)
    zq = torch.randn(1, 16, 3, 4, 6, generator=g).to(BF)
    dec, groups = vae.decoder, cfg["norm_num_groups"]
    # CogVideoXCausalConv3d.forward / fake_context_parallel_forward (first chunk: frame 0 repeated; second: the cache)
    vae._clear_fake_context_parallel_cache()
    ctx0 = dec.conv_in.fake_context_parallel_forward(zq.to(gpu))
    assert ctx0.shape == (1, 16, 5, 4, 6) and torch.equal(ctx0[:, :, 0], zq[:, :, 0].to(gpu)) and torch.equal(ctx0[:, :, 2:], zq.to(gpu))
    cache = {}
    h = dec.conv_in(zq.to(gpu))
    _check(h, ovae.causal_conv3d(p, sdf, "decoder.conv_in.", zq.float(), cache))
    z2 = torch.randn(1, 16, 2, 4, 6, generator=g).to(BF)
    ctx1 = dec.conv_in.fake_context_parallel_forward(z2.to(gpu))
    assert torch.equal(ctx1[:, :, :2], zq[:, :, -2:].to(gpu)) and torch.equal(ctx1[:, :, 2:], z2.to(gpu))
    _check(dec.conv_in(z2.to(gpu)), ovae.causal_conv3d(p, sdf, "decoder.conv_in.", z2.float(), cache))
    vae._clear_fake_context_parallel_cache()
    # CogVideoXSpatialNorm3D.forward(f, zq): no activation
    r0 = dec.mid_block.resnets[0]
    f = h
    _check(r0.norm1(f, zq.to(gpu)), ovae.spatial_norm3d(p, sdf, "decoder.mid_block.resnets.0.norm1.", h.float().cpu(), zq.float(), groups, {}, silu=False))
    # CogVideoXResnetBlock3D.forward(inputs, temb, zq), decoder (SpatialNorm) and encoder (GroupNorm) flavours, with a shortcut conv
    _check(r0(f, None, zq.to(gpu)), ovae.resnet_block3d(p, sdf, "decoder.mid_block.resnets.0.", h.float().cpu(), zq.float(), groups, cfg.get("norm_eps", 1e-6), {}))
    vae._clear_fake_context_parallel_cache()
    with pytest.raises(NotImplementedError, match="temb"):
        r0(f, torch.zeros(1, 512, device=gpu, dtype=BF), zq.to(gpu))
    x = torch.randn(1, 8, 5, 8, 12, generator=g).to(BF)                      # encoder down block 1: 8 -> 16 channels (1x1x1 shortcut)
    e1 = vae.encoder.down_blocks[1].resnets[0]
    assert e1.in_channels != e1.out_channels
    _check(e1(x.to(gpu)), ovae.resnet_block3d(p, sdf, "encoder.down_blocks.1.resnets.0.", x.float(), None, groups, cfg.get("norm_eps", 1e-6), {}))
    vae._clear_fake_context_parallel_cache()
    # CogVideoXSafeConv3d.forward: the module's own (valid-in-time) convolution, here the 1x1x1 shortcut and a 3x3x3 weight
    _check(e1.conv_shortcut(x.to(gpu)), p.R(F.conv3d(x.float(), sdf["encoder.down_blocks.1.resnets.0.conv_shortcut.weight"],
                                                     sdf["encoder.down_blocks.1.resnets.0.conv_shortcut.bias"])))
    xp = F.pad(x.float(), (1, 1, 1, 1))                                        # what CogVideoXCausalConv3d hands its SafeConv3d (:158-162)
    want = p.R(F.conv3d(xp, sdf["encoder.down_blocks.0.resnets.0.conv1.conv.weight"], sdf["encoder.down_blocks.0.resnets.0.conv1.conv.bias"]))
    got = vae.encoder.down_blocks[0].resnets[0].conv1.conv(xp.to(gpu, BF))
    assert got.shape == want.shape == (1, 8, 3, 8, 12)
    _check(got, want)
    # block-level forwards compose into the decoder / encoder exactly (same kernels, same order)
    z = t["z"].to(gpu, BF)[:, :, :3].contiguous()
    vae._clear_fake_context_parallel_cache()
    whole = dec(z)
    vae._clear_fake_context_parallel_cache()
    hh = dec.conv_in(z)
    hh = dec.mid_block(hh, None, z)
    for up in dec.up_blocks:
        hh = up(hh, None, z)
    from trajectorycrafter_amd.models.autoencoder_magvit import _from_cl, _groupnorm_silu, _to_cl
    # norm_out + conv_act is ONE fused kernel in the composed decoder: finish through the same kernel -> the same bits
    tail = _from_cl(dec.conv_out.forward_cl(dec.norm_out.forward_cl(_to_cl(hh), _to_cl(z), silu=True)))
    vae._clear_fake_context_parallel_cache()
    assert torch.equal(tail, whole)
    # ... and norm_out called the reference's way (no activation) against the oracle's building block
    _check(dec.norm_out(hh, z), ovae.spatial_norm3d(p, sdf, "decoder.norm_out.", hh.float().cpu(), z.float().cpu(), groups, {}, silu=False))
    v = t["video"].to(gpu, BF)[:, :, :5].contiguous()
    enc = vae.encoder
    whole = enc(v)
    vae._clear_fake_context_parallel_cache()
    hh = enc.conv_in(v)
    for blk in enc.down_blocks:
        hh = blk(hh, None, None)
    hh = enc.mid_block(hh, None, None)
    tail = _from_cl(enc.conv_out.forward_cl(_groupnorm_silu(enc.norm_out, _to_cl(hh))))
    vae._clear_fake_context_parallel_cache()
    assert torch.equal(tail, whole)
    # the diffusers resamplers on NCTHW
    up0 = dec.up_blocks[0].upsamplers[0]
    xu = torch.randn(1, up0.conv.in_channels, 3, 4, 6, generator=g).to(BF)
    _check(up0(xu.to(gpu)), dr.upsample3d(p, sdf, "decoder.up_blocks.0.upsamplers.0.", xu.float(), up0.compress_time))
    dn0 = enc.down_blocks[0].downsamplers[0]
    xd = torch.randn(1, dn0.conv.in_channels, 5, 8, 12, generator=g).to(BF)
    _check(dn0(xd.to(gpu)), dr.downsample3d(p, sdf, "encoder.down_blocks.0.downsamplers.0.", xd.float(), dn0.compress_time))


def test_softmax_path_knob_changes_launches_not_results(golden, gpu, monkeypatch):
    """CrossTransformer3DModel.set_softmax_path (bench.py's bracket of the weight-dependent attention path): "auto" with these
    LayerNorm parameters is the proven bound-centred loop; "unproven" adds the per-workgroup test (same loop -> same bits);
    "exact" sends every workgroup to the running-max kernel (k_sqmax withheld) -> equal within the contract's tolerance."""
    from trajectorycrafter_amd import ops
    from trajectorycrafter_amd.models.crosstransformer3d import CrossTransformer3DModel
    t, meta = golden("transformer_tiny.safetensors")
    cfg = ast.literal_eval(meta["config"])
    model = CrossTransformer3DModel(**cfg)
    model.load_state_dict(_weights(t), strict=True)
    model = model.to(gpu, BF).eval()
    rot = (t["rope_cos"].to(gpu), t["rope_sin"].to(gpu))
    seen = []
    a0 = ops.attn_fwd
    monkeypatch.setattr(ops, "attn_fwd", lambda q, k, v, *a, **kw: (seen.append((q.shape[-1], kw.get("k_sqmax") is not None, kw.get("bound_proven", False))), a0(q, k, v, *a, **kw))[1])
    call = lambda: model(t["hidden_states"].to(gpu, BF), t["encoder_hidden_states"].to(gpu, BF), t["timestep"].to(gpu),
                         inpaint_latents=t["inpaint_latents"].to(gpu, BF), cross_latents=t["cross_latents"].to(gpu, BF),
                         image_rotary_emb=rot, return_dict=False)[0]
    assert model.softmax_path_in_use() == {"self": "proven", "cross": "bound-tested"}
    auto = call()
    assert seen == [(64, True, True), (64, True, False), (64, True, True)]          # block 0, cross-attention 0, block 1
    seen.clear()
    model.set_softmax_path("unproven")
    assert model.softmax_path_in_use()["self"] == "unproven"
    unproven = call()
    assert [s[2] for s in seen] == [False, False, False] and all(s[1] for s in seen)
    assert torch.equal(unproven, auto)
    seen.clear()
    model.set_softmax_path("exact")
    assert model.softmax_path_in_use() == {"self": "exact", "cross": "exact"}
    exact = call()
    assert not any(s[1] for s in seen)
    # another correct bf16 execution of a 2-block model: not element-wise equal (other exponent origin -> P rounds differently), as
    # accurate against the reference's fp32 output, and close to the default path in the mean
    ex = t["out_sample"].float()
    e_auto, e_exact = float((auto.float().cpu() - ex).abs().mean()), float((exact.float().cpu() - ex).abs().mean())
    assert not torch.equal(exact, auto) and e_exact <= 1.1 * e_auto + 1e-5, (e_exact, e_auto)
    assert float((exact.float() - auto.float()).abs().mean()) <= 1.5 * e_auto
    model.set_softmax_path("auto")
    assert torch.equal(call(), auto)
    with pytest.raises(ValueError, match="softmax path"):
        model.set_softmax_path("fast")
    # LayerNorm parameters that do NOT prove the bound: "auto" falls to the tested launch by itself
    with torch.no_grad():
        model.transformer_blocks[0].attn1.norm_q.weight.mul_(40.0)
    assert model.softmax_path_in_use()["self"] == "unproven"
