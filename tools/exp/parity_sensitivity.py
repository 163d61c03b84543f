"""What does the deep-pipeline criterion (`tests/test_models_gpu._check_deep`) catch?  ("a deliberately wrong
epsilon or a dropped rounding point fails the suite — try one".)

Mutations are injected at the ops layer (monkeypatching the arguments the models hand to the kernels; the library is untouched), the
reference-run default-width cases of tests/test_default_width_gpu.py are re-run, and the three ratios of `_check_deep`
  mean|hip - fp32| / mean|contract - fp32|,  p99.9 ratio,  mean|hip - contract| / mean|contract - fp32|
are reported next to the round-3 factors (1.5 / 2 / 2) and the round-4 factors (1.15 / 1.3 / 1.5).  A mutation is "caught" when any
ratio exceeds its factor.  usage (GPU box): python tools/exp/parity_sensitivity.py  -> gpurun_out/r4_parity_sensitivity.json
"""
import ast, contextlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import default_cases as dc
from tests.conftest import load_golden
from oracle import transformer as otr, vae as ovae
from trajectorycrafter_amd import ops
from trajectorycrafter_amd.models.crosstransformer3d import CrossTransformer3DModel
from trajectorycrafter_amd.models.autoencoder_magvit import AutoencoderKLCogVideoX

BF, gpu = torch.bfloat16, torch.device("cuda:0")
OLD, NEW = (1.5, 2.0, 2.0), (1.15, 1.3, 1.5)


def ratios(got, con, ex):
    got, con, ex = got.float().cpu(), con.float().cpu(), ex.float().cpu()
    q = lambda t: float(torch.quantile(t.flatten()[:4_000_000], 0.999))
    e_h, e_c, e_hc = (got - ex).abs(), (con - ex).abs(), (got - con).abs()
    return [float(e_h.mean()) / float(e_c.mean()), q(e_h) / q(e_c), float(e_hc.mean()) / float(e_c.mean())]


@contextlib.contextmanager
def patched(name, wrapper):
    orig = getattr(ops, name)
    setattr(ops, name, wrapper(orig))
    try:
        yield
    finally:
        setattr(ops, name, orig)


def kw_scale(name, key, pos, factor=None, value=None):
    """ops.<name> with keyword / positional argument `key` (position `pos`) multiplied by `factor` or replaced by `value`."""
    def wrapper(orig):
        def f(*a, **kw):
            a = list(a)
            if key in kw:
                kw[key] = value if value is not None else kw[key] * factor
            elif pos is not None and len(a) > pos:
                a[pos] = value if value is not None else a[pos] * factor
            return orig(*a, **kw)
        return f
    return patched(name, wrapper)


def drop_bias_every(name, pos, key, nth):
    """ops.<name> without its bias on every `nth` call."""
    cnt = [0]
    def wrapper(orig):
        def f(*a, **kw):
            cnt[0] += 1
            if cnt[0] % nth == 0:
                a = list(a)
                if key in kw: kw[key] = None
                elif len(a) > pos: a[pos] = None
            return orig(*a, **kw)
        return f
    return patched(name, wrapper)


def main():
    t, meta = load_golden("transformer_default.safetensors")
    cfg = ast.literal_eval(meta["config"])
    sd = dc.transformer_weights()
    x = dc.transformer_inputs()
    model = CrossTransformer3DModel(**cfg); model.load_state_dict(sd, strict=True); model = model.to(gpu, BF).eval()
    rot = (t["rope_cos"].to(gpu), t["rope_sin"].to(gpu))
    sdg = {k: v.to(gpu) for k, v in sd.items()}
    con = otr.transformer_forward(sdg, cfg, *(x[n].to(gpu) for n in ("hidden_states", "encoder_hidden_states", "timestep", "inpaint_latents", "cross_latents")), rot, prec="bf16")
    fwd = lambda: model(x["hidden_states"].to(gpu, BF), x["encoder_hidden_states"].to(gpu, BF), x["timestep"].to(gpu), inpaint_latents=x["inpaint_latents"].to(gpu, BF),
                        cross_latents=x["cross_latents"].to(gpu, BF), image_rotary_emb=rot, return_dict=False)[0]
    tv, mv = load_golden("vae_default.safetensors")
    vsd = dc.vae_weights(); vi = dc.vae_inputs()
    vae = AutoencoderKLCogVideoX(); vae.load_state_dict(vsd, strict=True); vae = vae.to(gpu, BF).eval()
    vsdg = {k: v.to(gpu) for k, v in vsd.items()}
    vcon = ovae.vae_decode(vsdg, {}, vi["z"].to(gpu), prec="bf16")
    dec = lambda: vae.decode(vi["z"].to(gpu, BF)).sample

    T, V = "2-layer 5B-geometry transformer", "default-width VAE decode"
    cases = [(T, "none (the shipped path)", contextlib.nullcontext)]
    cases += [(T, f"q/k LayerNorm eps 1e-6 -> {e:g}", lambda e=e: kw_scale("qk_layernorm_rope", "eps", 9, value=e)) for e in (1e-5, 1e-3, 1e-2)]
    cases += [(T, f"LayerNormZero / AdaLN eps 1e-5 -> {e:g}", lambda e=e: kw_scale("layernorm_modulate", "eps", 3, value=e)) for e in (1e-4, 1e-3, 1e-2)]
    cases += [(T, f"attention scale x {f:g}", lambda f=f: kw_scale("qk_layernorm_rope", "q_scale", None, factor=f)) for f in (1.002, 1.005, 1.01, 1.02, 1.05)]
    cases += [(T, "bias dropped in every 7th Linear", lambda: drop_bias_every("gemm_bf16", 2, "bias", 7))]
    cases += [(T, "running-max softmax everywhere (a legitimate other path)", None)]
    cases += [(V, "none (the shipped path)", contextlib.nullcontext)]
    cases += [(V, f"GroupNorm eps 1e-6 -> {e:g}", lambda e=e: kw_scale("groupnorm_stats", "eps", 2, value=e)) for e in (1e-5, 1e-4, 1e-3, 1e-2)]
    cases += [(V, "bias dropped in every 9th conv", lambda: drop_bias_every("conv3d_cl", 2, "bias", 9))]
    out = {"criterion": "tests/test_models_gpu._check_deep: ratios (mean, p99.9, hip-vs-contract) against factors", "factors_round3": OLD, "factors_round4": NEW, "cases": []}
    for where, name, ctx in cases:
        if ctx is None:
            model.set_softmax_path("exact"); r = ratios(fwd(), con, t["out_sample"]); model.set_softmax_path("auto")
        else:
            with ctx():
                r = ratios(fwd(), con, t["out_sample"]) if where == T else ratios(dec(), vcon, tv["decoded"])
        rec = {"case": where, "mutation": name, "ratio_mean": round(r[0], 4), "ratio_p999": round(r[1], 4), "ratio_hip_vs_contract": round(r[2], 4),
               "caught_round3_factors": any(a > b for a, b in zip(r, OLD)), "caught_round4_factors": any(a > b for a, b in zip(r, NEW))}
        print(rec, flush=True)
        out["cases"].append(rec)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r4_parity_sensitivity.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
