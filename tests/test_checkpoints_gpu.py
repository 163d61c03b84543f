"""SURVEY §8 rows f4 and f2 on the GPU: checkpoint directories in the diffusers layout (config.json + sharded
`*.safetensors` + index, `patch_embed.proj.weight` trained with another `in_channels`) loaded through the reference's loader
entry points (crosstransformer3d.py:873-1092) onto the MI355X must run bit-identically to the directly constructed model; the
`prompt=` path (pipeline :248-296) through a real `transformers.T5EncoderModel` (random-init, tiny: no checkpoints offline)
and the `.safetensors` conditioning hand-off (demo.py:94-148) through the whole pipeline."""
import ast
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

BF = torch.bfloat16
CFG = dict(num_attention_heads=2, num_layers=2, in_channels=33, text_embed_dim=32, time_embed_dim=32,
           use_rotary_positional_embeddings=True, is_train_cross=True, cross_attn_dim_head=64, cross_attn_num_heads=2)


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _inputs(gpu, seed=0):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g).to(gpu, BF)
    from oracle.pipeline import prepare_rotary
    cos, sin = prepare_rotary(32, 48, 3, 2, 64)
    return dict(hidden_states=rn(2, 3, 16, 4, 6), encoder_hidden_states=rn(2, 10, 32), timestep=torch.tensor([500, 500], device=gpu),
                inpaint_latents=rn(2, 3, 17, 4, 6), cross_latents=rn(2, 2, 16, 4, 6), image_rotary_emb=(cos.to(gpu), sin.to(gpu)), return_dict=False)


def test_sharded_checkpoint_dirs_load_and_run_bit_identically(gpu, tmp_path):
    from safetensors.torch import load_file, save_file
    from trajectorycrafter_amd import init_weights as iw
    from trajectorycrafter_amd.models.crosstransformer3d import CrossTransformer3DModel
    m = CrossTransformer3DModel(**CFG)
    sd = {k: v.to(BF) for k, v in iw.random_state_dict(iw.transformer_param_shapes(dict(m.config)), 5).items()}
    m.load_state_dict(sd, strict=True)
    m = m.to(gpu, BF).eval()
    inp = _inputs(gpu)
    want = m(**inp)[0]

    d = tmp_path / "transformer"
    m.save_pretrained(str(d), max_shard_size=150_000)
    assert sum(f.endswith(".safetensors") for f in os.listdir(d)) >= 3 and os.path.exists(d / "diffusion_pytorch_model.safetensors.index.json")
    for load in (lambda: CrossTransformer3DModel.from_pretrained(str(tmp_path), subfolder="transformer", torch_dtype=BF),     # demo.py:636
                 lambda: CrossTransformer3DModel.from_pretrained_2d(str(tmp_path), subfolder="transformer"),
                 lambda: CrossTransformer3DModel.from_pretrained_cus(str(tmp_path), subfolder="transformer")):
        m2 = load().to(gpu, BF).eval()
        assert torch.equal(m2(**inp)[0], want)

    # a checkpoint trained with in_channels = 17 (16 latent + 1 mask channels) into the 33-channel model: zero-padded input
    # channels (:944-951) == the 33-channel model with those weight columns zeroed
    idx = json.load(open(d / "diffusion_pytorch_model.safetensors.index.json"))
    shard = idx["weight_map"]["patch_embed.proj.weight"]
    t = load_file(str(d / shard))
    full = t["patch_embed.proj.weight"].clone()
    t["patch_embed.proj.weight"] = full[:, :17].contiguous()
    save_file(t, str(d / shard))
    m3 = CrossTransformer3DModel.from_pretrained_cus(str(d)).to(gpu, BF).eval()
    ref_sd = dict(m.state_dict())
    padded = torch.zeros_like(ref_sd["patch_embed.proj.weight"])
    padded[:, :17] = ref_sd["patch_embed.proj.weight"][:, :17]
    ref_sd["patch_embed.proj.weight"] = padded
    m_ref = CrossTransformer3DModel(**CFG)
    m_ref.load_state_dict(ref_sd, strict=True)
    m_ref = m_ref.to(gpu, BF).eval()
    got3 = m3(**inp)[0]
    assert torch.equal(got3, m_ref(**inp)[0]) and not torch.equal(got3, want)
    # ... and the other direction: the 33-channel checkpoint into a 17-channel model keeps the leading 17 (:952-958)
    t["patch_embed.proj.weight"] = full
    save_file(t, str(d / shard))
    m4 = CrossTransformer3DModel.from_pretrained_2d(str(d), transformer_additional_kwargs=dict(in_channels=17))
    assert torch.equal(m4.state_dict()["patch_embed.proj.weight"].to(BF), full[:, :17])
    inp17 = dict(inp, inpaint_latents=inp["inpaint_latents"][:, :, :1])
    assert torch.isfinite(m4.to(gpu, BF).eval()(**inp17)[0].float()).all()
    # plain from_pretrained refuses that shape mismatch
    with pytest.raises(RuntimeError, match="patch_embed.proj.weight"):
        CrossTransformer3DModel.from_pretrained(str(d), in_channels=17)


class _Tokenizer:
    """Stand-in for the T5 tokenizer's call surface (pipeline :261-272): bytes -> ids, pad / truncate to max_length."""

    def __call__(self, texts, padding=None, max_length=None, truncation=False, add_special_tokens=True, return_tensors="pt"):
        rows = []
        for s in texts:
            ids = [3 + (b % 90) for b in s.encode()] + ([1] if add_special_tokens else [])     # 1 = </s>
            if padding == "max_length":
                ids = (ids[:max_length - 1] + [1] if truncation and len(ids) > max_length else ids) + [0] * max(0, max_length - len(ids))
            rows.append(ids)
        n = max(len(r) for r in rows)
        out = type("Enc", (), {})()
        out.input_ids = torch.tensor([r + [0] * (n - len(r)) for r in rows])
        return out

    def batch_decode(self, ids):
        return ["".join(chr(32 + int(i) % 90) for i in row) for row in ids]


@pytest.fixture(scope="module")
def tiny_pipe(gpu, golden):
    from transformers import T5Config, T5EncoderModel
    from tests.test_models_gpu import _weights
    from trajectorycrafter_amd.models.autoencoder_magvit import AutoencoderKLCogVideoX
    from trajectorycrafter_amd.models.crosstransformer3d import CrossTransformer3DModel
    from trajectorycrafter_amd.models.pipeline_trajectorycrafter import TrajCrafter_Pipeline
    tt, mt = golden("transformer_tiny.safetensors")
    tv, mv = golden("vae_tiny.safetensors")
    tr_cfg, vae_cfg = ast.literal_eval(mt["config"]), ast.literal_eval(mv["config"])
    tr = CrossTransformer3DModel(**tr_cfg)
    tr.load_state_dict(_weights(tt), strict=True)
    vae = AutoencoderKLCogVideoX(**vae_cfg)
    vae.load_state_dict(_weights(tv), strict=True)
    torch.manual_seed(0)
    t5 = T5EncoderModel(T5Config(vocab_size=100, d_model=tr_cfg["text_embed_dim"], d_kv=8, d_ff=64, num_layers=2, num_heads=2,
                                 decoder_start_token_id=0, pad_token_id=0, eos_token_id=1)).to(gpu, BF).eval()
    return TrajCrafter_Pipeline(_Tokenizer(), t5, vae.to(gpu, BF).eval(), tr.to(gpu, BF).eval())


def test_prompt_path_through_t5(gpu, tiny_pipe, golden):
    """`pipe(prompt="...")`: tokenizer -> T5EncoderModel on the GPU -> [1,226,D] embeddings -> the denoise loop; equal to
    passing the same embeddings as `prompt_embeds=` (pipeline :831-843), different prompts give different videos."""
    tp, _ = golden("pipeline_tiny.safetensors")
    pipe = tiny_pipe
    kw = dict(height=32, width=48, num_frames=9, num_inference_steps=2, guidance_scale=6.0, latents=tp["latents0"].to(BF),
              video=tp["video"], mask_video=tp["mask_video"], reference=tp["reference"], output_type="pt")
    torch.manual_seed(1)
    a = pipe(prompt="a camera orbits a red car", negative_prompt="blurry", **kw).videos
    pe, ne = pipe.encode_prompt("a camera orbits a red car", "blurry", True, device=gpu)
    assert pe.shape == (1, 226, 32) and ne.shape == (1, 226, 32) and pe.dtype == BF and pe.is_cuda
    torch.manual_seed(1)
    b = pipe(prompt=None, prompt_embeds=pe, negative_prompt_embeds=ne, **kw).videos
    assert a.shape == (1, 3, 9, 32, 48) and torch.isfinite(a).all() and torch.equal(a, b)
    torch.manual_seed(1)
    c = pipe(prompt="an entirely different scene", **kw).videos            # negative prompt defaults to "" (:356)
    assert not torch.equal(a, c)
    torch.manual_seed(1)
    two = pipe(prompt=["a camera orbits a red car"], negative_prompt=["blurry"], **kw).videos    # list form
    assert torch.equal(two, a)
    with pytest.raises(ValueError, match="batch size"):
        pipe.encode_prompt(["x"], ["a", "b"], True, device=gpu)


def test_conditioning_safetensors_hand_off(gpu, tiny_pipe, golden, tmp_path):
    """demo.py:94-148 seam as a file: written by the conditioning stage, read by the denoiser process."""
    from trajectorycrafter_amd.conditioning import load_conditioning, save_conditioning
    tp, _ = golden("pipeline_tiny.safetensors")
    pipe = tiny_pipe
    f = str(tmp_path / "clip0.safetensors")
    save_conditioning(f, cond_video=tp["video"], cond_masks=tp["mask_video"], frames_ref=tp["reference"],
                      prompt_embeds=tp["prompt_embeds"], negative_prompt_embeds=tp["negative_prompt_embeds"],
                      latents=tp["latents0"], height=32, width=48, num_frames=9, num_inference_steps=2, guidance_scale=6.0)
    kw = load_conditioning(f, device=gpu)
    assert set(kw) >= {"video", "mask_video", "reference", "prompt_embeds", "negative_prompt_embeds", "latents", "height", "width"}
    assert kw["prompt"] is None and kw["video"].is_cuda and torch.equal(kw["video"].cpu(), tp["video"])
    direct = dict(prompt=None, height=32, width=48, num_frames=9, num_inference_steps=2, guidance_scale=6.0,
                  prompt_embeds=tp["prompt_embeds"].to(BF), negative_prompt_embeds=tp["negative_prompt_embeds"].to(BF),
                  latents=tp["latents0"].to(BF), video=tp["video"], mask_video=tp["mask_video"], reference=tp["reference"])
    torch.manual_seed(2)
    a = pipe(output_type="pt", **kw).videos
    torch.manual_seed(2)
    b = pipe(output_type="pt", **direct).videos
    assert torch.equal(a, b)
    # pre-encoded form + prompt text + seed in the header (what a data-parallel rank receives)
    inpaint, ref = pipe._build_conditioning(tp["video"], tp["mask_video"], tp["reference"], 32, 48, True, BF, gpu)
    g = str(tmp_path / "clip0_latents.safetensors")
    save_conditioning(g, inpaint_latents=inpaint, ref_latents=ref, prompt="a street at night", negative_prompt="blurry", seed=43,
                      height=32, width=48, num_frames=9, num_inference_steps=2)
    kw2 = load_conditioning(g, device=gpu)
    assert kw2["prompt"] == "a street at night" and isinstance(kw2["generator"], torch.Generator) and kw2["inpaint_latents"].dtype == BF
    out = pipe(output_type="pt", **kw2).videos
    kw3 = load_conditioning(g, device=gpu)
    assert torch.equal(out, pipe(output_type="pt", **kw3).videos) and torch.isfinite(out).all()
    with pytest.raises(ValueError, match="cond_video"):
        save_conditioning(str(tmp_path / "bad.safetensors"), prompt_embeds=tp["prompt_embeds"])
    with pytest.raises(ValueError, match="not a trajectorycrafter-conditioning"):
        load_conditioning(os.path.join(os.path.dirname(__file__), "golden", "pipeline_tiny.safetensors"))
