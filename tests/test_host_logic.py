"""Host-side logic of the product package against the oracle (CPU only, no kernels launched)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import diffusers_restated as dr
from oracle import pipeline as opl
from trajectorycrafter_amd import init_weights as iw
from trajectorycrafter_amd.models.autoencoder_magvit import AutoencoderKLCogVideoX, upsample_t_map, zq_t_map
from trajectorycrafter_amd.models.crosstransformer3d import CrossTransformer3DModel, Timesteps
from trajectorycrafter_amd.models.pipeline_trajectorycrafter import (TrajCrafter_Pipeline, get_3d_rotary_pos_embed,
                                                                    get_resize_crop_region_for_grid, resize_mask)
from oracle.prec import Prec
from trajectorycrafter_amd.scheduler import DDIMScheduler


def test_scheduler_matches_oracle_tables():
    a, b = DDIMScheduler(), dr.DDIMScheduler()
    torch.testing.assert_close(a.alphas_cumprod, b.alphas_cumprod, rtol=0, atol=0)
    for n in (1, 2, 25, 50):
        a.set_timesteps(n), b.set_timesteps(n)
        assert a.timesteps.tolist() == b.timesteps.tolist()
        for t in a.timesteps.tolist():
            at, ap = a.coeffs(t)
            bt, bp = b.coeffs(t)
            assert at == float(bt) and ap == float(bp)
    assert a.init_noise_sigma == 1.0 and a.order == 1 and a.config.prediction_type == "v_prediction"
    with pytest.raises(ValueError):
        DDIMScheduler(prediction_type="epsilon")
    with pytest.raises(ValueError):
        a.set_timesteps(5000)


def test_scheduler_from_pretrained(tmp_path):
    d = tmp_path / "scheduler"
    d.mkdir()
    (d / "scheduler_config.json").write_text(json.dumps({"_class_name": "DDIMScheduler", "num_train_timesteps": 1000,
                                                         "beta_start": 0.00085, "beta_end": 0.012, "timestep_spacing": "trailing",
                                                         "rescale_betas_zero_snr": True, "snr_shift_scale": 1.0}))
    s = DDIMScheduler.from_pretrained(str(tmp_path), subfolder="scheduler")
    s.set_timesteps(50)
    assert s.timesteps[0] == 999


def test_cog_scheduler_matches_oracle_and_sampler_table(tmp_path):
    """`DDIM_Cog` (demo.py:652): product tables / step coefficients == the oracle's, float64; the sampler table of
    `trajectorycrafter_amd.run` resolves the reference's names, reads scheduler/scheduler_config.json, and says which of the
    reference's choices are not built."""
    from trajectorycrafter_amd.run import make_scheduler
    from trajectorycrafter_amd.scheduler import CogVideoXDDIMScheduler
    a, b = CogVideoXDDIMScheduler(), dr.CogVideoXDDIMScheduler()
    assert a.alphas_cumprod.dtype == torch.float64 and torch.equal(a.alphas_cumprod, b.alphas_cumprod)
    a3, b3 = CogVideoXDDIMScheduler(snr_shift_scale=3.0), dr.CogVideoXDDIMScheduler(snr_shift_scale=3.0)
    assert torch.equal(a3.alphas_cumprod, b3.alphas_cumprod) and not torch.equal(a3.alphas_cumprod, a.alphas_cumprod)
    assert a3.config.snr_shift_scale == 3.0 and a.config.timestep_spacing == "trailing"
    for n in (2, 50):
        a.set_timesteps(n), b.set_timesteps(n)
        assert a.timesteps.tolist() == b.timesteps.tolist()
        for t in a.timesteps.tolist():
            sa, sb, ca, cb = a.step_coeffs(t)
            at, ap = b.coeffs(t)
            assert sa == float(at ** 0.5) and sb == float((1 - at) ** 0.5)
            assert ca == float(((1 - ap) / (1 - at)) ** 0.5) and cb == float(ap ** 0.5 - at ** 0.5 * ((1 - ap) / (1 - at)) ** 0.5)
    sa, sb, ca, cb = a.step_coeffs(a.timesteps.tolist()[-1])
    assert ca == 0.0 and cb == 1.0                              # the last step lands on x0 (final alpha = 1)
    assert type(make_scheduler("DDIM_Origin", None)) is DDIMScheduler and type(make_scheduler("DDIM_Cog", None)) is CogVideoXDDIMScheduler
    d = tmp_path / "scheduler"
    d.mkdir()
    (d / "scheduler_config.json").write_text(json.dumps({"_class_name": "CogVideoXDDIMScheduler", "snr_shift_scale": 3.0,
                                                         "clip_sample_range": 1.0, "sample_max_value": 1.0, "trained_betas": None}))
    assert make_scheduler("DDIM_Cog", str(tmp_path)).config.snr_shift_scale == 3.0
    from trajectorycrafter_amd.scheduler import PNDMScheduler
    assert type(make_scheduler("PNDM", None)) is PNDMScheduler
    with pytest.raises(ValueError, match="unknown sampler"):
        make_scheduler("LCM", None)

    class _Foreign:
        init_noise_sigma = 1.0
    with pytest.raises(NotImplementedError, match="DDIM_Cog"):
        TrajCrafter_Pipeline(None, None, None, None, _Foreign())


def test_scheduler_add_noise_matches_oracle_in_the_sample_dtype():
    """`add_noise` of the `strength < 1` branch: schedule cast to the sample dtype, every op in it (CPU tensors here: torch glue)."""
    from trajectorycrafter_amd.scheduler import CogVideoXDDIMScheduler
    g = torch.Generator().manual_seed(3)
    x0, nz = torch.randn(1, 3, 16, 4, 6, generator=g), torch.randn(1, 3, 16, 4, 6, generator=g)
    for prod, orc in ((DDIMScheduler(), dr.DDIMScheduler()), (CogVideoXDDIMScheduler(), dr.CogVideoXDDIMScheduler())):
        for t in (999, 499, 19):
            for dt, mode in ((torch.bfloat16, "bf16"), (torch.float32, "fp32")):
                got = prod.add_noise(x0.to(dt), nz.to(dt), torch.tensor([t]))
                want = orc.add_noise(Prec(mode), x0.to(dt).float(), nz.to(dt).float(), t)
                assert got.dtype == dt and torch.equal(got.float(), want), (type(prod).__name__, t, mode)
    a = DDIMScheduler().add_noise(x0, nz, torch.tensor([999]))
    torch.testing.assert_close(a, nz, rtol=0, atol=1e-6)                 # zero terminal SNR: pure noise at t = 999


def test_run_cli_parses_the_reference_options():
    from trajectorycrafter_amd.run import parse
    a = parse(["generate", "--model-dir", "ckpt", "--conditioning", "c.safetensors", "--out", "o.safetensors", "--sampler", "DDIM_Cog",
               "--steps", "30", "--seed", "7"])
    assert (a.cmd, a.sampler, a.steps, a.seed, a.transformer_dir, a.guidance_scale) == ("generate", "DDIM_Cog", 30, 7, None, None)
    o = parse(["orbits", "--model-dir", "ckpt", "--clip", "clip.safetensors", "--out", "o.safetensors", "--variants", "left_-30,right_90",
               "--radius", "0.5"])
    assert (o.cmd, o.variants, o.radius, o.no_mask) == ("orbits", "left_-30,right_90", 0.5, False)
    with pytest.raises(SystemExit):
        parse(["generate", "--out", "o"])


@pytest.mark.parametrize("hw", [(480, 720), (384, 672), (256, 256)])
def test_rope_tables_match_oracle(hw):
    H, W = hw
    cos, sin = opl.prepare_rotary(H, W, 13, 2, 64)
    gh, gw = H // 16, W // 16
    crops = get_resize_crop_region_for_grid((gh, gw), 45, 30)
    c2, s2 = get_3d_rotary_pos_embed(64, crops, (gh, gw), 13)
    assert torch.equal(cos, c2) and torch.equal(sin, s2)


def test_timesteps_module_matches_oracle():
    t = torch.tensor([999, 19, 0])
    torch.testing.assert_close(Timesteps(3072, True, 0)(t), dr.timesteps_proj(t, 3072, True, 0), rtol=0, atol=0)


@pytest.mark.parametrize("T", [1, 2, 3, 4, 5, 8, 9])
def test_temporal_index_maps_match_interpolate(T):
    x = torch.arange(T, dtype=torch.float32)[None, None, :, None, None].expand(1, 1, T, 2, 2)
    for compress in (True, False):
        ref = dr.upsample3d_nearest(x, compress)[0, 0, :, 0, 0].long().tolist()
        assert upsample_t_map(T, compress) == ref
    for Tz in (1, 2, 3):
        z = torch.arange(Tz, dtype=torch.float32)[None, None, :, None, None].expand(1, 1, Tz, 2, 2)
        if T > 1 and T % 2 == 1:
            if Tz < 2:
                continue
            first = F.interpolate(z[:, :, :1], size=(1, 2, 2))
            rest = F.interpolate(z[:, :, 1:], size=(T - 1, 2, 2))
            ref = torch.cat([first, rest], 2)
        else:
            ref = F.interpolate(z, size=(T, 2, 2))
        assert zq_t_map(T, Tz) == ref[0, 0, :, 0, 0].long().tolist()


def test_resize_mask_matches_oracle():
    m = (torch.rand(1, 1, 9, 16, 24) > 0.5).float()
    lat = torch.zeros(1, 16, 3, 2, 3)
    torch.testing.assert_close(resize_mask(m, lat), opl.resize_mask(m, lat))


def test_config_and_state_dict_round_trip(tmp_path):
    cfg = dict(num_attention_heads=2, num_layers=2, in_channels=33, text_embed_dim=32, time_embed_dim=32,
               use_rotary_positional_embeddings=True, is_train_cross=True, cross_attn_dim_head=64, cross_attn_num_heads=2)
    m = CrossTransformer3DModel(**cfg)
    assert m.config.patch_size == 2 and m.config["in_channels"] == 33 and m.config.num_layers == 2
    sd = iw.random_state_dict(iw.transformer_param_shapes(dict(m.config)), 3)
    m.load_state_dict(sd, strict=True)
    m.save_pretrained(str(tmp_path / "tr"))
    assert json.load(open(tmp_path / "tr" / "config.json"))["cross_attn_dim_head"] == 64
    m2 = CrossTransformer3DModel.from_pretrained(str(tmp_path), subfolder="tr")
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    # reference from_pretrained_2d semantics: patch_embed input channels zero-padded when the checkpoint has fewer
    small = dict(sd)
    small["patch_embed.proj.weight"] = sd["patch_embed.proj.weight"][:, :16].clone()
    from safetensors.torch import save_file
    os.makedirs(tmp_path / "tr16")
    save_file(small, str(tmp_path / "tr16" / "diffusion_pytorch_model.safetensors"))
    json.dump(dict(m.config), open(tmp_path / "tr16" / "config.json", "w"))
    m3 = CrossTransformer3DModel.from_pretrained_2d(str(tmp_path), subfolder="tr16")
    w = m3.state_dict()["patch_embed.proj.weight"]
    assert torch.equal(w[:, :16], small["patch_embed.proj.weight"]) and float(w[:, 16:].abs().max()) == 0
    with pytest.raises(RuntimeError):
        CrossTransformer3DModel.from_pretrained_2d(str(tmp_path / "nope"))
    # attribute surface used by callers: length-dynamic module lists, processor API
    assert len(m.transformer_blocks) == 2 and len(m.perceiver_cross_attention) == 1
    assert len(m.attn_processors) == 2
    with pytest.raises(ValueError):
        m.set_attn_processor({"x": 1})
    v = AutoencoderKLCogVideoX(block_out_channels=(8, 16, 16, 32), norm_num_groups=4, layers_per_block=1)
    assert v.config.scaling_factor == 1.15258426 and list(v.config.block_out_channels) == [8, 16, 16, 32]
    # tile geometry as the reference derives it (:1081-1098, 1109-1153): half the sample size, /8 in latent rows / columns
    assert (v.tile_sample_min_height, v.tile_sample_min_width, v.tile_latent_min_height, v.tile_latent_min_width) == (240, 360, 30, 45)
    assert not v.use_tiling and (v.tile_overlap_factor_height, v.tile_overlap_factor_width) == (1 / 6, 1 / 5)
    v.enable_tiling(tile_sample_min_height=96, tile_overlap_factor_width=0.25)
    assert v.use_tiling and (v.tile_sample_min_height, v.tile_latent_min_height, v.tile_sample_min_width) == (96, 12, 360)
    assert (v.tile_overlap_factor_height, v.tile_overlap_factor_width) == (1 / 6, 0.25)
    v.disable_tiling()
    assert not v.use_tiling


def test_loaders_report_and_refuse_incomplete_checkpoints(tmp_path, capsys):
    """A checkpoint that does not cover the model must not load silently (reference loaders only print counts, :952-973):
    missing keys raise unless explicitly allowed, and the key names are printed."""
    cfg = dict(num_attention_heads=2, num_layers=2, in_channels=33, text_embed_dim=32, time_embed_dim=32,
               use_rotary_positional_embeddings=True, is_train_cross=True, cross_attn_dim_head=64, cross_attn_num_heads=2)
    m = CrossTransformer3DModel(**cfg)
    m.load_state_dict(iw.random_state_dict(iw.transformer_param_shapes(dict(m.config)), 3), strict=True)
    d = tmp_path / "sharded"
    m.save_pretrained(str(d), max_shard_size=200_000)                         # diffusers sharded layout + index.json
    files = sorted(os.listdir(d))
    assert "diffusion_pytorch_model.safetensors.index.json" in files and sum(f.endswith(".safetensors") for f in files) >= 3
    m2 = CrossTransformer3DModel.from_pretrained(str(d))
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    # (1) a parameter of the base model missing -> error naming it, from every loader
    from safetensors.torch import load_file, save_file
    idx = json.load(open(d / "diffusion_pytorch_model.safetensors.index.json"))
    victim = "transformer_blocks.1.ff.net.2.weight"
    shard = idx["weight_map"][victim]
    t = load_file(str(d / shard))
    t.pop(victim)
    save_file(t, str(d / shard))
    for load in (CrossTransformer3DModel.from_pretrained, CrossTransformer3DModel.from_pretrained_2d,
                 CrossTransformer3DModel.from_pretrained_cus):
        with pytest.raises(RuntimeError, match="transformer_blocks.1.ff.net.2.weight"):
            load(str(d))
    assert victim in capsys.readouterr().out
    assert CrossTransformer3DModel.from_pretrained(str(d), allow_missing=("transformer_blocks.1.ff.",)) is not None
    # (2) the base CogVideoX-Fun checkpoint lacks the cross-attention modules this model adds: from_pretrained_2d / _cus accept
    #     exactly that (the reference's use case), plain from_pretrained does not
    base = {k: v for k, v in m.state_dict().items() if not k.startswith(CrossTransformer3DModel.NEW_CROSS_MODULES)}
    b = tmp_path / "base"
    os.makedirs(b)
    save_file({k: v.contiguous() for k, v in base.items()}, str(b / "diffusion_pytorch_model.safetensors"))
    json.dump(dict(m.config), open(b / "config.json", "w"))
    m3 = CrossTransformer3DModel.from_pretrained_2d(str(b))
    assert all(torch.equal(m3.state_dict()[k], v) for k, v in base.items())
    assert "perceiver_cross_attention.0.to_q.weight" in capsys.readouterr().out
    with pytest.raises(RuntimeError, match="perceiver_cross_attention"):
        CrossTransformer3DModel.from_pretrained(str(b))
    # (3) from_pretrained_cus(config_path=...): config.json read from another directory (reference :977-1003)
    c = tmp_path / "cfgdir"
    os.makedirs(c)
    json.dump(dict(m.config, num_layers=1), open(c / "config.json", "w"))
    m4 = CrossTransformer3DModel.from_pretrained_cus(str(b), config_path=str(c))
    assert len(m4.transformer_blocks) == 1
    with pytest.raises(RuntimeError, match="does not exist"):
        CrossTransformer3DModel.from_pretrained_cus(str(b), config_path=str(tmp_path / "nowhere"))
    # (4) a shape mismatch is an error for from_pretrained (diffusers semantics) unless ignore_mismatched_sizes
    bad = dict(m.state_dict())
    bad["proj_out.weight"] = torch.zeros(8, 128)
    e = tmp_path / "bad"
    os.makedirs(e)
    save_file({k: v.contiguous() for k, v in bad.items()}, str(e / "diffusion_pytorch_model.safetensors"))
    json.dump(dict(m.config), open(e / "config.json", "w"))
    with pytest.raises(RuntimeError, match="proj_out.weight"):
        CrossTransformer3DModel.from_pretrained(str(e))
    assert CrossTransformer3DModel.from_pretrained(str(e), ignore_mismatched_sizes=True) is not None


def test_add_noise_to_reference_video_matches_reference_formula():
    """reference pipeline :163-175 (applied at :488-491 when transformer.config.add_noise_in_inpaint_model)."""
    from trajectorycrafter_amd.models.pipeline_trajectorycrafter import add_noise_to_reference_video
    img = torch.rand(2, 3, 5, 8, 8) * 2 - 1
    img[:, :, :, :4] = -1.0                                              # masked-out pixels stay exactly -1
    torch.manual_seed(9)
    out = add_noise_to_reference_video(img, ratio=0.0563)
    torch.manual_seed(9)
    want = img + torch.where(img == -1, torch.zeros_like(img), torch.randn_like(img) * 0.0563)
    assert torch.equal(out, want) and bool((out[:, :, :, :4] == -1).all()) and not torch.equal(out, img)
    torch.manual_seed(9)
    o2 = add_noise_to_reference_video(img)                               # ratio None: sigma = exp(N(-3, 0.5)) per sample
    torch.manual_seed(9)
    sigma = torch.exp(torch.normal(mean=-3.0, std=0.5, size=(2,)))
    want2 = img + torch.where(img == -1, torch.zeros_like(img), torch.randn_like(img) * sigma[:, None, None, None, None])
    assert torch.equal(o2, want2)


def test_attention_bound_proof_from_layernorm_parameters():
    """TCX_ATTN_BOUND_PROVEN is only passed when the q/k LayerNorm parameters prove |q| |k| < 60 for every input (|LN(x)|_2 <= 8)."""
    from trajectorycrafter_amd.models.crosstransformer3d import Attention, LOG2E
    a = Attention(query_dim=128, dim_head=64, heads=2)
    qs = 64 ** -0.5 * LOG2E
    assert a._bound_is_proven(qs)                                    # gamma = 1, beta = 0: 0.18 * 8 * 8 = 11.5
    # the bound really bounds: random inputs through torch's LayerNorm never exceed it
    with torch.no_grad():
        a.norm_q.weight.mul_(1.3); a.norm_q.bias.add_(0.1); a.norm_k.weight.mul_(0.7); a.norm_k.bias.sub_(0.2)
    assert a._bound_is_proven(qs)                                    # the in-place edits bumped the parameter versions
    x = torch.randn(4096, 64) * torch.rand(4096, 1) * 50
    nq = (a.norm_q(x) * qs).norm(dim=-1).max()
    nk = a.norm_k(x).norm(dim=-1).max()
    r = 8.0
    with torch.no_grad():
        bq = qs * float(r * a.norm_q.weight.abs().max() + a.norm_q.bias.norm())
        bk = float(r * a.norm_k.weight.abs().max() + a.norm_k.bias.norm())
        assert float(nq) <= bq and float(nk) <= bk and float(nq * nk) < 60
    with torch.no_grad():
        a.norm_k.weight.mul_(5.0)                                    # 0.18 * (8 * 1.3 + ..) * (8 * 3.5 + ..) > 60: no proof
    assert not a._bound_is_proven(qs)


def test_pipeline_rejects_cpu_models_and_checks_inputs():
    from trajectorycrafter_amd._lib import TcxError
    tr = CrossTransformer3DModel(num_attention_heads=2, num_layers=1, in_channels=33, text_embed_dim=32, time_embed_dim=32,
                                 use_rotary_positional_embeddings=True)
    vae = AutoencoderKLCogVideoX(block_out_channels=(8, 16, 16, 32), norm_num_groups=4, layers_per_block=1)
    pipe = TrajCrafter_Pipeline(None, None, vae, tr)
    assert pipe.vae_scale_factor_spatial == 8 and pipe.vae_scale_factor_temporal == 4
    pe = torch.zeros(1, 10, 32)
    with pytest.raises(TcxError, match="no CPU fallback"):
        pipe(prompt=None, prompt_embeds=pe, negative_prompt_embeds=pe, height=32, width=48, num_frames=9)
    with pytest.raises(ValueError):
        pipe(prompt="a", prompt_embeds=pe, height=32, width=48, num_frames=9)
    with pytest.raises(ValueError, match="no text encoder"):
        pipe.encode_prompt("hello", None, True, device="cpu")


def test_gemm_row_geometry_decoding():
    """Host logic of ops.gemm_bf16: how [.., N] views are mapped onto the C ABI's (rows_per_batch, ld, stride_b)."""
    from trajectorycrafter_amd.ops import _gemm_rows, gemm_supported
    from trajectorycrafter_amd._lib import TcxError
    N = 16
    assert _gemm_rows(torch.zeros(6, N), "t", N, 6) == (0, N, 0)                       # flat
    assert _gemm_rows(torch.zeros(2, 3, N), "t", N, 6) == (0, N, 0)                    # contiguous [B, rows, N] collapses
    assert _gemm_rows(torch.zeros(6, 3 * N)[:, N:2 * N], "t", N, 6) == (0, 3 * N, 0)   # column slice: ld > N
    assert _gemm_rows(torch.zeros(2, 5, N)[:, 2:], "t", N, 6) == (3, N, 5 * N)         # row range of a joint buffer
    assert _gemm_rows(torch.zeros(1, 5, N)[:, 2:], "t", N, 3) == (0, N, 0)             # batch of one: flat again
    with pytest.raises(TcxError):
        _gemm_rows(torch.zeros(6, N).t(), "t", 6, N)                                   # last dim not contiguous
    with pytest.raises(TcxError):
        _gemm_rows(torch.zeros(2, 2, 5, N)[:, :, 1:], "t", N, 16)                      # two strided levels
    assert gemm_supported(3072, 3072) and gemm_supported(64, 3072) and gemm_supported(3072, 256)
    assert not gemm_supported(3072, 132) and not gemm_supported(60, 128)


def test_warper_and_driver_need_a_gpu():
    """No CPU fallback in the product path: the CPU route is the oracle (oracle.warp)."""
    from trajectorycrafter_amd.models.utils import Warper
    from trajectorycrafter_amd._lib import TcxError
    w = Warper(device="cpu")
    f, d = torch.zeros(1, 3, 4, 4), torch.ones(1, 1, 4, 4)
    eye, k = torch.eye(4)[None], torch.eye(3)[None]
    with pytest.raises(TcxError):
        w.forward_warp(f, None, d, eye, eye, k, None, False, twice=False)
    with pytest.raises(NotImplementedError, match="mask=False"):          # twice=True is built for mask=False only (checked before the device)
        w.forward_warp(f, None, d, eye, eye, k, None, True, twice=True)
    with pytest.raises(TcxError):
        w.forward_warp(f, None, d, eye, eye, k, None, False, twice=True)


def test_sincos_table_and_position_rows_match_oracle():
    """Product `get_3d_sincos_pos_embed` == the oracle's restatement at fp32; the rows the non-rotary model adds
    (`CrossTransformer3DModel._position_rows`: buffer in bf16 -> trilinear resize -> cut) == oracle.transformer.sincos_position_table
    under the bf16 contract, at the configured size (identity resize) and at a smaller latent with fewer frames."""
    from oracle import transformer as otr
    from trajectorycrafter_amd.models.crosstransformer3d import get_3d_sincos_pos_embed
    for D, sz, T, ss, ts in ((128, (6, 4), 3, 1.875, 1.0), (1920, (45, 30), 2, 1.875, 1.0), (64, (5, 7), 2, 1.0, 2.0)):
        a = torch.from_numpy(dr.get_3d_sincos_pos_embed(D, sz, T, ss, ts)).float()
        assert torch.equal(get_3d_sincos_pos_embed(D, sz, T, ss, ts).float(), a)
    cfg = dict(num_attention_heads=2, num_layers=1, in_channels=33, text_embed_dim=32, time_embed_dim=32, max_text_seq_length=10,
               sample_width=12, sample_height=8, sample_frames=9, use_rotary_positional_embeddings=False)
    m = CrossTransformer3DModel(**cfg)
    assert m.pos_embedding.shape == (1, 10 + 3 * 4 * 6, 128) and "pos_embedding" not in m.state_dict()
    assert float(m.pos_embedding[:, :10].abs().max()) == 0
    m = m.to(torch.bfloat16)
    ocfg = dict(otr.DEFAULT_CONFIG)
    ocfg.update(cfg)
    for Tn, h, w in ((3, 8, 12), (2, 6, 10), (3, 16, 8)):
        rows = m._position_rows(10, Tn, h, w, torch.device("cpu"))
        want = otr.sincos_position_table(Prec("bf16"), ocfg, 128, h, w, 10 + Tn * h * w // 4)
        assert rows.dtype == torch.bfloat16 and rows.shape == want.shape and torch.equal(rows.float(), want), (Tn, h, w)
    with pytest.raises(ValueError, match="max_text_seq_length"):
        m._position_rows(7, 3, 8, 12, torch.device("cpu"))
    assert not hasattr(CrossTransformer3DModel(**dict(cfg, use_rotary_positional_embeddings=True)), "pos_embedding")


def test_sigma_samplers_match_oracle_tables(tmp_path):
    """"Euler" / "Euler A" / "DPM++" (demo.py:647-654): timesteps (values and dtype), sigma tables, init_noise_sigma and every step's
    fp32 coefficients of the product schedulers == the oracle's restatement, bit for bit; sampler table + from_pretrained on the
    CogVideoX scheduler_config.json (unknown keys ignored, like the library)."""
    from trajectorycrafter_amd import scheduler as S
    from trajectorycrafter_amd.run import make_scheduler
    pairs = ((S.EulerDiscreteScheduler, dr.EulerDiscreteScheduler), (S.EulerAncestralDiscreteScheduler, dr.EulerAncestralDiscreteScheduler),
             (S.DPMSolverMultistepScheduler, dr.DPMSolverMultistepScheduler))
    for pc, oc in pairs:
        a, b = pc(), oc()
        for n in (1, 2, 25, 50):
            a.set_timesteps(n), b.set_timesteps(n)
            assert a.timesteps.dtype == b.timesteps.dtype and torch.equal(a.timesteps, b.timesteps) and torch.equal(a.sigmas, b.sigmas)
            assert float(a.init_noise_sigma) == float(b.init_noise_sigma)
            for t in a.timesteps.tolist():
                cb = b.step_coeffs(t)
                want = [float(v) for v in cb[:4]] + [float(cb[4]) if cb[4] is not None else 0.0]
                if pc is S.DPMSolverMultistepScheduler:
                    ca, second = a.step_coeffs(t)
                    assert ca == want and second == (cb[4] is not None), (n, t)
                    a._lower_order_nums = min(a._lower_order_nums + 1, 2)
                    b.lower_order_nums = min(b.lower_order_nums + 1, 2)
                else:
                    assert a.step_coeffs(t) == want, (n, t)
        with pytest.raises(ValueError, match="not on the schedule"):
            a.step_coeffs(998)
        # add_noise (strength < 1): arithmetic in the sample dtype, bit-equal to the oracle's bf16 / fp32 evaluation
        g = torch.Generator().manual_seed(4)
        x0, nz = torch.randn(1, 3, 16, 4, 6, generator=g), torch.randn(1, 3, 16, 4, 6, generator=g)
        a.set_timesteps(50), b.set_timesteps(50)
        for t in (999, 499, 19):
            for dt, mode in ((torch.bfloat16, "bf16"), (torch.float32, "fp32")):
                got = a.add_noise(x0.to(dt), nz.to(dt), a.timesteps[a._i(t)].reshape(1))
                want = b.add_noise(Prec(mode), x0.to(dt).float(), nz.to(dt).float(), t)
                assert got.dtype == dt and torch.equal(got.float(), want), (pc.__name__, t, mode)
    d = tmp_path / "scheduler"
    d.mkdir()
    (d / "scheduler_config.json").write_text(json.dumps({"_class_name": "CogVideoXDDIMScheduler", "_diffusers_version": "0.31.0.dev0",
        "beta_end": 0.012, "beta_schedule": "scaled_linear", "beta_start": 0.00085, "clip_sample": False, "clip_sample_range": 1.0,
        "num_train_timesteps": 1000, "prediction_type": "v_prediction", "rescale_betas_zero_snr": True, "sample_max_value": 1.0,
        "set_alpha_to_one": True, "snr_shift_scale": 1.0, "steps_offset": 0, "timestep_spacing": "trailing", "trained_betas": None}))
    for name, pc in (("Euler", S.EulerDiscreteScheduler), ("Euler A", S.EulerAncestralDiscreteScheduler), ("DPM++", S.DPMSolverMultistepScheduler)):
        sch = make_scheduler(name, str(tmp_path))
        assert type(sch) is pc and type(make_scheduler(name, None)) is pc and sch.config.timestep_spacing == "trailing"
        sch.set_timesteps(50)
        assert int(sch.timesteps[0]) == 999


@pytest.mark.parametrize("T", list(range(1, 14)))
def test_decoded_frame_count_matches_the_chunked_temporal_upsampling(T):
    """`AutoencoderKLCogVideoX.decoded_frames` (sizes the fused frames epilogue's output) == the frame count of the reference's chunking
    (:1235-1241) pushed through two temporal nearest upsamplings of diffusers CogVideoXUpsample3D (oracle restatement): 4 (T - 1) + 1
    for odd T, 4 T for even T (whose first chunk has an even frame count)."""
    v = AutoencoderKLCogVideoX(block_out_channels=(8, 16, 16, 32), norm_num_groups=4, layers_per_block=1)
    chunks = [1] if T == 1 else [2 + (T % 2 if i == 0 else 0) for i in range(T // 2)]
    want = 0
    for t in chunks:
        x = torch.zeros(1, 1, t, 2, 2)
        for _ in range(2):
            x = dr.upsample3d_nearest(x, True)[:, :, :, :2, :2]
        want += x.shape[2]
    assert v.decoded_frames(T) == want == (1 if T == 1 else (4 * (T - 1) + 1 if T % 2 else 4 * T))


def test_pndm_schedule_and_coefficients_match_oracle():
    """"PNDM" (demo.py:651): the 59-long evaluation schedule of 50 inference steps (12 Runge-Kutta evaluations + 47 multistep updates) and
    every `_get_prev_sample` coefficient set of the product scheduler == the oracle's restatement, bit for bit; add_noise as DDIM's."""
    from trajectorycrafter_amd.scheduler import PNDMScheduler
    a, b = PNDMScheduler(), dr.PNDMScheduler()
    assert torch.equal(a.alphas_cumprod, b.alphas_cumprod) and float(a.alphas_cumprod[-1]) > 1e-3          # no zero-terminal-SNR rescale
    for n in (4, 10, 25, 50):
        a.set_timesteps(n), b.set_timesteps(n)
        assert a.timesteps.dtype == torch.int64 and torch.equal(a.timesteps, b.timesteps) and len(a.timesteps) == n + 9
    assert a.timesteps[:13].tolist() == [999, 989, 989, 979, 979, 969, 969, 959, 959, 949, 949, 939, 939] and a.timesteps[-1] == 19
    for t, pt in ((999, 989), (999, 979), (939, 919), (19, -1)):
        assert a.prev_coeffs(t, pt) == [float(v) for v in b.prev_coeffs(t, pt)]
    with pytest.raises(ValueError, match="at least"):
        a.set_timesteps(3)
    with pytest.raises(ValueError, match="skip_prk_steps"):
        PNDMScheduler(skip_prk_steps=True)
    g = torch.Generator().manual_seed(8)
    x0, nz = torch.randn(1, 3, 16, 4, 6, generator=g).to(torch.bfloat16), torch.randn(1, 3, 16, 4, 6, generator=g).to(torch.bfloat16)
    ddim_like = DDIMScheduler(rescale_betas_zero_snr=False)
    assert torch.equal(a.add_noise(x0, nz, torch.tensor([499])), ddim_like.add_noise(x0, nz, torch.tensor([499])))


def test_hashed_weight_stream_known_answers():
    """`init_weights.hashed_normal` / `hashed_state_dict` (the host-independent stream the default-width fixtures regenerate their
    weights from): fixed bit patterns (integer hashing + one fp32 multiply: the same on every host), the recipe's statistics, and the
    independence of streams / seeds / chunk boundaries."""
    from trajectorycrafter_amd import init_weights as iw
    x = iw.hashed_normal((2, 5), 3, 5)
    assert x.view(torch.int32).flatten().tolist() == [1038109939, 1069986748, 1069251804, -1086144235, 1066918607, -1078836870,
                                                      -1093203958, 1059999443, -1081601279, -1085032175]
    sd = iw.hashed_state_dict({"a.weight": (4, 6), "norm.weight": (6,), "b.bias": (5,)}, 9)
    assert iw.state_dict_digest(sd) == "7afc28ea34d0f60477ece4b9d7a0d892679ebf4933ccace851291682b126cf66"
    assert all(torch.equal(v, v.to(torch.bfloat16).float()) for v in sd.values())          # bf16-representable
    assert abs(float(sd["norm.weight"].mean()) - 1.0) < 0.2 and float(sd["b.bias"].abs().max()) < 0.1
    big = iw.hashed_normal((1 << 20,), 1, 2)
    assert abs(float(big.mean())) < 5e-3 and abs(float(big.std()) - 1.0) < 5e-3 and 3.0 < float(big.abs().max()) <= 3.4642
    # element i does not depend on how the tensor is chunked or shaped; streams and seeds differ
    n = (1 << 24) + 7
    long = iw.hashed_normal((n,), 1, 2)
    assert torch.equal(long[:1 << 20], big) and torch.equal(iw.hashed_normal((7, 3), 1, 2).flatten(), big[:21])
    assert not torch.equal(iw.hashed_normal((64,), 1, 3), big[:64]) and not torch.equal(iw.hashed_normal((64,), 2, 2), big[:64])
    assert torch.equal(iw.hashed_normal((4,), 1, 2, scale=0.5), big[:4] * 0.5)
