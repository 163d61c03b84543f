#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own classes.

Run ONLY in the build container (needs /root/reference; the GPU box never sees it):

    python tests/golden/make_golden.py

What is executed from the reference (imported, never copied): models/crosstransformer3d.py
(CogVideoXPatchEmbed, RefPatchEmbed, CogVideoXBlock, PerceiverCrossAttention,
CrossTransformer3DModel.forward), models/autoencoder_magvit.py (every VAE class,
encode/_decode) and models/pipeline_trajectorycrafter.py (TrajCrafter_Pipeline.__call__).

`diffusers` is not installed.  The names the reference imports from it are provided
here as *scaffolding*: config/model mixins, output dataclasses and thin nn.Module shells whose
arithmetic is oracle/diffusers_restated.py.  Consequently the fixtures pin the reference's own
code against the oracle's restatement of it; the diffusers-resident arithmetic itself stays
"parity unpinned" (covered by analytic KATs in tests/test_oracle_kat.py).

Fixtures are safetensors files: weights (bf16-representable, stored as bf16), inputs, outputs.
"""
from __future__ import annotations

import inspect
import os
import sys
import types
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import diffusers_restated as dr            # noqa: E402
from oracle.prec import Prec                          # noqa: E402
from trajectorycrafter_amd import init_weights as iw  # noqa: E402

P32 = Prec("fp32")


# --------------------------------------------------------------------------- scaffolding
class _Cfg(dict):
    __getattr__ = dict.__getitem__


def register_to_config(init):
    sig = inspect.signature(init)

    def wrapped(self, *a, **kw):
        ba = sig.bind(self, *a, **kw)
        ba.apply_defaults()
        cfg = _Cfg({k: v for k, v in ba.arguments.items() if k != "self"})
        object.__setattr__(self, "_cfg", cfg)
        init(self, *a, **kw)

    return wrapped


class ConfigMixin:
    @property
    def config(self):
        return self._cfg

    @classmethod
    def from_config(cls, config, **kw):
        c = dict(config)
        c.update(kw)
        return cls(**{k: v for k, v in c.items() if k in inspect.signature(cls.__init__).parameters})


class ModelMixin(nn.Module):
    @property
    def dtype(self):
        ps = list(self.parameters())
        return ps[0].dtype if ps else torch.float32

    @property
    def device(self):
        ps = list(self.parameters())
        return ps[0].device if ps else torch.device("cpu")


class _Logger:
    def __getattr__(self, _):
        return lambda *a, **k: None


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _sd(module, prefix=""):
    return {prefix + k: v for k, v in module.state_dict().items()}


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift, scale=1):
        super().__init__()
        self.a = (num_channels, flip_sin_to_cos, downscale_freq_shift, scale)

    def forward(self, t):
        return dr.timesteps_proj(t, *self.a)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu"):
        super().__init__()
        assert act_fn == "silu"
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, x, cond=None):
        return dr.timestep_embedding(P32, _sd(self), "", x)


class CogVideoXLayerNormZero(nn.Module):
    def __init__(self, conditioning_dim, embedding_dim, elementwise_affine=True, eps=1e-5, bias=True):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = nn.Linear(conditioning_dim, 6 * embedding_dim, bias=bias)
        self.norm = nn.LayerNorm(embedding_dim, eps=eps, elementwise_affine=elementwise_affine)
        self.eps = eps

    def forward(self, hidden_states, encoder_hidden_states, temb):
        return dr.layer_norm_zero(P32, _sd(self), "", hidden_states, encoder_hidden_states, temb, self.eps)


class AdaLayerNorm(nn.Module):
    def __init__(self, embedding_dim, output_dim, norm_elementwise_affine=True, norm_eps=1e-5, chunk_dim=0):
        super().__init__()
        assert chunk_dim == 1
        self.silu = nn.SiLU()
        self.linear = nn.Linear(embedding_dim, output_dim)
        self.norm = nn.LayerNorm(output_dim // 2, norm_eps, norm_elementwise_affine)
        self.eps = norm_eps

    def forward(self, x, temb=None):
        return dr.ada_layer_norm(P32, _sd(self), "", x, temb, self.eps)


class CogVideoXAttnProcessor2_0:
    pass


class Attention(nn.Module):
    def __init__(self, query_dim, dim_head, heads, qk_norm, eps, bias, out_bias, processor):
        super().__init__()
        assert qk_norm == "layer_norm"
        inner = dim_head * heads
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(query_dim, inner, bias=bias)
        self.to_v = nn.Linear(query_dim, inner, bias=bias)
        self.norm_q = nn.LayerNorm(dim_head, eps=eps)
        self.norm_k = nn.LayerNorm(dim_head, eps=eps)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=out_bias), nn.Dropout(0.0)])
        self.eps = eps

    def forward(self, hidden_states, encoder_hidden_states, image_rotary_emb=None):
        return dr.cogvideox_attention(P32, _sd(self), "", hidden_states, encoder_hidden_states, self.heads,
                                      image_rotary_emb, self.eps)


class _GELU(nn.Module):
    def __init__(self, d, inner, bias):
        super().__init__()
        self.proj = nn.Linear(d, inner, bias=bias)


class FeedForward(nn.Module):
    def __init__(self, dim, dropout=0.0, activation_fn="gelu-approximate", final_dropout=True, inner_dim=None, bias=True):
        super().__init__()
        assert activation_fn == "gelu-approximate"
        inner = inner_dim or 4 * dim
        self.net = nn.ModuleList([_GELU(dim, inner, bias), nn.Dropout(dropout), nn.Linear(inner, dim, bias=bias),
                                  nn.Dropout(dropout)])

    def forward(self, x):
        return dr.feed_forward(P32, _sd(self), "", x)


class CogVideoXUpsample3D(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, compress_time=False):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding)
        self.compress_time = compress_time

    def forward(self, x):
        return dr.upsample3d(P32, _sd(self), "", x, self.compress_time)


class CogVideoXDownsample3D(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=2, padding=0, compress_time=False):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding)
        self.compress_time = compress_time

    def forward(self, x):
        return dr.downsample3d(P32, _sd(self), "", x, self.compress_time)


@dataclass
class _Out:
    sample: torch.Tensor = None
    latent_dist: object = None

    def __getitem__(self, i):
        return [v for v in (self.sample, self.latent_dist) if v is not None][i]


class _Sched:
    """diffusers-shaped shell around the restated DDIMScheduler."""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self):
        self.s = dr.DDIMScheduler()

    def set_timesteps(self, n, device=None):
        self.s.set_timesteps(n)
        self.timesteps = self.s.timesteps

    def scale_model_input(self, x, t):
        return x

    def step(self, model_output, t, sample, eta=0.0, return_dict=False, generator=None):
        return (self.s.step(P32, model_output, int(t), sample, eta),)


class DiffusionPipeline:
    def register_modules(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    _execution_device = torch.device("cpu")
    device = torch.device("cpu")

    @property
    def dtype(self):
        return self.transformer.dtype

    def progress_bar(self, total=None):
        class _PB:
            def __enter__(s):
                return s

            def __exit__(s, *a):
                return False

            def update(s, *a):
                pass

        return _PB()

    def maybe_free_model_hooks(self):
        pass


class VaeImageProcessor:
    def __init__(self, vae_scale_factor=8, do_normalize=True, do_binarize=False, do_convert_grayscale=False):
        self.kw = dict(do_normalize=do_normalize, do_binarize=do_binarize)

    def preprocess(self, image, height=None, width=None):
        return dr.vae_image_preprocess(image, height, width, **self.kw)


def install_scaffolding(plain=False):
    """plain=False: the diffusers names are thin shells around the ORACLE's fp32 functions (the first fixture generation).
    plain=True: they are the plain-torch nn.Modules of tests/golden/diffusers_plain.py (no oracle code behind the reference's
    classes; runs in fp32 and, after `.to(bfloat16)`, as the reference's own eager bf16 execution)."""
    ident = lambda f: f  # noqa: E731
    g = globals()
    if plain:
        sys.path.insert(0, HERE)
        import diffusers_plain as dp
        names = {n: getattr(dp, n) for n in ("Timesteps", "TimestepEmbedding", "CogVideoXLayerNormZero", "AdaLayerNorm", "Attention",
                                             "FeedForward", "CogVideoXAttnProcessor2_0", "CogVideoXUpsample3D", "CogVideoXDownsample3D")}
        gauss, get_act = dp.DiagonalGaussianDistribution, dp.get_activation
    else:
        names = {n: g[n] for n in ("Timesteps", "TimestepEmbedding", "CogVideoXLayerNormZero", "AdaLayerNorm", "Attention",
                                   "FeedForward", "CogVideoXAttnProcessor2_0", "CogVideoXUpsample3D", "CogVideoXDownsample3D")}
        gauss, get_act = dr.DiagonalGaussian, (lambda n: nn.SiLU())
    Timesteps, TimestepEmbedding, CogVideoXLayerNormZero, AdaLayerNorm = (names[n] for n in (
        "Timesteps", "TimestepEmbedding", "CogVideoXLayerNormZero", "AdaLayerNorm"))
    Attention, FeedForward, CogVideoXAttnProcessor2_0 = names["Attention"], names["FeedForward"], names["CogVideoXAttnProcessor2_0"]
    CogVideoXUpsample3D, CogVideoXDownsample3D = names["CogVideoXUpsample3D"], names["CogVideoXDownsample3D"]
    _mod("diffusers")
    _mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config)
    _mod("diffusers.utils", is_torch_version=lambda *a: True, logging=types.SimpleNamespace(get_logger=lambda n: _Logger()),
         BaseOutput=object, replace_example_docstring=lambda s: ident, WEIGHTS_NAME="diffusion_pytorch_model.bin")
    _mod("diffusers.utils.torch_utils", maybe_allow_in_graph=ident,
         randn_tensor=lambda shape, generator=None, device=None, dtype=None: torch.randn(shape, generator=generator, dtype=dtype))
    _mod("diffusers.utils.accelerate_utils", apply_forward_hook=ident)
    _mod("diffusers.loaders")
    _mod("diffusers.loaders.single_file_model", FromOriginalModelMixin=type("FromOriginalModelMixin", (), {}))
    _mod("diffusers.models", AutoencoderKLCogVideoX=object)
    _mod("diffusers.models.attention", Attention=Attention, FeedForward=FeedForward)
    _mod("diffusers.models.attention_processor", AttentionProcessor=object,
         CogVideoXAttnProcessor2_0=CogVideoXAttnProcessor2_0, FusedCogVideoXAttnProcessor2_0=CogVideoXAttnProcessor2_0)
    _mod("diffusers.models.embeddings", TimestepEmbedding=TimestepEmbedding, Timesteps=Timesteps,
         get_3d_sincos_pos_embed=dr.get_3d_sincos_pos_embed,
         get_3d_rotary_pos_embed=lambda embed_dim, crops_coords, grid_size, temporal_size, use_real=True:
         dr.get_3d_rotary_pos_embed(embed_dim, crops_coords, grid_size, temporal_size))
    _mod("diffusers.models.modeling_outputs", Transformer2DModelOutput=_Out,
         AutoencoderKLOutput=lambda latent_dist: _Out(latent_dist=latent_dist))
    _mod("diffusers.models.modeling_utils", ModelMixin=ModelMixin)
    _mod("diffusers.models.normalization", AdaLayerNorm=AdaLayerNorm, CogVideoXLayerNormZero=CogVideoXLayerNormZero)
    _mod("diffusers.models.activations", get_activation=get_act)
    _mod("diffusers.models.downsampling", CogVideoXDownsample3D=CogVideoXDownsample3D)
    _mod("diffusers.models.upsampling", CogVideoXUpsample3D=CogVideoXUpsample3D)
    _mod("diffusers.models.autoencoders")
    _mod("diffusers.models.autoencoders.vae", DecoderOutput=_Out, DiagonalGaussianDistribution=gauss)
    _mod("diffusers.callbacks", MultiPipelineCallbacks=type("M", (), {}), PipelineCallback=type("PC", (), {}))
    _mod("diffusers.pipelines")
    _mod("diffusers.pipelines.pipeline_utils", DiffusionPipeline=DiffusionPipeline)
    _mod("diffusers.schedulers", CogVideoXDDIMScheduler=type("A", (), {}), CogVideoXDPMScheduler=type("B", (), {}))
    _mod("diffusers.video_processor", VideoProcessor=lambda vae_scale_factor=8: None)
    _mod("diffusers.image_processor", VaeImageProcessor=VaeImageProcessor)
    if "transformers" not in sys.modules:
        try:
            import transformers  # noqa: F401
        except Exception:
            _mod("transformers", T5EncoderModel=object, T5Tokenizer=object)


# --------------------------------------------------------------------------- fixtures
TINY_TR = dict(num_attention_heads=2, attention_head_dim=64, in_channels=33, out_channels=16, num_layers=2,
               text_embed_dim=32, time_embed_dim=32, max_text_seq_length=10, sample_width=12, sample_height=8,
               sample_frames=9, use_rotary_positional_embeddings=True, is_train_cross=True,
               cross_attn_in_channels=16, cross_attn_interval=2, cross_attn_dim_head=64, cross_attn_num_heads=2)
TINY_VAE = dict(block_out_channels=(8, 16, 16, 32), norm_num_groups=4, layers_per_block=1)


def bf16_round(sd):
    return {k: v.to(torch.bfloat16).float() for k, v in sd.items()}


def save(name, tensors, meta):
    from safetensors.torch import save_file
    out = {}
    for k, v in tensors.items():
        out[k] = v.contiguous()
    save_file(out, os.path.join(HERE, name), metadata={k: str(v) for k, v in meta.items()})
    sz = os.path.getsize(os.path.join(HERE, name))
    print(f"wrote {name}: {len(out)} tensors, {sz / 1e6:.2f} MB")


def main():
    if not os.path.isdir(REF):
        raise SystemExit("make_golden.py needs /root/reference (build container only)")
    install_scaffolding()
    sys.path.insert(0, REF)
    import warnings
    warnings.filterwarnings("ignore")
    from models.crosstransformer3d import CrossTransformer3DModel
    from models.autoencoder_magvit import AutoencoderKLCogVideoX
    from models import pipeline_trajectorycrafter as ref_pl

    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1234)

    # ---- transformer forward ------------------------------------------------
    tr_sd = bf16_round(iw.random_state_dict(iw.transformer_param_shapes(TINY_TR), seed=0))
    model = CrossTransformer3DModel(**TINY_TR).eval()
    missing = model.load_state_dict(tr_sd, strict=True)
    print("transformer load:", missing)
    B, T, h, w = 2, 3, 8, 12
    hs = torch.randn(B, T, 16, h, w, generator=g)
    enc = torch.randn(B, 10, 32, generator=g)
    inp = torch.randn(B, T, 17, h, w, generator=g)
    cross = torch.randn(B, 2, 16, h, w, generator=g)
    ts = torch.tensor([961, 961])
    from oracle.pipeline import prepare_rotary
    cos, sin = prepare_rotary(h * 8, w * 8, T, 2, 64)
    with torch.no_grad():
        out = model(hs, enc, ts, inpaint_latents=inp, cross_latents=cross, image_rotary_emb=(cos, sin),
                    return_dict=False)[0]
        # per-component taps from the reference's own sub-modules
        emb = model.time_embedding(model.time_proj(ts))
        pe = model.patch_embed(enc, torch.cat([hs, inp], 2))
        ref_tok = model.ref_patch_embed(cross)
        blk_h, blk_e = model.transformer_blocks[0](pe[:, 10:], pe[:, :10], emb, (cos, sin))
        ca = model.perceiver_cross_attention[0](ref_tok, blk_h)
    tens = {"w." + k: v.to(torch.bfloat16) for k, v in tr_sd.items()}
    tens.update(hidden_states=hs, encoder_hidden_states=enc, inpaint_latents=inp, cross_latents=cross,
                timestep=ts, rope_cos=cos, rope_sin=sin, out_sample=out, tap_patch_embed=pe,
                tap_ref_tokens=ref_tok, tap_block0_hidden=blk_h, tap_block0_encoder=blk_e, tap_cross0=ca)
    save("transformer_tiny.safetensors", tens, dict(config=repr(TINY_TR), source="reference CrossTransformer3DModel.forward"))

    # ---- VAE decode / encode --------------------------------------------------
    vae_sd = bf16_round(iw.random_state_dict(iw.vae_param_shapes(TINY_VAE), seed=1))
    vae = AutoencoderKLCogVideoX(**TINY_VAE).eval()
    print("vae load:", vae.load_state_dict(vae_sd, strict=True))
    z = torch.randn(1, 16, 5, 4, 6, generator=g)
    video = torch.rand(1, 3, 9, 32, 48, generator=g) * 2 - 1
    with torch.no_grad():
        dec = vae.decode(z).sample
        post = vae.encode(video)[0]
        dec1 = vae.decode(z[:, :, :1]).sample
    tens = {"w." + k: v.to(torch.bfloat16) for k, v in vae_sd.items()}
    tens.update(z=z, decoded=dec, decoded_single_frame=dec1, video=video, enc_mean=post.mean.contiguous(),
                enc_logvar=post.logvar.contiguous())
    save("vae_tiny.safetensors", tens, dict(config=repr(TINY_VAE), source="reference AutoencoderKLCogVideoX.decode/encode"))

    # ---- full pipeline (2 steps, CFG 6, 9 frames 32x48) -----------------------
    pipe = ref_pl.TrajCrafter_Pipeline(tokenizer=None, text_encoder=None, vae=vae, transformer=model, scheduler=_Sched())
    Fv, H, W = 9, 32, 48
    vid = torch.rand(1, 3, Fv, H, W, generator=g)
    mask = (torch.rand(1, 1, Fv, H // 8, W // 8, generator=g) < 0.3).float().repeat_interleave(8, 3).repeat_interleave(8, 4) * 255
    mask[:, :, 0] = 0
    ref = vid[:, :, :5].clone()
    pe_pos = torch.randn(1, 10, 32, generator=g)
    pe_neg = torch.randn(1, 10, 32, generator=g)
    lat0 = torch.randn(1, 3, 16, H // 8, W // 8, generator=g)
    torch.manual_seed(77)                                  # reference latents come from the GLOBAL rng (:886)
    with torch.no_grad():
        frames = pipe(prompt=None, negative_prompt=None, height=H, width=W, video=vid, mask_video=mask,
                      reference=ref, num_frames=Fv, num_inference_steps=2, guidance_scale=6.0,
                      latents=lat0.clone(), prompt_embeds=pe_pos, negative_prompt_embeds=pe_neg).videos
        torch.manual_seed(77)
        lat_out = pipe(prompt=None, negative_prompt=None, height=H, width=W, video=vid, mask_video=mask,
                       reference=ref, num_frames=Fv, num_inference_steps=2, guidance_scale=6.0,
                       latents=lat0.clone(), prompt_embeds=pe_pos, negative_prompt_embeds=pe_neg,
                       output_type="latent", return_dict=True).videos
    save("pipeline_tiny.safetensors",
         dict(video=vid, mask_video=mask, reference=ref, prompt_embeds=pe_pos, negative_prompt_embeds=pe_neg,
              latents0=lat0, frames=frames.float(), latents_out=lat_out.float()),
         dict(tr_config=repr(TINY_TR), vae_config=repr(TINY_VAE), steps=2, guidance_scale=6.0, global_seed=77,
              weights="transformer_tiny.safetensors + vae_tiny.safetensors",
              source="reference TrajCrafter_Pipeline.__call__"))


def make_warp():
    """warp_tiny.safetensors: the reference's own Warper.forward_warp(mask=False, twice=False) (models/utils.py:220-293)
    on a seeded scene: two views, a camera that moves sideways + yaws, depth with a near slab (occlusion ordering).
    models/utils.py imports cv2 / decord / skimage / torchvision at module level for its video IO helpers; none is used
    by the Warper path, so absent ones are stubbed with empty modules (scaffolding, like the diffusers stubs)."""
    if not os.path.isdir(REF):
        raise SystemExit("make_golden.py needs /root/reference (build container only)")
    import importlib
    for name in ("cv2", "decord", "skimage", "skimage.io", "torchvision", "torchvision.transforms", "PIL", "PIL.Image",
                 "matplotlib", "matplotlib.pyplot", "tqdm", "imageio"):
        try:
            importlib.import_module(name)
        except Exception:
            _mod(name, VideoReader=object, cpu=lambda *a: None, imread=None, ToTensor=object, Image=object)
    sys.path.insert(0, REF)
    from models.utils import Warper
    g = torch.Generator().manual_seed(4242)
    b, h, w = 2, 24, 40
    frame = torch.rand(b, 3, h, w, generator=g) * 2 - 1
    depth = 2.0 + torch.rand(b, 1, h, w, generator=g)
    depth[:, :, 6:14, 10:22] = 0.8                                  # near slab: must win the splat where it lands
    depth[1, :, 0:2, 0:3] = 1e-3
    k = torch.tensor([[30.0, 0, w / 2], [0, 30.0, h / 2], [0, 0, 1]])[None].repeat(b, 1, 1)
    t1 = torch.eye(4)[None].repeat(b, 1, 1)
    t2 = torch.eye(4)[None].repeat(b, 1, 1)
    for n, (yaw, tx, tz) in enumerate(((0.08, 0.25, -0.1), (-0.3, -0.6, 2.2))):   # second pose pushes points behind the camera
        c, s_ = np.cos(yaw), np.sin(yaw)
        t2[n, :3, :3] = torch.tensor([[c, 0, s_], [0, 1, 0], [-s_, 0, c]], dtype=torch.float32)
        t2[n, :3, 3] = torch.tensor([tx, 0.05, -tz])
    with torch.no_grad():
        warped, mask2, wdepth, flow = Warper(device="cpu").forward_warp(frame, None, depth, t1, t2, k, None, False, twice=False)
        tw_frame, tw_mask, tw_depth, tw_none = Warper(device="cpu").forward_warp(frame, None, depth, t1, t2, k, None, False, twice=True)
    assert tw_none is None
    save("warp_tiny.safetensors",
         dict(frame=frame, depth=depth, t1=t1, t2=t2, K=k, warped=warped.float(), mask2=mask2.float(),
              warped_depth=wdepth.float(), flow=flow.float(), twice_frame=tw_frame.float(), twice_mask=tw_mask.float(),
              twice_depth=tw_depth.float()),
         dict(source="reference models/utils.py Warper.forward_warp(frame, None, depth, t1, t2, K, None, False, twice=False) and (..., twice=True)",
              seed=4242))


def _stub_utils_imports():
    import importlib
    for name in ("cv2", "decord", "skimage", "skimage.io", "torchvision", "torchvision.transforms", "PIL", "PIL.Image",
                 "matplotlib", "matplotlib.pyplot", "tqdm", "imageio"):
        try:
            importlib.import_module(name)
        except Exception:
            _mod(name, VideoReader=object, cpu=lambda *a: None, imread=None, ToTensor=object, Image=object)


def make_poses():
    """orbit_poses.safetensors: the reference's own `generate_traj_specified` (models/utils.py:134-158, with `sphere2pose`
    :83-131) on the eight target poses of inference_orbits.py:258-283, at radius 1.0 and at the clamped / scaled radii
    `get_poses` (demo.py:538-586) produces for a centre depth of 2.5 and 7.0 (-> min(radius, 5)); 49 frames, plus one variant
    that also moves in x / y / r (d_x, d_y are unused by the orbit set but part of the function)."""
    if not os.path.isdir(REF):
        raise SystemExit("make_golden.py needs /root/reference (build container only)")
    _stub_utils_imports()
    sys.path.insert(0, REF)
    from models.utils import generate_traj_specified
    c2w_init = torch.tensor([[-1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0], [0.0, 0.0, -1.0, 0.0], [0.0, 0.0, 0.0, 1.0]]).unsqueeze(0)
    variants = [[0, -30, 1.0, 0, 0], [0, 30, 1.0, 0, 0], [30, 0, 1.0, 0, 0], [0, -45, 1.0, 0, 0], [0, 45, 1.0, 0, 0],
                [45, 0, 1.0, 0, 0], [0, -90, 1.0, 0, 0], [0, 90, 1.0, 0, 0], [10, -20, 0.5, 0.3, -0.2]]
    out = {}
    for radius in (1.0, 2.5, 5.0):
        ps = []
        for th, ph, dr, dx, dy in variants:
            ps.append(generate_traj_specified(c2w_init, th, ph, dr * radius, dx, dy, 49, "cpu"))
        out[f"poses_r{radius}"] = torch.stack(ps).float()
    out["variants"] = torch.tensor(variants, dtype=torch.float32)
    # `camera == 'traj'`: the reference's own generate_traj_txt on the key values of its two trajectory files (test/trajs/loop1.txt,
    # loop2.txt: theta keys / phi keys / r keys; read here as data) and on a short, linearly interpolated one; r keys times radius 2.0
    from models.utils import generate_traj_txt
    trajs = {"short": ([0, 10, 0], [0, -30, -60], [0, 0.3, 0.1])}
    for name in ("loop1", "loop2"):
        with open(os.path.join(REF, "test", "trajs", name + ".txt")) as f:
            lines = f.readlines()
        trajs[name] = tuple([float(v) for v in lines[i].split()] for i in range(3))
    for name, (theta, phi, r) in trajs.items():
        out[f"traj_{name}_keys_theta"], out[f"traj_{name}_keys_phi"], out[f"traj_{name}_keys_r"] = (torch.tensor(v, dtype=torch.float64) for v in (theta, phi, r))
        out[f"traj_{name}_poses"] = generate_traj_txt(c2w_init, phi, theta, [v * 2.0 for v in r], 49, "cpu").float()
    save("orbit_poses.safetensors", out,
         dict(source="reference models/utils.py generate_traj_specified(c2w_init, theta, phi, d_r * radius, d_x, d_y, 49, 'cpu')",
              frames=49))


TILED_VAE = dict(TINY_VAE, sample_height=96, sample_width=80)   # tiles 48 x 40 px = 6 x 5 latent rows / columns, strides 5 / 4, blends 8 / 8


def make_tiled():
    """vae_tiled_tiny.safetensors: the reference's `enable_tiling()` + `decode` (autoencoder_magvit.py:1109-1153, 1222-1225,
    1303-1392) on a 12 x 10 latent: 3 x 3 tiles with ragged last row / column, 2 temporal chunks per tile.  Weights are those of
    vae_tiny.safetensors (same seed, same shapes)."""
    if not os.path.isdir(REF):
        raise SystemExit("make_golden.py needs /root/reference (build container only)")
    install_scaffolding()
    sys.path.insert(0, REF)
    from models.autoencoder_magvit import AutoencoderKLCogVideoX
    vae_sd = bf16_round(iw.random_state_dict(iw.vae_param_shapes(TINY_VAE), seed=1))
    vae = AutoencoderKLCogVideoX(**TILED_VAE).eval()
    print("vae load:", vae.load_state_dict(vae_sd, strict=True))
    g = torch.Generator().manual_seed(777)
    z = torch.randn(1, 16, 5, 12, 10, generator=g)
    with torch.no_grad():
        vae.enable_tiling()
        dec = vae.decode(z).sample
        vae.enable_tiling(tile_sample_min_height=64, tile_sample_min_width=64, tile_overlap_factor_height=0.25,
                          tile_overlap_factor_width=0.25)          # 8 x 8 latent tiles, stride 6, blend 16, limit 48
        dec2 = vae.decode(z[:, :, :3]).sample                      # one temporal chunk
    print("tiled decode:", tuple(dec.shape), tuple(dec2.shape))
    save("vae_tiled_tiny.safetensors", dict(z=z, decoded_tiled=dec, decoded_tiled_64=dec2),
         dict(config=repr(TILED_VAE), weights="vae_tiny.safetensors",
              source="reference AutoencoderKLCogVideoX.enable_tiling() / .decode (tiled_decode, blend_v, blend_h)"))


def make_sincos():
    """transformer_sincos_tiny.safetensors: the reference's NON-rotary branch (crosstransformer3d.py:752-784, the 2B model's
    position embedding: the `pos_embedding` buffer resized trilinearly to the call's latent size and added to the joint
    sequence) on TINY_TR with use_rotary_positional_embeddings=False — once at the configured sample size and once at a smaller
    latent with fewer frames (interpolation + row cut).  Weights are those of transformer_tiny.safetensors.  The buffer itself
    comes from oracle/diffusers_restated.get_3d_sincos_pos_embed (diffusers is absent): the branch is pinned, the table is not."""
    if not os.path.isdir(REF):
        raise SystemExit("make_golden.py needs /root/reference (build container only)")
    install_scaffolding()
    sys.path.insert(0, REF)
    from models.crosstransformer3d import CrossTransformer3DModel
    cfg = dict(TINY_TR, use_rotary_positional_embeddings=False)
    tr_sd = bf16_round(iw.random_state_dict(iw.transformer_param_shapes(TINY_TR), seed=0))
    model = CrossTransformer3DModel(**cfg).eval()
    print("transformer load:", model.load_state_dict(tr_sd, strict=True))
    g = torch.Generator().manual_seed(4321)
    out = {"pos_embedding": model.pos_embedding.clone()}
    for tag, (T, h, w) in (("a", (3, 8, 12)), ("b", (2, 6, 10))):
        hs = torch.randn(2, T, 16, h, w, generator=g)
        enc = torch.randn(2, 10, 32, generator=g)
        inp = torch.randn(2, T, 17, h, w, generator=g)
        cross = torch.randn(2, 2, 16, h, w, generator=g)
        ts = torch.tensor([541, 541])
        with torch.no_grad():
            o = model(hs, enc, ts, inpaint_latents=inp, cross_latents=cross, image_rotary_emb=None, return_dict=False)[0]
        out.update({f"hidden_states_{tag}": hs, f"encoder_hidden_states_{tag}": enc, f"inpaint_latents_{tag}": inp,
                    f"cross_latents_{tag}": cross, f"timestep_{tag}": ts, f"out_sample_{tag}": o})
    save("transformer_sincos_tiny.safetensors", out,
         dict(config=repr(cfg), weights="transformer_tiny.safetensors",
              source="reference CrossTransformer3DModel.forward, use_rotary_positional_embeddings=False, image_rotary_emb=None"))


def make_default():
    """transformer_default.safetensors / vae_default.safetensors / pipeline_tiny_bf16.safetensors.

    The REFERENCE's own classes at the widths the product dispatches on — `AutoencoderKLCogVideoX()` (128/256/256/512: every conv
    with Cin % 64 == 0 goes to conv_mfma) and a 2-layer `CrossTransformer3DModel` at the 5B geometry (48 x 64 heads, fused-QKV
    layout, K = 3072 / 12288 GEMMs, cross-attention 16 x 128 -> attn_fwd_kernel<128>) — with the plain-torch diffusers stand-ins
    (diffusers_plain.py; no oracle code under the reference's classes), run twice: fp32, and `.to(bfloat16)` EAGER on the CPU =
    the reference's own bf16 execution, which `oracle.prec.Prec("bf16_ref")` claims to emulate.  Weights and the wide inputs are
    regenerated by the tests from the host-independent stream (default_cases.py); stored: digests + outputs (+ strided taps).
    Third file: the tiny 2-step CFG pipeline of pipeline_tiny.safetensors once more in eager bf16 (DESIGN §4's table), and the
    fp32 re-run with the plain stand-ins is asserted equal to the committed oracle-backed fixtures (two independent restatements
    of the diffusers arithmetic agree)."""
    if not os.path.isdir(REF):
        raise SystemExit("make_golden.py needs /root/reference (build container only)")
    install_scaffolding(plain=True)
    sys.path.insert(0, REF)
    sys.path.insert(0, HERE)
    import time
    import warnings
    warnings.filterwarnings("ignore")
    import default_cases as dc
    import diffusers_plain as dp
    from models.crosstransformer3d import CrossTransformer3DModel
    from models.autoencoder_magvit import AutoencoderKLCogVideoX
    from models import pipeline_trajectorycrafter as ref_pl
    from oracle.pipeline import prepare_rotary
    from safetensors.torch import load_file
    BF = torch.bfloat16

    # ---- (b) 2-layer transformer at the 5B geometry ---------------------------------------------
    t0 = time.time()
    sd = dc.transformer_weights()
    digest = iw.state_dict_digest(sd)
    model = CrossTransformer3DModel(**dc.DEFAULT_TR2).eval()
    print("transformer load:", model.load_state_dict(sd, strict=True), f"{sum(v.numel() for v in sd.values()) / 1e6:.0f} M params, {time.time() - t0:.0f} s")
    x = dc.transformer_inputs()
    B, T, C, h, w = dc.TR_LATENT
    cos, sin = prepare_rotary(h * 8, w * 8, T, 2, 64)

    def run_tr(m, dt):
        with torch.no_grad():
            a = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in x.items()}
            out = m(a["hidden_states"], a["encoder_hidden_states"], a["timestep"], inpaint_latents=a["inpaint_latents"],
                    cross_latents=a["cross_latents"], image_rotary_emb=(cos, sin), return_dict=False)[0]
            emb = m.time_embedding(m.time_proj(a["timestep"]).to(dt))
            pe = m.patch_embed(a["encoder_hidden_states"], torch.cat([a["hidden_states"], a["inpaint_latents"]], 2))
            ref_tok = m.ref_patch_embed(a["cross_latents"])
            blk_h, blk_e = m.transformer_blocks[0](pe[:, 226:], pe[:, :226], emb, (cos, sin))
            ca = m.perceiver_cross_attention[0](ref_tok, blk_h)
        return dict(out_sample=out, tap_block0_hidden=blk_h[:, ::4, ::16].contiguous(), tap_block0_encoder=blk_e[:, ::4, ::16].contiguous(),
                    tap_cross0=ca[:, ::3, ::8].contiguous(), tap_temb=emb)

    t0 = time.time()
    r32 = run_tr(model, torch.float32)
    print(f"  fp32 forward {time.time() - t0:.1f} s")
    t0 = time.time()
    r16 = run_tr(model.to(BF), BF)
    print(f"  bf16 eager forward {time.time() - t0:.1f} s;  mean|bf16 - fp32| {float((r16['out_sample'].float() - r32['out_sample']).abs().mean()):.3e}"
          f"  scale {float(r32['out_sample'].abs().mean()):.3e}")
    tens = {k: v.float() for k, v in r32.items()}
    tens.update({k + "_bf16_eager": v.to(BF) for k, v in r16.items()})
    tens.update(rope_cos=cos, rope_sin=sin)
    save("transformer_default.safetensors", tens,
         dict(config=repr(dc.DEFAULT_TR2), weights_digest=digest, weights=f"init_weights.hashed_state_dict(transformer_param_shapes(cfg), {dc.TR_SEED})",
              inputs="tests/golden/default_cases.transformer_inputs()", taps="block0 hidden / encoder [:, ::4, ::16]; cross0 [:, ::3, ::8]",
              source="reference CrossTransformer3DModel.forward (fp32, and .to(bfloat16) eager on the CPU) over tests/golden/diffusers_plain.py"))
    del sd
    model5b = model

    # ---- (a) default-width VAE ------------------------------------------------------------------------
    t0 = time.time()
    vsd = dc.vae_weights()
    vdigest = iw.state_dict_digest(vsd)
    vae = AutoencoderKLCogVideoX(**dc.DEFAULT_VAE).eval()
    print("vae load:", vae.load_state_dict(vsd, strict=True), f"{sum(v.numel() for v in vsd.values()) / 1e6:.0f} M params, {time.time() - t0:.0f} s")
    vi = dc.vae_inputs()

    def run_vae(m, dt):
        with torch.no_grad():
            dec = m.decode(vi["z"].to(dt)).sample
            dec1 = m.decode(vi["z"][:, :, :1].to(dt)).sample
            post = m.encode(vi["video"].to(dt))[0]
        return dict(decoded=dec, decoded_single_frame=dec1, enc_mean=post.mean.contiguous(), enc_logvar=post.logvar.contiguous())

    t0 = time.time()
    v32 = run_vae(vae, torch.float32)
    print(f"  fp32 decode + encode {time.time() - t0:.1f} s")
    t0 = time.time()
    v16 = run_vae(vae.to(BF), BF)
    print(f"  bf16 eager decode + encode {time.time() - t0:.1f} s;  mean|bf16 - fp32| decode {float((v16['decoded'].float() - v32['decoded']).abs().mean()):.3e}"
          f"  scale {float(v32['decoded'].abs().mean()):.3e}")
    tens = {k: v.float() for k, v in v32.items()}
    tens.update({k + "_bf16_eager": v.to(BF) for k, v in v16.items()})
    save("vae_default.safetensors", tens,
         dict(config=repr(dc.DEFAULT_VAE), weights_digest=vdigest, weights=f"init_weights.hashed_state_dict(vae_param_shapes(cfg), {dc.VAE_SEED})",
              inputs="tests/golden/default_cases.vae_inputs()",
              source="reference AutoencoderKLCogVideoX.decode / .encode (fp32, and .to(bfloat16) eager on the CPU) over tests/golden/diffusers_plain.py"))
    del vsd

    # ---- (b + a) the whole pipeline at those widths: 2 CFG / DDIM steps + decode, 9 frames 32x48 ----------------------
    pin = dc.pipeline_inputs()

    def run_pipe_default(dt):
        from oracle import diffusers_restated as dr_
        m, v = model5b.to(dt), vae.to(dt)
        pipe = ref_pl.TrajCrafter_Pipeline(tokenizer=None, text_encoder=None, vae=v, transformer=m,
                                           scheduler=dp.DDIMSchedulerPlain(dr_.DDIMScheduler()))
        kw = dict(prompt=None, negative_prompt=None, height=32, width=48, video=pin["video"], mask_video=pin["mask_video"],
                  reference=pin["reference"], num_frames=9, num_inference_steps=2, guidance_scale=6.0,
                  prompt_embeds=pin["prompt_embeds"].to(dt), negative_prompt_embeds=pin["negative_prompt_embeds"].to(dt))
        with torch.no_grad():
            torch.manual_seed(dc.PIPE_GLOBAL_SEED)
            frames = pipe(latents=pin["latents0"].clone().to(dt), **kw).videos
            torch.manual_seed(dc.PIPE_GLOBAL_SEED)
            lat = pipe(latents=pin["latents0"].clone().to(dt), output_type="latent", return_dict=True, **kw).videos
        return frames, lat

    t0 = time.time()
    pf32, pl32 = run_pipe_default(torch.float32)
    pf16, pl16 = run_pipe_default(BF)
    print(f"  default-width pipeline: fp32 + bf16 eager in {time.time() - t0:.1f} s; mean|frames bf16 - fp32| {float((pf16.float() - pf32.float()).abs().mean()):.3e}"
          f"  latents {float((pl16.float() - pl32.float()).abs().mean()):.3e} (scale {float(pl32.float().abs().mean()):.3e})")
    save("pipeline_default.safetensors",
         dict(frames=pf32.float(), latents_out=pl32.float(), frames_bf16_eager=pf16.float(), latents_out_bf16_eager=pl16.to(BF)),
         dict(tr_config=repr(dc.DEFAULT_TR2), vae_config=repr(dc.DEFAULT_VAE), steps=2, guidance_scale=6.0, global_seed=dc.PIPE_GLOBAL_SEED,
              weights_digest_transformer=digest, weights_digest_vae=vdigest, inputs="tests/golden/default_cases.pipeline_inputs()",
              source="reference TrajCrafter_Pipeline.__call__ (fp32, and every module .to(bfloat16) eager on the CPU) over tests/golden/diffusers_plain.py"))
    del model5b, vae, model

    # ---- (c') the tiny pipeline: plain stand-ins reproduce the committed fp32 fixtures, then eager bf16 -------------
    tt, tv, tp = (load_file(os.path.join(HERE, n)) for n in ("transformer_tiny.safetensors", "vae_tiny.safetensors", "pipeline_tiny.safetensors"))
    tr_sd = {k[2:]: v.float() for k, v in tt.items() if k.startswith("w.")}
    vae_sd = {k[2:]: v.float() for k, v in tv.items() if k.startswith("w.")}
    model = CrossTransformer3DModel(**TINY_TR).eval()
    model.load_state_dict(tr_sd, strict=True)
    vae = AutoencoderKLCogVideoX(**TINY_VAE).eval()
    vae.load_state_dict(vae_sd, strict=True)
    with torch.no_grad():
        out = model(tt["hidden_states"], tt["encoder_hidden_states"], tt["timestep"], inpaint_latents=tt["inpaint_latents"],
                    cross_latents=tt["cross_latents"], image_rotary_emb=(tt["rope_cos"], tt["rope_sin"]), return_dict=False)[0]
        dec = vae.decode(tv["z"]).sample
        post = vae.encode(tv["video"])[0]
    for name, a, b in (("transformer_tiny out_sample", out, tt["out_sample"]), ("vae_tiny decoded", dec, tv["decoded"]),
                       ("vae_tiny enc_mean", post.mean, tv["enc_mean"])):
        err = float((a - b).abs().max())
        print(f"  plain stand-ins vs oracle-backed stand-ins, {name}: max |diff| {err:.3e}")
        torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-6)

    def run_pipe(dt):
        from oracle import diffusers_restated as dr_
        m, v = model.to(dt), vae.to(dt)
        pipe = ref_pl.TrajCrafter_Pipeline(tokenizer=None, text_encoder=None, vae=v, transformer=m,
                                           scheduler=dp.DDIMSchedulerPlain(dr_.DDIMScheduler()))
        kw = dict(prompt=None, negative_prompt=None, height=32, width=48, video=tp["video"], mask_video=tp["mask_video"],
                  reference=tp["reference"], num_frames=9, num_inference_steps=2, guidance_scale=6.0,
                  prompt_embeds=tp["prompt_embeds"].to(dt), negative_prompt_embeds=tp["negative_prompt_embeds"].to(dt))
        with torch.no_grad():
            torch.manual_seed(77)
            frames = pipe(latents=tp["latents0"].clone().to(dt), **kw).videos
            torch.manual_seed(77)
            lat = pipe(latents=tp["latents0"].clone().to(dt), output_type="latent", return_dict=True, **kw).videos
        return frames, lat

    f32, l32 = run_pipe(torch.float32)
    print(f"  plain stand-ins vs committed pipeline fixture: frames max |diff| {float((f32 - tp['frames']).abs().max()):.3e}, "
          f"latents {float((l32 - tp['latents_out']).abs().max()):.3e}")
    torch.testing.assert_close(l32.float(), tp["latents_out"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(f32.float(), tp["frames"], rtol=1e-4, atol=1e-5)
    f16, l16 = run_pipe(BF)
    print(f"  eager bf16 pipeline: mean|frames - fp32| {float((f16.float() - tp['frames']).abs().mean()):.3e}  max {float((f16.float() - tp['frames']).abs().max()):.3e}")
    save("pipeline_tiny_bf16.safetensors", dict(frames_bf16_eager=f16.float(), latents_out_bf16_eager=l16.to(BF)),
         dict(inputs="pipeline_tiny.safetensors", weights="transformer_tiny.safetensors + vae_tiny.safetensors", global_seed=77,
              source="reference TrajCrafter_Pipeline.__call__ with every module .to(bfloat16), eager on the CPU, over tests/golden/diffusers_plain.py"))


SIG_MODULES = {
    "models.crosstransformer3d": "trajectorycrafter_amd.models.crosstransformer3d",
    "models.autoencoder_magvit": "trajectorycrafter_amd.models.autoencoder_magvit",
    "models.pipeline_trajectorycrafter": "trajectorycrafter_amd.models.pipeline_trajectorycrafter",
}


def describe_module(mod):
    """{name: signature description} of everything DEFINED in `mod`: module-level functions and, per class, `__init__` and every
    method / property the class body itself defines (inherited nn.Module / mixin members are not the module's surface).
    A parameter is [name, kind, default repr | None]; shared by make_signatures (reference side) and tests/test_signatures.py
    (mirror side), so both are described by the same code."""
    def params(fn):
        try:
            sig = inspect.signature(fn)
        except (TypeError, ValueError):
            return None
        return [[p.name, p.kind.name, None if p.default is inspect.Parameter.empty else repr(p.default)] for p in sig.parameters.values()]

    out = {"functions": {}, "classes": {}}
    for name, obj in vars(mod).items():
        if getattr(obj, "__module__", None) != mod.__name__ or name.startswith("_"):
            continue
        if inspect.isfunction(obj):
            out["functions"][name] = params(obj)
        elif inspect.isclass(obj):
            members = {}
            for mname, m in vars(obj).items():
                if mname.startswith("_") and mname not in ("__init__", "__call__", "_decode", "_set_gradient_checkpointing",
                                                           "_clear_fake_context_parallel_cache", "_get_t5_prompt_embeds", "_init_cross_inputs"):
                    continue
                if isinstance(m, (staticmethod, classmethod)):
                    members[mname] = {"kind": type(m).__name__, "params": params(m.__func__)}
                elif isinstance(m, property):
                    members[mname] = {"kind": "property"}
                elif inspect.isfunction(m) or (callable(m) and hasattr(m, "__wrapped__")):
                    members[mname] = {"kind": "method", "params": params(inspect.unwrap(m))}
            out["classes"][name] = {"bases": [b.__name__ for b in obj.__mro__[1:] if b.__module__ not in ("builtins",)][:4], "members": members}
    return out


def make_signatures():
    """signatures.json: the call surface of the reference's three hot-path modules as `inspect` sees it —
    every function / class / method name with parameter names, kinds and defaults — plus the state-dict key -> shape lists of the
    5B transformer and the default VAE built from the REFERENCE's constructors (meta device).  tests/test_signatures.py compares
    the mirrors in trajectorycrafter_amd/models against it mechanically."""
    if not os.path.isdir(REF):
        raise SystemExit("make_golden.py needs /root/reference (build container only)")
    install_scaffolding(plain=True)
    sys.path.insert(0, REF)
    import importlib
    import json
    import warnings
    warnings.filterwarnings("ignore")
    out = {"source": "inspect.signature over the reference's models/{crosstransformer3d,autoencoder_magvit,pipeline_trajectorycrafter}.py "
                     "(imported in the build container over tests/golden/diffusers_plain.py)", "modules": {}}
    for ref_name in SIG_MODULES:
        out["modules"][ref_name] = describe_module(importlib.import_module(ref_name))
    from models.crosstransformer3d import CrossTransformer3DModel
    from models.autoencoder_magvit import AutoencoderKLCogVideoX
    with torch.device("meta"):
        tr = CrossTransformer3DModel(**iw.TRANSFORMER_5B)
        vae = AutoencoderKLCogVideoX()
    out["state_dict"] = {"CrossTransformer3DModel(5B)": {k: list(v.shape) for k, v in tr.state_dict().items()},
                         "AutoencoderKLCogVideoX()": {k: list(v.shape) for k, v in vae.state_dict().items()}}
    out["config"] = {"CrossTransformer3DModel(5B)": {k: (list(v) if isinstance(v, tuple) else v) for k, v in dict(tr.config).items()},
                     "AutoencoderKLCogVideoX()": {k: (list(v) if isinstance(v, tuple) else v) for k, v in dict(vae.config).items()}}
    with open(os.path.join(HERE, "signatures.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    n = sum(len(c["members"]) for m in out["modules"].values() for c in m["classes"].values()) + sum(len(m["functions"]) for m in out["modules"].values())
    print(f"wrote signatures.json: {n} callables, {len(out['state_dict']['CrossTransformer3DModel(5B)'])} + "
          f"{len(out['state_dict']['AutoencoderKLCogVideoX()'])} state-dict keys, {os.path.getsize(os.path.join(HERE, 'signatures.json')) / 1e3:.0f} kB")


if __name__ == "__main__":
    if sys.argv[1:] == ["signatures"]:
        make_signatures()
    elif sys.argv[1:] == ["default"]:
        make_default()                                     # own process: the stand-in generation is fixed when the reference is imported
    elif sys.argv[1:] == ["tiled"]:
        make_tiled()
    elif sys.argv[1:] == ["sincos"]:
        make_sincos()
    elif sys.argv[1:] == ["warp"]:
        make_warp()
    elif sys.argv[1:] == ["poses"]:
        make_poses()
    else:
        main()
        make_warp()
        make_poses()
        make_tiled()
        make_sincos()
