"""Minimal diffusers-style config / model plumbing (diffusers itself is not a dependency).

Keeps the reference's conventions: `@register_to_config` on `__init__` (reference
models/crosstransformer3d.py:459, models/autoencoder_magvit.py:991), `.config.<name>` attribute
access (models/pipeline_trajectorycrafter.py:221-236), `config.json` round trip and
`from_pretrained(dir, subfolder=...)` reading `*.safetensors` (demo.py:636-645).
"""
from __future__ import annotations

import functools
import inspect
import json
import os
from typing import Any, Dict, Sequence

import torch
from torch import nn


class FrozenConfig(dict):
    """dict with attribute access (diffusers FrozenDict behaviour used by the pipeline)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def register_to_config(init):
    sig = inspect.signature(init)

    @functools.wraps(init)
    def wrapped(self, *args, **kwargs):
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: (list(v) if isinstance(v, tuple) else v) for k, v in bound.arguments.items() if k != "self"}
        object.__setattr__(self, "_internal_config", FrozenConfig(cfg))
        init(self, *args, **kwargs)

    return wrapped


class ConfigMixin:
    config_name = "config.json"

    @property
    def config(self) -> FrozenConfig:
        return self._internal_config

    @classmethod
    def from_config(cls, config: Dict[str, Any], **overrides):
        cfg = dict(config)
        cfg.update(overrides)
        params = inspect.signature(cls.__init__).parameters
        return cls(**{k: v for k, v in cfg.items() if k in params})

    def save_config(self, save_directory: str) -> None:
        os.makedirs(save_directory, exist_ok=True)
        out = dict(self.config)
        out["_class_name"] = type(self).__name__
        with open(os.path.join(save_directory, self.config_name), "w") as f:
            json.dump(out, f, indent=2, sort_keys=True)

    @classmethod
    def load_config(cls, directory: str) -> Dict[str, Any]:
        path = os.path.join(directory, cls.config_name)
        if not os.path.isfile(path):
            raise RuntimeError(f"{path} does not exist")
        with open(path) as f:
            return json.load(f)


def load_state_dict_from_dir(directory: str) -> Dict[str, torch.Tensor]:
    """Read `diffusion_pytorch_model*.safetensors` (single file or sharded with an index) or a `.bin`
    (weights_only) from a local directory.  No network access, nothing executed from the files."""
    from safetensors.torch import load_file

    idx = os.path.join(directory, "diffusion_pytorch_model.safetensors.index.json")
    if os.path.isfile(idx):
        with open(idx) as f:
            files = sorted(set(json.load(f)["weight_map"].values()))
        sd: Dict[str, torch.Tensor] = {}
        for fn in files:
            sd.update(load_file(os.path.join(directory, fn)))
        return sd
    single = os.path.join(directory, "diffusion_pytorch_model.safetensors")
    if os.path.isfile(single):
        return load_file(single)
    shards = sorted(f for f in os.listdir(directory) if f.endswith(".safetensors"))
    if shards:
        sd = {}
        for fn in shards:
            sd.update(load_file(os.path.join(directory, fn)))
        return sd
    binf = os.path.join(directory, "diffusion_pytorch_model.bin")
    if os.path.isfile(binf):
        return torch.load(binf, map_location="cpu", weights_only=True)
    raise RuntimeError(f"no weights found in {directory}")


def load_checked(model: nn.Module, state_dict: Dict[str, torch.Tensor], what: str, allow_missing: Sequence[str] = (),
                 ignore_mismatched_sizes: bool = False) -> None:
    """`model.load_state_dict` that reports by NAME what the checkpoint does not cover: unexpected keys are printed and
    dropped (the reference's loaders do the same, crosstransformer3d.py:952-960); shape-mismatched keys are printed and
    either dropped (`ignore_mismatched_sizes`) or raise; missing keys are printed and raise unless their name starts with a
    prefix in `allow_missing` (those parameters keep their initialisation)."""
    own = model.state_dict()
    keep, mismatched, unexpected = {}, [], []
    for k, v in state_dict.items():
        if k not in own:
            unexpected.append(k)
        elif tuple(own[k].shape) != tuple(v.shape):
            mismatched.append(f"{k}: checkpoint {tuple(v.shape)} vs model {tuple(own[k].shape)}")
        else:
            keep[k] = v
    mismatched_names = {m.split(":")[0] for m in mismatched}
    missing = [k for k in own if k not in keep and k not in mismatched_names]
    allowed = tuple(allow_missing)
    fatal_missing = [k for k in missing if not (allowed and k.startswith(allowed))]
    for title, names in (("unexpected keys (ignored)", unexpected), ("shape-mismatched keys", mismatched),
                         ("missing keys (keep their initialisation)", [k for k in missing if k not in fatal_missing]),
                         ("MISSING keys", fatal_missing)):
        if names:
            print(f"[{what}] {len(names)} {title}: {names[:20]}{' ...' if len(names) > 20 else ''}")
    if mismatched and not ignore_mismatched_sizes:
        raise RuntimeError(f"{what}: {len(mismatched)} checkpoint tensors do not fit the model: {mismatched[:8]} "
                           "(pass ignore_mismatched_sizes=True to skip them)")
    if fatal_missing:
        raise RuntimeError(f"{what}: the checkpoint lacks {len(fatal_missing)} parameters of the model: {fatal_missing[:8]} "
                           "(pass allow_missing=(prefix, ...) for parameters that are meant to stay at their initialisation)")
    model.load_state_dict(keep, strict=False)


class ModelMixin(nn.Module):
    """nn.Module + the slice of diffusers.ModelMixin the reference's callers use."""

    @property
    def dtype(self) -> torch.dtype:
        for p in self.parameters():
            return p.dtype
        return torch.float32

    @property
    def device(self) -> torch.device:
        for p in self.parameters():
            return p.device
        return torch.device("cpu")

    @classmethod
    def from_pretrained(cls, pretrained_model_path: str, subfolder: str | None = None, torch_dtype=None,
                        allow_missing: Sequence[str] = (), ignore_mismatched_sizes: bool = False, **kwargs):
        """diffusers `ModelMixin.from_pretrained` for a local directory (demo.py:636-645): config.json + one / sharded
        `*.safetensors`.  A checkpoint that does not cover the model is an error, not a partly random model: a missing key
        raises unless its name starts with one of `allow_missing`, a shape mismatch raises unless
        `ignore_mismatched_sizes=True` (the diffusers kwarg); the offending key names are always printed."""
        path = os.path.join(pretrained_model_path, subfolder) if subfolder else pretrained_model_path
        config = cls.load_config(path)
        model = cls.from_config(config, **kwargs)
        load_checked(model, load_state_dict_from_dir(path), f"{cls.__name__}.from_pretrained({path})", allow_missing,
                     ignore_mismatched_sizes)
        if torch_dtype is not None:
            model = model.to(torch_dtype)
        return model.eval()

    def save_pretrained(self, save_directory: str, max_shard_size: int | None = None) -> None:
        """config.json + `diffusion_pytorch_model.safetensors`, or (state dict larger than `max_shard_size` bytes) the
        diffusers sharded layout: `diffusion_pytorch_model-0000i-of-0000n.safetensors` + `...safetensors.index.json`."""
        from safetensors.torch import save_file

        self.save_config(save_directory)
        sd = {k: v.contiguous().cpu() for k, v in self.state_dict().items()}
        total = sum(v.numel() * v.element_size() for v in sd.values())
        if max_shard_size is None or total <= max_shard_size:
            save_file(sd, os.path.join(save_directory, "diffusion_pytorch_model.safetensors"))
            return
        shards, cur, size = [], {}, 0
        for k, v in sd.items():
            nbytes = v.numel() * v.element_size()
            if cur and size + nbytes > max_shard_size:
                shards.append(cur)
                cur, size = {}, 0
            cur[k] = v
            size += nbytes
        shards.append(cur)
        weight_map = {}
        for i, shard in enumerate(shards):
            fn = f"diffusion_pytorch_model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
            save_file(shard, os.path.join(save_directory, fn))
            weight_map.update({k: fn for k in shard})
        with open(os.path.join(save_directory, "diffusion_pytorch_model.safetensors.index.json"), "w") as f:
            json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, f, indent=2)
