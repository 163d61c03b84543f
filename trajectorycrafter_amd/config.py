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
from typing import Any, Dict

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
    def from_pretrained(cls, pretrained_model_path: str, subfolder: str | None = None, torch_dtype=None, **kwargs):
        path = os.path.join(pretrained_model_path, subfolder) if subfolder else pretrained_model_path
        config = cls.load_config(path)
        model = cls.from_config(config, **kwargs)
        sd = load_state_dict_from_dir(path)
        own = model.state_dict()
        keep, skipped = {}, []
        for k, v in sd.items():
            if k in own and own[k].shape == v.shape:
                keep[k] = v
            else:
                skipped.append(k)
        missing, unexpected = model.load_state_dict(keep, strict=False)
        if skipped or missing:
            print(f"[{cls.__name__}.from_pretrained] skipped {len(skipped)} mismatched keys, {len(missing)} missing")
        if torch_dtype is not None:
            model = model.to(torch_dtype)
        return model.eval()

    def save_pretrained(self, save_directory: str) -> None:
        from safetensors.torch import save_file

        self.save_config(save_directory)
        save_file({k: v.contiguous().cpu() for k, v in self.state_dict().items()},
                  os.path.join(save_directory, "diffusion_pytorch_model.safetensors"))
