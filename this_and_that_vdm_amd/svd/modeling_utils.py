"""Minimal stand-ins for the diffusers plumbing the reference models inherit from
(ModelMixin / ConfigMixin / register_to_config / BaseOutput; imported at
svd/unet_spatio_temporal_condition.py:7-12, svd/temporal_controlnet.py:26-37): config capture,
``.dtype`` / ``.device``, and the diffusers on-disk folder format (config.json +
diffusion_pytorch_model.safetensors) used by test_code/inference.py:331-336,373-378."""
from __future__ import annotations

import functools
import inspect
import json
import os
from collections import OrderedDict
from dataclasses import fields, is_dataclass
from typing import Any, Dict

import torch
import torch.nn as nn

CONFIG_NAME = "config.json"
WEIGHTS_NAME = "diffusion_pytorch_model.safetensors"
WEIGHTS_NAME_BIN = "diffusion_pytorch_model.bin"


def _load_weights(path: str, variant: str = None) -> Dict[str, torch.Tensor]:
    """State dict of a diffusers model folder: a single ``diffusion_pytorch_model[.variant].safetensors`` / ``.bin``, or
    a sharded checkpoint described by ``diffusion_pytorch_model.safetensors.index[.variant].json`` (``weight_map``:
    parameter name -> shard file), as diffusers' save_pretrained(max_shard_size=...) writes it."""
    stem = "diffusion_pytorch_model" + (f".{variant}" if variant else "")
    st_file, bin_file = os.path.join(path, stem + ".safetensors"), os.path.join(path, stem + ".bin")
    index = os.path.join(path, "diffusion_pytorch_model.safetensors.index" + (f".{variant}" if variant else "") + ".json")
    if os.path.isfile(st_file):
        from safetensors.torch import load_file
        return load_file(st_file)
    if os.path.isfile(index):
        from safetensors.torch import load_file
        with open(index) as f:
            weight_map = json.load(f)["weight_map"]
        sd: Dict[str, torch.Tensor] = {}
        for shard in sorted(set(weight_map.values())):
            shard_file = os.path.join(path, shard)
            if not os.path.isfile(shard_file):
                raise OSError(f"{index} names shard {shard}, which is missing under {path}")
            part = load_file(shard_file)
            sd.update({k: v for k, v in part.items() if weight_map.get(k) == shard})
        absent = [k for k in weight_map if k not in sd]
        if absent:
            raise OSError(f"sharded checkpoint under {path}: {len(absent)} tensors named by the index are in no shard, e.g. {absent[:3]}")
        return sd
    if os.path.isfile(bin_file):
        return torch.load(bin_file, map_location="cpu")
    raise OSError(f"no weights file ({stem}.safetensors | .bin | safetensors index) under {path}")


class FrozenDict(OrderedDict):
    """dict with attribute access (``unet.config.in_channels``; pipeline...controlnet.py:236,493,588)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e


class BaseOutput(OrderedDict):
    """dataclass-style output that also indexes like a tuple (``out[0]``) and a dict (``out["sample"]``)."""

    def __post_init__(self):
        for f in fields(self):
            v = getattr(self, f.name)
            if v is not None:
                OrderedDict.__setitem__(self, f.name, v)

    def __getitem__(self, k):
        if isinstance(k, str):
            return OrderedDict.__getitem__(self, k)
        return self.to_tuple()[k]

    def to_tuple(self):
        return tuple(self[k] for k in self.keys())


def register_to_config(init):
    """Store the constructor's keyword arguments (with defaults) in ``self.config``."""
    sig = inspect.signature(init)

    @functools.wraps(init)
    def wrapper(self, *args, **kwargs):
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        skip = ("self",) + tuple(getattr(type(self), "_config_exclude", ()))      # run-time objects are not config entries
        cfg = {k: v for k, v in bound.arguments.items() if k not in skip}
        init(self, *args, **kwargs)
        cfg["_class_name"] = type(self).__name__
        self._internal_dict = FrozenDict(cfg)
    return wrapper


class ConfigMixin:
    config_name = CONFIG_NAME

    @property
    def config(self) -> FrozenDict:
        return self._internal_dict


class ModelMixin(nn.Module):
    """device/dtype introspection + diffusers-folder load/save."""

    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    @property
    def dtype(self) -> torch.dtype:
        for p in self.parameters():
            if p.is_floating_point():
                return p.dtype
        return torch.float32

    def save_pretrained(self, save_directory: str, safe_serialization: bool = True, **_):
        os.makedirs(save_directory, exist_ok=True)
        cfg = {k: (list(v) if isinstance(v, tuple) else v) for k, v in self.config.items()}
        with open(os.path.join(save_directory, CONFIG_NAME), "w") as f:
            json.dump(cfg, f, indent=2, sort_keys=True)
        sd = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(sd, os.path.join(save_directory, WEIGHTS_NAME), metadata={"format": "pt"})
        else:
            torch.save(sd, os.path.join(save_directory, WEIGHTS_NAME_BIN))

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, subfolder: str = None, torch_dtype=None,
                        low_cpu_mem_usage: bool = True, variant: str = None, **kwargs):
        """Local diffusers-format folder only (this image has no network; the reference pulls from the HF hub)."""
        path = pretrained_model_name_or_path
        if subfolder:
            path = os.path.join(path, subfolder)
        cfg_file = os.path.join(path, CONFIG_NAME)
        if not os.path.isfile(cfg_file):
            raise OSError(f"{cfg_file} not found: from_pretrained needs a local diffusers-format folder")
        with open(cfg_file) as f:
            cfg = json.load(f)
        accepted = set(inspect.signature(cls.__init__).parameters) - {"self"}
        init_kwargs = {k: (tuple(v) if isinstance(v, list) else v) for k, v in cfg.items() if k in accepted}
        init_kwargs.update({k: v for k, v in kwargs.items() if k in accepted})
        model = cls(**init_kwargs)
        sd = _load_weights(path, variant)
        missing, unexpected = model.load_state_dict(sd, strict=False)
        if missing or unexpected:
            raise RuntimeError(f"state dict mismatch loading {cls.__name__}: missing={missing[:5]} unexpected={unexpected[:5]}")
        if torch_dtype is not None:
            model = model.to(torch_dtype)
        return model.eval()
