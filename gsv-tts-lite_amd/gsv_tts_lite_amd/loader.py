"""Checkpoint loading for the MI355X hot path: same entry points and on-disk formats as the
reference's gsv_tts/Loader.py (get_gpt_weights :111-170, get_sovits_weights :59-103), producing
this package's runtime objects instead of nn.Modules.

Accepted sources
  * GPT    `.ckpt`  torch pickle {"config": {...}, "weight": {...}} with upstream key names
                    (`model.h.layers.{i}.self_attn.in_proj_weight` ...), remapped to the lite names;
           directory {config.json, model.safetensors}  (TTS.to_safetensors layout)
  * SoVITS `.pth`   torch pickle {"config": ..., "weight": ...}; the first two bytes may be a
                    version tag (b"01" v2, b"05" v2Pro, b"06" v2ProPlus) instead of b"PK";
           directory {hps.json, model.safetensors}
  * `synthetic://gpt?seed=1234&n_layer=24&eos_gain=1.0` and
    `synthetic://sovits?version=v2Pro&seed=1234` -- seeded weights of the real architecture
    (gsv_tts_lite_amd.synth); there are no checkpoints in the build/bench environment.
Weight-norm: `dec.*` arrives with weight_g/weight_v in upstream checkpoints and is folded here
(the reference calls dec.remove_weight_norm() after load); `flow.*` keeps g/v and is folded by the
native library at finalize.
"""
from __future__ import annotations

import io
import json
import os
from urllib.parse import parse_qs, urlparse

import numpy as np
import torch

from . import synth
from .sovits import SynthesizerTrn
from .t2s import Text2SemanticDecoder

HEAD2VERSION = {b"01": "v2", b"05": "v2Pro", b"06": "v2ProPlus"}
# md5 of the first 8 KiB of the official pretrained s2G*.pth files (they start with b"PK", so the two-byte tag says
# nothing): reference Loader.py:22-40
HASH_PRETRAINED = {
    "dc3c97e17592963677a4a1681f30c653": "v2",         # s2G488k.pth
    "6642b37f3dbb1f76882b69937c95a5f3": "v2",         # s2G2333K.pth
    "c7e9fce2223f3db685cdfa1e6368728a": "v2Pro",      # s2Gv2Pro.pth
    "66b313e39455b57ab1b0bc0b239c9d0a": "v2ProPlus",  # s2Gv2ProPlus.pth
}


def get_hash_from_file(path) -> str:
    """Loader.py:35-40"""
    import hashlib
    with open(path, "rb") as f:
        return hashlib.md5(f.read(8192)).hexdigest()

_GPT_KEY_MAP = [
    ("self_attn.in_proj_weight", "qkv.weight"), ("self_attn.in_proj_bias", "qkv.bias"),
    ("self_attn.out_proj.weight", "out_proj.weight"), ("self_attn.out_proj.bias", "out_proj.bias"),
    ("linear1.weight", "mlp.0.weight"), ("linear1.bias", "mlp.0.bias"),
    ("linear2.weight", "mlp.2.weight"), ("linear2.bias", "mlp.2.bias"),
    ("norm1.weight", "norm1.weight"), ("norm1.bias", "norm1.bias"),
    ("norm2.weight", "norm2.weight"), ("norm2.bias", "norm2.bias"),
]


class Gpt:
    def __init__(self, t2s_model, config):
        self.t2s_model = t2s_model
        self.config = config


class Sovits:
    def __init__(self, vq_model, hps):
        self.vq_model = vq_model
        self.hps = hps


class AttrDict(dict):
    """dict with attribute access, recursively (the reference's DictToAttrRecursive role)."""

    def __init__(self, d):
        super().__init__()
        for k, v in d.items():
            self[k] = AttrDict(v) if isinstance(v, dict) else v

    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def remap_gpt_keys(weights: dict, n_layer: int) -> dict:
    """upstream -> lite names (reference Loader.py:130-154)"""
    w = dict(weights)
    for i in range(n_layer):
        for old, new in _GPT_KEY_MAP:
            k = "model.h.layers.%d.%s" % (i, old)
            if k in w:
                w["t2s_transformer.blocks.%d.%s" % (i, new)] = w.pop(k)
    return {(k[len("model."):] if k.startswith("model.") else k): v for k, v in w.items()}


def _synthetic(spec: str):
    u = urlparse(spec)
    q = {k: v[0] for k, v in parse_qs(u.query).items()}
    return u.netloc, q


def read_gpt_checkpoint(gpt_path):
    """(config, state dict under the lite names): what the reference's module holds after Loader.get_gpt_weights
    (:111-170) has loaded `gpt_path`; device-free, so that it can be compared with the reference on a CPU"""
    gpt_path = str(gpt_path)
    if gpt_path.startswith("synthetic://"):
        _, q = _synthetic(gpt_path)
        config = synth.gpt_config(n_layer=int(q.get("n_layer", 24)))
        weights = synth.gpt_weights(config, seed=int(q.get("seed", 1234)), eos_gain=float(q.get("eos_gain", 1.0)))
    elif os.path.isdir(gpt_path):
        from safetensors.torch import load_file
        with open(os.path.join(gpt_path, "config.json")) as f:
            config = json.load(f)
        weights = load_file(os.path.join(gpt_path, "model.safetensors"))
    else:
        blob = torch.load(gpt_path, map_location="cpu", weights_only=False)
        config = blob["config"]
        weights = remap_gpt_keys(blob["weight"], config["model"]["n_layer"])
    return config, weights


def get_gpt_weights(gpt_path, tts_config) -> Gpt:
    config, weights = read_gpt_checkpoint(gpt_path)
    model = Text2SemanticDecoder(config)
    model.load_state_dict(weights)
    model.eval()
    model.initialize_runtime(tts_config.dtype, tts_config.device, tts_config.gpt_cache)
    return Gpt(model, config)


def read_sovits_file(path):
    """handles the 2-byte version header that replaces b"PK", and the md5 table of the official pretrained files
    (reference Loader.py:42-57)"""
    with open(path, "rb") as f:
        head = f.read(2)
        rest = f.read()
    version = HEAD2VERSION.get(head)
    if version is None:
        version = HASH_PRETRAINED.get(get_hash_from_file(path))
    data = (b"PK" + rest) if head != b"PK" else (head + rest)
    return torch.load(io.BytesIO(data), map_location="cpu", weights_only=False), version


def fold_dec_weight_norm(weights: dict) -> dict:
    """dec.*.weight_g / weight_v -> dec.*.weight  (what dec.remove_weight_norm() leaves behind)"""
    out = dict(weights)
    for k in [k for k in weights if k.startswith("dec.") and k.endswith(".weight_g")]:
        base = k[: -len("_g")]
        g, v = out.pop(k).float(), out.pop(base + "_v").float()
        norm = v.pow(2).sum(dim=tuple(range(1, v.dim())), keepdim=True).sqrt()
        out[base] = v * (g / norm)
    return out


def _build_sovits(hps: dict, weights: dict, tts_config) -> Sovits:
    hp = AttrDict(hps)
    m = dict(hps["model"])
    m.setdefault("semantic_frame_rate", "25hz")
    vq = SynthesizerTrn(hp.data.filter_length // 2 + 1, hp.train.segment_size // hp.data.hop_length,
                        n_speakers=hp.data.n_speakers, **m)
    vq.load_state_dict(fold_dec_weight_norm(weights), strict=False)
    vq.eval()
    vq.initialize_runtime(tts_config.dtype, tts_config.device, tts_config.sovits_cache)
    return Sovits(vq, hp)


def read_sovits_checkpoint(sovits_path):
    """(hps dict with the version resolved, state dict with the Generator's weight norm folded): what the reference's
    module holds after Loader.get_sovits_weights (:59-103); device-free"""
    sovits_path = str(sovits_path)
    if sovits_path.startswith("synthetic://"):
        _, q = _synthetic(sovits_path)
        hps = synth.sovits_hps(q.get("version", "v2Pro"))
        seed = int(q.get("seed", 1234))
        weights = {k: torch.from_numpy(v) for k, v in synth.sovits_weights(hps, seed=seed).items()}
        weights.update({k: torch.from_numpy(v) for k, v in synth.ref_audio_weights(hps, seed=seed).items()})
        return hps, weights
    if os.path.isdir(sovits_path):
        from safetensors.torch import load_file
        with open(os.path.join(sovits_path, "hps.json")) as f:
            hps = json.load(f)
        return hps, load_file(os.path.join(sovits_path, "model.safetensors"))
    blob, version = read_sovits_file(sovits_path)
    hps = json.loads(json.dumps(blob["config"], default=lambda o: dict(o)))
    hps["model"]["semantic_frame_rate"] = "25hz"
    if version is None:
        version = hps["model"].get("version")
        if version not in ("v2", "v2Pro", "v2ProPlus"):
            raise ValueError("The SoVITS checkpoint is not a v2 / v2Pro / v2ProPlus model")
    hps["model"]["version"] = version
    return hps, fold_dec_weight_norm(blob["weight"])


def get_sovits_weights(sovits_path, tts_config) -> Sovits:
    hps, weights = read_sovits_checkpoint(sovits_path)
    return _build_sovits(hps, weights, tts_config)


def convert_to_safetensors(checkpoint_path, output_dir=None) -> str:
    """TTS.to_safetensors (gsv_tts/TTS.py:1482-1523): a `.pth` SoVITS checkpoint becomes {hps.json, model.safetensors},
    a `.ckpt` GPT checkpoint {config.json, model.safetensors} -- the directory form both loaders read.  The tensors
    are what the reference's module holds after loading: GPT keys remapped (Loader.py:130-154), the Generator's
    weight norm folded (`dec.remove_weight_norm()`, Loader.py:95), the flow's weight_g / weight_v kept."""
    from safetensors.torch import save_file
    checkpoint_path = str(checkpoint_path)
    root, ext = os.path.splitext(checkpoint_path)
    if output_dir is None:
        output_dir = root
    os.makedirs(output_dir, exist_ok=True)
    as_tensor = lambda v: (torch.from_numpy(v) if not torch.is_tensor(v) else v).detach().cpu().contiguous()
    if ext == ".pth":
        blob, version = read_sovits_file(checkpoint_path)
        hps = json.loads(json.dumps(blob["config"], default=lambda o: dict(o)))
        hps["model"]["semantic_frame_rate"] = "25hz"
        if version is not None:
            hps["model"]["version"] = version
        weights = {k: as_tensor(v) for k, v in fold_dec_weight_norm(blob["weight"]).items() if torch.is_tensor(v) or hasattr(v, "shape")}
        save_file(weights, os.path.join(output_dir, "model.safetensors"))
        with open(os.path.join(output_dir, "hps.json"), "w") as f:
            json.dump(hps, f, indent=4, ensure_ascii=False)
    elif ext == ".ckpt":
        blob = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
        config = blob["config"]
        weights = {k: as_tensor(v) for k, v in remap_gpt_keys(blob["weight"], config["model"]["n_layer"]).items()}
        save_file(weights, os.path.join(output_dir, "model.safetensors"))
        with open(os.path.join(output_dir, "config.json"), "w") as f:
            json.dump(config, f, indent=4, ensure_ascii=False)
    else:
        raise ValueError("to_safetensors converts .pth (SoVITS) and .ckpt (GPT) checkpoints, got %r" % ext)
    return output_dir
