"""Checkpoint formats (reference Loader.py:17-57, 111-170): upstream GPT key names in a `.ckpt`, the 2-byte
version tag that replaces b"PK" in a SoVITS `.pth`, dec weight-norm folding, safetensors directories.
CPU tests cover the parsing; the GPU test loads real files through TTS and must reproduce the synthetic://
route bit for bit."""
import json
import os

import numpy as np
import pytest
import torch

from gsv_tts_lite_amd import loader, synth


def _upstream_gpt_blob(cfg, w):
    inv = {new: old for old, new in loader._GPT_KEY_MAP}
    out = {}
    for k, v in w.items():
        t = torch.from_numpy(np.ascontiguousarray(v))
        if k.startswith("t2s_transformer.blocks."):
            i, rest = k[len("t2s_transformer.blocks."):].split(".", 1)
            out["model.h.layers.%s.%s" % (i, inv[rest])] = t
        else:
            out["model." + k] = t
    return {"config": cfg, "weight": out}


def test_gpt_ckpt_upstream_keys_are_remapped():
    cfg = synth.gpt_config(n_layer=2)
    w = synth.gpt_weights(cfg, seed=5)
    blob = _upstream_gpt_blob(cfg, w)
    assert "model.h.layers.1.self_attn.in_proj_weight" in blob["weight"] and "model.ar_predict_layer.weight" in blob["weight"]
    back = loader.remap_gpt_keys(blob["weight"], 2)
    assert set(back) == set(w)
    for k in w:
        assert np.array_equal(back[k].numpy(), w[k]), k


def test_sovits_pth_version_header_and_weight_norm_fold(tmp_path):
    hps = synth.sovits_hps("v2Pro")
    w = {k: torch.from_numpy(v) for k, v in synth.sovits_weights(hps, seed=3, hot_path_only=True).items()}
    # upstream checkpoints carry dec.* as weight_g / weight_v: split one conv that way
    k = "dec.ups.0.weight"
    v = w.pop(k)
    g = v.pow(2).sum(dim=(1, 2), keepdim=True).sqrt() * 1.7
    w[k + "_g"], w[k + "_v"] = g, v
    path = tmp_path / "s2.pth"
    torch.save({"config": hps, "weight": w}, str(path))
    raw = path.read_bytes()
    assert raw[:2] == b"PK"
    for head, ver in loader.HEAD2VERSION.items():
        path.write_bytes(head + raw[2:])           # the reference stores the version in place of b"PK"
        blob, version = loader.read_sovits_file(str(path))
        assert version == ver and set(blob["weight"]) == set(w)
    path.write_bytes(raw)
    blob, version = loader.read_sovits_file(str(path))
    assert version is None
    folded = loader.fold_dec_weight_norm(blob["weight"])
    assert k in folded and k + "_g" not in folded
    np.testing.assert_allclose(folded[k].numpy(), (v * 1.7).numpy(), rtol=1e-6)


@pytest.mark.gpu
def test_tts_loads_ckpt_pth_and_safetensors_dirs_like_synthetic(tmp_path):
    from safetensors.torch import save_file
    from gsv_tts import TTS
    dev = "cuda:0"
    cfg = synth.gpt_config(n_layer=3)
    gw = synth.gpt_weights(cfg, seed=9, eos_gain=1.0)
    torch.save(_upstream_gpt_blob(cfg, gw), str(tmp_path / "s1.ckpt"))
    gdir = tmp_path / "s1_st"; gdir.mkdir()
    save_file({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in gw.items()}, str(gdir / "model.safetensors"))
    (gdir / "config.json").write_text(json.dumps(cfg))
    hps = synth.sovits_hps("v2Pro")
    sw = {k: torch.from_numpy(v) for k, v in synth.sovits_weights(hps, seed=9).items()}
    p = tmp_path / "s2.pth"
    torch.save({"config": hps, "weight": sw}, str(p))
    p.write_bytes(b"05" + p.read_bytes()[2:])
    sdir = tmp_path / "s2_st"; sdir.mkdir()
    save_file({k: v.contiguous() for k, v in sw.items()}, str(sdir / "model.safetensors"))
    (sdir / "hps.json").write_text(json.dumps(hps))

    def run(gpt, sov):
        tts = TTS(gpt_cache=[(1, 128)], sovits_cache=[50], device=dev, dtype="float32")
        tts.load_gpt_model(gpt); tts.load_sovits_model(sov)
        tts.set_text_frontend(lambda text: ([1 + (ord(c) * 7) % 690 for c in text if not c.isspace()], {"word": list(text), "ph": [1] * len(text)}, None, text))
        tts.cache_spk_audio("spk.wav", ge=torch.from_numpy(synth.synth_ge(0, 1024)))
        x, y, _, _ = synth.synth_request(0, 12, 0, 30)
        tts.cache_prompt_audio("prompt.wav", "prompt text.", prompt=torch.from_numpy(y)[None], phones1=x.tolist())
        return tts.infer("spk.wav", "prompt.wav", "prompt text.", "Loading formats.", top_k=1, noise_scale=0.0).audio_data

    # TTS.to_safetensors (TTS.py:1482-1523): the converted directories load like the originals
    conv = TTS(gpt_cache=[(1, 128)], sovits_cache=[50], device=dev, dtype="float32")
    conv.to_safetensors(str(tmp_path / "s1.ckpt"))
    conv.to_safetensors(str(p), str(tmp_path / "s2_conv"))
    assert (tmp_path / "s1" / "config.json").exists() and (tmp_path / "s2_conv" / "hps.json").exists()
    with pytest.raises(ValueError):
        conv.to_safetensors(str(tmp_path / "model.bin"))

    ref = run("synthetic://gpt?seed=9&n_layer=3&eos_gain=1.0", "synthetic://sovits?version=v2Pro&seed=9")
    for gpt, sov in ((str(tmp_path / "s1.ckpt"), str(p)), (str(gdir), str(sdir)), (str(tmp_path / "s1"), str(tmp_path / "s2_conv"))):
        out = run(gpt, sov)
        assert out.shape == ref.shape
        np.testing.assert_allclose(out, ref, atol=1e-5)


def test_convert_to_safetensors_on_cpu(tmp_path):
    """the converter itself needs no device: keys remapped / weight norm folded, config files as the reference writes them"""
    from safetensors.torch import load_file
    cfg = synth.gpt_config(n_layer=2)
    gw = synth.gpt_weights(cfg, seed=4)
    torch.save(_upstream_gpt_blob(cfg, gw), str(tmp_path / "g.ckpt"))
    out = loader.convert_to_safetensors(str(tmp_path / "g.ckpt"))
    assert out == str(tmp_path / "g") and json.load(open(tmp_path / "g" / "config.json")) == cfg
    back = load_file(str(tmp_path / "g" / "model.safetensors"))
    assert set(back) == set(gw) and all(np.array_equal(back[k].numpy(), gw[k]) for k in gw)
    hps = synth.sovits_hps("v2")
    sw = {k: torch.from_numpy(v) for k, v in synth.sovits_weights(hps, seed=4, hot_path_only=True).items()}
    v = sw.pop("dec.conv_pre.weight")
    sw["dec.conv_pre.weight_v"] = v
    sw["dec.conv_pre.weight_g"] = v.pow(2).sum(dim=(1, 2), keepdim=True).sqrt() * 0.5
    p = tmp_path / "s.pth"
    torch.save({"config": hps, "weight": sw}, str(p))
    p.write_bytes(b"01" + p.read_bytes()[2:])          # "01": the v2 marker that replaces b"PK" (Loader.py:17-21)
    loader.convert_to_safetensors(str(p), str(tmp_path / "sdir"))
    h2 = json.load(open(tmp_path / "sdir" / "hps.json"))
    assert h2["model"]["version"] == "v2" and h2["model"]["semantic_frame_rate"] == "25hz"
    st = load_file(str(tmp_path / "sdir" / "model.safetensors"))
    assert "dec.conv_pre.weight" in st and "dec.conv_pre.weight_g" not in st and "flow.flows.0.enc.in_layers.0.weight_g" in st
    np.testing.assert_allclose(st["dec.conv_pre.weight"].numpy(), (v * 0.5).numpy(), rtol=1e-6)
