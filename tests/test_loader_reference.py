"""`reference`-marked CPU tests (build container only: they import /root/reference): the SAME checkpoint files are
loaded by the reference's own Loader (gsv_tts/Loader.py:59-170, nn.Modules on the CPU) and by this package's loader,
and the resulting state must agree tensor for tensor.  This pins SURVEY.md 8(a) row a3 to the reference instead of
to the package's own key map: the GPT file below spells the UPSTREAM names out literally, and the SoVITS file is the
reference module's own state dict (weight-normed Generator and flow, as upstream checkpoints are)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from oracle import ref_harness  # noqa: E402

from gsv_tts_lite_amd import loader, synth  # noqa: E402

pytestmark = [pytest.mark.reference,
              pytest.mark.skipif(not ref_harness.reference_available(), reason="/root/reference is only in the build container")]

CPU_CFG = types.SimpleNamespace(device=torch.device("cpu"), dtype=torch.float32, use_flash_attn=False,
                                gpt_cache=[(1, 64)], sovits_cache=[])

# upstream (RVC-Boss/GPT-SoVITS AR model) parameter names of one decoder layer, in the order of the lite names below
UPSTREAM_LAYER = ["self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight", "self_attn.out_proj.bias",
                  "linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias",
                  "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias"]
LITE_LAYER = ["qkv.weight", "qkv.bias", "out_proj.weight", "out_proj.bias", "mlp.0.weight", "mlp.0.bias", "mlp.2.weight",
              "mlp.2.bias", "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias"]


def _ref_loader():
    ref_harness.import_reference()
    import gsv_tts.Loader as RL
    return RL


def test_gpt_ckpt_reference_loader_vs_product_loader(tmp_path):
    cfg = synth.gpt_config(n_layer=3)
    w = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.gpt_weights(cfg, seed=9).items()}
    up = {}
    for l in range(3):
        for u, s in zip(UPSTREAM_LAYER, LITE_LAYER):
            up["model.h.layers.%d.%s" % (l, u)] = w["t2s_transformer.blocks.%d.%s" % (l, s)]
    for k in w:
        if not k.startswith("t2s_transformer."):
            up["model." + k] = w[k]
    path = tmp_path / "s1.ckpt"
    torch.save({"config": cfg, "weight": up}, str(path))
    ref = _ref_loader().get_gpt_weights(str(path), CPU_CFG)
    ref_sd = ref.t2s_model.state_dict()
    config, mine = loader.read_gpt_checkpoint(str(path))
    assert config == ref.config
    assert set(mine) == set(ref_sd), set(mine) ^ set(ref_sd)
    for k, v in ref_sd.items():
        assert torch.equal(v, mine[k]), k
    # and the safetensors directory the reference's TTS.to_safetensors layout names
    out = loader.convert_to_safetensors(str(path), str(tmp_path / "s1_dir"))
    ref2 = _ref_loader().get_gpt_weights(out, CPU_CFG)
    for k, v in ref2.t2s_model.state_dict().items():
        assert torch.equal(v, mine[k]), k


@pytest.mark.parametrize("version,head", [("v2Pro", b"05"), ("v2", b"01"), ("v2ProPlus", b"PK")])
def test_sovits_pth_reference_loader_vs_product_loader(tmp_path, version, head):
    RL = _ref_loader()
    from gsv_tts.GPT_SoVITS.SoVITS.models import SynthesizerTrn
    hps = synth.sovits_hps(version)
    if head != b"PK":
        hps["model"].pop("version", None)      # the 2-byte tag must supply it
    torch.manual_seed(3)
    m = SynthesizerTrn(hps["data"]["filter_length"] // 2 + 1, hps["train"]["segment_size"] // hps["data"]["hop_length"],
                       n_speakers=hps["data"]["n_speakers"], **{**hps["model"], "version": version})
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    assert any(k.startswith("dec.") and k.endswith("weight_g") for k in sd), "the Generator must arrive weight-normed"
    for k in sd:                                # weight_g of a fresh module equals |v|: make the fold observable
        if k.endswith("weight_g"):
            sd[k] = sd[k] * 1.3
    path = tmp_path / "s2.pth"
    torch.save({"config": hps, "weight": sd}, str(path))
    raw = path.read_bytes()
    assert raw[:2] == b"PK"
    path.write_bytes(head + raw[2:])
    ref = RL.get_sovits_weights(str(path), CPU_CFG)
    ref_sd = ref.vq_model.state_dict()
    mine_hps, mine = loader.read_sovits_checkpoint(str(path))
    assert mine_hps["model"]["version"] == ref.hps.model.version == version
    hot = [k for k in ref_sd if k.startswith(("dec.", "flow.", "enc_p.", "ref_enc.", "quantizer.", "ssl_proj.", "sv_emb.", "prelu."))]
    assert len(hot) > 100
    for k in hot:
        assert k in mine, k
        np.testing.assert_allclose(mine[k].float().numpy(), ref_sd[k].float().numpy(), rtol=2e-6, atol=1e-7, err_msg=k)
    assert not any(k.startswith("dec.") and k.endswith(("weight_g", "weight_v")) for k in mine)


def test_pretrained_md5_table_is_the_references():
    RL = _ref_loader()
    assert loader.HASH_PRETRAINED == RL.hash_pretrained_dict
    assert loader.HEAD2VERSION == RL.head2version


def test_md5_version_lookup_on_a_pk_file(tmp_path, monkeypatch):
    """a b"PK" file whose first 8 KiB hash is in the table gets its version from the table, as Loader.py:47-49"""
    hps = synth.sovits_hps("v2Pro")
    hps["model"].pop("version", None)
    path = tmp_path / "s2G.pth"
    torch.save({"config": hps, "weight": {}}, str(path))
    h = loader.get_hash_from_file(str(path))
    RL = _ref_loader()
    assert h == RL.get_hash_from_file(str(path))
    monkeypatch.setitem(loader.HASH_PRETRAINED, h, "v2ProPlus")
    _, version = loader.read_sovits_file(str(path))
    assert version == "v2ProPlus"
