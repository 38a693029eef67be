"""CPU tests of the boundary: the C-ABI library builds/loads and exports every symbol that
include/gsv_tts_hip.h declares; the product path refuses to run without the GPU/extension."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "gsv_tts_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gsv_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    import __graft_entry__ as g
    g.build_hip()
    from gsv_tts_lite_amd import _native
    lib = ctypes.CDLL(_native.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libgsv_hip.so does not export %s" % n
    assert sorted(_native.EXPORTS) == names, "ctypes binding and header disagree"
    lib.gsv_version.restype = ctypes.c_int
    assert lib.gsv_version() >= 1


def test_no_cpu_fallback():
    """initialize_runtime on a non-GPU device must fail loudly, never route through the oracle."""
    import torch
    from gsv_tts_lite_amd import synth
    from gsv_tts_lite_amd.t2s import Text2SemanticDecoder
    from gsv_tts_lite_amd.sovits import SynthesizerTrn
    cfg = synth.gpt_config(n_layer=1)
    m = Text2SemanticDecoder(cfg)
    m.load_state_dict(synth.gpt_weights(cfg))
    with pytest.raises(RuntimeError):
        m.initialize_runtime(torch.float32, torch.device("cpu"), [(1, 64)])
    hps = synth.sovits_hps("v2Pro")
    s = SynthesizerTrn(1025, 32, **hps["model"])
    s.load_state_dict(synth.sovits_weights(hps, hot_path_only=True))
    with pytest.raises(RuntimeError):
        s.initialize_runtime(torch.float32, torch.device("cpu"), [])


def test_product_does_not_import_oracle():
    """only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "gsv-tts-lite_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "libgsv_oracle" not in txt, f
