"""MI355X-native GPT-SoVITS inference hot path behind the reference's Python entry points.

Package layout:  csrc/ + lib/  HIP kernels and the C ABI (include/gsv_tts_hip.h);
t2s.py / sovits.py  host mirrors of the reference's Text2SemanticDecoder / SynthesizerTrn;
loader.py  checkpoint formats;  tts.py  the `TTS` facade;  scheduler.py  utterance sharding
over the GPUs of a node;  synth.py  deterministic synthetic weights/inputs.
"""
from . import synth  # noqa: F401

__all__ = ["synth", "TTS", "AudioClip", "Text2SemanticDecoder", "SynthesizerTrn"]


def __getattr__(name):  # lazy: importing the package must not require torch/GPU
    if name == "Text2SemanticDecoder":
        from .t2s import Text2SemanticDecoder
        return Text2SemanticDecoder
    if name == "SynthesizerTrn":
        from .sovits import SynthesizerTrn
        return SynthesizerTrn
    if name in ("TTS", "AudioClip", "cut_text"):
        from . import tts
        return getattr(tts, name)
    raise AttributeError(name)
