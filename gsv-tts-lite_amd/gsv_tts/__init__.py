"""Drop-in alias: `from gsv_tts import TTS, AudioClip, cut_text` resolves to the MI355X package
(the reference exports exactly these three names, gsv_tts/__init__.py:1-11)."""
from gsv_tts_lite_amd.tts import TTS, AudioClip, cut_text  # noqa: F401

__all__ = ["TTS", "AudioClip", "cut_text"]
