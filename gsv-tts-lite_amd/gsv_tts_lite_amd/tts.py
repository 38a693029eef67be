"""`TTS`: the reference's engine facade (gsv_tts/TTS.py) over the MI355X hot path.

Drop-in surface kept from the reference:
    TTS(gpt_cache, sovits_cache, models_dir, device, dtype, use_flash_attn, use_bert, auto_bert,
        use_jieba_fast, always_load_cnhubert, always_load_sv)                 TTS.py:39-52
    infer(...) -> AudioClip                                                    TTS.py:150-286
    infer_batched(...) -> tuple[AudioClip]                                     TTS.py:507-868
    infer_stream(...) -> generator of AudioClip, infer_vc(...), *_async wrappers  TTS.py:289-504, 871-1262
    to_safetensors, cache_* / del_* / get_*_list                                TTS.py:1346-1523
    load_gpt_model / load_sovits_model / unload_* / get_*_list                 TTS.py:1264-1345
    AudioClip(audio_data, samplerate, audio_len_s, subtitles, orig_text)       Player.py:68-99

What sits in front of the hot path in the reference -- G2P text frontends, audio file decoding / resampling,
the CN-HuBERT and ERes2Net models -- is OUT OF SCOPE of this build (CPU string
processing and third-party models whose packages are not installable here).  Their *outputs* enter through the
same caches the reference keeps:
    cache_spk_audio(path, ge=...)  or  cache_spk_audio(path, audio=<waveform>, sv_emb=<ERes2Net embedding>)
                                   (spectrogram + get_ge on the device; the reference: TTS.py:1346, 1576)
    cache_prompt_audio(path, text, prompt=... | ssl_content=<CN-HuBERT features>, phones1=..., bert1=...)
                                   (extract_latent on the device; TTS.py:1391, 1556)
    set_text_frontend(fn)   fn(text) -> (phones2, word2ph, bert2[P,1024], norm_text)
With those in place infer()/infer_batched()/infer_stream() behave as in the reference, including
`return_subtitles=True`: the frame->phoneme alignment runs on the device (subtitles.viterbi_monotonic ->
gsv_align_viterbi, replacing TTS.py:1744-1797) and the word-timing / text-span bookkeeping is subtitles.py.
"""
from __future__ import annotations

import logging
import os
import re
import threading
from pathlib import Path

import numpy as np
import torch

from . import subtitles as sub
from .batchmath import balance_order, split_bounds
from .stream import ChunkSplicer
from .loader import Gpt, Sovits, convert_to_safetensors, get_gpt_weights, get_sovits_weights

log = logging.getLogger("gsv_tts_lite_amd")

PAUSE_MARKS = tuple("…。？！.?!,，:：;；~、・—")


class Config:
    def __init__(self):
        if torch.cuda.is_available():
            self.device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
            self.dtype = torch.bfloat16
        else:
            self.device = torch.device("cpu")
            self.dtype = torch.float32
        self.use_flash_attn = False
        self.gpt_cache = []
        self.sovits_cache = []


class AudioClip:
    """Player.py:68-99 fields; playback needs `sounddevice`, saving uses `soundfile` when present
    and falls back to a 16-bit PCM WAV writer."""

    def __init__(self, audio_queue, audio_data, samplerate, audio_len_s, subtitles, orig_text):
        self.audio_queue = audio_queue
        self.audio_data = audio_data
        self.samplerate = samplerate
        self.audio_len_s = audio_len_s
        self.subtitles = subtitles
        self.orig_text = orig_text

    def play(self, volume: float = 1.0):
        if self.audio_queue is None:
            raise RuntimeError("audio playback needs the optional `sounddevice` package")
        data = self.audio_data if volume == 1.0 else np.clip(self.audio_data * volume, -1.0, 1.0)
        self.audio_queue.put(data)

    def save(self, save_path: str, is_save_subtitles: bool = False):
        try:
            import soundfile as sf
            sf.write(save_path, self.audio_data, self.samplerate)
        except ImportError:
            import wave
            pcm = (np.clip(self.audio_data, -1.0, 1.0) * 32767.0).astype("<i2")
            with wave.open(save_path, "wb") as w:
                w.setnchannels(1); w.setsampwidth(2); w.setframerate(self.samplerate)
                w.writeframes(pcm.tobytes())
        if is_save_subtitles:
            import json
            with open(os.path.splitext(save_path)[0] + ".json", "w", encoding="utf-8") as f:
                json.dump(self.subtitles, f, ensure_ascii=False, indent=2)


def cut_text(text: str, minlen: int = 10) -> list:
    """Split text into sentence-like segments at pause punctuation, merging segments shorter than
    `minlen` into their neighbour (role of TextProcessor.cut_text, TextProcessor.py:18-59; the
    reference delegates sentence boundaries to pysbd, unavailable here)."""
    parts = [p for p in re.split(r"(?<=[。！？!?\.…;；\n])", text) if p and p.strip()]
    out, cur = [], ""
    for p in parts:
        cur += p
        if len(cur) >= minlen:
            out.append(cur)
            cur = ""
    if cur:
        if out:
            out[-1] += cur
        else:
            out.append(cur)
    return [s.strip() for s in out if s.strip()]


class _EngineLock:
    """One inference at a time per TTS (the reference's `_infer_lock`, TTS.py:145).  A streaming generator owns the
    engine -- KV cache, enc_p overlap state -- from its first chunk to its last, so the lock stays held across its
    yields; what must not happen is the SAME thread calling back in between two chunks and waiting for itself forever:
    that raises instead.  An abandoned generator releases the lock when it is closed or collected."""

    def __init__(self):
        self._lock = threading.Lock()
        self._owner = None

    def __enter__(self):
        me = threading.get_ident()
        if self._owner == me:
            raise RuntimeError("this TTS is in the middle of an infer_stream() on this thread: exhaust or close() the "
                               "generator before starting another inference")
        self._lock.acquire()
        self._owner = me
        return self

    def __exit__(self, *exc):
        self._owner = None
        self._lock.release()
        return False


class TTS:
    def __init__(self, gpt_cache=[(1, 512), (1, 768), (1, 1024), (4, 512), (4, 1024)], sovits_cache=[50, 55],
                 models_dir: str = None, device: str = None, dtype: str = None, use_flash_attn: bool = False,
                 use_bert: bool = False, auto_bert: bool = True, use_jieba_fast: bool = False,
                 always_load_cnhubert: bool = False, always_load_sv: bool = False):
        self.tts_config = Config()
        if device is not None:
            self.tts_config.device = torch.device(device)
        if dtype is not None:
            self.tts_config.dtype = {"float32": torch.float32, "bfloat16": torch.bfloat16}.get(dtype.lower())
            if self.tts_config.dtype is None:
                raise ValueError("dtype must be float32 or bfloat16 on the MI355X path (float16 is not implemented)")
        if use_flash_attn:
            log.warning("use_flash_attn is ignored: the HIP decode kernel already reads only kv_len entries")
        self.tts_config.gpt_cache = list(gpt_cache)
        self.tts_config.sovits_cache = list(sovits_cache)
        self.models_dir = models_dir if models_dir is not None else Path.home() / ".cache" / "gsv"
        self.default_gpt_path = Path(self.models_dir) / "s1v3.ckpt"
        self.default_sovits_path = Path(self.models_dir) / "s2Gv2ProPlus.pth"
        self.gpt_models: dict = {}
        self.sovits_models: dict = {}
        self.spk_audio_cache: dict = {}
        self.prompt_audio_cache: dict = {}
        self.samplerate, self.gpt_hz, self.sovits_hz = 32000, 25, 50
        self.audio_queue = None
        self._infer_lock = _EngineLock()
        self._text_frontend = None
        # multi-GPU (one process per GPU): the rank on which infer_batched returns the clips; the other ranks return None.
        # None = every rank gets every clip (an all-gather of the audio instead of point-to-point sends to one rank)
        self.gather_dst = 0

    # ------------------------------------------------------------------ model management
    def load_gpt_model(self, *model_paths):
        for p in (model_paths or (self.default_gpt_path,)):
            self.gpt_models[p] = get_gpt_weights(p, self.tts_config)
            log.info("Loaded GPT model: %s", p)

    def load_sovits_model(self, *model_paths):
        for p in (model_paths or (self.default_sovits_path,)):
            self.sovits_models[p] = get_sovits_weights(p, self.tts_config)
            log.info("Loaded SoVITS model: %s", p)

    def unload_gpt_model(self, *model_paths):
        try:
            for p in model_paths:
                if self.gpt_models.pop(p, None) is None:
                    log.warning("GPT model %s not found.", p)
        finally:
            self._empty_cache()

    def unload_sovits_model(self, *model_paths):
        try:
            for p in model_paths:
                if self.sovits_models.pop(p, None) is None:
                    log.warning("SoVITS model %s not found.", p)
                for a in self.spk_audio_cache.values():
                    a["ge"].pop(p, None)
        finally:
            self._empty_cache()

    def to_safetensors(self, checkpoint_path: str, output_dir: str = None):
        """TTS.py:1482-1523: convert a .pth / .ckpt checkpoint to the safetensors directory form."""
        try:
            out = convert_to_safetensors(checkpoint_path, output_dir)
            log.info("Successfully converted and saved to: %s", out)
        finally:
            self._empty_cache()

    def get_gpt_list(self):
        return list(self.gpt_models.keys())

    def get_sovits_list(self):
        return list(self.sovits_models.keys())

    def _empty_cache(self):
        if torch.cuda.is_available():
            torch.cuda.empty_cache()

    # ------------------------------------------------------------------ front-of-hot-path inputs
    def set_text_frontend(self, fn):
        """fn(text) -> (phones2 list[int], word2ph dict, bert2 [P,1024] tensor, norm_text)"""
        self._text_frontend = fn

    def cache_spk_audio(self, spk_audio_paths, sovits_model=None, ge=None, audio=None, sv_emb=None):
        """TTS.py:1346-1389.  Either the finished embedding `ge` [1, gin, 1], or the reference waveform `audio`
        (mono fp32 at the model rate, what TTS._load_audio + _resample give) plus, for v2Pro / v2ProPlus, the ERes2Net
        embedding `sv_emb` [1, 20480]: then the spectrogram (TTS._get_spec) and get_ge run on the device."""
        sovits_model = self._pick(self.sovits_models, sovits_model, self.default_sovits_path)
        if ge is None:
            if audio is None:
                raise NotImplementedError("decoding / resampling audio files and the ERes2Net model are outside this build's "
                                          "scope; pass ge=[1, gin, 1], or audio=<waveform> (+ sv_emb=[1, 20480])")
            if sovits_model not in self.sovits_models:
                self.load_sovits_model(sovits_model)
            vq = self.sovits_models[sovits_model].vq_model
            audio = audio.to(self.tts_config.device).float().reshape(1, -1)
            peak = audio.abs().max()
            if peak > 1:                       # TTS.py:1586-1588
                audio = audio / min(2, float(peak))
            ge = vq.get_ge(vq.spectrogram(audio), sv_emb)
        entry = self.spk_audio_cache.setdefault(spk_audio_paths, {"ge": {}})
        entry["ge"][sovits_model] = ge.to(self.tts_config.device)

    def cache_prompt_audio(self, prompt_audio_paths, prompt_audio_texts, prompt=None, phones1=None, bert1=None,
                           ssl_content=None, sovits_model=None):
        """TTS.py:1391-1440.  `prompt` int64 [1, Ly], or `ssl_content` [1, 768, Th] (CN-HuBERT last_hidden_state,
        transposed as in TTS._get_prompt): then extract_latent runs on the device."""
        if not prompt_audio_texts:
            raise ValueError("prompt_audio_text must not be empty")
        if prompt is None and ssl_content is not None:
            sovits_model = self._pick(self.sovits_models, sovits_model, self.default_sovits_path)
            if sovits_model not in self.sovits_models:
                self.load_sovits_model(sovits_model)
            prompt = self.sovits_models[sovits_model].vq_model.extract_latent(ssl_content)[0, 0].unsqueeze(0)
        if prompt is None or phones1 is None:
            raise NotImplementedError("CN-HuBERT and G2P are outside this build's scope; pass prompt=int64[1,Ly] or "
                                      "ssl_content=[1,768,Th], and phones1=list[int] (and bert1=[Lx1,1024])")
        if bert1 is None:
            bert1 = torch.zeros(len(phones1), 1024)
        self.prompt_audio_cache[prompt_audio_paths] = {
            "prompt": prompt.to(self.tts_config.device), "phones1": list(phones1),
            "bert1": bert1.to(self.tts_config.device), "text": prompt_audio_texts}

    def del_spk_audio(self, *spk_audio_list):
        """TTS.py:1436-1448"""
        for p in spk_audio_list:
            if self.spk_audio_cache.pop(p, None) is None:
                log.warning("Speaker audio %s not found in cache.", p)

    def del_prompt_audio(self, *prompt_audio_list):
        """TTS.py:1450-1462"""
        for p in prompt_audio_list:
            if self.prompt_audio_cache.pop(p, None) is None:
                log.warning("Prompt audio %s not found in cache.", p)

    def get_spk_audio_list(self):
        return list(self.spk_audio_cache.keys())

    def get_prompt_audio_list(self):
        return list(self.prompt_audio_cache.keys())

    def _pick(self, table, name, default):
        if name is None:
            name = next(iter(table)) if table else default
        return name

    def _phones_and_bert(self, text):
        if self._text_frontend is None:
            raise NotImplementedError("no text frontend installed: call set_text_frontend(fn); the reference's G2P stack "
                                      "(pypinyin/jieba/pyopenjtalk/...) is outside this build's scope")
        phones2, word2ph, bert2, norm_text = self._text_frontend(text)
        if not phones2:
            raise ValueError("text produced no phonemes")
        if bert2 is None:
            bert2 = torch.zeros(len(phones2), 1024)
        return list(phones2), word2ph, bert2.to(self.tts_config.device), norm_text

    def _ge_for(self, spk_audio_path, sovits_model):
        def one(p):
            if p not in self.spk_audio_cache or sovits_model not in self.spk_audio_cache[p]["ge"]:
                self.cache_spk_audio(p, sovits_model=sovits_model)
            return self.spk_audio_cache[p]["ge"][sovits_model]
        if isinstance(spk_audio_path, dict):  # multi-speaker fusion, TTS.py:668-679
            total = sum(spk_audio_path.values())
            ge = None
            for p, wgt in spk_audio_path.items():
                g = one(p) * (wgt / total)
                ge = g if ge is None else ge + g
            return ge
        return one(spk_audio_path)

    def _prompt_for(self, path, text):
        if path not in self.prompt_audio_cache:
            self.cache_prompt_audio(path, text)
        c = self.prompt_audio_cache[path]
        return c["prompt"], c["phones1"], c["bert1"]

    # ------------------------------------------------------------------ multi-GPU (one process per GPU, engine.py)
    def _engine(self, t2s):
        """the continuous-batching engine of this process group, or None in a single process"""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return None
        from .engine import ContinuousBatchingEngine, SpeakerBook
        if getattr(self, "_speaker_book", None) is None:
            self._speaker_book = SpeakerBook(self.tts_config.device)
        return ContinuousBatchingEngine(t2s, slots=max(t2s.cuda_graph_buckets), chunk=2)

    def _sync_speakers(self, prompt_paths, prompt_texts, spk_paths, sovits_model, src=0):
        """SPMD: every rank calls infer_batched with the same arguments; the reference-audio models ran on rank `src`
        only (cache_spk_audio / cache_prompt_audio there).  Each prompt / speaker key is broadcast once, ever."""
        import torch.distributed as dist
        book, me = self._speaker_book, dist.get_rank()
        dev = self.tts_config.device
        for p, text in dict(zip(prompt_paths, prompt_texts)).items():
            key = "prompt:%s" % p
            if key in book.entries:
                continue
            c = self.prompt_audio_cache.get(p) if me == src else None
            if me == src and c is None:
                self.cache_prompt_audio(p, text)
                c = self.prompt_audio_cache[p]
            got = book.sync(key, [c["prompt"], torch.tensor(c["phones1"], dtype=torch.int64, device=dev), c["bert1"]] if me == src else None, src=src)
            if me != src:
                self.prompt_audio_cache[p] = {"prompt": got[0], "phones1": got[1].tolist(), "bert1": got[2], "text": text}
        names = []
        for sp in spk_paths:
            names += list(sp.keys()) if isinstance(sp, dict) else [sp]
        for p in dict.fromkeys(names):
            key = "ge:%s:%s" % (sovits_model, p)
            if key in book.entries:
                continue
            if me == src and (p not in self.spk_audio_cache or sovits_model not in self.spk_audio_cache[p]["ge"]):
                self.cache_spk_audio(p, sovits_model=sovits_model)
            got = book.sync(key, [self.spk_audio_cache[p]["ge"][sovits_model]] if me == src else None, src=src)
            if me != src:
                self.spk_audio_cache.setdefault(p, {"ge": {}})["ge"][sovits_model] = got[0]

    # ------------------------------------------------------------------ trimming (TTS.py:1629-1664)
    @staticmethod
    def _rms_frames(x, frame=512, hop=256):
        if x.shape[0] < frame:
            return x.new_zeros(0)
        return x.unfold(0, frame, hop).pow(2).mean(dim=1).sqrt()

    def _find_head_threshold_offsets(self, audio, threshold=0.02, search_len=64000, margin=3200):
        head = audio[:search_len]
        idx = torch.nonzero(self._rms_frames(head) > threshold)
        return max(0, int(idx[0]) * 256 - margin) if idx.numel() else head.shape[0]

    def _find_tail_threshold_offsets(self, audio, threshold=0.01, search_len=64000, margin=3200):
        tail = audio[-search_len:]
        idx = torch.nonzero(self._rms_frames(tail) > threshold)
        return max(1, tail.shape[0] - int(idx[-1]) * 256 - margin) if idx.numel() else tail.shape[0]

    @staticmethod
    def _check_pause(text):
        return len(text) > 0 and text[-1] in PAUSE_MARKS

    def _close_subtitles(self, subtitles, word2ph, tail_s):
        """TTS.py:255-261 / 472-478 / 772-777: make the list end on a pause mark (a zero-length entry for the
        last word when it is not one) and let that last entry cover the trailing silence."""
        if not self._check_pause(subtitles[-1]["text"]):
            subtitles.append({"text": word2ph["word"][-1], "start_s": subtitles[-1]["end_s"], "end_s": subtitles[-1]["end_s"]})
        if tail_s is not None:
            subtitles[-1]["end_s"] += tail_s

    # ------------------------------------------------------------------ inference
    @torch.inference_mode()
    def infer(self, spk_audio_path, prompt_audio_path, prompt_audio_text, text, return_subtitles=False, top_k=15,
              top_p=1.0, temperature=1.0, repetition_penalty=1.35, noise_scale=0.5, speed=1.0, gpt_model=None,
              sovits_model=None):
        with self._infer_lock:
            try:
                if not self._check_pause(text):
                    text += "."
                gpt_model = self._pick(self.gpt_models, gpt_model, self.default_gpt_path)
                sovits_model = self._pick(self.sovits_models, sovits_model, self.default_sovits_path)
                if gpt_model not in self.gpt_models:
                    self.load_gpt_model(gpt_model)
                if sovits_model not in self.sovits_models:
                    self.load_sovits_model(sovits_model)
                t2s = self.gpt_models[gpt_model].t2s_model
                vq = self.sovits_models[sovits_model].vq_model
                dev = self.tts_config.device
                ge = self._ge_for(spk_audio_path, sovits_model)
                prompt, phones1, bert1 = self._prompt_for(prompt_audio_path, prompt_audio_text)
                phones2, word2ph, bert2, norm_text = self._phones_and_bert(text)
                ids = torch.tensor(phones1 + phones2, dtype=torch.int64, device=dev).unsqueeze(0)
                bert = torch.cat([bert1, bert2]).unsqueeze(0)
                pred = t2s.infer(ids, prompt, bert, top_k=top_k, top_p=top_p, temperature=temperature,
                                 repetition_penalty=repetition_penalty)
                audio, attn = vq.decode(pred, torch.tensor(phones2, dtype=torch.int64, device=dev).unsqueeze(0), ge,
                                        noise_scale=noise_scale, speed=speed)
                audio = audio[0, 0, :]
                subtitles = []
                if return_subtitles:   # TTS.py:250-263 (the reference aligns even when nobody asked; only done on request here)
                    subtitles = sub.get_subtitles(word2ph, sub.viterbi_monotonic(attn), speed, sovits_hz=self.sovits_hz)
                    self._close_subtitles(subtitles, word2ph, 0.2)
                    subtitles = sub.sub2text_index(subtitles, norm_text, text)
                head_offset = self._find_head_threshold_offsets(audio)
                audio = audio[head_offset:]
                if subtitles:
                    sub.increment_subtitle_times(subtitles, -head_offset / self.samplerate)
                    subtitles[0]["start_s"] = max(0, subtitles[0]["start_s"])
                audio = audio.float().cpu().numpy()
                peak = np.abs(audio).max() if audio.size else 0.0
                if peak > 1:
                    audio = audio / peak
                audio = np.concatenate([audio, np.zeros(int(0.2 * self.samplerate), dtype=audio.dtype)])
                return AudioClip(self.audio_queue, audio, self.samplerate, len(audio) / self.samplerate, subtitles, text)
            finally:
                self._empty_cache()

    @torch.inference_mode()
    def infer_vc(self, spk_audio_path, prompt_audio_path, prompt_audio_text, noise_scale=0.5, speed=1.0, sovits_model=None):
        """TTS.py:871-964, voice conversion: the prompt's own semantic tokens and phonemes go straight to the SoVITS
        decoder with the target speaker's `ge`; word timings always come back (the reference aligns unconditionally here)."""
        with self._infer_lock:
            try:
                if not self._check_pause(prompt_audio_text):
                    prompt_audio_text += "."
                sovits_model = self._pick(self.sovits_models, sovits_model, self.default_sovits_path)
                if sovits_model not in self.sovits_models:
                    self.load_sovits_model(sovits_model)
                vq = self.sovits_models[sovits_model].vq_model
                dev = self.tts_config.device
                ge = self._ge_for(spk_audio_path, sovits_model)
                if prompt_audio_path not in self.prompt_audio_cache:
                    self.cache_prompt_audio(prompt_audio_path, prompt_audio_text)
                prompt = self.prompt_audio_cache[prompt_audio_path]["prompt"]
                phones, word2ph, _, norm_text = self._phones_and_bert(prompt_audio_text)
                audio, attn = vq.decode(prompt.unsqueeze(0), torch.tensor(phones, dtype=torch.int64, device=dev).unsqueeze(0), ge,
                                        noise_scale=noise_scale, speed=speed)
                audio = audio[0, 0, :].float().cpu().numpy()
                subtitles = sub.get_subtitles(word2ph, sub.viterbi_monotonic(attn), speed, sovits_hz=self.sovits_hz)
                self._close_subtitles(subtitles, word2ph, 0.2)
                subtitles = sub.sub2text_index(subtitles, norm_text, prompt_audio_text)
                peak = np.abs(audio).max() if audio.size else 0.0
                if peak > 1:
                    audio = audio / peak
                audio = np.concatenate([audio, np.zeros(int(0.2 * self.samplerate), dtype=audio.dtype)])
                return AudioClip(self.audio_queue, audio, self.samplerate, len(audio) / self.samplerate, subtitles, prompt_audio_text)
            finally:
                self._empty_cache()

    # ------------------------------------------------------------------ asyncio front (TTS.py:966-1262)
    # The reference's wrappers take the engine lock around the synchronous call inside an executor thread; here the
    # synchronous methods hold that (non-reentrant) lock themselves, so the wrappers only move the call off the loop.
    async def infer_async(self, *args, executor=None, **kwargs):
        import asyncio
        import functools
        return await asyncio.get_running_loop().run_in_executor(executor, functools.partial(self.infer, *args, **kwargs))

    async def infer_batched_async(self, *args, executor=None, **kwargs):
        import asyncio
        import functools
        return await asyncio.get_running_loop().run_in_executor(executor, functools.partial(self.infer_batched, *args, **kwargs))

    async def infer_stream_async(self, *args, executor=None, **kwargs):
        """async generator of AudioClip chunks: infer_stream runs in an executor thread and hands chunks over a queue"""
        import asyncio
        loop = asyncio.get_running_loop()
        queue = asyncio.Queue()
        failure = []

        def pump():
            try:
                for chunk in self.infer_stream(*args, **kwargs):
                    loop.call_soon_threadsafe(queue.put_nowait, chunk)
            except BaseException as exc:        # surfaced on the consumer side instead of dying in the worker
                failure.append(exc)
            finally:
                loop.call_soon_threadsafe(queue.put_nowait, None)
        loop.run_in_executor(executor, pump)
        while True:
            chunk = await queue.get()
            if chunk is None:
                break
            yield chunk
        if failure:
            raise failure[0]

    def _sola_algorithm(self, f1_overlap, f2, overlap_len, search_len: int = 320):
        """TTS.py:1612-1627, kept by name for callers of the reference's method: one library call (stream.sola -> gsv_sola)."""
        from .stream import sola
        out, k = sola(f1_overlap.reshape(-1)[-overlap_len:], f2.reshape(-1), search_len)
        return out.reshape(1, 1, -1), torch.tensor([[k]], device=f2.device)

    def infer_stream(self, spk_audio_path, prompt_audio_path, prompt_audio_text, text, return_subtitles=False,
                     is_cut_text=True, cut_minlen=10, cut_mute=0.4,
                     cut_mute_scale_map={"…": 2.0, ".": 1.5, "。": 1.5, "?": 1.5, "？": 1.5, "!": 1.5, "！": 1.5, ",": 1.0,
                                         "，": 1.0, ":": 1.0, "：": 1.0, ";": 1.0, "；": 1.0, "~": 1.0, "、": 0.8, "・": 0.8},
                     stream_mode="token", stream_chunk=25, overlap_len=5, boost_first_chunk=True, top_k=15, top_p=1.0,
                     temperature=1.0, repetition_penalty=1.35, noise_scale=0.5, speed=1.0, gpt_model=None,
                     sovits_model=None, debug=True):
        """TTS.py:289-504: generator of AudioClip chunks.  Per text segment the GPT streams cumulative token
        chunks (t2s.infer_stream); every chunk is decoded from the start of the segment with
        decode(stream_mode=True) -- only frames past `valid_start_idx` reach the flow / Generator -- and joined
        to the previous one by SOLA over `overlap_len` frames."""
        with self._infer_lock:
            try:
                if not self._check_pause(text):
                    text += "."
                if stream_mode == "sentence":
                    stream_chunk = 10000
                if not is_cut_text:
                    cut_minlen = 10000
                cut_mute = cut_mute / speed
                gpt_model = self._pick(self.gpt_models, gpt_model, self.default_gpt_path)
                sovits_model = self._pick(self.sovits_models, sovits_model, self.default_sovits_path)
                if gpt_model not in self.gpt_models:
                    self.load_gpt_model(gpt_model)
                if sovits_model not in self.sovits_models:
                    self.load_sovits_model(sovits_model)
                t2s = self.gpt_models[gpt_model].t2s_model
                vq = self.sovits_models[sovits_model].vq_model
                dev = self.tts_config.device
                ge = self._ge_for(spk_audio_path, sovits_model)
                prompt, phones1, bert1 = self._prompt_for(prompt_audio_path, prompt_audio_text)
                overlap_samples = overlap_len * vq.samples_per_frame
                audio_len_s, cur_text_l, last_end_s = 0.0, 0, 0
                for i, text_cut in enumerate(cut_text(text, cut_minlen)):
                    phones2, word2ph, bert2, norm_text = self._phones_and_bert(text_cut)
                    ids = torch.tensor(phones1 + phones2, dtype=torch.int64, device=dev).unsqueeze(0)
                    bert = torch.cat([bert1, bert2]).unsqueeze(0)
                    phones2_t = torch.tensor(phones2, dtype=torch.int64, device=dev).unsqueeze(0)
                    splicer, valid_start_idx, chunk_idx, last_subtitles_end = ChunkSplicer(overlap_samples), 0, 0, 0
                    for pred, is_final in t2s.infer_stream(ids, prompt, bert, top_k=top_k, top_p=top_p, temperature=temperature,
                                                           repetition_penalty=repetition_penalty, stream_chunk=stream_chunk,
                                                           boost_first_chunk=boost_first_chunk if i == 0 else False, debug=debug):
                        with torch.inference_mode():
                            audio, attn = vq.decode(pred, phones2_t, ge, noise_scale=noise_scale, speed=speed, stream_mode=True,
                                                    valid_start_idx=valid_start_idx, overlap_len=overlap_len)
                            audio = splicer.push(audio, is_final)     # aligned to the previous chunk's tail, its own tail kept back
                            if not is_final:
                                attn = attn[:, :-overlap_len, :]
                                valid_start_idx = attn.shape[1]
                            subtitles = []
                            if return_subtitles:   # TTS.py:444-451: a chunk whose path is mostly single frames is not trusted yet
                                assign = sub.viterbi_monotonic(attn)
                                if is_final or sub.is_normal_assign(assign):
                                    subtitles = sub.get_subtitles(word2ph, assign, speed, last_end_s=last_end_s, sovits_hz=self.sovits_hz)
                            if chunk_idx == 0:
                                head_offset = self._find_head_threshold_offsets(audio)
                                audio = audio[head_offset:]
                            if subtitles:
                                sub.increment_subtitle_times(subtitles, -head_offset / self.samplerate)
                                subtitles[0]["start_s"] = max(last_end_s, subtitles[0]["start_s"])
                            if is_final:
                                if text_cut[-1] in cut_mute_scale_map:
                                    scale = cut_mute_scale_map[text_cut[-1]]
                                elif "…" in cut_mute_scale_map and text_cut[-3:] in ["...", "。。。"]:
                                    scale = cut_mute_scale_map["…"]
                                else:
                                    scale = 1.0
                                audio = torch.cat([audio, torch.zeros(int(cut_mute * scale * self.samplerate), dtype=audio.dtype, device=audio.device)])
                                if subtitles:
                                    self._close_subtitles(subtitles, word2ph, cut_mute * scale)
                                    last_end_s = subtitles[-1]["end_s"]
                            new_subtitles = []
                            if subtitles:   # TTS.py:481-486: chunks are cumulative, hand out what is new; the last word stays open
                                subtitles = sub.sub2text_index(subtitles, norm_text, text_cut)
                                sub.increment_subtitle_indices(subtitles, cur_text_l)
                                new_subtitles = subtitles[last_subtitles_end:]
                                last_subtitles_end = len(subtitles) - 1
                                if not is_final and new_subtitles:
                                    new_subtitles[-1]["end_s"] = None
                            audio = audio.float().cpu().numpy()
                        audio_len_s += len(audio) / self.samplerate
                        yield AudioClip(self.audio_queue, audio, self.samplerate, audio_len_s, new_subtitles, text)
                        chunk_idx += 1
                    vq.enc_p.y_overlap = None
                    cur_text_l += len(text_cut)
            finally:
                try:   # an abandoned stream must not leave its cross-fade state to the next one (other speed -> other shape)
                    self.sovits_models[sovits_model].vq_model.enc_p.y_overlap = None
                except Exception:
                    pass
                self._empty_cache()

    @torch.inference_mode()
    def infer_batched(self, spk_audio_paths, prompt_audio_paths, prompt_audio_texts, texts, return_subtitles=False,
                      is_cut_text=True, cut_minlen=10, cut_mute=0.4,
                      cut_mute_scale_map={"…": 2.0, ".": 1.5, "。": 1.5, "?": 1.5, "？": 1.5, "!": 1.5, "！": 1.5,
                                          ",": 1.0, "，": 1.0, ":": 1.0, "：": 1.0, ";": 1.0, "；": 1.0, "~": 1.0,
                                          "、": 0.8, "・": 0.8},
                      top_k=15, top_p=1.0, temperature=1.0, repetition_penalty=1.35, noise_scale=0.5, speed=1.0,
                      bert_batch_size=20, sovits_batch_size=10, gpt_model=None, sovits_model=None):
        with self._infer_lock:
            try:
                if isinstance(texts, str):
                    texts = [texts]
                texts = [t if self._check_pause(t) else t + "." for t in texts]
                if not is_cut_text:
                    cut_minlen = 10000
                cut_mute = cut_mute / speed
                n = len(texts)
                bc = lambda v, kinds: [v] * n if isinstance(v, kinds) else list(v)
                spk_audio_paths = bc(spk_audio_paths, (str, dict))
                prompt_audio_paths = bc(prompt_audio_paths, str)
                prompt_audio_texts = bc(prompt_audio_texts, str)
                gpt_model = self._pick(self.gpt_models, gpt_model, self.default_gpt_path)
                sovits_model = self._pick(self.sovits_models, sovits_model, self.default_sovits_path)
                if gpt_model not in self.gpt_models:
                    self.load_gpt_model(gpt_model)
                if sovits_model not in self.sovits_models:
                    self.load_sovits_model(sovits_model)
                t2s = self.gpt_models[gpt_model].t2s_model
                vq = self.sovits_models[sovits_model].vq_model
                dev = self.tts_config.device

                segs, seg2orig = [], []
                for i, t in enumerate(texts):
                    for c in cut_text(t, cut_minlen):
                        segs.append(c)
                        seg2orig.append(i)
                eng = self._engine(t2s)     # None in a single process
                if eng is not None:         # reference-speaker tensors exist on rank 0 only: ONE broadcast per new key
                    self._sync_speakers(prompt_audio_paths, prompt_audio_texts, spk_audio_paths, sovits_model)
                feats = [self._phones_and_bert(s) for s in segs]
                ids, prompts, berts, ges, phones2_all = [], [], [], [], []
                word2ph_all = [f[1] for f in feats]
                norm_all = [f[3] for f in feats]
                for k, (ph2, _, b2, _) in enumerate(feats):
                    o = seg2orig[k]
                    prompt, ph1, b1 = self._prompt_for(prompt_audio_paths[o], prompt_audio_texts[o])
                    ids.append(torch.tensor(ph1 + ph2, dtype=torch.int64, device=dev))
                    prompts.append(prompt.squeeze(0))
                    berts.append(torch.cat([b1, b2]))
                    ges.append(self._ge_for(spk_audio_paths[o], sovits_model).squeeze(0))
                    phones2_all.append(ph2)

                if eng is None:
                    # staged refill: same tokens per request as the reference-order loop (greedy: rows are independent;
                    # sampling: the noise stream is the request's), no stall of the other slots on a prompt pass
                    pred, orig_idx = t2s.infer_batched(ids, prompts, berts, top_k=top_k, top_p=top_p,
                                                       temperature=temperature, repetition_penalty=repetition_penalty,
                                                       async_refill=True)
                    tokens = [None] * len(segs)
                    for p_, o in zip(pred, orig_idx.tolist()):
                        tokens[o] = p_
                else:   # this rank's share of the segment queue (engine.py), then every rank learns every segment's tokens
                    pred, orig_idx = eng.run_gpt(ids, prompts, berts, costs=[int(i.shape[0]) for i in ids], top_k=top_k, top_p=top_p,
                                                 temperature=temperature, repetition_penalty=repetition_penalty,
                                                 async_refill=True)
                    tokens = eng.exchange({int(o): p_ for p_, o in zip(pred, orig_idx.tolist())}, len(segs), dst=None)
                    eng._retire_cursors(None)
                # TTS.py:705-716 sorts the COMPLETION-order list by length; completion order depends on slot timing (and on
                # the rank count), the request order does not: the balance runs over the request-order lengths, so one
                # process and N ranks form the same vocoder batches and return the same samples
                lengths_all = torch.tensor([len(p_) for p_ in tokens])
                order_all = balance_order(lengths_all)
                batches = [order_all[s:s + sovits_batch_size] for s in range(0, len(order_all), sovits_batch_size)]
                my_batches = range(len(batches)) if eng is None else eng.deal_batches(len(batches))

                audios, subs_out, orig_done = [], [], []
                for b in my_batches:
                    oi = batches[b].tolist()
                    sem = [tokens[o] for o in oi]
                    ln = lengths_all[batches[b]]
                    orig_done += oi
                    ge_cat = torch.cat([ges[o].expand(-1, int(l)) for o, l in zip(oi, ln)], dim=1).unsqueeze(0)
                    ph_cat = torch.cat([torch.tensor(phones2_all[o], dtype=torch.int64, device=dev) for o in oi]).unsqueeze(0)
                    plens = torch.tensor([len(phones2_all[o]) for o in oi], device=dev)
                    ends = torch.cumsum(plens, 0)
                    pairs = torch.stack([ends - plens, ends], dim=1)
                    slice_indices = torch.repeat_interleave(pairs, (ln * 2).to(dev), dim=0)
                    audio, attn = vq.decode(torch.cat(sem).unsqueeze(0).unsqueeze(0), ph_cat, ge_cat, noise_scale=noise_scale,
                                            speed=speed, cuda_graph=False, slice_indices=slice_indices)
                    audio = audio[0, 0, :]
                    if return_subtitles:   # TTS.py:768-777: one alignment over the time-concatenated batch
                        w2p_cat = {"word": [w for o in oi for w in word2ph_all[o]["word"]],
                                   "ph": [c for o in oi for c in word2ph_all[o]["ph"]]}
                        subtitles = sub.get_subtitles(w2p_cat, sub.viterbi_monotonic(attn), speed, sovits_hz=self.sovits_hz)
                        self._close_subtitles(subtitles, w2p_cat, None)
                    peak = audio.abs().max()
                    if peak > 1.0:
                        audio = audio / peak
                    if return_subtitles:   # TTS.py:783-804: the word timings, not the token counts, cut the batch apart
                        last_i = 0
                        for o in oi:
                            best_i = sub.find_subtitles(subtitles, word2ph_all[o], last_i)
                            part = subtitles[last_i:best_i]
                            last_i = best_i
                            a = audio[int(part[0]["start_s"] * self.samplerate):int(part[-1]["end_s"] * self.samplerate)]
                            h, t = self._find_head_threshold_offsets(a), self._find_tail_threshold_offsets(a)
                            audios.append(a[h:-t].float())
                            part[0]["start_s"] += h / self.samplerate
                            part[-1]["end_s"] -= t / self.samplerate
                            subs_out.append(sub.sub2text_index(part, norm_all[o], segs[o]))
                        continue
                    for lo, hi in split_bounds(ln.tolist(), vq.samples_per_frame, speed):   # TTS.py:806-811
                        a = audio[lo:hi]
                        h, t = self._find_head_threshold_offsets(a), self._find_tail_threshold_offsets(a)
                        audios.append(a[h:-t].float())

                if eng is None:
                    ordered = [None] * len(segs)
                    ordered_subs = [None] * len(segs)
                    for cur, o in enumerate(orig_done):
                        ordered[o] = audios[cur].cpu().numpy()
                        if return_subtitles:
                            ordered_subs[o] = subs_out[cur]
                else:   # every rank vocoded its batches; the samples meet on rank `gather_dst` over RCCL (TTS.py:820-865)
                    dst = self.gather_dst
                    full = eng.exchange({int(o): audios[cur].contiguous() for cur, o in enumerate(orig_done)}, len(segs), dst=dst)
                    ordered_subs = [None] * len(segs)
                    if return_subtitles:
                        ordered_subs = eng.gather({int(o): subs_out[cur] for cur, o in enumerate(orig_done)}, len(segs), dst=dst)
                    if full is None:
                        return None     # not the gathering rank
                    ordered = [a.cpu().numpy() for a in full]
                per_text = [[] for _ in range(n)]
                per_text_subs = [[] for _ in range(n)]
                last_orig, cur_text_l = None, 0
                for k, a in enumerate(ordered):
                    per_text[seg2orig[k]].append(a)
                    tail = segs[k][-1]
                    if tail in cut_mute_scale_map:
                        sc = cut_mute_scale_map[tail]
                    elif "…" in cut_mute_scale_map and segs[k][-3:] in ("...", "。。。"):
                        sc = cut_mute_scale_map["…"]
                    else:
                        sc = 1.0
                    per_text[seg2orig[k]].append(np.zeros(int(cut_mute * sc * self.samplerate), dtype=a.dtype))
                    if return_subtitles:   # TTS.py:843-852: spans are per segment; shift them into the whole text
                        if seg2orig[k] != last_orig:
                            cur_text_l, last_orig = 0, seg2orig[k]
                        ordered_subs[k][-1]["end_s"] += cut_mute * sc
                        sub.increment_subtitle_indices(ordered_subs[k], cur_text_l)
                        per_text_subs[seg2orig[k]].append(ordered_subs[k])
                        cur_text_l += len(segs[k])
                clips = []
                for parts, sparts, t in zip(per_text, per_text_subs, texts):
                    a = np.concatenate(parts) if parts else np.zeros(0, np.float32)
                    subtitles = sub.cat_subtitles(*sparts) if return_subtitles else []
                    clips.append(AudioClip(self.audio_queue, a, self.samplerate, len(a) / self.samplerate, subtitles, t))
                return tuple(clips)
            finally:
                self._empty_cache()
