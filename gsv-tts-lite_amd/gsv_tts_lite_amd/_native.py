"""ctypes binding of the C ABI in include/gsv_tts_hip.h (libgsv_hip.so, gfx950 only).

There is no fallback: if the shared library is missing, or a call fails, this raises.
PyTorch is used by callers only for device memory and streams; every pointer crossing
this boundary is a raw device address (`tensor.data_ptr()`).
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSV_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libgsv_hip.so")   # GSV_HIP_LIB: an experimental build of the same ABI (tools/)
CSRC = os.path.join(os.path.dirname(_HERE), "csrc")

GSV_F32, GSV_BF16, GSV_FP8 = 0, 1, 2

EXPORTS = [
    "gsv_version", "gsv_last_error",
    "gsv_t2s_create", "gsv_t2s_destroy", "gsv_t2s_load_tensor", "gsv_t2s_finalize", "gsv_t2s_bind_state", "gsv_t2s_unbind_state",
    "gsv_t2s_embed_prompt", "gsv_t2s_prefill_workspace", "gsv_t2s_prefill", "gsv_t2s_set_eos_mirror", "gsv_t2s_prefill_slots", "gsv_t2s_prefill_slots_staged", "gsv_t2s_commit_slots", "gsv_t2s_adopt_slots", "gsv_t2s_move_slots", "gsv_t2s_decode_hidden",
    "gsv_t2s_decode", "gsv_t2s_flush", "gsv_t2s_time_kernels", "gsv_t2s_set_debug", "gsv_t2s_batched_min", "gsv_t2s_ffn_slices", "gsv_t2s_device_bytes",
    "gsv_voc_create", "gsv_voc_destroy", "gsv_voc_load_tensor", "gsv_voc_finalize", "gsv_voc_workspace",
    "gsv_voc_flow_dec", "gsv_voc_flow_dec_graph", "gsv_voc_resample_linear", "gsv_voc_flow", "gsv_voc_dec", "gsv_voc_has_enc_p", "gsv_voc_enc_workspace", "gsv_voc_enc_p", "gsv_voc_decode_workspace", "gsv_voc_decode",
    "gsv_align_workspace", "gsv_align_viterbi", "gsv_sola_workspace", "gsv_sola",
    "gsv_ref_create", "gsv_ref_destroy", "gsv_ref_load_tensor", "gsv_ref_finalize", "gsv_ref_workspace",
    "gsv_ref_spectrogram", "gsv_ref_get_ge", "gsv_ref_extract_latent",
]


class T2SConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in
                ("n_layer", "hidden", "n_head", "vocab", "eos", "n_pos", "n_phoneme", "dtype")]


class T2SState(ctypes.Structure):
    _fields_ = [("batch", ctypes.c_int), ("max_kv", ctypes.c_int)] + [(n, ctypes.c_void_p) for n in (
        "k_cache", "v_cache", "kv_len", "x_len", "pre_tokens", "seen", "step", "eos_at", "logits", "hidden",
        "tok_override", "ctl", "fctl")]


class VocConfig(ctypes.Structure):
    _fields_ = [("inter_channels", ctypes.c_int), ("hidden_channels", ctypes.c_int), ("gin_channels", ctypes.c_int),
                ("upsample_initial_channel", ctypes.c_int), ("n_upsample", ctypes.c_int),
                ("upsample_rates", ctypes.c_int * 8), ("upsample_kernel_sizes", ctypes.c_int * 8),
                ("n_resblock_kernels", ctypes.c_int), ("resblock_kernel_sizes", ctypes.c_int * 4),
                ("resblock_dilations", ctypes.c_int * 4), ("n_flows", ctypes.c_int), ("dtype", ctypes.c_int)]


class RefConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in
                ("n_fft", "hop", "spec_bins", "hidden", "n_head", "kernel", "gin", "sv_dim", "ssl_dim", "bins")]


_LIB = None


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "gsv_tts_lite_amd: HIP extension %s is missing. Build it with `python -c \"import __graft_entry__ as g; "
            "g.build()\"` (hipcc --offload-arch=gfx950). There is no CPU or PyTorch fallback for the hot path." % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp, i, i64, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t
    L.gsv_version.restype = i
    L.gsv_last_error.restype = ctypes.c_char_p
    sig = {
        "gsv_t2s_create": [ctypes.POINTER(T2SConfig), ctypes.POINTER(vp)],
        "gsv_t2s_destroy": [vp],
        "gsv_t2s_load_tensor": [vp, ctypes.c_char_p, vp, i64, vp],
        "gsv_t2s_finalize": [vp, vp],
        "gsv_t2s_bind_state": [vp, ctypes.POINTER(T2SState)],
        "gsv_t2s_unbind_state": [vp, i],
        "gsv_t2s_embed_prompt": [vp, i, i, i, i, vp, vp, vp, vp, vp, vp, vp, vp],
        "gsv_t2s_prefill": [vp, i, i, i, i, vp, vp, vp, vp, sz, vp],
        "gsv_t2s_set_eos_mirror": [vp, i, vp],
        "gsv_t2s_prefill_slots": [vp, i, vp, i, i, vp, vp, vp, vp, sz, vp],
        "gsv_t2s_prefill_slots_staged": [vp, i, vp, i, i, vp, vp, vp, vp, sz, vp],
        "gsv_t2s_commit_slots": [vp, i, vp, i, vp],
        "gsv_t2s_adopt_slots": [vp, i, vp, i, vp, vp, i, vp],
        "gsv_t2s_move_slots": [vp, i, vp, i, vp, i, vp],
        "gsv_t2s_decode_hidden": [vp, i, vp, vp],
        "gsv_t2s_decode": [vp, i, i, i, vp],
        "gsv_t2s_flush": [vp, i, vp],
        "gsv_t2s_set_debug": [vp, vp],
        "gsv_t2s_batched_min": [vp],
        "gsv_t2s_ffn_slices": [vp, ctypes.c_int],
        "gsv_t2s_time_kernels": [vp, i, i, ctypes.POINTER(ctypes.c_float), vp],
        "gsv_voc_create": [ctypes.POINTER(VocConfig), ctypes.POINTER(vp)],
        "gsv_voc_destroy": [vp],
        "gsv_voc_load_tensor": [vp, ctypes.c_char_p, vp, i64, vp],
        "gsv_voc_finalize": [vp, vp],
        "gsv_voc_flow_dec": [vp, vp, vp, vp, i, i, vp, vp, sz, vp],
        "gsv_voc_flow_dec_graph": [vp, vp, vp, vp, i, i, vp, vp, sz, vp],
        "gsv_voc_resample_linear": [vp, i, i, vp, i, vp],
        "gsv_voc_flow": [vp, vp, vp, vp, i, i, vp, vp, sz, vp],
        "gsv_voc_dec": [vp, vp, vp, i, i, vp, vp, sz, vp],
        "gsv_voc_has_enc_p": [vp],
        "gsv_voc_enc_p": [vp, vp, i, vp, i, vp, i, vp, vp, vp, vp, vp, sz, vp],
        "gsv_voc_decode": [vp, vp, i, vp, i, vp, i, vp, ctypes.c_float, ctypes.c_uint64, i, i, i, vp, i, i, vp, vp, vp, sz, vp],
        "gsv_align_viterbi": [vp, i, i, i, vp, vp, sz, vp],
        "gsv_sola": [vp, vp, i, i, i, vp, vp, vp, sz, vp],
        "gsv_ref_create": [ctypes.POINTER(RefConfig), ctypes.POINTER(vp)],
        "gsv_ref_destroy": [vp],
        "gsv_ref_load_tensor": [vp, ctypes.c_char_p, vp, i64, vp],
        "gsv_ref_finalize": [vp, vp],
        "gsv_ref_spectrogram": [vp, vp, i, vp, vp, sz, vp],
        "gsv_ref_get_ge": [vp, vp, i, vp, vp, vp, sz, vp],
        "gsv_ref_extract_latent": [vp, vp, i, vp, vp, vp, sz, vp],
    }
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = i
    L.gsv_t2s_prefill_workspace.argtypes = [vp, i, i]
    L.gsv_t2s_prefill_workspace.restype = sz
    L.gsv_t2s_device_bytes.argtypes = [vp]
    L.gsv_t2s_device_bytes.restype = sz
    L.gsv_voc_workspace.argtypes = [vp, i]
    L.gsv_voc_workspace.restype = sz
    L.gsv_voc_enc_workspace.argtypes = [vp, i, i]
    L.gsv_voc_enc_workspace.restype = sz
    L.gsv_voc_decode_workspace.argtypes = [vp, i, i, i, i, i]
    L.gsv_voc_decode_workspace.restype = sz
    L.gsv_ref_workspace.argtypes = [vp, i, i, i]
    L.gsv_ref_workspace.restype = sz
    L.gsv_align_workspace.argtypes = [i, i]
    L.gsv_align_workspace.restype = sz
    L.gsv_sola_workspace.argtypes = [i]
    L.gsv_sola_workspace.restype = sz
    _LIB = L
    return L


def check(rc: int):
    if rc != 0:
        msg = lib().gsv_last_error()
        raise RuntimeError("gsv_tts_hip error %d: %s" % (rc, msg.decode("utf-8", "replace") if msg else "?"))


def dtype_code(torch_dtype) -> int:
    import torch
    if torch_dtype == torch.float32:
        return GSV_F32
    if torch_dtype == torch.bfloat16:
        return GSV_BF16
    if torch_dtype == torch.float8_e4m3fn:   # gsv_t2s only: bf16 + e4m3 QKV / FFN in the batched decode step
        return GSV_FP8
    raise ValueError("the MI355X hot path supports float32 (parity), bfloat16 (production) and float8_e4m3fn (GPT batched step); "
                     "got %s" % torch_dtype)


def current_stream_ptr(device=None) -> int:
    import torch
    return torch.cuda.current_stream(device).cuda_stream
