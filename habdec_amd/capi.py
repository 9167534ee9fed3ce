"""ctypes binding of include/habdec_amd.h and include/habdec_amd_host.h (1:1, no logic)."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

LIB_PATH = Path(__file__).resolve().parent / "libhabdec_amd.so"
_lib = None


class HabdecError(RuntimeError):
    pass


class hd_engine_config(C.Structure):
    _fields_ = [("device", C.c_int32), ("n_streams", C.c_uint32), ("max_chunk", C.c_uint32), ("sampling_rate", C.c_double),
                ("decimation", C.c_uint32), ("baud", C.c_double), ("rtty_bits", C.c_uint32), ("rtty_stops", C.c_float),
                ("lowpass_bw_hz", C.c_float), ("lowpass_trans", C.c_float), ("dc_remove", C.c_int32), ("lookup_mode", C.c_int32),
                ("enable_spectrum", C.c_int32), ("ungated", C.c_int32), ("keep_filtered", C.c_int32), ("pipeline", C.c_int32), ("arith", C.c_int32)]


class hd_afc_info(C.Structure):
    _fields_ = [("frequency_correction", C.c_double), ("shift_hz", C.c_double), ("noise_floor", C.c_double),
                ("noise_variance", C.c_double), ("peak_left", C.c_int32), ("peak_right", C.c_int32), ("spectra", C.c_uint64)]


class hd_timing(C.Structure):
    _fields_ = [("ms_total", C.c_double), ("ms_front", C.c_double), ("front_bytes", C.c_uint64), ("samples", C.c_uint64),
                ("host_enqueue_us", C.c_double), ("host_wait_us", C.c_double), ("host_text_us", C.c_double), ("timed_calls", C.c_uint64),
                ("path", C.c_uint32), ("step_variant", C.c_uint32), ("host_calls_in_place", C.c_uint64), ("lowpass_fft_calls", C.c_uint64)]


SENTENCE_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_char_p, C.c_char_p, C.c_char_p)
MATCH_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int)
CHARS_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.POINTER(C.c_char), C.c_size_t)

_vp, _u32, _sz, _dbl, _int, _f = C.c_void_p, C.c_uint32, C.c_size_t, C.c_double, C.c_int, C.c_float
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")

# name -> (restype, argtypes): every symbol declared in include/habdec_amd.h
ENGINE_API = {
    "hd_engine_config_default": (None, [C.POINTER(hd_engine_config)]),
    "hd_engine_create": (_int, [C.POINTER(hd_engine_config), C.POINTER(_vp)]),
    "hd_engine_destroy": (None, [_vp]),
    "hd_pinned_alloc": (_vp, [_sz]),
    "hd_pinned_free": (None, [_vp]),
    "hd_last_error": (C.c_char_p, []),
    "hd_engine_streams": (_u32, [_vp]),
    "hd_engine_decimation": (_u32, [_vp]),
    "hd_engine_decimated_rate": (_dbl, [_vp]),
    "hd_stream_set_baud": (_int, [_vp, _u32, _dbl]),
    "hd_stream_set_rtty": (_int, [_vp, _u32, _u32, _f]),
    "hd_stream_set_lowpass_bw": (_int, [_vp, _u32, _f]),
    "hd_stream_set_lowpass_trans": (_int, [_vp, _u32, _f]),
    "hd_stream_set_dc_remove": (_int, [_vp, _u32, _int]),
    "hd_stream_reset_frequency_correction": (_int, [_vp, _u32, _dbl]),
    "hd_set_sentence_callback": (None, [_vp, SENTENCE_CB, _vp]),
    "hd_set_match_callback": (None, [_vp, MATCH_CB, _vp]),
    "hd_set_chars_callback": (None, [_vp, CHARS_CB, _vp]),
    "hd_process_host": (_int, [_vp, _vp, _sz, _vp, _u32]),
    "hd_process_device": (_int, [_vp, _vp, _sz, _vp, _u32]),
    "hd_flush": (_int, [_vp]),
    "hd_ingest_run": (_int, [_vp, _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "hd_stream_rtty": (_sz, [_vp, _u32, C.c_char_p, _sz]),
    "hd_stream_last_sentence": (_sz, [_vp, _u32, C.c_char_p, _sz]),
    "hd_stream_take_sentences": (_sz, [_vp, _u32, C.c_char_p, _sz]),
    "hd_stream_take_matches": (_sz, [_vp, _u32, C.c_char_p, _sz]),
    "hd_stream_take_chars": (_sz, [_vp, _u32, C.c_char_p, _sz]),
    "hd_engine_sentences_ok": (C.c_uint64, [_vp]),
    "hd_stream_afc": (_int, [_vp, _u32, C.POINTER(hd_afc_info)]),
    "hd_stream_spectrum": (_sz, [_vp, _u32, _f32p, _sz]),
    "hd_stream_power": (_sz, [_vp, _u32, _f32p, _sz]),
    "hd_stream_demodulated": (_sz, [_vp, _u32, _f32p, _sz]),
    "hd_stream_decimated": (_sz, [_vp, _u32, _f32p, _sz]),
    "hd_stream_filtered": (_sz, [_vp, _u32, _f32p, _sz]),
    "hd_stream_bits": (_sz, [_vp, _u32, _u8p, _sz]),
    "hd_stream_flips": (_sz, [_vp, _u32, _u32p, _sz]),
    "hd_stream_fir_taps": (_sz, [_vp, _u32, _f32p, _sz]),
    "hd_stream_symbol_backlog": (_u32, [_vp, _u32]),
    "hd_stream_bits_total": (C.c_uint64, [_vp, _u32]),
    "hd_stream_flip_list_full": (C.c_uint64, [_vp, _u32]),
    "hd_stream_demod_checksum": (C.c_int, [_vp, _u32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "hd_stream_demod_checksum_total": (C.c_int, [_vp, _u32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "hd_min_chunk": (_u32, [_u32]),
    "hd_engine_timing": (_int, [_vp, C.POINTER(hd_timing)]),
    "hd_engine_set_timing": (None, [_vp, _int]),
}
# every symbol declared in include/habdec_amd_host.h
HOST_API = {
    "hd_host_decim_plan": (_int, [C.c_uint, C.POINTER(_int), C.POINTER(C.c_uint)]),
    "hd_host_decim_taps": (_sz, [C.c_uint, _int, _f32p, _sz]),
    "hd_host_lowpass_design": (_sz, [_f, _f, _sz, _sz, _int, _f32p, _sz]),
    "hd_host_rtty_new": (_vp, [_sz, _f]),
    "hd_host_rtty_free": (None, [_vp]),
    "hd_host_rtty_pending": (_sz, [_vp]),
    "hd_host_rtty_push_run": (_sz, [_vp, _u8p, _sz, C.c_char_p, _sz]),
    "hd_host_crc16": (None, [C.c_char_p, _sz, C.c_char_p]),
    "hd_host_extract_sentence": (_int, [C.c_char_p, _sz, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, _sz]),
    "hd_host_text_new": (_vp, [_sz, _f]),
    "hd_host_text_free": (None, [_vp]),
    "hd_host_text_push_bits": (None, [_vp, _u8p, _sz]),
    "hd_host_text_get": (_sz, [_vp, _int, C.c_char_p, _sz]),
    "hd_host_afc_new": (_vp, []),
    "hd_host_afc_free": (None, [_vp]),
    "hd_host_afc_step": (None, [_vp, _int, _int, _int, _int, _f, _f, _dbl, _dbl, _sz, _dbl]),
    "hd_host_afc_reset": (None, [_vp, _dbl, _sz, _dbl]),
    "hd_host_afc_get": (None, [_vp] + [C.POINTER(_dbl)] * 4 + [C.POINTER(_int)] * 2),
    "hd_host_atan2f": (None, [_f32p, _f32p, _f32p, _sz]),
    "hd_host_discriminate": (None, [_f32p, _sz, _f, _f, _f32p]),
    "hd_host_spectrum_payload": (_sz, [_f32p, _sz, _dbl, _dbl, _dbl, _dbl, _int, _int, _f, _int, _int, _u8p, _sz, C.POINTER(_sz)]),
    "hd_host_demod_payload": (_sz, [_f32p, _sz, _int, _int, _u8p, _sz, C.POINTER(_sz)]),
    "hd_host_parse_time": (_int, [C.c_char_p, C.POINTER(_int), C.POINTER(_int), C.POINTER(_f)]),
    "hd_host_parse_gps_pos": (_int, [C.c_char_p, C.POINTER(_f)]),
    "hd_host_parse_sentence": (_int, [C.c_char_p, _vp]),
    "hd_host_timestamp_from_hms": (_sz, [C.c_int64, _int, _int, _f, C.c_char_p, _sz]),
    "hd_host_gps_distance": (None, [_dbl] * 6 + [C.POINTER(_dbl)]),
    "hd_host_sondehub_new": (_vp, [C.c_char_p, C.c_char_p]),
    "hd_host_sondehub_free": (None, [_vp]),
    "hd_host_sondehub_push_sentence": (_int, [_vp, _u32, C.c_char_p, C.c_char_p, C.c_int64]),
    "hd_host_sondehub_push": (_int, [_vp, C.c_char_p, C.c_char_p, C.c_char_p, _int, _f, _f, _f]),
    "hd_host_sondehub_size": (_sz, [_vp]),
    "hd_host_sondehub_take": (_sz, [_vp, C.c_int64, C.c_char_p, _sz, C.POINTER(_sz)]),
    "hd_host_utc_iso": (_sz, [C.c_int64, C.c_char_p, _sz]),
    "hd_host_json_number": (_sz, [_dbl, C.c_char_p, _sz]),
    "hd_host_iqfiles_open": (_vp, [C.POINTER(C.c_char_p), C.c_uint32, _int, C.c_uint32, C.c_uint32, _dbl]),
    "hd_host_iqfiles_close": (None, [_vp]),
    "hd_host_iqfiles_streams": (C.c_uint32, [_vp]),
    "hd_host_iqfiles_count": (C.c_uint64, [_vp, C.c_uint32]),
    "hd_host_iqfiles_rewinds": (C.c_uint64, [_vp, C.c_uint32]),
    "hd_host_iqfiles_next": (C.c_uint32, [_vp, _f32p, _sz, C.POINTER(C.c_uint32)]),
}


def lib() -> C.CDLL:
    """Load libhabdec_amd.so (built in-tree by `python -m habdec_amd.build`).  Fails loudly: no fallback exists."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise HabdecError(f"{LIB_PATH} is missing: build it with `python -m habdec_amd.build` (hipcc, gfx950). "
                              "habdec_amd has no CPU or PyTorch fallback for the data path.")
        L = C.CDLL(str(LIB_PATH))
        for table in (ENGINE_API, HOST_API):
            for name, (res, args) in table.items():
                f = getattr(L, name)   # AttributeError = header and library out of sync
                f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise HabdecError(f"habdec_amd error {rc}: {lib().hd_last_error().decode()}")
