"""Thin Python mirror of the C ABI: one `Engine` = one GPU, S independent IQ streams (tests and bench plumbing)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import capi
from .capi import HabdecError, check, lib


@dataclass
class EngineConfig:
    n_streams: int = 1
    max_chunk: int = 65536
    sampling_rate: float = 2.048e6
    decimation: int = 64
    baud: float = 300.0
    rtty_bits: int = 8
    rtty_stops: float = 2.0
    lowpass_bw_hz: float = 1500.0
    lowpass_trans: float = 0.025
    dc_remove: bool = False
    lookup_mode: int = 1
    enable_spectrum: bool = True
    ungated: bool = False
    keep_filtered: bool = False
    pipeline: int = 0        # 0 synchronous, 1 batch mode (results one call later), 2 batch mode with one more call in flight
    arith: int = 0           # 0 exact (separately rounded multiply and add: bit-identical floats), 1 fast (fused multiply-add, tolerance 1e-5)
    device: int = 0


class Engine:
    def __init__(self, cfg: EngineConfig | None = None, **kw):
        cfg = cfg or EngineConfig(**kw)
        self.cfg = cfg
        L = lib()
        c = capi.hd_engine_config()
        L.hd_engine_config_default(C.byref(c))
        for k in ("device", "n_streams", "max_chunk", "sampling_rate", "decimation", "baud", "rtty_bits", "rtty_stops",
                  "lowpass_bw_hz", "lowpass_trans", "lookup_mode"):
            setattr(c, k, getattr(cfg, k))
        c.dc_remove, c.enable_spectrum, c.ungated, c.keep_filtered = int(cfg.dc_remove), int(cfg.enable_spectrum), int(cfg.ungated), int(cfg.keep_filtered)
        c.pipeline = int(cfg.pipeline)
        c.arith = int(cfg.arith)
        h = C.c_void_p()
        check(L.hd_engine_create(C.byref(c), C.byref(h)))
        self.h, self.L = h, L
        self.S = cfg.n_streams
        self._cbs = []

    def close(self):
        if getattr(self, "h", None):
            self.L.hd_engine_destroy(self.h)
            self.h = None
        for p in getattr(self, "_pinned", []):
            self.L.hd_pinned_free(p)
        self._pinned = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- data path
    def process_host(self, iq: np.ndarray, n: int | None = None):
        """iq: complex64 [S, stride]; stream s hands over its first n samples."""
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        if iq.ndim == 1:
            iq = iq[None, :]
        assert iq.shape[0] == self.S
        n = iq.shape[1] if n is None else n
        check(self.L.hd_process_host(self.h, iq.ctypes.data, iq.shape[1], None, n))

    def pinned_array(self, shape, dtype=np.complex64) -> np.ndarray:
        """A numpy array in page-locked, GPU-mapped memory (hd_pinned_alloc): process_host() of a synchronous engine reads it in place over PCIe."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = self.L.hd_pinned_alloc(n)
        if not p:
            raise HabdecError("hd_pinned_alloc failed")
        buf = (C.c_char * n).from_address(p)
        arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
        self._pinned = getattr(self, "_pinned", []) + [p]
        return arr

    def process_device(self, dev_ptr: int, stride: int, n: int):
        check(self.L.hd_process_device(self.h, dev_ptr, stride, None, n))

    def ingest(self, files: "IqFiles", max_rounds: int = 1 << 62) -> int:
        """Pump a batch of cf32 files (one per stream) through the engine (hd_ingest_run); returns the samples consumed."""
        import ctypes as C
        done = C.c_uint64(0)
        check(self.L.hd_ingest_run(self.h, files.h, max_rounds, C.byref(done)))
        return int(done.value)

    def flush(self):
        check(self.L.hd_flush(self.h))

    # ---- control
    def set_baud(self, s, baud): check(self.L.hd_stream_set_baud(self.h, s, baud))
    def set_rtty(self, s, bits, stops): check(self.L.hd_stream_set_rtty(self.h, s, bits, stops))
    def set_lowpass_bw(self, s, hz): check(self.L.hd_stream_set_lowpass_bw(self.h, s, hz))
    def set_lowpass_trans(self, s, t): check(self.L.hd_stream_set_lowpass_trans(self.h, s, t))
    def set_dc_remove(self, s, on): check(self.L.hd_stream_set_dc_remove(self.h, s, int(on)))
    def reset_frequency_correction(self, s, c): check(self.L.hd_stream_reset_frequency_correction(self.h, s, c))

    def on_sentence(self, fn):
        cb = capi.SENTENCE_CB(lambda user, s, call, data, crc: fn(s, call.decode("latin-1"), data.decode("latin-1"), crc.decode("latin-1")))
        self._cbs.append(cb)
        self.L.hd_set_sentence_callback(self.h, cb, None)

    # ---- results
    def _text(self, fn, s, cap=1 << 16) -> str:
        buf = C.create_string_buffer(cap)
        n = fn(self.h, s, buf, cap)
        if n >= cap:
            return self._text(fn, s, n + 1)
        return buf.raw[:n].decode("latin-1")

    def rtty(self, s=0): return self._text(self.L.hd_stream_rtty, s)
    def last_sentence(self, s=0): return self._text(self.L.hd_stream_last_sentence, s)
    def take_sentences(self, s=0): return [x for x in self._text(self.L.hd_stream_take_sentences, s).split("\n") if x]
    def take_matches(self, s=0): return [x for x in self._text(self.L.hd_stream_take_matches, s).split("\n") if x]
    def take_chars(self, s=0): return self._text(self.L.hd_stream_take_chars, s)
    def sentences_ok(self) -> int: return self.L.hd_engine_sentences_ok(self.h)

    def afc(self, s=0) -> dict:
        a = capi.hd_afc_info()
        check(self.L.hd_stream_afc(self.h, s, C.byref(a)))
        return {"correction": a.frequency_correction, "shift_hz": a.shift_hz, "noise_floor": a.noise_floor,
                "noise_var": a.noise_variance, "peak_l": a.peak_left, "peak_r": a.peak_right, "spectra": a.spectra}

    def _arr(self, fn, s, cplx, cap):
        buf = np.zeros(cap * (2 if cplx else 1), np.float32)
        n = fn(self.h, s, buf, cap)
        if n > cap:
            return self._arr(fn, s, cplx, n)
        out = buf[: n * (2 if cplx else 1)]
        return out.view(np.complex64) if cplx else out

    def spectrum(self, s=0): return self._arr(self.L.hd_stream_spectrum, s, True, 4096)
    def power(self, s=0): return self._arr(self.L.hd_stream_power, s, False, 4096)
    def demodulated(self, s=0): return self._arr(self.L.hd_stream_demodulated, s, False, 1 << 15)
    def decimated(self, s=0): return self._arr(self.L.hd_stream_decimated, s, True, 1 << 16)
    def filtered(self, s=0): return self._arr(self.L.hd_stream_filtered, s, True, 1 << 15)
    def fir_taps(self, s=0): return self._arr(self.L.hd_stream_fir_taps, s, False, 8192)

    def bits(self, s=0) -> np.ndarray:
        cap = 1 << 16
        buf = np.zeros(cap, np.uint8)
        n = self.L.hd_stream_bits(self.h, s, buf, cap)
        return buf[:n].copy()

    def flips(self, s=0) -> np.ndarray:
        buf = np.zeros(4096, np.uint32)
        n = self.L.hd_stream_flips(self.h, s, buf, 4096)
        return buf[:n].copy()

    def symbol_backlog(self, s=0) -> int: return self.L.hd_stream_symbol_backlog(self.h, s)

    def bits_total(self, s=0) -> int: return int(self.L.hd_stream_bits_total(self.h, s))
    def flip_list_full(self, s=0) -> int: return int(self.L.hd_stream_flip_list_full(self.h, s))

    def demod_checksum(self, s=0):
        """(call index, n, ck0, ck1) of the call delivered last for stream s -- no flush; n is None where the launch path does not compute it."""
        ci, n, ck = C.c_uint64(0), C.c_uint32(0), (C.c_uint32 * 2)()
        check(self.L.hd_stream_demod_checksum(self.h, s, C.byref(ci), C.byref(n), ck))
        return int(ci.value), (None if n.value == 0xFFFFFFFF else int(n.value)), int(ck[0]), int(ck[1])

    def demod_checksum_total(self, s=0):
        """(calls folded in, delivered calls without a checksum, hash): every delivered call's discriminator checksum folded into one word -- no flush."""
        n, u, h = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        check(self.L.hd_stream_demod_checksum_total(self.h, s, C.byref(n), C.byref(u), C.byref(h)))
        return int(n.value), int(u.value), int(h.value)

    def timing(self) -> dict:
        t = capi.hd_timing()
        check(self.L.hd_engine_timing(self.h, C.byref(t)))
        return {"ms_total": t.ms_total, "ms_front": t.ms_front, "front_bytes": t.front_bytes, "samples": t.samples,
                "host_enqueue_us": t.host_enqueue_us, "host_wait_us": t.host_wait_us, "host_text_us": t.host_text_us,
                "timed_calls": t.timed_calls, "path": t.path, "step_variant": t.step_variant, "host_calls_in_place": t.host_calls_in_place, "lowpass_fft_calls": t.lowpass_fft_calls}

    def set_timing(self, every: int):
        """HIP-event timing on every `every`-th call (0 = off); see hd_engine_set_timing."""
        self.L.hd_engine_set_timing(self.h, every)


class IqFiles:
    """Batched cf32 file source (hd_host_iqfiles_*): one IQSource_File-like reader per stream, read in lock step."""

    def __init__(self, paths, chunk: int, granule: int, loop: bool = False, realtime_rate: float = 0.0):
        import ctypes as C
        self.L = lib()
        arr = (C.c_char_p * len(paths))(*[str(p).encode() for p in paths])
        self.h = self.L.hd_host_iqfiles_open(arr, len(paths), int(loop), chunk, granule, realtime_rate)
        if not self.h:
            raise HabdecError("hd_host_iqfiles_open failed (missing file, chunk < granule, ...)")
        self.S, self.chunk = len(paths), chunk

    def close(self):
        if getattr(self, "h", None):
            self.L.hd_host_iqfiles_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def count(self, s: int) -> int: return int(self.L.hd_host_iqfiles_count(self.h, s))
    def rewinds(self, s: int) -> int: return int(self.L.hd_host_iqfiles_rewinds(self.h, s))

    def next(self, stride: int | None = None):
        """One round: returns (slab complex64 [S, stride], n_per_stream uint32 [S], streams that read anything)."""
        import ctypes as C
        stride = stride or self.chunk
        slab = np.zeros((self.S, stride), np.complex64)
        n = np.zeros(self.S, np.uint32)
        alive = self.L.hd_host_iqfiles_next(self.h, slab.view(np.float32), stride, n.ctypes.data_as(C.POINTER(C.c_uint32)))
        return slab, n, int(alive)
