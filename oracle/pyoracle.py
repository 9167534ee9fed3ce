"""ctypes bindings for the CPU oracle (oracle/liboracle.so) and, when built, for the compiled
reference stage classes (oracle/_ref/libhabdec_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (habdec_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_vp, _sz, _dbl, _int = C.c_void_p, C.c_size_t, C.c_double, C.c_int
_pf = C.POINTER(C.POINTER(C.c_float))
_pu8 = C.POINTER(C.POINTER(C.c_uint8))
_pc = C.POINTER(C.c_char_p)
_psz = C.POINTER(C.POINTER(C.c_size_t))

FFT_FN = C.CFUNCTYPE(None, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_size_t)


def build(ref: bool = False) -> None:
    subprocess.run(["make", "-C", str(HERE)] + (["ref"] if ref else []), check=True, capture_output=True)


def _load(path: Path) -> C.CDLL:
    if not path.exists():
        raise FileNotFoundError(f"{path} missing: run `make -C oracle` (and `make -C oracle ref` where /root/reference exists)")
    return C.CDLL(str(path))


def _cf(a) -> np.ndarray:
    """complex64 array -> contiguous float32 view (interleaved I,Q)."""
    a = np.ascontiguousarray(a, dtype=np.complex64)
    return a.view(np.float32)


class _Lib:
    """Shared signature table: the oracle (`orc_`) and the reference harness (`ref_`) expose the same shapes."""

    def __init__(self, lib: C.CDLL, p: str):
        self.lib, self.p = lib, p

    def fn(self, name, restype, *argtypes):
        f = getattr(self.lib, self.p + name)
        f.restype, f.argtypes = restype, list(argtypes)
        return f


class Stages:
    """Stage-level API; `kind` is 'oracle' (our restatement) or 'ref' (compiled reference classes)."""

    def __init__(self, kind: str = "oracle"):
        self.kind = kind
        if kind == "oracle":
            self.L = _Lib(_load(HERE / "liboracle.so"), "orc_")
        elif kind in ("ref", "ref_mathh"):
            # "ref_mathh": the same reference sources compiled with `-include math.h` (float sin/cos overloads
            # win inside habdec_windows.h; pins the oracle's FIR design mode 1)
            so = "libhabdec_ref.so" if kind == "ref" else "libhabdec_ref_mathh.so"
            self.L = _Lib(_load(HERE / "_ref" / so), "ref_")
        else:
            raise ValueError(kind)
        fn = self.L.fn
        self._decim_taps = fn("decim_taps", _sz, _int, _int, _pf)
        self._dec_new = fn("decimator_new", _vp, _int, _f32p, _sz)
        self._dec_free = fn("decimator_free", None, _vp)
        self._dec_run = fn("decimator_run", _sz, _vp, _f32p, _sz)
        self._fir_new = fn("fir_new", _vp)
        self._fir_free = fn("fir_free", None, _vp)
        self._fir_design = fn("fir_design", None, _vp, C.c_float, C.c_float)
        self._demod_new = fn("demod_new", _vp)
        self._demod_free = fn("demod_free", None, _vp)
        self._demod_run = fn("demod_run", None, _vp, _f32p, _sz, _f32p)
        self._symex_new = fn("symex_new", _vp)
        self._symex_free = fn("symex_free", None, _vp)
        self._symex_rates = fn("symex_rates", None, _vp, _dbl, _dbl)
        self._symex_push = fn("symex_push", None, _vp, _f32p, _sz)
        self._symex_run = fn("symex_run", None, _vp)
        self._symex_get = fn("symex_get", _sz, _vp, _u8p, _sz)
        self._rtty_new = fn("rtty_new", _vp, _sz, C.c_float)
        self._rtty_free = fn("rtty_free", None, _vp)
        self._rtty_push = fn("rtty_push", None, _vp, _u8p, _sz)
        self._rtty_run = fn("rtty_run", _sz, _vp)
        self._rtty_get = fn("rtty_get", _sz, _vp, C.c_char_p, _sz)
        self._afc_new = fn("afc_new", _vp)
        self._afc_free = fn("afc_free", None, _vp)
        self._afc_set = fn("afc_set_spectrum", None, _vp, _f32p, _sz, _dbl)
        self._afc_process = fn("afc_process", _dbl, _vp)
        self._afc_reset = fn("afc_reset_correction", None, _vp, _dbl)
        self._afc_power = fn("afc_power", _sz, _vp, _pf)
        self._afc_get = fn("afc_get", None, _vp, *([C.POINTER(_dbl)] * 4), *([C.POINTER(_int)] * 2))
        self._crc = fn("crc16", None, C.c_char_p, _sz, C.c_char_p)
        self._extract = fn("extract_sentence", _int, C.c_char_p, _sz, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, _sz)
        if kind == "oracle":
            self._fir_set_n = fn("fir_set_input_size", None, _vp, _sz)
            self._fir_mode = fn("fir_design_mode", None, _vp, _int)
            self._fir_taps = fn("fir_taps", _sz, _vp, _pf)
            self._fir_run = fn("fir_run", _int, _vp, _f32p, _sz, _f32p)
            self._symex_held = fn("symex_held", _sz, _vp)
            self._symex_abs = fn("symex_abs_mode", None, _vp, _int)
            self._symex_flips = fn("symex_last_flips", _sz, _vp, _psz)
            self._plan = fn("decim_plan", _int, _int, C.POINTER(_int), C.POINTER(_int))
            self._fft = fn("fft_shifted", None, _f32p, _f32p, _sz)
            self._dc = fn("dc_remove", None, _f32p, _sz)
        else:
            self._fir_set_in = fn("fir_set_input", None, _vp, _f32p, _sz)
            self._fir_ntaps = fn("fir_ntaps", _sz, _vp)
            self._fir_run = fn("fir_run", None, _vp, _f32p, _sz, _f32p)

    # -- tables
    def decim_taps(self, total: int, ratio: int) -> np.ndarray:
        p = C.POINTER(C.c_float)()
        n = self._decim_taps(total, ratio, C.byref(p))
        return np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0, np.float32)

    def decim_plan(self, factor: int):
        r, nm = (_int * 2)(), (_int * 2)()
        ns = self._plan(factor, r, nm)
        return [(r[i], nm[i]) for i in range(max(ns, 0))] if ns >= 0 else None

    # -- stage objects
    def decimator(self, factor, taps):
        return _Decimator(self, factor, np.ascontiguousarray(taps, np.float32))

    def fir(self):
        return _Fir(self)

    def demod(self):
        return _Demod(self)

    def symex(self, fs, baud):
        return _Symex(self, fs, baud)

    def rtty(self, nbits, nstops):
        return _Rtty(self, nbits, nstops)

    def afc(self):
        return _Afc(self)

    def crc16(self, s: str) -> str:
        out = C.create_string_buffer(5)
        b = s.encode("latin-1")
        self._crc(b, len(b), out)
        return out.value.decode()

    def extract_sentence(self, stream: str):
        b = stream.encode("latin-1")
        cap = len(b) + 1
        bufs = [C.create_string_buffer(cap) for _ in range(4)]
        ok = self._extract(b, len(b), *bufs, cap)
        if not ok:
            return None
        call, data, crc, rest = (x.value.decode("latin-1") for x in bufs)
        return {"callsign": call, "data": data, "crc": crc, "stream": rest}

    def fft_shifted(self, x: np.ndarray) -> np.ndarray:
        xin = _cf(x).copy()
        out = np.empty_like(xin)
        self._fft(xin, out, len(xin) // 2)
        return out.view(np.complex64)

    def dc_remove(self, x: np.ndarray) -> np.ndarray:
        buf = _cf(x).copy()
        self._dc(buf, len(buf) // 2)
        return buf.view(np.complex64)


class _Decimator:
    def __init__(self, S, factor, taps):
        self.S, self.factor = S, factor
        self.h = S._dec_new(factor, taps, len(taps))

    def __call__(self, x: np.ndarray) -> np.ndarray:
        buf = _cf(x).copy()
        n = self.S._dec_run(self.h, buf, len(buf) // 2)
        if n == C.c_size_t(-1).value:
            raise ValueError("input shorter than history (undefined in the reference)")
        return buf.view(np.complex64)[:n].copy()

    def __del__(self):
        self.S._dec_free(self.h)


class _Fir:
    def __init__(self, S):
        self.S = S
        self.h = S._fir_new()
        self._dummy = None

    def set_input_size(self, n: int):
        if self.S.kind == "oracle":
            self.S._fir_set_n(self.h, n)
        else:  # the reference only records the size through setInput(ptr, n)
            self._dummy = np.zeros(2 * max(n, 1), np.float32)
            self.S._fir_set_in(self.h, self._dummy, n)

    def design_mode(self, float_trig: int):
        self.S._fir_mode(self.h, int(float_trig))

    def design(self, rel_width: float, trans: float):
        self.S._fir_design(self.h, rel_width, trans)

    def ntaps(self) -> int:
        if self.S.kind == "oracle":
            p = C.POINTER(C.c_float)()
            return self.S._fir_taps(self.h, C.byref(p))
        return self.S._fir_ntaps(self.h)

    def taps(self) -> np.ndarray:
        if self.S.kind == "oracle":
            p = C.POINTER(C.c_float)()
            n = self.S._fir_taps(self.h, C.byref(p))
            return np.ctypeslib.as_array(p, shape=(n,)).copy()
        # reference taps are private: recover them exactly as the impulse response on a fresh clone
        raise NotImplementedError("use impulse response")

    def __call__(self, x: np.ndarray) -> np.ndarray:
        xin = _cf(x).copy()
        out = np.zeros_like(xin)
        self.S._fir_run(self.h, xin, len(xin) // 2, out)
        return out.view(np.complex64)

    def __del__(self):
        self.S._fir_free(self.h)


class _Demod:
    def __init__(self, S):
        self.S = S
        self.h = S._demod_new()

    def __call__(self, x: np.ndarray) -> np.ndarray:
        xin = _cf(x).copy()
        out = np.zeros(len(xin) // 2, np.float32)
        self.S._demod_run(self.h, xin, len(out), out)
        return out

    def __del__(self):
        self.S._demod_free(self.h)


class _Symex:
    def __init__(self, S, fs, baud):
        self.S = S
        self.h = S._symex_new()
        S._symex_rates(self.h, fs, baud)

    def push(self, v):
        v = np.ascontiguousarray(v, np.float32)
        self.S._symex_push(self.h, v, len(v))

    def run(self) -> np.ndarray:
        self.S._symex_run(self.h)
        cap = 1 << 20
        out = np.zeros(cap, np.uint8)
        n = self.S._symex_get(self.h, out, cap)
        return out[:n].copy()

    def abs_mode(self, float_abs: int):
        self.S._symex_abs(self.h, int(float_abs))

    def held(self) -> int:
        return self.S._symex_held(self.h)

    def last_flips(self) -> np.ndarray:
        p = C.POINTER(C.c_size_t)()
        n = self.S._symex_flips(self.h, C.byref(p))
        return np.ctypeslib.as_array(p, shape=(n,)).copy().astype(np.int64) if n else np.zeros(0, np.int64)

    def __del__(self):
        self.S._symex_free(self.h)


class _Rtty:
    def __init__(self, S, nbits, nstops):
        self.S = S
        self.h = S._rtty_new(nbits, nstops)

    def push(self, bits):
        bits = np.ascontiguousarray(bits, np.uint8)
        self.S._rtty_push(self.h, bits, len(bits))

    def run(self) -> bytes:
        self.S._rtty_run(self.h)
        cap = 1 << 16
        out = C.create_string_buffer(cap)
        n = self.S._rtty_get(self.h, out, cap)
        return out.raw[:n]

    def __del__(self):
        self.S._rtty_free(self.h)


class _Afc:
    def __init__(self, S):
        self.S = S
        self.h = S._afc_new()

    def set_spectrum(self, spec: np.ndarray, rate: float):
        s = _cf(spec).copy()
        self.S._afc_set(self.h, s, len(s) // 2, rate)

    def process(self) -> float:
        return self.S._afc_process(self.h)

    def reset_correction(self, c: float):
        self.S._afc_reset(self.h, c)

    def power(self) -> np.ndarray:
        p = C.POINTER(C.c_float)()
        n = self.S._afc_power(self.h, C.byref(p))
        return np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0, np.float32)

    def state(self) -> dict:
        d = [_dbl() for _ in range(4)]
        i = [_int() for _ in range(2)]
        self.S._afc_get(self.h, *[C.byref(x) for x in d], *[C.byref(x) for x in i])
        return {"correction": d[0].value, "shift_hz": d[1].value, "noise_floor": d[2].value,
                "noise_var": d[3].value, "peak_l": i[0].value, "peak_r": i[1].value}

    def __del__(self):
        self.S._afc_free(self.h)


class Decoder:
    """Whole-chain decoder: `kind='oracle'` = orc_decoder_*, `kind='ref'` = ref_chain_* (reference stage
    classes sequenced by the harness; its FFT is borrowed from the oracle because FFTW is absent)."""

    def __init__(self, kind: str = "oracle", *, factor: int = 64, baud: float = 300, bits: int = 8, stops: float = 2,
                 lowpass_bw: float | None = None, lowpass_trans: float | None = None, dc_remove: bool = False,
                 with_fft: bool = True, mathh_context: int = 1, ungated: bool = False):
        self.kind = kind
        if kind == "oracle":
            self.L = _Lib(_load(HERE / "liboracle.so"), "orc_decoder_")
            self.h = self.L.fn("new", _vp)()
        else:
            so = "libhabdec_ref_mathh.so" if mathh_context else "libhabdec_ref.so"
            self.L = _Lib(_load(HERE / "_ref" / so), "ref_chain_")
            orc = _load(HERE / "liboracle.so")
            self._fft_keep = C.cast(orc.orc_fft_shifted, FFT_FN) if with_fft else C.cast(None, FFT_FN)
            self.h = self.L.fn("new", _vp, FFT_FN)(self._fft_keep)
        fn = self.L.fn
        self._free = fn("free", None, _vp)
        self._push = fn("push", None, _vp, _f32p, _sz, _dbl)
        self._process = fn("process", None, _vp)
        self._strs = {k: fn(k, _sz, _vp, _pc) for k in ("rtty_stream", "last_sentence", "sentence_log", "match_log", "chars_log")}
        self._arrs = {k: fn(k, _sz, _vp, _pf) for k in ("last_decimated", "last_filtered", "last_demod", "power")}
        self._bits = fn("last_bits", _sz, _vp, _pu8)
        self._afc = fn("afc", None, _vp, *([C.POINTER(_dbl)] * 4), *([C.POINTER(_int)] * 2))
        if kind == "oracle":
            fn("with_fft", None, _vp, _int)(self.h, 1 if with_fft else 0)
            fn("lookup_mode", None, _vp, _int)(self.h, int(mathh_context))
            fn("ungated", None, _vp, _int)(self.h, int(ungated))
            self._arrs["spectrum"] = fn("spectrum", _sz, _vp, _pf)
            self._arrs["fir_taps"] = fn("fir_taps", _sz, _vp, _pf)
            self._held = fn("symex_held", _sz, _vp)
            self._reset = fn("reset_correction", None, _vp, _dbl)
        assert fn("setup_factor", _int, _vp, _sz if kind == "oracle" else _int)(self.h, factor) == factor or factor == 1
        self._baud = fn("baud", None, _vp, _dbl)
        self._rtty = fn("rtty", None, _vp, _sz, C.c_float)
        self._baud(self.h, baud)
        self._rtty(self.h, bits, stops)
        self._dc_remove = fn("dc_remove", None, _vp, _int)
        self._dc_remove(self.h, int(dc_remove))
        self._lp_bw = fn("lowpass_bw", None, _vp, C.c_float)
        self._lp_trans = fn("lowpass_trans", None, _vp, C.c_float)
        # like websocketServer/main.cpp:550-551 these run before any input and only store the values
        if lowpass_bw is not None:
            self._lp_bw(self.h, lowpass_bw)
        if lowpass_trans is not None:
            self._lp_trans(self.h, lowpass_trans)

    def lowpass_bw(self, hz):
        self._lp_bw(self.h, hz)

    def lowpass_trans(self, t):
        self._lp_trans(self.h, t)

    def push(self, iq: np.ndarray, fs: float):
        x = _cf(iq)
        self._push(self.h, x, len(x) // 2, fs)

    def process(self):
        self._process(self.h)

    def __call__(self, iq, fs):
        self.push(iq, fs)
        self.process()

    def text(self, which: str) -> str:
        p = C.c_char_p()
        n = self._strs[which](self.h, C.byref(p))
        return C.string_at(p, n).decode("latin-1") if n else ""

    def sentences(self):
        return [s for s in self.text("sentence_log").split("\n") if s]

    def set_dc_remove(self, on: bool):
        self._dc_remove(self.h, int(on))

    def set_baud(self, baud: float):
        self._baud(self.h, baud)

    def set_rtty(self, bits: int, stops: float):
        self._rtty(self.h, bits, stops)

    def array(self, which: str) -> np.ndarray:
        p = C.POINTER(C.c_float)()
        n = self._arrs[which](self.h, C.byref(p))
        cplx = which in ("last_decimated", "last_filtered", "spectrum")
        if not n:
            return np.zeros(0, np.complex64 if cplx else np.float32)
        a = np.ctypeslib.as_array(p, shape=(n * (2 if cplx else 1),)).copy()
        return a.view(np.complex64) if cplx else a

    def bits(self) -> np.ndarray:
        p = C.POINTER(C.c_uint8)()
        n = self._bits(self.h, C.byref(p))
        return np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0, np.uint8)

    def afc(self) -> dict:
        d = [_dbl() for _ in range(4)]
        i = [_int() for _ in range(2)]
        self._afc(self.h, *[C.byref(x) for x in d], *[C.byref(x) for x in i])
        return {"correction": d[0].value, "shift_hz": d[1].value, "noise_floor": d[2].value,
                "noise_var": d[3].value, "peak_l": i[0].value, "peak_r": i[1].value}

    def symex_held(self) -> int:
        return self._held(self.h)

    def reset_correction(self, c: float):
        self._reset(self.h, c)

    def __del__(self):
        try:
            self._free(self.h)
        except Exception:
            pass


def have_ref() -> bool:
    return (HERE / "_ref" / "libhabdec_ref.so").exists()


class _BenchCfg(C.Structure):
    _fields_ = [("sampling_rate", _dbl), ("baud", _dbl), ("factor", _sz), ("bits", _sz), ("stops", C.c_float),
                ("lowpass_bw", C.c_float), ("lowpass_trans", C.c_float), ("mathh_context", _int), ("ungated", _int), ("with_fft", _int)]


def bench_run(streams, chunk_idx, chunk, repeats, *, fs, factor, baud, bits, stops, lowpass_bw=1500.0, lowpass_trans=0.025,
              mathh_context=1, ungated=False, with_fft=True):
    """Time `len(streams)` oracle decoders on as many threads (C++ std::thread inside liboracle.so, no GIL).
    streams: list of complex64 arrays; chunk_idx: which chunk of the array each call consumes.
    Returns (seconds, [sentence list of each stream's first pass])."""
    lib = _load(HERE / "liboracle.so")
    fn = lib.orc_bench_run
    fn.restype = _dbl
    fn.argtypes = [C.POINTER(_BenchCfg), C.POINTER(C.c_void_p), C.c_void_p, _sz, _sz, _int, _int, C.c_char_p, _sz]
    cfg = _BenchCfg(fs, baud, factor, bits, stops, lowpass_bw, lowpass_trans, int(mathh_context), int(ungated), int(with_fft))
    keep = [_cf(x) for x in streams]
    ptrs = (C.c_void_p * len(keep))(*[k.ctypes.data for k in keep])
    idx = np.ascontiguousarray(chunk_idx, dtype=np.uint32)
    cap = 1 << 22
    buf = C.create_string_buffer(cap)
    dt = fn(C.byref(cfg), ptrs, idx.ctypes.data, len(idx), chunk, repeats, len(keep), buf, cap)
    parts = buf.value.decode("latin-1").split("\x1e")[:len(keep)]
    out = []
    for p in parts:
        f = (p.split("\x1f") + ["", "0", "0"])[:4]
        out.append(BenchLog([x for x in f[0].split("\n") if x], f[1], int(f[2] or 0), int(f[3] or 0)))
    return dt, out


class BenchLog(list):
    """One stream's first-pass results of bench_run: the list itself is the sentence list; .chars is every printable character the
    decoder emitted, .bits the number of symbols it produced, .demod_hash every call's discriminator checksum folded into one word (the engine's
    hd_stream_demod_checksum_total folds the same way)."""
    def __init__(self, sentences, chars, bits, demod_hash=0):
        super().__init__(sentences)
        self.chars, self.bits, self.demod_hash = chars, bits, demod_hash


def atan2f_libm_mismatches(n=200000, seed=1):
    """How many of n argument pairs this box's libm atan2f answers differently from the oracle's fdlibm restatement (0 on glibc 2.35)."""
    lib = _load(HERE / "liboracle.so")
    fn = lib.orc_atan2f_libm_mismatches
    fn.restype = _sz
    fn.argtypes = [C.c_uint64, _sz]
    return int(fn(seed, n))
