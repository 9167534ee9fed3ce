"""GUI payloads: hd_host_spectrum_payload / hd_host_demod_payload against bytes produced by the
reference's own SerializeSpectrum / SerializeDemodulation + CompressedVector (tools/gen_golden_gui.py), and against that
build directly on random inputs where it exists."""
import ctypes as C
import json
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLD = json.loads((ROOT / "tests" / "golden" / "gui_payloads.json").read_text())
REF_SO = ROOT / "oracle" / "_ref" / "libhabdec_ref_gui.so"
META = (-91.25, 7.5, 32000.0, 512.5)


def spectrum(seed, n):
    r = np.random.default_rng(seed)
    x = (-90 + 8 * r.standard_normal(n)).astype(np.float32)
    x[n // 2 - n // 100] += 35; x[n // 2 + n // 97] += 33
    return x


def demod(seed, n):
    r = np.random.default_rng(seed)
    return (0.05 * np.sign(np.sin(np.arange(n) / 37.0)) + 0.01 * r.standard_normal(n)).astype(np.float32)


@pytest.fixture(scope="module")
def L():
    import habdec_amd
    return habdec_amd.lib()


def ours_spectrum(L, x, pl, pr, zoom, res, ts):
    buf = np.zeros(1 << 16, np.uint8); sent = C.c_size_t(0)
    nb = L.hd_host_spectrum_payload(x, x.size, *META, pl, pr, zoom, res, ts, buf, buf.size, C.byref(sent))
    return bytes(buf[:nb]), sent.value


def ours_demod(L, x, res, ts):
    buf = np.zeros(1 << 16, np.uint8); sent = C.c_size_t(0)
    nb = L.hd_host_demod_payload(x, x.size, res, ts, buf, buf.size, C.byref(sent))
    return bytes(buf[:nb]), sent.value


def test_spectrum_golden(L):
    for e in GOLD["spectrum"]:
        got, sent = ours_spectrum(L, spectrum(e["seed"], e["n"]), e["peak_left"], e["peak_right"], e["zoom"], e["resolution"], e["type_size"])
        assert sent == e["bins_sent"] and got.hex() == e["payload"], e
        assert len(got) == 52 + sent * e["type_size"] and int.from_bytes(got[:4], "little") == 52


def test_demod_golden(L):
    for e in GOLD["demod"]:
        got, sent = ours_demod(L, demod(e["seed"], e["n"]), e["resolution"], e["type_size"])
        assert sent == e["values_sent"] and got.hex() == e["payload"], e
        assert len(got) == 20 + sent * e["type_size"]


@pytest.mark.skipif(not REF_SO.exists(), reason="reference build only exists where /root/reference does")
def test_random_against_the_reference_build(L):
    R = C.CDLL(str(REF_SO))
    F32P = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS"); U8P = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
    R.ref_spectrum_payload.restype = C.c_size_t
    R.ref_spectrum_payload.argtypes = [F32P, C.c_size_t] + [C.c_double] * 4 + [C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, U8P, C.c_size_t, C.POINTER(C.c_size_t)]
    R.ref_demod_payload.restype = C.c_size_t
    R.ref_demod_payload.argtypes = [F32P, C.c_size_t, C.c_int, C.c_int, U8P, C.c_size_t, C.POINTER(C.c_size_t)]
    rng = np.random.default_rng(99)
    buf = np.zeros(1 << 16, np.uint8)
    for i in range(300):
        n = int(rng.choice([64, 1000, 4096]))
        x = spectrum(1000 + i, n)
        pl, pr = int(rng.integers(-n, n)), int(rng.integers(-n, n))
        zoom, res, ts = float(rng.uniform(-0.2, 1.2)), int(rng.integers(1, 2 * n)), int(rng.choice([1, 2, 4]))
        sent = C.c_size_t(0)
        nb = R.ref_spectrum_payload(x, n, *META, pl, pr, zoom, res, ts, buf, buf.size, C.byref(sent))
        assert ours_spectrum(L, x, pl, pr, zoom, res, ts) == (bytes(buf[:nb]), sent.value), (i, n, pl, pr, zoom, res, ts)
        y = demod(2000 + i, n)
        nb = R.ref_demod_payload(y, n, res, ts, buf, buf.size, C.byref(sent))
        assert ours_demod(L, y, res, ts) == (bytes(buf[:nb]), sent.value), (i, n, res, ts)


@pytest.mark.gpu
def test_payloads_from_a_live_engine():
    """End to end: the engine's getters feed the payload writers; what comes out parses back to the getter values."""
    import habdec_amd
    from habdec_amd import synth
    fs, C_ = 2.048e6, 65536
    x = synth.fsk_iq(synth.rtty_bits(synth.make_sentence("GUI", "1,2,3") * 2, 8, 2, 4, 4), fs, 300, seed=5, sigma=0.05)
    x = x[:len(x) // C_ * C_]
    eng = habdec_amd.Engine(n_streams=1, max_chunk=C_, sampling_rate=fs, decimation=64)
    for k in range(len(x) // C_):
        eng.process_host(x[None, k * C_:(k + 1) * C_])
    L = eng.L
    p, a = eng.power(0), eng.afc(0)
    got, sent = ours_spectrum(L, p, a["peak_l"], a["peak_r"], 0.5, 512, 4)
    assert sent == 512
    vals = np.frombuffer(got[52:], np.float32)
    mid = p[int(0.25 * 4096):int(0.75 * 4096)]
    assert np.array_equal(vals, mid[(np.arange(512, dtype=np.float32) / np.float32(512) * np.float32(mid.size)).astype(np.int64)])
    d = eng.demodulated(0)
    got, sent = ours_demod(L, d, 256, 2)
    assert sent == 256 and len(got) == 20 + 512
    # the 8- and 16-bit modes of CompressedVector (CompressedVector.cpp:75-118) on live data: what the web client decodes from the bytes
    # (min + q / (2^bits - 1) * (max - min), the header carries min and max) is the thinned getter values to within one quantisation step
    for ts in (1, 2):
        got, sent = ours_spectrum(L, p, a["peak_l"], a["peak_r"], 0.5, 512, ts)
        assert sent == 512 and len(got) == 52 + 512 * ts
        thin = mid[(np.arange(512, dtype=np.float32) / np.float32(512) * np.float32(mid.size)).astype(np.int64)]
        q = np.frombuffer(got[52:], np.uint8 if ts == 1 else np.uint16).astype(np.float64)
        lo, hi = float(thin.min()), float(thin.max())
        back = lo + q / (2 ** (8 * ts) - 1) * (hi - lo)
        assert np.max(np.abs(back - thin)) <= (hi - lo) / (2 ** (8 * ts) - 1) * 1.01
        got, sent = ours_demod(L, d, 256, ts)
        assert sent == 256 and len(got) == 20 + 256 * ts


LIVE = json.loads((ROOT / "tests" / "golden" / "gui_live.json").read_text())


def _spectrum_header(b):
    """SpectrumInfoHeader (NetTransport.h:29-46): 13 little-endian words."""
    i = np.frombuffer(b[:52], np.int32); f = np.frombuffer(b[:52], np.float32)
    return dict(header_size=int(i[0]), noise_floor=float(f[1]), noise_variance=float(f[2]), sampling_rate=float(f[3]), shift=float(f[4]),
                peak_left=int(i[5]), peak_right=int(i[6]), peak_left_valid=int(i[7]), peak_right_valid=int(i[8]), min=float(f[9]), max=float(f[10]),
                type_size=int(i[11]), size=int(i[12]))


@pytest.mark.gpu
def test_live_engine_payloads_against_the_reference_bytes():
    """SURVEY 8(f) row 1, pinned: tests/golden/gui_live.json holds the payload bytes the REFERENCE's serializers (NetTransport.h + CompressedVector.cpp,
    compiled as they are) produce from the CPU oracle's power spectrum / AFC read-outs / discriminator output of one seeded stream
    (tools/gen_golden_gui_live.py).  The GPU engine decodes the same stream and its getters go through hd_host_*_payload:
      * demodulation payloads must be the reference's bytes exactly (the discriminator output is bit-identical);
      * spectrum payloads: every integer header field equal (sizes, peaks and their validity, type size), float header fields and the
        32-bit values within the spectrum's parity tolerance (the transform is compared norm-wise everywhere: 1e-5 of the strongest bin,
        which is 5e-3 dB on bins within 40 dB of it), 8/16-bit values within one quantisation step of the reference's."""
    import habdec_amd
    from habdec_amd import synth
    inp = LIVE["input"]
    x = synth.fsk_iq(synth.rtty_bits(synth.make_sentence(*inp["text"]) * 2, 8, 2, 4, 4), inp["fs"], inp["baud"], seed=inp["seed"], sigma=inp["sigma"])
    x = x[:len(x) // inp["chunk"] * inp["chunk"]]
    assert len(x) == inp["samples"]
    eng = habdec_amd.Engine(n_streams=1, max_chunk=inp["chunk"], sampling_rate=inp["fs"], decimation=inp["factor"], baud=inp["baud"])
    for k in range(len(x) // inp["chunk"]):
        eng.process_host(x[None, k * inp["chunk"]:(k + 1) * inp["chunk"]])
    L = eng.L
    p, a, d = eng.power(0), eng.afc(0), eng.demodulated(0)
    assert p.size == LIVE["n_power"] and d.size == LIVE["n_demod"]
    assert (a["peak_l"], a["peak_r"]) == (LIVE["afc"]["peak_l"], LIVE["afc"]["peak_r"])
    for e in LIVE["demod"]:
        got, sent = ours_demod(L, d, e["resolution"], e["type_size"])
        assert sent == e["values_sent"] and got.hex() == e["payload"], ("demod payload", e["resolution"], e["type_size"])
    fsd = inp["fs"] / inp["factor"]
    for e in LIVE["spectrum"]:
        buf = np.zeros(1 << 16, np.uint8); sent = C.c_size_t(0)
        nb = L.hd_host_spectrum_payload(p, p.size, a["noise_floor"], a["noise_var"], fsd, a["shift_hz"], a["peak_l"], a["peak_r"], e["zoom"], e["resolution"],
                                        e["type_size"], buf, buf.size, C.byref(sent))
        got, want = bytes(buf[:nb]), bytes.fromhex(e["payload"])
        assert sent.value == e["bins_sent"] and len(got) == len(want)
        hg, hw = _spectrum_header(got), _spectrum_header(want)
        for k in ("header_size", "peak_left", "peak_right", "peak_left_valid", "peak_right_valid", "type_size", "size"):
            assert hg[k] == hw[k], (k, hg[k], hw[k])
        assert hg["sampling_rate"] == hw["sampling_rate"] and hg["shift"] == hw["shift"]
        assert hg["noise_floor"] == pytest.approx(hw["noise_floor"], rel=1e-5) and hg["noise_variance"] == pytest.approx(hw["noise_variance"], rel=1e-4)
        ts, n = e["type_size"], e["bins_sent"]
        span = hw["max"] - hw["min"]
        if ts == 4:
            vg, vw = np.frombuffer(got[52:], np.float32), np.frombuffer(want[52:], np.float32)
            strong = vw > vw.max() - 40.0
            assert np.max(np.abs(vg - vw)[strong]) <= 5e-3                     # dB, bins within 40 dB of the strongest
            assert np.max(np.abs(vg - vw)) <= 0.5                               # dB, the noise bins (a rounding difference in a near-empty bin is a large ratio)
        else:
            assert hg["min"] == pytest.approx(hw["min"], abs=0.5) and hg["max"] == pytest.approx(hw["max"], abs=5e-3)
            qg = np.frombuffer(got[52:], np.uint8 if ts == 1 else np.uint16).astype(np.float64)
            qw = np.frombuffer(want[52:], np.uint8 if ts == 1 else np.uint16).astype(np.float64)
            step = span / (2 ** (8 * ts) - 1)
            back_g = hg["min"] + qg / (2 ** (8 * ts) - 1) * (hg["max"] - hg["min"])
            back_w = hw["min"] + qw / (2 ** (8 * ts) - 1) * span
            strong = back_w > hw["max"] - 40.0
            assert np.max(np.abs(back_g - back_w)[strong]) <= step + 5e-3, (ts, n)
    eng.close()
