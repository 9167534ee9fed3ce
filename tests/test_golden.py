"""Pin the CPU oracle (and the product's host-side routines) against tests/golden/: fixtures that tools/gen_golden.py
recorded from the reference's own stage classes compiled out of /root/reference.  Runs anywhere (no reference, no GPU)."""
import ctypes as C
import hashlib
import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLD = ROOT / "tests" / "golden"
sys.path.insert(0, str(ROOT / "tools"))
TABLES = [(2, 2), (4, 4), (8, 8), (16, 8), (32, 16), (64, 32), (128, 32), (256, 64)]


def to_c64(q, scale=32768.0):
    return (q[:, 0].astype(np.float32) / np.float32(scale) + 1j * (q[:, 1].astype(np.float32) / np.float32(scale))).astype(np.complex64)


def same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def O():
    from oracle import pyoracle
    return pyoracle.Stages("oracle")


@pytest.fixture(scope="module", params=["mathh", "cmath"])
def gold(request):
    return request.param, np.load(GOLD / f"stages_{request.param}.npz"), json.loads((GOLD / f"stages_{request.param}.json").read_text())


def test_decimator_tables_and_outputs(O, gold):
    ctx, g, _ = gold
    for total, ratio in TABLES:
        assert same_bits(O.decim_taps(total, ratio), g[f"taps_{total}_{ratio}"])
        dec = O.decimator(ratio, g[f"taps_{total}_{ratio}"])
        for k in range(3):
            assert same_bits(dec(to_c64(g[f"dec_{total}_{ratio}_in{k}"])), g[f"dec_{total}_{ratio}_out{k}"]), (total, ratio, k)


def test_lowpass_design_and_filtering(O, gold):
    ctx, g, meta = gold
    for i in range(4):
        rel, trans, m, T = meta[f"fir{i}"]
        f = O.fir(); f.design_mode(1 if ctx == "mathh" else 0); f.set_input_size(int(m)); f.design(np.float32(rel), np.float32(trans))
        assert f.ntaps() == T and same_bits(f.taps(), g[f"fir{i}_taps"]), i
    f = O.fir(); f.design_mode(1 if ctx == "mathh" else 0); f.set_input_size(256); f.design(np.float32(1500 / 32000), np.float32(0.025))
    for k in range(3):
        assert same_bits(f(to_c64(g[f"firrun_in{k}"])), g[f"firrun_out{k}"]), k


def test_discriminator(O, gold):
    _, g, _ = gold
    dm = O.demod()
    for k in range(3):
        assert same_bits(dm(to_c64(g[f"demod_in{k}"])), g[f"demod_out{k}"]), k


@pytest.mark.parametrize("name", ["sym300", "sym50", "sym600"])
def test_symbol_extractor(O, gold, name):
    ctx, g, meta = gold
    fs, baud = meta[name]
    trace = g[f"{name}_trace"].astype(np.float32) / np.float32(8192)
    se = O.symex(fs, baud)
    se.abs_mode(1 if ctx == "mathh" else 0)
    outs = []
    for i in range(0, len(trace) - 1023, 1024):
        se.push(trace[i:i + 1024])
        outs.append(se.run())
    assert [len(o) for o in outs] == g[f"{name}_counts"].tolist()
    assert np.array_equal(np.concatenate(outs) if outs else np.zeros(0, np.uint8), g[f"{name}_bits"])
    assert g[f"{name}_bits"].size > 0


def test_text_fixtures_oracle_and_product(O):
    from habdec_amd.build import build
    build()
    import habdec_amd
    L = habdec_amd.lib()
    t = json.loads((GOLD / "text_afc.json").read_text())
    for s, want in t["extract"]:
        assert O.extract_sentence(s) == want, repr(s)
        b = s.encode("latin-1"); cap = len(b) + 1
        bufs = [C.create_string_buffer(cap) for _ in range(4)]
        ok = L.hd_host_extract_sentence(b, len(b), *bufs, cap)
        got = None if not ok else dict(zip(("callsign", "data", "crc", "stream"), (x.value.decode("latin-1") for x in bufs)))
        assert got == want, repr(s)
    out = C.create_string_buffer(5)
    for s, want in t["crc"]:
        assert O.crc16(s) == want
        L.hd_host_crc16(s.encode("latin-1"), len(s), out)
        assert out.value.decode() == want
    for key, chunks in t["rtty"].items():
        nb, ns = int(key[0]), int(key[2])
        ro = O.rtty(nb, ns)
        h = L.hd_host_rtty_new(nb, float(ns))
        for bits, want in chunks:
            bits = np.array(bits, np.uint8)
            ro.push(bits)
            assert list(ro.run()) == want
            buf = C.create_string_buffer(1 << 14)
            n = L.hd_host_rtty_push_run(h, bits, len(bits), buf, len(buf))
            assert list(buf.raw[:n]) == want
        L.hd_host_rtty_free(h)


def test_afc_states(O):
    from gen_golden import afc_spectrum
    t = json.loads((GOLD / "text_afc.json").read_text())
    a = O.afc()
    for call, want in enumerate(t["afc"]):
        if call % 3 != 2:
            a.set_spectrum(afc_spectrum(call), 32000.0)
        a.process()
        got = a.state()
        got["power_sha1"] = sha(a.power())
        assert got == want, call
        if call == 20:
            a.reset_correction(want["correction"])
    assert any(s["correction"] != 0 for s in t["afc"]) and any(s["peak_l"] > 0 for s in t["afc"])


@pytest.mark.parametrize("ctx", ["mathh", "cmath"])
def test_whole_chain_small(ctx):
    """End to end on a fully stored input: per-call hashes of the decimated / filtered / demodulated floats, the bits, the
    AFC scalars (the reference side used the oracle's DFT for its spectrum: FFTW is unpinned) and the decoded text."""
    from oracle import pyoracle
    meta = json.loads((GOLD / "chain_small.json").read_text())
    x = to_c64(np.load(GOLD / "chain_small_input.npz")["iq_int16"], meta["scale"])
    want = meta["expected"][ctx]
    d = pyoracle.Decoder("oracle", factor=meta["factor"], baud=meta["baud"], bits=8, stops=2, mathh_context=1 if ctx == "mathh" else 0)
    Cn = meta["chunk"]
    for k, i in enumerate(range(0, len(x), Cn)):
        d(x[i:i + Cn], meta["fs"])
        w = want["per_call"][k]
        assert [sha(d.array("last_decimated")), sha(d.array("last_filtered")), sha(d.array("last_demod"))] == w[:3], k
        assert d.bits().tolist() == w[3], k
        assert d.afc() == w[4], k
    assert d.sentences() == want["sentences"] and len(want["sentences"]) == 2
    assert d.text("chars_log") == want["chars"] and d.text("rtty_stream") == want["rtty"] and d.text("last_sentence") == want["last"]
