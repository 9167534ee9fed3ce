#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own stage classes (oracle/_ref, compiled from /root/reference).

Run only where /root/reference exists (`make -C oracle ref` first).  The fixtures are data -- inputs and the outputs the
reference produced for them -- never reference source text.  Two lookup contexts are recorded: "mathh" (the
reference compiled with <math.h> in front: float sin/cos/abs overloads) and "cmath" (C++ headers only).
"""
import hashlib
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from habdec_amd import synth  # noqa: E402
from oracle import pyoracle  # noqa: E402

OUT = ROOT / "tests" / "golden"
TABLES = [(2, 2), (4, 4), (8, 8), (16, 8), (32, 16), (64, 32), (128, 32), (256, 64)]


def qiq(n, seed, amp=0.5):
    """int16-quantised complex noise: exactly representable, small on disk."""
    r = np.random.default_rng(seed)
    q = r.integers(-20000, 20000, (n, 2)).astype(np.int16)
    return q


def to_c64(q):
    return (q[:, 0].astype(np.float32) / 32768.0 + 1j * (q[:, 1].astype(np.float32) / 32768.0)).astype(np.complex64) * np.float32(1.0)


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def stages(ctx):
    R = pyoracle.Stages("ref_mathh" if ctx == "mathh" else "ref")
    d = {}
    meta = {}
    # decimator stages: two consecutive chunks (history carry), then a longer one (Q5 re-zero)
    for total, ratio in TABLES:
        taps = R.decim_taps(total, ratio)
        dec = R.decimator(ratio, taps)
        n = max(32 * ratio, 512)
        for k, nn in enumerate([n, n, 2 * n]):
            q = qiq(nn, 1000 + total * 10 + k)
            d[f"dec_{total}_{ratio}_in{k}"] = q
            d[f"dec_{total}_{ratio}_out{k}"] = dec(to_c64(q))
        d[f"taps_{total}_{ratio}"] = taps
    # low-pass designs (recovered exactly as impulse responses) and filtering
    designs = [(1500 / 32000, 0.025, 1024), (3000 / 156250, 0.025, 4096), (1500 / 512000, 0.025, 16384), (1500 / 39062.5, 4 / 4096, 4096)]
    for i, (rel, trans, m) in enumerate(designs):
        f = R.fir(); f.set_input_size(m); f.design(np.float32(rel), np.float32(trans))
        T = f.ntaps()
        imp = np.zeros(m, np.complex64); imp[0] = 1
        resp = np.concatenate([f(imp), f(np.zeros(m, np.complex64))])      # T may be m+1: the last tap shows up in the second call
        d[f"fir{i}_taps"] = resp[:T].real[::-1].copy()
        meta[f"fir{i}"] = [float(np.float32(rel)), float(np.float32(trans)), m, T]
    f = R.fir(); f.set_input_size(256); f.design(np.float32(1500 / 32000), np.float32(0.025))
    for k in range(3):
        q = qiq(256, 50 + k)
        d[f"firrun_in{k}"] = q
        d[f"firrun_out{k}"] = f(to_c64(q))
    # discriminator over three calls (first-sample rule, carry)
    dm = R.demod()
    for k, n in enumerate([256, 64, 512]):
        q = qiq(n, 70 + k)
        d[f"demod_in{k}"] = q
        d[f"demod_out{k}"] = dm(to_c64(q))
    # symbol extractor: a demodulated trace (produced by the reference FIR + discriminator on synthetic FSK) pushed in calls
    for name, fs, baud, nb, ns, text in [("sym300", 32000, 300, 8, 2, "$$ABC,12,3*1F2E\n"), ("sym50", 32000, 50, 7, 2, "$A"), ("sym600", 40000, 600, 7, 1, "$$XY,1*0000\n")]:
        bits = synth.rtty_bits(text, nb, ns, 10, 10)
        iq = synth.fsk_iq(bits, fs, baud, sigma=0.05, seed=5)
        n = (len(iq) // 256) * 256
        fir = R.fir(); fir.set_input_size(n); fir.design(np.float32(1500 / fs), np.float32(0.025))
        trace = R.demod()(fir(iq[:n]))
        trace = (np.round(trace * 8192) / 8192).astype(np.float32)       # quantised: compresses, still a valid input
        se = R.symex(fs, baud)
        outs = []
        for i in range(0, len(trace) - 1023, 1024):
            se.push(trace[i:i + 1024])
            outs.append(se.run())
        d[f"{name}_trace"] = (trace * 8192).astype(np.int16)
        d[f"{name}_bits"] = np.concatenate(outs) if outs else np.zeros(0, np.uint8)
        d[f"{name}_counts"] = np.array([len(o) for o in outs], np.int32)
        meta[name] = [fs, baud]
    np.savez_compressed(OUT / f"stages_{ctx}.npz", **d)
    (OUT / f"stages_{ctx}.json").write_text(json.dumps(meta, indent=1))


def text_and_afc():
    R = pyoracle.Stages("ref_mathh")
    r = np.random.default_rng(0)
    good = synth.make_sentence("CALLSIGN", "1,12:00:00,52.1234,21.4321,1000")
    cases = [good, "xx" + good + "yy", good + good, "$$$A-B C,1,2$ABCD tail", "no star at all", "*", "$$A,b*12", "$$A,b*1234",
             "garbage$$X,1*0000\n$$Y,2*1111\n", "$$CALL,da\nta*12AB", "$,*AAAA", "$$A,,*AAAA*BBBB"]
    alphabet = list("$*,-_ abAB019\n")
    for _ in range(300):
        cases.append("".join(r.choice(alphabet, size=r.integers(1, 60))))
    out = {"extract": [[c, R.extract_sentence(c)] for c in cases],
           "crc": [[s, R.crc16(s)] for s in ["CALLSIGN,1,12:00:00,52.1234,21.4321,1000", "", "A", "HAB1,7,52.1,21.4,999"] +
                   ["".join(r.choice(alphabet, size=20)) for _ in range(20)]]}
    # RTTY framing
    rt = {}
    for nb, ns in [(7, 1), (7, 2), (8, 1), (8, 2)]:
        bits = np.concatenate([synth.rtty_bits("$$CALL,1,2,3*ABCD\n\x01\x7f~", nb, ns, 5, 0), r.integers(0, 2, 300).astype(np.uint8)])
        rr = R.rtty(nb, ns)
        chunks, pos = [], 0
        for step in [7, 50, 13, 200, 1, 1, 1, 400, 10000]:
            c = bits[pos:pos + step]; pos += step
            if not len(c):
                break
            rr.push(c)
            chunks.append([c.tolist(), list(rr.run())])
        rt[f"{nb}N{ns}"] = chunks
    out["rtty"] = rt
    # AFC: deterministic integer-hash spectra (reproducible anywhere), expected state after every call
    a = R.afc()
    fsd, N = 32000.0, 4096
    states = []
    for call in range(30):
        spec = afc_spectrum(call, N)
        if call % 3 != 2:
            a.set_spectrum(spec, fsd)
        a.process()
        st = a.state()
        st["power_sha1"] = sha(a.power())
        states.append(st)
        if call == 20:
            a.reset_correction(st["correction"])
    out["afc"] = states
    (OUT / "text_afc.json").write_text(json.dumps(out))


def afc_spectrum(call, N=4096):
    i = np.arange(N, dtype=np.uint64)
    h = (i * np.uint64(2654435761) + np.uint64(call) * np.uint64(40503)) % np.uint64(1000)
    h2 = (i * np.uint64(40503) + np.uint64(call) * np.uint64(2654435761)) % np.uint64(1000)
    spec = ((h.astype(np.float32) - 500) / 250 + 1j * ((h2.astype(np.float32) - 500) / 250)).astype(np.complex64)
    off = 30 if call < 15 else 95
    spec[2048 + off - 32] += 4000
    spec[2048 + off + 32] += 3500
    return spec


def chain_small():
    """End to end on a small, fully stored input: 48 kS/s, /2, 600 baud 8N2 -- the reference stage classes sequenced like
    Decoder::process() (oracle/ref_harness.cpp).  Input stored as int16 I/Q."""
    fs, baud, C = 48000.0, 600, 4096
    text = synth.make_sentence("GOLD", "1,52.1,21.4") * 2
    iq = synth.fsk_iq_for_text(text, fs, baud, 8, 2, chunk=C, sigma=0.06, seed=77, idle_before=6, idle_after=12)
    q = np.stack([np.round(iq.real * 16384), np.round(iq.imag * 16384)], axis=1).astype(np.int16)
    x = (q[:, 0].astype(np.float32) / 16384 + 1j * (q[:, 1].astype(np.float32) / 16384)).astype(np.complex64)
    res = {}
    for ctx in (1, 0):
        d = pyoracle.Decoder("ref", factor=2, baud=baud, bits=8, stops=2, mathh_context=ctx)
        per_call = []
        for i in range(0, len(x), C):
            d(x[i:i + C], fs)
            per_call.append([sha(d.array("last_decimated")), sha(d.array("last_filtered")), sha(d.array("last_demod")), d.bits().tolist(), d.afc()])
        res["mathh" if ctx else "cmath"] = {"per_call": per_call, "sentences": d.sentences(), "chars": d.text("chars_log"),
                                            "rtty": d.text("rtty_stream"), "last": d.text("last_sentence")}
    np.savez_compressed(OUT / "chain_small_input.npz", iq_int16=q)
    (OUT / "chain_small.json").write_text(json.dumps({"fs": fs, "baud": baud, "chunk": C, "factor": 2, "scale": 16384, "expected": res}))


if __name__ == "__main__":
    OUT.mkdir(parents=True, exist_ok=True)
    pyoracle.build(ref=True)
    for ctx in ("mathh", "cmath"):
        stages(ctx)
    text_and_afc()
    chain_small()
    for p in sorted(OUT.iterdir()):
        print(f"{p.name:28s} {p.stat().st_size / 1024:8.1f} KiB")
