"""Pin the CPU oracle (oracle/habdec_oracle.cpp) bit-for-bit against the reference's own stage classes
compiled from /root/reference (oracle/_ref/libhabdec_ref.so).  Runs only where that build exists."""
import numpy as np
import pytest

from habdec_amd import synth
from oracle import pyoracle

pytestmark = pytest.mark.needs_ref

TABLES = [(2, 2), (4, 4), (8, 8), (16, 8), (32, 16), (64, 32), (128, 32), (256, 64)]


@pytest.fixture(scope="module")
def O():
    return pyoracle.Stages("oracle")


@pytest.fixture(scope="module", params=[1, 0], ids=["mathh_ctx", "cmath_ctx"])
def ctx(request):
    """Lookup context the reference was compiled in: 1 = <math.h> force-included first (float sin/cos/abs
    overloads visible: the default), 0 = only C++ standard headers (double trig, integer abs).  DESIGN.md."""
    return request.param


@pytest.fixture(scope="module")
def R(ctx):
    return pyoracle.Stages("ref_mathh" if ctx else "ref")


def rnd_iq(n, seed, amp=0.5):
    r = np.random.default_rng(seed)
    return (amp * (r.standard_normal(n) + 1j * r.standard_normal(n))).astype(np.complex64)


def same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))


@pytest.mark.parametrize("total,ratio", TABLES)
def test_tap_tables_identical(O, R, total, ratio):
    assert same_bits(O.decim_taps(total, ratio), R.decim_taps(total, ratio))


@pytest.mark.parametrize("total,ratio", TABLES)
def test_decimator_stage_bit_exact_with_history(O, R, total, ratio):
    taps = R.decim_taps(total, ratio)
    do, dr = O.decimator(ratio, taps), R.decimator(ratio, taps)
    # three consecutive chunks: history carry, then a LARGER chunk (Q5: history re-zeroed), then smaller
    for k, n in enumerate([8192, 8192, 16384, 4096]):
        x = rnd_iq(n, 100 + k)
        assert same_bits(do(x), dr(x)), (total, ratio, k)


def test_decimator_short_input_history_quirk(O, R):
    """Q4: history is read back from the in-place buffer after the outputs overwrote its head."""
    taps = R.decim_taps(2, 2)  # 69 taps, /2
    do, dr = O.decimator(2, taps), R.decimator(2, taps)
    for k, n in enumerate([512, 100, 80, 70, 68, 512]):  # 68 = T-1 still defined; n - 68 < n/2 for n < 136
        x = rnd_iq(n, 7 + k)
        assert same_bits(do(x), dr(x)), (k, n)


@pytest.mark.parametrize("rel,trans,m", [
    (1500 / 32000, 0.025, 1024), (3000 / 156250, 0.025, 4096), (1500 / 512000, 0.025, 16384),
    (1500 / 39062.5, 4 / 4096, 4096), (1500 / 32000, 0.0, 1024), (1500 / 32000, 0.025, 100), (0.1, 0.9, 256)])
def test_fir_design_and_run_bit_exact(O, R, ctx, rel, trans, m):
    """float_trig=0: reference compiled with only C++ standard headers in front of habdec_windows.h (its
    unqualified sin/cos bind to the double overloads); float_trig=1: same sources with <math.h> force-included
    first (float overloads).  Both are 'the reference'; see DESIGN.md 'tap design hazard'."""
    float_trig = ctx
    fo, fr = O.fir(), R.fir()
    fo.design_mode(float_trig)
    rel, trans = np.float32(rel), np.float32(trans)
    for f in (fo, fr):
        f.set_input_size(m)
        f.design(rel, trans)
    assert fo.ntaps() == fr.ntaps()
    T = fo.ntaps()
    if T == 0:
        return
    # reference taps are private: recover them exactly as the impulse response of a fresh filter:
    # out[i] = sum_t buf[i+t]*tap[t], buf = [zeros(T-1) | x]; x = delta[0]  =>  out[i] = tap[T-1-i], i < T
    if m >= T:
        imp = np.zeros(m, np.complex64)
        imp[0] = 1.0
        fr2 = R.fir(); fr2.set_input_size(m); fr2.design(rel, trans)
        rec = fr2(imp)[:T].real[::-1].copy()
        assert same_bits(rec, fo.taps()), "tap design differs"
    for k in range(3):
        x = rnd_iq(m, 40 + k)
        assert same_bits(fo(x), fr(x)), k


def test_demod_bit_exact_including_first_sample_rule(O, R):
    do, dr = O.demod(), R.demod()
    for k, n in enumerate([1024, 256, 4096]):
        x = rnd_iq(n, 900 + k, amp=0.3)
        a, b = do(x), dr(x)
        assert same_bits(a, b)
        if k == 0:
            assert a[0] == 0.0  # first d[0] = arg(x0 * conj(x0)) (Q12)


def _demod_trace(fs, baud, text, nbits, nstops, seed, sigma=0.05):
    """Demodulated trace produced by the oracle chain front-end on synthetic FSK at the decimated rate."""
    bits = synth.rtty_bits(text, nbits, nstops, 12, 12)
    iq = synth.fsk_iq(bits, fs, baud, sigma=sigma, seed=seed)
    S = pyoracle.Stages("oracle")
    fir = S.fir(); n = (len(iq) // 256) * 256
    fir.set_input_size(n); fir.design(np.float32(1500 / fs), np.float32(0.025))
    return S.demod()(fir(iq[:n]))


@pytest.mark.parametrize("fs,baud,nbits,nstops", [(32000, 300, 8, 2), (156250, 300, 8, 2), (32000, 50, 7, 2), (40000, 600, 7, 1)])
def test_symbol_extractor_bits_identical(O, R, ctx, fs, baud, nbits, nstops):
    text = "$$ABC,12,3*1F2E\n" if baud > 50 else "$$A,1*"
    d = _demod_trace(fs, baud, text, nbits, nstops, seed=5)
    so, sr = O.symex(fs, baud), R.symex(fs, baud)
    so.abs_mode(ctx)
    step = 1024 if fs < 100000 else 4096
    allo, allr = [], []
    for i in range(0, len(d) - step + 1, step):
        so.push(d[i:i + step]); sr.push(d[i:i + step])
        bo, br = so.run(), sr.run()
        assert np.array_equal(bo, br), i
        allo.append(bo); allr.append(br)
    assert sum(len(b) for b in allo) > 20


def test_symbol_extractor_noise_and_vent(O, R, ctx):
    """Pure noise (many spurious flips) and a long flat run that trips the >30000-sample vent (Q14)."""
    r = np.random.default_rng(3)
    so, sr = O.symex(32000, 300), R.symex(32000, 300)
    so.abs_mode(ctx)
    for k in range(20):
        v = r.standard_normal(1024).astype(np.float32)
        so.push(v); sr.push(v)
        assert np.array_equal(so.run(), sr.run())
    flat = np.full(1024, 0.3, np.float32)
    for k in range(40):
        so.push(flat); sr.push(flat)
        assert np.array_equal(so.run(), sr.run())


@pytest.mark.parametrize("nbits,nstops", [(7, 1), (7, 2), (8, 1), (8, 2)])
def test_rtty_framing_identical(O, R, nbits, nstops):
    r = np.random.default_rng(nbits * 10 + nstops)
    text = "$$CALL,1,2,3*ABCD\n\x01\x7f~"
    bits = synth.rtty_bits(text, nbits, nstops, 5, 0)
    bits = np.concatenate([bits, r.integers(0, 2, 300).astype(np.uint8), bits[:-3]])  # truncated tail
    ro, rr = O.rtty(nbits, nstops), R.rtty(nbits, nstops)
    pos = 0
    for step in [7, 50, 13, 200, 1, 1, 1, 400, 10000]:
        chunk = bits[pos:pos + step]; pos += step
        if not len(chunk):
            break
        ro.push(chunk); rr.push(chunk)
        assert ro.run() == rr.run()


def test_crc_and_sentence_extraction_identical(O, R):
    assert O.crc16("CALLSIGN,1,12:00:00,52.1234,21.4321,1000") == "BF8A" == R.crc16("CALLSIGN,1,12:00:00,52.1234,21.4321,1000")
    r = np.random.default_rng(0)
    good = synth.make_sentence("CALLSIGN", "1,12:00:00,52.1234,21.4321,1000")
    cases = [good, "xx" + good + "yy", good + good, "$$$A-B C,1,2$ABCD tail", "no star at all", "*", "$$A,b*12",
             "$$A,b*1234", "garbage$$X,1*0000\n$$Y,2*1111\n", "$$CALL,da\nta*12AB", "$,*AAAA", "$$A,,*AAAA*BBBB"]
    alphabet = list("$*,-_ abAB019\n")
    for _ in range(400):
        cases.append("".join(r.choice(alphabet, size=r.integers(1, 60))))
    for s in cases:
        assert O.extract_sentence(s) == R.extract_sentence(s), repr(s)
    for _ in range(50):
        s = "".join(r.choice(alphabet, size=30))
        assert O.crc16(s) == R.crc16(s)


def test_afc_state_machine_identical(O, R):
    r = np.random.default_rng(11)
    ao, ar = O.afc(), R.afc()
    fsd, N = 32000.0, 4096
    for call in range(40):
        spec = (r.standard_normal(N) + 1j * r.standard_normal(N)).astype(np.complex64)
        off = 30 if call < 20 else 90  # move the tone pair so the correction latches and changes
        spec[2048 + off - 32] += 4000; spec[2048 + off + 32] += 3500
        if call % 3 != 2:  # AFC also runs on stale spectra (Q21)
            ao.set_spectrum(spec, fsd); ar.set_spectrum(spec, fsd)
        assert ao.process() == ar.process()
        assert same_bits(ao.power(), ar.power())
        assert ao.state() == ar.state()
        if call == 25:
            ao.reset_correction(ao.state()["correction"]); ar.reset_correction(ar.state()["correction"])
            assert ao.state() == ar.state()
    # degenerate spectra: empty -> no-op; NaN -> correction cleared
    bad = np.full(N, np.nan + 0j, np.complex64)
    ao.set_spectrum(bad, fsd); ar.set_spectrum(bad, fsd)
    assert ao.process() == ar.process() == 0.0


CHAIN_CASES = [
    dict(fs=2.048e6, factor=64, baud=300, bits=8, stops=2, text=synth.make_sentence("CALLSIGN", "1,12:00:00,52.1234,21.4321,1000") * 2),
    dict(fs=2.5e6, factor=16, baud=300, bits=8, stops=2, lowpass_bw=3000.0, text=synth.make_sentence("HAB1", "7,52.1,21.4,999")),
    dict(fs=2.048e6, factor=64, baud=50, bits=7, stops=2, text=synth.make_sentence("A", "1") * 3, f0=100.0),  # >20 chars needed before the scan runs
    dict(fs=2.048e6, factor=4, baud=300, bits=8, stops=2, text="$$A,1*0000\n"),  # above the 160 kHz gate: no decode
    dict(fs=2.048e6, factor=64, baud=300, bits=8, stops=2, dc_remove=True, text=synth.make_sentence("DCBLOCK", "1,2,52.0,21.0,100") * 2),
]


@pytest.mark.parametrize("case", CHAIN_CASES, ids=lambda c: f"D{c['factor']}_{c['baud']}bd_{c['bits']}N{c['stops']}" + ("_dc" if c.get("dc_remove") else ""))
def test_whole_chain_identical(case, ctx):
    """End to end: the oracle Decoder vs the reference stage classes sequenced after Decoder::process()."""
    fs, text = case["fs"], case["text"]
    iq = synth.fsk_iq_for_text(text, fs, case["baud"], case["bits"], case["stops"], sigma=0.08, seed=42, f0=case.get("f0", 0.0),
                               idle_before=8, idle_after=16)
    kw = dict(factor=case["factor"], baud=case["baud"], bits=case["bits"], stops=case["stops"],
              lowpass_bw=case.get("lowpass_bw"), dc_remove=case.get("dc_remove", False), mathh_context=ctx)
    do, dr = pyoracle.Decoder("oracle", **kw), pyoracle.Decoder("ref", **kw)
    C = 65536
    for i in range(0, len(iq), C):
        do(iq[i:i + C], fs); dr(iq[i:i + C], fs)
        for which in ("last_decimated", "last_filtered", "last_demod", "power"):
            assert same_bits(do.array(which), dr.array(which)), (which, i // C)
        assert np.array_equal(do.bits(), dr.bits())
        assert do.afc() == dr.afc()
    for which in ("rtty_stream", "last_sentence", "sentence_log", "match_log", "chars_log"):
        assert do.text(which) == dr.text(which), which
    if case["factor"] == 4:
        assert do.sentences() == [] and do.text("chars_log") == ""
    else:
        want = [s[2:] for s in text.strip().split("\n")]
        got = do.sentences()   # the scan only runs while >20 chars are buffered, so a short last line may wait
        assert got == want[:len(got)] and len(got) >= len(want) - 1 and len(got) >= 1


def test_whole_chain_with_control_plane_changes_mid_stream(ctx):
    """baud / framing / low-pass bandwidth and transition changed while samples flow: the oracle's Decoder must follow the
    reference's stage classes through every change (symbol-extractor restart, framer restart, the Q8 'same tap count -> no
    redesign' rule, and a transition change that does redesign)."""
    fs, C = 2.048e6, 65536
    text = synth.make_sentence("CTRL", "1,12:00:00,52.1234,21.4321,1000") * 4
    iq = synth.fsk_iq_for_text(text, fs, 300, 8, 2, sigma=0.08, seed=77, idle_before=8, idle_after=16)
    kw = dict(factor=64, baud=300, bits=8, stops=2, mathh_context=ctx)
    do, dr = pyoracle.Decoder("oracle", **kw), pyoracle.Decoder("ref", **kw)
    plan = {2: ("baud", 100.0), 3: ("baud", 300.0), 4: ("lp_bw", 2000.0), 5: ("rtty", (7, 1.0)), 6: ("rtty", (8, 2.0)), 7: ("lp_trans", 0.05),
            9: ("lp_bw", 1500.0), 10: ("lp_trans", 0.025)}
    for k, i in enumerate(range(0, len(iq), C)):
        if k in plan:
            what, v = plan[k]
            for d in (do, dr):
                if what == "baud": d.set_baud(v)
                elif what == "lp_bw": d.lowpass_bw(v)
                elif what == "lp_trans": d.lowpass_trans(v)
                else: d.set_rtty(int(v[0]), v[1])
        do(iq[i:i + C], fs); dr(iq[i:i + C], fs)
        for which in ("last_decimated", "last_filtered", "last_demod"):
            assert same_bits(do.array(which), dr.array(which)), (which, k)
        assert np.array_equal(do.bits(), dr.bits()), k
    for which in ("rtty_stream", "last_sentence", "sentence_log", "match_log", "chars_log"):
        assert do.text(which) == dr.text(which), which
    assert len(do.sentences()) >= 1
