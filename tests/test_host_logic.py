"""Host-side logic of the product (the hd_host_* C ABI inside libhabdec_amd.so: stage plan + coefficient tables, low-pass
design, RTTY framing, sentence extraction + CRC, text stage, AFC state machine, the discriminator's float-only atan2f)
against the CPU oracle.  Runs without a GPU."""
import ctypes as C

import numpy as np
import pytest

from habdec_amd import synth


@pytest.fixture(scope="module")
def L():
    from habdec_amd.build import build
    build()
    import habdec_amd
    return habdec_amd.lib()


@pytest.fixture(scope="module")
def O():
    from oracle import pyoracle
    return pyoracle.Stages("oracle")


def same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))


@pytest.mark.parametrize("factor", [1, 2, 4, 8, 16, 32, 64, 128, 256, 3, 512, 0])
def test_stage_plan_and_tables(L, O, factor):
    ratio, ntaps = (C.c_int * 2)(), (C.c_uint * 2)()
    ns = L.hd_host_decim_plan(factor, ratio, ntaps)
    plan = O.decim_plan(factor)
    if plan is None:
        assert ns == -1
        return
    assert ns == len(plan)
    for i, (r, name) in enumerate(plan):
        want = O.decim_taps(name, r)
        got = np.zeros(ntaps[i], np.float32)
        assert L.hd_host_decim_taps(factor, i, got, len(got)) == len(want)
        assert ratio[i] == r and same_bits(got, want)


@pytest.mark.parametrize("float_trig", [1, 0])
@pytest.mark.parametrize("rel,trans,batch", [(1500 / 32000, 0.025, 1024), (3000 / 156250, 0.025, 4096), (1500 / 512000, 0.025, 16384),
                                             (1500 / 39062.5, 4 / 4096, 4096), (1500 / 32000, 0.0, 1024), (1500 / 32000, 0.025, 100), (0.1, 0.9, 256)])
def test_lowpass_design_matches_oracle(L, O, rel, trans, batch, float_trig):
    f = O.fir()
    f.design_mode(float_trig)
    f.set_input_size(batch)
    f.design(np.float32(rel), np.float32(trans))
    want = f.taps()
    got = np.zeros(8192, np.float32)
    n = L.hd_host_lowpass_design(np.float32(rel), np.float32(trans), batch, 0, float_trig, got, len(got))
    assert n == len(want) and same_bits(got[:n], want)
    if n and trans:   # a second design of the same length is skipped (Q8), also when the cutoff changed
        assert L.hd_host_lowpass_design(np.float32(rel * 1.5), np.float32(trans), batch, n, float_trig, got, len(got)) == 0


@pytest.mark.parametrize("nbits,nstops", [(7, 1), (7, 2), (8, 1), (8, 2), (8, 1.5)])
def test_rtty_framing_matches_oracle(L, O, nbits, nstops):
    r = np.random.default_rng(int(nbits * 10 + nstops))
    bits = synth.rtty_bits("$$CALL,1,2,3*ABCD\n\x01\x7f~", nbits, int(np.ceil(nstops)), 5, 0)
    bits = np.concatenate([bits, r.integers(0, 2, 500).astype(np.uint8), bits[:-3], r.integers(0, 2, 2000).astype(np.uint8)])
    ro = O.rtty(nbits, nstops)
    h = L.hd_host_rtty_new(nbits, nstops)
    pos = 0
    for step in [7, 50, 13, 200, 1, 1, 1, 400, 3, 10000]:
        chunk = bits[pos:pos + step]
        pos += step
        if not len(chunk):
            break
        ro.push(chunk)
        want = ro.run()
        buf = C.create_string_buffer(1 << 14)
        n = L.hd_host_rtty_push_run(h, np.ascontiguousarray(chunk), len(chunk), buf, len(buf))
        assert buf.raw[:n] == want
    L.hd_host_rtty_free(h)


@pytest.mark.parametrize("nbits,nstops", [(7, 2), (8, 1), (8, 1.5)])
def test_rtty_framer_forgets_what_can_never_frame(L, O, nbits, nstops):
    """Long stretches that never frame (one level for thousands of bits, a start bit whose stop bit never comes) in front of, between and behind real
    characters, pushed one to a few bits at a time: the bytes are the reference's, although the framer keeps only the bits that can still start a frame
    (the reference keeps every unframed bit for ever and scans them all on every push)."""
    r = np.random.default_rng(int(nbits * 7 + nstops * 3))
    chars = synth.rtty_bits("$$IDLE,4,2*1F2E\n", nbits, int(np.ceil(nstops)), 0, 0)
    ones, zeros = np.ones(3000, np.uint8), np.zeros(2500, np.uint8)
    broken = np.tile(np.concatenate([[0], r.integers(0, 2, nbits), [0, 0]]).astype(np.uint8), 150)       # start bits whose stop bits are wrong
    bits = np.concatenate([ones, chars, zeros, chars[:-2], broken, ones[:777], chars, r.integers(0, 2, 1500).astype(np.uint8), zeros[:40], chars])
    ro = O.rtty(nbits, nstops)
    h = L.hd_host_rtty_new(nbits, nstops)
    pos, out_ref, out_got = 0, b"", b""
    while pos < len(bits):
        step = int(r.integers(1, 12))
        chunk = bits[pos:pos + step]
        pos += step
        ro.push(chunk)
        want = ro.run()
        buf = C.create_string_buffer(1 << 12)
        n = L.hd_host_rtty_push_run(h, np.ascontiguousarray(chunk), len(chunk), buf, len(buf))
        assert buf.raw[:n] == want, pos
        out_ref += bytes(want); out_got += buf.raw[:n]
    assert out_got == out_ref and out_got.count(b"IDLE") >= 2
    L.hd_host_rtty_free(h)


def test_rtty_framer_at_its_bound_of_kept_bits(L, O):
    """The one place the framer departs from the reference: beyond 2^18 unframed bits it forgets the oldest ones that can no longer
    frame in the current format.  At, just below and far beyond the bound the BYTES stay the reference's (the oracle keeps every bit and scans them all),
    and what is held stays bounded."""
    KEEP = 1 << 18
    nbits, nstops = 7, 2.0
    chars = synth.rtty_bits("$$EDGE,9,9*0A0B\n", nbits, 2, 0, 0)
    r = np.random.default_rng(18)
    broken = np.tile(np.concatenate([[0], r.integers(0, 2, nbits), [0, 0]]).astype(np.uint8), 3000)        # 30 000 bits of start bits whose stop bits never come
    pieces = [np.ones(KEEP - 7, np.uint8), chars,                   # idle up to just below the bound, then characters
              np.ones(12, np.uint8), chars,                         # ... across it
              np.zeros(KEEP + 4321, np.uint8), chars,               # a stuck-at-zero stretch longer than the bound (every bit a start bit without stop bits)
              broken, np.ones(KEEP, np.uint8), broken, chars[:-1],  # unframeable structure, idle, more of it, and a character that is one bit short
              np.ones(1, np.uint8), chars]
    ro = O.rtty(nbits, nstops)
    h = L.hd_host_rtty_new(nbits, nstops)
    got_all, held = b"", []
    for piece in pieces:
        ro.push(piece)
        want = ro.run()
        buf = C.create_string_buffer(1 << 12)
        n = L.hd_host_rtty_push_run(h, np.ascontiguousarray(piece), len(piece), buf, len(buf))
        assert buf.raw[:n] == want
        got_all += buf.raw[:n]
        held.append(L.hd_host_rtty_pending(h))
    assert got_all.count(b"EDGE") == 5
    assert max(held) <= KEEP + len(chars) + 16, held                # bounded: the kept bits + what can still frame
    assert sum(len(x) for x in pieces) > 3 * KEEP
    L.hd_host_rtty_free(h)


def _extract(L, s):
    b = s.encode("latin-1")
    cap = len(b) + 1
    bufs = [C.create_string_buffer(cap) for _ in range(4)]
    if not L.hd_host_extract_sentence(b, len(b), *bufs, cap):
        return None
    call, data, crc, rest = (x.value.decode("latin-1") for x in bufs)
    return {"callsign": call, "data": data, "crc": crc, "stream": rest}


def test_sentence_extraction_equals_std_regex(L, O):
    """The product spells the reference regex out as loops (std::regex is ~100x slower); the oracle keeps std::regex."""
    r = np.random.default_rng(0)
    good = synth.make_sentence("CALLSIGN", "1,12:00:00,52.1234,21.4321,1000")
    cases = [good, "xx" + good + "yy", good + good, "$$$A-B C,1,2$ABCD tail", "no star at all", "*", "$$A,b*12", "$$A,b*1234",
             "garbage$$X,1*0000\n$$Y,2*1111\n", "$$CALL,da\nta*12AB", "$,*AAAA", "$$A,,*AAAA*BBBB", "$$$$", "$$a,b$cdef*", "$ ,x*____",
             "$$A B-C_d,,,x*y*zzzz$$", "*$$A,b*cdefg", "$$A,b*cde$$A,b*cdef", "$-,-*----", "$$,x*abcd"]
    alphabet = list("$*,-_ abAB019\n")
    for _ in range(3000):
        cases.append("".join(r.choice(alphabet, size=r.integers(1, 70))))
    heavy = list("$$$***,,, ab1\n")
    for _ in range(2000):
        cases.append("".join(r.choice(heavy, size=r.integers(5, 40))))
    for s in cases:
        assert _extract(L, s) == O.extract_sentence(s), repr(s)


def test_crc16(L, O):
    out = C.create_string_buffer(5)
    kat = b"CALLSIGN,1,12:00:00,52.1234,21.4321,1000"
    L.hd_host_crc16(kat, len(kat), out)
    assert out.value == b"BF8A"
    r = np.random.default_rng(5)
    for _ in range(300):
        s = bytes(r.integers(1, 256, r.integers(0, 60)).astype(np.uint8))
        L.hd_host_crc16(s, len(s), out)
        assert out.value.decode() == O.crc16(s.decode("latin-1"))
        assert out.value.decode() == synth.crc16_ccitt(s.decode("latin-1")) or max(s) >= 0x80


def test_text_stage_matches_oracle_text_rules(L, O):
    """bits -> chars -> printable filter -> sentence loop -> >1000-char trim, against the oracle's primitives composed the
    way Decoder::process() composes them (Decoder.h:568-637)."""
    r = np.random.default_rng(9)
    texts = [synth.make_sentence("HAB", f"{i},52.{i},21.{i}") for i in range(30)] + ["$$BADCRC,1,2*0000\n", "noise" * 50, "$" * 30]
    bits = np.concatenate([np.concatenate([synth.rtty_bits(t, 8, 2, int(r.integers(0, 9)), 0), r.integers(0, 2, int(r.integers(0, 40))).astype(np.uint8)])
                           for t in texts] + [synth.rtty_bits("x" * 1200 + "$tail", 8, 2, 3, 3)])
    h = L.hd_host_text_new(8, 2.0)
    ro = O.rtty(8, 2)
    stream, last, ok, matches, chars = "", "", [], [], ""
    pos = 0
    while pos < len(bits):
        step = int(r.integers(1, 400))
        chunk = np.ascontiguousarray(bits[pos:pos + step])
        pos += step
        L.hd_host_text_push_bits(h, chunk, len(chunk))
        ro.push(chunk)
        raw = ro.run()
        if not raw:
            continue
        printable = "".join(chr(c) for c in raw if 0x20 <= c <= 0x7e or c == 0x0a)
        stream += printable
        chars += printable
        if len(stream) > 20:
            while True:
                m = O.extract_sentence(stream)
                if m is None:
                    break
                stream = m["stream"]
                last = m["callsign"] + "," + m["data"] + "*" + m["crc"]
                matches.append(last)
                if m["crc"] == O.crc16(m["callsign"] + "," + m["data"]):
                    ok.append(last)
        if len(stream) > 1000:
            k = stream.rfind("$")
            stream = "" if k < 0 else stream[k:]

    def get(which):
        buf = C.create_string_buffer(1 << 20)
        n = L.hd_host_text_get(h, which, buf, len(buf))
        return buf.raw[:n].decode("latin-1")
    assert get(0) == stream and get(1) == last
    assert [x for x in get(2).split("\n") if x] == ok and len(ok) >= 30
    assert [x for x in get(3).split("\n") if x] == matches and len(matches) > len(ok)
    assert get(4) == chars
    L.hd_host_text_free(h)


@pytest.mark.parametrize("seed", [21, 22, 23, 24])
def test_text_stage_on_noise_with_sentences_arriving_a_character_at_a_time(L, O, seed):
    """The text stage scans its stream only when the new characters can have completed a sentence (a terminator and four word characters, or the '*' the
    reference insists on): same text, sentences and logs as the reference's scan-after-every-push, on mostly-noise input rich in '$', ',' and '*', with
    sentences (good and bad CRC, '$'-terminated ones waiting for a '*') embedded, pushed a few bits at a time."""
    r = np.random.default_rng(seed)
    alphabet = np.frombuffer(b"$$$,,,**ABCDEFabcdef0123456789_- \n;:!?", dtype=np.uint8)
    pieces = []
    for i in range(60):
        noise = bytes(r.choice(alphabet, int(r.integers(0, 90)))).decode("latin-1")
        kind = int(r.integers(0, 5))
        if kind == 0: sent = synth.make_sentence("NOISY", f"{i},1.{i},2.{i}")
        elif kind == 1: sent = "$$BAD,%d,9*12AB\n" % i
        elif kind == 2: sent = "$$DOLLAR,%d,7$ABCD" % i            # terminator '$': matches only once a '*' is somewhere in the stream
        elif kind == 3: sent = "$,*,$" * int(r.integers(1, 4))
        else: sent = ""
        pieces.append(noise + sent)
    text = "".join(pieces)
    bits = synth.rtty_bits(text, 8, 2, 0, 0)
    h = L.hd_host_text_new(8, 2.0)
    ro = O.rtty(8, 2)
    stream, last, ok, matches, chars = "", "", [], [], ""
    pos = 0
    while pos < len(bits):
        step = int(r.integers(1, 30))
        chunk = np.ascontiguousarray(bits[pos:pos + step])
        pos += step
        L.hd_host_text_push_bits(h, chunk, len(chunk))
        ro.push(chunk)
        raw = ro.run()
        if not raw:
            continue
        printable = "".join(chr(c) for c in raw if 0x20 <= c <= 0x7e or c == 0x0a)
        stream += printable
        chars += printable
        if len(stream) > 20:
            while True:
                m = O.extract_sentence(stream)
                if m is None:
                    break
                stream = m["stream"]
                last = m["callsign"] + "," + m["data"] + "*" + m["crc"]
                matches.append(last)
                if m["crc"] == O.crc16(m["callsign"] + "," + m["data"]):
                    ok.append(last)
        if len(stream) > 1000:
            k = stream.rfind("$")
            stream = "" if k < 0 else stream[k:]

    def get(which):
        buf = C.create_string_buffer(1 << 20)
        n = L.hd_host_text_get(h, which, buf, len(buf))
        return buf.raw[:n].decode("latin-1")
    assert get(4) == chars and len(chars) > 2000
    assert [x for x in get(3).split("\n") if x] == matches and len(matches) >= 8
    assert [x for x in get(2).split("\n") if x] == ok and len(ok) >= 2
    assert get(0) == stream and get(1) == last
    L.hd_host_text_free(h)


def test_afc_state_machine_matches_oracle(L, O):
    """Feed the oracle AFC full spectra and the product tracker the reductions the GPU kernel produces from them."""
    r = np.random.default_rng(11)
    ao = O.afc()
    h = L.hd_host_afc_new()
    fsd, N = 32000.0, 4096
    have = False
    stats = None
    for call in range(60):
        spec = (r.standard_normal(N) + 1j * r.standard_normal(N)).astype(np.complex64)
        off = 30 if call < 25 else 95
        spec[2048 + off - 32] += 4000
        spec[2048 + off + 32] += 3500 if call % 7 else 0.0     # sometimes only one tone
        if call % 3 != 2:
            ao.set_spectrum(spec, fsd)
            have = True
        ao.process()
        if call % 3 != 2:
            P = ao.power()
            acc = np.cumsum(P.astype(np.float64))[-1]                       # sequential, like std::accumulate
            mean = acc / N
            var = np.cumsum((P.astype(np.float64) - mean) ** 2)[-1] / N
            sep = max(8, int(round(float(np.float32(500.0 / fsd)) * N)))
            p1 = int(np.argmax(P))
            p2, p2v = 0, P[0]
            for i in range(max(p1 - 2 * sep, 0), min(p1 + 2 * sep, N)):
                if P[i] > p2v and abs(i - p1) > sep // 2:
                    p2, p2v = i, P[i]
            a, b = (p1, p2) if p1 <= p2 else (p2, p1)
            stats = (1, a, b, float(P[a]), float(P[b]), mean, float(np.sqrt(var)))
        L.hd_host_afc_step(h, int(have), *stats, N, fsd)
        d = [C.c_double() for _ in range(4)]
        i2 = [C.c_int() for _ in range(2)]
        L.hd_host_afc_get(h, *[C.byref(x) for x in d], *[C.byref(x) for x in i2])
        want = ao.state()
        got = {"correction": d[0].value, "shift_hz": d[1].value, "noise_floor": d[2].value, "noise_var": d[3].value,
               "peak_l": i2[0].value, "peak_r": i2[1].value}
        assert got == want, (call, got, want)
        if call == 40:
            ao.reset_correction(want["correction"])
            L.hd_host_afc_reset(h, want["correction"], N, fsd)
    L.hd_host_afc_free(h)


def test_atan2_restatement_matches_libm(L, O):
    """The discriminator kernel's float-only atan2f (compiled here for the host from the same header) is bit-identical to
    the libm the reference calls through std::arg -- on ordinary products, on every special class and on random bit patterns."""
    r = np.random.default_rng(2)
    n = 400000
    iq = (r.standard_normal(n) * r.uniform(1e-3, 2.0, n) + 1j * r.standard_normal(n) * r.uniform(1e-3, 2.0, n)).astype(np.complex64)
    specials = np.array([0, -0.0, 1, -1, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1e-38, 3.4e38, 2.0 ** 25, 2.0 ** -30, 0.4375, 0.6875, 1.1875, 2.4375], np.float32)
    g = np.array(np.meshgrid(specials, specials)).reshape(2, -1)
    iq = np.concatenate([iq, (g[0] + 1j * g[1]).astype(np.complex64)])
    bits = r.integers(0, 2 ** 32, 2 * 200000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    iq = np.concatenate([iq, bits.view(np.complex64)])
    want = O.demod()(iq)                                     # libm atan2f via the oracle
    got = np.zeros(len(iq), np.float32)
    L.hd_host_discriminate(np.ascontiguousarray(iq).view(np.float32), len(iq), float(iq[0].real), float(iq[0].imag), got)
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan)
    assert same_bits(got[~nan], want[~nan])


def test_oracle_discriminator_is_the_fdlibm_restatement_not_the_box_libm():
    """The oracle evaluates arg() through its own fdlibm restatement (oracle/orc_atan2f.h), so bit-exact discriminator asserts compare
    like with like on any libm.  On glibc 2.35 (this image) the box's atan2f agrees with it on every probe; elsewhere a difference is
    RECORDED in the results as an expected failure (xfail), not passed over in silence: it means "another libm" -- the reference built on
    this host would differ from the golden vectors in the same way -- not "a wrong kernel"."""
    from oracle import pyoracle
    bad = pyoracle.atan2f_libm_mismatches(400000, seed=7)
    if bad:
        pytest.xfail(f"this box's libm atan2f differs from the fdlibm restatement on {bad} of 400289 probes: "
                     "discriminator parity is checked against the restatement (glibc 2.35 behaviour), not against this libm")


@pytest.mark.parametrize("factor,want", [(1, 1), (2, 68), (4, 140), (8, 280), (16, 544), (32, 1088), (64, 2176), (128, 4480), (256, 8960), (3, 0)])
def test_min_chunk_covers_every_stage_history(L, factor, want):
    """hd_min_chunk: the smallest push the engine accepts = a multiple of the factor that covers T1-1 input samples and (T2-1)*R1 of them
    for the second stage (shorter inputs are undefined behaviour in the reference's Decimator.h:140-143)."""
    L.hd_min_chunk.restype = C.c_uint32
    got = L.hd_min_chunk(factor)
    assert got == want
    if want:
        assert got % factor == 0
