"""SURVEY 8(f) row 3, last hop: the sondehub upload batch (habdec_amd/csrc/host/sondehub.hpp through the hd_host_* C ABI).

Pinned byte for byte against the reference's own serializer and date library (common/json.hpp, common/date.h) compiled as they are in
oracle/_ref/libhabdec_ref_sondehub.so: tests/golden/sondehub.json was recorded from that build (tools/gen_golden_sondehub.py) and
travels; where the build exists the number formatting is also compared directly on a large random sample.  The eleven field
assignments of sondehub_uploader.cpp:55-65 are restated in the harness (by inspection); the HTTP PUT is out of scope."""
import ctypes as C
import json
import struct
import threading
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLD = json.loads((ROOT / "tests" / "golden" / "sondehub.json").read_text())


@pytest.fixture(scope="module")
def L():
    from habdec_amd.build import build
    build()
    import habdec_amd
    return habdec_amd.lib()


def num(L, v):
    buf = C.create_string_buffer(64)
    L.hd_host_json_number(v, buf, len(buf))
    return buf.value.decode()


def test_float_fields_print_like_the_reference_serializer(L):
    for hexbits, want in GOLD["numbers"]:
        v = struct.unpack("<d", bytes.fromhex(hexbits))[0]
        assert num(L, v) == want, (hexbits, v)


def test_clock_strings(L):
    buf = C.create_string_buffer(64)
    for ns, want in GOLD["clocks"]:
        L.hd_host_utc_iso(ns, buf, len(buf))
        assert buf.value.decode() == want, ns


def test_batch_bodies(L):
    for case in GOLD["bodies"]:
        h = L.hd_host_sondehub_new(case["uploader"].encode("latin-1"), case["version"].encode())
        for r in case["records"]:
            assert L.hd_host_sondehub_push(h, r["payload_callsign"].encode("latin-1"), r["time_received"].encode(), r["datetime"].encode(), r["frame"],
                                           r["lat"], r["lon"], r["alt"]) == 1
        assert L.hd_host_sondehub_size(h) == len(case["records"])
        # upload_time is the clock at take time: the golden bodies use 2021-03-04T12:35:00.000000001Z
        now_ns = 1614861300 * 10 ** 9 + 1
        n = C.c_size_t(0)
        need = L.hd_host_sondehub_take(h, now_ns, None, 0, C.byref(n))
        assert need == len(case["body"].encode("latin-1")) and n.value == len(case["records"])
        assert L.hd_host_sondehub_size(h) == 0                      # drained into the pending body ...
        buf = C.create_string_buffer(need + 1)
        assert L.hd_host_sondehub_take(h, now_ns + 5, buf, len(buf), C.byref(n)) == need     # ... which is handed out unchanged once the buffer fits
        assert buf.value.decode("latin-1") == case["body"]
        assert L.hd_host_sondehub_take(h, now_ns, buf, len(buf), C.byref(n)) == 0
        L.hd_host_sondehub_free(h)


def test_sentences_from_many_decoder_threads_fan_in(L):
    """What SentenceCallback does per CRC-valid sentence (main.cpp:286-306), from many threads at once: every parsable sentence becomes one
    record, the unparsable ones are dropped like the reference drops them, nothing is lost or duplicated."""
    h = L.hd_host_sondehub_new(b"FANIN", b"deadbeefcafe")
    T, N = 8, 500
    now = 1614861296 * 10 ** 9

    def work(t):
        for i in range(N):
            if i % 10 == 9:
                assert L.hd_host_sondehub_push_sentence(h, t, b"$$BAD", b"1,2", now) == 0                  # fewer than six fields
            else:
                data = f"{i},12:34:{i % 60:02d},52.{1000 + i},21.{4000 + t},{100 + i}".encode()
                assert L.hd_host_sondehub_push_sentence(h, t, f"$$T{t}".encode(), data, now + i) == 1
    th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert L.hd_host_sondehub_size(h) == T * N * 9 // 10
    n = C.c_size_t(0)
    need = L.hd_host_sondehub_take(h, now, None, 0, C.byref(n))
    buf = C.create_string_buffer(need + 1)
    L.hd_host_sondehub_take(h, now, buf, len(buf), C.byref(n))
    recs = json.loads(buf.value.decode())
    assert len(recs) == T * N * 9 // 10
    seen = {(r["payload_callsign"], r["frame"]) for r in recs}
    assert len(seen) == len(recs)
    r0 = [r for r in recs if r["payload_callsign"] == "T3" and r["frame"] == 7][0]
    assert list(r0) == sorted(r0) and r0["datetime"] == "2021-03-04T12:34:07Z" and r0["alt"] == 107 and r0["software_version"] == "deadbee"
    assert r0["time_received"].startswith("2021-03-04T12:34:56.0000000") and r0["uploader_callsign"] == "FANIN"
    L.hd_host_sondehub_free(h)


@pytest.mark.needs_ref
def test_number_formatting_against_the_reference_library_directly(L):
    ref = C.CDLL(str(ROOT / "oracle" / "_ref" / "libhabdec_ref_sondehub.so"))
    ref.ref_json_number.restype = C.c_size_t
    ref.ref_json_number.argtypes = [C.c_double, C.c_char_p, C.c_size_t]
    r = np.random.default_rng(5)
    vals = np.concatenate([r.uniform(-180, 180, 60000).astype(np.float32).astype(np.float64),
                           r.integers(0, 2 ** 32, 60000, dtype=np.uint64).astype(np.uint32).view(np.float32).astype(np.float64),
                           r.integers(0, 2 ** 63, 30000, dtype=np.uint64).view(np.float64)])
    vals = vals[np.isfinite(vals)]
    a, b = C.create_string_buffer(64), C.create_string_buffer(64)
    bad = 0
    for v in vals:
        ref.ref_json_number(float(v), a, 64)
        L.hd_host_json_number(float(v), b, 64)
        bad += a.value != b.value
    assert bad == 0


@pytest.mark.gpu
def test_decoded_sentences_of_a_batch_become_one_upload_body(L):
    """End of the path: GPU engine -> sentence callback -> sondehub batch.  The records of the body are exactly the CRC-valid sentences
    the CPU oracle decodes from the same streams (callsign, frame, position, altitude), in delivery order per stream."""
    import habdec_amd
    from habdec_amd import synth
    from oracle import pyoracle
    fs, S, Cn = 2.048e6, 8, 65536
    texts = [synth.make_sentence(f"SONDE{s}", f"{10 + s},12:{s:02d}:30,52.{1000 + s},21.{4000 + s},{1500 + 10 * s}") * 2 for s in range(S)]
    n = int(np.ceil((max(len(t) for t in texts) * 11 + 40) * fs / 300 / Cn)) + 1
    iq = np.stack([synth.fsk_iq(synth.rtty_bits(texts[s], 8, 2, 6 + s, 10), fs, 300, sigma=0.07, seed=300 + s, n_samples=n * Cn) for s in range(S)])
    h = L.hd_host_sondehub_new(b"GPUBOX", b"0123456789")
    eng = habdec_amd.Engine(n_streams=S, max_chunk=Cn, sampling_rate=fs, decimation=64, pipeline=True)
    now = 1614861296 * 10 ** 9
    eng.on_sentence(lambda s, call, data, crc: L.hd_host_sondehub_push_sentence(h, s, call.encode(), data.encode(), now))
    for k in range(n):
        eng.process_host(np.ascontiguousarray(iq[:, k * Cn:(k + 1) * Cn]))
    eng.flush()
    want = []
    for s in range(S):
        o = pyoracle.Decoder("oracle", factor=64)
        for k in range(n):
            o(iq[s, k * Cn:(k + 1) * Cn], fs)
        want += [x for x in o.sentences()]
    assert len(want) == 2 * S
    cnt = C.c_size_t(0)
    need = L.hd_host_sondehub_take(h, now + 1, None, 0, C.byref(cnt))
    buf = C.create_string_buffer(need + 1)
    L.hd_host_sondehub_take(h, now + 1, buf, len(buf), C.byref(cnt))
    recs = json.loads(buf.value.decode())
    assert len(recs) == len(want)
    got = sorted(f'{r["payload_callsign"]},{r["frame"]},{r["alt"]}' for r in recs)
    exp = sorted(f'{x.split(",")[0]},{x.split(",")[1]},{x.split(",")[5].split("*")[0]}' for x in want)
    assert got == exp
    assert all(r["datetime"].startswith("2021-03-04T12:0") and abs(r["lat"] - 52.1) < 0.1 for r in recs)
    L.hd_host_sondehub_free(h)
    eng.close()
