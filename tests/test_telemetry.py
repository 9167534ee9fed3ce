"""Post-decode telemetry: hd_host_parse_* / hd_host_gps_distance against fixtures generated from the
reference's own sentence_parse.cpp + GpsDistance.cpp (tools/gen_golden_telemetry.py), and -- where that reference build exists
(this container, not the GPU box) -- against the reference directly on random inputs."""
import ctypes as C
import json
import random
import struct
import time
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLD = json.loads((ROOT / "tests" / "golden" / "telemetry.json").read_text())
REF_SO = ROOT / "oracle" / "_ref" / "libhabdec_ref_telemetry.so"


class Tele(C.Structure):
    _fields_ = [("callsign", C.c_char * 64), ("frame", C.c_int32), ("hour", C.c_int32), ("minute", C.c_int32),
                ("second", C.c_float), ("lat", C.c_float), ("lon", C.c_float), ("alt", C.c_float)]


def f32hex(x):
    return struct.pack("<f", x).hex()


@pytest.fixture(scope="module")
def L():
    import habdec_amd
    return habdec_amd.lib()


def parse_time(L, t):
    h, m, s = C.c_int(0), C.c_int(0), C.c_float(0)
    rc = L.hd_host_parse_time(t.encode(), C.byref(h), C.byref(m), C.byref(s))
    return (rc, h.value, m.value, f32hex(s.value)) if rc == 1 else (rc,)


def parse_pos(L, t):
    v = C.c_float(0)
    rc = L.hd_host_parse_gps_pos(t.encode(), C.byref(v))
    return (rc, f32hex(v.value)) if rc == 1 else (rc,)


def parse_sentence(L, t):
    o = Tele()
    rc = L.hd_host_parse_sentence(t.encode(), C.byref(o))
    if rc != 1:
        return (rc,)
    buf = C.create_string_buffer(64)
    L.hd_host_timestamp_from_hms(0, o.hour, o.minute, o.second, buf, 64)
    return (rc, o.callsign.decode(), o.frame, f32hex(o.lat), f32hex(o.lon), f32hex(o.alt), buf.value.decode()[10:])


def test_time_golden(L):
    for e in GOLD["time"]:
        want = (1, e["h"], e["m"], e["s"]) if e["rc"] == 1 else (e["rc"],)
        assert parse_time(L, e["in"]) == want, e["in"]


def test_gps_pos_golden(L):
    for e in GOLD["gps_pos"]:
        want = (1, e["v"]) if e["rc"] == 1 else (e["rc"],)
        assert parse_pos(L, e["in"]) == want, e["in"]


def test_sentence_golden(L):
    for e in GOLD["sentence"]:
        want = (1, e["callsign"], e["frame"], e["lat"], e["lon"], e["alt"], e["time_of_day"]) if e["rc"] == 1 else (e["rc"],)
        assert parse_sentence(L, e["in"]) == want, e["in"]


def test_distance_golden(L):
    for e in GOLD["distance"]:
        o = (C.c_double * 5)()
        L.hd_host_gps_distance(*[float.fromhex(x) for x in e["in"]], o)
        assert [float(x).hex() for x in o] == e["out"]


def test_timestamp_midnight_window(L):
    def ts(now, h, m, s):
        b = C.create_string_buffer(64)
        L.hd_host_timestamp_from_hms(now, h, m, s, b, 64)
        return b.value.decode()
    day = 1_700_000_000 // 86400 * 86400            # 2023-11-14 00:00:00 UTC
    assert ts(day + 12 * 3600, 11, 5, 7.0) == "2023-11-14T11:05:07Z"
    assert ts(day + 600, 23, 59, 58.0) == "2023-11-13T23:59:58Z"             # our clock is past midnight, the payload's is not
    assert ts(day + 86400 - 30, 0, 0, 1.5) == "2023-11-15T00:00:1.5Z"        # the other way round; "%g" seconds under setw(2)
    assert ts(day + 600, 0, 0, 0.0) == "2023-11-14T00:00:00Z"
    assert ts(951782400 + 3600, 1, 2, 3.0) == "2000-02-29T01:02:03Z"         # leap day


@pytest.mark.skipif(not REF_SO.exists(), reason="reference build only exists where /root/reference does")
def test_random_inputs_against_the_reference_build(L):
    R = C.CDLL(str(REF_SO))
    R.ref_parse_time.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float)]
    R.ref_parse_gps_pos.argtypes = [C.c_char_p, C.POINTER(C.c_float)]
    R.ref_timestamp_now.argtypes = [C.c_int, C.c_int, C.c_float, C.c_char_p, C.c_size_t]
    R.ref_gps_distance.argtypes = [C.c_double] * 6 + [C.POINTER(C.c_double)]
    rng = random.Random(7)
    for _ in range(4000):
        t = "".join(rng.choice("0123456789:._-x ") for _ in range(rng.randint(0, 11)))
        h, m, s = C.c_int(0), C.c_int(0), C.c_float(0)
        rc = R.ref_parse_time(t.encode(), C.byref(h), C.byref(m), C.byref(s))
        assert parse_time(L, t) == ((1, h.value, m.value, f32hex(s.value)) if rc == 1 else (rc,)), t
        c = "".join(rng.choice("0123456789.-") for _ in range(rng.randint(1, 11)))
        v = C.c_float(0)
        rc = R.ref_parse_gps_pos(c.encode(), C.byref(v))
        assert parse_pos(L, c) == ((1, f32hex(v.value)) if rc == 1 else (rc,)), c
    for _ in range(2000):
        p = [rng.uniform(-90, 90), rng.uniform(-180, 180), rng.uniform(0, 4e4), rng.uniform(-90, 90), rng.uniform(-180, 180), rng.uniform(0, 4e4)]
        a, b = (C.c_double * 5)(), (C.c_double * 5)()
        R.ref_gps_distance(*p, a); L.hd_host_gps_distance(*p, b)
        assert list(a) == list(b)
    # the wall-clock variant: same string as the reference when both look at the clock in the same second
    for h, m, s in [(12, 0, 0.0), (23, 59, 59.5), (0, 0, 1.0), (7, 8, 9.25)]:
        for _ in range(3):
            t0 = int(time.time())
            a = C.create_string_buffer(64); R.ref_timestamp_now(h, m, s, a, 64)
            b = C.create_string_buffer(64); L.hd_host_timestamp_from_hms(t0, h, m, s, b, 64)
            if int(time.time()) == t0:
                break
        assert a.value == b.value
