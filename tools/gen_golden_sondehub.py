#!/usr/bin/env python3
"""Golden vectors for the sondehub upload body (tests/golden/sondehub.json), produced by the reference's own serializer
(common/json.hpp) and date library (common/date.h) compiled in oracle/_ref/libhabdec_ref_sondehub.so (oracle/Makefile `ref`).
Run where /root/reference exists:  python tools/gen_golden_sondehub.py"""
import ctypes as C
import json
import struct
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ref = C.CDLL(str(ROOT / "oracle" / "_ref" / "libhabdec_ref_sondehub.so"))
ref.ref_sondehub_body.restype = C.c_size_t
ref.ref_utc_iso.restype = C.c_size_t
ref.ref_utc_iso.argtypes = [C.c_int64, C.c_char_p, C.c_size_t]
ref.ref_json_number.restype = C.c_size_t
ref.ref_json_number.argtypes = [C.c_double, C.c_char_p, C.c_size_t]


def body(uploader, version, upload_time, recs):
    n = len(recs)
    arr = lambda key: (C.c_char_p * n)(*[r[key].encode("latin-1") for r in recs])
    frame = (C.c_int * n)(*[r["frame"] for r in recs])
    f = lambda key: (C.c_float * n)(*[r[key] for r in recs])
    buf = C.create_string_buffer(1 << 20)
    ref.ref_sondehub_body(uploader.encode(), version.encode(), upload_time.encode(), C.c_size_t(n), arr("payload_callsign"), arr("time_received"),
                          arr("datetime"), frame, f("lat"), f("lon"), f("alt"), buf, C.c_size_t(len(buf)))
    return buf.value.decode("latin-1")


def main():
    r = np.random.default_rng(20261002)
    buf = C.create_string_buffer(256)
    # 1. numbers: geographic floats, round numbers, ties of the 17-digit form, tiny and huge magnitudes, arbitrary float bit patterns
    vals = [0.0, -0.0, 1.0, -1.0, 100.0, 52.1234, -21.4321, 1e-5, 1e-4, 123456789.0, 1e15, 1e16, 1e20, 3.4e38, 1.17549435e-38, 1e-45,
            19.411880493164062, 64.21279907226562, 0.1, 0.5, 179.99999, -179.99999]
    vals = [float(np.float32(v)) for v in vals]
    vals += [float(x) for x in (r.uniform(-180, 180, 400).astype(np.float32))]
    bits = r.integers(0, 2 ** 32, 600, dtype=np.uint64).astype(np.uint32).view(np.float32)
    vals += [float(x) for x in bits if np.isfinite(x)]
    numbers = []
    for v in vals:
        ref.ref_json_number(v, buf, len(buf))
        numbers.append([struct.pack("<d", v).hex(), buf.value.decode()])
    # 2. clock strings
    clocks = []
    for ns in [0, 1, 999999999, 1000000000, 1614861296123456789, 1709164799999999999, 1709164800000000000, 4102444800000000000, 951782400000000000,
               1582934400500000000] + [int(x) for x in r.integers(0, 2 ** 62, 40)]:
        ref.ref_utc_iso(ns, buf, len(buf))
        clocks.append([ns, buf.value.decode()])
    # 3. whole bodies
    bodies = []
    for case in range(12):
        n = int(r.integers(1, 9))
        recs = []
        for i in range(n):
            recs.append({"payload_callsign": ["HAB1", "N0CALL-11", 'Q"uote', "back\\slash", "tab\there", "Zzz"][int(r.integers(0, 6))],
                         "time_received": "2021-03-04T12:34:56.123456789Z", "datetime": "2021-03-04T12:34:%02dZ" % int(r.integers(0, 60)),
                         "frame": int(r.integers(-5, 100000)), "lat": float(np.float32(r.uniform(-90, 90))), "lon": float(np.float32(r.uniform(-180, 180))),
                         "alt": float(np.float32(r.uniform(-100, 40000)))})
        up, ver, now = ["SP7HAB", "rx_1", "up\"loader"][case % 3], ["a1b2c3d4e5f6", "abc", "0123456789abcdef"][case % 3], "2021-03-04T12:35:00.000000001Z"
        bodies.append({"uploader": up, "version": ver, "upload_time": now, "records": recs, "body": body(up, ver, now, recs)})
    out = ROOT / "tests" / "golden" / "sondehub.json"
    out.write_text(json.dumps({"numbers": numbers, "clocks": clocks, "bodies": bodies}, indent=0) + "\n")
    print(out, len(numbers), len(clocks), len(bodies))


if __name__ == "__main__":
    main()
