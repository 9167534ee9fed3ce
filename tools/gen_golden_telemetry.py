#!/usr/bin/env python3
"""Generate tests/golden/telemetry.json from the REFERENCE build (oracle/_ref/libhabdec_ref_telemetry.so = the reference's
own sentence_parse.cpp + GpsDistance.cpp, compiled where they lie).  Run in the container that has /root/reference:
    make -C oracle ref && python tools/gen_golden_telemetry.py
The fixture is data only: inputs and the reference's outputs."""
import ctypes as C, json, random
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
L = C.CDLL(str(ROOT / "oracle" / "_ref" / "libhabdec_ref_telemetry.so"))
L.ref_parse_time.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float)]
L.ref_parse_gps_pos.argtypes = [C.c_char_p, C.POINTER(C.c_float)]
L.ref_parse_sentence.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)] + [C.POINTER(C.c_float)] * 3 + [C.c_char_p, C.c_size_t]
L.ref_gps_distance.argtypes = [C.c_double] * 6 + [C.POINTER(C.c_double)]


def f32hex(x: float) -> str:
    import struct
    return struct.pack("<f", x).hex()


def times():
    fixed = ["123456", "12:34:56", "12_34_56", "1234", "12:34", "12:34:56.7", "12:34:56.", "12:34:5", "1:34:56", "12:34:", "12x34y56",
             "235959", "00:00:00", "12:34:56.123456", "12::34", "", "12", "12:3", "12:34:56:78", "aa:bb:cc", "12.34.56", "12-34-56.5x",
             "123", "12345", "1234567", "12:34 56", " 12:34:56", "12:34:56 "]
    rng = random.Random(1)
    for _ in range(60):
        fixed.append("".join(rng.choice("0123456789:._ x") for _ in range(rng.randint(3, 10))))
    return fixed


def coords():
    fixed = ["52.1234", "21.34", "121.345", "-52.1234", "-121.345", "5205.5857", "02112.7309", "-5205.5857", "-02112.7309", "0.0", "0",
             "5.5", "1.25", "123456.7", "52", "5205", "52.", ".5", "9959.9999", "17959.9999", "0000.0000", "00000.0000", "+52.1234",
             "52,1234", "52.12.34", "5205.58.57", "abc.d", "12.ab", "ab12.5", "1234.x", "12345.x", "-", "-.5", "--52.1"]
    rng = random.Random(2)
    for _ in range(40):
        d = rng.choice([2, 3, 4, 5])
        fixed.append(("-" if rng.random() < 0.3 else "") + "".join(rng.choice("0123456789") for _ in range(d)) + "." + "".join(rng.choice("0123456789") for _ in range(rng.randint(1, 5))))
    return fixed


def sentences():
    fixed = ["CALLSIGN,1,12:00:00,52.1234,21.4321,1000", "$$CALLSIGN,1,12:00:00,52.1234,21.4321,1000", "$$$HAB,42,123456,5205.5857,02112.7309,31000,extra,fields",
             "HAB,42,12:34,-52.5,-21.25,12.5", "HAB,42,12:34:56.5,52.5,21.25,-3", "HAB,1,12:00:00,0.0,0.0,100", "HAB,1,12:00:00,0,21.5,100",
             "HAB,1,12:00:00,52.1,21.1", "HAB,x,12:00:00,52.1,21.1,100", "HAB,1,12:00:00,52.1,21.1,abc", "HAB,1,nope,52.1,21.1,100",
             "HAB,1,12:00:00,,21.1,100", "HAB,1,12:00:00,52.1,,100", "$$,1,12:00:00,52.1,21.1,100", "A$B,1,12:00:00,52.1,21.1,100", "A$,1,12:00:00,52.1,21.1,100",
             "a b-c,007,00:00:00,9959.9999,17959.9999,99999", ",1,12:00:00,52.1,21.1,100", "HAB, 5,12:00:00, 52.1,21.1, 100", "HAB,5.9,12:00:00,52.1,21.1,1e3",
             "HAB,1,12:00:00,52.1,21.1,100,", "R0S0001,1,52,21", "HAB,99999999999,12:00:00,52.1,21.1,100", "", "$$$$"]
    return fixed


def main():
    out = {"time": [], "gps_pos": [], "sentence": [], "distance": []}
    for t in times():
        h, m, s = C.c_int(0), C.c_int(0), C.c_float(0)
        rc = L.ref_parse_time(t.encode(), C.byref(h), C.byref(m), C.byref(s))
        out["time"].append({"in": t, "rc": rc, "h": h.value, "m": m.value, "s": f32hex(s.value)} if rc == 1 else {"in": t, "rc": rc})
    for c in coords():
        v = C.c_float(0)
        rc = L.ref_parse_gps_pos(c.encode(), C.byref(v))
        out["gps_pos"].append({"in": c, "rc": rc, "v": f32hex(v.value)} if rc == 1 else {"in": c, "rc": rc})
    for t in sentences():
        cs, dt = C.create_string_buffer(128), C.create_string_buffer(64)
        fr, la, lo, al = C.c_int(0), C.c_float(0), C.c_float(0), C.c_float(0)
        rc = L.ref_parse_sentence(t.encode(), cs, 128, C.byref(fr), C.byref(la), C.byref(lo), C.byref(al), dt, 64)
        e = {"in": t, "rc": rc}
        if rc == 1:
            e.update(callsign=cs.value.decode(), frame=fr.value, lat=f32hex(la.value), lon=f32hex(lo.value), alt=f32hex(al.value), time_of_day=dt.value.decode()[10:])
        out["sentence"].append(e)
    rng = random.Random(3)
    pts = [(52.0, 21.0, 100.0, 52.0, 21.0, 100.0), (52.0, 21.0, 100.0, 52.5, 21.5, 30000.0), (0, 0, 0, 0, 180, 0), (90, 0, 0, -90, 0, 0), (-33.9, 151.2, 5, 51.5, -0.1, 11000)]
    pts += [tuple([rng.uniform(-90, 90), rng.uniform(-180, 180), rng.uniform(0, 40000), rng.uniform(-90, 90), rng.uniform(-180, 180), rng.uniform(0, 40000)]) for _ in range(40)]
    for p in pts:
        o = (C.c_double * 5)()
        L.ref_gps_distance(*p, o)
        out["distance"].append({"in": [float(x).hex() for x in p], "out": [float(x).hex() for x in o]})
    dst = ROOT / "tests" / "golden" / "telemetry.json"
    dst.write_text(json.dumps(out, indent=0) + "\n")
    print(dst, {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
