#!/usr/bin/env python3
"""Generate tests/golden/gui_payloads.json from the REFERENCE build (oracle/_ref/libhabdec_ref_gui.so: the reference's
NetTransport.h + CompressedVector.cpp compiled as they are; zoom/ShrinkVector restated in the harness, see its header).
    make -C oracle ref && python tools/gen_golden_gui.py
Fixture = inputs (seeded generators, parameters) and the reference's payload bytes (hex)."""
import ctypes as C, json
from pathlib import Path
import numpy as np

ROOT = Path(__file__).resolve().parent.parent
R = C.CDLL(str(ROOT / "oracle" / "_ref" / "libhabdec_ref_gui.so"))
F32P = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
U8P = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
R.ref_spectrum_payload.restype = C.c_size_t
R.ref_spectrum_payload.argtypes = [F32P, C.c_size_t] + [C.c_double] * 4 + [C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, U8P, C.c_size_t, C.POINTER(C.c_size_t)]
R.ref_demod_payload.restype = C.c_size_t
R.ref_demod_payload.argtypes = [F32P, C.c_size_t, C.c_int, C.c_int, U8P, C.c_size_t, C.POINTER(C.c_size_t)]


def spectrum(seed, n):
    r = np.random.default_rng(seed)
    x = (-90 + 8 * r.standard_normal(n)).astype(np.float32)
    x[n // 2 - n // 100] += 35; x[n // 2 + n // 97] += 33
    return x


def main():
    out = {"spectrum": [], "demod": []}
    buf = np.zeros(1 << 16, np.uint8)
    for seed, n, pl, pr, zoom, res, ts in [(1, 4096, 2008, 2090, 0.5, 512, 1), (2, 4096, 2008, -2090, 0.0, 4096, 2), (3, 4096, -2008, 2090, 0.9, 100, 4),
                                           (4, 4096, 100, 4000, 0.75, 256, 1), (5, 4096, 2048, 2049, 1.5, 8000, 2), (6, 1000, 400, 600, 0.3, 333, 1),
                                           (7, 4096, 0, 0, 0.25, 1, 2), (8, 64, 30, 34, 0.5, 7, 4)]:
        x = spectrum(seed, n)
        sent = C.c_size_t(0)
        nb = R.ref_spectrum_payload(x, n, -91.25, 7.5, 32000.0, 512.5, pl, pr, zoom, res, ts, buf, buf.size, C.byref(sent))
        out["spectrum"].append(dict(seed=seed, n=n, peak_left=pl, peak_right=pr, zoom=zoom, resolution=res, type_size=ts, bins_sent=sent.value,
                                    payload=bytes(buf[:nb]).hex()))
    for seed, n, res, ts in [(11, 1024, 300, 1), (12, 5000, 1000, 2), (13, 256, 256, 4), (14, 777, 100, 1), (15, 2048, 5000, 2)]:
        r = np.random.default_rng(seed)
        x = (0.05 * np.sign(np.sin(np.arange(n) / 37.0)) + 0.01 * r.standard_normal(n)).astype(np.float32)
        sent = C.c_size_t(0)
        nb = R.ref_demod_payload(x, n, res, ts, buf, buf.size, C.byref(sent))
        out["demod"].append(dict(seed=seed, n=n, resolution=res, type_size=ts, values_sent=sent.value, payload=bytes(buf[:nb]).hex()))
    dst = ROOT / "tests" / "golden" / "gui_payloads.json"
    dst.write_text(json.dumps(out, indent=0) + "\n")
    print(dst, {k: len(v) for k, v in out.items()}, dst.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
