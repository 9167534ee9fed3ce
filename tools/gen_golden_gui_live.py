#!/usr/bin/env python3
"""tests/golden/gui_live.json: the websocket GUI payloads of ONE seeded stream, as the reference writes them.

The CPU oracle decodes the stream (tests regenerate the same IQ from the seed); its power spectrum, AFC read-outs and discriminator output go
through the REFERENCE's own SerializeSpectrum / SerializeDemodulation + CompressedVector, compiled as they are (oracle/_ref/libhabdec_ref_gui.so,
`make -C oracle ref`), for a few (zoom, resolution, type size) requests.  The fixture holds the request parameters, the AFC read-outs the header
carries and the payload bytes -- data only.  tests/test_gui_payload.py::test_live_engine_payloads_against_the_reference_bytes runs the GPU engine on
the same IQ and compares.
    make -C oracle ref && python tools/gen_golden_gui_live.py"""
import ctypes as C, json, sys
from pathlib import Path
import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from habdec_amd import synth          # (signal generator only: numpy, no GPU)
from oracle import pyoracle

R = C.CDLL(str(ROOT / "oracle" / "_ref" / "libhabdec_ref_gui.so"))
F32P = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
U8P = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
R.ref_spectrum_payload.restype = C.c_size_t
R.ref_spectrum_payload.argtypes = [F32P, C.c_size_t] + [C.c_double] * 4 + [C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, U8P, C.c_size_t, C.POINTER(C.c_size_t)]
R.ref_demod_payload.restype = C.c_size_t
R.ref_demod_payload.argtypes = [F32P, C.c_size_t, C.c_int, C.c_int, U8P, C.c_size_t, C.POINTER(C.c_size_t)]

FS, CHUNK, FACTOR, BAUD, SEED, SIGMA = 2.048e6, 65536, 64, 300, 5, 0.05
TEXT = ("GUI", "1,2,3")


def stream():
    x = synth.fsk_iq(synth.rtty_bits(synth.make_sentence(*TEXT) * 2, 8, 2, 4, 4), FS, BAUD, seed=SEED, sigma=SIGMA)
    return x[:len(x) // CHUNK * CHUNK]


def main():
    x = stream()
    o = pyoracle.Decoder("oracle", factor=FACTOR, baud=BAUD, bits=8, stops=2)
    for k in range(len(x) // CHUNK):
        o(x[k * CHUNK:(k + 1) * CHUNK], FS)
    power, demod, a = o.array("power"), o.array("last_demod"), o.afc()
    fsd = FS / FACTOR
    out = {"input": dict(fs=FS, chunk=CHUNK, factor=FACTOR, baud=BAUD, seed=SEED, sigma=SIGMA, text=list(TEXT), samples=int(len(x))),
           "afc": {k: (float(v) if isinstance(v, float) else int(v)) for k, v in a.items()}, "rate": fsd, "n_power": int(power.size), "n_demod": int(demod.size),
           "spectrum": [], "demod": []}
    buf = np.zeros(1 << 16, np.uint8)
    for zoom, res, ts in [(0.5, 512, 4), (0.0, 4096, 4), (0.75, 300, 2), (0.5, 512, 1), (0.9, 64, 2)]:
        sent = C.c_size_t(0)
        nb = R.ref_spectrum_payload(power, power.size, a["noise_floor"], a["noise_var"], fsd, a["shift_hz"], a["peak_l"], a["peak_r"], zoom, res, ts,
                                    buf, buf.size, C.byref(sent))
        out["spectrum"].append(dict(zoom=zoom, resolution=res, type_size=ts, bins_sent=sent.value, payload=bytes(buf[:nb]).hex()))
    for res, ts in [(256, 4), (256, 2), (100, 1), (5000, 4)]:
        sent = C.c_size_t(0)
        nb = R.ref_demod_payload(demod, demod.size, res, ts, buf, buf.size, C.byref(sent))
        out["demod"].append(dict(resolution=res, type_size=ts, values_sent=sent.value, payload=bytes(buf[:nb]).hex()))
    dst = ROOT / "tests" / "golden" / "gui_live.json"
    dst.write_text(json.dumps(out, indent=0) + "\n")
    print(dst, dst.stat().st_size, "bytes; afc", out["afc"])


if __name__ == "__main__":
    main()
