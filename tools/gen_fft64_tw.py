#!/usr/bin/env python3
"""Writes habdec_amd/csrc/kernels/fft64_tw.inc: the 64-point twiddles of the single-wave spectrum kernel, rounded once from double."""
import math, struct, pathlib
f = lambda x: repr(struct.unpack('f', struct.pack('f', x))[0]) + 'f'
cs = [math.cos(2 * math.pi * m / 64) for m in range(32)]
sn = [math.sin(2 * math.pi * m / 64) for m in range(32)]
out = pathlib.Path(__file__).resolve().parent.parent / "habdec_amd/csrc/kernels/fft64_tw.inc"
out.write_text("// cos(2 pi m / 64), sin(2 pi m / 64), m = 0..31, rounded once from double (generated: tools/gen_fft64_tw.py)\n"
               "{" + ", ".join(f(c) for c in cs) + "},\n{" + ", ".join(f(s) for s in sn) + "}\n")
