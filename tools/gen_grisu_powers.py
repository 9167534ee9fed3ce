#!/usr/bin/env python3
"""habdec_amd/csrc/host/grisu_powers.inc: the cached powers of ten of Grisu2 (Loitsch, PLDI 2010) -- 10^k for k = -300, -292, ..., 324 as
normalised 64-bit significands (round to nearest) with their binary exponents, computed exactly with rationals."""
from fractions import Fraction
from pathlib import Path

rows = []
for k in range(-300, 325, 8):
    v = Fraction(10) ** k
    e = v.numerator.bit_length() - v.denominator.bit_length() - 64
    while True:
        f = v / (Fraction(2) ** e)
        if f >= 2 ** 64:
            e += 1
        elif f < 2 ** 63:
            e -= 1
        else:
            break
    fi = int(f)
    if f - fi >= Fraction(1, 2):
        fi += 1
    if fi == 2 ** 64:
        fi >>= 1
        e += 1
    rows.append((fi, e, k))
out = Path(__file__).resolve().parent.parent / "habdec_amd" / "csrc" / "host" / "grisu_powers.inc"
out.write_text(",\n".join("    {0x%016XULL, %d, %d}" % r for r in rows) + "\n")
print(out, len(rows))
