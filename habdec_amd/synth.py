"""Portable, seeded synthetic RTTY/2-FSK IQ generator (cf32, `IQSource_File` layout).

The reference ships no recording, so every test and benchmark input is generated here.  The stream
layout is what `code/IQSource/IQSource_File.h:156-157` reads: raw interleaved little-endian float32
I,Q, no header.  Signal model: continuous-phase 2-FSK, `shift` Hz between
space (bit 0, carrier - shift/2) and mark (bit 1, carrier + shift/2; `bit = mean > 0` in
`SymbolExtractor.h:149`), idle = mark, additive noise from an integer PRNG (Irwin-Hall sum of four
16-bit uniforms taken from the raw PCG64 stream, which numpy guarantees to be stable).
"""
from __future__ import annotations

import numpy as np

__all__ = ["crc16_ccitt", "make_sentence", "rtty_bits", "fsk_iq", "fsk_iq_for_text", "to_iqfile_bytes"]


def crc16_ccitt(text: str) -> str:
    """CRC16-CCITT-FALSE (init 0xFFFF, poly 0x1021) as 4 upper-case hex digits (reference CRC.cpp:21-47)."""
    crc = 0xFFFF
    for ch in text.encode("latin-1"):
        crc ^= ch << 8
        for _ in range(8):
            crc = ((crc << 1) ^ 0x1021) if (crc & 0x8000) else (crc << 1)
            crc &= 0xFFFFFFFF
    return "%04X" % (crc & 0xFFFF)


def make_sentence(callsign: str, fields: str) -> str:
    """`$$CALLSIGN,fields*CRC\\n` -- the habhub/UKHAS telemetry line format the text stage extracts."""
    body = f"{callsign},{fields}"
    return f"$${body}*{crc16_ccitt(body)}\n"


def rtty_bits(text: str, nbits: int = 8, nstops: int = 2, idle_before: int = 20, idle_after: int = 20) -> np.ndarray:
    """Asynchronous serial framing: start bit 0, `nbits` data bits LSB first, `nstops` stop bits 1."""
    out = [1] * idle_before
    for ch in text.encode("latin-1"):
        out.append(0)
        out.extend((ch >> k) & 1 for k in range(nbits))
        out.extend([1] * nstops)
    out.extend([1] * idle_after)
    return np.asarray(out, dtype=np.uint8)


def _noise(n: int, seed: int) -> np.ndarray:
    """2n unit-variance approximately-Gaussian float64 values from the raw PCG64 stream (portable)."""
    raw = np.random.PCG64(seed).random_raw(2 * n)  # one uint64 -> four 16-bit uniforms
    s = ((raw & 0xFFFF) + ((raw >> 16) & 0xFFFF) + ((raw >> 32) & 0xFFFF) + (raw >> 48)).astype(np.float64)
    # Irwin-Hall(4) on [0, 4*65535]: mean 2*65535, variance 4 * 65536^2 / 12
    return (s - 2.0 * 65535.0) / (65536.0 * np.sqrt(1.0 / 3.0))


def fsk_iq(bits: np.ndarray, fs: float, baud: float, *, shift: float = 500.0, f0: float = 0.0, amp: float = 0.5,
           sigma: float = 0.05, seed: int = 0, n_samples: int | None = None, phase0: float = 0.0) -> np.ndarray:
    """Continuous-phase 2-FSK rendering of `bits` at `baud` into complex64 samples at rate `fs`.

    Sample k carries bit floor(k * baud / fs); beyond the last bit the stream idles at mark.
    """
    bits = np.asarray(bits, dtype=np.uint8)
    total = int(np.ceil(len(bits) * fs / baud)) if n_samples is None else int(n_samples)
    k = np.arange(total, dtype=np.int64)
    idx = np.floor(k * (baud / fs) + 1e-9).astype(np.int64)
    b = np.where(idx < len(bits), bits[np.minimum(idx, len(bits) - 1)], 1).astype(np.float64)
    freq = f0 + (b * 2.0 - 1.0) * (shift / 2.0)
    phase = phase0 + 2.0 * np.pi * np.cumsum(freq / fs)
    phase = np.mod(phase, 2.0 * np.pi)
    x = amp * np.exp(1j * phase)
    if sigma:
        nz = _noise(total, seed)
        x = x + sigma * (nz[0::2] + 1j * nz[1::2])
    return x.astype(np.complex64)


def fsk_iq_for_text(text: str, fs: float, baud: float, nbits: int = 8, nstops: int = 2, *, chunk: int = 65536,
                    idle_before: int = 30, idle_after: int = 40, **kw) -> np.ndarray:
    """IQ for `text`, padded with mark idle to a whole number of `chunk`-sample pushes."""
    bits = rtty_bits(text, nbits, nstops, idle_before, idle_after)
    n = int(np.ceil(len(bits) * fs / baud))
    n = ((n + chunk - 1) // chunk) * chunk
    return fsk_iq(bits, fs, baud, n_samples=n, **kw)


def to_iqfile_bytes(iq: np.ndarray) -> bytes:
    """Serialise to the raw cf32 file layout `IQSource_File` replays."""
    return np.ascontiguousarray(iq, dtype=np.complex64).tobytes()
