"""The source-compatible habdec::Decoder<float> facade (C++17 over the C ABI) driven like the reference's
DECODER_THREAD on a recorded-style cf32 IQ file; its callbacks/getters must equal the CPU oracle's."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

from habdec_amd import synth

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("fs,dec,baud,bits,stops,lowpass", [(2.048e6, 6, 300, 8, 2, 1500.0), (2.5e6, 4, 300, 8, 2, 3000.0), (2.048e6, 6, 50, 7, 2, 1500.0)])
def test_decoder_facade_on_iq_file(tmp_path, fs, dec, baud, bits, stops, lowpass):
    from habdec_amd.build import build_facade_demo
    from oracle import pyoracle
    exe = build_facade_demo()
    text = synth.make_sentence("FACADE", "1,52.1,21.4,100") * (2 if baud > 50 else 1) + ("" if baud > 50 else synth.make_sentence("F", "2"))
    iq = synth.fsk_iq_for_text(text, fs, baud, bits, stops, sigma=0.08, seed=11, idle_before=8, idle_after=14)
    # a file length that is NOT a multiple of 65536: the last read is short, like a real recording
    iq = iq[: len(iq) - 12345 - (len(iq) - 12345) % 64]
    path = tmp_path / "iq.cf32"
    path.write_bytes(synth.to_iqfile_bytes(iq))
    out = subprocess.run([str(exe), str(path), str(fs), str(dec), str(baud), str(bits), str(stops), str(lowpass)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    got_sent = [l[len("SENTENCE "):] for l in lines if l.startswith("SENTENCE ")]
    got_rtty = [l[len("RTTY "):] for l in lines if l.startswith("RTTY ")][-1]
    got_last = [l[len("LAST "):] for l in lines if l.startswith("LAST ")][-1]
    o = pyoracle.Decoder("oracle", factor=1 << dec, baud=baud, bits=bits, stops=stops, lowpass_bw=lowpass, lowpass_trans=0.025)
    C = 65536
    for i in range(0, len(iq), C):
        o(iq[i:i + C], fs)
    assert got_sent == o.sentences() and len(got_sent) >= 1
    assert got_rtty == o.text("rtty_stream").replace("\n", "\n").split("\n")[0] or got_rtty.startswith(o.text("rtty_stream").split("\n")[0])
    assert got_last == o.text("last_sentence")
    # stdout carries what the reference prints for every sentence its scan finds (Decoder.h:601 -> print_habhub_sentence.cpp:33-62): the line in
    # colour with its CRC verdict and the running OK / ERR tally
    matches = [m for m in o.text("match_log").split("\n") if m]
    ok = 0
    for m in matches:
        good = m in o.sentences()
        ok += good
        # (the line opens with "\x1b[2K\r"; text mode hands the carriage return over as a newline, so the part behind it is compared)
        want = ("\x1b[1;35m" if good else "\x1b[1;31m") + m + (" OK" if good else " ERR") + "\x1b[0m\t\tOK:" + str(ok) + "  ERR:"
        assert want in out.stdout and "\x1b[2K" in out.stdout, (m, good)
    info = [l for l in lines if l.startswith("INFO ")][-1]
    assert f"dec={1 << dec} " in info and "bins=4096" in info and f"samples={len(iq)} " in info
    # every sentence callback asked the decoder for getLastSentence()/getRTTY() from inside the callback (no deadlock, right answers)
    assert f"reentered={len(got_sent)}" in info
    # setupDecimationStagesBW: fs/10 -> /16, 2*fs -> /1, fs/1000 -> more than /256: refused (0)
    bw = [l for l in lines if l.startswith("BW ")][-1].split()
    assert bw[1:] == ["16", "1", "0"]


def test_facade_keeps_short_pushes_queued(tmp_path):
    """A push shorter than the first decimation stage's history (undefined behaviour in the reference, refused by the engine) must
    stay queued until the next push makes it long enough -- not be dropped: feeding the same file in odd-sized reads gives the same text."""
    from habdec_amd.build import build_facade_demo
    exe = build_facade_demo()
    fs, baud = 2.048e6, 300
    text = synth.make_sentence("SHORT", "7,52.1,21.4,100") * 2
    iq = synth.fsk_iq_for_text(text, fs, baud, 8, 2, sigma=0.08, seed=12, idle_before=8, idle_after=14)
    C = 65536
    iq = iq[: (len(iq) // C) * C + 128]              # the last read of the demo is 128 samples: 2 x 64, shorter than the 211-sample history
    path = tmp_path / "iq.cf32"
    path.write_bytes(synth.to_iqfile_bytes(iq))
    out = subprocess.run([str(exe), str(path), str(fs), "6", str(baud), "8", "2", "1500.0"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "habdec_amd:" not in out.stdout          # no engine error surfaced
    assert len([l for l in out.stdout.splitlines() if l.startswith("SENTENCE ")]) == 2
