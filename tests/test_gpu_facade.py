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
    info = [l for l in lines if l.startswith("INFO ")][-1]
    assert f"dec={1 << dec} " in info and "bins=4096" in info and f"samples={len(iq)} " in info
