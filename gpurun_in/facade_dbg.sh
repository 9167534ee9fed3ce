#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_facade.py -m gpu -x -q 2>&1 | grep -B2 -A12 "AssertionError" | head -40
python3 - <<P
import subprocess, numpy as np, sys
sys.path.insert(0, ".")
from habdec_amd import synth
from habdec_amd.build import build_facade_demo
exe = build_facade_demo()
fs, baud = 2.048e6, 300
text = synth.make_sentence("FACADE", "1,52.1,21.4,100") * 2
iq = synth.fsk_iq_for_text(text, fs, baud, 8, 2, sigma=0.08, seed=11, idle_before=8, idle_after=14)
open("/tmp/iq.cf32", "wb").write(synth.to_iqfile_bytes(iq))
out = subprocess.run([str(exe), "/tmp/iq.cf32", str(fs), "6", str(baud), "8", "2", "1500.0"], capture_output=True, text=True)
for l in out.stdout.split("\n"):
    if "FACADE" in l: print(repr(l))
P
