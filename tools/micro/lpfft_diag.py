"""Fast mode at /256 with the 4097-tap low-pass: the transform route against the direct sums and the oracle, stream by stream (how many streams' symbols differ, and by how much the filtered floats)."""
import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]; sys.path.insert(0, str(ROOT))
import torch, bench, habdec_amd
from oracle import pyoracle
w = dict(bench.WORKLOADS["cfg5"]); S = 64; C = w["C"]; fs = w["fs"]
ring, rc, _ = bench.generate_ring(torch, torch.device("cuda", 0), w, S, 0, seed=1234)
steps = int(os.environ.get("STEPS", "30"))
kw = dict(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"], lowpass_bw_hz=w["lp_bw"], lowpass_trans=w["lp_trans"], keep_filtered=True)
orcs = [pyoracle.Decoder("oracle", factor=w["D"], baud=w["baud"], bits=w["bits"], stops=w["stops"], lowpass_bw=w["lp_bw"], lowpass_trans=w["lp_trans"]) for _ in range(S)]
host = [ring[:, s].cpu().numpy().view(np.complex64).reshape(rc, C) for s in range(S)]
engs = {}
for name, env, arith in (("exact", {}, 0), ("fast_direct", {"HD_NO_LP_FFT": "1"}, 1), ("fast_fft", {}, 1)):
    for k, v in env.items(): os.environ[k] = v
    engs[name] = habdec_amd.Engine(arith=arith, **kw)
    for k in env: os.environ.pop(k)
worst = {n: 0.0 for n in engs}; bitdiff = {n: set() for n in engs}; firstdiff = {}
for k in range(steps):
    for e in engs.values(): e.process_device(ring[k % rc].data_ptr(), C, C)
    for s in range(S):
        o = orcs[s]; o(host[s][k % rc], fs)
        fo, do = o.array("last_filtered"), o.array("last_decimated")
        for n, e in engs.items():
            worst[n] = max(worst[n], bench.fir_normwise(e.filtered(s), fo, do))
            if not np.array_equal(e.bits(s), o.bits()):
                bitdiff[n].add(s); firstdiff.setdefault(n, (k, s, e.bits(s).tolist(), o.bits().tolist()))
print("lowpass_fft_calls", {n: e.timing()["lowpass_fft_calls"] for n, e in engs.items()})
print("worst filtered normwise", worst)
print("streams with a differing call", {n: len(v) for n, v in bitdiff.items()})
print("first difference", firstdiff)
