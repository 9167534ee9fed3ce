import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import habdec_amd
from oracle import pyoracle
from test_gpu_parity import make_streams, same_bits, C
S, fs = 3, 2.048e6
iq, _ = make_streams(S, fs, 300, 8, 2, seed0=900)
nch = iq.shape[1] // C
eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=64, pipeline=True)
orcs = [pyoracle.Decoder("oracle", factor=64) for _ in range(S)]
for k in range(nch):
    on = k % 3 == 1
    for s in range(S):
        eng.set_dc_remove(s, on); orcs[s].set_dc_remove(on)
    eng.process_host(np.ascontiguousarray(iq[:, k * C:(k + 1) * C]))
    eng.flush()
    for s in range(S):
        orcs[s](iq[s, k * C:(k + 1) * C], fs)
        g = eng.decimated(s); o = orcs[s].array("last_decimated")
        bad = np.nonzero(g.view(np.uint64) != o.view(np.uint64))[0]
        gd = eng.demodulated(s); od = orcs[s].array("last_demod")
        badd = np.nonzero(gd.view(np.uint32) != od.view(np.uint32))[0]
        if len(bad) or len(badd): print(k, s, "dc", on, "decimated bad:", len(bad), bad[:5], bad[-3:] if len(bad) else "", "demod bad:", len(badd), badd[:5])
