"""One stream, synchronous delivery (what the Decoder facade does per push): wall time per 65536-sample push of an HBM-resident slab.
Usage: one_stream_latency.py [pushes=300]"""
import pathlib, sys, time
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
import habdec_amd
from habdec_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
fs, C = 2.048e6, 65536
frame = synth.rtty_bits(synth.make_sentence("ONE", "1,52.1,21.4,100"), 7, 2, 3, 3)
iq = synth.fsk_iq(np.concatenate([frame] * 2), fs, 50, sigma=0.05, seed=3, n_samples=8 * C)
slab = torch.from_numpy(np.ascontiguousarray(iq).view(np.float32).reshape(8, C, 2)).cuda()
for pipeline in (0,):
    eng = habdec_amd.Engine(n_streams=1, max_chunk=C, sampling_rate=fs, decimation=64, baud=50, rtty_bits=7, rtty_stops=2, pipeline=pipeline)
    for k in range(20):
        eng.process_device(slab[k % 8].data_ptr(), C, C)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        eng.process_device(slab[k % 8].data_ptr(), C, C)
    eng.flush()
    dt = (time.perf_counter() - t0) / n
    print(f"pipeline {pipeline}: {dt * 1e6:.1f} us per push, {C / dt / 1e9:.3f} GS/s, path {eng.timing()['path']}")
    eng.close()
