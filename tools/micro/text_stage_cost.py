"""Host text stage per call (hd_timing.host_text_us) at 1024 streams, synchronous delivery and batch mode.  Usage: text_stage_cost.py"""
import pathlib, sys
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
import habdec_amd
import bench

w = dict(bench.WORKLOADS["cfg4"])
S, fs, C = w["S"], w["fs"], w["C"]
ring, ring_chunks, _ = bench.generate_ring(torch, torch.device("cuda", 0), w, S, 0, seed=1)
torch.cuda.synchronize()
for pipeline in (0, 2):
    eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"],
                            lowpass_bw_hz=w["lp_bw"], lowpass_trans=w["lp_trans"], pipeline=pipeline)
    t = []
    for k in range(60):
        eng.process_device(ring[k % ring_chunks].data_ptr(), C, C)
        if k >= 10: t.append(eng.timing()["host_text_us"])
    eng.flush()
    print(f"pipeline {pipeline}: host_text_us median {np.median(t):.1f} min {min(t):.1f} max {max(t):.1f}")
    eng.close()
