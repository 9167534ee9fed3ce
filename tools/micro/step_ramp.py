"""How long does the step loop take to reach the launch time it sustains?  The headline workload from a fresh engine (the ring's generation has kept the GPU busy
until a moment before), HIP events on every launch; prints the launch time in bins of ten launches.  Usage: step_ramp.py [launches=600]"""
import pathlib, sys
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
import habdec_amd
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
w = dict(bench.WORKLOADS["cfg4"]); S, fs, C = w["S"], w["fs"], w["C"]
ring, rc, _ = bench.generate_ring(torch, torch.device("cuda", 0), w, S, 0, seed=1234)
torch.cuda.synchronize()
eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"],
                        lowpass_bw_hz=w["lp_bw"], lowpass_trans=w["lp_trans"], pipeline=2)
eng.set_timing(1)
us, seen = [], 0
for k in range(n):
    eng.process_device(ring[k % rc].data_ptr(), C, C)
    t = eng.timing()
    if t["timed_calls"] != seen:
        seen = t["timed_calls"]; us.append(t["ms_front"] * 1e3)
eng.flush()
us = np.array(us)
print("launch us, medians of ten consecutive launches (every launch carries its timing events here: a few us more than in bench.py):")
print(" ".join(f"{np.median(us[i:i + 10]):.0f}" for i in range(0, len(us) - 9, 10)))
print(f"first 20: {np.mean(us[:20]):.1f}   launches 20-39: {np.mean(us[20:40]):.1f}   150-169: {np.mean(us[150:170]):.1f}   last 100: {np.mean(us[-100:]):.1f}")
