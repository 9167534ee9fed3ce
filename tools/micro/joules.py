"""Energy per step: board power (hwmon power1_average) and shader clock (freq1_input) sampled from a thread of THIS process every 50 ms while one engine
runs a loop of at least --seconds; J/step = mean power x time per step.  Variants as in ab_step.py: label[:ENV=val,...], label
`default` = the product library, anything else gpurun_in/variants/libhd_<label>.so; the pseudo-variant `idle` sleeps instead of launching.

    python3 tools/micro/joules.py [--workload cfg4] [--seconds 3] [--sync] default default+f:ARITH=1 timingexp+s1:HD_CU_EXP=1 timingexp+tails:HD_CU_EXP=2 idle
(`timingexp`: HD_BUILD_VARIANT=timingexp python3 -m habdec_amd.build -- the timing-experiment build; HD_CU_EXP does not exist in the product library.  ARITH=1: the fast mode.)
"""
import argparse, glob, os, sys, threading, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch, bench
from habdec_amd import capi, engine
from ab_step import load


def hwmon(dev):
    pr = torch.cuda.get_device_properties(dev)
    want = "%04x:%02x:%02x." % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", -1) & 0xFF, getattr(pr, "pci_device_id", 0))
    card = next((c for c in glob.glob("/sys/class/drm/card[0-9]*/device") if want in os.path.realpath(c)), None)
    return (glob.glob(f"{card}/hwmon/hwmon*") or [None])[0] if card else None


class Sampler(threading.Thread):
    def __init__(self, hw, period=0.05):
        super().__init__(daemon=True); self.hw, self.period, self.stop, self.w, self.mhz = hw, period, False, [], []

    def run(self):
        while not self.stop:
            try:
                pf = f"{self.hw}/power1_average" if os.path.exists(f"{self.hw}/power1_average") else f"{self.hw}/power1_input"
                self.w.append(int(open(pf).read()) / 1e6)
                self.mhz.append(int(open(f"{self.hw}/freq1_input").read()) / 1e6)
            except Exception:
                pass
            time.sleep(self.period)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg4"); ap.add_argument("--seconds", type=float, default=3.0); ap.add_argument("--sync", action="store_true")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    w = dict(bench.WORKLOADS[a.workload]); S = w["S"]; Cn = w["C"]
    dev = torch.device("cuda", 0)
    hw = hwmon(dev)
    ring, rc, _ = bench.generate_ring(torch, dev, w, S, 0, 1234)
    base = ring.data_ptr()
    torch.cuda.synchronize()
    print(f"workload {a.workload}: S={S} C={Cn}; hwmon {hw}; each variant runs >= {a.seconds} s; the first second is not sampled (the power average settles)")
    for v in a.variants:
        label, _, envs = v.partition(":")
        env = dict(kv.split("=", 1) for kv in envs.split(",") if kv)
        if label == "idle":
            smp = Sampler(hw); smp.start(); time.sleep(a.seconds); smp.stop = True; smp.join()
            print(f"{v:40s} idle: {np.mean(smp.w):7.1f} W  {np.mean(smp.mhz):6.0f} MHz")
            continue
        for k, x in env.items(): os.environ[k] = x
        capi._lib = load(label)
        eng = engine.Engine(n_streams=S, max_chunk=Cn, sampling_rate=w["fs"], decimation=w["D"], baud=w["baud"], rtty_bits=w["bits"], rtty_stops=w["stops"],
                            lowpass_bw_hz=w["lp_bw"], lowpass_trans=w["lp_trans"], ungated=w["ungated"], pipeline=0 if a.sync else 2,
                            **({"arith": int(env["ARITH"])} if "ARITH" in env else {}))
        eng.set_timing(0)
        i = 0
        t_end = time.perf_counter() + 1.0
        while time.perf_counter() < t_end:                  # settle
            for _ in range(50): eng.process_device(base + (i % rc) * S * Cn * 8, Cn, Cn); i += 1
        eng.flush(); torch.cuda.synchronize()
        smp = Sampler(hw); smp.start()
        n = 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < a.seconds:
            for _ in range(50): eng.process_device(base + (i % rc) * S * Cn * 8, Cn, Cn); i += 1; n += 1
        eng.flush(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        smp.stop = True; smp.join()
        for k in env: os.environ.pop(k, None)
        tm = eng.timing()
        eng.close()
        W, M = np.array(smp.w[2:]), np.array(smp.mhz[2:])
        print(f"{v:40s} {dt / n * 1e3:.4f} ms/step  {W.mean():7.1f} W (min {W.min():.0f} max {W.max():.0f})  {M.mean():6.0f} MHz  -> {W.mean() * dt / n * 1e3:7.2f} mJ/step   path {tm['path']} variant {tm['step_variant']}  ({n} steps)")


if __name__ == "__main__":
    main()
