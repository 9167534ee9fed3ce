#!/usr/bin/env python3
"""BASELINE configs[0] and configs[1] as they are written -- ONE stream, synchronous delivery (what `habdec::Decoder<T>` does per push: the text of a push is
there when the call returns, main.cpp:234-245) -- with the one-core CPU oracle beside each line.

Per configuration two rates: `host_fed` -- every push hands over a HOST buffer (hd_process_host: H2D copy over PCIe included, the facade's path) -- and
`hbm_resident` (hd_process_device on a slab already on the GPU).  The oracle decodes the same pushes on one thread; characters, sentences and the last
discriminator output must be identical.  One JSON line per configuration (-> profiles/r06_single_stream.jsonl).

    python3 tools/single_stream.py [--pushes 300]
"""
import argparse, json, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
import habdec_amd
from habdec_amd import synth
from oracle import pyoracle

CONFIGS = [
    dict(name="configs[0]", desc="1 stream, 2.048 MS/s, /64, 300 baud 8N2 (the recorded-file configuration; synthetic IQ here)", fs=2.048e6, D=64, baud=300, bits=8, stops=2, lp_bw=1500.0),
    dict(name="configs[1]", desc="1 stream, 2.5 MS/s, dec=16 (/8 + /2), low-pass 3 kHz, 300 baud 8N2", fs=2.5e6, D=16, baud=300, bits=8, stops=2, lp_bw=3000.0),
]


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--pushes", type=int, default=300); ap.add_argument("--chunk", type=int, default=65536)
    a = ap.parse_args()
    C, N = a.chunk, a.pushes
    for cfg in CONFIGS:
        fs = cfg["fs"]
        text = synth.make_sentence("ONE", "1,52.1,21.4,100") * 3
        bits = synth.rtty_bits(text, cfg["bits"], cfg["stops"], 4, 4)
        nchunks = int(np.ceil(len(bits) * fs / cfg["baud"] / C)) + 1
        iq = synth.fsk_iq(bits, fs, cfg["baud"], sigma=0.05, seed=5, n_samples=nchunks * C).astype(np.complex64)
        pushes = [i % nchunks for i in range(N)]
        kw = dict(n_streams=1, max_chunk=C, sampling_rate=fs, decimation=cfg["D"], baud=cfg["baud"], rtty_bits=cfg["bits"], rtty_stops=cfg["stops"], lowpass_bw_hz=cfg["lp_bw"], pipeline=0)
        # --- host-fed (the facade's path): pageable host memory in, text out, per push
        eng = habdec_amd.Engine(**kw)
        host = [np.ascontiguousarray(iq[k * C:(k + 1) * C]) for k in range(nchunks)]
        for k in range(10): eng.process_host(host[k % nchunks][None, :])
        eng.close()
        eng = habdec_amd.Engine(**kw)
        t0 = time.perf_counter()
        for k in pushes: eng.process_host(host[k][None, :])
        eng.flush()
        dt_host = time.perf_counter() - t0
        chars_host, sent_host, demod_host = eng.take_chars(0), eng.take_sentences(0), eng.demodulated(0)
        path = eng.timing()["path"]
        eng.close()
        # --- host-fed from page-locked memory (what the Decoder facade's input queue lives in since round 6: hd_pinned_alloc): read in place by stage 1, no copy
        eng = habdec_amd.Engine(**kw)
        pin = eng.pinned_array((nchunks, C))
        pin[:] = iq.reshape(nchunks, C)
        for k in range(10): eng.process_host(pin[k % nchunks][None, :])
        eng.flush()
        c0 = eng.timing()["host_calls_in_place"]
        _ = eng.take_chars(0), eng.take_sentences(0)
        t0 = time.perf_counter()
        for k in pushes: eng.process_host(pin[k][None, :])
        eng.flush()
        dt_pin = time.perf_counter() - t0
        in_place = eng.timing()["host_calls_in_place"] - c0
        eng.close()
        # --- HBM-resident
        slab = torch.from_numpy(iq.view(np.float32).reshape(nchunks, C, 2)).cuda()
        eng = habdec_amd.Engine(**kw)
        t0 = time.perf_counter()
        for k in pushes: eng.process_device(slab[k].data_ptr(), C, C)
        eng.flush()
        dt_dev = time.perf_counter() - t0
        chars_dev, sent_dev = eng.take_chars(0), eng.take_sentences(0)
        eng.close()
        # --- the CPU oracle on one core, same pushes
        o = pyoracle.Decoder("oracle", factor=cfg["D"], baud=cfg["baud"], bits=cfg["bits"], stops=cfg["stops"], lowpass_bw=cfg["lp_bw"])
        t0 = time.perf_counter()
        for k in pushes: o(host[k], fs)
        dt_cpu = time.perf_counter() - t0
        same = (chars_host == o.text("chars_log") == chars_dev and sent_host == o.sentences() == sent_dev and
                np.array_equal(demod_host.view(np.uint32), o.array("last_demod").view(np.uint32)))
        print(json.dumps({"workload": f"{cfg['name']}: {cfg['desc']}", "streams": 1, "mode": "sync (text delivered by the call that pushed the samples)", "chunk_samples": C, "pushes": N,
                          "host_fed": {"value": round(N * C / dt_host / 1e6, 1), "unit": "MS/s", "us_per_push": round(dt_host / N * 1e6, 1), "note": "hd_process_host: pageable host buffer, H2D over PCIe inside the call"},
                          "host_fed_pinned": {"value": round(N * C / dt_pin / 1e6, 1), "unit": "MS/s", "us_per_push": round(dt_pin / N * 1e6, 1), "calls_read_in_place": int(in_place),
                                              "note": "hd_process_host on hd_pinned_alloc memory (the facade's input queue): stage 1 reads the buffer over PCIe, no staging copy"},
                          "hbm_resident": {"value": round(N * C / dt_dev / 1e6, 1), "unit": "MS/s", "us_per_push": round(dt_dev / N * 1e6, 1)},
                          "cpu_baseline": {"value": round(N * C / dt_cpu / 1e6, 1), "unit": "MS/s", "cores": 1, "kind": "port", "us_per_push": round(dt_cpu / N * 1e6, 1),
                                           "sample": f"the same {N} pushes through oracle/liboracle.so on one thread"},
                          "realtime_factor_host_fed": round(N * C / dt_host / fs, 1),
                          "launch_path": path, "gpu_matches_oracle": bool(same), "sentences": len(sent_host), "chars": len(chars_host)}), flush=True)
        del slab


if __name__ == "__main__":
    main()
