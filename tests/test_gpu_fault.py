"""The give-up path of the loader / consumer ring kernels (k_stage1_cu at /8 and /4; the /32 stages' worker waves share nothing and wait for nothing
but hardware counters), end to end, on a /16 plan: a fault-injection build of the SAME sources
(habdec_amd/libhabdec_amd_fault.so, -DHD_RING_FAULT: one LDS-DMA loader of the process never publishes its second tile and stops; or -- the second
mode -- one feeding wave stops handing out runs, so that its CU's loaders starve) must

  * not hang the device: every wait in stage1_ring.h is bounded, the launch ends by itself;
  * report: the waves whose wait ran out set the mapped host word, collect() fails the call with HD_ERR_DEVICE;
  * stay failed: up to three calls are undelivered when the word is seen, so the engine cannot say which one was hit -- every later call
    and the flush fail too, until the engine is destroyed.

The product library carries none of this (the hook is compiled out); the test loads the fault build into its own handle."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HD_ERR_DEVICE = -2


def load_fault_lib():
    from habdec_amd import capi
    path = capi.LIB_PATH.with_name("libhabdec_amd_fault.so")
    if not path.exists():
        pytest.skip("fault-injection library not built (python -c 'from habdec_amd.build import build; build(variant=\"fault\")')")
    L = C.CDLL(str(path))
    for table in (capi.ENGINE_API, capi.HOST_API):
        for name, (res, args) in table.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
    return L


@pytest.mark.parametrize("pipeline", [1, 0])
def test_a_dropped_publish_is_reported_and_the_engine_stays_failed(pipeline):
    torch = pytest.importorskip("torch")
    from habdec_amd import capi
    L = load_fault_lib()
    L.hd_debug_ring_fault_arm()                          # one loader of this process will drop one publish from here on
    S, CH, fs = 64, 16384, 2.5e6
    cfg = capi.hd_engine_config()
    L.hd_engine_config_default(C.byref(cfg))
    cfg.n_streams, cfg.max_chunk, cfg.sampling_rate, cfg.decimation, cfg.pipeline = S, CH, fs, 16, pipeline
    h = C.c_void_p()
    assert L.hd_engine_create(C.byref(cfg), C.byref(h)) == 0, L.hd_last_error()
    iq = torch.randn((S, CH, 2), device="cuda", dtype=torch.float32) * 0.1
    codes = []
    for k in range(8):                                   # the first call restarts the histories (classic grid); the per-CU kernels serve the rest
        codes.append(L.hd_process_device(h, iq.data_ptr(), CH, None, CH))
    codes.append(L.hd_flush(h))
    t = capi.hd_timing()
    L.hd_engine_timing(h, C.byref(t))
    assert t.step_variant == 1, "the per-CU ring kernel did not run: nothing was injected"
    assert HD_ERR_DEVICE in codes, codes                 # reported ...
    first = codes.index(HD_ERR_DEVICE)
    assert all(c == HD_ERR_DEVICE for c in codes[first:]), codes      # ... and sticky
    assert b"bounded wait" in L.hd_last_error()
    assert L.hd_process_device(h, iq.data_ptr(), CH, None, CH) == HD_ERR_DEVICE
    L.hd_engine_destroy(h)
    # the device is fine: a fresh engine of the PRODUCT library decodes as usual right after
    import habdec_amd
    eng = habdec_amd.Engine(n_streams=S, max_chunk=CH, sampling_rate=fs, decimation=16, pipeline=pipeline)
    for k in range(4):
        eng.process_device(iq.data_ptr(), CH, CH)
    eng.flush()
    eng.close()


@pytest.mark.parametrize("pipeline", [1, 0])
def test_starved_loaders_report_instead_of_waiting_forever(pipeline):
    """The other bounded wait: a CU's feeding wave stops handing out runs after its second (the fault build's second mode), so that CU's loaders find
    nothing to issue -- their wait runs out, they report and count themselves out, the computing waves leave, the launch ends; the engine fails the
    call and stays failed.  (Round 4's second mode -- a computing wave napping until its publication word had been overwritten -- tested a hazard the
    sixteen-entry publication queue had; round 5's per-slot words have no queue to overflow: a READY word is replaced only by the wave that took it.)"""
    torch = pytest.importorskip("torch")
    from habdec_amd import capi
    L = load_fault_lib()
    if not hasattr(L, "hd_debug_ring_fault_arm_starve"):
        pytest.skip("fault-injection library without the starvation mode (rebuild it)")
    L.hd_debug_ring_fault_arm_starve()
    S, CH, fs = 1024, 65536, 2.5e6                       # 128 tiles per CU: far more than two runs
    cfg = capi.hd_engine_config()
    L.hd_engine_config_default(C.byref(cfg))
    cfg.n_streams, cfg.max_chunk, cfg.sampling_rate, cfg.decimation, cfg.pipeline = S, CH, fs, 16, pipeline
    h = C.c_void_p()
    assert L.hd_engine_create(C.byref(cfg), C.byref(h)) == 0, L.hd_last_error()
    iq = torch.randn((S, CH, 2), device="cuda", dtype=torch.float32) * 0.1
    codes = [L.hd_process_device(h, iq.data_ptr(), CH, None, CH) for _ in range(6)]
    codes.append(L.hd_flush(h))
    t = capi.hd_timing()
    L.hd_engine_timing(h, C.byref(t))
    assert t.step_variant == 1, "the per-CU ring kernel did not run: nothing was injected"
    assert HD_ERR_DEVICE in codes, codes
    assert all(c == HD_ERR_DEVICE for c in codes[codes.index(HD_ERR_DEVICE):]), codes
    assert b"bounded wait" in L.hd_last_error(), L.hd_last_error()
    L.hd_engine_destroy(h)
    del iq
    torch.cuda.empty_cache()


def test_a_result_slot_without_its_tag_fails_the_engine_instead_of_delivering_stale_results():
    """Round 5: the engine's completion events carry no system-scope fence, so the host takes a result slot as delivered when it reads the call's tag in
    the slot's header (BitsHeader::seq, stored last by the wave that wrote the slot) -- not because the event fired.  The fault build makes ONE stream tail
    return without storing its tag: collect() must wait for it, give up after 200 ms, fail the call with HD_ERR_DEVICE, deliver nothing and stay failed."""
    torch = pytest.importorskip("torch")
    from habdec_amd import capi
    L = load_fault_lib()
    if not hasattr(L, "hd_debug_tail_fault_arm_no_tag"):
        pytest.skip("fault-injection library without the missing-tag mode (rebuild it)")
    S, CH, fs = 64, 16384, 2.048e6
    cfg = capi.hd_engine_config()
    L.hd_engine_config_default(C.byref(cfg))
    cfg.n_streams, cfg.max_chunk, cfg.sampling_rate, cfg.decimation, cfg.pipeline = S, CH, fs, 64, 0
    h = C.c_void_p()
    assert L.hd_engine_create(C.byref(cfg), C.byref(h)) == 0, L.hd_last_error()
    iq = torch.randn((S, CH, 2), device="cuda", dtype=torch.float32) * 0.1
    assert L.hd_process_device(h, iq.data_ptr(), CH, None, CH) == 0          # a healthy call first
    L.hd_debug_tail_fault_arm_no_tag()
    codes = [L.hd_process_device(h, iq.data_ptr(), CH, None, CH) for _ in range(3)]
    assert codes[0] == HD_ERR_DEVICE and all(c == HD_ERR_DEVICE for c in codes), codes
    assert b"does not carry its call's tag" in L.hd_last_error(), L.hd_last_error()      # (every later call of the failed engine names the cause it failed with)
    assert L.hd_flush(h) == HD_ERR_DEVICE
    assert b"tag" in L.hd_last_error(), L.hd_last_error()
    L.hd_engine_destroy(h)
