// libhabdec_amd.so -- host side of the MI355X RTTY demodulation engine: the C ABI of include/habdec_amd.h.
//
// Responsibilities: mirror the reference Decoder's integer bookkeeping per stream (how many samples each stage
// sees, when histories restart, when the spectrum buffer completes, when the low-pass is redesigned:
// reference code/Decoder/Decoder.h:416-542), upload it as one small parameter array per call, launch the gfx950
// kernels over all streams on one HIP stream, fetch the packed bits and spectrum statistics, and run the host
// text stage (RTTY framing, sentence extraction, CRC, callbacks: Decoder.h:559-637) and the AFC state machine.
// There is no CPU fallback: without a gfx950 device and the kernels in this library hd_engine_create fails.
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/habdec_amd.h"
#include "dev_types.h"
#include "host/afc_tracker.hpp"
#include "host/decim_plan.hpp"
#include "host/fir_design.hpp"
#include "host/iq_file_batch.hpp"
#include "host/text_stage.hpp"
#include "kernels/launch.h"

// A launcher (or layout helper) of the kernels that exist once per arithmetic mode (kernels/arith.h): hd::exact::fn or hd::fast::fn by the engine's mode.
#define HDK(fn, ...) (e->fast ? hd::fast::fn(__VA_ARGS__) : hd::exact::fn(__VA_ARGS__))

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HD_HIP(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) return fail(HD_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count)
    {
        n = count;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) e = hipMemset(p, 0, std::max<size_t>(count, 1) * sizeof(T));
        return e;
    }
    ~DevBuf() { if (p) (void)hipFree(p); }
};
template <typename T>
struct PinBuf {
    T* p = nullptr;
    T* dev = nullptr;      // the same memory as the GPU addresses it (kernels write results straight into it)
    size_t n = 0;
    hipError_t alloc(size_t count)
    {
        n = count;
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(count, 1) * sizeof(T), hipHostMallocMapped | hipHostMallocCoherent);
        if (e == hipSuccess) std::memset(p, 0, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void**>(&dev), p, 0);
        return e;
    }
    ~PinBuf() { if (p) (void)hipHostFree(p); }
};

struct StreamHost {
    // control plane
    double baud = 300;
    float lp_bw = 1500, lp_trans = 0.025f;
    bool dc = false;
    // size bookkeeping mirrored from the reference
    size_t stage_buf[2] = {0, 0};   // Decimator::p_buff_.size()
    size_t pending = 0;             // iq_samples_decimated_.size()
    size_t fft_fill = 0;            // freq_in_.size()
    bool have_spectrum = false;     // freq_out_.size() == 4096
    uint64_t spectra = 0;
    size_t fir_buf = 0;             // FirFilter::buff_.size()
    uint32_t fir_T_run = 0;         // tap count of the last low-pass run (FirHistory, dev_types.h)
    uint32_t head_n = 0;            // FirHistory head, lazily: how many samples of the last run's input a later run may ask for ...
    uint64_t head_call = ~0ull;     // ... the call that run was in (its input sits in THAT call's low-pass buffer until the buffer's turn comes again) ...
    bool head_aside = false;        // ... or: the head has been copied to the side buffer (the stream did not run in the call behind its last run)
    hd::LowpassDesigner lp;
    bool taps_dirty = false;
    // data-dependent state learned back from the device
    uint32_t held = 0, win_ub = 0, inflight_m = 0;   // win_ub: upper bound of backlog samples without final window results
    bool sym_reset = true;          // symbol parameters changed since the window cache was built
    // host stages
    hd::TextStage text;
    hd::AfcTracker afc;
    hd::SpectrumStats stats{};
    // last call (for the getters)
    uint32_t last_n2 = 0, last_pend_before = 0, last_m = 0, last_nbits = 0, last_nflips = 0;
    uint64_t flip_list_full = 0;   // calls in which the device's flip list filled up (kMaxFlipsPerCall) and the search went on a call later
    uint64_t bits_total = 0;
    int last_buf = 0;
    std::vector<uint32_t> last_words;
    uint32_t demod_ck[2] = {0, 0}, demod_ck_n = 0xFFFFFFFFu;   // discriminator checksum of the call delivered last (BitsHeader::demod_ck)
    uint64_t demod_ck_call = 0;                               // ... and that call's index (0 = the engine's first hd_process_* call)
    uint64_t demod_ck_hash = 0xCBF29CE484222325ull;           // every delivered call's (n, ck[0], ck[1]) folded into one word (hd_stream_demod_checksum_total)
    uint64_t demod_ck_calls = 0, demod_ck_unknown = 0;        // calls folded in / calls whose slot carried no checksum (the fused back end)
};

}  // namespace

struct hd_engine {
    hd_engine_config cfg{};
    double fs = 0, fsd = 0;
    std::vector<hd::DecimStage> stages;
    uint32_t S = 0, D = 1;
    uint32_t n1_cap = 0, n2_cap = 0;        // per-call capacities after stage 1 / last stage
    uint32_t taps_cap = 0, fir_hist_cap = 0; // low-pass tap capacity, history slots in front of the pending samples
    size_t fbuf_stride = 0;
    uint32_t tail_cap = 0, backlog_cap = 0, slot_words = 0, flips_cap = 0, max_R = 0, min_R = 4;
    int bins_sep = 8;
    bool decode_enabled = true;
    bool fast = false;         // hd_engine_config.arith == HD_ARITH_FAST: the hd::fast kernels (fused multiply-add in every FIR; tolerance 1e-5, not bit-identical floats)
    bool one_stream = false;
    bool no_fuse = false;      // HD_NO_FUSE: never use the fused back end (kernels/backend.hip); A/B measurements
    bool no_tail = false;      // HD_NO_TAIL: never use the one-wave stream tail (kernels/tail_body.h); A/B measurements
    uint32_t tail_max_n2 = 2048;   // HD_TAIL_MAX_N2: most decimated samples per call for which the stream tail is used
    int last_fuse = -1;
    // Step mode (batch decoding, kernels/decimate.hip k_step): the stream tails of call k ride in the stage-1 launch of call k+1.
    struct PendingTail { bool valid = false; hd::TailArgs ta{}; int slot = 0; bool any_fft = false; int r2 = 0, t2 = 0; } pend;
    bool own_fft = false;      // the spectrum, where it is a launch of its own, as single-wave workgroups (kernels/spectrum_wave.hip) instead of rocFFT + commit: the default; HD_ROCFFT=1 without HD_OWN_FFT=1 switches it off
    bool tail_fft = true;      // a stream tail transforms its stream's completed spectrum buffer itself (kernels/spectrum_wave.h); HD_ROCFFT=1: separate launches
    DevBuf<float2> fft_tw;     // (cos, -sin)(2 pi m / 4096), rounded once from double
    bool no_claim = false;     // HD_NO_CLAIM: step launches with fixed shares of tiles (A/B measurements)
    uint32_t step_run = 0;     // HD_STEP_RUN: tiles per drawn run (default 4, minimum 2)
    bool no_cu_step = false;   // HD_NO_CU_STEP: step launches as single-wave workgroups (k_step) instead of one workgroup per CU (k_step_cu)
    uint32_t ring_run = 0;     // HD_RING_RUN: tiles per drawn run of the per-CU ring kernels (default: pick_ring_run)
    uint32_t ring_short_pct = 25;   // HD_RING_SHORT_PCT: share of an XCD's tiles the worker waves draw as SINGLE tiles at the end of a launch (guided hand-out; 0 = whole runs to the end)
    static constexpr uint32_t kS1Loaders = 2;   // LDS-DMA loader waves of k_stage1_cu at /8 and /4 (round 4: with the nt policy on the body rows one loader's three
                                                // tiles in flight bound the launch -- 102.7 us with one loader, 94.7 with two, one box, alternating)
#ifndef HD_S1_SLOTS_BATCH
#define HD_S1_SLOTS_BATCH 4
#endif
    static constexpr uint32_t kS1Slots = HD_S1_SLOTS_BATCH;     // tile slots of k_stage1_cu in batch mode on the separate-kernels path: four leave half of a CU's LDS to the back-half
                                                // workgroups of the previous call on the other queue (/16: 0.334-0.343 ms per step against 0.342-0.351 with eight)
    PinBuf<unsigned int> ring_gave_up;     // mapped host word a wave of k_step_cu / k_stage1_cu sets when a bounded wait runs out (never in a correct run)
    std::string fail_cause;                // what put the engine into its failed state (reported again by every later call)
    bool device_failed = false;            // ... after which the engine stays failed: the launch that gave up left stage-1 output incomplete, and up to
                                           // three calls are undelivered by the time a collect() sees the word -- which of them it was cannot be told
    uint64_t step_launches = 0;
    DevBuf<unsigned int> step_ctr;         // two sets of per-XCD run counters ([2][16][32] u32), alternating per step launch
    uint32_t pend_max_taps = 0;
    bool dec_wgs_forced = false;   // HD_DEC_WGS_PER_CU was given (tests of the linear split)
    // Timing experiments (results wrong: step launches without their tails -- 1 -- or without stage 1 -- 2; tools/micro/joules.py) exist in a VARIANT build only
    // (-DHD_TIMING_EXPERIMENT, HD_BUILD_VARIANT=timingexp: reads HD_CU_EXP): no environment variable makes the product library skip the result-slot tag
    // check or deliver wrong results.
#ifdef HD_TIMING_EXPERIMENT
    int cu_exp = 0;
#else
    static constexpr int cu_exp = 0;
#endif
    uint32_t n_cus = 0, dec_wgs_per_cu = 0;   // stage-1 linear split: workgroups per CU (0 = classic grid), see kernels/decimate.hip        // path of the previous call (the two paths use the stage-2 buffers on different queues)
    hipStream_t qa = nullptr, qb = nullptr, qc = nullptr;   // front (decimation, spectrum) and back (FIR, symbols, results) HIP streams
    // hd_process_host: copies run on their own stream into two alternating slabs; a call returns once ITS copy has landed, so the
    // caller may reuse (or free) the buffer at once, like Decoder::pushSamples' copy -- whatever the engine still has queued.
    hipStream_t qh = nullptr;
    DevBuf<float2> staging2;
    hipEvent_t ev_copy[2] = {nullptr, nullptr}, ev_staging_free[2] = {nullptr, nullptr};
    bool staging_used[2] = {false, false};
    uint64_t host_calls = 0, host_calls_in_place = 0;   // hd_process_host calls through the staging slabs / served in place from page-locked memory
    uint32_t timing_every = 8;   // HIP-event timing on every Nth call (0 = off): each event record is a barrier packet worth ~6 us of queue time
    hd_timing last_timing{};
    rocfft_plan fft_plan = nullptr;
    rocfft_execution_info fft_info = nullptr;
    // Fast mode, long low-pass filters (>= kLpFftMinTaps taps: configs[4]'s 4097): the filter as transforms (kernels/fir_demod.hip: k_lp_gather / k_lp_mul around
    // rocFFT, k_fir_demod with the filtered samples handed in).  Two transform lengths, the smaller one where [history | inputs] of every stream fits it.
    static constexpr uint32_t kLpFftMinTaps = 1024;
    uint32_t lpf_N[2] = {0, 0};
    rocfft_plan lpf_fwd[2] = {nullptr, nullptr}, lpf_inv[2] = {nullptr, nullptr};
    rocfft_execution_info lpf_info = nullptr;
    DevBuf<char> lpf_workbuf;
    DevBuf<float2> lpf_x, lpf_k;           // [S][lpf_N[1]]: the input images / filtered samples of the call in the back half; the taps' spectra
    PinBuf<uint32_t> lpf_ntaps;            // [S]: tap counts the spectra in lpf_k were made from
    uint32_t lpf_k_N = 0;                  // ... and the transform length (0: none yet)
    bool lpf_k_stale = true;
    uint64_t lpf_calls = 0;
    DevBuf<char> fft_work;

    DevBuf<float2> staging, dec1, dec1b, dec1c, hist1[2], hist2[2], fbuf[3], fft_in, fft_raw, spec, filtered;   // dec1/b/c: stage-1 output, rotating per call
    DevBuf<float> stage_taps[2], lp_taps, power, demod, tail, weight;
    DevBuf<unsigned long long> flipmask;
    DevBuf<uint32_t> flips_dbg;
    DevBuf<hd::SymState> d_symstate;
    DevBuf<hd::DemodCarry> carry[2];
    DevBuf<float2> fir_head;          // [S][head_cap]: FirHistory heads moved aside -- the first samples of a stream's last low-pass run's input, copied here by the first call in
                                      // which the stream did not run (dev_types.h: the head, lazily); a stream that ran in the previous call finds them in that call's buffer
    DevBuf<uint32_t> demod_ck_acc;    // [S][2]: the discriminator checksum of the call in the back half of the separate-kernels path (k_fir_demod adds, k_symbols collects and clears)
    uint32_t head_cap = 0;
    DevBuf<hd::SymbolParams> d_sym;
    PinBuf<hd::SymbolParams> h_sym;
    // Everything a call in flight owns, double-buffered so call k+1 can be enqueued (and its front half can run)
    // while call k's back half is still executing and its results have not been read yet.
    static constexpr int kSlots = 4;   // calls whose host-visible blocks (parameters, result slots, spectrum statistics) may be in use at once: in flight + being delivered
    struct CallSlot {
        DevBuf<hd::StreamCall> d_call;
        PinBuf<hd::StreamCall> h_call;
        PinBuf<uint32_t> h_slots;                 // written by the symbol scan kernel over PCIe (zero-copy), read after ev_done
        PinBuf<hd::SpectrumStatsDev> h_stats;    // written by the spectrum kernel
        bool timed = false, timed_step = false;   // this call carries the timing events (a step call: only the two around its one launch)
        hipEvent_t ev_front = nullptr, ev_done = nullptr, ev_params = nullptr, ev_spec = nullptr, t0 = nullptr, t1 = nullptr, t2 = nullptr, t3 = nullptr;
        bool busy = false;
        uint32_t seq = 0;                         // this call's tag: what every result slot's BitsHeader::seq must read before the slot is taken as delivered
        uint64_t total_in = 0;
        uint32_t r1 = 1;
        ~CallSlot() { for (hipEvent_t ev : {ev_front, ev_done, ev_params, ev_spec, t0, t1, t2, t3}) if (ev) (void)hipEventDestroy(ev); }
    } slot[kSlots];
    uint64_t calls = 0;
    uint64_t delivered = 0;   // calls whose results have been delivered; calls - delivered <= 2 (pipelined mode)
    int cur = 0;          // which fbuf receives this call's chunk (three take turns: the one a call writes was last read by the back half of call k-3, which the host has collected)
    int hist_cur = 0;     // which stage-history buffers are read this call (the others are written)
    int carry_cur = 0;
    bool sym_dirty = true;

    std::vector<StreamHost> st;
    std::recursive_mutex mtx;   // recursive: a sentence / character callback (fired under it, on the calling thread) may call the text getters
    bool in_callback = false;   // ... but not the data getters that would have to flush the pipeline from inside a delivery
    hd_sentence_cb sentence_cb = nullptr; void* sentence_user = nullptr;
    hd_match_cb match_cb = nullptr; void* match_user = nullptr;
    hd_chars_cb chars_cb = nullptr; void* chars_user = nullptr;
    uint64_t sentences_ok = 0;

    ~hd_engine()
    {
        if (fft_plan) rocfft_plan_destroy(fft_plan);
        if (fft_info) rocfft_execution_info_destroy(fft_info);
        for (rocfft_plan pl : {lpf_fwd[0], lpf_fwd[1], lpf_inv[0], lpf_inv[1]}) if (pl) rocfft_plan_destroy(pl);
        if (lpf_info) rocfft_execution_info_destroy(lpf_info);
        if (qa) (void)hipStreamDestroy(qa);
        if (qb && qb != qa) (void)hipStreamDestroy(qb);
        if (qc && qc != qa) (void)hipStreamDestroy(qc);
        if (qh) (void)hipStreamDestroy(qh);
        for (hipEvent_t ev : {ev_copy[0], ev_copy[1], ev_staging_free[0], ev_staging_free[1]}) if (ev) (void)hipEventDestroy(ev);
    }
};

namespace {

std::once_flag g_rocfft_once;

hd::SymbolParams symbol_params(const hd_engine* e, const StreamHost& s)
{
    hd::SymbolParams p{};
    p.float_abs = e->cfg.lookup_mode ? 1u : 0u;
    p.min_held = 0xFFFFFFFFu;
    p.spb = 0; p.R = 4;
    if (e->fsd > 0 && s.baud > 0) {
        const double spb = std::round(e->fsd / s.baud);
        if (spb >= 1 && spb < 1e9) {
            p.spb = (uint32_t)spb;
            p.R = (uint32_t)std::max(4, int(p.spb / 4));
            p.min_held = (uint32_t)std::ceil(e->fsd / s.baud * 3);
        }
    }
    return p;
}

float cutoff_rel(const hd_engine* e, const StreamHost& s) { return (float)(s.lp_bw / e->fsd); }

int check_stream(const hd_engine* e, uint32_t s)
{
    if (!e) return fail(HD_ERR_INVALID, "null engine");
    if (s >= e->S) return fail(HD_ERR_INVALID, "stream index out of range");
    return HD_OK;
}

}  // namespace

extern "C" {

const char* hd_last_error(void) { return g_err.c_str(); }

/* Smallest non-empty per-stream sample count hd_process_* accepts for a decimation factor: a multiple of the factor that covers the
 * history of every stage (shorter inputs are undefined behaviour in the reference, Q4).  0 = unsupported factor. */
uint32_t hd_min_chunk(uint32_t decimation)
{
    std::vector<hd::DecimStage> st;
    if (!hd::decim_plan(decimation, st)) return 0;
    uint64_t need = decimation ? decimation : 1;
    uint64_t r = 1;
    for (const auto& g : st) { need = std::max<uint64_t>(need, (uint64_t)(g.taps.size() - 1) * r); r *= (uint64_t)g.ratio; }
    const uint64_t d = decimation ? decimation : 1;
    return (uint32_t)((need + d - 1) / d * d);
}

void hd_engine_config_default(hd_engine_config* c)
{
    std::memset(c, 0, sizeof(*c));
    c->device = 0; c->n_streams = 1; c->max_chunk = 65536; c->sampling_rate = 2.048e6; c->decimation = 64;
    c->baud = 300; c->rtty_bits = 8; c->rtty_stops = 2; c->lowpass_bw_hz = 1500; c->lowpass_trans = 0.025f;
    c->dc_remove = 0; c->lookup_mode = 1; c->enable_spectrum = 1; c->ungated = 0; c->keep_filtered = 0; c->pipeline = 0; c->arith = HD_ARITH_EXACT;
}

int hd_engine_create(const hd_engine_config* cfg, hd_engine** out)
{
    if (!cfg || !out) return fail(HD_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->n_streams < 1) return fail(HD_ERR_INVALID, "n_streams must be >= 1");
    if (!(cfg->sampling_rate > 0)) return fail(HD_ERR_INVALID, "sampling_rate must be > 0");
    if (cfg->max_chunk < 1) return fail(HD_ERR_INVALID, "max_chunk must be >= 1");
    if (cfg->arith != HD_ARITH_EXACT && cfg->arith != HD_ARITH_FAST) return fail(HD_ERR_INVALID, "arith must be HD_ARITH_EXACT (0) or HD_ARITH_FAST (1)");
    std::unique_ptr<hd_engine> e(new hd_engine);
    e->cfg = *cfg;
    e->fast = cfg->arith == HD_ARITH_FAST;
    if (!hd::decim_plan(cfg->decimation, e->stages))
        return fail(HD_ERR_INVALID, "Unsupported decimation factor: " + std::to_string(cfg->decimation));   // Decoder.h:317-319
    e->S = cfg->n_streams;
    e->D = cfg->decimation;
    if (cfg->max_chunk % e->D) return fail(HD_ERR_INVALID, "max_chunk must be a multiple of the decimation factor");
    e->fs = (double)(float)cfg->sampling_rate;        // Decoder::init(const float) (Decoder.h:223-225)
    e->fsd = e->fs / (double)e->D;
    e->decode_enabled = cfg->ungated || !(e->fsd > 4 * 40e3);   // Decoder.h:522
    if (cfg->baud > 0 && e->fsd / cfg->baud < 3.5)
        return fail(HD_ERR_UNSUPPORTED, "symbol rate above a third of the decimated rate (fewer than 4 samples per bit): the symbol extractor's windows would overlap");
    {   // AFC::FindPeaks separation in bins (AFC.h:110-111, 299-300)
        const float rel = (float)(500.0f / e->fsd);
        e->bins_sep = std::max(8, (int)std::round((double)rel * (double)hd::kFftBins));
    }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= cfg->device || cfg->device < 0)
        return fail(HD_ERR_DEVICE, "no HIP device " + std::to_string(cfg->device) + " (this library has no CPU path)");
    HD_HIP(hipSetDevice(cfg->device));
    HD_HIP(hipStreamCreateWithFlags(&e->qa, hipStreamNonBlocking));
    // Synchronous delivery (pipeline = 0) has nothing to overlap between calls: one queue, no cross-queue event between the front and the back half
    // (0.283 -> 0.268 ms per step at 1024 streams, 12 us of it the hop from one queue to the other).
    e->one_stream = cfg->pipeline == 0;
    e->no_fuse = getenv("HD_NO_FUSE") != nullptr;
    e->no_tail = getenv("HD_NO_TAIL") != nullptr;
    e->no_claim = getenv("HD_NO_CLAIM") != nullptr;
    if (const char* v = getenv("HD_STEP_RUN")) e->step_run = (uint32_t)atoi(v);
    e->no_cu_step = getenv("HD_NO_CU_STEP") != nullptr;
    if (const char* v = getenv("HD_RING_RUN")) e->ring_run = (uint32_t)atoi(v);
    if (const char* v = getenv("HD_RING_SHORT_PCT")) e->ring_short_pct = std::min<uint32_t>((uint32_t)atoi(v), 100u);
    if (const char* v = getenv("HD_TAIL_MAX_N2")) e->tail_max_n2 = (uint32_t)strtoul(v, nullptr, 0);
    {
        hipDeviceProp_t prop;
        HD_HIP(hipGetDeviceProperties(&prop, cfg->device));
        e->n_cus = (uint32_t)prop.multiProcessorCount;
        // Stage 1 of equally sized pushes runs as a linear split over k workgroups per CU.  Synchronous mode: all eight LDS slots of a
        // CU (measured against the classic per-stream grid on the same box: 125 vs 137 us per launch; 12, 16, 24, 32, 64 per CU: 150,
        // 135, 141, 135, 132 us).  Batch mode when the step kernel does not apply: six, so that the previous call's back half, on the
        // second queue, finds room underneath.
        e->dec_wgs_per_cu = cfg->pipeline ? 6u : 8u;
        if (const char* v = getenv("HD_DEC_WGS_PER_CU")) { e->dec_wgs_per_cu = (uint32_t)atoi(v); e->dec_wgs_forced = true; }
#ifdef HD_TIMING_EXPERIMENT
        if (const char* v = getenv("HD_CU_EXP")) e->cu_exp = atoi(v);
#endif
    }
    for (hipEvent_t* ev : {&e->ev_copy[0], &e->ev_copy[1], &e->ev_staging_free[0], &e->ev_staging_free[1]}) HD_HIP(hipEventCreateWithFlags(ev, hipEventDisableTiming));
    if (e->one_stream) e->qb = e->qc = e->qa;
    else {
        if (!e->qb) HD_HIP(hipStreamCreateWithFlags(&e->qb, hipStreamNonBlocking));
        HD_HIP(hipStreamCreateWithFlags(&e->qc, hipStreamNonBlocking));   // parameter fetches: tiny kernels that need not queue behind stage 1
    }
    for (auto& sl : e->slot) {
        // No system-scope fence at these events (hipEventDisableSystemFence): what queues hand each other on this device needs agent scope only, and with the
        // default flags every record wrote back and invalidated the L2s -- ~5 us of queue time per event between two step launches (kernel trace, round 4),
        // and a system-scope release on the dispatch that carries ev_done costs 2.8 % of a step (round 5, one box, alternating: 0.1316-0.1323 against
        // 0.1283-0.1285 ms).  What the HOST reads behind ev_done -- result slots, spectrum statistics, the give-up word -- lives in coherent mapped host memory and
        // is written straight over PCIe; HIP does not promise that an event without the fence orders those stores before the host's reads, so
        // collect() does not rely on it: every result slot carries the call's tag, stored last behind a wait for the writing wave's other stores
        // (BitsHeader::seq), and a slot counts as delivered when the host reads that tag.
        static const unsigned ev_flags = (unsigned)hipEventDisableSystemFence;
        static const unsigned done_flags = ev_flags;
        HD_HIP(hipEventCreateWithFlags(&sl.ev_front, hipEventDisableTiming | ev_flags));
        HD_HIP(hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming | done_flags));
        HD_HIP(hipEventCreateWithFlags(&sl.ev_params, hipEventDisableTiming | ev_flags));
        HD_HIP(hipEventCreateWithFlags(&sl.ev_spec, hipEventDisableTiming | ev_flags));
        for (hipEvent_t* ev : {&sl.t0, &sl.t1, &sl.t2, &sl.t3}) HD_HIP(hipEventCreateWithFlags(ev, ev_flags));
    }

    const uint32_t S = e->S;
    const uint32_t r0 = e->stages.size() > 0 ? e->stages[0].ratio : 1;
    e->n1_cap = cfg->max_chunk / r0;
    e->n2_cap = cfg->max_chunk / e->D;
    // low-pass tap capacity: the design clamps to the batch length m <= pending_max and forces odd (FirFilter.h:185-188)
    const uint32_t m_cap = ((e->n2_cap + hd::kFirBatch - 1) / hd::kFirBatch) * hd::kFirBatch + hd::kFirBatch;
    e->taps_cap = (m_cap | 1u) + 1u;
    e->fir_hist_cap = (e->taps_cap + 1u) & ~1u;       // even, so the chunk behind it stays 16-byte aligned
    e->fbuf_stride = (size_t)e->fir_hist_cap + hd::kFirBatch + e->n2_cap + 2;
    e->fbuf_stride = (e->fbuf_stride + 1) & ~(size_t)1;
    e->backlog_cap = hd::kVentLimit + 1 + m_cap;       // most samples the symbol extractor can hold (SymbolExtractor.h:116)
    e->tail_cap = 1u;                                  // per-stream symbol ring: power of two >= vent limit + one batch + slack
    while (e->tail_cap < hd::kVentLimit + 1 + m_cap + 1024) e->tail_cap <<= 1;

    e->st.resize(S);
    for (auto& s : e->st) {
        s.baud = cfg->baud; s.lp_bw = cfg->lowpass_bw_hz; s.lp_trans = cfg->lowpass_trans; s.dc = cfg->dc_remove != 0;
        s.text.framer.nbits = cfg->rtty_bits; s.text.framer.nstops = cfg->rtty_stops;
        s.lp.float_trig = cfg->lookup_mode != 0;
    }
    {   // result slot: header + packed bits.  bits per call <= 3*backlog/spb (SymbolExtractor run lengths)
        const hd::SymbolParams p = symbol_params(e.get(), e->st[0]);
        const uint32_t spb = std::max<uint32_t>(p.spb, 1);
        uint32_t bits = 3u * e->backlog_cap / spb + 64u;
        bits = std::max<uint32_t>(256u, (bits + 31u) & ~31u);
        e->slot_words = (uint32_t)(sizeof(hd::BitsHeader) / 4) + bits / 32;
        e->max_R = std::max<uint32_t>(p.R, 4);
        e->min_R = e->max_R;
    }
    e->flips_cap = cfg->keep_filtered ? 256 : 0;

    // device memory
    HD_HIP(e->staging.alloc((size_t)S * cfg->max_chunk));
    if (e->stages.size() == 2) for (auto* b : {&e->dec1, &e->dec1b, &e->dec1c}) HD_HIP(b->alloc((size_t)S * e->n1_cap));
    HD_HIP(e->step_ctr.alloc(2 * 16 * 32 + 32));
    HD_HIP(e->ring_gave_up.alloc(4));
    e->ring_gave_up.p[0] = 0;
    if (!e->no_claim && e->n_cus % 32u == 0) {
        // The drawn runs of the step launches are split by the hardware's XCC id (kernels/decimate.hip, stage1_ring.h): every id in
        // [0, n_cus / 32) must show up in a chip-filling grid and no other may.  A CU mask or a partition mode that breaks that
        // turns the draws off (fixed shares: slower, never wrong).
        const uint32_t n_xcd = e->n_cus / 32u;
        // (the allocations above clear their memory on the null stream, which a non-blocking stream does not wait for: without this
        // the clearing of step_ctr could land on top of the probe's answer -- seen in one test run of three as a silent fall-back to
        // fixed shares)
        HD_HIP(hipDeviceSynchronize());
        const uint32_t seen = HDK(probe_xcc_mask, e->qa, e->n_cus, e->step_ctr.p + 2 * 16 * 32);
        if (n_xcd > 16 || seen != (n_xcd >= 32 ? 0xFFFFFFFFu : (1u << n_xcd) - 1u)) e->no_claim = true;
    } else
        e->no_claim = true;
    if (e->stages.size() >= 1) {
        for (auto& h : e->hist1) HD_HIP(h.alloc((size_t)S * (e->stages[0].taps.size() - 1)));
        HD_HIP(e->stage_taps[0].alloc(e->stages[0].taps.size() + 1));       // (+1: the ring kernels read the taps as aligned pairs; the word behind an odd count is never used)
        HD_HIP(hipMemcpy(e->stage_taps[0].p, e->stages[0].taps.data(), e->stages[0].taps.size() * 4, hipMemcpyHostToDevice));
    }
    if (e->stages.size() == 2) {
        for (auto& h : e->hist2) HD_HIP(h.alloc((size_t)S * (e->stages[1].taps.size() - 1)));
        HD_HIP(e->stage_taps[1].alloc(e->stages[1].taps.size() + 1));
        HD_HIP(hipMemcpy(e->stage_taps[1].p, e->stages[1].taps.data(), e->stages[1].taps.size() * 4, hipMemcpyHostToDevice));
    }
    for (auto& b : e->fbuf) HD_HIP(b.alloc((size_t)S * e->fbuf_stride));
    HD_HIP(e->lp_taps.alloc((size_t)S * e->taps_cap));
    HD_HIP(e->demod.alloc((size_t)S * m_cap));
    if (cfg->keep_filtered) HD_HIP(e->filtered.alloc((size_t)S * m_cap));
    HD_HIP(e->tail.alloc((size_t)S * e->tail_cap));
    HD_HIP(e->weight.alloc((size_t)S * e->tail_cap));
    HD_HIP(e->flipmask.alloc((size_t)S * (e->tail_cap / 64)));
    HD_HIP(e->d_symstate.alloc(S));
    if (e->flips_cap) HD_HIP(e->flips_dbg.alloc((size_t)S * e->flips_cap));
    for (auto& c : e->carry) HD_HIP(c.alloc(S));
    e->head_cap = std::min<uint32_t>(e->taps_cap, 8192u);   // a tap-count jump of more than 8192 reads zeros beyond (documented)
    HD_HIP(e->fir_head.alloc((size_t)S * e->head_cap));
    HD_HIP(e->demod_ck_acc.alloc((size_t)2 * S));
    HD_HIP(e->d_sym.alloc(S));
    HD_HIP(e->h_sym.alloc(S));
    for (auto& sl : e->slot) {
        HD_HIP(sl.d_call.alloc(S));
        HD_HIP(sl.h_call.alloc(S));
        HD_HIP(sl.h_slots.alloc((size_t)S * e->slot_words));
    }
    if (cfg->enable_spectrum) {
        HD_HIP(e->fft_in.alloc((size_t)S * hd::kFftBins));
        HD_HIP(e->fft_raw.alloc((size_t)S * hd::kFftBins));
        HD_HIP(e->spec.alloc((size_t)S * hd::kFftBins));
        HD_HIP(e->power.alloc((size_t)S * hd::kFftBins));
        for (auto& sl : e->slot) HD_HIP(sl.h_stats.alloc(S));
        {
            std::vector<float2> tw(hd::kFftBins);
            for (size_t m = 0; m < hd::kFftBins; ++m) {
                const double a = 2.0 * 3.14159265358979323846264338327950288 * (double)m / (double)hd::kFftBins;
                tw[m] = make_float2((float)std::cos(a), (float)-std::sin(a));
            }
            HD_HIP(e->fft_tw.alloc(hd::kFftBins));
            HD_HIP(hipMemcpy(e->fft_tw.p, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice));
            e->tail_fft = getenv("HD_ROCFFT") == nullptr;
            // (round 6: where the spectrum is a launch of its own -- the separate-kernels path -- it is the single-wave kernel by default: it reads a chunk that
            // fills the buffer by itself in place, and transform + commit are one pass; HD_ROCFFT=1 alone: rocFFT + k_spectrum_commit)
            e->own_fft = e->tail_fft || getenv("HD_OWN_FFT") != nullptr;
        }
        std::call_once(g_rocfft_once, [] { rocfft_setup(); });
        const size_t len = hd::kFftBins;
        if (rocfft_plan_create(&e->fft_plan, rocfft_placement_notinplace, rocfft_transform_type_complex_forward,
                               rocfft_precision_single, 1, &len, S, nullptr) != rocfft_status_success)
            return fail(HD_ERR_DEVICE, "rocfft_plan_create failed");
        if (rocfft_execution_info_create(&e->fft_info) != rocfft_status_success)
            return fail(HD_ERR_DEVICE, "rocfft_execution_info_create failed");
        size_t wsz = 0;
        rocfft_plan_get_work_buffer_size(e->fft_plan, &wsz);
        if (wsz) {
            HD_HIP(e->fft_work.alloc(wsz));
            rocfft_execution_info_set_work_buffer(e->fft_info, e->fft_work.p, wsz);
        }
        rocfft_execution_info_set_stream(e->fft_info, e->qa);
    }
    if (e->fast && e->taps_cap >= hd_engine::kLpFftMinTaps && getenv("HD_NO_LP_FFT") == nullptr) {     // (HD_NO_LP_FFT=1: long filters as direct sums in fast mode too)
        std::call_once(g_rocfft_once, [] { rocfft_setup(); });
        uint32_t N = 1;
        while (N < m_cap + e->taps_cap) N <<= 1;
        e->lpf_N[1] = N; e->lpf_N[0] = N / 2;
        size_t wmax = 0;
        for (int w = 0; w < 2; ++w) {
            const size_t len = e->lpf_N[w];
            if (rocfft_plan_create(&e->lpf_fwd[w], rocfft_placement_inplace, rocfft_transform_type_complex_forward, rocfft_precision_single, 1, &len, S, nullptr) != rocfft_status_success ||
                rocfft_plan_create(&e->lpf_inv[w], rocfft_placement_inplace, rocfft_transform_type_complex_inverse, rocfft_precision_single, 1, &len, S, nullptr) != rocfft_status_success)
                return fail(HD_ERR_DEVICE, "rocfft_plan_create (low-pass transforms) failed");
            for (rocfft_plan pl : {e->lpf_fwd[w], e->lpf_inv[w]}) { size_t wsz = 0; rocfft_plan_get_work_buffer_size(pl, &wsz); wmax = std::max(wmax, wsz); }
        }
        if (rocfft_execution_info_create(&e->lpf_info) != rocfft_status_success) return fail(HD_ERR_DEVICE, "rocfft_execution_info_create failed");
        if (wmax) { HD_HIP(e->lpf_workbuf.alloc(wmax)); rocfft_execution_info_set_work_buffer(e->lpf_info, e->lpf_workbuf.p, wmax); }
        HD_HIP(e->lpf_x.alloc((size_t)S * N));
        HD_HIP(e->lpf_k.alloc((size_t)S * N));
        HD_HIP(e->lpf_ntaps.alloc(S));
    }
    // Pay first-use costs now (rocFFT's first execute alone stalls ~7 ms): run every kernel once on an all-idle call.
    {
        hd_engine::CallSlot& sl = e->slot[0];
        hipStream_t q = e->qa;
        const uint32_t T1 = e->stages.size() > 0 ? (uint32_t)e->stages[0].taps.size() : 0, T2 = e->stages.size() > 1 ? (uint32_t)e->stages[1].taps.size() : 0;
        if (e->stages.empty()) HDK(launch_passthrough, q, S, 1, e->staging.p, cfg->max_chunk, e->fbuf[0].p, e->fbuf_stride, sl.d_call.p, e->fir_hist_cap);
        if (e->stages.size() >= 1)
            HDK(launch_decimate, q, e->stages[0].ratio, T1, S, 1, e->staging.p, cfg->max_chunk, e->hist1[0].p, e->hist1[1].p, e->stage_taps[0].p,
                                e->stages.size() == 1 ? e->fbuf[0].p : e->dec1.p, e->stages.size() == 1 ? e->fbuf_stride : e->n1_cap, sl.d_call.p, 0,
                                e->stages.size() == 1, e->fir_hist_cap, nullptr);
        if (e->stages.size() == 2)
            HDK(launch_decimate, q, e->stages[1].ratio, T2, S, 1, e->dec1.p, e->n1_cap, e->hist2[0].p, e->hist2[1].p, e->stage_taps[1].p, e->fbuf[0].p,
                                e->fbuf_stride, sl.d_call.p, 1, 1, e->fir_hist_cap, nullptr);
        HDK(launch_dc_remove, q, S, e->fbuf[0].p, e->fbuf_stride, sl.d_call.p, e->fir_hist_cap);
        if (cfg->enable_spectrum) {
            HDK(launch_fft_feed, q, S, e->fbuf[0].p, e->fbuf_stride, e->fft_in.p, sl.d_call.p, e->fir_hist_cap);
            void* in[1] = {e->fft_in.p};
            void* outb[1] = {e->fft_raw.p};
            if (rocfft_execute(e->fft_plan, in, outb, e->fft_info) != rocfft_status_success) return fail(HD_ERR_DEVICE, "rocfft_execute failed");
            hd::launch_spectrum_commit(q, S, e->fft_raw.p, e->spec.p, e->power.p, sl.h_stats.dev, sl.d_call.p, e->fsd, e->bins_sep, 0u);
        }
        HDK(launch_fir_demod, q, S, 0, 0, e->fbuf[0].p, e->fbuf_stride, e->lp_taps.p, e->taps_cap, e->demod.p, e->demod.n / S, nullptr, e->carry[0].p,
                             e->carry[1].p, sl.d_call.p, e->fir_hist_cap, e->tail.p, e->tail_cap, e->d_symstate.p, e->fbuf[1].p,
                             e->fir_head.p, e->head_cap, e->fbuf[2].p);
        HDK(launch_symbols, q, S, 1, 1, e->max_R, e->tail.p, e->tail_cap, e->d_symstate.p, e->flipmask.p, e->weight.p, e->d_sym.p, sl.d_call.p,
                           sl.h_slots.dev, e->slot_words, nullptr, 0, e->min_R, 0u);
        HD_HIP(hipStreamSynchronize(q));
        HD_HIP(hipGetLastError());        // a kernel this shape cannot launch (LDS, grid) fails the creation, not every later call
        // the all-idle call moved nothing, but the history / carry ping-pong "out" buffers were written: restore zeros
        for (auto& h : e->hist1) if (h.p) HD_HIP(hipMemset(h.p, 0, h.n * sizeof(float2)));
        for (auto& h : e->hist2) if (h.p) HD_HIP(hipMemset(h.p, 0, h.n * sizeof(float2)));
        for (auto& c : e->carry) HD_HIP(hipMemset(c.p, 0, c.n * sizeof(hd::DemodCarry)));
        HD_HIP(hipMemset(e->d_symstate.p, 0, S * sizeof(hd::SymState)));
        if (const char* b0 = getenv("HD_SYM_BASE0")) {
            // test hook: start every stream's monotonic 32-bit sample position somewhere else than 0 (e.g. just below 2^32, which
            // a 32 kHz stream otherwise reaches after 37 hours), to exercise the wrap-around of the symbol rings' positions
            hd::SymState st0{};
            st0.base = st0.cached = st0.run_pos = (uint32_t)strtoul(b0, nullptr, 0);
            std::vector<hd::SymState> v(S, st0);
            HD_HIP(hipMemcpy(e->d_symstate.p, v.data(), S * sizeof(hd::SymState), hipMemcpyHostToDevice));
        }
    }
    HD_HIP(hipDeviceSynchronize());
    *out = e.release();
    return HD_OK;
}

void hd_engine_destroy(hd_engine* e)
{
    if (!e) return;
    (void)hipSetDevice(e->cfg.device);
    (void)hipDeviceSynchronize();
    delete e;
}

uint32_t hd_engine_streams(const hd_engine* e) { return e ? e->S : 0; }
uint32_t hd_engine_decimation(const hd_engine* e) { return e ? e->D : 0; }
double hd_engine_decimated_rate(const hd_engine* e) { return e ? e->fsd : 0; }
uint64_t hd_engine_sentences_ok(const hd_engine* e) { return e ? e->sentences_ok : 0; }
void hd_engine_set_timing(hd_engine* e, int on) { if (e) e->timing_every = on < 0 ? 0u : (uint32_t)on; }
int hd_engine_timing(hd_engine* e, hd_timing* out)
{
    if (!e || !out) return fail(HD_ERR_INVALID, "null argument");
    *out = e->last_timing;
    out->host_calls_in_place = e->host_calls_in_place;
    out->lowpass_fft_calls = e->lpf_calls;
    return HD_OK;
}

// diagnostic (not in include/habdec_amd.h): the two sets of per-XCD run counters of the step launches, [2][16] (after a device-wide wait)
extern "C" int hd_debug_step_counters(hd_engine* e, unsigned int* out32)
{
    if (!e || !out32) return HD_ERR_INVALID;
    std::vector<unsigned int> h(2 * 16 * 32);
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h.data(), e->step_ctr.p, h.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return HD_ERR_DEVICE;
    for (int i = 0; i < 32; ++i) out32[i] = h[(size_t)i * 32];
    return HD_OK;
}

void hd_set_sentence_callback(hd_engine* e, hd_sentence_cb cb, void* user) { if (e) { e->sentence_cb = cb; e->sentence_user = user; } }
void hd_set_match_callback(hd_engine* e, hd_match_cb cb, void* user) { if (e) { e->match_cb = cb; e->match_user = user; } }
void hd_set_chars_callback(hd_engine* e, hd_chars_cb cb, void* user) { if (e) { e->chars_cb = cb; e->chars_user = user; } }

/* ---------------------------------------------------------------- control plane -------------------------------- */

int hd_stream_set_baud(hd_engine* e, uint32_t s, double baud)
{
    if (int r = check_stream(e, s)) return r;
    std::lock_guard<std::recursive_mutex> l(e->mtx);
    if (baud > 0 && e->fsd / baud < 3.5)
        return fail(HD_ERR_UNSUPPORTED, "symbol rate above a third of the decimated rate (fewer than 4 samples per bit): the symbol extractor's windows would overlap");
    StreamHost probe = e->st[s];
    probe.baud = baud;
    const hd::SymbolParams p = symbol_params(e, probe);
    if (p.spb) {
        const uint32_t need = 3u * e->backlog_cap / p.spb + 64u;
        if (need > (e->slot_words - sizeof(hd::BitsHeader) / 4) * 32)
            return fail(HD_ERR_CAPACITY, "baud too high for the result slots sized at engine creation");
        if (p.R > e->max_R) e->max_R = p.R;
        if (p.R < e->min_R) e->min_R = std::max<uint32_t>(p.R, 4);
    }
    e->st[s].baud = baud;
    e->st[s].sym_reset = true;
    e->sym_dirty = true;
    return HD_OK;
}
int hd_stream_set_rtty(hd_engine* e, uint32_t s, uint32_t bits, float stops)
{
    if (int r = check_stream(e, s)) return r;
    std::lock_guard<std::recursive_mutex> l(e->mtx);
    e->st[s].text.framer.nbits = bits;
    e->st[s].text.framer.nstops = stops;
    return HD_OK;
}
int hd_stream_set_lowpass_bw(hd_engine* e, uint32_t s, float hz)         // Decoder.h:238-243
{
    if (int r = check_stream(e, s)) return r;
    std::lock_guard<std::recursive_mutex> l(e->mtx);
    StreamHost& st = e->st[s];
    st.lp_bw = hz;
    if (st.lp.design(cutoff_rel(e, st), st.lp_trans)) st.taps_dirty = true;
    return HD_OK;
}
int hd_stream_set_lowpass_trans(hd_engine* e, uint32_t s, float trans)   // Decoder.h:252-257
{
    if (int r = check_stream(e, s)) return r;
    std::lock_guard<std::recursive_mutex> l(e->mtx);
    StreamHost& st = e->st[s];
    st.lp_trans = trans;
    if (st.lp.design(cutoff_rel(e, st), st.lp_trans)) st.taps_dirty = true;
    return HD_OK;
}
int hd_stream_set_dc_remove(hd_engine* e, uint32_t s, int on)
{
    if (int r = check_stream(e, s)) return r;
    std::lock_guard<std::recursive_mutex> l(e->mtx);
    e->st[s].dc = on != 0;
    return HD_OK;
}
int hd_stream_reset_frequency_correction(hd_engine* e, uint32_t s, double c)
{
    if (int r = check_stream(e, s)) return r;
    std::lock_guard<std::recursive_mutex> l(e->mtx);
    e->st[s].afc.reset(c, hd::kFftBins, e->fsd);
    return HD_OK;
}

/* ---------------------------------------------------------------- data path ------------------------------------ */

namespace {

// rocFFT + commit for the streams whose 4096-sample buffer completed in the call that owns `sl` (Decoder.h:475-489)
int transform_and_commit(hd_engine* e, hipStream_t q, hd::SpectrumStatsDev* stats_dev, const hd::StreamCall* dcall, uint32_t seq, const float2* chunk = nullptr)
{
    if (e->own_fft) {   // one launch, one wave per stream (kernels/spectrum_wave.hip)
        hd::launch_spectrum_wave(q, e->S, e->fft_in.p, e->fft_tw.p, e->spec.p, e->power.p, stats_dev, dcall, e->fsd, e->bins_sep, seq, chunk, e->fbuf_stride, e->fir_hist_cap);
        return HD_OK;
    }
    void* in[1] = {e->fft_in.p};
    void* outb[1] = {e->fft_raw.p};
    rocfft_execution_info_set_stream(e->fft_info, q);
    if (rocfft_execute(e->fft_plan, in, outb, e->fft_info) != rocfft_status_success) return fail(HD_ERR_DEVICE, "rocfft_execute failed");
    hd::launch_spectrum_commit(q, e->S, e->fft_raw.p, e->spec.p, e->power.p, stats_dev, dcall, e->fsd, e->bins_sep, seq);
    return HD_OK;
}

int run_spectrum(hd_engine* e, hipStream_t q, hd_engine::CallSlot& sl, bool any_fft)
{
    if (!e->cfg.enable_spectrum || !any_fft) return HD_OK;
    return transform_and_commit(e, q, sl.h_stats.dev, sl.d_call.p, sl.seq);
}

// Step mode: the tails of the newest call have not been launched yet (they wait for the next call's launch).  Run them now, as a
// kernel of their own -- the next call is not coming (flush), or its results are wanted first.
int run_pending_tail(hd_engine* e)
{
    if (!e->pend.valid) return HD_OK;
    hd_engine::CallSlot& ps = e->slot[e->pend.slot];
    e->pend.valid = false;
    hd::TailArgs ta = e->pend.ta;
    const int lanes = (e->S >= 2 * e->n_cus ? 64 : 256);
    hd::TailArgs lay = ta;
    if (lanes != 64) {      // the pending arguments carry the 64-lane carve of the step launch; a 256-lane kernel needs its own
        // (same buffers, other LDS offsets)
        if (!HDK(tail_layout, lay, lanes, e->pend.r2, e->pend.t2, e->pend_max_taps, e->max_R, e->min_R, e->tail_cap, e->pend.ta.pend_max, 64 * 1024)) return fail(HD_ERR_INVALID, "stream tail layout");
    }
    const bool own_spectrum = !ta.fft_tw && e->cfg.enable_spectrum && e->pend.any_fft;       // a transform launch follows the tails: the event goes behind that one
    if (!HDK(launch_tail, e->qa, lanes, e->pend.r2, e->pend.t2, e->S, lay, own_spectrum ? nullptr : ps.ev_done)) return fail(HD_ERR_INVALID, "stream tail refused a shape it was selected for");
    if (own_spectrum) { if (!ta.fft_tw) { if (const int r = run_spectrum(e, e->qa, ps, e->pend.any_fft)) return r; } HD_HIP(hipEventRecord(ps.ev_done, e->qa)); }
    HD_HIP(hipGetLastError());
    return HD_OK;
}

// Deliver the results of the call that ran in `sl`: wait for its back half, then the host stages, stream by stream
// (AFC state machine Decoder.h:501-515; RTTY framing, sentence scan, callbacks Decoder.h:559-637).
int collect(hd_engine* e, hd_engine::CallSlot& sl)
{
    if (!sl.busy) return HD_OK;
    if (e->pend.valid && &e->slot[e->pend.slot] == &sl) { if (const int r = run_pending_tail(e)) return r; }
    const auto w0 = std::chrono::steady_clock::now();
    HD_HIP(hipEventSynchronize(sl.ev_done));
    const auto w1 = std::chrono::steady_clock::now();
    sl.busy = false;
    if (sl.timed) {
        float a = 0, b = 0;
        HD_HIP(hipEventElapsedTime(&b, sl.t1, sl.t2));
        if (sl.timed_step) a = b; else HD_HIP(hipEventElapsedTime(&a, sl.t0, sl.t3));
        e->last_timing.ms_total = a;
        e->last_timing.ms_front = b;
        e->last_timing.samples = sl.total_in;
        e->last_timing.front_bytes = sl.total_in * 8 + (sl.total_in / sl.r1) * 8;
        ++e->last_timing.timed_calls;
    }
    int rc = HD_OK;
    {   // every stream's slot must carry this call's tag (the kernels store it last); in practice it is there when the event has fired -- if not, wait for it
        const auto t_lim = std::chrono::steady_clock::now() + std::chrono::milliseconds(200);
        const bool timing_experiment = e->cu_exp != 0;      // (only in the -DHD_TIMING_EXPERIMENT variant build: step launches without their tails or without stage 1 -- nothing writes the slots)
        auto await = [&](const volatile uint32_t* tag, const char* what, uint32_t s) {
            for (uint32_t spins = 0; *tag != sl.seq; ++spins) {
                if (std::chrono::steady_clock::now() > t_lim) {
                    e->device_failed = true;
                    e->fail_cause = std::string(what) + " of stream " + std::to_string(s) + " does not carry its call's tag 200 ms after the completion event";
                    rc = fail(HD_ERR_DEVICE, e->fail_cause + ": the results of this engine are unreliable -- destroy the engine");
                    return;
                }
                if (spins > 64u) std::this_thread::yield();      // (in practice the tag is there when the event has fired)
            }
        };
        for (uint32_t s = 0; s < e->S && !e->device_failed && !timing_experiment; ++s) {
            await(&reinterpret_cast<const volatile hd::BitsHeader*>(sl.h_slots.p + (size_t)s * e->slot_words)->seq, "result slot", s);
            // ... and the spectrum statistics of a stream whose buffer completed in this call: written by the tail itself (behind the same tag) or by a
            // commit / spectrum launch of its own, whose stores no event of this engine fences
            if (!e->device_failed && e->cfg.enable_spectrum && sl.h_stats.p && sl.h_call.p[s].fft_run)
                await(reinterpret_cast<const volatile uint32_t*>(&sl.h_stats.p[s].seq), "spectrum statistics", s);
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (e->ring_gave_up.p && e->ring_gave_up.p[0] && !e->device_failed) {
        e->device_failed = true;
        e->fail_cause = "k_stage1_cu (loader / consumer waves): a bounded wait ran out inside a launch (word " + std::to_string(e->ring_gave_up.p[0]) + ")";
    }
    if (e->device_failed && rc == HD_OK)         // (a later call of a failed engine: the cause recorded when it failed)
        rc = fail(HD_ERR_DEVICE, e->fail_cause + ": the results of this and of every later call of this engine are unreliable -- destroy the engine");
    if (e->device_failed) {
        // Nothing of this slot is delivered: the launch that gave up left stage-1 output incomplete, so the characters -- possibly CRC-passing
        // sentences -- framed from it would be handed to the callbacks as if they were results.  No AFC step, no framer push, no callbacks, no counters.
        e->last_timing.host_wait_us = std::chrono::duration<double, std::micro>(w1 - w0).count();
        e->last_timing.host_text_us = 0;
        return rc;
    }
    for (uint32_t s = 0; s < e->S; ++s) {
        StreamHost& st = e->st[s];
        const hd::StreamCall& c = sl.h_call.p[s];
        if (c.fft_run) {
            std::memcpy(&st.stats, &sl.h_stats.p[s], sizeof(st.stats));
            st.have_spectrum = true;
            ++st.spectra;
        }
        const bool past_batch_gate = c.fir_m || c.clear_pending;
        if (past_batch_gate && e->cfg.enable_spectrum) st.afc.step(st.have_spectrum, st.stats, hd::kFftBins, e->fsd);
        const uint32_t* slot = sl.h_slots.p + (size_t)s * e->slot_words;
        const hd::BitsHeader* hdr = reinterpret_cast<const hd::BitsHeader*>(slot);
        st.held = hdr->held_after;
        st.inflight_m -= c.fir_m;
        st.win_ub = hdr->uncached + st.inflight_m;
        st.last_nbits = hdr->nbits; st.last_nflips = hdr->nflips;
        st.bits_total += hdr->nbits;
        st.demod_ck[0] = hdr->demod_ck[0]; st.demod_ck[1] = hdr->demod_ck[1]; st.demod_ck_n = hdr->demod_n; st.demod_ck_call = e->delivered;
        if (hdr->demod_n != 0xFFFFFFFFu) {
            for (const uint32_t x : {hdr->demod_n, hdr->demod_ck[0], hdr->demod_ck[1]}) st.demod_ck_hash = (st.demod_ck_hash ^ x) * 0x100000001B3ull;
            ++st.demod_ck_calls;
        } else ++st.demod_ck_unknown;
        if (hdr->overflow & 1u) rc = fail(HD_ERR_CAPACITY, "symbol result slot overflow on stream " + std::to_string(s));
        if (hdr->overflow & 2u) ++st.flip_list_full;     // (the search stopped at the flip-list bound and resumes next call: bits arrive a call later)
        const uint32_t* words = slot + sizeof(hd::BitsHeader) / 4;
        st.last_words.assign(words, words + (hdr->nbits + 31) / 32);
        if (!c.fir_m) continue;
        if (hdr->nbits) st.text.framer.push_packed(words, hdr->nbits);
        const std::string chars = st.text.run(hdr->nbits != 0, [&](const hd::SentenceMatch& m) {
            ++e->sentences_ok;
            if (e->sentence_cb) { e->in_callback = true; e->sentence_cb(e->sentence_user, s, m.callsign.c_str(), m.data.c_str(), m.crc.c_str()); e->in_callback = false; }
        }, [&](const hd::SentenceMatch& m, bool ok) {
            if (e->match_cb) { e->in_callback = true; e->match_cb(e->match_user, s, m.callsign.c_str(), m.data.c_str(), m.crc.c_str(), ok ? 1 : 0); e->in_callback = false; }
        });
        if (!chars.empty() && e->chars_cb) { e->in_callback = true; e->chars_cb(e->chars_user, s, chars.data(), chars.size()); e->in_callback = false; }
    }
    e->last_timing.host_wait_us = std::chrono::duration<double, std::micro>(w1 - w0).count();
    e->last_timing.host_text_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w1).count();
    return rc;
}

int flush_locked(hd_engine* e)
{
    int rc = HD_OK;
    // step mode: the newest call's tails have not been launched yet -- start them before the host spends time on the older call's text
    if (e->delivered < e->calls) { if (const int r = run_pending_tail(e)) return r; }
    while (e->delivered < e->calls) {           // oldest first
        const int r = collect(e, e->slot[e->delivered % hd_engine::kSlots]);
        if (r) rc = r;
        ++e->delivered;
    }
    return rc;
}

}  // namespace

int hd_flush(hd_engine* e)
{
    if (!e) return fail(HD_ERR_INVALID, "null engine");
    std::lock_guard<std::recursive_mutex> lock(e->mtx);
    if (e->in_callback) return fail(HD_ERR_INVALID, "hd_flush cannot be called from a sentence / character callback (the delivery it would join is the one running)");
    HD_HIP(hipSetDevice(e->cfg.device));
    const int rc = flush_locked(e);
    if (e->device_failed) return fail(HD_ERR_DEVICE, "this engine is in its failed state (" + e->fail_cause + ") -- destroy the engine");
    return rc;
}

int hd_process_device(hd_engine* e, const void* d_iq, size_t stride, const uint32_t* n_per_stream, uint32_t n_uniform)
{
    if (!e) return fail(HD_ERR_INVALID, "null engine");
    if (!d_iq) return fail(HD_ERR_INVALID, "null IQ pointer");
    if ((reinterpret_cast<uintptr_t>(d_iq) & 15) || (stride & 1)) return fail(HD_ERR_INVALID, "IQ base must be 16-byte aligned and stream_stride even");
    std::lock_guard<std::recursive_mutex> lock(e->mtx);
    if (e->in_callback) return fail(HD_ERR_INVALID, "hd_process_* cannot be called from a sentence / character callback");   // (a nested delivery would count the slot being delivered twice)
    if (e->device_failed) return fail(HD_ERR_DEVICE, "this engine is in its failed state (" + e->fail_cause + "): no new calls are accepted -- destroy the engine");
    const auto h0 = std::chrono::steady_clock::now();
    HD_HIP(hipSetDevice(e->cfg.device));
    const uint32_t S = e->S;
    const size_t nst = e->stages.size();
    const uint32_t T1 = nst > 0 ? (uint32_t)e->stages[0].taps.size() : 0, T2 = nst > 1 ? (uint32_t)e->stages[1].taps.size() : 0;
    const uint32_t R1 = nst > 0 ? e->stages[0].ratio : 1, R2 = nst > 1 ? e->stages[1].ratio : 1;

    // ---- validate first, so a rejected call leaves every stream untouched
    for (uint32_t s = 0; s < S; ++s) {
        const uint32_t n = n_per_stream ? n_per_stream[s] : n_uniform;
        if (n > e->cfg.max_chunk) return fail(HD_ERR_CAPACITY, "more samples than max_chunk");
        if (n % e->D) return fail(HD_ERR_INVALID, "sample count must be a multiple of the decimation factor (queue the remainder)");
        if (n && nst > 0 && n + 1 < T1) return fail(HD_ERR_UNSUPPORTED, "chunk shorter than the first stage's history (undefined in the reference)");
        if (n && nst > 1 && n / R1 + 1 < T2) return fail(HD_ERR_UNSUPPORTED, "chunk shorter than the second stage's history (undefined in the reference)");
    }
    // Control-plane changes (new low-pass design, new symbol parameters) rewrite device tables the previous call's
    // back half may still be reading: drain the pipeline first.  Rare.
    bool tables_dirty = e->sym_dirty;
    for (uint32_t s = 0; s < S && !tables_dirty; ++s) tables_dirty = e->st[s].taps_dirty;
    if (tables_dirty) { if (int rc = flush_locked(e)) return rc; }

    hd_engine::CallSlot& sl = e->slot[e->calls % hd_engine::kSlots];
    while (sl.busy) {                            // cannot happen with depth <= 3, but never reuse a slot in flight
        if (int rc = collect(e, e->slot[e->delivered % hd_engine::kSlots])) return rc;
        ++e->delivered;
    }

    // ---- host mirror of the reference's size bookkeeping -> one StreamCall per stream
    uint32_t max_in = 0, min_in = 0xFFFFFFFFu, max_n1 = 0, max_n2 = 0, max_m = 0, max_taps = 0, max_new = 0, max_pend = 0;
    bool any_fft = false, any_dc = false, any_zero1 = false;
    uint64_t total_in = 0;
    for (uint32_t s = 0; s < S; ++s) {
        StreamHost& st = e->st[s];
        hd::StreamCall c{};
        const uint32_t n = n_per_stream ? n_per_stream[s] : n_uniform;
        c.n_in = n;
        c.n1 = n / R1;
        c.n2 = c.n1 / R2;
        if (n) {   // Decimator::setInput growth rule (Decimator.h:74-79, Q5)
            if (nst > 0) { const size_t want = (size_t)n + T1 + R1; if (st.stage_buf[0] < want) { st.stage_buf[0] = want; c.zero_hist1 = 1; } }
            if (nst > 1) { const size_t want = (size_t)c.n1 + T2 + R2; if (st.stage_buf[1] < want) { st.stage_buf[1] = want; c.zero_hist2 = 1; } }
        }
        c.dc_remove = st.dc && c.n2;
        c.pend_before = (uint32_t)st.pending;
        st.pending += c.n2;
        // spectrum collection (Decoder.h:467-489)
        c.fft_fill = (uint32_t)st.fft_fill;
        if (e->cfg.enable_spectrum && n && st.fft_fill < (size_t)hd::kFftBins && c.n2) {
            c.fft_take = (uint32_t)std::min<size_t>(hd::kFftBins - st.fft_fill, c.n2);
            st.fft_fill += c.fft_take;
        }
        if (e->cfg.enable_spectrum && n && st.fft_fill >= (size_t)hd::kFftBins) { c.fft_run = 1; st.fft_fill = 0; any_fft = true; }
        // batch gate, rate gate, low-pass batch (Decoder.h:492-542)
        const bool past_batch_gate = n && st.pending >= hd::kFirBatch;
        st.last_m = 0;
        if (past_batch_gate && !e->decode_enabled) { c.clear_pending = 1; st.pending = 0; }
        else if (past_batch_gate) {
            const uint32_t m = (uint32_t)(st.pending - st.pending % hd::kFirBatch);
            st.lp.batch = m;                                                       // FirFilter::setInput
            if (st.lp.design(cutoff_rel(e, st), st.lp_trans)) {                    // LP_BlackmanHarris, Decoder.h:538
                st.taps_dirty = true;
                if (int rc = flush_locked(e)) return rc;                           // (only when a stream's batch size changes)
            }
            const uint32_t T = (uint32_t)st.lp.taps.size();
            if (!T) return fail(HD_ERR_UNSUPPORTED, "low-pass transition width leaves no taps (reference filters nothing then)");
            if (T > e->taps_cap) return fail(HD_ERR_CAPACITY, "low-pass tap count exceeds engine capacity");
            if (st.fir_buf < (size_t)m + T) { st.fir_buf = (size_t)m + T; c.fir_zero_hist = 1; }   // FirFilter.h:141-147
            c.fir_m = m; c.fir_taps = T;
            // (tap count of the previous run, and where the first samples of that run's input are: in the previous call's buffer if that is where it ran, else aside)
            c.fir_taps_prev = hd::sc_pack_taps_prev(st.fir_T_run ? st.fir_T_run : T, st.head_n, !st.head_aside && st.head_call + 1 == e->calls, false);
            st.fir_T_run = T;
            st.head_n = std::min<uint32_t>(m, std::min<uint32_t>(e->head_cap, 0x3FFFu)); st.head_call = e->calls; st.head_aside = false;
            st.pending -= m;
            st.last_m = m;
            max_m = std::max(max_m, m);
            max_taps = std::max(max_taps, T);
            // positions whose windows can become computable this call (device: k_sym_avg range), as an upper bound
            const uint32_t pending_windows = st.sym_reset ? std::max(st.held, st.win_ub) : st.win_ub;
            max_new = std::max(max_new, pending_windows + m + 64u);
            st.inflight_m += m;
            st.win_ub += m;
        }
        if (!c.fir_m && st.head_n && !st.head_aside) {
            // no run in this call: a head that still sits in the previous call's buffer moves aside now (that buffer is rewritten two calls on)
            if (st.head_call + 1 == e->calls) c.fir_taps_prev = hd::sc_pack_taps_prev(0, st.head_n, false, true);
            else st.head_n = 0;          // (cannot happen: the call behind a run either runs or saves)
            st.head_aside = true;
        }
        c.pend_after = (uint32_t)st.pending;
        max_pend = std::max(max_pend, std::max(c.pend_before, c.pend_after));
        st.last_n2 = c.n2; st.last_pend_before = c.pend_before; st.last_buf = e->cur;
        sl.h_call.p[s] = c;
        max_in = std::max(max_in, n); min_in = std::min(min_in, n); max_n1 = std::max(max_n1, c.n1); max_n2 = std::max(max_n2, c.n2);
        any_dc |= c.dc_remove != 0;
        any_zero1 |= c.zero_hist1 != 0;
        total_in += n;
    }
    sl.total_in = total_in;
    sl.r1 = R1;
    sl.seq = (uint32_t)(e->calls + 1u);           // (never the tag this slot carried four calls ago)
    if (!sl.seq) sl.seq = 1u;                     // (0 is what a slot holds before its first call: skipped when the counter wraps after 2^32 calls)
    // ---- uploads: per-call parameters, changed low-pass designs, changed symbol parameters
    hipStream_t qa = e->qa, qb = e->qb;
    // Two-stage plans at batch-decoding sizes: stage 2, low-pass, discriminator and slide run as ONE kernel per call on qb
    // (kernels/backend.hip) when a stream's call fits in LDS and no DC blocker sits in between.
    // The stream tail (kernels/tail_body.h) goes further: stage 2 through the symbol extractor as one wave per stream that walks the
    // call in pieces, so neither the call size nor the tap count has to fit an LDS image.
    hd::TailArgs ta{};
    const int tail_lanes = (S >= 2 * e->n_cus ? 64 : 256);
    // One wave per stream is the right shape while a call is short: its time grows with the decimated samples per call, and beyond
    // ~2048 of them (e.g. /16 with 65536-sample pushes: 4096) the many-workgroup kernels finish a batch sooner (measured: 156 vs 173 GS/s).
    // (a handful of streams, long calls: the 256-lane tail in pieces of 1024 outputs where its windows fit -- one piece for a 65536-sample push at /64 --, of 256 otherwise)
    const bool tail = nst == 2 && !any_dc && !e->no_tail && max_n2 <= e->tail_max_n2 &&
                      ((tail_lanes == 256 && max_n2 >= 512u && HDK(tail_layout, ta, tail_lanes, (int)R2, (int)T2, max_taps, e->max_R, e->min_R, e->tail_cap, max_pend, 64 * 1024, 4)) ||
                       HDK(tail_layout, ta, tail_lanes, (int)R2, (int)T2, max_taps, e->max_R, e->min_R, e->tail_cap, max_pend, 64 * 1024));
    const bool fuse = tail || (nst == 2 && !any_dc && !e->no_fuse && ((R2 == 2 && T2 == 69) || (R2 == 4 && T2 == 139)) &&
                      HDK(backend_lds_bytes, (int)T2, max_n1, max_n2, max_taps) <= 64 * 1024);
    // Step mode: batch decoding of equally sized pushes through a single-wave first stage -- ONE launch per call, on one queue: this
    // call's stage 1 with the previous call's stream tails in front (kernels/decimate.hip k_step).
    hd::TailArgs ta_step{};
    const bool step = tail && e->cfg.pipeline && !e->one_stream && min_in == max_in && max_in && (R1 == 32 || R1 == 64) &&
                      HDK(step_lds_bytes, (int)R1, (int)T1) &&
                      HDK(tail_layout, ta_step, 64, (int)R2, (int)T2, max_taps, e->max_R, e->min_R, e->tail_cap, max_pend, HDK(step_lds_bytes, (int)R1, (int)T1));
    const int path = step ? 3 : tail ? 2 : fuse ? 1 : 0;
    if (e->last_fuse >= 0 && e->last_fuse != path) {   // path switch: drain (a pending tail first)
        if (const int r = run_pending_tail(e)) return r;
        HD_HIP(hipStreamSynchronize(qa)); HD_HIP(hipStreamSynchronize(qb));
    }
    e->last_fuse = path;
    e->last_timing.path = (uint32_t)path;
    if (path == 0 && e->own_fft && e->cfg.enable_spectrum && !any_dc && any_fft) {
        // A call whose decimated chunk alone fills a stream's (empty) spectrum buffer: the spectrum launch reads the chunk's head in place and nothing is copied
        // into the collection buffer (fft_take = 0 for the stage that would have fed it; fft_run = 2 for the spectrum kernel) -- at 4096 and more decimated
        // samples per call that is every call: 8 + 8 bytes per decimated sample less traffic, and transform + commit are one pass instead of three.
        for (uint32_t s = 0; s < S; ++s) {
            hd::StreamCall& c = sl.h_call.p[s];
            if (c.fft_run == 1 && c.fft_fill == 0 && c.fft_take == (uint32_t)hd::kFftBins) { c.fft_take = 0; c.fft_run = 2; }
        }
    }
    // Cross-queue event waits cost ~18 us each on this platform (kernel trace: stage 1 of call k+1 started 37 us after stage 1
    // of call k ended, two barrier packets later).  On the fused path the stage-1 queue therefore waits for NOTHING: it reads the
    // call's parameters straight from the mapped host block (32 bytes per workgroup), its output rotates over three buffers --
    // the one it writes was last read by call k-2, which the host has already collected -- and the parameter fetch for the back
    // half rides in front of the back end on qb.  The unfused path keeps the fetch queue and the ordering behind call k-2 (its
    // front half writes the low-pass buffer the back half of call k-2 may still be reading).
    const bool lean = fuse && !e->one_stream;
    // Round 5: the unfused TWO-stage plans (calls too large for an LDS image or a one-wave tail: /16 and /256 at 4096 decimated samples per call) do not wait
    // either.  What the wait protected was the low-pass buffer -- with three of them taking turns, the one this call's second stage writes was last read by
    // the back half of call k-3, which the host has collected before it submits call k -- and the parameter copy rides on qa behind stage 1, which reads the
    // mapped host block.  (Kernel trace before: 15-22 us of barrier packets in front of every stage 1, on the queue a step waits for.)
    const bool free_front = !fuse && nst == 2 && !e->one_stream;
    if (!lean && !free_front && e->calls >= 2 && !e->one_stream) HD_HIP(hipStreamWaitEvent(qa, e->slot[(e->calls - 2) % hd_engine::kSlots].ev_done, 0));
    sl.timed = e->timing_every && (e->calls % e->timing_every) == 0;
    sl.timed_step = sl.timed && step;                       // a step call is ONE launch: two event records (each a barrier packet), not four
    if (sl.timed && !step) HD_HIP(hipEventRecord(sl.t0, qa));
    // One queue and the stream tails behind stage 1 (synchronous delivery): both kernels read the call's parameters from the mapped host block -- one
    // round trip over PCIe at the start of a workgroup instead of a copy kernel in front of stage 1.  (Not where a spectrum launch reads the device copy.)
    const bool host_params = tail && !step && e->one_stream && (!e->cfg.enable_spectrum || (e->tail_fft && !any_dc));
    if (lean || free_front) {
        if (e->sym_dirty && e->calls > e->delivered) { if (int rc = flush_locked(e)) return rc; }   // symbol parameters are uploaded below: nothing may still read them
    } else if (!host_params) {
        // the parameter block is pulled on its own queue, so it does not wait for the previous call's stage 1 to drain
        HDK(launch_fetch_params, e->qc, sl.h_call.dev, sl.d_call.p, S * sizeof(hd::StreamCall));
        if (!e->one_stream) { HD_HIP(hipEventRecord(sl.ev_params, e->qc)); HD_HIP(hipStreamWaitEvent(qa, sl.ev_params, 0)); }
    }
    for (uint32_t s = 0; s < S; ++s) {
        StreamHost& st = e->st[s];
        if (!st.taps_dirty) continue;
        HD_HIP(hipMemcpyAsync(e->lp_taps.p + (size_t)s * e->taps_cap, st.lp.taps.data(), st.lp.taps.size() * 4, hipMemcpyHostToDevice, qa));
        st.taps_dirty = false;
        e->lpf_k_stale = true;
    }
    if (e->sym_dirty) {
        uint32_t mr = 4;
        for (uint32_t s = 0; s < S; ++s) {
            e->h_sym.p[s] = symbol_params(e, e->st[s]);
            e->h_sym.p[s].reset = e->st[s].sym_reset ? 1u : 0u;
            mr = std::max(mr, e->h_sym.p[s].R);
        }
        e->max_R = mr;
        { uint32_t lo = mr; for (uint32_t s2 = 0; s2 < S; ++s2) lo = std::min(lo, std::max<uint32_t>(e->h_sym.p[s2].R, 4)); e->min_R = lo; }
        HD_HIP(hipMemcpyAsync(e->d_sym.p, e->h_sym.p, S * sizeof(hd::SymbolParams), hipMemcpyHostToDevice, qa));
        e->sym_dirty = false;
        for (uint32_t s = 0; s < S; ++s)       // a reset flag is consumed by exactly one call: upload again without it next time
            if (e->st[s].sym_reset) { e->st[s].sym_reset = false; e->sym_dirty = true; }
    }
    // ---- front half on qa: decimation, DC blocker, spectrum
    const float2* iq = static_cast<const float2*>(d_iq);
    float2* fcur = e->fbuf[e->cur].p;
    float2* fnext = e->fbuf[(e->cur + 1) % 3].p;
    const hd::StreamCall* dcall = host_params ? sl.h_call.dev : sl.d_call.p;
    float2* d1 = (e->calls % 3 == 0) ? e->dec1.p : (e->calls % 3 == 1) ? e->dec1b.p : e->dec1c.p;   // never a buffer a call in flight still reads
    const int hin = e->hist_cur, hout = e->hist_cur ^ 1;
    // spectrum collection rides in the final stage's epilogue unless the DC blocker must see the samples first
    float2* feed = (e->cfg.enable_spectrum && !any_dc) ? e->fft_in.p : nullptr;
    const int cin = e->carry_cur, cout = e->carry_cur ^ 1;
    auto fill_tail = [&](hd::TailArgs& t) {      // buffers of THIS call's stream tails (the LDS carve is already in t)
        t.dec1 = d1; t.dec1_stride = e->n1_cap; t.hist2_in = e->hist2[hin].p; t.hist2_out = e->hist2[hout].p; t.taps2 = e->stage_taps[1].p;
        t.fbuf = fcur; t.fbuf_next = fnext; t.fbuf_stride = e->fbuf_stride; t.fir_hist_cap = e->fir_hist_cap;
        t.lp_taps = e->lp_taps.p; t.taps_stride = e->taps_cap; t.demod = e->demod.p; t.demod_stride = e->demod.n / S;
        t.filtered = e->cfg.keep_filtered ? e->filtered.p : nullptr; t.carry_in = e->carry[cin].p; t.carry_out = e->carry[cout].p;
        t.call = dcall; t.fft_in = feed; t.head_buf = e->fir_head.p; t.fbuf_prev = e->fbuf[(e->cur + 2) % 3].p; t.head_cap = e->head_cap;
        t.n_streams = S;
        t.ring = e->tail.p; t.ring_cap = e->tail_cap; t.sym = e->d_symstate.p; t.flipmask = e->flipmask.p; t.wsum = e->weight.p;
        t.sp = e->d_sym.p; t.slots = sl.h_slots.dev; t.slot_words = e->slot_words; t.seq = sl.seq;
        t.flips_dbg = e->flips_cap ? e->flips_dbg.p : nullptr; t.flips_cap = e->flips_cap;
        const bool in_tail = e->tail_fft && e->cfg.enable_spectrum && feed;  // the tail transforms a completed buffer itself
        t.fft_tw = in_tail ? e->fft_tw.p : nullptr; t.spec = e->spec.p; t.power = e->power.p; t.stats = sl.h_stats.dev; t.rate = e->fsd; t.bins_sep = e->bins_sep;
    };
    // Equally sized pushes through a single-wave first stage of a two-stage plan: the stage-1 workgroups (eight resident per CU) draw
    // runs of tiles from per-XCD counters (kernels/decimate.hip) -- no cold start per run, no fixed shares that end ragged.
    auto make_claim = [&](uint32_t lin_wgs /* stage 1 as a launch of its own: the workgroup count of its linear split (which needs four tiles per workgroup); 0 = step launch */,
                          uint32_t run_len_cu = 0 /* != 0: runs for a per-CU ring kernel's loaders */) {
        hd::StepClaim claim{};
        // (tiles: 64 outputs of a single-wave /32 or /64 first stage; for the per-CU ring kernels hd::ring_tiles -- 64 rows of 32 samples advancing by 57 or 58 at /32, 2048 input samples at the smaller ratios)
        const uint32_t ntiles = run_len_cu ? HDK(ring_tiles, (int)R1, (int)T1, max_in) : (max_n1 + 63) / 64, n_xcd = e->n_cus / 32u, run_len = run_len_cu ? run_len_cu : e->step_run >= 2 ? e->step_run : 4u;   // (a run must hold the tile in front of which the next draw is issued: at least two; four re-read fewer halos than two)
        const uint64_t runs = (uint64_t)S * ntiles / run_len;
        const bool shape_ok = run_len_cu ? (HDK(stage1_cu_supported, (int)R1, (int)T1) && max_in % 2048u == 0) : ((R1 == 32 || R1 == 64) && max_n1 % 64 == 0);
        if (!e->no_claim && (nst == 2 || (run_len_cu && nst == 1)) && shape_ok && min_in == max_in && max_in && !any_zero1 && n_xcd && e->n_cus % 32u == 0 && n_xcd <= 16 &&
            ntiles && (ntiles % run_len == 0 || (run_len_cu && R1 >= 32 && ((uint64_t)S * ntiles) % run_len == 0)) && runs % n_xcd == 0 && (uint64_t)S * ntiles < (1ull << 32) && (uint64_t)ntiles * S >= 4ull * lin_wgs) {   // (the two counter sets alternate: a launch that takes one must really run that way)
            claim.ctr = e->step_ctr.p + (size_t)(e->step_launches & 1u) * 16 * 32;
            claim.ctr_next = e->step_ctr.p + (size_t)((e->step_launches & 1u) ^ 1u) * 16 * 32;
            claim.n_xcd = n_xcd; claim.runs_per_xcd = (uint32_t)(runs / n_xcd); claim.run_len = run_len;
            if (run_len_cu && R1 >= 32 && e->ring_short_pct) {
                // the worker waves' guided hand-out (stage1_ring.h): the last ring_short_pct per cent of an XCD's tiles go out as single tiles
                const uint32_t tpx = claim.runs_per_xcd * run_len, whole = (uint32_t)((uint64_t)tpx * (100u - e->ring_short_pct) / 100u) / run_len;
                claim.tiles_per_xcd = tpx; claim.short_from = whole; claim.runs_per_xcd = whole + (tpx - whole * run_len);
            }
            ++e->step_launches;
        }
        return claim;
    };
    // Tiles per drawn run of the per-CU ring kernels: consecutive tiles of one stream (the rows neighbouring tiles share are re-read from the XCD's L2), so it
    // must divide the stream's tile count, and the launch's runs must divide among the XCDs.  The /32 stages' worker waves draw their own runs and the
    // last runs of a launch are its ragged end: four measured best among 3 / 4 / 6 / 9 / 12 (0.1360-0.1368 ms per step against 0.1386 with nine, one box,
    // alternating); the loader / consumer kernels of the smaller ratios keep eight (round 4).  The nearest usable length, the longer one first.
    auto pick_ring_run = [&](const uint32_t ntiles) -> uint32_t {
        const uint32_t n_xcd = e->n_cus / 32u ? e->n_cus / 32u : 1u, want = R1 >= 32 ? 4u : 8u;
        // (the worker waves of the /32 and /64 stages walk a run on into the next stream: it need not divide a stream's tiles, only the slab's)
        auto ok = [&](uint32_t r) { return r >= 2u && r <= ntiles && (R1 >= 32 ? ((uint64_t)S * ntiles) % r == 0 : ntiles % r == 0) && ((uint64_t)S * ntiles / r) % n_xcd == 0; };
        if (e->ring_run >= 2 && ok(e->ring_run)) return e->ring_run;
        for (uint32_t d = 0; d <= 8u; ++d) {
            if (ok(want + d)) return want + d;
            if (d && d < want && ok(want - d)) return want - d;
        }
        return 2u;                                     // (k_step treats shorter runs as "not drawn" while the counter sets have already alternated; make_claim refuses what does not divide)
    };
    if (step) {
        // One launch: [tails of the previous call | this call's stage 1].  Stage 1 reads its parameters from the mapped host block and
        // leaves the device copy the tails (next launch) and the spectrum commit read.
        // Where the per-CU step kernel can serve the plan and the sizes, this call's tails are laid out for ITS slice of LDS -- a quarter of what four
        // worker slots leave of the CU's 160 KB (23 KB: larger caches for the search phase than the 20 KB slot of the single-wave fallback) -- now; the
        // launch that runs them decides.
        // (... and only where the per-CU kernel's runs can be drawn at all for this stream count -- an odd count never divides among the XCDs --: such
        // batches keep the 20 KB layout and ride in k_step as before.  What is left for the fallback further down are the transitions: a launch that cannot
        // be the per-CU kernel although the previous call expected it.)
        const uint32_t ring_nt = HDK(ring_tiles, (int)R1, (int)T1, max_in), ring_run0 = pick_ring_run(ring_nt), n_xcd0 = e->n_cus / 32u;
        const bool runs_ok = ring_nt && n_xcd0 && ring_run0 >= 2u && ((uint64_t)S * ring_nt) % ring_run0 == 0 && ((uint64_t)S * ring_nt / ring_run0) % n_xcd0 == 0;
        const bool cu_shape = !e->no_cu_step && !e->no_claim && !any_zero1 && max_in % 2048u == 0 && runs_ok && HDK(step_cu_supported, (int)R1, (int)T1, (int)R2, (int)T2);
        const uint32_t cu_tail = cu_shape ? HDK(step_cu_tail_lds, (int)R1, (int)T1) : 0u;
        if (cu_tail) {
            hd::TailArgs tacu{};
            if (HDK(tail_layout, tacu, 64, (int)R2, (int)T2, max_taps, e->max_R, e->min_R, e->tail_cap, max_pend, cu_tail)) ta_step = tacu;
        }
        fill_tail(ta_step);
        hd_engine::PendingTail prev = e->pend;
        hd_engine::CallSlot* ps = prev.valid ? &e->slot[prev.slot] : nullptr;
        // Events of a k_step_cu launch ride on its dispatch packet (launch_step_cu): an event recorded behind it is a barrier packet of its own and
        // costs the queue ~5 us before the next launch starts.  (Not where the spectra are a launch of their own: ev_done follows that one.)
        const bool ext_events = true;
        bool ev_on_dispatch = false;
        uint32_t wgs = 32u * e->n_cus;                          // short runs of tiles: the dispatcher evens out the tail of the launch
        // One workgroup per CU (four stage-1 worker waves, the tails in the other four) where the plan and the sizes allow it
        const uint32_t ring_run = ring_run0;
        const int cu_exp0 = e->cu_exp;             // timing experiments only (results wrong): 1 = no tails, 2 = no stage 1
        const bool want_cu = cu_tail && ((cu_exp0 & 1) || (ta_step.lds_bytes <= cu_tail && (!prev.valid || prev.ta.lds_bytes <= cu_tail)));
        const hd::StepClaim claim = make_claim(0, want_cu ? ring_run : 0u);
        if (claim.ctr) wgs = 8u * e->n_cus;
        bool launched = false;
        e->last_timing.step_variant = 0;
        if (want_cu && claim.ctr) {
            const uint32_t tb = std::max(ta_step.lds_bytes, prev.valid ? prev.ta.lds_bytes : 0u);
            hd::StepClaim cl = claim;
            if (cu_exp0 & 2) { cl.runs_per_xcd = 0; cl.tiles_per_xcd = 0; cl.short_from = 0xFFFFFFFFu; }
            ev_on_dispatch = ext_events && (!ps || prev.ta.fft_tw || !prev.any_fft);
            if (sl.timed && !ev_on_dispatch) HD_HIP(hipEventRecord(sl.t1, qa));
            launched = HDK(launch_step_cu, qa, (int)R1, (int)T1, prev.valid ? prev.r2 : (int)R2, prev.valid ? prev.t2 : (int)T2, S, e->n_cus, iq, stride,
                                          e->hist1[hin].p, e->hist1[hout].p, e->stage_taps[0].p, d1, e->n1_cap, sl.h_call.dev, sl.d_call.p, prev.ta,
                                          (prev.valid && !(cu_exp0 & 1)) ? S : 0u, max_in, cl, tb,
                                          ev_on_dispatch && sl.timed ? sl.t1 : nullptr, !ev_on_dispatch ? nullptr : sl.timed ? sl.t2 : ps ? ps->ev_done : nullptr);
            if (!launched && ev_on_dispatch) { ev_on_dispatch = false; if (sl.timed) HD_HIP(hipEventRecord(sl.t1, qa)); }
            e->last_timing.step_variant = launched ? 1u : 0u;
        }
        else if (sl.timed) HD_HIP(hipEventRecord(sl.t1, qa));
        if (!launched) {
            hd::StepClaim fb = claim;
            if (want_cu && claim.ctr) {                        // (runs cut for the ring kernel's tiles are not k_step's: fixed shares, and the counter set was not drawn from)
                --e->step_launches; fb = hd::StepClaim{};
                wgs = 32u * e->n_cus;
            }
            // The previous call's tails were laid out when THAT call was enqueued -- for k_step_cu's 23 KB slice where its shape held -- and the launch that runs
            // them is this one.  If this launch is the single-wave fallback (the pushes grew, a stream count whose runs do not divide among the XCDs, a size
            // that is not a multiple of 2048), a tail carved for more than k_step's static slot would read and write LDS past the workgroup's allocation
            // (reads return 0, writes are dropped: wrong window sums and flips, no error -- ADVICE r05).  Such tails run as a launch of their own, in front,
            // with the LDS they were laid out for; this launch then carries stage 1 only.
            if (prev.valid && prev.ta.lds_bytes > HDK(step_lds_bytes, (int)R1, (int)T1)) {
                if (const int r = run_pending_tail(e)) return r;     // (e->pend is still the previous call's; launches the tails, their spectra, records its ev_done)
                prev.valid = false; ps = nullptr;
            }
            if (!HDK(launch_step, qa, (int)R1, (int)T1, prev.valid ? prev.r2 : (int)R2, prev.valid ? prev.t2 : (int)T2, S, max_n1, iq, stride, e->hist1[hin].p,
                                 e->hist1[hout].p, e->stage_taps[0].p, d1, e->n1_cap, sl.h_call.dev, sl.d_call.p, wgs, prev.ta, prev.valid ? S : 0u,
                                 any_zero1 ? 0u : max_in, fb))
                return fail(HD_ERR_INVALID, "no step kernel for this decimation plan");
        }
        if (sl.timed && !ev_on_dispatch) HD_HIP(hipEventRecord(sl.t2, qa));
        if (ps) {
            if (!prev.ta.fft_tw) { if (const int r = run_spectrum(e, qa, *ps, prev.any_fft)) return r; }
            if (!ev_on_dispatch || sl.timed) HD_HIP(hipEventRecord(ps->ev_done, qa));    // (a timed launch's own signal is its t2)
        }
        e->pend.valid = true; e->pend.ta = ta_step; e->pend.slot = (int)(e->calls % hd_engine::kSlots); e->pend.any_fft = any_fft;
        e->pend.r2 = (int)R2; e->pend.t2 = (int)T2; e->pend_max_taps = max_taps;
        HD_HIP(hipGetLastError());
        sl.busy = true;
        e->cur = (e->cur + 1) % 3;
        e->carry_cur ^= 1;
        e->hist_cur ^= 1;
        ++e->calls;
        e->last_timing.host_enqueue_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
        // Step launches run strictly in order on one queue, so how far the host lags behind with the results is only a matter of the
        // host-visible slots: pipeline = 2 keeps three calls undelivered instead of two -- results one call later still, and a host that
        // is held up for a whole launch (a descheduled thread, a burst of sentences) no longer leaves the queue empty.
        const uint64_t depth = e->cfg.pipeline >= 2 ? 3u : 2u;
        int rc = HD_OK;
        while (e->calls - e->delivered > depth) {
            const int r = collect(e, e->slot[e->delivered % hd_engine::kSlots]);
            if (r) rc = r;
            ++e->delivered;
        }
        return rc;
    }
    if (nst == 0) {
        if (sl.timed) HD_HIP(hipEventRecord(sl.t1, qa));
        HDK(launch_passthrough, qa, S, max_in, iq, stride, fcur, e->fbuf_stride, dcall, e->fir_hist_cap);
        if (sl.timed) HD_HIP(hipEventRecord(sl.t2, qa));
    } else {
        const bool single = nst == 1;
        float2* out1 = single ? fcur : d1;
        const size_t out1_stride = single ? e->fbuf_stride : e->n1_cap;
        // (a /64 workgroup holds 36.7 KB of LDS and ~400 VGPRs: four fit a CU, and six per CU would be a launch of one round and a half --
        // measured at 10 MS/s, /256: 1.73 against 1.84 ms per step)
        const uint32_t wgs_cu = (R1 == 64 && !e->dec_wgs_forced && e->dec_wgs_per_cu > 4u) ? 4u : e->dec_wgs_per_cu;
        const uint32_t lin1 = (min_in == max_in && max_in) ? wgs_cu * e->n_cus : 0u;
        // A /32 first stage over equally sized pushes: one workgroup per CU, LDS-DMA loader waves + computing waves (k_stage1_cu, stage1_ring.h)
        bool s1_cu = false;
        if ((single ? R1 == 4 : R1 != 4) && min_in == max_in && max_in && !e->no_cu_step && !any_zero1 && max_in % 2048u == 0 && HDK(stage1_cu_supported, (int)R1, (int)T1)) {
            const hd::StepClaim cl = make_claim(0, pick_ring_run(HDK(ring_tiles, (int)R1, (int)T1, max_in)));
            if (cl.ctr) {
                if (sl.timed) HD_HIP(hipEventRecord(sl.t1, qa));
                s1_cu = HDK(launch_stage1_cu, qa, (int)R1, (int)T1, e->n_cus, iq, stride, e->hist1[hin].p, e->hist1[hout].p, e->stage_taps[0].p, out1, out1_stride,
                                             max_in, cl, e->ring_gave_up.dev, hd_engine::kS1Loaders, 16u, (e->cfg.pipeline && !fuse && !single) ? hd_engine::kS1Slots : 8u,   // (/4 as the only stage is bound by the vector pipes: eight slots, 0.838 against 0.904 ms per step with four)
                                             single ? (lean ? sl.h_call.dev : dcall) : nullptr, e->fir_hist_cap, single ? feed : nullptr);
                if (!s1_cu) --e->step_launches;                // (the counter sets alternate per launch that really draws: this one did not)
            }
        }
        e->last_timing.step_variant = s1_cu ? 1u : 0u;
        const hd::StepClaim claim1{};                      // (stage 1 alone is HBM-bound: drawing runs there cost 3 % -- more halo re-reads -- where the step launch gains 4 %: fixed shares)
        if (sl.timed && !s1_cu) HD_HIP(hipEventRecord(sl.t1, qa));
        if (!s1_cu)
        if (!HDK(launch_decimate, qa, R1, T1, S, max_n1, iq, stride, e->hist1[hin].p, e->hist1[hout].p, e->stage_taps[0].p, out1, out1_stride,
                                 (lean || free_front) ? sl.h_call.dev : dcall, 0, single ? 1 : 0, e->fir_hist_cap, single ? feed : nullptr, lin1,
                                 nullptr, claim1.ctr ? max_in : 0u, claim1))
            return fail(HD_ERR_INVALID, "no kernel for this decimation stage");
        if (sl.timed) HD_HIP(hipEventRecord(sl.t2, qa));
        if (free_front) HDK(launch_fetch_params, qa, sl.h_call.dev, sl.d_call.p, S * sizeof(hd::StreamCall));
        if (!single && !fuse) {
            // (Round 4 measured the second stage as a ring kernel too -- /2 with sixteen outputs per lane, /4 with eight: bit-identical, and SLOWER in the
            // pipelined step (/16: 0.338-0.342 against 0.326-0.338 ms): a ring kernel takes a CU's whole LDS, and the back half of the previous call
            // can no longer share the CU with it.  The classic grid stays.  NOTES.md.)
            if (!HDK(launch_decimate, qa, R2, T2, S, max_n2, d1, e->n1_cap, e->hist2[hin].p, e->hist2[hout].p, e->stage_taps[1].p, fcur,
                                     e->fbuf_stride, dcall, 1, 1, e->fir_hist_cap, feed))
                return fail(HD_ERR_INVALID, "no kernel for this decimation stage");
        }
    }
    auto spectrum = [&](hipStream_t q) -> int {
        if (!(e->cfg.enable_spectrum && max_n2)) return HD_OK;
        if (!fuse && (any_dc || nst == 0)) HDK(launch_fft_feed, q, S, fcur, e->fbuf_stride, e->fft_in.p, dcall, e->fir_hist_cap);
        if (any_fft) {   // only when some stream's 4096-sample buffer completed (every call at >= 4096 decimated samples per push)
            if (const int r = transform_and_commit(e, q, sl.h_stats.dev, dcall, sl.seq, fcur)) return r;
        }
        return HD_OK;
    };
    // Where every spectrum of the call is read in place out of this call's chunk (fft_run == 2 for all of them), the spectrum launch hangs on nothing but that
    // chunk: it goes to the third queue behind the front half's event, off the queue a step waits for (round 5 could not: the next call's second stage refilled
    // the collection buffer a transform on another queue was still reading -- there is no collection buffer on this route).
    bool spectrum_on_qc = false;
    if (!fuse) {
        if (any_dc) HDK(launch_dc_remove, qa, S, fcur, e->fbuf_stride, dcall, e->fir_hist_cap);
        if (path == 0 && !e->one_stream && e->own_fft && e->cfg.enable_spectrum && !any_dc && any_fft && nst != 0) {
            spectrum_on_qc = true;
            for (uint32_t s = 0; s < S && spectrum_on_qc; ++s) spectrum_on_qc = sl.h_call.p[s].fft_run != 1u;
        }
        if (!spectrum_on_qc) { if (const int r = spectrum(qa)) return r; }
    }
    if (!e->one_stream) HD_HIP(hipEventRecord(sl.ev_front, qa));
    if (spectrum_on_qc) {
        HD_HIP(hipStreamWaitEvent(e->qc, sl.ev_front, 0));
        if (const int r = spectrum(e->qc)) return r;
        HD_HIP(hipEventRecord(sl.ev_spec, e->qc));
    }
    // ---- back half on qb: [stage 2 +] low-pass + discriminator + buffer slide, [spectrum,] symbol extractor, results.  It may
    // still be running when the NEXT call's front half starts on qa: the two halves touch disjoint buffers (DESIGN.md
    // "two-stream pipeline").
    if (!e->one_stream) HD_HIP(hipStreamWaitEvent(qb, sl.ev_front, 0));
    bool done_on_dispatch = false;
    const bool ev_ride = !sl.timed && e->one_stream;   // (two queues: the event is not on the step's critical path, and /16 measured no better with it on the dispatch)
    if (tail) {
        if (lean) HDK(launch_fetch_params, qb, sl.h_call.dev, sl.d_call.p, S * sizeof(hd::StreamCall));
        fill_tail(ta);
        // (the call's completion event rides on the tails' dispatch where nothing is launched behind them)
        done_on_dispatch = ev_ride && (ta.fft_tw || !(e->cfg.enable_spectrum && max_n2));
        if (!HDK(launch_tail, qb, tail_lanes, (int)R2, (int)T2, S, ta, done_on_dispatch ? sl.ev_done : nullptr)) return fail(HD_ERR_INVALID, "stream tail refused a shape it was selected for");
        if (!ta.fft_tw) { if (const int r = spectrum(qb)) return r; }
    } else if (fuse) {
        if (lean) HDK(launch_fetch_params, qb, sl.h_call.dev, sl.d_call.p, S * sizeof(hd::StreamCall));
        if (!HDK(launch_backend, qb, (int)R2, (int)T2, S, max_n1, max_n2, max_taps, d1, e->n1_cap, e->hist2[hin].p, e->hist2[hout].p,
                                e->stage_taps[1].p, fcur, fcur, fnext, e->fbuf_stride, e->fir_hist_cap, e->lp_taps.p, e->taps_cap, e->demod.p,
                                e->demod.n / S, e->cfg.keep_filtered ? e->filtered.p : nullptr, e->carry[cin].p, e->carry[cout].p, dcall, feed,
                                e->tail.p, e->tail_cap, e->d_symstate.p, e->fir_head.p, e->head_cap, e->fbuf[(e->cur + 2) % 3].p))
            return fail(HD_ERR_INVALID, "fused back end refused a shape it was selected for");
        if (const int r = spectrum(qb)) return r;
    } else {
        // Fast mode, a long low-pass (configs[4]: 4097 taps): filter through transforms -- [history | inputs | zeros] of N per stream, X conj(K), back -- and hand
        // k_fir_demod the filtered samples (it still does the discriminator, the ring append, the carries, the slide).
        uint32_t lpN = 0;
        // (Not the run that starts a stream's filter from zeros: a transform's rounding is ~1e-6 of the input's PEAK everywhere, and the start-up transient -- outputs
        // that are exactly zero or tiny in the direct sum -- would come out as noise at that floor, which the discriminator turns into arbitrary phases.)
        bool any_lp_restart = false;
        for (uint32_t s = 0; s < S && !any_lp_restart; ++s) any_lp_restart = sl.h_call.p[s].fir_m && sl.h_call.p[s].fir_zero_hist;
        if (e->fast && e->lpf_N[1] && max_taps >= hd_engine::kLpFftMinTaps && max_m && !any_lp_restart) {
            const uint32_t Lmax = max_m + max_taps - 1;
            const int w = Lmax <= e->lpf_N[0] ? 0 : Lmax <= e->lpf_N[1] ? 1 : -1;
            if (w >= 0) {
                lpN = e->lpf_N[w];
                rocfft_execution_info_set_stream(e->lpf_info, qb);
                if (e->lpf_k_stale || e->lpf_k_N != lpN) {           // the taps' spectra, once per design and transform length (a design change has drained the pipeline)
                    for (uint32_t s = 0; s < S; ++s) e->lpf_ntaps.p[s] = (uint32_t)e->st[s].lp.taps.size();
                    HDK(launch_lp_taps_gather, qb, S, e->lp_taps.p, e->taps_cap, e->lpf_ntaps.dev, e->lpf_k.p, lpN);
                    void* kb[1] = {e->lpf_k.p};
                    if (rocfft_execute(e->lpf_fwd[w], kb, nullptr, e->lpf_info) != rocfft_status_success) return fail(HD_ERR_DEVICE, "rocfft_execute (low-pass taps) failed");
                    e->lpf_k_N = lpN; e->lpf_k_stale = false;
                }
                HDK(launch_lp_gather, qb, S, fcur, e->fbuf_stride, e->lpf_x.p, lpN, dcall, e->fir_hist_cap, e->fir_head.p, e->head_cap, e->fbuf[(e->cur + 2) % 3].p);
                void* xb[1] = {e->lpf_x.p};
                if (rocfft_execute(e->lpf_fwd[w], xb, nullptr, e->lpf_info) != rocfft_status_success) return fail(HD_ERR_DEVICE, "rocfft_execute (low-pass forward) failed");
                HDK(launch_lp_mul, qb, S, e->lpf_x.p, e->lpf_k.p, lpN, dcall);
                if (rocfft_execute(e->lpf_inv[w], xb, nullptr, e->lpf_info) != rocfft_status_success) return fail(HD_ERR_DEVICE, "rocfft_execute (low-pass inverse) failed");
                ++e->lpf_calls;
            }
        }
        HDK(launch_fir_demod, qb, S, max_m, lpN ? 0u : max_taps, fcur, e->fbuf_stride, e->lp_taps.p, e->taps_cap, e->demod.p, e->demod.n / S,
                             e->cfg.keep_filtered ? e->filtered.p : nullptr, e->carry[cin].p, e->carry[cout].p, dcall, e->fir_hist_cap,
                             e->tail.p, e->tail_cap, e->d_symstate.p, fnext, e->fir_head.p, e->head_cap, e->fbuf[(e->cur + 2) % 3].p, e->demod_ck_acc.p,
                             lpN ? e->lpf_x.p : nullptr, lpN, lpN ? 1.0f / (float)lpN : 1.0f);
    }
    if (!tail)
    HDK(launch_symbols, qb, S, max_m, max_new, e->max_R, e->tail.p, e->tail_cap, e->d_symstate.p, e->flipmask.p, e->weight.p, e->d_sym.p,
                       dcall, sl.h_slots.dev, e->slot_words, e->flips_cap ? e->flips_dbg.p : nullptr, e->flips_cap, e->min_R, sl.seq, ev_ride ? sl.ev_done : nullptr,
                       fuse ? nullptr : e->demod_ck_acc.p);      // (the fused back end, k_backend, leaves no checksum: its slots say demod_n = 0xFFFFFFFF)
    if (!tail && ev_ride) done_on_dispatch = true;
    if (sl.timed) HD_HIP(hipEventRecord(sl.t3, qb));
    if (spectrum_on_qc) HD_HIP(hipStreamWaitEvent(qb, sl.ev_spec, 0));      // the call is complete when its spectra are
    if (!done_on_dispatch) HD_HIP(hipEventRecord(sl.ev_done, qb));
    HD_HIP(hipGetLastError());
    sl.busy = true;
    e->cur = (e->cur + 1) % 3;
    e->carry_cur ^= 1;
    if (max_in && nst) e->hist_cur ^= 1;
    ++e->calls;
    e->last_timing.host_enqueue_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
    if (!e->cfg.pipeline) { ++e->delivered; return collect(e, sl); }
    // pipelined: keep up to two calls in flight; deliver the oldest one's results now (its host stage overlaps the GPU work
    // of the newer calls, and the GPU always has the next call queued)
    int rc = HD_OK;
    while (e->calls - e->delivered > 2) {
        const int r = collect(e, e->slot[e->delivered % hd_engine::kSlots]);
        if (r) rc = r;
        ++e->delivered;
    }
    return rc;
}

void* hd_pinned_alloc(size_t bytes)
{
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void hd_pinned_free(void* p) { if (p) (void)hipHostFree(p); }

int hd_process_host(hd_engine* e, const float* iq, size_t stride, const uint32_t* n_per_stream, uint32_t n_uniform)
{
    if (!e) return fail(HD_ERR_INVALID, "null engine");
    if (!iq) return fail(HD_ERR_INVALID, "null IQ pointer");
    std::lock_guard<std::recursive_mutex> lock(e->mtx);   // (recursive: hd_process_device below takes it again)
    if (e->in_callback) return fail(HD_ERR_INVALID, "hd_process_* cannot be called from a sentence / character callback");
    HD_HIP(hipSetDevice(e->cfg.device));
    // Synchronous delivery from page-locked, mapped memory (hd_pinned_alloc, or any hipHostMalloc'ed / registered buffer): the kernels are done when the
    // call returns, so the first decimation stage may read the caller's buffer in place over PCIe -- no staging copy in front of it.  (Pageable memory,
    // batch mode -- the caller may reuse the buffer while the call is still queued -- or a base the 16-byte loads cannot take: the copy below.)
    if (!e->cfg.pipeline && !(reinterpret_cast<uintptr_t>(iq) & 15) && !(stride & 1)) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, iq) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer) {
            ++e->host_calls_in_place;
            return hd_process_device(e, at.devicePointer, stride, n_per_stream, n_uniform);
        }
        (void)hipGetLastError();                          // (an ordinary pointer: the query's error is not the call's)
    }
    const size_t dstride = e->cfg.max_chunk;
    // validate the sizes before anything is queued (hd_process_device checks the rest and leaves every stream untouched on error)
    for (uint32_t s = 0; s < e->S; ++s)
        if ((n_per_stream ? n_per_stream[s] : n_uniform) > e->cfg.max_chunk) return fail(HD_ERR_CAPACITY, "more samples than max_chunk");
    // The copy stream is created on first use, AFTER the engine's compute streams: HIP deals streams to a handful of hardware queues in
    // creation order, and a copy stream created in between put the two compute streams on one queue (their kernels took turns).
    if (!e->qh) HD_HIP(hipStreamCreateWithFlags(&e->qh, hipStreamNonBlocking));
    const int j = (int)(e->host_calls & 1u);
    if (j == 1 && !e->staging2.p) {
        HD_HIP(e->staging2.alloc((size_t)e->S * e->cfg.max_chunk));
        HD_HIP(hipDeviceSynchronize());      // the allocation's memset runs on the null stream: it must not trail the copy below
    }
    float2* dst = j ? e->staging2.p : e->staging.p;
    if (e->staging_used[j]) HD_HIP(hipStreamWaitEvent(e->qh, e->ev_staging_free[j], 0));    // the call that read this slab two calls ago is past its stage 1
    if (!n_per_stream) {
        if (n_uniform)
            HD_HIP(hipMemcpy2DAsync(dst, dstride * sizeof(float2), iq, stride * sizeof(float2), (size_t)n_uniform * sizeof(float2),
                                    e->S, hipMemcpyHostToDevice, e->qh));
    } else {
        for (uint32_t s = 0; s < e->S; ++s) {
            const uint32_t n = n_per_stream[s];
            if (n) HD_HIP(hipMemcpyAsync(dst + (size_t)s * dstride, iq + 2 * (size_t)s * stride, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, e->qh));
        }
    }
    HD_HIP(hipEventRecord(e->ev_copy[j], e->qh));
    HD_HIP(hipStreamWaitEvent(e->qa, e->ev_copy[j], 0));
    HD_HIP(hipEventSynchronize(e->ev_copy[j]));          // from here on the caller's buffer is his again
    const int rc = hd_process_device(e, dst, dstride, n_per_stream, n_uniform);
    HD_HIP(hipEventRecord(e->ev_staging_free[j], e->qa));
    e->staging_used[j] = true;
    ++e->host_calls;
    return rc;
}

/* ---------------------------------------------------------------- batched file ingest -------------------------- */

int hd_ingest_run(hd_engine* e, hd_host_iqfiles* src, uint64_t max_rounds, uint64_t* samples_done)
{
    if (samples_done) *samples_done = 0;
    if (!e || !src) return fail(HD_ERR_INVALID, "null engine or file batch");
    const uint32_t S = e->S;
    if (src->batch.streams() != S) return fail(HD_ERR_INVALID, "the file batch must hold one file per stream");
    if (src->batch.chunk() > e->cfg.max_chunk) return fail(HD_ERR_CAPACITY, "file batch chunk larger than max_chunk");
    HD_HIP(hipSetDevice(e->cfg.device));
    constexpr int kSlabs = 4;      // round r's slab is refilled for round r+4: by then call r is delivered even with two calls in flight
    const size_t stride = src->batch.chunk();
    struct Slab { float* iq = nullptr; std::vector<uint32_t> n; uint32_t alive = 0; };
    Slab slab[kSlabs];
    auto release = [&] { for (auto& b : slab) if (b.iq) (void)hipHostFree(b.iq); };
    for (auto& b : slab) {
        if (hipHostMalloc(reinterpret_cast<void**>(&b.iq), (size_t)S * stride * sizeof(float2), hipHostMallocDefault) != hipSuccess) {
            release();
            return fail(HD_ERR_DEVICE, "pinned slab allocation failed");
        }
        b.n.assign(S, 0);
    }
    std::mutex m;
    std::condition_variable cv;
    uint64_t filled = 0, consumed = 0;       // rounds read by the reader / handed back by the pump
    bool stop = false;
    src->batch.set_min_take(hd_min_chunk(e->D));      // a file tail shorter than the stage histories waits for the next round
    const bool looping = src->batch.looping();
    std::thread reader([&] {
        for (uint64_t r = 0; r < max_rounds; ++r) {       // never read past the stop: a later hd_ingest_run resumes where this one ended
            {
                std::unique_lock<std::mutex> l(m);
                cv.wait(l, [&] { return stop || r < consumed + (uint64_t)kSlabs - 2; });   // two slabs may still belong to calls in flight
                if (stop) return;
            }
            Slab& b = slab[r % kSlabs];
            b.alive = src->batch.next(b.iq, stride, b.n.data());
            {
                std::lock_guard<std::mutex> l(m);
                filled = r + 1;
            }
            cv.notify_all();
            if (!b.alive && !looping) return;
        }
    });
    int rc = HD_OK;
    uint64_t total = 0;
    for (uint64_t r = 0; r < max_rounds; ++r) {
        {
            std::unique_lock<std::mutex> l(m);
            cv.wait(l, [&] { return filled > r; });
        }
        Slab& b = slab[r % kSlabs];
        if (!b.alive) {
            if (!looping) break;
            // looping files of equal length a whole number of chunks long: the reference's one empty read before the rewind
            { std::lock_guard<std::mutex> l(m); consumed = r + 1; }
            cv.notify_all();
            continue;
        }
        bool uniform = true;
        for (uint32_t s = 1; s < S; ++s) uniform = uniform && b.n[s] == b.n[0];
        rc = hd_process_host(e, b.iq, stride, uniform ? nullptr : b.n.data(), uniform ? b.n[0] : 0);
        if (rc) break;
        for (uint32_t s = 0; s < S; ++s) total += b.n[s];
        {
            std::lock_guard<std::mutex> l(m);
            consumed = r + 1;
        }
        cv.notify_all();
    }
    {
        std::lock_guard<std::mutex> l(m);
        stop = true;
    }
    cv.notify_all();
    reader.join();
    const int rf = hd_flush(e);      // nothing may still copy from the slabs when they are freed
    HD_HIP(hipDeviceSynchronize());
    release();
    if (samples_done) *samples_done = total;
    return rc ? rc : rf;
}

/* ---------------------------------------------------------------- results -------------------------------------- */

static size_t copy_out(const std::string& s, char* buf, size_t cap)
{
    if (buf && cap) { const size_t n = std::min(cap - 1, s.size()); std::memcpy(buf, s.data(), n); buf[n] = 0; }
    return s.size();
}
static size_t take_out(std::string& s, char* buf, size_t cap)
{
    const size_t n = copy_out(s, buf, cap);
    if (buf && cap > n) s.clear();
    return n;
}

size_t hd_stream_rtty(hd_engine* e, uint32_t s, char* buf, size_t cap)
{ if (check_stream(e, s)) return 0; std::lock_guard<std::recursive_mutex> l(e->mtx); return copy_out(e->st[s].text.stream, buf, cap); }
size_t hd_stream_last_sentence(hd_engine* e, uint32_t s, char* buf, size_t cap)
{ if (check_stream(e, s)) return 0; std::lock_guard<std::recursive_mutex> l(e->mtx); return copy_out(e->st[s].text.last_sentence, buf, cap); }
size_t hd_stream_take_sentences(hd_engine* e, uint32_t s, char* buf, size_t cap)
{ if (check_stream(e, s)) return 0; std::lock_guard<std::recursive_mutex> l(e->mtx); return take_out(e->st[s].text.ok_log, buf, cap); }
size_t hd_stream_take_matches(hd_engine* e, uint32_t s, char* buf, size_t cap)
{ if (check_stream(e, s)) return 0; std::lock_guard<std::recursive_mutex> l(e->mtx); return take_out(e->st[s].text.match_log, buf, cap); }
size_t hd_stream_take_chars(hd_engine* e, uint32_t s, char* buf, size_t cap)
{ if (check_stream(e, s)) return 0; std::lock_guard<std::recursive_mutex> l(e->mtx); return take_out(e->st[s].text.char_log, buf, cap); }

int hd_stream_afc(hd_engine* e, uint32_t s, hd_afc_info* out)
{
    if (int r = check_stream(e, s)) return r;
    if (!out) return fail(HD_ERR_INVALID, "null argument");
    std::lock_guard<std::recursive_mutex> l(e->mtx);
    const StreamHost& st = e->st[s];
    out->frequency_correction = st.afc.correction; out->shift_hz = st.afc.shift_hz;
    out->noise_floor = st.afc.noise_floor; out->noise_variance = st.afc.noise_sigma;
    out->peak_left = st.afc.gui_left; out->peak_right = st.afc.gui_right; out->spectra = st.spectra;
    return HD_OK;
}

static size_t fetch(hd_engine* e, const void* dev, size_t count, size_t elem, void* host, size_t cap)
{
    if (e->in_callback) { fail(HD_ERR_INVALID, "data getters cannot be called from a sentence / character callback (text getters can)"); return 0; }
    (void)flush_locked(e);                                  // pipelined mode: wait for the call in flight
    const size_t n = std::min(count, cap);
    if (!n || !host) return count;
    if (hipSetDevice(e->cfg.device) != hipSuccess) return 0;
    if (hipMemcpy(host, dev, n * elem, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return count;
}

size_t hd_stream_spectrum(hd_engine* e, uint32_t s, float* iq, size_t cap)
{
    if (check_stream(e, s)) return 0;
    std::lock_guard<std::recursive_mutex> l(e->mtx);
    if (!e->cfg.enable_spectrum || !e->st[s].have_spectrum) return 0;
    return fetch(e, e->spec.p + (size_t)s * hd::kFftBins, hd::kFftBins, sizeof(float2), iq, cap);
}
size_t hd_stream_power(hd_engine* e, uint32_t s, float* p, size_t cap)
{
    if (check_stream(e, s)) return 0;
    std::lock_guard<std::recursive_mutex> l(e->mtx);
    if (!e->cfg.enable_spectrum || !e->st[s].have_spectrum) return 0;
    return fetch(e, e->power.p + (size_t)s * hd::kFftBins, hd::kFftBins, sizeof(float), p, cap);
}
size_t hd_stream_demodulated(hd_engine* e, uint32_t s, float* v, size_t cap)
{
    if (check_stream(e, s)) return 0;
    std::lock_guard<std::recursive_mutex> l(e->mtx);
    return fetch(e, e->demod.p + (size_t)s * (e->demod.n / e->S), e->st[s].last_m, sizeof(float), v, cap);
}
size_t hd_stream_decimated(hd_engine* e, uint32_t s, float* iq, size_t cap)
{
    if (check_stream(e, s)) return 0;
    std::lock_guard<std::recursive_mutex> l(e->mtx);
    const StreamHost& st = e->st[s];
    const float2* src = e->fbuf[st.last_buf].p + (size_t)s * e->fbuf_stride + e->fir_hist_cap + st.last_pend_before;
    return fetch(e, src, st.last_n2, sizeof(float2), iq, cap);
}
size_t hd_stream_filtered(hd_engine* e, uint32_t s, float* iq, size_t cap)
{
    if (check_stream(e, s)) return 0;
    std::lock_guard<std::recursive_mutex> l(e->mtx);
    if (!e->cfg.keep_filtered) return 0;
    return fetch(e, e->filtered.p + (size_t)s * (e->demod.n / e->S), e->st[s].last_m, sizeof(float2), iq, cap);
}
size_t hd_stream_bits(hd_engine* e, uint32_t s, uint8_t* bits, size_t cap)
{
    if (check_stream(e, s)) return 0;
    std::lock_guard<std::recursive_mutex> l(e->mtx);
    const StreamHost& st = e->st[s];
    for (uint32_t i = 0; i < st.last_nbits && i < cap; ++i) bits[i] = (st.last_words[i >> 5] >> (i & 31)) & 1u;
    return st.last_nbits;
}
size_t hd_stream_flips(hd_engine* e, uint32_t s, uint32_t* flips, size_t cap)
{
    if (check_stream(e, s)) return 0;
    std::lock_guard<std::recursive_mutex> l(e->mtx);
    if (!e->flips_cap) return 0;
    const size_t n = std::min<size_t>(e->st[s].last_nflips, e->flips_cap);
    return fetch(e, e->flips_dbg.p + (size_t)s * e->flips_cap, n, sizeof(uint32_t), flips, cap);
}
size_t hd_stream_fir_taps(hd_engine* e, uint32_t s, float* taps, size_t cap)
{
    if (check_stream(e, s)) return 0;
    std::lock_guard<std::recursive_mutex> l(e->mtx);
    const auto& t = e->st[s].lp.taps;
    if (taps) std::memcpy(taps, t.data(), std::min(cap, t.size()) * sizeof(float));
    return t.size();
}
uint64_t hd_stream_bits_total(hd_engine* e, uint32_t s)
{
    if (check_stream(e, s)) return 0;
    std::lock_guard<std::recursive_mutex> l(e->mtx);
    return e->st[s].bits_total;
}
uint64_t hd_stream_flip_list_full(hd_engine* e, uint32_t s)
{
    if (check_stream(e, s)) return 0;
    std::lock_guard<std::recursive_mutex> l(e->mtx);
    return e->st[s].flip_list_full;
}
int hd_stream_demod_checksum(hd_engine* e, uint32_t s, uint64_t* call_index, uint32_t* n, uint32_t ck[2])
{
    if (check_stream(e, s)) return HD_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> l(e->mtx);
    const StreamHost& st = e->st[s];
    if (call_index) *call_index = st.demod_ck_call;
    if (n) *n = st.demod_ck_n;
    if (ck) { ck[0] = st.demod_ck[0]; ck[1] = st.demod_ck[1]; }
    return HD_OK;
}
int hd_stream_demod_checksum_total(hd_engine* e, uint32_t s, uint64_t* calls, uint64_t* calls_without, uint64_t* hash)
{
    if (check_stream(e, s)) return HD_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> l(e->mtx);
    const StreamHost& st = e->st[s];
    if (calls) *calls = st.demod_ck_calls;
    if (calls_without) *calls_without = st.demod_ck_unknown;
    if (hash) *hash = st.demod_ck_hash;
    return HD_OK;
}
uint32_t hd_stream_symbol_backlog(hd_engine* e, uint32_t s)
{
    if (check_stream(e, s)) return 0;
    std::lock_guard<std::recursive_mutex> l(e->mtx);
    return e->st[s].held;
}

}  // extern "C"
