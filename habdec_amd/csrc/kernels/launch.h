// Host-callable launchers of the gfx950 kernels (defined in kernels/*.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../dev_types.h"

namespace hd {

struct DemodCarry {        // per-stream discriminator carry (previous filtered sample), ping-ponged per call
    float re, im;
    uint32_t primed;       // 0 until the stream's first FIR batch (FSK2_Demod.h:35 static initialiser)
    uint32_t _pad;
};

// How the single-wave stage-1 workgroups of a launch (step launches, and stage 1 alone over equally sized pushes) get their tiles when they
// draw them (ctr != nullptr): per-XCD counters, one per 128-byte line
// (32 u32 apart), zero when the launch starts; every XCD resets its counter in ctr_next, the set the NEXT step launch uses.
struct StepClaim {
    unsigned int* ctr = nullptr;             // this launch's counters, [n_xcd][32]
    unsigned int* ctr_next = nullptr;        // the other set
    uint32_t n_xcd = 0, runs_per_xcd = 0, run_len = 0;
};
// One FIR-decimate stage for all streams.  `final_stage`: output lands behind the FIR history in the
// low-pass input buffer (offset fir_hist_cap + pend_before), else at offset 0 of `out`.
// Returns false when (ratio, ntaps) is not one of the eight reference designs.  The kernel also carries each
// stream's last ntaps-1 inputs into `hist_out` (Decimator.h:140-143, incl. the in-place quirk Q4); hist_in/hist_out
// are ping-ponged by the caller.
bool launch_decimate(hipStream_t st, int ratio, int ntaps, uint32_t n_streams, uint32_t max_out,
                     const float2* in, size_t in_stride, const float2* hist_in, float2* hist_out, const float* taps,
                     float2* out, size_t out_stride, const StreamCall* call, int stage, int final_stage,
                     uint32_t fir_hist_cap, float2* fft_in /* final stage: spectrum input buffer [S][4096], or null */,
                     uint32_t lin_wgs = 0 /* != 0: every stream has the same size; use exactly this many workgroups (single-wave kernels) */,
                     StreamCall* call_copy = nullptr /* linear split only: leave a device copy of each stream's parameters here */,
                     uint32_t uniform_n = 0 /* linear split, not the final stage: the streams' common sample count, no stream restarts its history */,
                     const StepClaim& claim = StepClaim{} /* with uniform_n: the lin_wgs workgroups draw their tiles */);
// copy `bytes` (multiple of 16) from mapped pinned host memory into device memory with a kernel
void launch_fetch_params(hipStream_t st, const void* host_mapped, void* dst, size_t bytes);
// factor 1: copy the chunk behind the FIR history.
void launch_passthrough(hipStream_t st, uint32_t n_streams, uint32_t max_n, const float2* in, size_t in_stride,
                        float2* out, size_t out_stride, const StreamCall* call, uint32_t fir_hist_cap);
void launch_dc_remove(hipStream_t st, uint32_t n_streams, float2* fbuf, size_t stride, const StreamCall* call, uint32_t fir_hist_cap);
void launch_fft_feed(hipStream_t st, uint32_t n_streams, const float2* fbuf, size_t stride, float2* fft_in,
                     const StreamCall* call, uint32_t fir_hist_cap);
void launch_spectrum_commit(hipStream_t st, uint32_t n_streams, const float2* raw, float2* spec, float* power,
                            SpectrumStatsDev* stats, const StreamCall* call, double rate, int bins_sep);
// The transform and the commit in one launch, one wave per stream (kernels/spectrum_wave.hip); tw4096[m] = (cos, -sin)(2 pi m / 4096).
void launch_spectrum_wave(hipStream_t st, uint32_t n_streams, const float2* fft_in, const float2* tw4096, float2* spec, float* power,
                          SpectrumStatsDev* stats, const StreamCall* call, double rate, int bins_sep);
void launch_fir_demod(hipStream_t st, uint32_t n_streams, uint32_t max_m, uint32_t max_taps, const float2* fbuf, size_t stride,
                      const float* taps, uint32_t taps_stride, float* demod, size_t demod_stride, float2* filtered /*or null*/,
                      const DemodCarry* carry_in, DemodCarry* carry_out, const StreamCall* call, uint32_t fir_hist_cap,
                      float* sym_ring, uint32_t ring_cap, const SymState* sym, float2* fbuf_next,
                      const float2* head_in, const uint32_t* head_n_in, float2* head_out, uint32_t* head_n_out, uint32_t head_cap /* FirHistory, dev_types.h */);
// Fused back end of a two-stage plan (second decimation stage + low-pass + discriminator + slide), one workgroup per stream;
// returns false when the stage design or the LDS footprint (see backend_lds_bytes) does not allow it -- the caller then
// runs launch_decimate(stage 2) + launch_fir_demod.  `fbuf` and `fbuf_w` are the same buffer (read: history + pending,
// written: the new decimated chunk).
size_t backend_lds_bytes(int ntaps2, uint32_t max_n1, uint32_t max_n2, uint32_t max_taps);
bool launch_backend(hipStream_t st, int ratio2, int ntaps2, uint32_t n_streams, uint32_t max_n1, uint32_t max_n2, uint32_t max_taps,
                    const float2* dec1, size_t dec1_stride, const float2* hist2_in, float2* hist2_out, const float* taps2,
                    const float2* fbuf, float2* fbuf_w, float2* fbuf_next, size_t fbuf_stride, uint32_t fir_hist_cap, const float* lp_taps,
                    uint32_t taps_stride, float* demod, size_t demod_stride, float2* filtered, const DemodCarry* carry_in,
                    DemodCarry* carry_out, const StreamCall* call, float2* fft_in, float* sym_ring, uint32_t ring_cap, const SymState* sym,
                    float2* head_buf /* [2][S][head_cap] */, uint32_t* head_cnt /* [2][S] */, uint32_t head_cap, uint32_t head_par);
// Symbol extractor: window kernel over the positions that became computable (at most max_new per stream) + scan kernel.
void launch_symbols(hipStream_t st, uint32_t n_streams, uint32_t max_m, uint32_t max_new, uint32_t max_R, const float* tail,
                    uint32_t ring_cap, SymState* sym, unsigned long long* flipmask, float* weight, const SymbolParams* sp,
                    const StreamCall* call, uint32_t* slots, uint32_t slot_words, uint32_t* flips_dbg, uint32_t flips_cap,
                    uint32_t min_R /* smallest averaging half-window over the streams: bounds the flips one call can find */,
                    uint32_t seq /* the call's tag, stored last into every result slot (BitsHeader::seq) */,
                    hipEvent_t ev_stop = nullptr /* signalled by the dispatch itself, as in launch_step_cu */);

// ---- the fused stream tail (tail_body.h / tail.hip): stage 2 + low-pass + discriminator + symbol extractor, one wave per stream
struct TailArgs {
    // stage 2 + low-pass + discriminator (same buffers and conventions as launch_backend)
    const float2* dec1; size_t dec1_stride;
    const float2* hist2_in; float2* hist2_out; const float* taps2;
    float2* fbuf; float2* fbuf_next; size_t fbuf_stride; uint32_t fir_hist_cap;
    const float* lp_taps; uint32_t taps_stride;
    float* demod; size_t demod_stride; float2* filtered;
    const DemodCarry* carry_in; DemodCarry* carry_out;
    const StreamCall* call; float2* fft_in;
    float2* head_buf; uint32_t* head_cnt; uint32_t head_cap, head_par, n_streams;
    // symbol extractor (same rings and state as launch_symbols)
    float* ring; uint32_t ring_cap; SymState* sym; unsigned long long* flipmask; float* wsum; const SymbolParams* sp;
    uint32_t* slots; uint32_t slot_words; uint32_t* flips_dbg; uint32_t flips_cap;
    uint32_t seq;                                   // the call's tag, stored last into the result slot (BitsHeader::seq)
    // spectrum of a stream whose 4096-sample buffer completes in this call, done by the tail itself when fft_tw != nullptr (spectrum_wave.h)
    const float2* fft_tw; float2* spec; float* power; SpectrumStatsDev* stats; double rate; int bins_sep;
    // LDS carve in bytes from the base of the workgroup's scratch (tail_layout)
    // (search phase, overlaying the stream windows: [header][flip list + run info][run-sum strips] at fixed offsets, then from dyn_off to lds_bytes a region
    // every stream carves for itself: the flag-mask image of its searchable backlog, a sample cache for the run sums, a window-sum cache for the edge search)
    uint32_t pend_max, f_off, v_off, ws_off, words_off, tp_off, h2_off, flips_off, fl_cap, strips_off, dyn_off, lds_bytes;
    uint32_t op;          // stage-2 outputs per lane and piece the carve was made for (tail_layout; launch_tail picks the kernel by it)
};
// Fills the LDS carve for `lanes` (64 or 256) lanes per stream; returns false when (ratio2, ntaps2) has no tail instantiation or the
// windows for max_taps / max_R do not fit into lds_limit bytes -- the caller then runs launch_backend / launch_decimate + launch_fir_demod
// and launch_symbols instead.
bool tail_layout(TailArgs& a, int lanes, int ratio2, int ntaps2, uint32_t max_taps, uint32_t max_R, uint32_t min_R, uint32_t ring_cap,
                 uint32_t pend_max /* most pending samples any stream has in front of or behind this call's low-pass run */, uint32_t lds_limit,
                 int op = 0 /* stage-2 outputs per lane and piece: 0 = the default of `lanes` (4 for 64 lanes, 1 for 256); 256 lanes also take 4 */);
bool launch_tail(hipStream_t st, int lanes, int ratio2, int ntaps2, uint32_t n_streams, const TailArgs& a, hipEvent_t ev_stop = nullptr);
// Batch mode, two-stage plans whose first stage is a single-wave design: ONE launch per step -- the stream tails of the previous call
// (workgroups [0, n_tail), arguments `ta`) in front of this call's stage 1 as a linear split over stage1_wgs workgroups; every stream
// has n_out stage-1 outputs.  Returns false when there is no instantiation for the plan.
bool launch_step(hipStream_t st, int ratio, int ntaps, int ratio2, int ntaps2, uint32_t n_streams, uint32_t n_out, const float2* in, size_t in_stride,
                 const float2* hist_in, float2* hist_out, const float* taps, float2* out, size_t out_stride, const StreamCall* call,
                 StreamCall* call_copy, uint32_t stage1_wgs, const TailArgs& ta, uint32_t n_tail,
                 uint32_t uniform_n /* != 0: the streams' common sample count, and no stream restarts its stage-1 history this call */,
                 const StepClaim& claim = StepClaim{});
// The same step as ONE 512-thread workgroup per CU (k_step_cu, decimate.hip): four stage-1 worker waves (stage1_ring.h: ring_worker -- LDS-DMA into the
// wave's own slot, systolic tap loop) and the tails in the other four waves, which become workers when they are done.  Needs a /32 first stage, equally
// sized pushes that are a multiple of 2048 samples, drawn runs (claim.ctr) and no history restart; returns false otherwise (the caller then launches
// k_step).  tail_bytes: LDS slice of one tail (<= step_cu_tail_lds).
bool launch_step_cu(hipStream_t st, int ratio, int ntaps, int ratio2, int ntaps2, uint32_t n_streams, uint32_t n_cus, const float2* in, size_t in_stride,
                    const float2* hist_in, float2* hist_out, const float* taps, float2* out, size_t out_stride, const StreamCall* call,
                    StreamCall* call_copy, const TailArgs& ta, uint32_t n_tail, uint32_t uniform_n, const StepClaim& claim, uint32_t tail_bytes,
                    hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr /* signalled by the dispatch itself (hipExtLaunchKernel), not by a packet behind it */);
// Stage 1 alone in the same shape (eight tile slots, one loader wave and seven computing waves by default): a /32 or /8 first stage, equally sized pushes that are a multiple of 2048
// samples, drawn runs, no history restart; false otherwise (the caller then launches k_decimate).  The call's parameter block is not copied.
uint32_t ring_tiles(int ratio, int ntaps, uint32_t n);   // tiles per stream and call of the per-CU ring kernels for n input samples (0: none); the drawn runs must divide it
bool step_cu_supported(int ratio, int ntaps, int ratio2, int ntaps2);   // plans k_step_cu is instantiated for: /64 (/32 212 + /2 69), /128 (/32 174 + /4 139)
bool stage1_cu_supported(int ratio, int ntaps);   // stages k_stage1_cu exists for: /32 (212, 174 taps), /8 (54 taps) as first stages; /4 (139 taps) as the only stage of a plan
bool launch_stage1_cu(hipStream_t st, int ratio, int ntaps, uint32_t n_cus, const float2* in, size_t in_stride, const float2* hist_in, float2* hist_out,
                      const float* taps, float2* out, size_t out_stride, uint32_t uniform_n, const StepClaim& claim, unsigned int* gave_up,
                      uint32_t n_loaders /* 1 or 2 */, uint32_t n_waves /* 8 .. 16 waves per workgroup: loaders + computing waves */,
                      uint32_t n_slots = 8 /* tile slots (2 per loader .. 8): fewer leave LDS for the other queue's kernels */,
                      const StreamCall* final_call = nullptr /* /4 only: the stage is the final one of a single-stage plan -- per-stream pend_before / fft_take / fft_fill */,
                      uint32_t fir_hist_cap = 0, float2* fft_in = nullptr);
uint32_t step_cu_tail_lds(int ratio, int ntaps);   // LDS a tail may use inside k_step_cu beside the four worker slots of that first stage; 0 = no such kernel
uint32_t probe_xcc_mask(hipStream_t st, uint32_t n_cus, unsigned int* d_word);   // bit x set = some workgroup of a chip-filling grid ran on XCC id x
uint32_t step_lds_bytes(int ratio, int ntaps);   // LDS of a step-launch workgroup for that first stage (its tile, at least kStepLdsBytes); 0 = no step kernel
constexpr uint32_t kStepLdsBytes = 20480;   // LDS of a stage-1 workgroup slot (eight per CU): what a tail riding in the stage-1 launch may use

}  // namespace hd
