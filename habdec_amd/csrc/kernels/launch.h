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
    // Worker-wave kernels (stage1_ring.h: ring_worker) only -- guided hand-out: tickets [0, short_from) are runs of run_len tiles, the tickets behind them
    // are SINGLE tiles, so that the end of a launch is ragged by one tile's time instead of one run's (runs_per_xcd counts the tickets of both kinds;
    // tiles_per_xcd = short_from * run_len + (runs_per_xcd - short_from)).  short_from = 0xFFFFFFFF: every ticket is a whole run.
    uint32_t short_from = 0xFFFFFFFFu, tiles_per_xcd = 0;
};

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
    float2* head_buf /* [S][head_cap]: FirHistory heads moved aside */; const float2* fbuf_prev /* the previous call's low-pass buffers (dev_types.h: FirHistory head, lazily) */; uint32_t head_cap, n_streams;
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

constexpr uint32_t kStepLdsBytes = 20480;   // LDS of a stage-1 workgroup slot (eight per CU): what a tail riding in the stage-1 launch may use

// ---- launchers that do not depend on the arithmetic mode (symbols.hip, spectrum.hip, spectrum_wave.hip)
void launch_spectrum_commit(hipStream_t st, uint32_t n_streams, const float2* raw, float2* spec, float* power,
                            SpectrumStatsDev* stats, const StreamCall* call, double rate, int bins_sep,
                            uint32_t seq /* the call's tag, stored last into every SpectrumStatsDev written (SpectrumStatsDev::seq) */);
// The transform and the commit in one launch, one wave per stream (kernels/spectrum_wave.hip); tw4096[m] = (cos, -sin)(2 pi m / 4096).
void launch_spectrum_wave(hipStream_t st, uint32_t n_streams, const float2* fft_in, const float2* tw4096, float2* spec, float* power,
                          SpectrumStatsDev* stats, const StreamCall* call, double rate, int bins_sep, uint32_t seq,
                          const float2* chunk = nullptr /* the low-pass input buffers [S][chunk_stride]: streams with StreamCall::fft_run == 2 take their 4096 samples from the head of this call's decimated chunk */,
                          size_t chunk_stride = 0, uint32_t fir_hist_cap = 0);

// ---- launchers of the mode-dependent kernels, once per arithmetic mode (arith.h)
namespace exact {
#include "launch_decls.inc"
}
namespace fast {
#include "launch_decls.inc"
}

}  // namespace hd
