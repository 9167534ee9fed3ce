// Structures shared by the host engine (engine.cpp) and the gfx950 kernels (kernels/*.hip).
#pragma once
#include <stdint.h>
#ifndef __HIPCC__
#ifndef __host__
#define __host__
#define __device__
#endif
#endif

namespace hd {

constexpr int kFftBins = 4096;
constexpr uint32_t kVentLimit = 30000;   // symbol-extractor overflow vent (reference SymbolExtractor.h:116)
constexpr uint32_t kFirBatch = 256;      // FIR batch granularity (reference Decoder.h:492)

// Per-stream, per-call launch parameters.  Everything here is a pure function of the sizes pushed so far,
// so the host mirrors the reference's integer bookkeeping (Decoder.h:426-436, 461-495, 532-542;
// Decimator.h:74-79; FirFilter.h:141-147, 185-194) and uploads one array of these per call.
struct StreamCall {
    uint32_t n_in;          // IQ samples consumed by stage 1 this call (multiple of the total factor)
    uint32_t n1;            // samples after stage 1 (== n_in when there are no stages)
    uint32_t n2;            // samples after the last stage = length of this call's decimated chunk
    uint32_t zero_hist1;    // Q5: stage history restarts from zeros (first call / input grew)
    uint32_t zero_hist2;
    uint32_t pend_before;   // decimated samples already pending in front of the FIR
    uint32_t fft_fill;      // samples already collected for the next spectrum
    uint32_t fft_take;      // samples of this chunk's head to append (0..n2)
    uint32_t fft_run;       // 1 = the spectrum buffer completes in this call; 2 = ... and this call's chunk alone fills it: the spectrum launch reads the chunk's head in place
                            //     (fft_take = 0: nothing is copied into the collection buffer; separate-kernels path only)
    uint32_t fir_m;         // samples to filter/demodulate this call (0 = FIR does not run)
    uint32_t fir_taps;      // tap count in use
    uint32_t fir_zero_hist; // Q5 for the low-pass
    uint32_t pend_after;    // pending decimated samples left for the next call
    uint32_t dc_remove;     // 1 = per-chunk DC blocker on the decimated chunk
    uint32_t clear_pending; // 1 = rate gate hit: drop everything pending (Decoder.h:522-527)
    uint32_t fir_taps_prev; // bits 0-15: tap count of the stream's previous low-pass run (== fir_taps unless the design changed; see FirHistory);
                            // bits 16-29: head_n, bit 30: head_prev, bit 31: head_save -- where the FirHistory head is (sc_* below)
};
// FirHistory head, lazily (round 6).  The head -- the first samples of the previous run's input -- is only read by the first run after a tap-count INCREASE, and
// the previous run's input still sits where that run read it: in the low-pass buffer of the call it ran in, which nothing overwrites for two more calls (three
// buffers take turns).  So no run writes a head any more (33 MB per step at /16, 67 at /4, 8.4 in the headline launch, for a change that may never come):
//   head_prev = 1: the stream's previous run was the PREVIOUS call -- a consumer reads head[j] = prev_buf[fir_hist_cap + j], j < head_n;
//   head_prev = 0: it was earlier -- the head was copied aside (head_save = 1) by the first call in which the stream did not run, and sits in the side buffer.
__host__ __device__ inline uint32_t sc_taps_prev(const StreamCall& c) { return c.fir_taps_prev & 0xFFFFu; }
__host__ __device__ inline uint32_t sc_head_n(const StreamCall& c) { return (c.fir_taps_prev >> 16) & 0x3FFFu; }
__host__ __device__ inline bool sc_head_prev(const StreamCall& c) { return (c.fir_taps_prev >> 30) & 1u; }
__host__ __device__ inline bool sc_head_save(const StreamCall& c) { return (c.fir_taps_prev >> 31) & 1u; }
__host__ __device__ inline uint32_t sc_pack_taps_prev(uint32_t taps_prev, uint32_t head_n, bool head_prev, bool head_save)
{ return (taps_prev & 0xFFFFu) | ((head_n & 0x3FFFu) << 16) | ((head_prev ? 1u : 0u) << 30) | ((head_save ? 1u : 0u) << 31); }

// Low-pass history across a change of the tap count.  The reference keeps ONE buffer [history (T-1) | input (m)] per filter
// (FirFilter.h:141-167): after a run its first T-1 slots hold the last T-1 inputs, the slots behind still hold the run's
// input from its first sample on.  A following run with T' != T taps simply takes the first T'-1 slots as its history:
// for T' < T the OLDEST T'-1 of the kept T-1 samples, for T' > T all T-1 of them followed by the first T'-T samples of the
// previous run's input.  The kernels rebuild exactly that: history[j] = j < Tp-1 ? kept[j] : head_prev[j - (Tp-1)], where
// `kept` are the last Tp-1 inputs (in the low-pass buffer) and `head_prev` the first samples of the previous run's input
// (a small per-stream side buffer, ping-ponged like the discriminator carry; entries past what was saved count as zero).

// Per-stream symbol-extractor constants (change only through the control plane).
struct SymbolParams {
    uint32_t spb;           // samples per bit = size_t(round(fsd / baud))       (SymbolExtractor.h:90)
    uint32_t R;             // averaging half-window = max(4, int(spb / 4))      (SymbolExtractor.h:170)
    uint32_t min_held;      // smallest backlog that satisfies `size >= fsd/baud*3` (SymbolExtractor.h:134); 0xFFFFFFFF = disabled
    uint32_t float_abs;     // lookup context of the flip weight |avg_r - avg_l|, see DESIGN.md
    uint32_t reset;         // 1 = parameters changed: cached window results are stale, recompute the whole backlog
};

// Per-stream symbol-extractor ring state (device resident, updated by the scan kernel only).  Positions are
// monotonic 32-bit counters; a sample's slot in the per-stream rings is (position & (ring_cap - 1)).
struct SymState {
    uint32_t base;          // position of backlog sample 0
    uint32_t held;          // backlog length (SymbolExtractor::samples_.size())
    uint32_t cached;        // window sums W(p) are final for every position p < cached (flags for base + R <= p < cached)
    uint32_t run_pos;       // the current run's sequential sample sum is carried across calls: it covers [base, run_pos)
    float    run_sum;       //   ... = v[base] + v[base+1] + ... + v[run_pos-1], accumulated left to right
    uint32_t _pad[3];
};

// Header of a stream's result slot written by the symbol scan kernel, followed by packed bits.
struct BitsHeader {
    uint32_t nbits;         // symbols produced this call
    uint32_t held_after;    // backlog kept by the symbol extractor after this call
    uint32_t nflips;        // flip points found this call
    uint32_t overflow;      // bit 0: more bits than the result slot can hold (never with the derived capacities; the call fails with HD_ERR_CAPACITY);
                            // bit 1: the flip list was full -- the search stopped at its last flip and goes on in the next call (not an error)
    uint32_t uncached;      // backlog samples whose windows are not final yet (sizes the next call's window kernel)
    // Stream tail only (tail_body.h): a checksum of the call's discriminator output, so that a test can compare every call of a free-running
    // batch with a CPU reference without asking for the samples (which would flush the pipeline and change the launch shape):
    // demod_ck[0] = sum bits(d[i]), demod_ck[1] = sum (i + 1) * bits(d[i]) (mod 2^32) over the demod_n = fir_m samples of the call.
    uint32_t demod_ck[2];
    uint32_t demod_n;       // 0xFFFFFFFF when the kernel that wrote the slot does not compute it
    // The call's tag (engine: call number + 1), stored LAST -- behind a wait for every other store of the wave that wrote the slot (header, bits, and in the
    // stream tail the spectrum statistics).  Result slots live in mapped host memory and the engine's completion events carry no system-scope fence (they
    // cost 3 % of a step): the host takes a slot as delivered when it reads the tag it expects, not because an event said so (collect(), engine.cpp).
    uint32_t seq;
    uint32_t _pad[3];
};
static_assert(sizeof(BitsHeader) == 48, "result slot header");

struct SpectrumStatsDev {   // must match hd::SpectrumStats (host/afc_tracker.hpp)
    int32_t valid;
    int32_t peak1, peak2;
    float power1, power2;
    uint32_t seq;           // the call's tag, stored LAST by the wave that wrote the statistics (as BitsHeader::seq: the host takes them as delivered when it reads the tag it expects)
    double mean, sigma;
};

}  // namespace hd
