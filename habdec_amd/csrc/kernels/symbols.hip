// Symbol extractor on the GPU: demodulated samples -> bits, per stream.
//
// Reference behaviour (code/Decoder/SymbolExtractor.h:108-255): demodulated samples are appended to a per-stream
// backlog (dropped wholesale if more than 30000 are already held); once at least 3 symbols are held, bit edges
// ("flip points") are searched left to right: slide i until the mean of the R samples left of i and the mean of
// the R samples right of i differ in sign, keep sliding until they agree again, and take the position in that
// span where |mean_r - mean_l| is largest (first maximum); continue R samples after it.  The search stops at
// size - samples_per_bit.  Each run between flips yields round(len/spb) copies of (mean > 0); consumed samples
// are erased.
//
// The reference re-evaluates O(backlog * R) window sums on every call.  Here:
//   * the backlog lives in a per-stream power-of-two RING (the discriminator kernel appends straight into it), so
//     "erase" is just advancing the base position -- no copies;
//   * the two window means of a candidate position are pure functions of the samples around it, so they are
//     computed ONCE, when the position's right window is complete (k_sym_avg: one lane per new position, each lane
//     adding its R samples left-to-right from LDS exactly like std::accumulate, so the floats are bit-identical),
//     and cached as one "signs differ" bit (ballot -> 64-bit mask words) plus one weight per position;
//   * the inherently sequential edge search runs on those mask words with one wave per stream (k_sym_scan: 4096
//     positions per step through ballots and find-first-set); per-run sums are accumulated in element order, four runs
//     to a wave (16-lane groups, each adding its own run from its own LDS strip through broadcast reads); lane 0 packs bits.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdlib>

#include "arith.h"
#include "launch.h"
#include "sym_common.h"

namespace hd {
namespace HD_ARITH_NS {

#ifdef HD_STAMP   // diagnostic build only: s_memtime at the phase boundaries of k_symbols, per stream
__device__ unsigned long long g_sym_stamps[8192 * 8];
#define STAMP(i) do { if (threadIdx.x == 0 && s < 8192) g_sym_stamps[s * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" void HD_DBG_NAME(hd_debug_sym_stamps)(unsigned long long* host, size_t n) { (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_sym_stamps), n * 8); }
#else
#define STAMP(i) do { } while (0)
#endif
#ifdef HD_STAMP_SWEEP   // (with HD_STAMP) where a sweep of phase A spends its cycles: staging / window sums / flags, summed over the call's sweeps into stamps 6 and 7
#define SWEEP_DECL unsigned long long sw_t = __builtin_amdgcn_s_memtime(), sw_acc[3] = {0, 0, 0}
#define SWEEP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); sw_acc[i] += t_ - sw_t; sw_t = t_; } while (0)
#else
#define SWEEP_DECL do { } while (0)
#define SWEEP(i) do { } while (0)
#endif
constexpr int kSymLanes = 256;
// the call's tag goes into the result slot LAST, behind a wait for the header and bit stores of the same lane (BitsHeader::seq, dev_types.h)
#define HD_SLOT_DONE() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __hip_atomic_store(&hdr->seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } while (0)
// the call's discriminator checksum (BitsHeader::demod_ck): k_fir_demod's tiles left it in ck_acc; it goes into the slot and the accumulators are cleared for the next call
#define HD_SLOT_CK() do { if (ck_acc) { hdr->demod_ck[0] = ck_acc[2 * (size_t)s]; hdr->demod_ck[1] = ck_acc[2 * (size_t)s + 1]; hdr->demod_n = m; ck_acc[2 * (size_t)s] = 0u; ck_acc[2 * (size_t)s + 1] = 0u; } \
                          else hdr->demod_n = 0xFFFFFFFFu; } while (0)
constexpr uint32_t kRunStrip = 512;                       // samples per run-sum step


__global__ __launch_bounds__(kSymLanes) void k_symbols(const float* __restrict__ tail, uint32_t ring_cap, SymState* __restrict__ sym,
                                                        unsigned long long* __restrict__ flipmask, float* __restrict__ weight,
                                                        const SymbolParams* __restrict__ sp, const StreamCall* __restrict__ call,
                                                        uint32_t* __restrict__ slots, uint32_t slot_words,
                                                        uint32_t* __restrict__ flips_dbg, uint32_t flips_cap, uint32_t fl_cap, const uint32_t seq,
                                                        uint32_t* __restrict__ ck_acc /* [S][2] from k_fir_demod, or null */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // carve: [lmask: ring_cap/64 u64][flips, runinfo: fl_cap u32 each][win: span + R + 16 floats][wl: span + R floats];
    // the run-sum strips (4 waves x 4 groups x kRunStrip / 2 floats) reuse win/wl, which are dead by then
    unsigned long long* lmask = reinterpret_cast<unsigned long long*>(smem);
    __shared__ unsigned long long words[kAvgSpan / 64 + 1];
    __shared__ uint32_t s_nfl, s_overflow, s_frontier;
    __shared__ float s_carry;
    __shared__ uint32_t s_flagged;
    // Workgroups are dealt round-robin to the 8 XCDs.  The kernel ends with its slowest stream, and slow streams (off-tune,
    // noisy: ten flips per call instead of one) tend to come with a period in the stream index (every 8th receiver of a
    // bank, ...), which would pile them onto one XCD's CUs; give XCD j the contiguous block j of the streams instead.
    const uint32_t nS = gridDim.x;
    const uint32_t s = (nS & 7u) ? blockIdx.x : (blockIdx.x & 7u) * (nS >> 3) + (blockIdx.x >> 3);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t* slot = slots + (size_t)s * slot_words;
    BitsHeader* hdr = reinterpret_cast<BitsHeader*>(slot);
    uint32_t* outw = slot + sizeof(BitsHeader) / 4;
    const uint32_t cap_bits = (slot_words - sizeof(BitsHeader) / 4) * 32;
    const uint32_t m = call[s].fir_m;
    const SymState old = sym[s];
    const SymbolParams q = sp[s];                           // (requested with the two above: one round trip, not two)
    if (!m) {                                               // symbol stage not reached this call
        if (tid == 0) { hdr->nbits = 0; hdr->held_after = old.held; hdr->nflips = 0; hdr->overflow = 0; hdr->uncached = old.base + old.held - old.cached; HD_SLOT_CK(); HD_SLOT_DONE(); }
        return;
    }
    SymState st = state_after_push(old, q, m);
    const uint32_t h = st.held;
    // Too little backlog for the extractor to run (SymbolExtractor.h:134): the reference returns; here the window sums of the
    // samples that arrived are still computed and cached (they are pure functions of the samples), so that the call that does
    // run finds one sweep of work instead of two -- busy streams alternate between the two kinds of call.
    const bool search = !(h < q.min_held || h < q.spb);
    if (!search && (q.min_held == 0xFFFFFFFFu || h < q.R)) {
        if (tid == 0) { sym[s] = st; hdr->nbits = 0; hdr->held_after = h; hdr->nflips = 0; hdr->overflow = 0; hdr->uncached = st.base + h - st.cached; HD_SLOT_CK(); HD_SLOT_DONE(); }
        return;
    }
    const uint32_t R = q.R, rmask = ring_cap - 1;
    uint32_t* flips = reinterpret_cast<uint32_t*>(lmask + ring_cap / 64);
    uint32_t* runinfo = flips + fl_cap;                      // (count << 1) | bit
    float* win = reinterpret_cast<float*>(runinfo + fl_cap);
    float* strips = win;                                    // phase C only; (kAvgSpan + R) * 2 floats >= 8 * kRunStrip (four waves x four groups x half a strip)
    const float* v = tail + (size_t)s * ring_cap;
    unsigned long long* gmask = flipmask + (size_t)s * (ring_cap / 64);
    float* gw = weight + (size_t)s * ring_cap;
    const uint32_t end = st.base + h;
    const uint32_t pend = end - R + 1;                      // first position whose right window is still incomplete
    const uint32_t limit = search ? h - q.spb : 0u;         // backlog indices searched: [R, limit)

    if (tid == 0) s_flagged = 0u;
    STAMP(0);
    // ---- A: window sums and flags for the new positions [cached, pend), kAvgSpan positions per sweep (one sweep per call
    // in steady state: a call appends m <= kAvgSpan samples).  Everything the first sweep and the search read from global
    // memory -- the samples, the R cached sums in front of them, the LDS image of the mask words the search can touch, the
    // partially filled mask word `cached` falls into -- is requested before the first value is used: one round trip.
    // W(p) = v[p] + ... + v[p+R-1] summed left to right is BOTH the reference's right window of p and its left window
    // of p+R (same elements, same order, same rounding), so one sum per position is computed and cached (ring `wsum`);
    // flag(p) = sgn(W(p-R)/R) != sgn(W(p)/R).
    float* wl = win + ((kAvgSpan + R + 16 + 3) & ~3u);      // W of [c0 - R, c0 + span)
    const uint32_t wn = kAvgSpan + R + kAvgPos;
    constexpr int WB = kAvgPos == 12 ? 14 : kAvgPos == 8 ? 10 : 6, LB = 2, MB = 2;                   // loads per lane issued back to back (covers R <= 504, 32768 searchable positions)
    unsigned long long first_word = 0ull;                   // mask word holding position c0, as earlier calls left it
    {
        const uint32_t c0 = st.cached;
        const uint32_t wr0 = (st.base + R) & ~63u;                     // ring position of the first searchable word
        const uint32_t nw = search ? ((st.base + limit) - wr0 + 63u) >> 6 : 0u;      // modular difference: safe across the 2^32 wrap
        const bool sweep = (int32_t)(pend - c0) > 0;
        float tw[WB], tl[LB];
        unsigned long long tm[MB];
#pragma unroll
        for (int u = 0; u < WB; ++u) { const uint32_t k = tid + u * kSymLanes; tw[u] = (sweep && k < wn) ? v[(c0 + k) & rmask] : 0.0f; }
#pragma unroll
        for (int u = 0; u < LB; ++u) { const uint32_t k = tid + u * kSymLanes; tl[u] = (sweep && k < R) ? gw[(c0 - R + k) & rmask] : 0.0f; }
#pragma unroll
        for (int u = 0; u < MB; ++u) { const uint32_t i = tid + u * kSymLanes; tm[u] = i < nw ? gmask[((wr0 + 64u * i) & rmask) >> 6] : 0ull; }
        if (sweep) first_word = gmask[(c0 & rmask) >> 6];
#pragma unroll
        for (int u = 0; u < WB; ++u) { const uint32_t k = tid + u * kSymLanes; if (sweep && k < wn) win[k] = tw[u]; }
#pragma unroll
        for (int u = 0; u < LB; ++u) { const uint32_t k = tid + u * kSymLanes; if (sweep && k < R) wl[k] = tl[u]; }
#pragma unroll
        for (int u = 0; u < MB; ++u) { const uint32_t i = tid + u * kSymLanes; if (i < nw) lmask[((wr0 + 64u * i) & rmask) >> 6] = tm[u]; }
        // larger R / backlog than the batches cover: plain loops
        if (sweep) {
            for (uint32_t k = tid + WB * kSymLanes; k < wn; k += kSymLanes) win[k] = v[(c0 + k) & rmask];
            for (uint32_t k = tid + LB * kSymLanes; k < R; k += kSymLanes) wl[k] = gw[(c0 - R + k) & rmask];
        }
        for (uint32_t i = tid + MB * kSymLanes; i < nw; i += kSymLanes) { const uint32_t wi = ((wr0 + 64u * i) & rmask) >> 6; lmask[wi] = gmask[wi]; }
    }
    STAMP(1);
    bool staged = true;
    uint32_t sweeps = 0, wl_c0 = st.cached;                 // wl holds W of [wl_c0 - R, wl_c0 + kAvgSpan) after the last sweep
    SWEEP_DECL;
    float nxw[WB];                                          // the NEXT sweep's samples, requested before this sweep's window sums: their round trip runs under the adds
    for (uint32_t c0 = st.cached; (int32_t)(pend - c0) > 0; c0 += kAvgSpan) {
        SWEEP(2);
        const uint32_t wb0 = c0 & ~63u, wsh = c0 & 63u;    // sweeps start wherever the previous call stopped: words are shared
        if (!staged) {
            // the R sums in front of this sweep are the previous sweep's last R: they are in LDS (wl[span .. span + R)), as is the mask word it ended in
            float tl[LB];
#pragma unroll
            for (int u = 0; u < LB; ++u) { const uint32_t k = tid + u * kSymLanes; tl[u] = k < R ? wl[kAvgSpan + k] : 0.0f; }
            __syncthreads();                                // previous sweep's LDS consumers are done
#pragma unroll
            for (int u = 0; u < WB; ++u) { const uint32_t k = tid + u * kSymLanes; if (k < wn) win[k] = nxw[u]; }
            for (uint32_t k = tid + WB * kSymLanes; k < wn; k += kSymLanes) win[k] = v[(c0 + k) & rmask];
#pragma unroll
            for (int u = 0; u < LB; ++u) { const uint32_t k = tid + u * kSymLanes; if (k < R) wl[k] = tl[u]; }
            for (uint32_t k = tid + LB * kSymLanes; k < R; k += kSymLanes) wl[k] = gw[(c0 - R + k) & rmask];   // (windows longer than the batches cover: from the ring, as written above)
            first_word = lmask[(c0 & rmask) >> 6];          // as the previous sweep left it
        }
        staged = false;
        ++sweeps; wl_c0 = c0;
        if (tid < kAvgSpan / 64 + 1) words[tid] = 0ull;
        __syncthreads();
        if ((int32_t)(pend - (c0 + kAvgSpan)) > 0) {        // another sweep follows: its samples
#pragma unroll
            for (int u = 0; u < WB; ++u) { const uint32_t k = tid + u * kSymLanes; nxw[u] = k < wn ? v[(c0 + kAvgSpan + k) & rmask] : 0.0f; }
        }
        SWEEP(0);
        const uint32_t p0 = c0 + tid * kAvgPos;
        const bool any = (int32_t)(pend - p0) > 0;
        float wp[kAvgPos];
        if (any) {
            // (fast mode: an exactly summed anchor per lane and its neighbours by sliding -- a seventh of the adds at R = 427; sym_common.h)
            if constexpr (kFastArith && kFastWindows) window_sums_slide<kAvgPos>(win + tid * kAvgPos, R, wp);
            else if constexpr (kAvgPos >= 8) window_sums_wide<kAvgPos>(win + tid * kAvgPos, R, wp); else window_sums(win + tid * kAvgPos, R, wp);
#pragma unroll
            for (int j = 0; j < kAvgPos; ++j) {
                wl[R + tid * kAvgPos + j] = wp[j];
                if ((int32_t)(pend - (p0 + j)) > 0) gw[(p0 + j) & rmask] = wp[j];
            }
        }
        __syncthreads();
        SWEEP(1);
        if (any) {
            // no clamping: p >= base + R is all the search reads, p + R <= end
            float wlv[kAvgPos];
            const float4* wl4 = reinterpret_cast<const float4*>(wl + tid * kAvgPos);
#pragma unroll
            for (int c4 = 0; c4 < kAvgPos / 4; ++c4) { const float4 x = wl4[c4]; wlv[4 * c4] = x.x; wlv[4 * c4 + 1] = x.y; wlv[4 * c4 + 2] = x.z; wlv[4 * c4 + 3] = x.w; }
            const unsigned int bits = sign_flags<kAvgPos>(wlv, wp, min((uint32_t)kAvgPos, pend - p0), R);
            if (bits) {
                const uint32_t bp = wsh + tid * kAvgPos;    // bit position of p0 relative to the sweep's first word
                const uint32_t sh = bp & 63u;
                atomicOr(&words[bp >> 6], (unsigned long long)bits << sh);
                if (sh > 64u - kAvgPos) atomicOr(&words[(bp >> 6) + 1], (unsigned long long)bits >> (64u - sh));
            }
        }
        __syncthreads();
        if (tid < kAvgSpan / 64 + 1 && (int32_t)(pend - (wb0 + tid * 64)) > 0 && (int32_t)((c0 + kAvgSpan) - (wb0 + tid * 64)) > 0) {
            const uint32_t wi = ((wb0 + tid * 64) & rmask) >> 6;
            unsigned long long w = words[tid];
            if (tid == 0 && wsh) w |= first_word & ((1ull << wsh) - 1ull);     // flags of the positions in front of c0
            gmask[wi] = w;
            lmask[wi] = w;
        }
    }
    // The edge search reads window sums this call has just written.  After a single sweep (the steady state) they are all
    // still in `wl`; only after several sweeps, or when the busy-stream cache below takes over win/wl, must the global copies
    // be visible first (a store round trip).
    bool lds_w = sweeps == 1;
    if (sweeps > 1) __threadfence_block();
    __syncthreads();
    if (!search) {                                          // nothing to extract yet: only the cache moved on
        if (tid == 0) {
            if ((int32_t)(pend - st.cached) > 0) st.cached = pend;
            sym[s] = st;
            hdr->nbits = 0; hdr->held_after = h; hdr->nflips = 0; hdr->overflow = 0; hdr->uncached = end - st.cached; HD_SLOT_CK(); HD_SLOT_DONE();
        }
        STAMP(2); STAMP(3); STAMP(4); STAMP(5);
        return;
    }
    // ---- A2: busy streams (off-tune or noisy: ten flips per call instead of one) would pay one global round trip per flip
    // for the zone weights.  When the mask image shows more flagged positions than a couple of clean edges produce, the
    // window sums of the searchable backlog (as much as fits) are pulled into LDS first, over win/wl, which are idle here.
    float* wc = win;
    uint32_t wc_n = 0;
    {
        const uint32_t wr0 = (st.base + R) & ~63u;
        const uint32_t nw = ((st.base + limit) - wr0 + 63u) >> 6;
        uint32_t cnt = 0;
        for (uint32_t i = tid; i < nw; i += kSymLanes) cnt += (uint32_t)__popcll(lmask[((wr0 + 64u * i) & rmask) >> 6]);
        for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
        if (lane == 0 && cnt) atomicAdd(&s_flagged, cnt);
        __syncthreads();
        if (s_flagged > 4u * R) {
            if (lds_w) { __threadfence_block(); __syncthreads(); lds_w = false; }   // the fill reads this call's sums from global, and overwrites wl
            const uint32_t cap = ((kAvgSpan + R + 16 + 3) & ~3u) + ((kAvgSpan + R + 3) & ~3u);
            wc_n = min(limit, cap);
            constexpr int CB = 9;
            for (uint32_t k0 = tid; k0 < wc_n; k0 += CB * kSymLanes) {
                float t[CB];
#pragma unroll
                for (int u = 0; u < CB; ++u) { const uint32_t k = k0 + u * kSymLanes; t[u] = k < wc_n ? gw[(st.base + k) & rmask] : 0.0f; }
#pragma unroll
                for (int u = 0; u < CB; ++u) { const uint32_t k = k0 + u * kSymLanes; if (k < wc_n) wc[k] = t[u]; }
            }
            __syncthreads();
        }
    }
    STAMP(2);

    // window sum of backlog index i: busy-stream cache, this call's sums still in LDS, or the global ring
    auto wsum = [&](uint32_t i) -> float {
        if (i < wc_n) return wc[i];
        const uint32_t pos = st.base + i, rel = pos - (wl_c0 - R);
        if (lds_w && rel < R + kAvgSpan && (int32_t)(pend - pos) > 0) return wl[rel];
        return gw[pos & rmask];
    };
    // ---- B: edge search (wave 0)
    if (wave == 0) {
        uint32_t pos = R, nfl = 0, overflow = 0;
        uint32_t frontier = 0xFFFFFFFFu;                    // backlog index no future flip point can precede
        while (pos < limit) {
            const uint32_t lo = find_flag_lds(lmask, st.base, rmask, pos, limit, true);
            if (lo == 0xFFFFFFFFu) break;
            const uint32_t hi = find_flag_lds(lmask, st.base, rmask, lo + 1, limit, false);
            if (hi == 0xFFFFFFFFu) { frontier = lo; break; }      // an edge zone is open: the next flip lies at or after lo
            // first maximum of the weight over [lo, hi): key = (weight bits, ~index) -- weights are >= 0, so their bit patterns
            // order like the values, and among equal weights the smaller index has the larger key
            unsigned long long key = 0ull;
            constexpr int ZB = 4;                             // 256 zone positions per round trip (a zone is about R long)
            for (uint32_t i0 = lo + lane; i0 < hi; i0 += 64 * ZB) {
                float wr_[ZB], wl_[ZB];
#pragma unroll
                for (int u = 0; u < ZB; ++u) {
                    const uint32_t i = i0 + 64 * u;
                    wr_[u] = i < hi ? wsum(i) : 0.0f;
                    wl_[u] = i < hi ? wsum(i - R) : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < ZB; ++u) {
                    const uint32_t i = i0 + 64 * u;
                    if (i < hi) {
                        const float d = wr_[u] / (float)R - wl_[u] / (float)R;   // avg_r - avg_l
                        const float w = q.float_abs ? __builtin_fabsf(d) : (float)abs((int)d);
                        const unsigned long long k = ((unsigned long long)__builtin_bit_cast(uint32_t, w) << 32) | (uint32_t)~i;
                        key = k > key ? k : key;
                    }
                }
            }
            const uint32_t f = ~(uint32_t)wave_max_u64(key);
            if (lane == 0) flips[nfl] = f;
            ++nfl;
            pos = f + R;
            // The flip list is full (kMaxFlipsPerCall; the reference has no such bound): stop HERE and leave the rest of the backlog to the next call.
            // The search is a pure function of the samples around each flip, and a search that resumes at the last flip's position + R is exactly
            // what the reference does next -- so the stream of bits is the one the reference produces, only delivered a call later (BitsHeader::
            // overflow bit 1, not an error; the one visible difference: a backlog that is consumed later reaches the 30 000-sample vent earlier).
            if (nfl == fl_cap && pos < limit) { frontier = f; overflow = 2; break; }
        }
        if (frontier == 0xFFFFFFFFu) frontier = max(pos, limit);   // nothing flagged in [pos, limit)
        if (lane == 0) { s_nfl = nfl; s_overflow = overflow; s_frontier = frontier; }
    }
    __syncthreads();
    const uint32_t nfl = s_nfl;
    STAMP(3);

    // ---- C: per-run sums in element order (std::accumulate), runs dealt round-robin to the waves.  The sum of the
    // run in progress is carried across calls (SymState::run_sum covers [base, run_pos)): every call extends it up to
    // the search frontier -- samples that can no longer become a flip point -- so when the flip finally shows up only
    // the few samples between the frontier and the flip remain.  The chain is the same left-to-right sequence of adds.
    // FOUR runs per wave at a time (round 5): lanes 16 g .. 16 g + 15 work on run 4 q + g -- their own strip of the wave's LDS, their own accumulator -- in ONE
    // instruction stream, so a call's runs cost a quarter of the vector instructions (each add used to be a wave instruction with one useful lane) and
    // sixteen runs are in flight per workgroup instead of four.  The runs of a batch differ in length: a strip is zero-filled behind its run's end and every
    // group adds the batch's longest length -- x + (+0.0f) is x for every x, and a sum that starts at +0 is never -0, so the extra adds change nothing.
    constexpr int G = 4, GL = 64 / G, SL = (int)kRunStrip / 2;           // groups per wave, lanes per group, samples per strip and group
    constexpr int NX = SL / GL;                                           // loads in flight per lane: one whole strip of its group
    static_assert(G * SL * (kSymLanes / 64) <= 2 * (kAvgSpan + 4), "the strips overlay win / wl");
    const uint32_t g = lane / GL, gl = lane % GL;
    float* sbg = strips + wave * (G * SL) + g * SL;                       // my group's strip (wave-private memory)
    const uint32_t carried_to = st.run_pos - st.base;       // run 0's sum is already known up to here
    // Run nfl is the carry for the next call: the run now in progress starts at the last flip (or continues) and is summed up to the frontier.
    const uint32_t frontier = s_frontier;
    for (uint32_t rb = (uint32_t)G * wave; rb <= nfl; rb += (uint32_t)G * (kSymLanes / 64)) {
        const uint32_t r = rb + g;
        const bool live = r <= nfl;
        uint32_t a = 0, b = 0;                                            // the samples run r still has to add: backlog indices [a, b)
        float acc = 0.0f;
        if (live) {
            if (r < nfl) { b = flips[r]; a = r ? flips[r - 1] : min(carried_to, b); }
            else { a = nfl ? flips[nfl - 1] : carried_to; b = max(a, frontier); }
            if (r == 0u || (r == nfl && !nfl)) acc = st.run_sum;
        }
        const uint32_t len = b > a ? b - a : 0u;
        const uint32_t maxlen = max(max((uint32_t)__builtin_amdgcn_readlane((int)len, 0), (uint32_t)__builtin_amdgcn_readlane((int)len, GL)),
                                    max((uint32_t)__builtin_amdgcn_readlane((int)len, 2 * GL), (uint32_t)__builtin_amdgcn_readlane((int)len, 3 * GL)));
        float nx[NX];
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const uint32_t k = a + GL * j + gl;
            nx[j] = k < b ? v[(st.base + k) & rmask] : 0.0f;
        }
        for (uint32_t k0 = 0; k0 < maxlen; k0 += SL) {
            // the previous strip's LDS reads have all been consumed by its adds (data dependence), so the strip can be
            // rewritten from the prefetched registers; the next strip's loads then fly under this strip's adds
#pragma unroll
            for (int j = 0; j < NX; ++j) sbg[GL * j + gl] = nx[j];
            if (k0 + SL < maxlen) {
#pragma unroll
                for (int j = 0; j < NX; ++j) {
                    const uint32_t k = a + k0 + SL + GL * j + gl;
                    nx[j] = k < b ? v[(st.base + k) & rmask] : 0.0f;
                }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);             // lgkmcnt(0): this wave's strip writes have landed (wave-private strips)
            __builtin_amdgcn_wave_barrier();
            const uint32_t cnt = (min((uint32_t)SL, maxlen - k0) + 31u) & ~31u;   // whole blocks of 32: the strip is zero-filled behind every run's end
            const float4* s4 = reinterpret_cast<const float4*>(sbg);
            // 32 samples per block through group-uniform 16-byte reads (broadcast within a group), the next block in flight under this block's 32 adds; two
            // register images take turns, so nothing is copied between blocks
            float4 c[8], n[8];
            auto ld = [&](float4 (&x)[8], const uint32_t at) {
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = s4[(at >> 2) + u];
            };
            auto adds = [&](const float4 (&x)[8]) {
#pragma unroll
                for (int u = 0; u < 8; ++u) { acc = acc + x[u].x; acc = acc + x[u].y; acc = acc + x[u].z; acc = acc + x[u].w; }
            };
            uint32_t i = 0;
            ld(c, 0u);                                                      // invariant: c holds samples [i, i + 32), i + 32 <= cnt
            for (; i + 96 <= cnt; i += 64) { ld(n, i + 32); adds(c); ld(c, i + 64); adds(n); }
            if (i + 64 <= cnt) { ld(n, i + 32); adds(c); adds(n); }
            else adds(c);
            __builtin_amdgcn_wave_barrier();
        }
        if (live && gl == 0) {
            if (r < nfl) {
                const uint32_t a0 = r ? flips[r - 1] : 0u;
                const float mean = acc / (float)(b - a0);
                const uint32_t cnt = (uint32_t)roundf((float)(b - a0) / (float)q.spb);
                runinfo[r] = (cnt << 1) | (mean > 0.0f ? 1u : 0u);
            } else s_carry = acc;
        }
    }
    __syncthreads();

    STAMP(4);
    // ---- D: bits, ring advance, result slot
    if (tid == 0) {
        uint32_t nbits = 0, cur = 0, overflow = s_overflow;
        for (uint32_t r = 0; r < nfl; ++r) {
            const uint32_t bit = runinfo[r] & 1u;
            for (uint32_t k = runinfo[r] >> 1; k; --k) {
                if (nbits >= cap_bits) { overflow |= 1u; break; }
                cur |= bit << (nbits & 31);
                if ((nbits & 31) == 31) { outw[nbits >> 5] = cur; cur = 0; }
                ++nbits;
            }
        }
        if (nbits & 31) outw[nbits >> 5] = cur;
        if (flips_dbg)
            for (uint32_t r = 0; r < nfl && r < flips_cap; ++r) flips_dbg[(size_t)s * flips_cap + r] = flips[r];
        const uint32_t last = nfl ? flips[nfl - 1] : 0u;   // erase the consumed prefix (SymbolExtractor.h:156-157) = advance the base
        st.run_pos = st.base + max(nfl ? last : carried_to, frontier);
        st.run_sum = s_carry;
        st.cached = pend;
        st.base += last;
        st.held = h - last;
        sym[s] = st;
        hdr->nbits = nbits; hdr->held_after = st.held; hdr->nflips = nfl; hdr->overflow = overflow; hdr->uncached = end - st.cached; HD_SLOT_CK(); HD_SLOT_DONE();
#if defined(HD_STAMP) && defined(HD_STAMP_SWEEP)
        g_sym_stamps[s * 8 + 6] = sw_acc[0]; g_sym_stamps[s * 8 + 7] = sw_acc[1];
#elif defined(HD_STAMP)
        g_sym_stamps[s * 8 + 6] = nfl; g_sym_stamps[s * 8 + 7] = nfl ? flips[nfl - 1] : 0;
#endif
    }
    STAMP(5);
}

void launch_symbols(hipStream_t st, uint32_t n_streams, uint32_t max_m, uint32_t max_new, uint32_t max_R, const float* tail,
                    uint32_t ring_cap, SymState* sym, unsigned long long* flipmask, float* weight, const SymbolParams* sp,
                    const StreamCall* call, uint32_t* slots, uint32_t slot_words, uint32_t* flips_dbg, uint32_t flips_cap, uint32_t min_R, uint32_t seq, hipEvent_t ev_stop, uint32_t* ck_acc)
{
    (void)max_m; (void)max_new;
    // a flip point moves the search on by R, so a call finds at most backlog / R + 1 of them (the backlog never exceeds the ring)
    uint32_t fl_cap = ring_cap / (min_R ? min_R : 4u) + 2u;
    fl_cap = (fl_cap + 63u) & ~63u;
    if (fl_cap > kMaxFlipsPerCall) fl_cap = kMaxFlipsPerCall;
    const size_t lds = (size_t)(ring_cap / 64) * 8 + (size_t)fl_cap * 8 +
                 (size_t)(((kAvgSpan + max_R + 16 + 3) & ~3u) + ((kAvgSpan + max_R + 3) & ~3u)) * 4;
    if (ev_stop) hipExtLaunchKernelGGL(k_symbols, dim3(n_streams), dim3(kSymLanes), (uint32_t)lds, st, nullptr, ev_stop, 0u, tail, ring_cap, sym, flipmask, weight, sp, call, slots,
                                       slot_words, flips_dbg, flips_cap, fl_cap, seq, ck_acc);
    else hipLaunchKernelGGL(k_symbols, dim3(n_streams), dim3(kSymLanes), lds, st, tail, ring_cap, sym, flipmask, weight, sp, call, slots,
                            slot_words, flips_dbg, flips_cap, fl_cap, seq, ck_acc);
}

}  // namespace HD_ARITH_NS
}  // namespace hd
