// Pieces of the symbol extractor shared by k_symbols (symbols.hip) and the fused stream tail (tail_body.h):
// exact-order window sums, sign test, wave-wide arg-max, search over the flag mask.  See symbols.hip for the
// algorithm and its mapping to the reference (code/Decoder/SymbolExtractor.h:108-255).
#pragma once
#include <hip/hip_runtime.h>

#include "launch.h"

namespace hd {

constexpr int kAvgLanes = 256;
constexpr uint32_t kMaxFlipsPerCall = 512;    // upper bound of the LDS flip list ON EVERY LAUNCH PATH (the launcher sizes it from backlog / R): 512 is what fits beside
                                              // a stage-1 slot inside a step launch, and a stream must get the same answer whichever path serves it.  Three symbols' worth
                                              // of backlog holds a dozen flip points; a call that finds more than the bound (noise with four-sample windows and thousands of
                                              // decimated samples per call) stops at the bound and goes on in the next call -- same bits, a call later (symbols.hip, tail_body.h)

__device__ __forceinline__ int sgnf(float v) { return (0.0f < v) - (v < 0.0f); }

// State after this call's samples were appended (SymbolExtractor.h:116-124: the vent happens before the append).
__device__ __forceinline__ SymState state_after_push(SymState st, const SymbolParams& q, uint32_t m)
{
    if (st.held > kVentLimit) { st.base += st.held; st.held = 0; st.cached = st.base; st.run_pos = st.base; st.run_sum = 0.0f; }
    if (q.reset) st.cached = st.base;
    st.held += m;
    return st;
}

// Window sums for the NEW candidate positions.  One lane owns 4 consecutive positions: it walks its R+3 left-window
// samples (then the R+3 right-window samples) once with 16-byte LDS reads and adds each sample to every one of its
// four accumulators whose window contains it.  Each accumulator still receives exactly its own R samples in index
// order, so the sums are bit-identical to std::accumulate, with 1/16 of the LDS instructions of the scalar form.
#ifndef HD_SYM_POS
#define HD_SYM_POS 8     // (8: half the LDS reads per position of 4 and four add chains per lane; /4 at 512 kHz: 0.868 against 0.898 ms per step.  12 -- lanes 48
#endif                   // bytes apart, 16-byte LDS reads without bank conflicts, see window_sums_wide -- measured SLOWER: /4 0.860 against 0.853 ms, /16 0.360 against
                         // 0.353: the workgroup's LDS grows by 8 KB, one fewer fits a CU, and the conflicts were latency the other waves covered)
constexpr int kAvgPos = HD_SYM_POS;                         // positions per lane of k_symbols (4, 8 or 12)
constexpr int kAvgSpan = kAvgLanes * kAvgPos;               // positions per workgroup and sweep

typedef float f32x2 __attribute__((ext_vector_type(2)));

// acc[j] = w[j] + w[j+1] + ... + w[j+R-1] for j = 0..3, each in index order.  Interior samples feed all four sums:
// two v_pk_add_f32 per sample.
__device__ __forceinline__ void window_sums(const float* __restrict__ w, uint32_t R, float acc[4])
{
    constexpr int kAvgPos = 4;
    f32x2 a01 = {0.0f, 0.0f}, a23 = {0.0f, 0.0f};
    const uint32_t total = R + kAvgPos - 1;                 // samples touched: w[0 .. R+3)
    {   // head chunk: element u feeds accumulators j <= u (R >= 4 always)
        const float4 x = *reinterpret_cast<const float4*>(w);
        a01.x = a01.x + x.x;
        a01 = a01 + (f32x2){x.y, x.y};
        a01 = a01 + (f32x2){x.z, x.z}; a23.x = a23.x + x.z;
        a01 = a01 + (f32x2){x.w, x.w}; a23 = a23 + (f32x2){x.w, x.w};
    }
    uint32_t e = 4;
    // one chunk ahead: the next 16-byte read is in flight during the adds; two register images take turns (two chunks per trip), so no
    // chunk is copied from "next" to "current" between trips
    auto feed4 = [&](const float4 x) {                      // interior: every element feeds all four
        a01 = a01 + (f32x2){x.x, x.x}; a23 = a23 + (f32x2){x.x, x.x};
        a01 = a01 + (f32x2){x.y, x.y}; a23 = a23 + (f32x2){x.y, x.y};
        a01 = a01 + (f32x2){x.z, x.z}; a23 = a23 + (f32x2){x.z, x.z};
        a01 = a01 + (f32x2){x.w, x.w}; a23 = a23 + (f32x2){x.w, x.w};
    };
    float4 xa = *reinterpret_cast<const float4*>(w + 4), xb;
    for (; e + 8 <= R; e += 8) {                            // xa holds elements e .. e+3
        xb = *reinterpret_cast<const float4*>(w + e + 4);
        __builtin_amdgcn_sched_barrier(0);
        feed4(xa);
        xa = *reinterpret_cast<const float4*>(w + e + 8);  // (callers pad their windows: a chunk past the last one is readable)
        __builtin_amdgcn_sched_barrier(0);
        feed4(xb);
    }
    if (e + 4 <= R) { feed4(xa); e += 4; }
    float acc0 = a01.x, acc1 = a01.y, acc2 = a23.x, acc3 = a23.y;
    for (; e < total; e += 4) {                             // tail chunks: element e+u feeds accumulators with e+u < j + R
        const float4 x = *reinterpret_cast<const float4*>(w + e);
        const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = e + u;
            if (i < R) acc0 = acc0 + xs[u];
            if (i < 1 + R) acc1 = acc1 + xs[u];
            if (i < 2 + R) acc2 = acc2 + xs[u];
            if (i < 3 + R) acc3 = acc3 + xs[u];
        }
    }
    acc[0] = acc0; acc[1] = acc1; acc[2] = acc2; acc[3] = acc3;
}

// The same for a lane that owns KP (8 or 12) consecutive positions: acc[j] = w[j] + ... + w[j+R-1], j = 0..KP-1.  KP/2 packed accumulators are
// KP/2 independent add chains -- what ONE wave needs to issue back to back (two chains wait out the add latency on every element).
// Element e feeds accumulator j iff j <= e <= j + R - 1.  `w` is 16-byte aligned and readable up to w[R + KP + 7].
// Eight positions per lane put consecutive lanes 32 bytes apart: two lanes of every ds_read_b128 lane group then share a bank quad (a 2-way
// conflict on every read); twelve put them 48 bytes apart -- three 16-byte chunks, an odd stride -- and the sixteen lanes of a group cover
// all sixteen quads.  The one-wave stream tail keeps eight (512 new positions per round = 64 lanes x 8: a full sweep); k_symbols was measured with
// twelve and stays at eight too (HD_SYM_POS).
template <int KP>
__device__ __forceinline__ void window_sums_wide(const float* __restrict__ w, uint32_t R, float (&acc)[KP])
{
    static_assert(KP % 4 == 0 && KP >= 8, "whole 16-byte chunks per lane");
    float s[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) s[j] = 0.0f;
    const float4* w4 = reinterpret_cast<const float4*>(w);
    if (R >= (uint32_t)KP) {
        {   // head: element u < KP feeds the accumulators j <= u
            float xs[KP];
#pragma unroll
            for (int c = 0; c < KP / 4; ++c) { const float4 x = w4[c]; xs[4 * c] = x.x; xs[4 * c + 1] = x.y; xs[4 * c + 2] = x.z; xs[4 * c + 3] = x.w; }
#pragma unroll
            for (int u = 0; u < KP; ++u)
#pragma unroll
                for (int j = 0; j <= u; ++j) s[j] = s[j] + xs[u];
        }
        f32x2 a[KP / 2];
#pragma unroll
        for (int p = 0; p < KP / 2; ++p) a[p] = (f32x2){s[2 * p], s[2 * p + 1]};
        uint32_t e = KP;
        auto feed = [&](const float4 x) {                   // interior: every element feeds every accumulator
#pragma unroll
            for (int p = 0; p < KP / 2; ++p) a[p] = a[p] + (f32x2){x.x, x.x};
#pragma unroll
            for (int p = 0; p < KP / 2; ++p) a[p] = a[p] + (f32x2){x.y, x.y};
#pragma unroll
            for (int p = 0; p < KP / 2; ++p) a[p] = a[p] + (f32x2){x.z, x.z};
#pragma unroll
            for (int p = 0; p < KP / 2; ++p) a[p] = a[p] + (f32x2){x.w, x.w};
        };
        float4 xa = w4[KP / 4], xb;                         // two images taking turns, one chunk ahead (no copies between trips)
        for (; e + 8 <= R; e += 8) {                        // xa holds elements e .. e+3
            xb = w4[(e >> 2) + 1];
            __builtin_amdgcn_sched_barrier(0);
            feed(xa);
            xa = w4[(e >> 2) + 2];
            __builtin_amdgcn_sched_barrier(0);
            feed(xb);
        }
        if (e + 4 <= R) { feed(xa); e += 4; }
#pragma unroll
        for (int p = 0; p < KP / 2; ++p) { s[2 * p] = a[p].x; s[2 * p + 1] = a[p].y; }
        for (; e < R; ++e) {                                // R % 4 elements that still feed every accumulator
            const float xe = w[e];
#pragma unroll
            for (int j = 0; j < KP; ++j) s[j] = s[j] + xe;
        }
        // tail: element R + d (d = 0 .. KP-2) feeds the accumulators j > d
        float xt[KP - 1];
#pragma unroll
        for (int d = 0; d < KP - 1; ++d) xt[d] = w[R + d];
#pragma unroll
        for (int d = 0; d < KP - 1; ++d)
#pragma unroll
            for (int j = d + 1; j < KP; ++j) s[j] = s[j] + xt[d];
    } else {                                                // tiny windows (fewer than 32 samples per bit): plain loops
        for (uint32_t e = 0; e < R + KP - 1; ++e) {
            const float xe = w[e];
#pragma unroll
            for (int j = 0; j < KP; ++j)
                if (e < (uint32_t)j + R && (uint32_t)j <= e) s[j] = s[j] + xe;
        }
    }
#pragma unroll
    for (int j = 0; j < KP; ++j) acc[j] = s[j];
}
// flag(p) = sgn(W(p - R) / R) != sgn(W(p) / R) for a lane's KP consecutive positions, of which the first nvalid exist: bit j of the result.  The quotient's sign
// is the sum's unless the quotient underflows to zero (|W| / R < 2^-149: only then does the division matter).  When every sum of every active lane of the WAVE
// is beyond that -- always, on real data -- two sums differ in sign iff their sign bits differ: three instructions per position.  (Written per position as
// `|W| > tiny ? sgn(W) : sgn(W / R)` the compiler turns the ?: into a select and evaluates BOTH IEEE divisions for every position: ~50 vector instructions
// each, a sixth of k_symbols' vector instructions at /4 -- round 5.)
template <int KP>
__device__ __forceinline__ unsigned int sign_flags(const float (&wl)[KP], const float (&wr)[KP], const uint32_t nvalid, const uint32_t R)
{
    const float tiny = (float)R * 2.8e-45f;             // > R * 2^-149, far below any non-zero window sum of real data
    bool plain = true;
#pragma unroll
    for (int j = 0; j < KP; ++j)
        if ((uint32_t)j < nvalid) plain = plain && __builtin_fabsf(wl[j]) > tiny && __builtin_fabsf(wr[j]) > tiny;
    unsigned int bits = 0;
    if (__builtin_amdgcn_ballot_w64(!plain) == 0ull) {  // (wave-uniform: every active lane's sums are ordinary -- non-zero, not NaN)
#pragma unroll
        for (int j = 0; j < KP; ++j) bits |= ((__builtin_bit_cast(uint32_t, wl[j]) ^ __builtin_bit_cast(uint32_t, wr[j])) >> 31) << j;
    } else {
        auto avg_sign = [&](float wsum) { return __builtin_fabsf(wsum) > tiny ? sgnf(wsum) : sgnf(wsum / (float)R); };
#pragma unroll
        for (int j = 0; j < KP; ++j) if (avg_sign(wl[j]) != avg_sign(wr[j])) bits |= 1u << j;
    }
    return nvalid >= (uint32_t)KP ? bits : bits & ((1u << nvalid) - 1u);
}

// FAST arithmetic mode (arith.h) only.  The same KP window sums, cheaper: acc[0] = w[0] + ... + w[R-1] as four interleaved partial sums (two packed chains
// over the 16-byte reads, joined at the end), then each neighbour from its predecessor: acc[j] = acc[j-1] + (w[j-1+R] - w[j-1]) -- R / 2 + 2 (KP - 1) adds
// per lane where the exact-order form needs (R + KP - 1) KP / 2 packed ones (R = 427 at /4 and 300 baud: a seventh).  Not std::accumulate's order, so not its
// rounding: a sum of R terms carries R roundings either way, the slide adds two per step over at most KP - 1 steps from an exactly summed anchor, and the
// difference stays at the level of the sums' own rounding noise (<= 1e-6 of the largest window sum; tests/test_gpu_fast.py compares every decision the
// sums feed -- flags, flip points, bits, characters -- with the reference's on every stream).  `w` is 16-byte aligned and readable up to w[R + KP + 7].
template <int KP>
__device__ __forceinline__ void window_sums_slide(const float* __restrict__ w, uint32_t R, float (&acc)[KP])
{
    static_assert(KP % 4 == 0 && KP >= 4, "whole 16-byte chunks per lane");
    const float4* w4 = reinterpret_cast<const float4*>(w);
    float head[KP];                                         // w[0 .. KP): what the slide subtracts
#pragma unroll
    for (int c = 0; c < KP / 4; ++c) { const float4 x = w4[c]; head[4 * c] = x.x; head[4 * c + 1] = x.y; head[4 * c + 2] = x.z; head[4 * c + 3] = x.w; }
    f32x2 a = {0.0f, 0.0f}, b = {0.0f, 0.0f};
    uint32_t e = 0;
    if (R >= 16u) {
        float4 xa = w4[0], xb = w4[1], ya, yb;              // two images of eight elements taking turns, one ahead
        for (; e + 24 <= R; e += 16) {
            ya = w4[(e >> 2) + 2]; yb = w4[(e >> 2) + 3];
            __builtin_amdgcn_sched_barrier(0);
            a = a + (f32x2){xa.x, xa.y}; b = b + (f32x2){xa.z, xa.w}; a = a + (f32x2){xb.x, xb.y}; b = b + (f32x2){xb.z, xb.w};
            xa = w4[(e >> 2) + 4]; xb = w4[(e >> 2) + 5];
            __builtin_amdgcn_sched_barrier(0);
            a = a + (f32x2){ya.x, ya.y}; b = b + (f32x2){ya.z, ya.w}; a = a + (f32x2){yb.x, yb.y}; b = b + (f32x2){yb.z, yb.w};
        }
        // xa, xb hold elements [e, e + 8), all inside the window (e + 8 <= R here: the loop leaves 8 <= R - e < 24)
        a = a + (f32x2){xa.x, xa.y}; b = b + (f32x2){xa.z, xa.w}; a = a + (f32x2){xb.x, xb.y}; b = b + (f32x2){xb.z, xb.w};
        e += 8;
        if (e + 8 <= R) {
            const float4 za = w4[e >> 2], zb = w4[(e >> 2) + 1];
            a = a + (f32x2){za.x, za.y}; b = b + (f32x2){za.z, za.w}; a = a + (f32x2){zb.x, zb.y}; b = b + (f32x2){zb.z, zb.w};
            e += 8;
        }
    }
    float sum = (a.x + a.y) + (b.x + b.y);
    for (; e < R; ++e) sum = sum + w[e];                    // the R % 8 elements left (the whole window when R < 16)
    acc[0] = sum;
    float tail[KP - 1];
#pragma unroll
    for (int d = 0; d < KP - 1; ++d) tail[d] = w[R + d];
#pragma unroll
    for (int j = 1; j < KP; ++j) { sum = sum + (tail[j - 1] - head[j - 1]); acc[j] = sum; }
}

constexpr int kWidePos = 8;
__device__ __forceinline__ void window_sums8(const float* __restrict__ w, uint32_t R, float (&acc)[kWidePos]) { window_sums_wide<kWidePos>(w, R, acc); }

// Wave-wide maximum of an unsigned 64-bit key with DPP moves (a ds_bpermute-based shuffle reduction costs an LDS round trip
// per step, ~1.5k cycles for an arg-max; this is a few dozen).  Result is uniform.
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k)
{
#define HD_DPP_MAX(ctrl, rmask_)                                                                                         \
    {                                                                                                                   \
        const uint32_t lo_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)k, ctrl, rmask_, 0xf, false);       \
        const uint32_t hi_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(k >> 32), ctrl, rmask_, 0xf, false); \
        const unsigned long long o_ = ((unsigned long long)hi_ << 32) | lo_;                                            \
        k = o_ > k ? o_ : k;                                                                                            \
    }
    HD_DPP_MAX(0x111, 0xf)   // row_shr:1
    HD_DPP_MAX(0x112, 0xf)   // row_shr:2
    HD_DPP_MAX(0x114, 0xf)   // row_shr:4
    HD_DPP_MAX(0x118, 0xf)   // row_shr:8   -> lane 15 of every row holds the row's maximum
    HD_DPP_MAX(0x142, 0xa)   // row_bcast:15 into rows 1 and 3
    HD_DPP_MAX(0x143, 0xc)   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's maximum
#undef HD_DPP_MAX
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)k, 63);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(k >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ uint32_t find_flag_lds(const unsigned long long* lmask, uint32_t base, uint32_t rmask,
                                                  uint32_t from, uint32_t to, bool want)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wr0 = (base + from) & ~63u;
    for (uint32_t it = 0;; ++it) {
        const uint32_t wstart = wr0 + it * 4096u;
        if ((int32_t)(wstart - base) >= (int32_t)to) break;
        const uint32_t wr = wstart + lane * 64u;
        const int32_t lw = (int32_t)(wr - base);
        unsigned long long w = 0;
        if (lw < (int32_t)to) {
            w = lmask[(wr & rmask) >> 6];
            if (!want) w = ~w;
            const int32_t lo = (int32_t)from - lw;
            if (lo > 0) w = lo >= 64 ? 0ull : (w & (~0ull << lo));
            const int32_t hi = (int32_t)to - lw;
            if (hi < 64) w &= (1ull << hi) - 1ull;
        }
        const unsigned long long hit = __ballot(w != 0ull);
        if (hit) {
            const int src = __ffsll((long long)hit) - 1;
            const unsigned long long ww = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(w >> 32), src) << 32) |
                                          (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)w, src);     // src is wave-uniform
            return (uint32_t)((int32_t)(wstart - base) + src * 64 + (__ffsll((long long)ww) - 1));
        }
    }
    return 0xFFFFFFFFu;
}

// The same search over a COMPACT image of the mask: lmask[i] is ring word (wr0 >> 6) + i, wr0 = the (64-aligned) ring position of the first
// word a search can touch -- the image then takes as many words as the searched backlog has, not as many as the ring (tail_body.h).
__device__ __forceinline__ uint32_t find_flag_rel(const unsigned long long* lmask, uint32_t base, uint32_t wr0,
                                                  uint32_t from, uint32_t to, bool want)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wrs = (base + from) & ~63u;
    for (uint32_t it = 0;; ++it) {
        const uint32_t wstart = wrs + it * 4096u;
        if ((int32_t)(wstart - base) >= (int32_t)to) break;
        const uint32_t wr = wstart + lane * 64u;
        const int32_t lw = (int32_t)(wr - base);
        unsigned long long w = 0;
        if (lw < (int32_t)to) {
            w = lmask[(wr - wr0) >> 6];
            if (!want) w = ~w;
            const int32_t lo = (int32_t)from - lw;
            if (lo > 0) w = lo >= 64 ? 0ull : (w & (~0ull << lo));
            const int32_t hi = (int32_t)to - lw;
            if (hi < 64) w &= (1ull << hi) - 1ull;
        }
        const unsigned long long hit = __ballot(w != 0ull);
        if (hit) {
            const int src = __ffsll((long long)hit) - 1;
            const unsigned long long ww = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(w >> 32), src) << 32) |
                                          (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)w, src);     // src is wave-uniform
            return (uint32_t)((int32_t)(wstart - base) + src * 64 + (__ffsll((long long)ww) - 1));
        }
    }
    return 0xFFFFFFFFu;
}

}  // namespace hd
