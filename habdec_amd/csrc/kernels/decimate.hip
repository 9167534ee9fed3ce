// FIR-decimate stage for batched IQ streams on gfx950 (CDNA4).
//
// What it computes (reference code/Decoder/Decimator.h:128-138): out[o] = sum_{t=0}^{T-1} buf[o*D + t] * tap[t]
// with buf = [history(T-1) | this call's input], complex sample times real tap, accumulated in ascending t
// with separately rounded multiply and add (this file is compiled with -ffp-contract=off, so the result is
// bit-identical to the CPU path).  The first stage is the only kernel of the pipeline that touches full-rate
// IQ: 8 bytes read per input sample, 8/D written -- HBM-bound for D >= 8 (kernel table).
//
// Mapping: one lane = one output sample, so the T-term sum stays inside a lane and in order.  A workgroup walks consecutive
// tiles: either `tiles_per_wg` tiles of one stream (grid = (tile groups, streams)), or -- linear split -- its even share of
// ALL the slab's tiles, crossing from one stream into the next where its range does.  Each tile's (TO-1)*D + T input
// samples go global -> registers -> LDS with 16-byte accesses, and the NEXT tile's global loads are issued before
// the current tile is computed, so HBM latency hides under the LDS/VALU phase (register prefetch, single LDS
// buffer, two barriers per tile).
// LDS layout: linear, with a 16-byte pad after every D samples (pitch D+2 samples per output).  Lane o reads
// samples o*D + t as 16-byte pairs at o*(D+2) + ...: a lane stride of 2*(D+2) dwords = 4 mod 64, i.e. consecutive
// lanes hit consecutive 16-byte bank groups -> ds_read_b128 / ds_write_b128 are conflict-free at full LDS rate.
// The tile is shifted by JS = (T-1)&1 slots so that 16-byte-aligned global pairs land on 16-byte-aligned LDS pairs.
// Taps are wave-uniform and come through the scalar cache (s_load), not LDS.
// History: the workgroup that owns a stream's last tile also writes the stream's next history (last T-1 inputs)
// into the OTHER history buffer (ping-pong, so the first tile of the same launch still reads the old one).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <type_traits>
#include <cstdlib>

#include "launch.h"
#include "tail_body.h"
#include "stage1_ring.h"

#define HD_DEC_MAXW 4     // most waves per SIMD the 256-lane stages are compiled for (their registers and LDS decide what they get: 3-4)

namespace hd {
namespace HD_ARITH_NS {

#ifdef HD_STAMP_DEC   // diagnostic build only (tools/micro/dec_stamps.py): phase clocks of the D = HD_STAMP_DEC_D (default 32) kernel, per workgroup
#ifndef HD_STAMP_DEC_D
#define HD_STAMP_DEC_D 32
#endif
__device__ unsigned long long g_dec_stamps[4096 * 8];
extern "C" void HD_DBG_NAME(hd_debug_dec_stamps)(unsigned long long* host, size_t n) { (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_dec_stamps), n * 8); }
#define DSTAMP_DECL unsigned long long ds_t = __builtin_amdgcn_s_memtime(), ds_acc[4] = {0, 0, 0, 0}; const unsigned long long ds_r0 = __builtin_amdgcn_s_memrealtime(); unsigned long long ds_r1 = 0
#define DSTAMP(i) do { if (D == HD_STAMP_DEC_D) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ds_acc[i] += t_ - ds_t; ds_t = t_; } } while (0)
#define DSTAMP_ARRIVED() do { if (D == HD_STAMP_DEC_D) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); DSTAMP(0); if (!ds_r1) ds_r1 = __builtin_amdgcn_s_memrealtime(); } } while (0)
#define DSTAMP_WRITE() do { if (D == HD_STAMP_DEC_D && threadIdx.x == 0) { const uint32_t w_ = by * gdx + bx; if (w_ < 4096) { \
        unsigned long long* g_ = g_dec_stamps + w_ * 8; g_[0] = ds_r0; g_[1] = ds_r1; g_[2] = __builtin_amdgcn_s_memrealtime(); \
        g_[3] = ds_acc[0]; g_[4] = ds_acc[1]; g_[5] = ds_acc[2]; g_[6] = ds_acc[3]; g_[7] = count; } } } while (0)
#else
#define DSTAMP_DECL do { } while (0)
#define DSTAMP(i) do { } while (0)
#define DSTAMP_ARRIVED() do { } while (0)
#define DSTAMP_WRITE() do { } while (0)
#endif

// Outputs per lane.  With a small ratio D neighbouring outputs share almost all of their T input samples, and one output per
// lane makes the kernel LDS-bound (8 bytes read per complex multiply-add).  A lane that owns OPL consecutive outputs reads its
// (OPL-1)*D + T samples once and feeds each to up to OPL accumulators; every accumulator still gets its own T products in
// ascending tap order.  OPL*D is 8 or 16 samples, so the padded pitch OPL*D + 2 keeps the 16-byte reads conflict-free.
template <int D> constexpr int dec_opl() { return D == 2 ? 4 : D == 4 ? 4 : D == 8 ? 2 : 1; }

template <int D, int T, int TO>
constexpr int dec_tile_f4()                        // float4 slots of a workgroup's LDS tile (+ this tile's outputs)
{
    constexpr int OPL = dec_opl<D>(), TOUT = TO * OPL, RD = OPL * D, JS = (T - 1) & 1;
    constexpr int NJJ = (TOUT - 1) * D + T + JS, NL = NJJ + 2 * (NJJ / RD) + 4;
    return (NL + 1) / 2 + TOUT / 2 + 1;
}

#ifdef HD_STAMP_TAIL  // diagnostic build only: the phase clocks of the tails that ran inside step launches (this translation unit's copy)
extern "C" void HD_DBG_NAME(hd_debug_step_tail_stamps)(unsigned long long* host, size_t n) { (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tail_stamps), n * 8); }
#endif

// The stage body.  (bx, by, gdx) are the workgroup's coordinates in the stage's own grid -- blockIdx / gridDim when the stage is a
// launch of its own (k_decimate), shifted when stream tails ride in front of it in the same launch (k_step).
template <int D, int T, int TO>
__device__ __forceinline__ void decimate_body(const float2* __restrict__ in, size_t in_stride,
                                              const float2* __restrict__ hist_in, float2* __restrict__ hist_out,
                                              const float* __restrict__ taps,
                                              float2* __restrict__ out, size_t out_stride,
                                              const StreamCall* __restrict__ call, int stage, int final_stage,
                                              uint32_t fir_hist_cap, uint32_t tiles_per_wg, float2* __restrict__ fft_in,
                                              uint32_t n_streams, uint32_t lin_ntiles, StreamCall* __restrict__ call_copy,
                                              const uint32_t bx, const uint32_t by, const uint32_t gdx, float4* __restrict__ tile4,
                                              const uint32_t uniform_n = 0 /* != 0: every stream brings this many samples and none restarts its history:
                                                                             the per-stream parameter block need not be read (it may sit across PCIe) */,
                                              const StepClaim claim = StepClaim{} /* runs of tiles handed out by per-XCD counters instead of a fixed share */)
{
    constexpr int OPL = dec_opl<D>();              // outputs per lane
    constexpr int TOUT = TO * OPL;                 // outputs per tile
    constexpr int RD = OPL * D;                    // samples per lane row (a 16-byte pad follows every row)
    constexpr int JS = (T - 1) & 1;                // LDS slot jj = j + JS for tile-local sample j
    constexpr int NJ = (TOUT - 1) * D + T;         // samples a tile needs
    constexpr int NJJ = NJ + JS;
    constexpr int NL = NJJ + 2 * (NJJ / RD) + 4;   // + two pad slots per row
    constexpr int NP = (NJJ + 1) / 2;              // 16-byte pairs per tile
    constexpr int ITER = (NP + TO - 1) / TO;
    static_assert((NL + 1) / 2 + TOUT / 2 + 1 == dec_tile_f4<D, T, TO>(), "tile size helper out of step");
    float2* tile = reinterpret_cast<float2*>(tile4);
    float2* ytile = tile + ((NL + 1) & ~1);        // this tile's outputs (only needed for the Q4 history quirk)

    // Two ways to hand out tiles.  Classic: grid = (tile groups, streams), a workgroup walks tiles_per_wg tiles of stream
    // blockIdx.y.  Linear split (lin_ntiles != 0; every stream has exactly lin_ntiles tiles of the same n): the S * lin_ntiles
    // tiles of the slab are dealt evenly to gridDim.x workgroups, whatever that number is -- it decides how many of a CU's LDS
    // slots this kernel occupies -- and a workgroup simply walks on into the next stream where its range crosses a seam.
    // Per-stream context (s, the stream's StreamCall head) therefore is a variable; the prefetch runs one tile ahead of it.
    struct CallHead { uint32_t n_in, n1, n2, zero_hist1, zero_hist2, pend_before, fft_fill, fft_take; };   // first 32 bytes of StreamCall
    static_assert(sizeof(CallHead) == 32, "CallHead mirrors the head of StreamCall");
    const bool linear = lin_ntiles != 0;
    uint32_t s, first, count;                               // current stream, first tile in it, tiles this workgroup walks in total
    // Claimed runs (step launches): the slab's tiles are cut into runs of claim.run_len consecutive tiles of one stream; XCD x owns the
    // runs [x * runs_per_xcd, (x + 1) * runs_per_xcd) and its workgroups draw them from the XCD's counter until none is left.  The draw
    // for the NEXT run is issued one tile before it is needed, in front of that tile's prefetch loads (returns are in order: by the
    // time the prefetched tile has arrived, so has the ticket), and the next run's first tile is prefetched like any other tile -- a
    // workgroup never sits through a cold start again, however unevenly the launch's slots free up.
    const bool claimed = claim.ctr != nullptr && claim.run_len >= 2;
    unsigned int* my_ctr = nullptr;
    unsigned int* next_ctr = nullptr;
    // A counter is only ever touched by its own XCD, with atomics (they execute in that XCD's L2): the workgroup that draws the first
    // ticket past the end -- exactly one per XCD and launch -- resets the XCD's counter of the other set for the next step launch.
    // (A plain store from whichever XCD runs workgroup 0 would leave this XCD's L2 with its stale copy.)
    auto retire = [&]() { if (threadIdx.x == 0) (void)__hip_atomic_exchange(next_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    // one draw per WAVE: lane 0 adds, everybody reads its answer (the value is only looked at a tile later -- the wait sits there)
    auto draw = [&]() -> unsigned int { unsigned int t = 0; if (threadIdx.x == 0) t = __hip_atomic_fetch_add(my_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return t; };
    auto drawn = [&](unsigned int t) -> unsigned int { return (unsigned int)__builtin_amdgcn_readfirstlane((int)t); };
    uint32_t xcd = 0;
    unsigned int ticket = 0;                                // the draw in flight (lane 0's register)
    bool jump = false; uint32_t jump_s = 0, jump_first = 0;
    if (claimed) {
        xcd = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u;       // XCC_ID
        if (xcd >= claim.n_xcd) xcd = claim.n_xcd - 1;
        my_ctr = claim.ctr + (size_t)xcd * 32;                                  // one counter per 128-byte line
        next_ctr = claim.ctr_next + (size_t)xcd * 32;
        const unsigned int r = drawn(draw());
        if (r >= claim.runs_per_xcd) { if (r == claim.runs_per_xcd) retire(); return; }
        const uint32_t g0 = (xcd * claim.runs_per_xcd + r) * claim.run_len;    // (the slab's tile count fits 32 bits: the host checks)
        s = g0 / lin_ntiles;
        first = g0 - s * lin_ntiles;
        count = claim.run_len;
    } else if (linear) {
        const uint64_t total = (uint64_t)n_streams * lin_ntiles;
        const uint64_t g0 = (uint64_t)bx * total / gdx, g1 = (uint64_t)(bx + 1) * total / gdx;
        s = (uint32_t)(g0 / lin_ntiles);
        first = (uint32_t)(g0 - (uint64_t)s * lin_ntiles);
        count = (uint32_t)(g1 - g0);
    } else {
        s = by; first = bx * tiles_per_wg; count = tiles_per_wg;
    }
    CallHead c{};
    if (uniform_n) { c.n_in = c.n1 = uniform_n; }            // (a workgroup's first tile is requested without waiting for a parameter fetch)
    else c = *reinterpret_cast<const CallHead*>(call + s);
    const uint32_t n = stage == 0 ? c.n_in : c.n1;          // (linear: the same for every stream)
    const uint32_t nout = n / D;
    const uint32_t ntiles = (nout + TOUT - 1) / TOUT;
    if (!linear) {
        if (!n && bx == 0)                                  // idle stream: its history passes through unchanged
            for (uint32_t j = threadIdx.x; j < (uint32_t)(T - 1); j += TO)
                hist_out[(size_t)s * (T - 1) + j] = hist_in[(size_t)s * (T - 1) + j];
        if (first >= ntiles) return;
        count = min(first + tiles_per_wg, ntiles) - first;
    }
    if (!count) return;
    // prefetch context: the stream the NEXT tile to load belongs to
    uint32_t pf_s = s;
    bool pf_zero_hist = (stage == 0 ? c.zero_hist1 : c.zero_hist2) != 0;
    CallHead c_next = c;                                    // head of stream s + 1 once the prefetch has crossed the seam

    // Edge tiles (the history tile of every stream): branch-free address selection so the loads still issue
    // back to back -- out-of-range samples read a safe address and are zeroed afterwards.
    float4 r[ITER];
    auto load_tile = [&](uint32_t tile_i, const bool fast) {   // tile tile_i of stream pf_s; fast: try the plain path for interior tiles
        const float2* in_s = in + (size_t)pf_s * in_stride;
        const float2* hist_s = hist_in + (size_t)pf_s * (T - 1);
        const bool zero_hist = pf_zero_hist;
        auto locate = [&](long xi, bool& ok) -> const float2* {
            const bool in_hist = xi < 0;
            ok = in_hist ? (!zero_hist && xi >= -(long)(T - 1)) : (xi < (long)n);
            const long hi = xi + (T - 1);
            const float2* ph = hist_s + (hi < 0 ? 0 : hi);
            const float2* pi = in_s + (xi < 0 ? 0 : (xi < (long)n ? xi : 0));
            return in_hist ? ph : pi;
        };
        const long xe = (long)tile_i * TOUT * D - (T - 1) - JS;    // stream sample of LDS slot 0; even
        const float4* src = reinterpret_cast<const float4*>(in_s + xe);
        // A tile that lies wholly inside this call's input -- all but a stream's first and (sometimes) last: one scalar test, then ITER
        // plain loads off one scalar base with immediate offsets.  (Tested per sweep, the compiler kept the bounds in vector registers
        // and put ten instructions between consecutive loads.)
        if (fast && __builtin_amdgcn_readfirstlane((int)(xe >= 0 && xe + 2L * ITER * TO <= (long)n))) {
            const uint64_t b64 = reinterpret_cast<uint64_t>(src);
            const char* base = reinterpret_cast<const char*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b64 >> 32)) << 32) |
                                                             (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b64));
            const uint32_t lane_off = threadIdx.x * 16u;
#pragma unroll
            for (int it = 0; it < ITER; ++it) r[it] = *reinterpret_cast<const float4*>(base + (lane_off + (uint32_t)it * (TO * 16u)));
            return;
        }
        // Per sweep of TO pairs: plain aligned 16-byte loads when the whole sweep lies inside this call's input (wave-uniform
        // test); only the sweeps that touch the history in front of the stream (first tile: the first ceil((T-1)/2/TO) sweeps)
        // or the end of the input take the address-selecting path.
        // (The lane number is made opaque here: with it visible the compiler hoists every sweep's `k < NP` and address comparisons out of the tile loop as
        // loop invariants -- dozens of scalar register pairs live across the loop, parked in vector lanes, and in k_step<32,212,2,69> two vector registers
        // spilled to scratch for it.  Recomputing a few compares per edge tile is free.)
        uint32_t tid_ = threadIdx.x;
        asm volatile("" : "+v"(tid_));
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int k = (int)tid_ + it * TO;
            const long lo = xe + 2L * it * TO, hi = lo + 2L * TO;
            if (lo >= 0 && hi <= (long)n) {
                r[it] = src[k];                              // (k < NP or not: the extra pairs of the last sweep are in range and unused)
            } else if (k < NP) {
                const long xi = xe + 2 * k;
                bool oka, okb;
                const float2* pa = locate(xi, oka);
                const float2* pb = locate(xi + 1, okb);
                float2 a = *pa, b = *pb;
                if (!oka) a = make_float2(0.f, 0.f);
                if (!okb) b = make_float2(0.f, 0.f);
                r[it] = make_float4(a.x, a.y, b.x, b.y);
            }
        }
    };

    DSTAMP_DECL;
    load_tile(first, false);                                // (a workgroup's first tile: once, through the general path only)
    const float2* p = tile + threadIdx.x * (RD + 2);
    // A tile's outputs are stored one iteration late, just BEFORE the next prefetch is issued: loads and stores retire
    // through one in-order counter, so a store issued after the prefetch would make the wait for the prefetched tile also
    // wait for the store's acknowledgement (measured: 8 % of the kernel).
    float2 y_prev[OPL];
    float2* dst_prev = nullptr;                             // where y_prev[0] goes; the lane's other outputs follow it
    float2* fdst_prev = nullptr;
    uint32_t nv_prev = 0, nf_prev = 0;                      // how many of the lane's outputs exist / belong to the spectrum feed
    auto store_prev = [&]() {
#pragma unroll
        for (int q = 0; q < OPL; ++q) {
            if ((uint32_t)q < nv_prev) dst_prev[q] = y_prev[q];
            if ((uint32_t)q < nf_prev) fdst_prev[q] = y_prev[q];   // spectrum input collection (reference Decoder.h:467-473)
        }
        nv_prev = nf_prev = 0;
    };
    uint32_t tile_i = first;                                // tile of stream s being computed; pf_* run one tile ahead
    // The call's parameters live in mapped host memory; whoever owns a stream's first tile leaves a device copy for the kernels
    // that read them later (the stream tail of this call runs inside the NEXT call's launch).
    // (The read crosses PCIe: it is requested here and stored a tile later, so nothing waits for it.)
    uint4 cpy = make_uint4(0, 0, 0, 0);
    uint32_t cpy_s = 0xFFFFFFFFu;
    auto leave_copy = [&](uint32_t sc) {
        if (call_copy) { cpy_s = sc; if (threadIdx.x < 4) cpy = reinterpret_cast<const uint4*>(call + sc)[threadIdx.x]; }
    };
    auto flush_copy = [&]() {
        if (cpy_s != 0xFFFFFFFFu) { if (threadIdx.x < 4) reinterpret_cast<uint4*>(call_copy + cpy_s)[threadIdx.x] = cpy; cpy_s = 0xFFFFFFFFu; }
    };
    if (first == 0) leave_copy(s);
    for (uint32_t done = 0; done < count; ++done) {
        DSTAMP_ARRIVED();
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int k = threadIdx.x + it * TO;
            if ((it + 1) * TO <= NP || k < NP) {                // (only the last sweep is partial: no per-sweep lane masks for the others)
                const int jj = 2 * k;
                *reinterpret_cast<float4*>(tile + jj + 2 * (jj / RD)) = r[it];
            }
        }
        __syncthreads();
        DSTAMP(1);
        store_prev();
        if (done) flush_copy();
        bool pf_do = false; uint32_t pf_which = 0;
        if (claimed && done + 2 == count)                   // one tile before the run ends: draw the next run (in front of the prefetch loads)
            ticket = draw();
        if (claimed && done + 1 == count) {                 // the run's last tile: prefetch the first tile of the run just drawn, if there is one
            const unsigned int tk = drawn(ticket);
            if (tk == claim.runs_per_xcd) retire();
            if (tk < claim.runs_per_xcd) {
                const uint32_t g0 = (xcd * claim.runs_per_xcd + tk) * claim.run_len;
                pf_s = g0 / lin_ntiles;
                const uint32_t pf_tile = g0 - pf_s * lin_ntiles;
                pf_do = true; pf_which = pf_tile;
                count = (uint32_t)__builtin_amdgcn_readfirstlane((int)(count + claim.run_len));   // (stays a scalar: the loop bound)
                jump = true; jump_s = pf_s; jump_first = pf_tile;
            }
        } else
        if (done + 1 < count) {                             // the next tile's loads stay in flight while this tile is computed
            uint32_t pf_tile = tile_i + 1;
            if (linear && pf_tile == ntiles) {              // ... across the seam into the next stream
                pf_tile = 0; ++pf_s;
                if (!uniform_n) c_next = *reinterpret_cast<const CallHead*>(call + pf_s);
                pf_zero_hist = (stage == 0 ? c_next.zero_hist1 : c_next.zero_hist2) != 0;
            }
            pf_do = true; pf_which = pf_tile;
        }
        if (pf_do) load_tile(pf_which, true);               // (one call site for the loop: the loader is big, and inlined)
        DSTAMP(2);

        // The T-term sum, in tap order.  Taps are consumed in blocks of B LDS slots (B/2 ds_read_b128 issued
        // together, then 2*B packed multiply/add) inside a rolled loop: that keeps ~B taps live in SGPRs instead
        // of all T (which spilled SGPRs through v_writelane) and puts B/2 LDS reads in flight per wave.
        float2 yq[OPL];
        if constexpr (OPL == 1 && D >= 32 && (D % 16) == 0) {
            // Single-wave first stages (/32, /64): 16-slot chunks (never across a row pad), the next chunk's LDS reads and taps requested
            // before the current chunk is summed (two register images, rolled loop over chunk pairs), products ahead of the adds.
            constexpr int CH = 16;
            constexpr int NS = T + JS;                          // slots [JS, NS) carry taps [0, T)
            constexpr int NCH = NS / CH;                        // full chunks; chunk 0 starts at slot JS
            static_assert(NCH >= 3, "filter shorter than three chunks");
            static_assert((TO - 1) * (RD + 2) + (NCH * CH + CH - 1) + 2 * ((NCH * CH + CH - 1) / D) < 2 * dec_tile_f4<D, T, TO>(),
                          "the look-ahead read of the last chunk must stay inside the workgroup's LDS");
            f32x2 acc = {0.f, 0.f};
            f32x4 xa[8], xb[8];
            f32x2 ka[8], kb[8];                                 // the chunk's taps as eight pairs (wave-uniform: scalar registers), requested with its samples
            auto rd = [&](f32x4 (&x)[8], f32x2 (&k)[8], const int c, auto j0, auto j1) {
                const float2* pc = p + c * CH + 2 * ((c * CH) / D);
#pragma unroll
                for (int q = 0; q < 8; ++q) x[q] = *reinterpret_cast<const f32x4*>(pc + 2 * q);
                const float* tb = taps + (c * CH - JS);
#pragma unroll
                for (int j = 0; j < 16; j += 2) {
                    k[j >> 1].x = (j >= decltype(j0)::value && j < decltype(j1)::value) ? tb[j] : 0.f;
                    k[j >> 1].y = (j + 1 >= decltype(j0)::value && j + 1 < decltype(j1)::value) ? tb[j + 1] : 0.f;
                }
            };
            auto mac_part = [&](const f32x4 (&x)[8], const f32x2 (&k)[8], auto j0, auto j1) {    // a chunk only part of whose slots carry taps
                HD_FIR_ARITH
#pragma unroll
                for (int j = decltype(j0)::value; j < decltype(j1)::value; ++j) {
                    const f32x2 smp = (j & 1) ? x[j >> 1].zw : x[j >> 1].xy;
                    acc = acc + smp * ((j & 1) ? k[j >> 1].y : k[j >> 1].x);
                }
            };
            using I0 = std::integral_constant<int, 0>;
            using I16 = std::integral_constant<int, 16>;
            using IJS = std::integral_constant<int, JS>;
            using IRem = std::integral_constant<int, NS % CH>;
            rd(xa, ka, 0, IJS{}, I16{});
            rd(xb, kb, 1, I0{}, I16{});
            if constexpr (JS == 0) ring_mac16_asm(acc, xa, ka); else mac_part(xa, ka, IJS{}, I16{});
            int c = 1;
#pragma unroll 1
            for (; c + 1 < NCH; c += 2) {                       // chunk c is in xb
                rd(xa, ka, c + 1, I0{}, I16{});
                ring_mac16_asm(acc, xb, kb);                    // (sixteen taps with the products three ahead of the adds: stage1_ring.h)
                if (c + 2 < NCH) rd(xb, kb, c + 2, I0{}, I16{});
                else rd(xb, kb, NCH, I0{}, IRem{});             // the partial chunk behind the last pair (or nothing)
                ring_mac16_asm(acc, xa, ka);
            }
            if constexpr ((NCH - 1) % 2) {                      // one full chunk left over (in xb)
                rd(xa, ka, NCH, I0{}, IRem{});
                ring_mac16_asm(acc, xb, kb);
                if constexpr (NS % CH) mac_part(xa, ka, I0{}, IRem{});
            } else {
                if constexpr (NS % CH) mac_part(xb, kb, I0{}, IRem{});
            }
            yq[0] = make_float2(acc.x, acc.y);
        } else
        if constexpr (OPL == 1) {
            float ar = 0.f, ai = 0.f;
            auto mac = [&](float xr, float xi, float k) { HD_FIR_ARITH ar = ar + xr * k; ai = ai + xi * k; };
            constexpr int B = D >= 32 ? 32 : 16;               // slots per block; pad inside a block is compile-time
            constexpr int NS = T + JS;                          // slots [JS, NS) carry taps [0, T)
            constexpr int NFULL = NS / B;                       // full blocks; block 0 is peeled when JS (slot 0 unused)
            // block 0 (peeled): slots [0, B) or the whole filter when it is shorter than a block
            {
                constexpr int END = NS < B ? NS : B;
    #pragma unroll
                for (int jj = 0; jj + 1 < END + 1; jj += 2) {
                    if (jj + 1 < END || jj < END) {
                        const float4 x = *reinterpret_cast<const float4*>(p + jj + 2 * (jj / D));
                        if (jj >= JS && jj < END) mac(x.x, x.y, taps[jj - JS]);
                        if (jj + 1 < END) mac(x.z, x.w, taps[jj + 1 - JS]);
                    }
                }
            }
            if (NFULL > 1) {
    #pragma unroll 1
                for (int b = 1; b < NFULL; ++b) {
                    const int j0 = b * B;
                    const float2* pb = p + j0 + 2 * (j0 / D);
                    const float* tb = taps + (j0 - JS);
                    float4 x[B / 2];
    #pragma unroll
                    for (int q = 0; q < B / 2; ++q) x[q] = *reinterpret_cast<const float4*>(pb + 2 * q + 2 * ((2 * q) / D));
    #pragma unroll
                    for (int q = 0; q < B / 2; ++q) {
                        mac(x[q].x, x[q].y, tb[2 * q]);
                        mac(x[q].z, x[q].w, tb[2 * q + 1]);
                    }
                }
            }
            if (NS >= B && NS % B) {                            // tail block: slots [NFULL*B, NS)
                constexpr int J0 = NFULL * B;
    #pragma unroll
                for (int jj = J0; jj < NS; jj += 2) {
                    const float4 x = *reinterpret_cast<const float4*>(p + jj + 2 * (jj / D));
                    mac(x.x, x.y, taps[jj - JS]);
                    if (jj + 1 < NS) mac(x.z, x.w, taps[jj + 1 - JS]);
                }
            }
            yq[0] = make_float2(ar, ai);
        } else {
            // Lane-relative slot r holds tile sample OPL*tid*D + r - JS; output q of the lane takes it with tap r - JS - q*D.
            float ar[OPL], ai[OPL];
#pragma unroll
            for (int q = 0; q < OPL; ++q) { ar[q] = 0.f; ai[q] = 0.f; }
            constexpr int B = 16;                                       // slots per block (a multiple of the row, so pads are compile-time)
            constexpr int W = (OPL - 1) * D + T + JS;                   // slots a lane reads
            constexpr int NB = (W + B - 1) / B;
            constexpr int B_LO = (JS + (OPL - 1) * D + B - 1) / B;      // first block in which every output takes every slot
            constexpr int B_HI = (T + JS) / B - 1;                      // last such block
            auto edge_block = [&](auto bc) {                            // head / tail blocks: per (slot, output) compile-time guards
                HD_FIR_ARITH
                constexpr int j0 = decltype(bc)::value * B;
#pragma unroll
                for (int k = 0; k < B; k += 2) {
                    if (j0 + k < W) {
                        const float4 x = *reinterpret_cast<const float4*>(p + j0 + k + 2 * ((j0 + k) / RD));
#pragma unroll
                        for (int q = 0; q < OPL; ++q) {
                            const int t0 = j0 + k - JS - q * D, t1 = t0 + 1;
                            if (t0 >= 0 && t0 < T) { ar[q] = ar[q] + x.x * taps[t0]; ai[q] = ai[q] + x.y * taps[t0]; }
                            if (t1 >= 0 && t1 < T) { ar[q] = ar[q] + x.z * taps[t1]; ai[q] = ai[q] + x.w * taps[t1]; }
                        }
                    }
                }
            };
            auto for_blocks = [&](auto lo, auto hi, auto self) {        // compile-time loop lo .. hi-1 over edge_block
                if constexpr (decltype(lo)::value < decltype(hi)::value) {
                    edge_block(lo);
                    self(std::integral_constant<int, decltype(lo)::value + 1>{}, hi, self);
                }
            };
            constexpr int HEAD_END = B_LO < NB ? (B_LO <= B_HI ? B_LO : NB) : NB;     // blocks [0, HEAD_END) are head blocks
            for_blocks(std::integral_constant<int, 0>{}, std::integral_constant<int, HEAD_END>{}, for_blocks);
            if constexpr (B_LO <= B_HI) {
#pragma unroll 1
                for (int b = B_LO; b <= B_HI; ++b) {                    // interior: rolled, ~B + (OPL-1)*D taps live in scalar registers
                    HD_FIR_ARITH
                    const int j0 = b * B;
                    const float2* pb = p + j0 + 2 * (j0 / RD);
                    const float* tb = taps + (j0 - JS);
                    float4 x[B / 2];
#pragma unroll
                    for (int k = 0; k < B / 2; ++k) x[k] = *reinterpret_cast<const float4*>(pb + 2 * k + 2 * ((2 * k) / RD));
#pragma unroll
                    for (int k = 0; k < B / 2; ++k) {
#pragma unroll
                        for (int q = 0; q < OPL; ++q) {
                            const float k0 = tb[2 * k - q * D], k1 = tb[2 * k + 1 - q * D];
                            ar[q] = ar[q] + x[k].x * k0; ai[q] = ai[q] + x[k].y * k0;
                            ar[q] = ar[q] + x[k].z * k1; ai[q] = ai[q] + x[k].w * k1;
                        }
                    }
                }
                for_blocks(std::integral_constant<int, B_HI + 1>{}, std::integral_constant<int, NB>{}, for_blocks);
            }
#pragma unroll
            for (int q = 0; q < OPL; ++q) yq[q] = make_float2(ar[q], ai[q]);
        }
        const uint32_t o = (tile_i * TO + threadIdx.x) * OPL;           // the lane's first output
#pragma unroll
        for (int q = 0; q < OPL; ++q) { y_prev[q] = yq[q]; ytile[threadIdx.x * OPL + q] = yq[q]; }
        nv_prev = o < nout ? min((uint32_t)OPL, nout - o) : 0u;
        dst_prev = out + (size_t)s * out_stride + (final_stage ? (size_t)fir_hist_cap + c.pend_before : 0) + o;
        nf_prev = (fft_in && o < c.fft_take) ? min((uint32_t)OPL, c.fft_take - o) : 0u;
        fdst_prev = fft_in ? fft_in + (size_t)s * kFftBins + c.fft_fill + o : nullptr;
        __syncthreads();                                    // everyone is done with this tile's LDS image
        DSTAMP(3);
        ++tile_i;
        const bool seam = tile_i == ntiles;
        if (seam) {
            // History carry for the next call (Decimator.h:140-143).  Q4: the reference decimates in place
            // (Decoder.h:443-444), so history positions that fall inside the first n/D samples hold OUTPUTS; that can only
            // happen for inputs so short that all outputs are in this (single) tile.
            if (n) {
                float2* hout = hist_out + (size_t)s * (T - 1);
                const float2* in_s = in + (size_t)s * in_stride;
                for (uint32_t j = threadIdx.x; j < (uint32_t)(T - 1); j += TO) {
                    const uint32_t idx = n - (T - 1) + j;   // host guarantees n >= T-1
                    hout[j] = idx < nout ? ytile[idx] : in_s[idx];
                }
            }
            if (!jump && done + 1 < count) {                // linear split: walk on into the next stream (its first tile is in flight)
                __syncthreads();                            // ytile is rewritten by the next tile
                ++s; tile_i = 0; c = c_next;
                leave_copy(s);
            }
        }
        if (jump) {                                         // claimed runs: on to the run drawn a tile ago (its first tile is in flight)
            __syncthreads();
            s = jump_s; tile_i = jump_first; jump = false;
            if (!tile_i) leave_copy(s);
        }
    }
    store_prev();
    flush_copy();
    DSTAMP_WRITE();
}

template <int D, int T, int TO>
__global__ __launch_bounds__(TO) __attribute__((amdgpu_waves_per_eu(D == 64 ? 1 : 2, TO == 64 ? 2 : HD_DEC_MAXW))) void k_decimate(const float2* __restrict__ in, size_t in_stride,
                                                   const float2* __restrict__ hist_in, float2* __restrict__ hist_out,
                                                   const float* __restrict__ taps,
                                                   float2* __restrict__ out, size_t out_stride,
                                                   const StreamCall* __restrict__ call, int stage, int final_stage,
                                                   uint32_t fir_hist_cap, uint32_t tiles_per_wg, float2* __restrict__ fft_in,
                                                   uint32_t n_streams, uint32_t lin_ntiles, StreamCall* __restrict__ call_copy,
                                                   const uint32_t uniform_n, const StepClaim claim)
{
    __shared__ float4 tile4[dec_tile_f4<D, T, TO>()];
    decimate_body<D, T, TO>(in, in_stride, hist_in, hist_out, taps, out, out_stride, call, stage, final_stage, fir_hist_cap, tiles_per_wg, fft_in,
                            n_streams, lin_ntiles, call_copy, blockIdx.x, blockIdx.y, gridDim.x, tile4, TO == 64 ? uniform_n : 0u,
                            TO == 64 ? claim : StepClaim{});
}

// One launch per step in batch mode: workgroups [0, n_tail) are the stream tails of the PREVIOUS call (tail_body.h: stage 2, low-pass,
// discriminator, symbol extractor -- one wave per stream), workgroups [n_tail, gridDim.x) share this call's stage-1 tiles as a linear
// split.  Every workgroup is one wave with the same register and LDS footprint, eight slots per CU; the tails are dispatched first
// and finish after a fraction of the launch, the hardware dispatcher hands their slots to stage-1 workgroups as they free up, and the
// HBM-bound stage-1 waves that are resident from the start keep the memory system busy meanwhile.  No second queue, no event waits.
template <int D, int T, int D2, int T2>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(D == 64 ? 1 : 2, 2))) void k_step(const float2* __restrict__ in, size_t in_stride,
                                                   const float2* __restrict__ hist_in, float2* __restrict__ hist_out,
                                                   const float* __restrict__ taps,
                                                   float2* __restrict__ out, size_t out_stride,
                                                   const StreamCall* __restrict__ call, uint32_t n_streams, uint32_t lin_ntiles,
                                                   StreamCall* __restrict__ call_copy, const TailArgs ta, const uint32_t n_tail,
                                                   const uint32_t uniform_n, const StepClaim claim)
{
    constexpr int kF4 = dec_tile_f4<D, T, 64>() > (int)(kStepLdsBytes / 16) ? dec_tile_f4<D, T, 64>() : (int)(kStepLdsBytes / 16);
    __shared__ float4 tile4[kF4];
    if (blockIdx.x < n_tail) {
        // the tails are latency chains with nobody to hide behind; the stage-1 waves beside them are waiting for HBM most of the
        // time and lose nothing when the arbiter prefers the tail
#define HD_STEP_PRIO 3
        __builtin_amdgcn_s_setprio(HD_STEP_PRIO);
        tail_body<64, 4, D2, T2>(ta, blockIdx.x, reinterpret_cast<unsigned char*>(tile4));
        return;
    }
    decimate_body<D, T, 64>(in, in_stride, hist_in, hist_out, taps, out, out_stride, call, 0, 0, 0u, 0u, nullptr, n_streams, lin_ntiles, call_copy,
                            blockIdx.x - n_tail, 0u, gridDim.x - n_tail, tile4, uniform_n, claim);
}

// The step launch as ONE workgroup per CU (512 threads, all of the CU's LDS), two waves per SIMD: a stage-1 WORKER (stage1_ring.h: ring_worker --
// it loads its own tiles with LDS-DMA into its own 64-row slot and sums them with the systolic tap loop) and a stream tail of the previous call --
// four streams per CU at 1024 streams, each in its own slice of LDS -- which leaves the device copy of its stream's parameter block for the next
// launch and then becomes a worker too, its slice the slot.  Compared with k_step
// (single-wave workgroups, dispatcher-scheduled) stage 1's loads never stop while the tails hold half of the CU's wave slots, a tile in flight costs
// no registers, and what runs where does not depend on the dispatcher; compared with rounds 3-4 (one loader wave feeding computing waves through
// shared slots) nothing is shared between the waves of a CU and the loads in flight grow with the waves that exist.
template <int T, int D2, int T2>
__global__ __launch_bounds__(512) void k_step_cu(const RingArgs ra, const StreamCall* __restrict__ call, StreamCall* __restrict__ call_copy,
                                                 const TailArgs ta, const uint32_t n_tail, const uint32_t n_streams, const uint32_t tail_bytes /* >= kWorkSlotBytes */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char cu_lds[];
    RingCtl* ctl = reinterpret_cast<RingCtl*>(cu_lds);
    unsigned char* slots = cu_lds + kRingCtlBytes;                           // worker slots of roles 0-3 (0-7 in a launch without tails)
    unsigned char* tails = slots + 4 * kWorkSlotBytes;                       // four tail slices of tail_bytes each
    ring_ctl_init(ctl, 0u);
    __syncthreads();
    // Roles by SIMD, not by wave number.  A 512-thread workgroup at 256 VGPRs puts exactly two waves on each of the CU's four SIMDs; which two
    // is the hardware's choice.  The first wave to arrive on a SIMD (an LDS counter) takes the SIMD's first role.
    const uint32_t simd = (__builtin_amdgcn_s_getreg((2 - 1) << 11 | 4 << 6 | 4)) & 3u;                    // HW_ID.SIMD_ID
    uint32_t rank = 0;
    if ((threadIdx.x & 63u) == 0) rank = __hip_atomic_fetch_add(&ctl->simd_rank[simd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    rank = (uint32_t)__builtin_amdgcn_readfirstlane((int)rank);
    // Every SIMD gets ONE worker and ONE tail: two tails on a SIMD slow each other by a fifth (83 against 102 us alone, round 3), two workers on a SIMD halve
    // the second one's tap loop (the arbiter prefers the older wave: 2.4k against 4.8k ticks per tile, in-kernel clocks) -- a tail and a worker fill each
    // other's gaps (one box, alternating, sustained: 0.1300-0.1311 ms per step against 0.1318-0.1326 with workers on SIMDs 0-1 and tails on 2-3).
    uint32_t w = rank ? 4u + simd : simd;                                                                  // workers 0-3, tails 4-7
    // (two waves per SIMD is what the register budget gives; should the hardware ever place a third on a SIMD, that wave takes one of the roles
    // nobody claimed -- every role must be filled exactly once, whatever the placement.)
    if (rank < 2u && (threadIdx.x & 63u) == 0) (void)__hip_atomic_fetch_or(&ctl->roles_taken, 1u << w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    if (rank >= 2u) {
        uint32_t pick = 0;
        if ((threadIdx.x & 63u) == 0) {
            for (;;) {                                                       // claim the lowest role still free
                const uint32_t taken = __hip_atomic_load(&ctl->roles_taken, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                pick = (uint32_t)__builtin_ctz(~taken);
                if (!(__hip_atomic_fetch_or(&ctl->roles_taken, 1u << pick, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & (1u << pick))) break;
            }
        }
        w = (uint32_t)__builtin_amdgcn_readfirstlane((int)pick);
    }
    if (w < 4u) {
        ring_worker<T>(ra, slots + w * (uint32_t)kWorkSlotBytes, w);
    } else {
        const uint32_t k = w - 4u, lane = threadIdx.x & 63u;
        const uint32_t s = blockIdx.x * 4u + k;                          // (the grid has at least n_streams / 4 workgroups)
        if (s < n_streams) {
            if (s < n_tail) {
                __builtin_amdgcn_s_setprio(HD_STEP_PRIO);
                tail_body<64, 4, D2, T2>(ta, s, tails + k * tail_bytes);           // (the latency chain of the SIMD at the higher priority; the worker beside it takes the issue slots it leaves)
                __builtin_amdgcn_s_setprio(0);
            }
            // this call's parameters live in mapped host memory; the tails of this call (next launch) read the device copy
            if (call_copy && lane < 4) reinterpret_cast<uint4*>(call_copy + s)[lane] = reinterpret_cast<const uint4*>(call + s)[lane];
        }
        // the tail is done: one more worker -- its slot is the tail's own slice (a launch without tails: the ring region's slots 4-7)
        ring_worker<T>(ra, n_tail ? tails + k * tail_bytes : slots + w * (uint32_t)kWorkSlotBytes, w);
    }
}

// Stage 1 ALONE in the same shape (synchronous delivery, the first half of a call whose tails run as a launch of their own): one workgroup per CU.
// D = 32: eight worker waves, each with its own slot (ring_worker); D = 64 (the /64 first stage of /256 plans: rows of 64 samples, 33 KB slots): four, or
// three where the previous call's 4097-tap low-pass on the other queue needs its 45 KB of the CU's LDS.  D = 8 (the /8 first stage of /16 plans, four outputs per lane row) and D = 4 (the
// only stage of a /4 plan, eight outputs per lane): a lane's window spans several rows there and the slot stays busy while it is summed, so these keep
// the loader / consumer arrangement -- one or two LDS-DMA loader waves with up to eight tile slots between them, computing waves in all the others.
template <int T, int D>
__global__ __launch_bounds__(D == 4 ? 1024 : D == 64 ? 256 : 512) void k_stage1_cu(const RingArgs ra, const uint32_t n_loaders, const uint32_t n_slots /* 2 .. 8 */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char cu_lds[];
    if constexpr (D == 32 || D == 64) {
        ring_worker<T, D>(ra, cu_lds + (threadIdx.x >> 6) * (uint32_t)work_slot_bytes<D>(), threadIdx.x >> 6);
    } else {
    unsigned char* ring = cu_lds;
    RingCtl* ctl = reinterpret_cast<RingCtl*>(cu_lds + n_slots * ring_slot_bytes<T>());
    ring_ctl_init(ctl, n_loaders);
    __syncthreads();
    // one loader per SIMD pair, the first computing wave beside a loader feeds the runs (roles by SIMD; any placement fills every role)
    const uint32_t simd = (__builtin_amdgcn_s_getreg((2 - 1) << 11 | 4 << 6 | 4)) & 3u;
    uint32_t rank = 0;
    if ((threadIdx.x & 63u) == 0) rank = __hip_atomic_fetch_add(&ctl->simd_rank[simd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    rank = (uint32_t)__builtin_amdgcn_readfirstlane((int)rank);
    uint32_t w = simd < 2u ? (rank ? 2u + simd : simd) : 4u + 2u * (simd - 2u) + rank;
    if (rank < 2u && (threadIdx.x & 63u) == 0) (void)__hip_atomic_fetch_or(&ctl->roles_taken, 1u << w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    if (rank >= 2u) {
        uint32_t pick = 0;
        if ((threadIdx.x & 63u) == 0) {
            for (;;) {
                const uint32_t taken = __hip_atomic_load(&ctl->roles_taken, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                pick = (uint32_t)__builtin_ctz(~taken);
                if (!(__hip_atomic_fetch_or(&ctl->roles_taken, 1u << pick, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & (1u << pick))) break;
            }
        }
        w = (uint32_t)__builtin_amdgcn_readfirstlane((int)pick);
    }
    const RingGeom geom{ring, n_slots};
    const uint32_t h0 = n_slots / 2u;                                    // two loaders: [0, h0) and [h0, n_slots)
    if (w < n_loaders) ring_loader<T>(ra, geom, ctl, (n_loaders == 1u || w == 0u) ? 0u : h0, n_loaders == 1u ? n_slots : (w == 0u ? h0 : n_slots - h0), w);
    else ring_consumer<T, D>(ra, geom, ctl, w == 2, w);
    }
}

// XCC ids seen by a grid of single-wave workgroups: the run counters of the step launches are per XCD and indexed by the hardware's id
// (a CU mask or a partition mode that hides an XCD, or numbers it differently, must switch the drawn runs off -- engine.cpp).
__global__ void k_xcc_probe(unsigned int* mask)
{
    if (threadIdx.x == 0) atomicOr(mask, 1u << (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u));
}

__global__ void k_passthrough(const float2* __restrict__ in, size_t in_stride, float2* __restrict__ out, size_t out_stride,
                              const StreamCall* __restrict__ call, uint32_t fir_hist_cap)
{
    const uint32_t s = blockIdx.y;
    const StreamCall c = call[s];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < c.n_in) out[(size_t)s * out_stride + fir_hist_cap + c.pend_before + i] = in[(size_t)s * in_stride + i];
}

// Per-chunk DC blocker (reference Decoder.h:450-459): wp = .97*x0; w = x + .97*wp; y = w - wp; wp = w.
// Strictly sequential per stream; one wave per stream walks the chunk 64 samples at a time, broadcasting
// each sample with v_readlane so the recurrence runs uniformly in every lane without LDS or barriers.
__global__ __launch_bounds__(64) void k_dc_remove(float2* __restrict__ fbuf, size_t stride, const StreamCall* __restrict__ call,
                                                    uint32_t fir_hist_cap)
{
    const uint32_t s = blockIdx.x;
    const StreamCall c = call[s];
    if (!c.dc_remove || !c.n2) return;
    float2* x = fbuf + (size_t)s * stride + fir_hist_cap + c.pend_before;
    const int lane = threadIdx.x;
    const float2 x0 = x[0];
    float wr = 0.97f * x0.x, wi = 0.97f * x0.y;
    for (uint32_t base = 0; base < c.n2; base += 64) {
        const uint32_t i = base + lane;
        float2 v = i < c.n2 ? x[i] : make_float2(0.f, 0.f);
        float yr = 0.f, yi = 0.f;
#pragma unroll 8                                 // (fully unrolled the compiler hoisted all 128 broadcasts to the front: 130 spilled SGPRs)
        for (int k = 0; k < 64; ++k) {
            const float xr = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v.x), k));
            const float xi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v.y), k));
            const float sr = 0.97f * wr, si = 0.97f * wi;
            const float nr = xr + sr, ni = xi + si;
            if (lane == k) { yr = nr - wr; yi = ni - wi; }
            // lanes past the end of the chunk must not disturb the carry (their outputs are not stored)
            if (base + k < c.n2) { wr = nr; wi = ni; }
        }
        if (i < c.n2) x[i] = make_float2(yr, yi);
    }
}

template <int D, int T, int TO>
static void launch_one(hipStream_t st, uint32_t n_streams, uint32_t max_out, const float2* in, size_t in_stride,
                       const float2* hist_in, float2* hist_out, const float* taps, float2* out, size_t out_stride,
                       const StreamCall* call, int stage, int final_stage, uint32_t fir_hist_cap, float2* fft_in, uint32_t lin_wgs, StreamCall* call_copy,
                       uint32_t uniform_n, const StepClaim& claim)
{
    constexpr uint32_t TOUT = TO * dec_opl<D>();
    const uint32_t ntiles = (max_out + TOUT - 1) / TOUT;
    // Linear split (single-wave instantiations, every stream the same size -- the caller vouches for that by passing lin_wgs):
    // exactly lin_wgs workgroups share the slab's tiles evenly.  The caller picks lin_wgs = k * CUs to decide how many of a CU's
    // eight 19.5 KB LDS slots this kernel takes -- the rest stays free for the back-half kernels of the previous call.
    if (TO == 64 && lin_wgs && (uint64_t)ntiles * n_streams >= 4ull * lin_wgs) {
        hipLaunchKernelGGL((k_decimate<D, T, TO>), dim3(lin_wgs), dim3(TO), 0, st, in, in_stride, hist_in, hist_out, taps, out, out_stride, call,
                           stage, final_stage, fir_hist_cap, 0u, fft_in, n_streams, ntiles, call_copy, final_stage ? 0u : uniform_n,
                           (final_stage || !uniform_n) ? StepClaim{} : claim);
        return;
    }
    // Walk several tiles per workgroup (prefetch pipelining) once there are enough workgroups to fill the chip:
    // 256 CUs x ~4 resident workgroups; keep >= ~2048 workgroups when the batch allows it.
    uint32_t per = 1;
    while (per < 16 && (uint64_t)((ntiles + 2 * per - 1) / (2 * per)) * n_streams >= 2048) per *= 2;
    dim3 grid((ntiles + per - 1) / per, n_streams);
    hipLaunchKernelGGL((k_decimate<D, T, TO>), grid, dim3(TO), 0, st, in, in_stride, hist_in, hist_out, taps, out, out_stride, call,
                       stage, final_stage, fir_hist_cap, per, fft_in, n_streams, 0u, (StreamCall*)nullptr, 0u, StepClaim{});
}

bool launch_decimate(hipStream_t st, int ratio, int ntaps, uint32_t n_streams, uint32_t max_out, const float2* in, size_t in_stride,
                     const float2* hist_in, float2* hist_out, const float* taps, float2* out, size_t out_stride, const StreamCall* call,
                     int stage, int final_stage, uint32_t fir_hist_cap, float2* fft_in, uint32_t lin_wgs, StreamCall* call_copy,
                     uint32_t uniform_n, const StepClaim& claim)
{
    if (!max_out) return true;
#define HD_CASE(D, T, TO) \
    if (ratio == D && ntaps == T) { launch_one<D, T, TO>(st, n_streams, max_out, in, in_stride, hist_in, hist_out, taps, out, out_stride, call, stage, final_stage, fir_hist_cap, fft_in, lin_wgs, call_copy, uniform_n, claim); return true; }
    HD_CASE(2, 69, 256) HD_CASE(4, 139, 256) HD_CASE(8, 280, 256) HD_CASE(8, 54, 256)
    HD_CASE(16, 107, 128) HD_CASE(32, 212, 64) HD_CASE(32, 174, 64) HD_CASE(64, 348, 64)
#undef HD_CASE
    return false;
}

uint32_t step_lds_bytes(int ratio, int ntaps)
{
    auto sz = [](int f4) { return (uint32_t)(f4 > (int)(kStepLdsBytes / 16) ? f4 : (int)(kStepLdsBytes / 16)) * 16u; };
    if (ratio == 32 && ntaps == 212) return sz(dec_tile_f4<32, 212, 64>());
    if (ratio == 32 && ntaps == 174) return sz(dec_tile_f4<32, 174, 64>());
    if (ratio == 64 && ntaps == 348) return sz(dec_tile_f4<64, 348, 64>());
    return 0;
}

bool launch_step(hipStream_t st, int ratio, int ntaps, int ratio2, int ntaps2, uint32_t n_streams, uint32_t n_out, const float2* in, size_t in_stride,
                 const float2* hist_in, float2* hist_out, const float* taps, float2* out, size_t out_stride, const StreamCall* call,
                 StreamCall* call_copy, uint32_t stage1_wgs, const TailArgs& ta, uint32_t n_tail, uint32_t uniform_n, const StepClaim& claim)
{
#define HD_STEP_CASE(D, T, D2, T2)                                                                                                    \
    if (ratio == D && ntaps == T && ratio2 == D2 && ntaps2 == T2) {                                                                   \
        const uint32_t ntiles = (n_out + 63) / 64;                                                                                    \
        hipLaunchKernelGGL((k_step<D, T, D2, T2>), dim3(n_tail + stage1_wgs), dim3(64), 0, st, in, in_stride, hist_in, hist_out, taps, out, \
                           out_stride, call, n_streams, ntiles, call_copy, ta, n_tail, uniform_n, claim);                             \
        return true;                                                                                                                  \
    }
    HD_STEP_CASE(32, 212, 2, 69) HD_STEP_CASE(32, 174, 4, 139) HD_STEP_CASE(64, 348, 4, 139)
#undef HD_STEP_CASE
    return false;
}

// Tiles per stream and call of the per-CU ring kernels: a /32 stage runs the systolic tap loop, whose 64-row tiles advance by 64 - HR rows
// (stage1_ring.h: ring_adv); the smaller ratios keep tiles of 64 rows + halo, 2048 input samples each.
uint32_t ring_tiles(int ratio, int ntaps, uint32_t n)
{
    if (ratio != 32 && ratio != 64) return n / 2048u;
    const uint32_t adv = 64u - (uint32_t)((ntaps - 1 + ratio - 1) / ratio), rows = n / (uint32_t)ratio;
    return rows >= 64u ? (rows + adv - 1u) / adv : 0u;
}

// what a publication word of the ring protocol can name (stage1_ring.h, RingCtl::pub): 24 bits of sequence number, 20 of stream, 12 of tile
static bool ring_limits_ok(uint32_t ntiles, const StepClaim& claim, bool workers = false /* worker waves: a run may run on into the next stream */)
{
    const uint64_t total = claim.tiles_per_xcd ? (uint64_t)claim.tiles_per_xcd * claim.n_xcd : (uint64_t)claim.runs_per_xcd * claim.n_xcd * claim.run_len;
    if (workers) return ntiles && claim.run_len >= 2u && total % ntiles == 0 && total < (1ull << 32) &&
                        (!claim.tiles_per_xcd || (uint64_t)claim.tiles_per_xcd == (uint64_t)(claim.short_from < claim.runs_per_xcd ? claim.short_from : claim.runs_per_xcd) * claim.run_len +
                                                                                    (claim.runs_per_xcd - (claim.short_from < claim.runs_per_xcd ? claim.short_from : claim.runs_per_xcd)));
    return ntiles && ntiles <= 4096u && claim.run_len && ntiles % claim.run_len == 0 && total < (1ull << 24) && total / ntiles <= (1ull << 20);
}

uint32_t step_cu_tail_lds(int ratio, int ntaps)
{
    if (ratio != 32 || (ntaps != 212 && ntaps != 174)) return 0;
    return ((163840u - (uint32_t)kRingCtlBytes - 4u * (uint32_t)kWorkSlotBytes) / 4u) & ~15u;      // four worker slots + four tail slices fill the CU's 160 KB
}

bool launch_step_cu(hipStream_t st, int ratio, int ntaps, int ratio2, int ntaps2, uint32_t n_streams, uint32_t n_cus, const float2* in, size_t in_stride,
                    const float2* hist_in, float2* hist_out, const float* taps, float2* out, size_t out_stride, const StreamCall* call,
                    StreamCall* call_copy, const TailArgs& ta, uint32_t n_tail, uint32_t uniform_n, const StepClaim& claim, uint32_t tail_bytes,
                    hipEvent_t ev_start, hipEvent_t ev_stop)
{
    if (ratio != 32 || !claim.ctr || !uniform_n || uniform_n % 2048u) return false;
    const uint32_t ntiles = ring_tiles(ratio, ntaps, uniform_n);
    if (!ring_limits_ok(ntiles, claim, true)) return false;
    if (tail_bytes < (uint32_t)kWorkSlotBytes) tail_bytes = (uint32_t)kWorkSlotBytes;       // (a finished tail's slice is its wave's tile slot)
    tail_bytes = (tail_bytes + 15u) & ~15u;
    RingArgs ra{in, in_stride, hist_in, hist_out, taps, out, out_stride, uniform_n, ntiles, claim, nullptr, nullptr, 0u, nullptr};
#define HD_CU_CASE(T, D2, T2)                                                                                                         \
    if (ntaps == T && ratio2 == D2 && ntaps2 == T2) {                                                                                 \
        const uint32_t lds = (uint32_t)kRingCtlBytes + (n_tail ? 4u * (uint32_t)kWorkSlotBytes + 4u * tail_bytes : 8u * (uint32_t)kWorkSlotBytes); \
        if (lds > 163840u) return false;                                                                                              \
        static bool attr_set[64] = {};                             /* per device: the attribute belongs to the function on the current device */ \
        int dev_ = 0;                                                                                                                 \
        if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) return false;                                                \
        if (!attr_set[dev_]) {                                                                                                        \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_step_cu<T, D2, T2>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840) != hipSuccess) return false; \
            attr_set[dev_] = true;                                                                                                    \
        }                                                                                                                             \
        const dim3 grid_(n_cus > (n_streams + 3u) / 4u ? n_cus : (n_streams + 3u) / 4u);                                             \
        /* events handed to the launch ride on the dispatch packet's own completion signal: no barrier packet behind the kernel */     \
        if (ev_start || ev_stop) hipExtLaunchKernelGGL((k_step_cu<T, D2, T2>), grid_, dim3(512), lds, st, ev_start, ev_stop, 0u, ra, call, call_copy, ta, n_tail, n_streams, tail_bytes); \
        else hipLaunchKernelGGL((k_step_cu<T, D2, T2>), grid_, dim3(512), lds, st, ra, call, call_copy, ta, n_tail, n_streams, tail_bytes); \
        return true;                                                                                                                  \
    }
    HD_CU_CASE(212, 2, 69) HD_CU_CASE(174, 4, 139)
#undef HD_CU_CASE
    return false;
}

bool step_cu_supported(int ratio, int ntaps, int ratio2, int ntaps2)
{
    return ratio == 32 && ((ntaps == 212 && ratio2 == 2 && ntaps2 == 69) || (ntaps == 174 && ratio2 == 4 && ntaps2 == 139));   // (HD_CU_CASE in launch_step_cu)
}

bool stage1_cu_supported(int ratio, int ntaps)
{
    return (ratio == 32 && (ntaps == 212 || ntaps == 174)) || (ratio == 64 && ntaps == 348) || (ratio == 8 && ntaps == 54) || (ratio == 4 && ntaps == 139);   // (/4: as a final stage)
}

bool launch_stage1_cu(hipStream_t st, int ratio, int ntaps, uint32_t n_cus, const float2* in, size_t in_stride, const float2* hist_in, float2* hist_out,
                      const float* taps, float2* out, size_t out_stride, uint32_t uniform_n, const StepClaim& claim, unsigned int* gave_up, uint32_t n_loaders, uint32_t n_waves,
                      uint32_t n_slots, const StreamCall* final_call, uint32_t fir_hist_cap, float2* fft_in)
{
    if (n_loaders != 1u) n_loaders = 2u;
    if (ratio <= 4) {            // 278 flop per input sample (/4): the vector pipes bind, not HBM -- one loader is plenty, and every other wave slot computes
        n_loaders = 1u;
        n_waves = 16u;
    }
    if (n_slots < 2u * n_loaders || n_slots > 8u) n_slots = 8u;
    if (n_waves < 8u || n_waves > 16u) n_waves = 8u;
    if (!claim.ctr || !uniform_n || uniform_n % 2048u) return false;
    const uint32_t ntiles = ring_tiles(ratio, ntaps, uniform_n);
    if (!ring_limits_ok(ntiles, claim, ratio >= 32)) return false;
    if ((ratio <= 4) != (final_call != nullptr)) return false;               // /4 exists as a FINAL stage only (the only stage of a plan), the others as first stages only
    RingArgs ra{in, in_stride, hist_in, hist_out, taps, out, out_stride, uniform_n, ntiles, claim, gave_up, final_call, fir_hist_cap, fft_in};
#define HD_S1_CASE(D, T)                                                                                                              \
    if (ratio == D && ntaps == T) {                                                                                                   \
        static_assert(D >= 32 || (uint32_t)ring_bytes<T, kRingNSLAlone>() <= 163840u, "eight tile slots must fit a CU's LDS");       \
        const uint32_t workers = D == 32 ? 8u : n_slots <= 4u ? 3u : 4u;            /* (/64: 33 KB slots; three leave room for the other queue's FIR tile) */ \
        const uint32_t lds = D >= 32 ? workers * (uint32_t)work_slot_bytes<(D >= 32 ? D : 32)>() : n_slots * (uint32_t)ring_slot_bytes<T>() + (uint32_t)kRingCtlBytes; \
        static bool attr_set[64] = {};                                                                                                \
        int dev_ = 0;                                                                                                                 \
        if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) return false;                                                \
        if (!attr_set[dev_]) {                                                                                                        \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_stage1_cu<T, D>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840) != hipSuccess) return false; \
            attr_set[dev_] = true;                                                                                                    \
        }                                                                                                                             \
        hipLaunchKernelGGL((k_stage1_cu<T, D>), dim3(n_cus), dim3(64u * (D == 4 ? n_waves : D >= 32 ? workers : 8u)), lds, st, ra, n_loaders, n_slots);   \
        return true;                                                                                                                  \
    }
    HD_S1_CASE(32, 212) HD_S1_CASE(32, 174) HD_S1_CASE(64, 348) HD_S1_CASE(8, 54) HD_S1_CASE(4, 139)
#undef HD_S1_CASE
    return false;
}

uint32_t probe_xcc_mask(hipStream_t st, uint32_t n_cus, unsigned int* d_word)
{
    if (hipMemsetAsync(d_word, 0, 4, st) != hipSuccess) return 0;
    hipLaunchKernelGGL(k_xcc_probe, dim3(n_cus * 8u), dim3(64), 0, st, d_word);
    unsigned int h = 0;
    if (hipMemcpyAsync(&h, d_word, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return 0;
    return h;
}

void launch_passthrough(hipStream_t st, uint32_t n_streams, uint32_t max_n, const float2* in, size_t in_stride, float2* out,
                        size_t out_stride, const StreamCall* call, uint32_t fir_hist_cap)
{
    if (!max_n) return;
    dim3 grid((max_n + 255) / 256, n_streams);
    hipLaunchKernelGGL(k_passthrough, grid, dim3(256), 0, st, in, in_stride, out, out_stride, call, fir_hist_cap);
}

void launch_dc_remove(hipStream_t st, uint32_t n_streams, float2* fbuf, size_t stride, const StreamCall* call, uint32_t fir_hist_cap)
{
    hipLaunchKernelGGL(k_dc_remove, dim3(n_streams), dim3(64), 0, st, fbuf, stride, call, fir_hist_cap);
}

}  // namespace HD_ARITH_NS
}  // namespace hd
