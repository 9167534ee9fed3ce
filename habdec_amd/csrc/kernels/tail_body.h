// The stream tail: everything behind the first decimation stage of a two-stage plan, for ONE stream and one call, as a
// device function -- second FIR-decimate stage, low-pass FIR, FSK discriminator, symbol extractor (window sums, flag
// mask, edge search, run means, bits), plus the bookkeeping the other kernels expect (history carries, buffer slide,
// spectrum feed, FirHistory head, result slot).
//
// Why it exists.  k_backend + k_symbols (backend.hip, symbols.hip) give a stream a 256-lane workgroup each and image the
// whole call in LDS.  Next to the HBM-bound stage-1 kernel only one such workgroup fits per CU, so in batch mode the back
// half of call k took about as long as stage 1 of call k+1 and the step was the SUM of the two.  This body needs one
// wave, ~17 KB of LDS and no barrier: the call is walked in pieces of P stage-2 outputs whose inputs are prefetched into
// registers one piece ahead, every stage hands its output to the next through small sliding LDS windows, and the symbol
// extractor consumes the discriminator output as it appears.  It runs either as its own kernel (k_tail, tail.hip) or as
// the first blockIdx range of the stage-1 launch of the NEXT call (k_step, decimate.hip), where the hardware dispatcher
// deals tails and stage-1 tile runs to the CUs' eight slots as they free up.
//
// Arithmetic and semantics are those of the kernels it replaces (and therefore of the reference): T-term sums in
// ascending tap order with separately rounded multiply and add (Decimator.h:128-138, FirFilter.h:155-161; compiled with
// -ffp-contract=off), arg(y[i] * conj(y[i-1])) through exact_math.h (FSK2_Demod.h:37-40), window sums and run sums
// added left to right by one lane (SymbolExtractor.h:162-224).  Global state (buffers, rings, SymState) is laid out as
// the other kernels leave and expect it, so the engine can switch between the paths from call to call.
#pragma once
#include <hip/hip_runtime.h>

#include "arith.h"
#include "exact_math.h"
#include "launch.h"
#include "sym_common.h"
#include "spectrum_wave.h"

namespace hd {
namespace HD_ARITH_NS {

constexpr uint32_t kTailHdrBytes = 64;          // small shared scalars at the base of the workgroup's LDS scratch
constexpr uint32_t kTailStrip = 512;            // samples per run-sum step (per wave)

typedef float f32x4 __attribute__((ext_vector_type(4)));
// Tap tables are read-only for the whole launch, but they hang off a by-value argument struct, so the compiler cannot prove that the
// kernel's own stores leave them alone and would fetch them with VECTOR loads inside the loops (a vmcnt(0) wait per block of taps).
// Reading them through the constant address space makes them scalar loads (wave-uniform address, scalar cache).
typedef const float __attribute__((address_space(4)))* cfloat_ptr;
__device__ __forceinline__ cfloat_ptr as_const(const float* p) { return (cfloat_ptr)(uintptr_t)p; }

// One tap for a lane's four adjacent outputs: four independent products, then four independent adds.  Every accumulator still
// receives its products in ascending tap order with separately rounded multiply and add; keeping the four chains apart is what lets
// ONE wave issue back to back (a single chain waits out the add latency on every tap).
#define HD_TB_TAP4(A_, B_, C_, D_, k_)                                                        \
    {                                                                                         \
        HD_FIR_ARITH                                                                          \
        const f32x2 p0_ = (A_) * (k_), p1_ = (B_) * (k_), p2_ = (C_) * (k_), p3_ = (D_) * (k_); \
        a0 = a0 + p0_; a1 = a1 + p1_; a2 = a2 + p2_; a3 = a3 + p3_;                           \
    }

#ifdef HD_RING_FAULT   // fault-injection build (libhabdec_amd_fault.so): ONE stream tail of the process leaves its result slot without the call's tag
__device__ unsigned int g_tail_fault_armed, g_tail_fault_fired;
#endif
#ifdef HD_STAMP_TAIL   // diagnostic build only (tools/micro/tail_stamps.py): cycles per phase of the tail, per stream
__device__ unsigned long long g_tail_stamps[8192 * 24];
#define TSTAMP_DECL unsigned long long ts_t = __builtin_amdgcn_s_memtime(), ts_acc[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long ts_r0 = __builtin_amdgcn_s_memrealtime()
#define TSTAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ts_acc[i] += t_ - ts_t; ts_t = t_; } while (0)
#define TSTAMP_WRITE() do { if (tid == 0 && s < 8192) { unsigned long long* g_ = g_tail_stamps + (size_t)s * 24; for (int i_ = 0; i_ < 20; ++i_) g_[i_] = ts_acc[i_]; g_[20] = ts_r0; g_[21] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define TSTAMP_DECL do { } while (0)
#define TSTAMP(i) do { } while (0)
#define TSTAMP_WRITE() do { } while (0)
#endif

#define HD_TB_TAPN(A_, B_, C_, D_, k_, a0_, a1_, a2_, a3_)                                    \
    {                                                                                         \
        HD_FIR_ARITH                                                                          \
        const f32x2 p0_ = (A_) * (k_), p1_ = (B_) * (k_), p2_ = (C_) * (k_), p3_ = (D_) * (k_); \
        a0_ = a0_ + p0_; a1_ = a1_ + p1_; a2_ = a2_ + p2_; a3_ = a3_ + p3_;                   \
    }
// Four taps K_ = (k, k+1, k+2, k+3) for four adjacent outputs whose inputs start in pair W0_: output q takes input q + u for tap u.
#define HD_TB_BLK(W0_, W1_, W2_, W3_, K_, a0_, a1_, a2_, a3_)                      \
    HD_TB_TAPN((W0_).xy, (W0_).zw, (W1_).xy, (W1_).zw, (K_).x, a0_, a1_, a2_, a3_) \
    HD_TB_TAPN((W0_).zw, (W1_).xy, (W1_).zw, (W2_).xy, (K_).y, a0_, a1_, a2_, a3_) \
    HD_TB_TAPN((W1_).xy, (W1_).zw, (W2_).xy, (W2_).zw, (K_).z, a0_, a1_, a2_, a3_) \
    HD_TB_TAPN((W1_).zw, (W2_).xy, (W2_).zw, (W3_).xy, (K_).w, a0_, a1_, a2_, a3_)

// One tap (position U of a block of four) for a lane's OP adjacent stage-2 outputs: sample D2*q + U of the rolling window for output q.
template <int U, int OP, int D2, int NWIN>
__device__ __forceinline__ void tb_s2_tap(f32x2 (&acc)[OP], const f32x4 (&win)[NWIN], const float k)
{
    HD_FIR_ARITH
    f32x2 pr[OP];
#pragma unroll
    for (int q = 0; q < OP; ++q) {
        const f32x4 w = win[(D2 * q + U) >> 1];
        if constexpr (U & 1) pr[q] = w.zw * k; else pr[q] = w.xy * k;
    }
#pragma unroll
    for (int q = 0; q < OP; ++q) acc[q] = acc[q] + pr[q];
}

template <int NT>
__device__ __forceinline__ void tb_sync()
{
    if constexpr (NT == 64) {
        // One wave per stream: what __syncthreads() is for a 64-lane workgroup once the compiler has dropped the s_barrier -- the
        // memory ordering alone.  Spelled out because the same body also runs as ONE WAVE OF A LARGER workgroup (k_step_cu,
        // decimate.hip), whose other waves must not be waited for.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    } else {
        __syncthreads();
    }
}

// NT lanes, OP stage-2 outputs per lane and piece, stage-2 design (D2, T2).
template <int NT, int OP, int D2, int T2>
__device__ __forceinline__ void tail_body(const TailArgs& a, const uint32_t s, unsigned char* __restrict__ lds)
{
    constexpr int P = NT * OP;                      // stage-2 outputs per piece
    constexpr int XCH = P * D2;                     // stage-1 samples a piece consumes
    constexpr int XN = (T2 - 1) + XCH;              // X image: [stage-2 history | piece]
    constexpr int XB = XCH / 2 / NT;                // 16-byte loads per lane and piece
    constexpr int HB = (T2 - 1 + NT - 1) / NT;
    constexpr int NW = NT / 64;
    const uint32_t B = 2 * P + a.pend_max;          // most discriminator outputs one low-pass round (two pieces) can release
    static_assert(((T2 - 1) & 1) == 0 && XCH % (2 * NT) == 0, "16-byte pairs must stay aligned");
    const uint32_t tid = NT == 64 ? (threadIdx.x & 63u) : threadIdx.x, lane = tid & 63u, wave = tid >> 6;   // (64 lanes: possibly one wave of a larger workgroup)

    uint32_t* sh = reinterpret_cast<uint32_t*>(lds);                    // [0] nfl [1] overflow [2] frontier [3] carry sum [4] flagged
    float2* X = reinterpret_cast<float2*>(lds + kTailHdrBytes);
    // Y[0]: predecessor of this pass's first output; Y[1 + l], Y[1 + NT + l]: lane l's last outputs.  One wave per stream (kCompact): Y sits INSIDE the
    // stage-1 image, behind the stage-2 history -- that part of X is dead from the end of a piece's stage 2 until the next piece is stored, which is
    // when the low-pass passes run; the carry Y[0] outlives a piece and is kept in the header (sh[8..9]) between rounds.
    constexpr bool kCompact = NT == 64;
    static_assert(!kCompact || 2 * NT + 2 <= XCH, "the exchange array must fit the dead part of the stage-1 image");
    // The PADDED stage-1 image (one wave, four outputs per lane, /2): a lane's stage-2 window starts OP * D2 = 8 samples = 64 bytes after its neighbour's, and
    // ds_read_b128 serves sixteen lanes per cycle from sixteen 16-byte bank groups -- at a 64-byte stride every fourth lane of a group hits the same one (a
    // four-way conflict on every read of the tap loop).  16 bytes of padding behind every 8 samples make the stride 80 bytes, an odd number of bank groups:
    // conflict-free (sample j lives at xpos(j); the lane's k-th 16-byte chunk at chunk k + k / 4 behind its first).  tail_layout (tail.hip) sizes the image.
#ifdef HD_X_NOPAD
    constexpr bool kXPad = false;
#else
    constexpr bool kXPad = kCompact && OP == 4 && D2 == 2;
#endif
    auto xpos = [](uint32_t j) -> uint32_t { return kXPad ? j + 2u * (j >> 3) : j; };
    // (positions this lane stores to, once per call: sample tid + NT u sits NT + NT / 4 slots behind sample tid + NT (u - 1), and so on -- NT is a multiple of 8)
    constexpr uint32_t kXStepH = kXPad ? NT + 2 * (NT >> 3) : NT, kXStepP = kXPad ? 2 * NT + 2 * (2 * NT >> 3) : 2 * NT, kXPiece = kXPad ? XCH + 2 * (XCH >> 3) : XCH;
    static_assert(NT % 8 == 0 && XCH % 8 == 0, "whole pad groups per step");
    const uint32_t xh0 = xpos(tid), xp0 = xpos((uint32_t)(T2 - 1) + 2u * tid);
    constexpr int YOFF = kXPad ? (((T2 - 2) + 2 * ((T2 - 2) >> 3) + 1 + 1) & ~1) : ((T2 - 1 + 1) & ~1);    // behind the (padded) stage-2 history
    static_assert(!kXPad || YOFF + 2 * NT + 2 <= XN, "the exchange array must fit the dead part of the padded stage-1 image");
    float2* Y = kCompact ? X + YOFF : X + ((XN + 4 + 1) & ~1);
    float2* carryY = reinterpret_cast<float2*>(sh + 8);
    float2* F = reinterpret_cast<float2*>(lds + a.f_off);               // low-pass input window, F[0] = input index fbase
    float* V = reinterpret_cast<float*>(lds + a.v_off);                 // discriminator output from position c0 on
    float* WS = reinterpret_cast<float*>(lds + a.ws_off);               // window sums of [c0 - R, c0), then the new ones
    unsigned long long* words = reinterpret_cast<unsigned long long*>(lds + a.words_off);
    float* TP = reinterpret_cast<float*>(lds + a.tp_off);              // this stream's low-pass taps
    float* H2 = reinterpret_cast<float*>(lds + a.h2_off);              // stage-2 taps

    TSTAMP_DECL;
    const StreamCall c = a.call[s];
    const uint32_t n1 = c.n1, n2 = c.n2, m = c.fir_m, T = c.fir_taps, pb = c.pend_before;
    const uint32_t H = T ? T - 1 : 0;
    const uint32_t Tp = sc_taps_prev(c) ? sc_taps_prev(c) : T;
    const bool run = m && T;
    const bool keepf = !c.clear_pending;
    const float2* in_s = a.dec1 + (size_t)s * a.dec1_stride;
    float2* cur = a.fbuf + (size_t)s * a.fbuf_stride;
    float2* nxt = a.fbuf_next + (size_t)s * a.fbuf_stride;
    // FirHistory head (dev_types.h): in the previous call's low-pass buffer, or in the side buffer a non-running call copied it to
    float2* head_side = a.head_buf + (size_t)s * a.head_cap;
    const float2* head_prevbuf = a.fbuf_prev + (size_t)s * a.fbuf_stride + a.fir_hist_cap;
    const float2* head_in = sc_head_prev(c) ? head_prevbuf : head_side;
    const float* tp = a.lp_taps + (size_t)s * a.taps_stride;
    const uint32_t fhc = a.fir_hist_cap;


    // ---- symbol extractor: state after this call's push (SymbolExtractor.h:116-124), what this call has to do
    const uint32_t ring_cap = a.ring_cap, rmask = ring_cap - 1;
    float* vring = a.ring + (size_t)s * ring_cap;
    float* gw = a.wsum + (size_t)s * ring_cap;
    unsigned long long* gmask = a.flipmask + (size_t)s * (ring_cap / 64);
    uint32_t* slot = a.slots + (size_t)s * a.slot_words;
    BitsHeader* hdr = reinterpret_cast<BitsHeader*>(slot);
    const SymState old = a.sym[s];
    const SymbolParams q = a.sp[s];
    const DemodCarry kin = a.carry_in[s];
    const uint32_t R = q.R;
    SymState st = old;
    bool search = false, do_sums = false;
    if (m) {
        st = state_after_push(old, q, m);
        search = !(st.held < q.min_held || st.held < q.spb);
        do_sums = search || !(q.min_held == 0xFFFFFFFFu || st.held < R);
    }
    const uint32_t end_old = old.base + old.held;                       // ring position of this call's first discriminator output
    uint32_t c0 = st.cached;                                            // next position whose window sum is due; V[0] is sample c0
    uint32_t vcnt = 0;                                                  // samples in V
    uint32_t ck0 = 0, ck1 = 0;                                          // this lane's share of the discriminator checksum (BitsHeader::demod_ck)
    unsigned long long carry_word = 0ull;                               // flags of the positions [c0 & ~63, c0), as earlier sweeps left them
    const uint32_t old_cnt = do_sums ? end_old - c0 : 0u;              // backlog samples whose windows are not final yet (R-1 in steady state)

    // ---- every global read the first piece needs, issued back to back (one round trip): a touch of each 64-byte line of the
    // stream's low-pass taps (scalar cache), stage-2 history, the first piece of the stage-1 chunk, low-pass history + pending,
    // the backlog samples and window sums the first window sums build on, the boundary word of the flag mask.
    const bool refold = Tp != T && !c.fir_zero_hist && run;           // first run after a tap-count change (FirHistory, dev_types.h)
    const uint32_t head_n = refold ? sc_head_n(c) : 0u;
    auto fold = [&](uint32_t k) -> float2 {                             // element k of [history (H) | pending]
        if (k < H && c.fir_zero_hist) return make_float2(0.f, 0.f);
        if (!refold || k >= H) return cur[fhc - H + k];
        if (k < Tp - 1) return cur[fhc - (Tp - 1) + k];
        const uint32_t h = k - (Tp - 1);
        return h < head_n ? head_in[h] : make_float2(0.f, 0.f);
    };
    const uint32_t f_old = H + pb;
    const float4* in4 = reinterpret_cast<const float4*>(in_s);
    const uint32_t n1p = n1 >> 1;
    float4 tx[XB];                                                      // register prefetch of the next piece
    auto prefetch = [&](uint32_t pc) {
#pragma unroll
        for (int u = 0; u < XB; ++u) {
            const uint32_t k = pc * (XCH / 2) + tid + u * NT;
            tx[u] = k < n1p ? in4[k] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    constexpr uint32_t kFirst = 256;                                    // backlog samples taken with the first batch (R-1 of them are due in steady state)
    const uint32_t take0 = min(old_cnt, kFirst);
    {
        // Tap tables go to LDS: inside the loops every operand then comes through ds_read (in-order, counted waits), so the next
        // block's reads stay in flight under the math; scalar loads there would force a full lgkmcnt(0) stall per block.
        float2 th[HB];
#pragma unroll
        for (int u = 0; u < HB; ++u) {
            const uint32_t k = tid + u * NT;
            th[u] = (k < (uint32_t)(T2 - 1) && !c.zero_hist2) ? a.hist2_in[(size_t)s * (T2 - 1) + k] : make_float2(0.f, 0.f);
        }
        prefetch(0);
        if (do_sums && (c0 & 63u)) carry_word = gmask[(c0 & rmask) >> 6] & ((1ull << (c0 & 63u)) - 1ull);
        constexpr int LB = 4;                                           // loads per lane in flight per batch
        const bool special = c.fir_zero_hist || refold;               // history restarts from zeros / first run after a redesign
        // The usual shape (a 161-tap low-pass, less than a batch pending, windows of a few hundred samples): EVERY load of the start-up goes out before the
        // first of them is waited for -- one round trip.  Inside a step launch a global-memory instruction of this wave queues behind the stage-1 waves' tile
        // loads (in-kernel clocks, round 5: the seven batches below, each waiting for its own loads before it stores them, took 27k of a tail's 200k cycles).
        constexpr int NB2 = ((T2 + 7) / 4 * 4 + NT - 1) / NT;           // batches of NT for the stage-2 tap table
        if (T + 8 <= 4u * NT && f_old <= 8u * NT && R <= 4u * NT && !special) {
            float tt[4], t2[NB2], tv[4], tw[4];
            float2 tf[8];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const uint32_t k = tid + u * NT; tt[u] = k < T ? tp[k] : 0.f; }
#pragma unroll
            for (int u = 0; u < NB2; ++u) { const uint32_t k = tid + u * NT; t2[u] = k < (uint32_t)T2 ? a.taps2[k] : 0.f; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const uint32_t k = tid + u * NT; tf[u] = k < f_old ? cur[fhc - H + k] : make_float2(0.f, 0.f); }
            if (do_sums) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { const uint32_t k = tid + u * NT; tv[u] = k < take0 ? vring[(c0 + k) & rmask] : 0.f; }
#pragma unroll
                for (int u = 0; u < 4; ++u) { const uint32_t k = tid + u * NT; tw[u] = k < R ? gw[(c0 - R + k) & rmask] : 0.f; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const uint32_t k = tid + u * NT; if (k < T + 8) TP[k] = tt[u]; }
#pragma unroll
            for (int u = 0; u < NB2; ++u) { const uint32_t k = tid + u * NT; if (k < (uint32_t)((T2 + 7) & ~3)) H2[k] = t2[u]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const uint32_t k = tid + u * NT; if (k < f_old) F[k] = tf[u]; }
            if (do_sums) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { const uint32_t k = tid + u * NT; if (k < take0) V[k] = tv[u]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) { const uint32_t k = tid + u * NT; if (k < R) WS[k] = tw[u]; }
            }
        } else {
#pragma unroll 1
        for (uint32_t k0 = 0; k0 < T + 8; k0 += LB * NT) {
            float t[LB];
#pragma unroll
            for (int u = 0; u < LB; ++u) { const uint32_t k = k0 + tid + u * NT; t[u] = k < T ? tp[k] : 0.f; }
#pragma unroll
            for (int u = 0; u < LB; ++u) { const uint32_t k = k0 + tid + u * NT; if (k < T + 8) TP[k] = t[u]; }
        }
#pragma unroll 1
        for (uint32_t k0 = 0; k0 < (uint32_t)((T2 + 7) & ~3); k0 += LB * NT) {
            float t[LB];
#pragma unroll
            for (int u = 0; u < LB; ++u) { const uint32_t k = k0 + tid + u * NT; t[u] = k < (uint32_t)T2 ? a.taps2[k] : 0.f; }
#pragma unroll
            for (int u = 0; u < LB; ++u) { const uint32_t k = k0 + tid + u * NT; if (k < (uint32_t)((T2 + 7) & ~3)) H2[k] = t[u]; }
        }
#pragma unroll 1
        for (uint32_t k0 = 0; k0 < f_old; k0 += LB * NT) {
            float2 t[LB];
#pragma unroll
            for (int u = 0; u < LB; ++u) { const uint32_t k = k0 + tid + u * NT; t[u] = k < f_old ? (special ? fold(k) : cur[fhc - H + k]) : make_float2(0.f, 0.f); }
#pragma unroll
            for (int u = 0; u < LB; ++u) { const uint32_t k = k0 + tid + u * NT; if (k < f_old) F[k] = t[u]; }
        }
        if (do_sums) {
#pragma unroll 1
            for (uint32_t k0 = 0; k0 < take0; k0 += LB * NT) {
                float t[LB];
#pragma unroll
                for (int u = 0; u < LB; ++u) { const uint32_t k = k0 + tid + u * NT; t[u] = k < take0 ? vring[(c0 + k) & rmask] : 0.f; }
#pragma unroll
                for (int u = 0; u < LB; ++u) { const uint32_t k = k0 + tid + u * NT; if (k < take0) V[k] = t[u]; }
            }
#pragma unroll 1
            for (uint32_t k0 = 0; k0 < R; k0 += LB * NT) {
                float t[LB];
#pragma unroll
                for (int u = 0; u < LB; ++u) { const uint32_t k = k0 + tid + u * NT; t[u] = k < R ? gw[(c0 - R + k) & rmask] : 0.f; }
#pragma unroll
                for (int u = 0; u < LB; ++u) { const uint32_t k = k0 + tid + u * NT; if (k < R) WS[k] = t[u]; }
            }
        }
        }
#pragma unroll
        for (int u = 0; u < HB; ++u) { const uint32_t k = tid + u * NT; if (k < (uint32_t)(T2 - 1)) X[xh0 + u * kXStepH] = th[u]; }
        if (tid == 0) { if (kCompact) carryY[0] = make_float2(kin.re, kin.im); else Y[0] = make_float2(kin.re, kin.im); }
    }
    vcnt = take0;
    tb_sync<NT>();
    TSTAMP(0);

    // ---- window sums + flags for every position whose right window is complete: positions [c0, c0 + vcnt - R + 1).
    // W(p) = v[p] + ... + v[p+R-1] is both the reference's right window of p and its left window of p + R (symbols.hip).
    auto sym_feed = [&]() {
        int32_t npos = (int32_t)vcnt - (int32_t)R + 1;
        uint32_t off = 0;
        constexpr uint32_t KP = kWidePos;                               // positions per lane
        while (npos > 0) {
            const uint32_t cnt = min((uint32_t)npos, (uint32_t)(KP * NT));
            const uint32_t wb0 = c0 & ~63u, wsh = c0 & 63u;
            const uint32_t nwords = (wsh + cnt + 63u) >> 6;
            if (tid < (uint32_t)(KP * NT / 64 + 2)) words[tid] = 0ull;
            tb_sync<NT>();
            const bool any = KP * tid < cnt;
            float wp[KP];
            if (any) {
                if constexpr (kFastArith && kFastWindows) window_sums_slide<(int)KP>(V + off + KP * tid, R, wp); else window_sums8(V + off + KP * tid, R, wp);
                TSTAMP(19);
#pragma unroll
                for (int j = 0; j < (int)KP; ++j) WS[off + R + KP * tid + j] = wp[j];
                float* gdst = gw + ((c0 + KP * tid) & rmask);           // c0 + KP*tid .. +7: contiguous unless the ring wraps inside
                if (KP * tid + KP <= cnt && ((c0 + KP * tid) & rmask) + KP <= ring_cap && (((c0 + KP * tid) & 3u) == 0)) {
                    *reinterpret_cast<float4*>(gdst) = make_float4(wp[0], wp[1], wp[2], wp[3]);
                    *reinterpret_cast<float4*>(gdst + 4) = make_float4(wp[4], wp[5], wp[6], wp[7]);
                } else {
#pragma unroll
                    for (int j = 0; j < (int)KP; ++j) if (KP * tid + j < cnt) gw[(c0 + KP * tid + j) & rmask] = wp[j];
                }
            }
            TSTAMP(15);
            tb_sync<NT>();
            if (any) {
                float wlv[KP];
                const float4* wl4 = reinterpret_cast<const float4*>(WS + off + KP * tid);    // (off is a multiple of the sweep, the window 16-byte aligned)
#pragma unroll
                for (int c4 = 0; c4 < (int)KP / 4; ++c4) { const float4 x = wl4[c4]; wlv[4 * c4] = x.x; wlv[4 * c4 + 1] = x.y; wlv[4 * c4 + 2] = x.z; wlv[4 * c4 + 3] = x.w; }
                const unsigned int bits = sign_flags<(int)KP>(wlv, wp, min(KP, cnt - KP * tid), R);
                if (bits) {
                    const uint32_t bp = wsh + tid * KP, shb = bp & 63u;
                    atomicOr(&words[bp >> 6], (unsigned long long)bits << shb);
                    if (shb > 64u - KP) atomicOr(&words[(bp >> 6) + 1], (unsigned long long)bits >> (64u - shb));
                }
            }
            tb_sync<NT>();
            if (tid < nwords) {
                unsigned long long w = words[tid];
                if (tid == 0) w |= carry_word;
                gmask[((wb0 + tid * 64u) & rmask) >> 6] = w;
            }
            {   // the word the next sweep starts in, as far as it is filled
                const uint32_t e = wsh + cnt, ei = e >> 6, eb = e & 63u;
                unsigned long long w = words[ei];
                if (ei == 0) w |= carry_word;
                carry_word = eb ? (w & ((1ull << eb) - 1ull)) : 0ull;
            }
            tb_sync<NT>();
            c0 += cnt; off += cnt; npos -= (int32_t)cnt;
            TSTAMP(16);
        }
        if (off) {                                                      // slide both windows down by the positions done
            const uint32_t nv = vcnt - off;
            for (uint32_t k0 = 0; k0 < nv; k0 += 4 * NT) {
                float t4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const uint32_t k = k0 + tid + u * NT; t4[u] = k < nv ? V[k + off] : 0.f; }
                tb_sync<NT>();
#pragma unroll
                for (int u = 0; u < 4; ++u) { const uint32_t k = k0 + tid + u * NT; if (k < nv) V[k] = t4[u]; }
            }
            for (uint32_t k0 = 0; k0 < R; k0 += 4 * NT) {
                float t4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const uint32_t k = k0 + tid + u * NT; t4[u] = k < R ? WS[k + off] : 0.f; }
                tb_sync<NT>();
#pragma unroll
                for (int u = 0; u < 4; ++u) { const uint32_t k = k0 + tid + u * NT; if (k < R) WS[k] = t4[u]; }
            }
            vcnt = nv;
            tb_sync<NT>();
            TSTAMP(17);
        }
    };
    TSTAMP(1);
    // ---- the call, piece by piece
    uint32_t fbase = 0, fcount = f_old, i_done = 0;
    const uint32_t npieces = (n2 + P - 1) / P;
    // One loop for both kinds of step -- first the backlog samples that are due beyond the first batch (only after a parameter
    // change: the whole backlog is due), then the pieces of the call -- so that the window-sum code exists once.
    uint32_t old_left = old_cnt - take0;
    for (uint32_t pc = 0;;) {
        if (do_sums) {
            tb_sync<NT>();
            sym_feed();
        }
        TSTAMP(6);
        if (old_left) {
            const uint32_t take = min(old_left, B);
            for (uint32_t k = tid; k < take; k += NT) V[vcnt + k] = vring[(c0 + vcnt + k) & rmask];
            vcnt += take; old_left -= take;
            continue;
        }
        if (pc == npieces) break;
        const uint32_t o0 = pc * P, po = min((uint32_t)P, n2 - o0);
#pragma unroll
        for (int u = 0; u < XB; ++u) {
            *reinterpret_cast<float4*>(X + xp0 + u * kXStepP) = tx[u];     // (sample T2 - 1 + 2 k; T2 - 1 is even: a pair never straddles a pad)
        }
        tb_sync<NT>();
        if (pc + 1 < npieces) prefetch(pc + 1);
        TSTAMP(2);

        // stage 2: y2[o] = sum_t x[o*D2 + t] * h2[t].  A lane owns OP ADJACENT outputs of the piece (o = OP*tid + q): their inputs
        // overlap, so a block of four taps costs two new 16-byte LDS reads (plus one for the taps) for 8*OP multiply-adds, and the OP
        // sums are independent chains that keep one wave issuing back to back.  Sample j of the block, relative to the lane's first
        // input, is D2*q + u for output q and tap u: pair (D2*q + u) / 2 of the rolling window, low or high half.
        {
            constexpr int NWIN = ((D2 * (OP - 1) + 3) >> 1) + 1;       // 16-byte pairs a block of four taps touches
            const f32x4* h2 = reinterpret_cast<const f32x4*>(H2);
            const f32x4* px = reinterpret_cast<const f32x4*>(X + xpos((uint32_t)(OP * tid) * D2));
            f32x4 kc = h2[0];
            f32x2 acc[OP];
            f32x4 win[NWIN];
#pragma unroll
            for (int q = 0; q < OP; ++q) acc[q] = (f32x2){0.f, 0.f};
            constexpr int NB4 = T2 / 4;
            auto block = [&](const f32x4 n0, const f32x4 n1, const f32x4 kn) {
                __builtin_amdgcn_sched_barrier(0);                      // keep those reads up here: one wave has nobody else to hide their latency
                tb_s2_tap<0, OP, D2, NWIN>(acc, win, kc.x);
                tb_s2_tap<1, OP, D2, NWIN>(acc, win, kc.y);
                tb_s2_tap<2, OP, D2, NWIN>(acc, win, kc.z);
                tb_s2_tap<3, OP, D2, NWIN>(acc, win, kc.w);
#pragma unroll
                for (int j = 0; j + 2 < NWIN; ++j) win[j] = win[j + 2];
                win[NWIN - 2] = n0; win[NWIN - 1] = n1;
                kc = kn;
            };
            if constexpr (kXPad) {
                // chunk k of the lane's window sits at px[k + k / 4]; a block of four taps takes chunks 2 b + NWIN and 2 b + NWIN + 1 (NWIN = 5), so two blocks
                // advance the chunk pointer by exactly five: immediate offsets inside the pair, one pointer bump per pair
                static_assert(NWIN == 5, "the padded image's offsets below are those of four outputs per lane at /2");
#pragma unroll
                for (int j = 0; j < NWIN; ++j) win[j] = px[j + (j >> 2)];
                const f32x4* pq = px;
                int b = 0;
#pragma unroll 1
                for (; b + 2 <= NB4; b += 2, pq += 5) {
                    { const f32x4 n0 = pq[6], n1 = pq[7]; const f32x4 kn = h2[b + 1]; block(n0, n1, kn); }
                    { const f32x4 n0 = pq[8], n1 = pq[10]; const f32x4 kn = h2[b + 2]; block(n0, n1, kn); }
                }
                if (b < NB4) { const f32x4 n0 = pq[6], n1 = pq[7]; const f32x4 kn = h2[b + 1]; block(n0, n1, kn); }   // (reads a few slots past its taps: in bounds, unused)
            } else {
#pragma unroll
                for (int j = 0; j < NWIN; ++j) win[j] = px[j];
#pragma unroll 2
                for (int b = 0; b < NB4; ++b) {
                    const f32x4 n0 = px[2 * b + NWIN], n1 = px[2 * b + NWIN + 1];   // (the last block reads a few slots past its taps: in bounds, unused)
                    const f32x4 kn = h2[b + 1];
                    block(n0, n1, kn);
                }
            }
            if constexpr (T2 % 4 >= 1) tb_s2_tap<0, OP, D2, NWIN>(acc, win, kc.x);   // the T2 % 4 taps left over
            if constexpr (T2 % 4 >= 2) tb_s2_tap<1, OP, D2, NWIN>(acc, win, kc.y);
            if constexpr (T2 % 4 >= 3) tb_s2_tap<2, OP, D2, NWIN>(acc, win, kc.z);
            TSTAMP(18);
            const bool keepF = keepf || npieces == 1;
            bool done4 = false;
            if constexpr (OP == 4) {
                // four adjacent outputs per lane: 16-byte stores whenever all four exist and the destinations are 16-byte aligned
                // (a global store instruction costs the lone wave a couple of hundred cycles: four times fewer of them)
                const uint32_t oo = o0 + 4 * tid;
                if (4 * tid + 3 < po && !(pb & 1u)) {
                    const float4 y01 = make_float4(acc[0].x, acc[0].y, acc[1].x, acc[1].y), y23 = make_float4(acc[2].x, acc[2].y, acc[3].x, acc[3].y);
                    if (keepF) { float4* f4 = reinterpret_cast<float4*>(F + (f_old + oo - fbase)); f4[0] = y01; f4[1] = y23; }
                    { float4* c4 = reinterpret_cast<float4*>(cur + fhc + pb + oo); c4[0] = y01; c4[1] = y23; }   // the decimated chunk stays readable (getters, the unfused path next call)
                    if (a.fft_in && oo < c.fft_take) {                  // Decoder.h:467-473
                        float2* fi = a.fft_in + (size_t)s * kFftBins + c.fft_fill + oo;
                        if (oo + 3 < c.fft_take && !(c.fft_fill & 1u)) { reinterpret_cast<float4*>(fi)[0] = y01; reinterpret_cast<float4*>(fi)[1] = y23; }
                        else {
#pragma unroll
                            for (int q = 0; q < 4; ++q) if (oo + q < c.fft_take) fi[q] = make_float2(acc[q].x, acc[q].y);
                        }
                    }
                    done4 = true;
                }
            }
            if (!done4) {
#pragma unroll
                for (int q = 0; q < OP; ++q) {
                    const uint32_t ol = OP * tid + q;
                    if (ol < po) {
                        const uint32_t oo = o0 + ol;
                        const float2 y = make_float2(acc[q].x, acc[q].y);
                        if (keepF) F[f_old + oo - fbase] = y;
                        cur[fhc + pb + oo] = y;
                        if (a.fft_in && oo < c.fft_take) a.fft_in[(size_t)s * kFftBins + c.fft_fill + oo] = y;
                    }
                }
            }
        }
        // the last T2-1 samples of this image are the next piece's history
        float2 xt[HB];
#pragma unroll
        for (int u = 0; u < HB; ++u) { const uint32_t k = tid + u * NT; xt[u] = k < (uint32_t)(T2 - 1) ? X[xh0 + kXPiece + u * kXStepH] : make_float2(0.f, 0.f); }
        if (pc + 1 == npieces && n1) {
            // stage-2 history carry for the next call (Decimator.h:140-143, with the in-place quirk Q4 of Decoder.h:443-444: history
            // positions inside the first n2 samples hold OUTPUTS -- only for inputs so short that this is the only piece)
            tb_sync<NT>();
            for (uint32_t k = tid; k < (uint32_t)(T2 - 1); k += NT) {
                const uint32_t idx = n1 - (T2 - 1) + k;                 // host guarantees n1 >= T2-1
                a.hist2_out[(size_t)s * (T2 - 1) + k] = idx < n2 ? F[f_old + idx - fbase] : in_s[idx];
            }
        }
        tb_sync<NT>();
#pragma unroll
        for (int u = 0; u < HB; ++u) { const uint32_t k = tid + u * NT; if (k < (uint32_t)(T2 - 1)) X[xh0 + u * kXStepH] = xt[u]; }
        fcount += po;
        TSTAMP(3);

        // low-pass + discriminator over the outputs that became computable, every second piece: 8*NT outputs per pass, a lane owns two
        // runs of four ADJACENT outputs (i0 + 4*lane .. and the same 4*NT further on).  Four adjacent outputs share their inputs, so a
        // block of four taps costs two new 16-byte reads per run for 32 multiply-adds, and the lane's eight sums are independent
        // chains.  The inner loop walks three blocks per trip so that the six pairs a run keeps rotate without a register move.
        if (run && ((pc & 1u) || pc + 1 == npieces)) {
            const uint32_t i_hi = (pc + 1 == npieces) ? m : min(m, (pb + o0 + po) & ~1u);
            const f32x4* hk = reinterpret_cast<const f32x4*>(TP);
            float2* Yl = Y;                                              // Yl[0]: predecessor of the pass's first output; Yl[1 + l], Yl[1 + NT + l]: lane l's last outputs
            if (kCompact) { if (tid == 0) Yl[0] = carryY[0]; }            // (visible to lane 0 itself, its only reader, and ordered by the sync behind the tap loop)
            const uint32_t c0v = c0;                                     // (the symbol extractor's position c0 does not move inside the passes)
            for (uint32_t i0 = i_done; i0 < i_hi; i0 += 8 * NT) {
                const uint32_t live = min((uint32_t)(8 * NT), i_hi - i0);
                const uint32_t nvA = 4u * tid < live ? min(4u, live - 4u * tid) : 0u;
                const uint32_t nvB = 4u * NT + 4u * tid < live ? min(4u, live - 4u * NT - 4u * tid) : 0u;
                f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, a2 = {0.f, 0.f}, a3 = {0.f, 0.f};
                f32x2 b0 = {0.f, 0.f}, b1 = {0.f, 0.f}, b2 = {0.f, 0.f}, b3 = {0.f, 0.f};
                {
                    const f32x4* pA = reinterpret_cast<const f32x4*>(F + (i0 - fbase)) + 2 * tid;   // pair j: inputs i + 2j, i + 2j + 1
                    const f32x4* pB = live > 4u * NT ? pA + 2 * NT : pA;                            // (a pass with one run only: the second computes along, unused)
                    f32x4 A0 = pA[0], A1 = pA[1], A2 = pA[2], A3 = pA[3], A4, A5;
                    f32x4 B0 = pB[0], B1 = pB[1], B2 = pB[2], B3 = pB[3], B4, B5;
                    f32x4 kc = hk[0], k1, k2;
                    uint32_t t = 0;
                    for (; t + 12 <= T; t += 12) {                      // (reads run up to ten pairs past t: in bounds, the windows are padded)
                        const uint32_t j = t >> 1, jk = t >> 2;
                        A4 = pA[j + 4]; A5 = pA[j + 5]; B4 = pB[j + 4]; B5 = pB[j + 5]; k1 = hk[jk + 1];
                        __builtin_amdgcn_sched_barrier(0);              // the next block's reads stay up here, in flight during this block's math
                        HD_TB_BLK(A0, A1, A2, A3, kc, a0, a1, a2, a3)
                        HD_TB_BLK(B0, B1, B2, B3, kc, b0, b1, b2, b3)
                        A0 = pA[j + 6]; A1 = pA[j + 7]; B0 = pB[j + 6]; B1 = pB[j + 7]; k2 = hk[jk + 2];
                        __builtin_amdgcn_sched_barrier(0);
                        HD_TB_BLK(A2, A3, A4, A5, k1, a0, a1, a2, a3)
                        HD_TB_BLK(B2, B3, B4, B5, k1, b0, b1, b2, b3)
                        A2 = pA[j + 8]; A3 = pA[j + 9]; B2 = pB[j + 8]; B3 = pB[j + 9]; kc = hk[jk + 3];
                        __builtin_amdgcn_sched_barrier(0);
                        HD_TB_BLK(A4, A5, A0, A1, k2, a0, a1, a2, a3)
                        HD_TB_BLK(B4, B5, B0, B1, k2, b0, b1, b2, b3)
                    }
                    for (; t + 4 <= T; t += 4) {                        // at most two blocks left
                        const uint32_t j = t >> 1;
                        A4 = pA[j + 4]; A5 = pA[j + 5]; B4 = pB[j + 4]; B5 = pB[j + 5]; k1 = hk[(t >> 2) + 1];
                        HD_TB_BLK(A0, A1, A2, A3, kc, a0, a1, a2, a3)
                        HD_TB_BLK(B0, B1, B2, B3, kc, b0, b1, b2, b3)
                        A0 = A2; A1 = A3; A2 = A4; A3 = A5; B0 = B2; B1 = B3; B2 = B4; B3 = B5; kc = k1;
                    }
                    if (t < T) { HD_TB_TAPN(A0.xy, A0.zw, A1.xy, A1.zw, kc.x, a0, a1, a2, a3) HD_TB_TAPN(B0.xy, B0.zw, B1.xy, B1.zw, kc.x, b0, b1, b2, b3) ++t; }
                    if (t < T) { HD_TB_TAPN(A0.zw, A1.xy, A1.zw, A2.xy, kc.y, a0, a1, a2, a3) HD_TB_TAPN(B0.zw, B1.xy, B1.zw, B2.xy, kc.y, b0, b1, b2, b3) ++t; }
                    if (t < T) { HD_TB_TAPN(A1.xy, A1.zw, A2.xy, A2.zw, kc.z, a0, a1, a2, a3) HD_TB_TAPN(B1.xy, B1.zw, B2.xy, B2.zw, kc.z, b0, b1, b2, b3) ++t; }
                    const f32x2 lastA = nvA == 4 ? a3 : nvA == 3 ? a2 : nvA == 2 ? a1 : a0;
                    const f32x2 lastB = nvB == 4 ? b3 : nvB == 3 ? b2 : nvB == 2 ? b1 : b0;
                    Yl[1 + tid] = make_float2(lastA.x, lastA.y);
                    Yl[1 + NT + tid] = make_float2(lastB.x, lastB.y);
                }
                TSTAMP(12);
                tb_sync<NT>();
                {
                    // discriminator, run by run (one copy of the code): d[i] = arg(y[i] * conj(y[i-1]))
                    f32x2 c0 = a0, c1 = a1, c2 = a2, c3 = a3;
                    uint32_t nv = nvA, i = i0 + 4 * tid;
                    float2 qv = Yl[tid];
#pragma unroll 1
                    for (int w = 0; w < 2; ++w) {
                        if (nv) {
                            float pr = qv.x, pi = qv.y;
                            if (i == 0 && !kin.primed) { pr = c0.x; pi = c0.y; }   // very first sample: arg(y0*conj(y0)) (FSK2_Demod.h:35)
                            float d[4];
                            d[0] = discriminate(c0.x, c0.y, pr, pi);
                            d[1] = discriminate(c1.x, c1.y, c0.x, c0.y);
                            d[2] = discriminate(c2.x, c2.y, c1.x, c1.y);
                            d[3] = discriminate(c3.x, c3.y, c2.x, c2.y);
                            const f32x2 av[4] = {c0, c1, c2, c3};
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                if ((uint32_t)u < nv) { const uint32_t b = __builtin_bit_cast(uint32_t, d[u]); ck0 += b; ck1 += (i + (uint32_t)u + 1u) * b; }
                            float* dm = a.demod + (size_t)s * a.demod_stride + i;   // i and demod_stride are even
                            const uint32_t pos = end_old + i;           // SymbolExtractor::pushSamples: append to the backlog ring
                            const float4 d4 = make_float4(d[0], d[1], d[2], d[3]);
                            if (nv == 4 && !(i & 3u)) *reinterpret_cast<float4*>(dm) = d4;
                            else {
                                if (nv >= 2) *reinterpret_cast<float2*>(dm) = make_float2(d[0], d[1]); else dm[0] = d[0];
                                if (nv == 4) *reinterpret_cast<float2*>(dm + 2) = make_float2(d[2], d[3]); else if (nv == 3) dm[2] = d[2];
                            }
                            if (nv == 4 && !(pos & 3u)) *reinterpret_cast<float4*>(vring + (pos & rmask)) = d4;   // (an aligned quad never straddles the ring's end)
                            else {
#pragma unroll
                                for (int u = 0; u < 4; ++u) if ((uint32_t)u < nv) vring[(pos + u) & rmask] = d[u];
                            }
                            if (do_sums) {
                                if (nv == 4 && !((pos - c0v) & 3u)) *reinterpret_cast<float4*>(V + (pos - c0v)) = d4;
                                else {
#pragma unroll
                                    for (int u = 0; u < 4; ++u) if ((uint32_t)u < nv) V[pos + u - c0v] = d[u];
                                }
                            }
                            if (a.filtered) {
#pragma unroll
                                for (int u = 0; u < 4; ++u) if ((uint32_t)u < nv) a.filtered[(size_t)s * a.demod_stride + i + u] = make_float2(av[u].x, av[u].y);
                            }
                            if (i + nv == m) {                          // the lane that owns the run's last output
                                DemodCarry k; k.primed = 1; k._pad = 0;
                                const f32x2 lastv = nv == 4 ? c3 : nv == 3 ? c2 : nv == 2 ? c1 : c0;
                                k.re = lastv.x; k.im = lastv.y;
                                a.carry_out[s] = k;
                            }
                        }
                        c0 = b0; c1 = b1; c2 = b2; c3 = b3; nv = nvB; i = i0 + 4 * NT + 4 * tid; qv = Yl[NT + tid];
                    }
                }
                TSTAMP(14);
                tb_sync<NT>();                                          // Yl is read above; its carry slot and the next pass rewrite it
                {   // the pass's last output becomes the next pass's predecessor
                    const uint32_t lo = live - 1, lw = lo / (4u * NT), ll = (lo % (4u * NT)) >> 2;
                    if (tid == ll) { const float2 cy = Yl[1 + lw * NT + tid]; Yl[0] = cy; if (kCompact) carryY[0] = cy; }
                }
            }
            i_done = i_hi;
            TSTAMP(4);
            {   // slide the low-pass window: what is still needed starts at input index i_done
                const uint32_t shf = i_done - fbase, keep = fcount - shf;
                if (shf) {
                    tb_sync<NT>();
                    for (uint32_t k0 = 0; k0 < keep; k0 += 4 * NT) {
                        float2 t4[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) { const uint32_t k = k0 + tid + u * NT; t4[u] = k < keep ? F[k + shf] : make_float2(0.f, 0.f); }
                        tb_sync<NT>();
#pragma unroll
                        for (int u = 0; u < 4; ++u) { const uint32_t k = k0 + tid + u * NT; if (k < keep) F[k] = t4[u]; }
                    }
                    fbase = i_done; fcount = keep;
                }
            }
            TSTAMP(5);
            if (do_sums) vcnt = end_old + i_done - c0;
        }
        ++pc;
    }
    tb_sync<NT>();

    // ---- The second half of the symbol extractor (edge search, run means) reads back what this call and earlier ones left in the global
    // rings: flag mask, window sums, samples.  Those reads are issued HERE, into registers, so that their round trip runs under the
    // carry / slide stores below; they land in LDS (over the stream windows, which are dead by then) afterwards.
    const uint32_t h = st.held;
    const uint32_t end = st.base + h;
    const uint32_t pend = end - R + 1;
    const bool hunt = m && do_sums && search;
    const uint32_t limit = hunt ? h - q.spb : 0u;                      // backlog indices searched: [R, limit)
    const uint32_t carried_to = st.run_pos - st.base;
    const uint32_t wr0 = (st.base + R) & ~63u;
    const uint32_t nw = hunt ? ((st.base + limit) - wr0 + 63u) >> 6 : 0u;
    // This stream's carve of the search-phase region [dyn_off, lds_bytes): the flag-mask image of the backlog it searches (nw words: a few dozen for
    // a few symbols of backlog, the whole ring's worth for a stream that has been idle for seconds), then the sample cache the run sums add up from
    // -- [carried_to, limit), i.e. about one call's worth -- and the window sums of the searchable backlog [0, limit) with what is left.  Either cache
    // may hold only a prefix: what lies behind it comes from the global rings as before (a round trip per edge zone / per strip of a run -- which is
    // what made the busiest streams' tails the longest waves of a step launch while the caches were all-or-nothing and a 4 KiB mask image sat in front).
    const uint32_t lmw = (nw + 1u) & ~1u;
    const uint32_t dyn_floats = hunt ? (((a.lds_bytes - a.dyn_off) - lmw * 8u) >> 2) & ~3u : 0u;
    const uint32_t vc_want = (hunt && limit > carried_to) ? limit - carried_to : 0u;
    // Both caches are images of RING-ALIGNED quads (round 5): their first element is the ring position rounded down to a multiple of four, so that every
    // load is an aligned 16-byte one that cannot straddle the ring's end -- a dozen load instructions where there were thirty-odd 4-byte ones (inside a step
    // launch every global-memory instruction of this wave queues behind the stage-1 workers' tile loads).  woff / voff: where backlog position 0 /
    // sample carried_to sits in its image.
    const uint32_t woff = st.base & 3u, voff = (st.base + carried_to) & 3u;
    const uint32_t vc_cap = min((vc_want + voff + 7u) & ~3u, (dyn_floats * 5u / 8u) & ~3u);
    const uint32_t wc_cap = dyn_floats - vc_cap;
    const uint32_t wc_n = hunt ? min(limit, wc_cap > woff + 4u ? wc_cap - woff - 4u : 0u) : 0u;   // window sums [0, wc_n) of the backlog
    // the run sums add up the backlog samples [carried_to, frontier) (frontier <= limit), in pieces
    const uint32_t vc_n = min(vc_want, vc_cap > voff + 4u ? vc_cap - voff - 4u : 0u);             // samples [carried_to, carried_to + vc_n)
    const uint32_t wq_n = wc_n ? (wc_n + woff + 3u) >> 2 : 0u, vq_n = vc_n ? (vc_n + voff + 3u) >> 2 : 0u;   // quads of each image
    const uint32_t wpos4 = st.base & ~3u, vpos4 = (st.base + carried_to) & ~3u;                     // ring positions of the images' first elements
    constexpr int MB = 512 / NT, CB = 1024 / NT / 4 > 0 ? 1024 / NT / 4 : 1, SB = CB;      // first batches: 512 mask words, 1024 window sums, 1024 samples
    unsigned long long tm0[MB];
    float4 tc0[CB], ts0[SB];
    if (hunt) {
        __threadfence_block();                                          // this call's own ring stores first
#pragma unroll
        for (int u = 0; u < MB; ++u) { const uint32_t i = tid + u * NT; tm0[u] = i < nw ? gmask[((wr0 + 64u * i) & rmask) >> 6] : 0ull; }
#pragma unroll
        for (int u = 0; u < CB; ++u) { const uint32_t qd = tid + u * NT; tc0[u] = qd < wq_n ? *reinterpret_cast<const float4*>(gw + ((wpos4 + 4u * qd) & rmask)) : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
        for (int u = 0; u < SB; ++u) { const uint32_t qd = tid + u * NT; ts0[u] = qd < vq_n ? *reinterpret_cast<const float4*>(vring + ((vpos4 + 4u * qd) & rmask)) : make_float4(0.f, 0.f, 0.f, 0.f); }
    }

    // ---- what the next call finds: stage-2 history of an idle stream, [history | leftover pending] at the front of the other
    // low-pass buffer, the head of this run's input (FirHistory), the discriminator carry
    if (!n1) for (uint32_t k = tid; k < (uint32_t)(T2 - 1); k += NT) a.hist2_out[(size_t)s * (T2 - 1) + k] = a.hist2_in[(size_t)s * (T2 - 1) + k];
    if (run) {
        // F[0 ..) is input index m on: the last H inputs of this run followed by the leftover pending samples
        for (uint32_t k = tid; k < H + c.pend_after; k += NT) nxt[fhc - H + k] = F[k];
    } else {
        // low-pass did not run: the buffer moves as it is (history of whatever length the last run had, pending, new)
        __threadfence_block();
        tb_sync<NT>();
        const uint32_t cnt = fhc + c.pend_after;
        for (uint32_t k = tid; k < cnt; k += NT) nxt[k] = cur[k];
    }
    if (sc_head_save(c)) {      // the stream ran in the previous call and does not in this one: its head moves aside before that buffer's turn comes again
        const uint32_t hk = sc_head_n(c);
        for (uint32_t k = tid; k < hk; k += NT) head_side[k] = head_prevbuf[k];
    }
    if (!run && tid == 0) a.carry_out[s] = kin;
    TSTAMP(7);

    // ---- symbol extractor, second half: edge search, run means, bits (SymbolExtractor.h:129-158).  (A lambda, so that its early exits
    // all lead to the one place behind it where a completed spectrum buffer is transformed.)
    auto second_half = [&]() {
    uint32_t* outw = slot + sizeof(BitsHeader) / 4;
    const uint32_t cap_bits = (a.slot_words - sizeof(BitsHeader) / 4) * 32;
    if (!m) {
        if (tid == 0) { hdr->nbits = 0; hdr->held_after = old.held; hdr->nflips = 0; hdr->overflow = 0; hdr->uncached = old.base + old.held - old.cached; }
        return;
    }
    if (!do_sums) {
        if (tid == 0) { a.sym[s] = st; hdr->nbits = 0; hdr->held_after = h; hdr->nflips = 0; hdr->overflow = 0; hdr->uncached = st.base + h - st.cached; }
        return;
    }
    if (!search) {
        if (tid == 0) {
            if ((int32_t)(pend - st.cached) > 0) st.cached = pend;
            a.sym[s] = st;
            hdr->nbits = 0; hdr->held_after = h; hdr->nflips = 0; hdr->overflow = 0; hdr->uncached = end - st.cached;
        }
        return;
    }
    unsigned long long* lmask = reinterpret_cast<unsigned long long*>(lds + a.dyn_off);   // lmask[i]: ring word (wr0 >> 6) + i
    uint32_t* flips = reinterpret_cast<uint32_t*>(lds + a.flips_off);
    uint32_t* runinfo = flips + a.fl_cap;
    float* strips = reinterpret_cast<float*>(lds + a.strips_off);
    float* vc4 = reinterpret_cast<float*>(lds + a.dyn_off + lmw * 8u); // the sample image (ring-aligned quads) ...
    float* wc4 = vc4 + vc_cap;                                          // ... and the window-sum image
    const float* vc = vc4 + voff;                                       // vc[k] = backlog sample carried_to + k
    const float* wc = wc4 + woff;                                       // wc[k] = window sum of backlog position k
    const uint32_t fl_cap = a.fl_cap;
    tb_sync<NT>();                                                      // the slide above read the stream windows these images overwrite
    {
#pragma unroll
        for (int u = 0; u < MB; ++u) { const uint32_t i = tid + u * NT; if (i < nw) lmask[i] = tm0[u]; }
#pragma unroll
        for (int u = 0; u < CB; ++u) { const uint32_t qd = tid + u * NT; if (qd < wq_n) reinterpret_cast<float4*>(wc4)[qd] = tc0[u]; }
#pragma unroll
        for (int u = 0; u < SB; ++u) { const uint32_t qd = tid + u * NT; if (qd < vq_n) reinterpret_cast<float4*>(vc4)[qd] = ts0[u]; }
        // longer backlogs than the first batches cover (idle or freshly re-parameterised streams): plain loops
        for (uint32_t i = tid + MB * NT; i < nw; i += NT) lmask[i] = gmask[((wr0 + 64u * i) & rmask) >> 6];
        constexpr int RB = 4;                                           // (four 16-byte loads per lane in flight per trip: a loop of single loads is a round trip per element)
#pragma unroll 1
        for (uint32_t q0 = tid + CB * NT; q0 < wq_n; q0 += RB * NT) {
            float4 t[RB];
#pragma unroll
            for (int u = 0; u < RB; ++u) { const uint32_t qd = q0 + u * NT; t[u] = qd < wq_n ? *reinterpret_cast<const float4*>(gw + ((wpos4 + 4u * qd) & rmask)) : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
            for (int u = 0; u < RB; ++u) { const uint32_t qd = q0 + u * NT; if (qd < wq_n) reinterpret_cast<float4*>(wc4)[qd] = t[u]; }
        }
#pragma unroll 1
        for (uint32_t q0 = tid + SB * NT; q0 < vq_n; q0 += RB * NT) {
            float4 t[RB];
#pragma unroll
            for (int u = 0; u < RB; ++u) { const uint32_t qd = q0 + u * NT; t[u] = qd < vq_n ? *reinterpret_cast<const float4*>(vring + ((vpos4 + 4u * qd) & rmask)) : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
            for (int u = 0; u < RB; ++u) { const uint32_t qd = q0 + u * NT; if (qd < vq_n) reinterpret_cast<float4*>(vc4)[qd] = t[u]; }
        }
    }
    tb_sync<NT>();
    TSTAMP(8);
    auto wsum = [&](uint32_t i) -> float { return i < wc_n ? wc[i] : gw[(st.base + i) & rmask]; };
    if (wave == 0) {
        uint32_t pos = R, nfl = 0, overflow = 0;
        uint32_t frontier = 0xFFFFFFFFu;                                // backlog index no future flip point can precede
        while (pos < limit) {
            const uint32_t lo = find_flag_rel(lmask, st.base, wr0, pos, limit, true);
            if (lo == 0xFFFFFFFFu) break;
            const uint32_t hi = find_flag_rel(lmask, st.base, wr0, lo + 1, limit, false);
            if (hi == 0xFFFFFFFFu) { frontier = lo; break; }            // an edge zone is open: the next flip lies at or after lo
            unsigned long long key = 0ull;                              // first maximum of the weight over [lo, hi): (weight bits, ~index)
            constexpr int ZB = 4;
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
        if (frontier == 0xFFFFFFFFu) frontier = max(pos, limit);        // nothing flagged in [pos, limit)
        if (lane == 0) { sh[0] = nfl; sh[1] = overflow; sh[2] = frontier; }
    }
    tb_sync<NT>();
    const uint32_t nfl = sh[0];
    TSTAMP(9);

    // per-run sums in element order (std::accumulate); the run in progress is carried across calls (SymState::run_sum covers
    // [base, run_pos)) up to the search frontier -- the same left-to-right chain of adds the reference performs in one go
    float* sb = strips + wave * kTailStrip;
    constexpr int NX = kTailStrip / 64;
    auto chain = [&](float acc, uint32_t a0, uint32_t b) -> float {     // acc + v[a0] + ... + v[b-1] (backlog indices)
        if (a0 >= b) return acc;
        if (a0 >= carried_to && b - carried_to <= vc_n) {               // the samples are in LDS already (wave-uniform addresses: broadcast reads)
            const float* v0 = vc + (a0 - carried_to);
            uint32_t n = b - a0, i = 0;
            for (; i < n && ((a0 - carried_to + voff + i) & 3u); ++i) acc = acc + v0[i];
            const float4* v4 = reinterpret_cast<const float4*>(v0 + i);
            const uint32_t nq = (n - i) >> 2;
            uint32_t q = 0;
#define HD_TB_ADD4(c_) { acc = acc + (c_).x; acc = acc + (c_).y; acc = acc + (c_).z; acc = acc + (c_).w; }
            if (nq >= 4) {
                float4 c0_ = v4[0], c1_ = v4[1], c2_ = v4[2], c3_ = v4[3], d0_, d1_, d2_, d3_;
                for (; q + 12 <= nq; q += 8) {                          // two blocks of 16 samples per trip: each block's reads fly under the other's adds
                    d0_ = v4[q + 4]; d1_ = v4[q + 5]; d2_ = v4[q + 6]; d3_ = v4[q + 7];
                    __builtin_amdgcn_sched_barrier(0);
                    HD_TB_ADD4(c0_) HD_TB_ADD4(c1_) HD_TB_ADD4(c2_) HD_TB_ADD4(c3_)
                    c0_ = v4[q + 8]; c1_ = v4[q + 9]; c2_ = v4[q + 10]; c3_ = v4[q + 11];
                    __builtin_amdgcn_sched_barrier(0);
                    HD_TB_ADD4(d0_) HD_TB_ADD4(d1_) HD_TB_ADD4(d2_) HD_TB_ADD4(d3_)
                }
                HD_TB_ADD4(c0_) HD_TB_ADD4(c1_) HD_TB_ADD4(c2_) HD_TB_ADD4(c3_)
                q += 4;
            }
            for (; q < nq; ++q) { const float4 c_ = v4[q]; HD_TB_ADD4(c_) }
#undef HD_TB_ADD4
            for (i += 4 * q; i < n; ++i) acc = acc + v0[i];
            return acc;
        }
        float nx[NX];
#pragma unroll
        for (int j = 0; j < NX; ++j) { const uint32_t k = a0 + 64 * j + lane; nx[j] = k < b ? vring[(st.base + k) & rmask] : 0.0f; }
        for (uint32_t k0 = a0; k0 < b; k0 += kTailStrip) {
#pragma unroll
            for (int j = 0; j < NX; ++j) sb[64 * j + lane] = nx[j];
            if (k0 + kTailStrip < b) {
#pragma unroll
                for (int j = 0; j < NX; ++j) { const uint32_t k = k0 + kTailStrip + 64 * j + lane; nx[j] = k < b ? vring[(st.base + k) & rmask] : 0.0f; }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);                         // lgkmcnt(0): the strip has landed (wave-private)
            __builtin_amdgcn_wave_barrier();
            const uint32_t cnt = min(kTailStrip, b - k0);
            const float4* s4 = reinterpret_cast<const float4*>(sb);
            uint32_t i = 0;
            if (cnt >= 32) {
                // (two register images taking turns: no copy of the block in flight into the current one -- symbols.hip, phase C)
                float4 cc[8], nn[8];
                auto ld = [&](float4 (&x)[8], const uint32_t at) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) x[u] = s4[(at >> 2) + u];
                };
                auto adds = [&](const float4 (&x)[8]) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) { acc = acc + x[u].x; acc = acc + x[u].y; acc = acc + x[u].z; acc = acc + x[u].w; }
                };
                ld(cc, 0u);                                                 // invariant: cc holds samples [i, i + 32), i + 32 <= cnt
                for (; i + 96 <= cnt; i += 64) { ld(nn, i + 32); adds(cc); ld(cc, i + 64); adds(nn); }
                if (i + 64 <= cnt) { ld(nn, i + 32); adds(cc); adds(nn); i += 64; }
                else { adds(cc); i += 32; }
            }
            for (; i < cnt; ++i) acc = acc + sb[i];
            __builtin_amdgcn_wave_barrier();
        }
        return acc;
    };
    const uint32_t frontier = sh[2];
    for (uint32_t r = wave; r <= nfl; r += NW) {                        // runs dealt to the waves; task nfl is the carry for the next call
        const bool tail_task = r == nfl;
        const uint32_t ra = r ? flips[r - 1] : 0u;
        uint32_t from, to;
        float acc0 = 0.0f;
        if (!tail_task) { to = flips[r]; from = r ? ra : min(carried_to, to); if (!r) acc0 = st.run_sum; }
        else { from = nfl ? ra : carried_to; to = max(from, frontier); if (!nfl) acc0 = st.run_sum; }
        const float acc = chain(acc0, from, to);
        if (lane == 0) {
            if (tail_task) sh[3] = __builtin_bit_cast(uint32_t, acc);
            else {
                const float mean = acc / (float)(to - ra);
                const uint32_t cnt = (uint32_t)roundf((float)(to - ra) / (float)q.spb);
                runinfo[r] = (cnt << 1) | (mean > 0.0f ? 1u : 0u);
            }
        }
    }
    tb_sync<NT>();
    TSTAMP(10);
    if (tid == 0) {
        uint32_t nbits = 0, curw = 0, overflow = sh[1];
        for (uint32_t r = 0; r < nfl; ++r) {
            const uint32_t bit = runinfo[r] & 1u;
            for (uint32_t k = runinfo[r] >> 1; k; --k) {
                if (nbits >= cap_bits) { overflow |= 1u; break; }
                curw |= bit << (nbits & 31);
                if ((nbits & 31) == 31) { outw[nbits >> 5] = curw; curw = 0; }
                ++nbits;
            }
        }
        if (nbits & 31) outw[nbits >> 5] = curw;
        if (a.flips_dbg)
            for (uint32_t r = 0; r < nfl && r < a.flips_cap; ++r) a.flips_dbg[(size_t)s * a.flips_cap + r] = flips[r];
        const uint32_t last = nfl ? flips[nfl - 1] : 0u;               // erase the consumed prefix (SymbolExtractor.h:156-157) = advance the base
        st.run_pos = st.base + max(nfl ? last : carried_to, frontier);
        st.run_sum = __builtin_bit_cast(float, sh[3]);
        st.cached = pend;
        st.base += last;
        st.held = h - last;
        a.sym[s] = st;
        hdr->nbits = nbits; hdr->held_after = st.held; hdr->nflips = nfl; hdr->overflow = overflow; hdr->uncached = end - st.cached;
    }
    TSTAMP(11);
    };
    second_half();
    {   // the call's discriminator checksum (BitsHeader::demod_ck): lanes -> wave -> (256-lane variant) workgroup
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { ck0 += (uint32_t)__shfl_xor((int)ck0, o, 64); ck1 += (uint32_t)__shfl_xor((int)ck1, o, 64); }
        if constexpr (NT == 64) {
            if (tid == 0) { hdr->demod_ck[0] = ck0; hdr->demod_ck[1] = ck1; hdr->demod_n = m; }
        } else {
            tb_sync<NT>();
            if (tid == 0) { sh[5] = 0; sh[6] = 0; }
            tb_sync<NT>();
            if (lane == 0) { atomicAdd(&sh[5], ck0); atomicAdd(&sh[6], ck1); }
            tb_sync<NT>();
            if (tid == 0) { hdr->demod_ck[0] = sh[5]; hdr->demod_ck[1] = sh[6]; hdr->demod_n = m; }
        }
    }
    // ---- the stream's spectrum, when its 4096-sample buffer completed in this call (Decoder.h:475-489): transform, half swap, power and
    // AFC statistics by this same wave (spectrum_wave.h) -- no launches of their own, no fft_raw round trip.  The buffer's last samples
    // were stored by this workgroup a moment ago.
    if (a.fft_tw && c.fft_run) {
        __threadfence_block();
        tb_sync<NT>();                                                  // every wave is done with the LDS images
        if (wave == 0)
            spectrum_wave_body(a.fft_in + (size_t)s * kFftBins, a.fft_tw, a.spec, a.power, a.stats, s, a.rate, a.bins_sep, reinterpret_cast<float*>(lds + kTailHdrBytes), a.seq);
    }
    // the call's tag, LAST: every store this wave has issued -- header, bits, spectrum statistics, all by wave 0 -- has been acknowledged before it goes out
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef HD_RING_FAULT
    if (tid == 0 && g_tail_fault_armed && atomicCAS(&g_tail_fault_fired, 0u, 1u) == 0u) return;     // (the host's wait for the tag runs out: tests/test_gpu_fault.py)
#endif
    if (tid == 0) __hip_atomic_store(&hdr->seq, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    TSTAMP_WRITE();
}

}  // namespace HD_ARITH_NS
}  // namespace hd
