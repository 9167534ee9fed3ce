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
//     positions per step through ballots and find-first-set); per-run sums are accumulated in element order with
//     the wave loading 256 samples at a time and v_readlane feeding a wave-uniform accumulator; lane 0 packs bits.
#include <hip/hip_runtime.h>

#include "launch.h"

namespace hd {

constexpr int kAvgLanes = 256;
constexpr uint32_t kMaxFlipsPerCall = 2048;   // LDS flip list of the scan kernel (overflow is flagged)

__device__ __forceinline__ int sgnf(float v) { return (0.0f < v) - (v < 0.0f); }

// State after this call's samples were appended (SymbolExtractor.h:116-124: the vent happens before the append).
__device__ __forceinline__ SymState state_after_push(SymState st, const SymbolParams& q, uint32_t m)
{
    if (st.held > kVentLimit) { st.base += st.held; st.held = 0; st.cached = st.base; }
    if (q.reset) st.cached = st.base;
    st.held += m;
    return st;
}

// Window sums for the NEW candidate positions.  One lane owns 4 consecutive positions: it walks its R+3 left-window
// samples (then the R+3 right-window samples) once with 16-byte LDS reads and adds each sample to every one of its
// four accumulators whose window contains it.  Each accumulator still receives exactly its own R samples in index
// order, so the sums are bit-identical to std::accumulate, with 1/16 of the LDS instructions of the scalar form.
constexpr int kAvgPos = 4;                                  // positions per lane
constexpr int kAvgSpan = kAvgLanes * kAvgPos;               // positions per workgroup (1024)

__device__ __forceinline__ void window_sums(const float* __restrict__ w, uint32_t R, float acc[kAvgPos])
{
    // accumulator j sums w[j .. j+R)
#pragma unroll
    for (int j = 0; j < kAvgPos; ++j) acc[j] = 0.0f;
    const uint32_t total = R + kAvgPos - 1;                 // samples touched: w[0 .. R+3)
    uint32_t e = 0;
    {   // head chunk: element e feeds accumulators j <= e
        const float4 x = *reinterpret_cast<const float4*>(w);
        const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < kAvgPos; ++j)
                if (j <= u && (uint32_t)u < (uint32_t)j + R) acc[j] = acc[j] + xs[u];
        e = 4;
    }
    for (; e + 4 <= R; e += 4) {                            // interior: every element feeds all four
        const float4 x = *reinterpret_cast<const float4*>(w + e);
#pragma unroll
        for (int j = 0; j < kAvgPos; ++j) acc[j] = acc[j] + x.x;
#pragma unroll
        for (int j = 0; j < kAvgPos; ++j) acc[j] = acc[j] + x.y;
#pragma unroll
        for (int j = 0; j < kAvgPos; ++j) acc[j] = acc[j] + x.z;
#pragma unroll
        for (int j = 0; j < kAvgPos; ++j) acc[j] = acc[j] + x.w;
    }
    for (; e < total; e += 4) {                             // tail chunks: element e feeds accumulators with e < j + R
        const float4 x = *reinterpret_cast<const float4*>(w + e);
        const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < kAvgPos; ++j)
                if (e + u >= (uint32_t)j && e + u < (uint32_t)j + R) acc[j] = acc[j] + xs[u];
    }
}

__global__ __launch_bounds__(kAvgLanes) void k_sym_avg(const float* __restrict__ tail, uint32_t ring_cap,
                                                        const SymState* __restrict__ sym, unsigned long long* __restrict__ flipmask,
                                                        float* __restrict__ weight, const SymbolParams* __restrict__ sp,
                                                        const StreamCall* __restrict__ call)
{
    extern __shared__ __attribute__((aligned(16))) float win[];   // tail[c0 - R, c0 + span + R + pad)
    __shared__ unsigned long long words[kAvgSpan / 64];
    const uint32_t s = blockIdx.y;
    const uint32_t m = call[s].fir_m;
    if (!m) return;
    const SymbolParams q = sp[s];
    const SymState st = state_after_push(sym[s], q, m);
    const uint32_t h = st.held;
    if (h < q.min_held || h < q.spb) return;
    const uint32_t R = q.R, rmask = ring_cap - 1;
    const uint32_t end = st.base + h;
    const uint32_t pend = end - R + 1;                     // first position whose right window is still incomplete
    const uint32_t c0 = (st.cached & ~63u) + blockIdx.x * kAvgSpan;   // mask words are written whole: start 64-aligned
    if ((int32_t)(pend - c0) <= 0) return;
    const float* v = tail + (size_t)s * ring_cap;
    const uint32_t w0 = c0 - R;
    const uint32_t wn = kAvgSpan + 2 * R + 8;
    for (uint32_t k = threadIdx.x; k < wn; k += kAvgLanes) win[k] = v[(w0 + k) & rmask];
    if (threadIdx.x < kAvgSpan / 64) words[threadIdx.x] = 0ull;
    __syncthreads();
    const uint32_t p0 = c0 + threadIdx.x * kAvgPos;        // this lane's first position; its left window starts at win[4*lane]
    float sl[kAvgPos], sr[kAvgPos];
    window_sums(win + threadIdx.x * kAvgPos, R, sl);
    // right windows start R samples later; R is not a multiple of 4 in general, so realign through a 4-float shift
    {
        const uint32_t off = threadIdx.x * kAvgPos + R;    // win index of position p0
        const uint32_t al = off & ~3u, sh = off & 3u;      // aligned start, residual shift
        // accumulator j sums win[off + j .. off + j + R) = aligned[sh + j .. sh + j + R)
        const float* w = win + al;
#pragma unroll
        for (int j = 0; j < kAvgPos; ++j) sr[j] = 0.0f;
        const uint32_t total = sh + R + kAvgPos - 1;
        for (uint32_t e = 0; e < total; e += 4) {
            const float4 x = *reinterpret_cast<const float4*>(w + e);
            const float xs[4] = {x.x, x.y, x.z, x.w};
            const bool interior = e >= sh + kAvgPos - 1 && e + 4 <= sh + R;
            if (interior) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < kAvgPos; ++j) sr[j] = sr[j] + xs[u];
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < kAvgPos; ++j)
                        if (e + u >= sh + (uint32_t)j && e + u < sh + (uint32_t)j + R) sr[j] = sr[j] + xs[u];
            }
        }
    }
    unsigned int bits = 0;
#pragma unroll
    for (int j = 0; j < kAvgPos; ++j) {
        const uint32_t p = p0 + j;
        if ((int32_t)(pend - p) > 0) {
            const float al = sl[j] / (float)R, ar = sr[j] / (float)R;   // no clamping: p >= base+R is all the search reads, p+R <= end
            if (sgnf(al) != sgnf(ar)) bits |= 1u << j;
            const float d = ar - al;
            weight[(size_t)s * ring_cap + (p & rmask)] = q.float_abs ? __builtin_fabsf(d) : (float)abs((int)d);
        }
    }
    // lane l owns bits [4l, 4l+4) of the workgroup's 1024-position span: 16 lanes per 64-bit word
    if (bits) atomicOr(&words[threadIdx.x >> 4], (unsigned long long)bits << ((threadIdx.x & 15) * 4));
    __syncthreads();
    if (threadIdx.x < kAvgSpan / 64)
        flipmask[(size_t)s * (ring_cap / 64) + (((c0 + threadIdx.x * 64) & rmask) >> 6)] = words[threadIdx.x];
}

// First backlog index l in [from, to) whose flag equals `want`, or 0xFFFFFFFF.  One 64-bit mask word per lane per step.
__device__ __forceinline__ uint32_t find_flag(const unsigned long long* __restrict__ masks, uint32_t base, uint32_t rmask,
                                              uint32_t from, uint32_t to, bool want)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t wr0 = (base + from) & ~63u;             // ring position of the first word
    for (uint32_t it = 0;; ++it) {
        const uint32_t wstart = wr0 + it * 4096u;
        if ((int32_t)(wstart - base) >= (int32_t)to) break;
        const uint32_t wr = wstart + lane * 64u;
        const int32_t lw = (int32_t)(wr - base);           // backlog index of this word's bit 0 (may be < from)
        unsigned long long w = 0;
        if (lw < (int32_t)to) {
            w = masks[(wr & rmask) >> 6];
            if (!want) w = ~w;
            const int32_t lo = (int32_t)from - lw;         // keep bits >= lo
            if (lo > 0) w = lo >= 64 ? 0ull : (w & (~0ull << lo));
            const int32_t hi = (int32_t)to - lw;           // keep bits < hi
            if (hi < 64) w &= (1ull << hi) - 1ull;
        }
        const unsigned long long hit = __ballot(w != 0ull);
        if (hit) {
            const int src = __ffsll((long long)hit) - 1;
            const unsigned long long ww = __shfl(w, src, 64);
            return (uint32_t)((int32_t)(wstart - base) + src * 64 + (__ffsll((long long)ww) - 1));
        }
    }
    return 0xFFFFFFFFu;
}

__global__ __launch_bounds__(64) void k_sym_scan(const float* __restrict__ tail, uint32_t ring_cap, SymState* __restrict__ sym,
                                                   const unsigned long long* __restrict__ flipmask, const float* __restrict__ weight,
                                                   const SymbolParams* __restrict__ sp, const StreamCall* __restrict__ call,
                                                   uint32_t* __restrict__ slots, uint32_t slot_words,
                                                   uint32_t* __restrict__ flips_dbg, uint32_t flips_cap)
{
    __shared__ uint32_t flips[kMaxFlipsPerCall];
    __shared__ uint32_t runinfo[kMaxFlipsPerCall];          // (count << 1) | bit
    const uint32_t s = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    uint32_t* slot = slots + (size_t)s * slot_words;
    BitsHeader* hdr = reinterpret_cast<BitsHeader*>(slot);
    uint32_t* words = slot + sizeof(BitsHeader) / 4;
    const uint32_t cap_bits = (slot_words - sizeof(BitsHeader) / 4) * 32;
    const uint32_t m = call[s].fir_m;
    const SymState old = sym[s];
    if (!m) {                                               // symbol stage not reached this call
        if (lane == 0) { hdr->nbits = 0; hdr->held_after = old.held; hdr->nflips = 0; hdr->overflow = 0; hdr->uncached = old.base + old.held - old.cached; }
        return;
    }
    const SymbolParams q = sp[s];
    SymState st = state_after_push(old, q, m);
    const uint32_t h = st.held;
    if (h < q.min_held || h < q.spb) {
        if (lane == 0) { sym[s] = st; hdr->nbits = 0; hdr->held_after = h; hdr->nflips = 0; hdr->overflow = 0; hdr->uncached = st.base + h - st.cached; }
        return;
    }
    const uint32_t rmask = ring_cap - 1;
    const float* v = tail + (size_t)s * ring_cap;
    const unsigned long long* masks = flipmask + (size_t)s * (ring_cap / 64);
    const float* wgt = weight + (size_t)s * ring_cap;
    const uint32_t limit = h - q.spb;                       // backlog indices searched: [R, limit)
    uint32_t pos = q.R, nfl = 0, overflow = 0;
    while (pos < limit) {
        const uint32_t lo = find_flag(masks, st.base, rmask, pos, limit, true);
        if (lo == 0xFFFFFFFFu) break;
        const uint32_t hi = find_flag(masks, st.base, rmask, lo + 1, limit, false);
        if (hi == 0xFFFFFFFFu) break;
        float bw = -1.0f;                                   // first maximum of the weight over [lo, hi)
        uint32_t bi = 0xFFFFFFFFu;
        for (uint32_t i = lo + lane; i < hi; i += 64) {
            const float w = wgt[(st.base + i) & rmask];
            if (bi == 0xFFFFFFFFu || w > bw) { bw = w; bi = i; }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const float ow = __shfl_down(bw, off, 64);
            const uint32_t oi = __shfl_down(bi, off, 64);
            if (oi != 0xFFFFFFFFu && (bi == 0xFFFFFFFFu || ow > bw || (ow == bw && oi < bi))) { bw = ow; bi = oi; }
        }
        const uint32_t f = __shfl(bi, 0, 64);
        if (nfl < kMaxFlipsPerCall) { if (lane == 0) flips[nfl] = f; ++nfl; } else { overflow = 1; break; }
        pos = f + q.R;
    }
    __syncthreads();
    // per-run sums in element order (std::accumulate): the wave fetches 256 samples per step (the next step's loads
    // are issued before the current 256 are consumed), v_readlane feeds a wave-uniform accumulator one sample at a time
    for (uint32_t r = 0; r < nfl; ++r) {
        const uint32_t a = r ? flips[r - 1] : 0u, b = flips[r];
        float acc = 0.0f;
        float nx[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t k = a + 64 * j + lane;
            nx[j] = k < b ? v[(st.base + k) & rmask] : 0.0f;
        }
        for (uint32_t k0 = a; k0 < b; k0 += 256) {
            int x[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = __builtin_bit_cast(int, nx[j]);
            if (k0 + 256 < b) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t k = k0 + 256 + 64 * j + lane;
                    nx[j] = k < b ? v[(st.base + k) & rmask] : 0.0f;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t c = k0 + 64 * j;
                if (c < b) {
                    const uint32_t cnt = b - c;
                    if (cnt >= 64) {
#pragma unroll
                        for (int i = 0; i < 64; ++i) acc = acc + __builtin_bit_cast(float, __builtin_amdgcn_readlane(x[j], i));
                    } else {
                        for (uint32_t i = 0; i < cnt; ++i) acc = acc + __builtin_bit_cast(float, __builtin_amdgcn_readlane(x[j], (int)i));
                    }
                }
            }
        }
        if (lane == 0) {
            const float mean = acc / (float)(b - a);
            const uint32_t cnt = (uint32_t)roundf((float)(b - a) / (float)q.spb);
            runinfo[r] = (cnt << 1) | (mean > 0.0f ? 1u : 0u);
        }
    }
    __syncthreads();
    if (lane == 0) {
        uint32_t nbits = 0, cur = 0;
        for (uint32_t r = 0; r < nfl; ++r) {
            const uint32_t bit = runinfo[r] & 1u;
            for (uint32_t k = runinfo[r] >> 1; k; --k) {
                if (nbits >= cap_bits) { overflow = 1; break; }
                cur |= bit << (nbits & 31);
                if ((nbits & 31) == 31) { words[nbits >> 5] = cur; cur = 0; }
                ++nbits;
            }
        }
        if (nbits & 31) words[nbits >> 5] = cur;
        if (flips_dbg)
            for (uint32_t r = 0; r < nfl && r < flips_cap; ++r) flips_dbg[(size_t)s * flips_cap + r] = flips[r];
        // erase the consumed prefix (SymbolExtractor.h:156-157) = advance the ring base
        const uint32_t last = nfl ? flips[nfl - 1] : 0u;
        const uint32_t end = st.base + h;
        st.cached = end - q.R + 1;
        st.base += last;
        st.held = h - last;
        sym[s] = st;
        hdr->nbits = nbits; hdr->held_after = st.held; hdr->nflips = nfl; hdr->overflow = overflow; hdr->uncached = end - st.cached;
    }
}

void launch_symbols(hipStream_t st, uint32_t n_streams, uint32_t max_m, uint32_t max_new, uint32_t max_R, const float* tail,
                    uint32_t ring_cap, SymState* sym, unsigned long long* flipmask, float* weight, const SymbolParams* sp,
                    const StreamCall* call, uint32_t* slots, uint32_t slot_words, uint32_t* flips_dbg, uint32_t flips_cap)
{
    if (max_m) {
        dim3 g2((max_new + kAvgSpan - 1) / kAvgSpan, n_streams);
        const size_t lds = (size_t)(kAvgSpan + 2 * max_R + 16) * sizeof(float);
        hipLaunchKernelGGL(k_sym_avg, g2, dim3(kAvgLanes), lds, st, tail, ring_cap, sym, flipmask, weight, sp, call);
    }
    hipLaunchKernelGGL(k_sym_scan, dim3(n_streams), dim3(64), 0, st, tail, ring_cap, sym, flipmask, weight, sp, call, slots,
                       slot_words, flips_dbg, flips_cap);
}

}  // namespace hd
