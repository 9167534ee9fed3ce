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
// The reference's cost is O(n*R) sequential.  Here the two window means of every candidate position are pure
// functions of the backlog, so they are evaluated for all positions in parallel (k_sym_avg: one lane per
// position, each lane adding its R samples left-to-right from LDS exactly like std::accumulate, so the float
// results are bit-identical), reduced to one "signs differ" bit (ballot -> 64-bit masks) and one weight per
// position.  The inherently sequential edge search then runs on those masks with one wave per stream
// (k_sym_scan: 4096 positions per step through ballots and find-first-set), the per-run sums are taken by
// separate lanes in element order, and lane 0 packs the bits for the host.
#include <hip/hip_runtime.h>

#include "launch.h"

namespace hd {

constexpr int kAvgLanes = 256;
constexpr uint32_t kMaxFlipsPerCall = 2048;   // LDS flip list of the scan kernel (overflow is flagged)

__device__ __forceinline__ uint32_t backlog_after_push(uint32_t held, uint32_t m)
{
    return (held > kVentLimit ? 0u : held) + m;          // SymbolExtractor.h:116-124 (vent happens before the append)
}

__global__ void k_sym_append(const float* __restrict__ demod, size_t demod_stride, float* __restrict__ tail, uint32_t tail_cap,
                             const uint32_t* __restrict__ held, const StreamCall* __restrict__ call)
{
    const uint32_t s = blockIdx.y;
    const uint32_t m = call[s].fir_m;
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const uint32_t h0 = held[s] > kVentLimit ? 0u : held[s];
    if (h0 + j < tail_cap) tail[(size_t)s * tail_cap + h0 + j] = demod[(size_t)s * demod_stride + j];
}

__device__ __forceinline__ int sgnf(float v) { return (0.0f < v) - (v < 0.0f); }

// One lane per candidate position i: left/right window means, sign-difference flag, flip weight.
__global__ __launch_bounds__(kAvgLanes) void k_sym_avg(const float* __restrict__ tail, uint32_t tail_cap,
                                                        const uint32_t* __restrict__ held, unsigned long long* __restrict__ flipmask,
                                                        float* __restrict__ weight, const SymbolParams* __restrict__ sp,
                                                        const StreamCall* __restrict__ call)
{
    extern __shared__ float win[];                         // tail[c0 - R, c0 + 256 + R)
    const uint32_t s = blockIdx.y;
    const uint32_t m = call[s].fir_m;
    if (!m) return;
    const SymbolParams q = sp[s];
    const uint32_t h = backlog_after_push(held[s], m);
    if (h < q.min_held || h < q.spb) return;
    const uint32_t limit = h - q.spb;                      // candidates are i in [R, limit)
    const uint32_t c0 = blockIdx.x * kAvgLanes;            // absolute position of lane 0 (mask words are 64-aligned)
    if (c0 >= limit || c0 + kAvgLanes <= q.R) return;
    const uint32_t R = q.R;
    const float* v = tail + (size_t)s * tail_cap;
    const long w0 = (long)c0 - (long)R;                    // absolute index of win[0]
    const uint32_t wn = kAvgLanes + 2 * R;
    for (uint32_t k = threadIdx.x; k < wn; k += kAvgLanes) {
        const long a = w0 + (long)k;
        win[k] = (a >= 0 && a < (long)h) ? v[a] : 0.0f;
    }
    __syncthreads();
    const uint32_t i = c0 + threadIdx.x;
    bool differ = false;
    float wgt = 0.0f;
    if (i >= R && i < limit) {
        const float* pl = win + threadIdx.x;               // win index of absolute i-R
        float sl = 0.0f, sr = 0.0f;
        for (uint32_t k = 0; k < R; ++k) sl = sl + pl[k];
        const uint32_t rn = min(i + R, h) - i;             // right window is clamped at the end of the backlog
        const float* pr = pl + R;
        for (uint32_t k = 0; k < rn; ++k) sr = sr + pr[k];
        const float al = sl / (float)R;                    // left window never clamps: i >= R
        const float ar = sr / (float)rn;
        differ = sgnf(al) != sgnf(ar);
        const float d = ar - al;
        wgt = q.float_abs ? __builtin_fabsf(d) : (float)abs((int)d);
        weight[(size_t)s * tail_cap + i] = wgt;
    }
    const unsigned long long mask = __ballot(differ);
    if ((threadIdx.x & 63) == 0) flipmask[(size_t)s * (tail_cap / 64) + (i >> 6)] = mask;
}

// first position in [from, to) whose flag equals `want`, or 0xFFFFFFFF.  One 64-bit mask word per lane per step.
__device__ __forceinline__ uint32_t find_flag(const unsigned long long* __restrict__ masks, uint32_t from, uint32_t to, bool want)
{
    const uint32_t lane = threadIdx.x;
    for (uint32_t base = from & ~63u; base < to; base += 64 * 64) {
        const uint32_t wbase = base + lane * 64;           // first position covered by this lane's word
        unsigned long long w = 0;
        if (wbase < to) {
            w = masks[wbase >> 6];
            if (!want) w = ~w;
            if (wbase < from) w &= ~0ull << (from - wbase);             // from - wbase < 64 here
            if (to - wbase < 64) w &= (1ull << (to - wbase)) - 1ull;
        }
        const unsigned long long hit = __ballot(w != 0ull);
        if (hit) {
            const int src = __ffsll((long long)hit) - 1;
            const unsigned long long ww = __shfl(w, src, 64);
            return base + (uint32_t)src * 64 + (uint32_t)(__ffsll((long long)ww) - 1);
        }
    }
    return 0xFFFFFFFFu;
}

__global__ __launch_bounds__(64) void k_sym_scan(float* __restrict__ tail, uint32_t tail_cap, uint32_t* __restrict__ held,
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
    const uint32_t h_old = held[s];
    if (!m) {                                               // symbol stage not reached this call
        if (lane == 0) { hdr->nbits = 0; hdr->held_after = h_old; hdr->nflips = 0; hdr->overflow = 0; }
        return;
    }
    const SymbolParams q = sp[s];
    const uint32_t h = backlog_after_push(h_old, m);
    if (h < q.min_held || h < q.spb) {
        if (lane == 0) { held[s] = h; hdr->nbits = 0; hdr->held_after = h; hdr->nflips = 0; hdr->overflow = 0; }
        return;
    }
    float* v = tail + (size_t)s * tail_cap;
    const unsigned long long* masks = flipmask + (size_t)s * (tail_cap / 64);
    const float* wgt = weight + (size_t)s * tail_cap;
    const uint32_t limit = h - q.spb;
    uint32_t pos = q.R, nfl = 0, overflow = 0;
    while (pos < limit) {
        const uint32_t lo = find_flag(masks, pos, limit, true);
        if (lo == 0xFFFFFFFFu) break;
        const uint32_t hi = find_flag(masks, lo + 1, limit, false);
        if (hi == 0xFFFFFFFFu) break;
        // first maximum of the weight over [lo, hi)
        float bw = -1.0f;
        uint32_t bi = 0xFFFFFFFFu;
        for (uint32_t i = lo + lane; i < hi; i += 64) {
            const float w = wgt[i];
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
    // per-run sums, one lane per run, elements added in index order (std::accumulate)
    for (uint32_t r = lane; r < nfl; r += 64) {
        const uint32_t a = r ? flips[r - 1] : 0u, b = flips[r];
        float acc = 0.0f;
        for (uint32_t k = a; k < b; ++k) acc = acc + v[k];
        const float mean = acc / (float)(b - a);
        const uint32_t cnt = (uint32_t)roundf((float)(b - a) / (float)q.spb);
        runinfo[r] = (cnt << 1) | (mean > 0.0f ? 1u : 0u);
    }
    __syncthreads();
    uint32_t nbits = 0;
    if (lane == 0) {
        uint32_t cur = 0;
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
    }
    // erase the consumed prefix (SymbolExtractor.h:156-157): move [last, h) to the front, 64 samples per step;
    // a step's loads complete before its stores and never reach back into an earlier step's destination.
    const uint32_t last = nfl ? flips[nfl - 1] : 0u;
    const uint32_t keep = h - last;
    if (last) {
        for (uint32_t base = 0; base < keep; base += 64) {
            const uint32_t k = base + lane;
            float x = 0.0f;
            if (k < keep) x = v[last + k];
            __syncthreads();
            if (k < keep) v[k] = x;
            __syncthreads();
        }
    }
    if (lane == 0) {
        held[s] = keep;
        hdr->nbits = nbits; hdr->held_after = keep; hdr->nflips = nfl; hdr->overflow = overflow;
    }
}

void launch_symbols(hipStream_t st, uint32_t n_streams, uint32_t max_m, uint32_t max_held, uint32_t max_R, const float* demod,
                    size_t demod_stride, float* tail, uint32_t tail_cap, uint32_t* held, unsigned long long* flipmask, float* weight,
                    const SymbolParams* sp, const StreamCall* call, uint32_t* slots, uint32_t slot_words, uint32_t* flips_dbg,
                    uint32_t flips_cap)
{
    if (max_m) {
        dim3 g1((max_m + 255) / 256, n_streams);
        hipLaunchKernelGGL(k_sym_append, g1, dim3(256), 0, st, demod, demod_stride, tail, tail_cap, held, call);
        const uint32_t span = max_held > tail_cap ? tail_cap : max_held;
        dim3 g2((span + kAvgLanes - 1) / kAvgLanes, n_streams);
        const size_t lds = (size_t)(kAvgLanes + 2 * max_R) * sizeof(float);
        hipLaunchKernelGGL(k_sym_avg, g2, dim3(kAvgLanes), lds, st, tail, tail_cap, held, flipmask, weight, sp, call);
    }
    hipLaunchKernelGGL(k_sym_scan, dim3(n_streams), dim3(64), 0, st, tail, tail_cap, held, flipmask, weight, sp, call, slots,
                       slot_words, flips_dbg, flips_cap);
}

}  // namespace hd
