// Cycles per 64-output stage-1 tile (/32, 212 taps) for ONE wave's tap loop, by where the taps come from and how the loop is pipelined.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/taploop.hip -o tools/micro/taploop && tools/micro/taploop
// Layout = the ring layout of the loader/consumer stage 1: rows of 32 samples at a pitch of 34 (16-byte pad), lane o's window starts in
// row o at sample 13 (slot s of the lane <-> row s/32, column s%32; taps sit on slots [13, 225)).
//   MODE 0: taps through the scalar cache (s_load), 16-slot chunks, next chunk's reads requested before the current one is summed
//   MODE 1: taps from an LDS table with broadcast ds_read_b128 (in-order returns: counted waits), same chunking
//   MODE 2: MODE 1 on two tiles at once (two independent sums per lane, 2 x 18 KB of LDS)
//   MODE 3: MODE 0 with the sixteen-tap chunk scheduled by hand (ring_mac16_asm of stage1_ring.h): products three taps ahead of the adds
//   MODE 4: the SYSTOLIC loop (built into stage1_ring.h's ring_worker in round 5): a lane reads only ITS row (16 x 16 bytes instead of 106), the accumulators travel -- output o
//           starts in lane o and after the taps that fall on row o + r moves to the next lane with one DPP rotate per component; same products in the same
//           order.  (Timing only here: the wrap at lane 63 is not handled -- a real version computes 57 outputs per 64 rows.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int T = 212, JS = 13, NS = JS + T, NCH = (NS + 15) / 16;   // 15 chunks: chunk 0 holds taps on slots 13..15, chunk 14 on slot 224 only
constexpr int ROWP = 34;                                            // samples per row incl. pad
constexpr int TILE_ROWS = 64 + 7;
constexpr int TILE_BYTES = TILE_ROWS * ROWP * 8;

template <int J0, int J1>
__device__ __forceinline__ void mac16(f32x2& acc, const f32x4 (&x)[8], const float (&k)[16])
{
#pragma unroll
    for (int j = J0; j < J1; ++j) {
        const f32x2 s = (j & 1) ? x[j >> 1].zw : x[j >> 1].xy;
        acc = acc + s * k[j];
    }
}
template <int J0, int J1>
__device__ __forceinline__ void mac16v(f32x2& acc, const f32x4 (&x)[8], const f32x4 (&k)[4])
{
#pragma unroll
    for (int j = J0; j < J1; ++j) {
        const f32x2 s = (j & 1) ? x[j >> 1].zw : x[j >> 1].xy;
        const f32x4 kk = k[j >> 2];
        const float kv = (j & 3) == 0 ? kk.x : (j & 3) == 1 ? kk.y : (j & 3) == 2 ? kk.z : kk.w;
        acc = acc + s * kv;
    }
}

// the hand-scheduled chunk of stage1_ring.h (products three taps ahead of the adds), taps as eight scalar pairs
__device__ __forceinline__ void mac16_asm(f32x2& acc, const f32x4 (&x)[8], const f32x2 (&kp)[8])
{
    f32x2 t0, t1, t2, t3;
#define HD_MUL_E(t, xi, ki) "v_pk_mul_f32 %" #t ", %" #xi ", %" #ki " op_sel_hi:[1,0]\n\t"
#define HD_MUL_O(t, xi, ki) "v_pk_mul_f32 %" #t ", %" #xi ", %" #ki " op_sel:[0,1]\n\t"
#define HD_ADD(t) "v_pk_add_f32 %0, %0, %" #t "\n\t"
    asm volatile(
        HD_MUL_E(1, 5, 21) HD_MUL_O(2, 6, 21) HD_MUL_E(3, 7, 22)
        HD_ADD(1) HD_MUL_O(4, 8, 22) HD_ADD(2) HD_MUL_E(1, 9, 23) HD_ADD(3) HD_MUL_O(2, 10, 23) HD_ADD(4) HD_MUL_E(3, 11, 24)
        HD_ADD(1) HD_MUL_O(4, 12, 24) HD_ADD(2) HD_MUL_E(1, 13, 25) HD_ADD(3) HD_MUL_O(2, 14, 25) HD_ADD(4) HD_MUL_E(3, 15, 26)
        HD_ADD(1) HD_MUL_O(4, 16, 26) HD_ADD(2) HD_MUL_E(1, 17, 27) HD_ADD(3) HD_MUL_O(2, 18, 27) HD_ADD(4) HD_MUL_E(3, 19, 28)
        HD_ADD(1) HD_MUL_O(4, 20, 28) HD_ADD(2) HD_ADD(3) "v_pk_add_f32 %0, %0, %4"
        : "+v"(acc), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(x[0].xy), "v"(x[0].zw), "v"(x[1].xy), "v"(x[1].zw), "v"(x[2].xy), "v"(x[2].zw), "v"(x[3].xy), "v"(x[3].zw),
          "v"(x[4].xy), "v"(x[4].zw), "v"(x[5].xy), "v"(x[5].zw), "v"(x[6].xy), "v"(x[6].zw), "v"(x[7].xy), "v"(x[7].zw),
          "s"(kp[0]), "s"(kp[1]), "s"(kp[2]), "s"(kp[3]), "s"(kp[4]), "s"(kp[5]), "s"(kp[6]), "s"(kp[7]));
#undef HD_MUL_E
#undef HD_MUL_O
#undef HD_ADD
}

template <int MODE>
__global__ __launch_bounds__(64) void k(unsigned long long* out, float2* res, const float* __restrict__ taps_g /* NCH*16, zero outside [13,225) */, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    typedef const float __attribute__((address_space(4)))* ctaps_t;
    const ctaps_t taps = (ctaps_t)(uintptr_t)taps_g;
    float2* tile = reinterpret_cast<float2*>(lds);
    constexpr int NT = MODE == 2 ? 2 : 1;
    float* tl = reinterpret_cast<float*>(lds + NT * TILE_BYTES);
    const int lane = threadIdx.x;
    for (int i = lane; i < NT * TILE_ROWS * ROWP; i += 64) tile[i] = make_float2(0.001f * (float)((i * 37 + blockIdx.x) % 1000) - 0.5f, 0.002f * (float)((i * 11) % 500) - 0.5f);
    for (int i = lane; i < NCH * 16; i += 64) tl[i] = taps[i];
    __syncthreads();
    const unsigned char* p = lds + lane * (ROWP * 8);
    auto coff = [](int c) { return (c >> 1) * (ROWP * 8) + (c & 1) * 128; };
    f32x2 total = {0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
            f32x2 acc = {0.f, 0.f};
            f32x4 xa[8], xb[8]; float ka[16], kb[16];
            auto rd = [&](f32x4 (&x)[8], float (&kk)[16], int c) {
                const unsigned char* pc = p + coff(c);
#pragma unroll
                for (int q = 0; q < 8; ++q) x[q] = *reinterpret_cast<const f32x4*>(pc + 16 * q);
#pragma unroll
                for (int j = 0; j < 16; ++j) kk[j] = taps[c * 16 + j];
            };
            rd(xa, ka, 0); rd(xb, kb, 1);
            mac16<13, 16>(acc, xa, ka);
            int c = 1;
#pragma unroll 1
            for (; c + 1 < NCH - 1; c += 2) {
                rd(xa, ka, c + 1);
                mac16<0, 16>(acc, xb, kb);
                rd(xb, kb, c + 2);
                mac16<0, 16>(acc, xa, ka);
            }
            // c == 13: chunk 13 in xb?  (NCH-1 = 14: loop runs c = 1,3,..,11 -> ends with c = 13, chunk 13 in xb)
            rd(xa, ka, 14);
            mac16<0, 16>(acc, xb, kb);
            mac16<0, 1>(acc, xa, ka);
            total = total + acc;
        } else if constexpr (MODE == 3) {
            f32x2 acc = {0.f, 0.f};
            f32x4 xa[8], xb[8]; f32x2 ka[8], kb[8];
            auto rd = [&](f32x4 (&x)[8], f32x2 (&kk)[8], int c) {
                const unsigned char* pc = p + coff(c);
#pragma unroll
                for (int q = 0; q < 8; ++q) x[q] = *reinterpret_cast<const f32x4*>(pc + 16 * q);
#pragma unroll
                for (int j = 0; j < 8; ++j) { kk[j].x = taps[c * 16 + 2 * j]; kk[j].y = taps[c * 16 + 2 * j + 1]; }
            };
            rd(xa, ka, 0); rd(xb, kb, 1);
            acc = acc + xa[6].zw * ka[6].y; acc = acc + xa[7].xy * ka[7].x; acc = acc + xa[7].zw * ka[7].y;
            int c = 1;
#pragma unroll 1
            for (; c + 1 < NCH - 1; c += 2) {
                rd(xa, ka, c + 1);
                mac16_asm(acc, xb, kb);
                rd(xb, kb, c + 2);
                mac16_asm(acc, xa, ka);
            }
            rd(xa, ka, 14);
            mac16_asm(acc, xb, kb);
            acc = acc + xa[0].xy * ka[0].x;
            total = total + acc;
        } else if constexpr (MODE == 4) {
            f32x2 acc = {0.f, 0.f};
            f32x4 xa[8], xb[8]; f32x2 ka[8], kb[8], na[8], nb[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { xa[q] = *reinterpret_cast<const f32x4*>(p + 16 * q); xb[q] = *reinterpret_cast<const f32x4*>(p + 128 + 16 * q); }
            auto ld = [&](f32x2 (&kk)[8], int c) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { kk[j].x = taps[c * 16 + 2 * j]; kk[j].y = taps[c * 16 + 2 * j + 1]; }
            };
            auto rot = [&]() {                                      // lane l takes lane l-1's accumulator (wave_ror:1)
                // (through scalars: __builtin_bit_cast of an ext-vector ELEMENT reads the vector's first element whichever was named -- round 4's version
                // of this loop rotated the real part twice; the timing is the same)
                const float re = acc.x, im = acc.y;
                acc.x = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, re), 0x13C, 0xF, 0xF, false));
                acc.y = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, im), 0x13C, 0xF, 0xF, false));
            };
            ld(ka, 0); ld(kb, 1);
            acc = acc + xa[6].zw * ka[6].y; acc = acc + xa[7].xy * ka[7].x; acc = acc + xa[7].zw * ka[7].y;      // step 0: slots 13..15 of chunk 0
            ld(na, 2); ld(nb, 3);
            mac16_asm(acc, xb, kb);
            rot();
#pragma unroll
            for (int r = 1; r <= 6; ++r) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { ka[j] = na[j]; kb[j] = nb[j]; }
                if (r < 6) { ld(na, 2 * r + 2); ld(nb, 2 * r + 3); } else ld(na, 14);
                mac16_asm(acc, xa, ka);
                mac16_asm(acc, xb, kb);
                rot();
            }
            acc = acc + xa[0].xy * na[0].x;                         // step 7: slot 224
            total = total + acc;
        } else if constexpr (MODE == 1) {
            f32x2 acc = {0.f, 0.f};
            f32x4 xa[8], xb[8]; f32x4 ka[4], kb[4];
            auto rd = [&](f32x4 (&x)[8], f32x4 (&kk)[4], int c) {
                const unsigned char* pc = p + coff(c);
#pragma unroll
                for (int q = 0; q < 8; ++q) x[q] = *reinterpret_cast<const f32x4*>(pc + 16 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) kk[j] = *reinterpret_cast<const f32x4*>(tl + c * 16 + 4 * j);
            };
            rd(xa, ka, 0); rd(xb, kb, 1);
            mac16v<13, 16>(acc, xa, ka);
            int c = 1;
#pragma unroll 1
            for (; c + 1 < NCH - 1; c += 2) {
                rd(xa, ka, c + 1);
                mac16v<0, 16>(acc, xb, kb);
                rd(xb, kb, c + 2);
                mac16v<0, 16>(acc, xa, ka);
            }
            rd(xa, ka, 14);
            mac16v<0, 16>(acc, xb, kb);
            mac16v<0, 1>(acc, xa, ka);
            total = total + acc;
        } else {
            f32x2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
            f32x4 xa[8], xb[8], ya[8], yb[8]; f32x4 ka[4], kb[4];
            auto rd = [&](f32x4 (&x)[8], f32x4 (&y)[8], f32x4 (&kk)[4], int c) {
                const unsigned char* pc = p + coff(c);
#pragma unroll
                for (int q = 0; q < 8; ++q) { x[q] = *reinterpret_cast<const f32x4*>(pc + 16 * q); y[q] = *reinterpret_cast<const f32x4*>(pc + TILE_BYTES + 16 * q); }
#pragma unroll
                for (int j = 0; j < 4; ++j) kk[j] = *reinterpret_cast<const f32x4*>(tl + c * 16 + 4 * j);
            };
            rd(xa, ya, ka, 0); rd(xb, yb, kb, 1);
            mac16v<13, 16>(acc0, xa, ka); mac16v<13, 16>(acc1, ya, ka);
            int c = 1;
#pragma unroll 1
            for (; c + 1 < NCH - 1; c += 2) {
                rd(xa, ya, ka, c + 1);
                mac16v<0, 16>(acc0, xb, kb); mac16v<0, 16>(acc1, yb, kb);
                rd(xb, yb, kb, c + 2);
                mac16v<0, 16>(acc0, xa, ka); mac16v<0, 16>(acc1, ya, ka);
            }
            rd(xa, ya, ka, 14);
            mac16v<0, 16>(acc0, xb, kb); mac16v<0, 16>(acc1, yb, kb);
            mac16v<0, 1>(acc0, xa, ka); mac16v<0, 1>(acc1, ya, ka);
            total = total + acc0 + acc1;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    res[blockIdx.x * 64 + lane] = make_float2(total.x, total.y);
    if (lane == 0) out[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, const float* d_taps)
{
    constexpr int NT = MODE == 2 ? 2 : 1;
    const size_t lds = NT * TILE_BYTES + NCH * 16 * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int per_cu : {1, 4, 8}) {
        if (per_cu * lds > 160 * 1024) continue;
        const int blocks = 256 * per_cu, iters = 200;
        unsigned long long* d; float2* r;
        hipMalloc(&d, blocks * 8); hipMalloc(&r, blocks * 64 * 8);
        k<MODE><<<blocks, 64, lds>>>(d, r, d_taps, 2);
        hipDeviceSynchronize();
        k<MODE><<<blocks, 64, lds>>>(d, r, d_taps, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
        double sum = 0; unsigned long long mx = 0;
        for (auto v : h) { sum += (double)v; mx = v > mx ? v : mx; }
        std::vector<float2> hr(64);
        hipMemcpy(hr.data(), r, 64 * 8, hipMemcpyDeviceToHost);
        printf("%-28s %d waves/CU: %.0f cycles per tile (mean), %.0f (max)   [check %.6f]\n", name, per_cu, sum / blocks / iters / NT, (double)mx / iters / NT, hr[5].x);
        hipFree(d); hipFree(r);
    }
}

int main()
{
    std::vector<float> taps(NCH * 16, 0.f);
    for (int t = 0; t < T; ++t) taps[JS + t] = 0.01f * (float)((t * 7) % 13) - 0.05f;
    float* d_taps; hipMalloc(&d_taps, taps.size() * 4);
    hipMemcpy(d_taps, taps.data(), taps.size() * 4, hipMemcpyHostToDevice);
    run<0>("taps via s_load", d_taps);
    run<1>("taps via LDS broadcast", d_taps);
    run<2>("LDS taps, two tiles per wave", d_taps);
    run<3>("s_load taps, hand-scheduled", d_taps);
    run<4>("systolic (own row only)", d_taps);
    return 0;
}
