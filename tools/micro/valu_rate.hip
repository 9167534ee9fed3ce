// Issue cost of packed / plain f32 VALU instructions from ONE wave (and from 2 / 4 waves per SIMD): cycles per instruction.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o tools/micro/valu_rate && tools/micro/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, float seed, int iters)
{
    f32x2 a0 = {seed, seed + 1}, a1 = {seed + 2, seed + 3}, a2 = {seed + 4, seed + 5}, a3 = {seed + 6, seed + 7};
    f32x2 x0 = {1.0f + seed, 1.00001f}, x1 = {0.5f, seed}, x2 = {seed, 2.f}, x3 = {3.f, seed};
    f32x2 k = {0.99f, 1.01f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {          // packed: 4 independent mul + 4 independent (chain) adds
                f32x2 p0, p1, p2, p3;
                asm volatile("v_pk_mul_f32 %0, %4, %8\n v_pk_mul_f32 %1, %5, %8\n v_pk_mul_f32 %2, %6, %8\n v_pk_mul_f32 %3, %7, %8"
                             : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(k));
                asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_add_f32 %2, %2, %6\n v_pk_add_f32 %3, %3, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3));
            } else if (MODE == 1) {   // plain: 8 independent mul + 8 chain adds (same flops)
                float p[8];
                asm volatile("v_mul_f32 %0, %8, %16\n v_mul_f32 %1, %9, %16\n v_mul_f32 %2, %10, %16\n v_mul_f32 %3, %11, %16\n"
                             "v_mul_f32 %4, %12, %16\n v_mul_f32 %5, %13, %16\n v_mul_f32 %6, %14, %16\n v_mul_f32 %7, %15, %16"
                             : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5]), "=&v"(p[6]), "=&v"(p[7])
                             : "v"(x0.x), "v"(x0.y), "v"(x1.x), "v"(x1.y), "v"(x2.x), "v"(x2.y), "v"(x3.x), "v"(x3.y), "v"(k.x));
                asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_add_f32 %2, %2, %10\n v_add_f32 %3, %3, %11\n"
                             "v_add_f32 %4, %4, %12\n v_add_f32 %5, %5, %13\n v_add_f32 %6, %6, %14\n v_add_f32 %7, %7, %15"
                             : "+v"(a0.x), "+v"(a0.y), "+v"(a1.x), "+v"(a1.y), "+v"(a2.x), "+v"(a2.y), "+v"(a3.x), "+v"(a3.y)
                             : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]));
            } else if (MODE == 2) {   // packed, ONE dependent chain: mul -> add -> mul -> add on the same registers
                f32x2 p0;
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=&v"(p0) : "v"(x0), "v"(k));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a0) : "v"(p0));
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=&v"(p0) : "v"(x1), "v"(k));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a0) : "v"(p0));
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=&v"(p0) : "v"(x2), "v"(k));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a0) : "v"(p0));
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=&v"(p0) : "v"(x3), "v"(k));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a0) : "v"(p0));
            } else if (MODE == 3) {   // plain dependent chain of adds only
                asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %2\n v_add_f32 %0, %0, %3\n v_add_f32 %0, %0, %4\n"
                             "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %2\n v_add_f32 %0, %0, %3\n v_add_f32 %0, %0, %4"
                             : "+v"(a0.x) : "v"(x0.x), "v"(x1.x), "v"(x2.x), "v"(x3.x));
            } else if (MODE == 6) {   // packed dependent chain of adds only
                asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %0, %0, %3\n v_pk_add_f32 %0, %0, %4\n"
                             "v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %0, %0, %3\n v_pk_add_f32 %0, %0, %4"
                             : "+v"(a0) : "v"(x0), "v"(x1), "v"(x2), "v"(x3));
            } else if (MODE == 7) {   // one packed chain, products two taps ahead: mul(i+2) between add(i) and add(i+1)
                f32x2 p0, p1;
                asm volatile("v_pk_mul_f32 %0, %2, %4\n v_pk_mul_f32 %1, %3, %4" : "=&v"(p0), "=&v"(p1) : "v"(x0), "v"(x1), "v"(k));
                asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_mul_f32 %1, %3, %5\n v_pk_add_f32 %0, %0, %2\n v_pk_mul_f32 %2, %4, %5\n"
                             "v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %2"
                             : "+v"(a0), "+v"(p0), "+v"(p1) : "v"(x2), "v"(x3), "v"(k));
            } else if (MODE == 4) {   // packed fma, independent
                asm volatile("v_pk_fma_f32 %0, %4, %8, %0\n v_pk_fma_f32 %1, %5, %8, %1\n v_pk_fma_f32 %2, %6, %8, %2\n v_pk_fma_f32 %3, %7, %8, %3\n"
                             "v_pk_fma_f32 %0, %5, %8, %0\n v_pk_fma_f32 %1, %6, %8, %1\n v_pk_fma_f32 %2, %7, %8, %2\n v_pk_fma_f32 %3, %4, %8, %3"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(k));
            } else {                  // plain fma, independent
                asm volatile("v_fma_f32 %0, %8, %12, %0\n v_fma_f32 %1, %9, %12, %1\n v_fma_f32 %2, %10, %12, %2\n v_fma_f32 %3, %11, %12, %3\n"
                             "v_fma_f32 %4, %8, %12, %4\n v_fma_f32 %5, %9, %12, %5\n v_fma_f32 %6, %10, %12, %6\n v_fma_f32 %7, %11, %12, %7"
                             : "+v"(a0.x), "+v"(a0.y), "+v"(a1.x), "+v"(a1.y), "+v"(a2.x), "+v"(a2.y), "+v"(a3.x), "+v"(a3.y)
                             : "v"(x0.x), "v"(x1.x), "v"(x2.x), "v"(x3.x), "v"(k.x));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (a0.x + a1.x + a2.x + a3.x + a0.y + a1.y + a2.y + a3.y == 12345.f) out[1] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int MODE>
void run(const char* name, int per_iter_instrs)
{
    unsigned long long* d; hipMalloc(&d, 16);
    for (int threads : {64, 256, 512, 1024}) {    // 1 wave on one SIMD; 1, 2, 4 waves per SIMD (one workgroup per CU)
        const int iters = 2000;
        (void)hipMemset(d, 0, 16);
        k<MODE><<<256, threads>>>(d, 0.001f, iters); (void)hipDeviceSynchronize();
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, 0);
        k<MODE><<<256, threads>>>(d, 0.001f, iters);
        (void)hipEventRecord(e1, 0); (void)hipDeviceSynchronize();
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        const hipError_t err = hipGetLastError();
        unsigned long long c; (void)hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost);
        // wall clock: instructions one SIMD executed = iters * 8 * per_iter * (waves per SIMD)
        const double wps = threads >= 256 ? threads / 256.0 : 1.0;
        printf("%-44s %4d threads/CU: %6.2f s_memtime ticks per instruction per wave | kernel %.3f ms -> %.2f ns per instruction per SIMD (%s)\n", name, threads,
               (double)c / ((double)iters * 8 * per_iter_instrs), ms, ms * 1e6 / ((double)iters * 8 * per_iter_instrs * wps), hipGetErrorString(err));
    }
    hipFree(d);
}
int main()
{
    run<0>("v_pk_mul_f32 + v_pk_add_f32, 4 chains", 8);
    run<1>("v_mul_f32 + v_add_f32, 8 chains", 16);
    run<2>("v_pk_mul -> v_pk_add, one dependent chain", 8);
    run<3>("v_add_f32 dependent chain", 8);
    run<6>("v_pk_add_f32 dependent chain", 8);
    run<7>("one pk chain, products 2 ahead (4 taps: 8 instr)", 8);
    run<4>("v_pk_fma_f32 independent", 8);
    run<5>("v_fma_f32 independent", 8);
    return 0;
}
