// Does s_getreg_b32(HW_REG_IB_STS) show the wave's outstanding vector-memory count (vmcnt) on gfx950, and in which bits?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float4* src, uint32_t* out, float4* sink)
{
    f4 r[40];
    const uint32_t before = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 7);
    uint32_t mid[5];
#pragma unroll
    for (int i = 0; i < 40; ++i) {
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r[i]) : "v"((uint32_t)(threadIdx.x * 16 + i * 65536)), "s"(src));
        if (i % 8 == 7) mid[i / 8] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 7);
    }
    const uint32_t after = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 7);
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    const uint32_t at10 = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 7);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t at0 = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 7);
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 40; ++i) { f4 t = r[i]; asm volatile("" : "+v"(t)); acc.x += t.x; }
    sink[threadIdx.x] = acc;
    if (threadIdx.x == 0) { out[0] = before; for (int i = 0; i < 5; ++i) out[1 + i] = mid[i]; out[6] = after; out[7] = at10; out[8] = at0; }
}
int main()
{
    float4 *src, *sink; uint32_t* out;
    hipMalloc(&src, 64 << 20); hipMemset(src, 0, 64 << 20); hipMalloc(&sink, 4096); hipMalloc(&out, 64);
    k<<<1, 64>>>(src, out, sink);
    uint32_t h[9]; hipMemcpy(h, out, 36, hipMemcpyDeviceToHost);
    const char* names[9] = {"before", "after 8", "after 16", "after 24", "after 32", "after 40", "after all", "vmcnt(10)", "vmcnt(0)"};
    for (int i = 0; i < 9; ++i) printf("%-10s IB_STS %08x  vm_cnt lo %u hi %u -> %u\n", names[i], h[i], h[i] & 15, (h[i] >> 22) & 3, (h[i] & 15) | (((h[i] >> 22) & 3) << 4));
    return 0;
}
