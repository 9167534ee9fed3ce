// Where does an LDS-DMA instruction put its bytes for LDS addresses beyond 64 KiB, with part of the lanes masked off, 16 and 4 bytes wide?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/glds_test.hip -o tools/micro/glds_test && tools/micro/glds_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>

__device__ __forceinline__ void glds16(const void* base, uint32_t off, uint32_t lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(base), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4(const void* base, uint32_t off, uint32_t lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(base), "s"(lds_dst) : "memory");
}

__global__ __launch_bounds__(256) void k(const uint32_t* src, uint32_t* dump, uint32_t dst, int mode, uint32_t nwords)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    uint32_t* w = reinterpret_cast<uint32_t*>(lds);
    for (uint32_t i = threadIdx.x; i < nwords; i += 256) w[i] = 0xDEAD0000u | (i & 0xFFFF);
    __syncthreads();
    if (threadIdx.x < 64) {
        const uint32_t lane = threadIdx.x;
        if (mode == 0) glds16(src, lane * 16u, dst);
        else if (mode == 1) { if (lane < 55) glds16(src, lane * 16u, dst); }
        else if (mode == 2) glds4(src, lane * 4u + 4u, dst);
        else if (mode == 3) glds16(src, (63u - lane) * 16u, dst);     // permuted source
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nwords; i += 256) dump[i] = w[i];
}

int main()
{
    const uint32_t lds_bytes = 160 * 1024, nwords = lds_bytes / 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    std::vector<uint32_t> hs(4096);
    for (uint32_t i = 0; i < hs.size(); ++i) hs[i] = 0xA0000000u | i;
    uint32_t *src, *dump;
    hipMalloc(&src, hs.size() * 4); hipMalloc(&dump, lds_bytes);
    hipMemcpy(src, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    std::vector<uint32_t> hd(nwords);
    for (int mode = 0; mode < 4; ++mode)
        for (uint32_t dst : {0u, 4096u, 60000u - 60000u % 16u, 65536u, 70000u - 70000u % 16u, 100000u - 100000u % 16u, 150000u - 150000u % 16u, 57936u, 57936u + 17u * 1024u}) {
            k<<<1, 256, lds_bytes>>>(src, dump, dst, mode, nwords);
            hipError_t e = hipDeviceSynchronize();
            hipMemcpy(hd.data(), dump, lds_bytes, hipMemcpyDeviceToHost);
            // find the changed words
            uint32_t first = 0xFFFFFFFFu, last = 0, cnt = 0;
            for (uint32_t i = 0; i < nwords; ++i) if (hd[i] != (0xDEAD0000u | (i & 0xFFFF))) { if (first == 0xFFFFFFFFu) first = i; last = i; ++cnt; }
            printf("mode %d dst %6u: err %d changed %4u words, bytes [%u, %u]  first value %08x  word@dst %08x %08x %08x %08x\n", mode, dst, (int)e, cnt,
                   first == 0xFFFFFFFFu ? 0 : first * 4, last * 4 + 3, first == 0xFFFFFFFFu ? 0 : hd[first], hd[dst / 4], hd[dst / 4 + 1], hd[dst / 4 + 4], hd[dst / 4 + 5]);
        }
    return 0;
}
