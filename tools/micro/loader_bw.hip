// What one (or two, or four) loader wave(s) per CU can stream into LDS with LDS-DMA, nothing else running: GB/s over the chip by
// source address pattern (contiguous 1 KiB per instruction / the padded-row pattern of stage1_ring.h), tiles in flight per wave and loaders per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/loader_bw.hip -o tools/micro/loader_bw && tools/micro/loader_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ void glds16_x17(const void* base, const uint32_t (&off)[17], uint32_t lds_dst)
{
    unsigned keep, scc_keep;
#define HD_G1(n) "global_load_lds_dwordx4 %" #n ", %19\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
    asm volatile("s_cselect_b32 %1, 1, 0\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %20\n\ts_nop 0\n\t"
                 HD_G1(2) HD_G1(3) HD_G1(4) HD_G1(5) HD_G1(6) HD_G1(7) HD_G1(8) HD_G1(9) HD_G1(10) HD_G1(11) HD_G1(12) HD_G1(13) HD_G1(14) HD_G1(15) HD_G1(16) HD_G1(17)
                 "global_load_lds_dwordx4 %18, %19\n\ts_mov_b32 m0, %0\n\ts_cmp_lg_u32 %1, 0"
                 : "=&s"(keep), "=&s"(scc_keep)
                 : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "v"(off[5]), "v"(off[6]), "v"(off[7]), "v"(off[8]), "v"(off[9]),
                   "v"(off[10]), "v"(off[11]), "v"(off[12]), "v"(off[13]), "v"(off[14]), "v"(off[15]), "v"(off[16]), "s"(base), "s"(lds_dst)
                 : "memory");
#undef HD_G1
}

// pattern 0: contiguous (instruction i: bytes [1024 i, 1024 i + 1024) of the tile's 17 KiB); 1: padded rows (17 chunks per 256-byte row, the 17th a repeat)
template <int DEPTH>
__global__ __launch_bounds__(256) void k(const unsigned char* src, size_t tile_stride, uint32_t tiles_per_wave, int pattern, uint32_t* sink, unsigned long long* cyc)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
    uint32_t off[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) {
        const uint32_t P = 64u * i + lane;
        if (pattern == 0) off[i] = P * 16u;
        else { const uint32_t row = P / 17u, col = P - row * 17u; off[i] = row * 256u + (col < 16u ? col : 15u) * 16u; }
    }
    const uint32_t wg_tiles = tiles_per_wave * nw;
    const unsigned char* base0 = src + ((size_t)blockIdx.x * wg_tiles + wave) * tile_stride;
    const uint32_t slot_bytes = 17u * 1024u;
    const uint32_t lds0 = wave * DEPTH * slot_bytes;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t t = 0; t < tiles_per_wave; ++t) {
        const unsigned char* b = base0 + (size_t)t * nw * tile_stride;
        const void* bu = reinterpret_cast<const void*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)b >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint64_t)b));
        glds16_x17(bu, off, lds0 + (t % DEPTH) * slot_bytes);
        if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(34)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (threadIdx.x == 0) { sink[blockIdx.x] = reinterpret_cast<uint32_t*>(lds)[5]; cyc[blockIdx.x] = t1 - t0; }
}

template <int DEPTH>
void run(const unsigned char* src, size_t bytes, int waves, int pattern, uint32_t* sink, unsigned long long* cyc)
{
    const size_t tile_stride = 16384;                     // a tile = 16 KiB of new samples (the instructions read 17 x 60 or 64 chunks from it)
    const uint32_t n_tiles = (uint32_t)(bytes / tile_stride) - 2;
    const uint32_t tiles_per_wave = n_tiles / (256 * waves);
    const size_t lds = (size_t)waves * DEPTH * 17 * 1024;
    if (lds > 160 * 1024) return;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<DEPTH><<<256, 64 * waves, 160 * 1024 - 256>>>(src, tile_stride, tiles_per_wave, pattern, sink, cyc);    // all of a CU's LDS: one workgroup per CU
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<DEPTH><<<256, 64 * waves, 160 * 1024 - 256>>>(src, tile_stride, tiles_per_wave, pattern, sink, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double moved = (double)tiles_per_wave * 256 * waves * 16384.0;
    printf("loaders/CU %d  in flight/wave %d  pattern %s : %.1f us, %.2f TB/s (%.1f B/clk/CU at 2.3 GHz)\n", waves, DEPTH, pattern ? "padded rows" : "contiguous ", ms * 1e3,
           moved / (ms * 1e-3) / 1e12, moved / 256 / (ms * 1e-3 * 2.3e9));
}

int main()
{
    const size_t bytes = 512ull << 20;
    unsigned char* src; uint32_t* sink; unsigned long long* cyc;
    hipMalloc(&src, bytes + (1 << 20)); hipMemset(src, 1, bytes + (1 << 20)); hipMalloc(&sink, 4096); hipMalloc(&cyc, 4096);
    for (int pattern = 0; pattern < 2; ++pattern)
        for (int waves : {1, 2, 4}) {
            run<1>(src, bytes, waves, pattern, sink, cyc);
            run<2>(src, bytes, waves, pattern, sink, cyc);
            run<3>(src, bytes, waves, pattern, sink, cyc);
        }
    return 0;
}
