// Microbenchmark: which compute units a CU-masked queue gets (per XCD), and what read bandwidth single-wave tile walkers reach on such a subset.
// Question behind it: can stage 1 keep HBM saturated from HALF of every XCD's CUs, leaving the other half to the stream tails?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(64) void k_where(unsigned* out, int spin)
{
    extern __shared__ unsigned char lds[];
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);    // XCC_ID[3:0]
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) { }   // stay resident so that the grid spreads out
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
    if (spin < 0) lds[threadIdx.x] = 1;
}

template <int TPB, int ITER>
__global__ __launch_bounds__(TPB) void k_tiles(const float4* __restrict__ in, int tiles_per_wg, float* out)
{
    float acc = 0.f;
    const size_t tile4 = (size_t)TPB * ITER;
    const float4* p = in + ((size_t)blockIdx.x * tiles_per_wg) * tile4 + threadIdx.x;
    float4 r[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) r[it] = p[(size_t)it * TPB];
    for (int t = 0; t < tiles_per_wg; ++t) {
        float4 c[ITER];
#pragma unroll
        for (int it = 0; it < ITER; ++it) c[it] = r[it];
        if (t + 1 < tiles_per_wg) {
            p += tile4;
#pragma unroll
            for (int it = 0; it < ITER; ++it) r[it] = p[(size_t)it * TPB];
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) acc += c[it].x + c[it].y + c[it].z + c[it].w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("CUs: %d\n", ncu);
    const size_t bytes = 512ull << 20; const int nbuf = 4;
    float4* d; CK(hipMalloc(&d, bytes * nbuf)); CK(hipMemset(d, 1, bytes * nbuf));
    float* o; CK(hipMalloc(&o, 64));
    unsigned* w; CK(hipMalloc(&w, 8192 * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    struct Pat { const char* name; std::function<int(int)> f; };
    Pat pats[] = {
        {"all", [](int i) { return 1; }},
        {"even bits", [](int i) { return (i % 2) == 0; }},
        {"low half", [](int i) { return i < 128; }},
        {"(i/8)%2==0", [](int i) { return ((i / 8) % 2) == 0; }},
        {"(i/16)%2==0", [](int i) { return ((i / 16) % 2) == 0; }},
        {"(i/32)%2==0", [](int i) { return ((i / 32) % 2) == 0; }},
        {"(i/64)%2==0", [](int i) { return ((i / 64) % 2) == 0; }},
        {"i%4<3 (three quarters)", [](int i) { return (i % 4) < 3; }},
        {"(i/8)%4<3 (three quarters)", [](int i) { return ((i / 8) % 4) < 3; }},
    };
    for (const Pat& p : pats) {
        std::vector<uint32_t> mask((ncu + 31) / 32, 0);
        int bits = 0;
        for (int i = 0; i < ncu; ++i) if (p.f(i)) { mask[i / 32] |= 1u << (i % 32); ++bits; }
        hipStream_t st; CK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
        const int nwg = 4096;
        CK(hipMemsetAsync(w, 0, nwg * 8, st));
        hipLaunchKernelGGL(k_where, dim3(nwg), dim3(64), 19 * 1024, st, w, 2000);   // 20 us each
        CK(hipStreamSynchronize(st));
        std::vector<unsigned> h(nwg * 2); CK(hipMemcpy(h.data(), w, nwg * 8, hipMemcpyDeviceToHost));
        std::map<unsigned, std::set<unsigned>> per;    // xcc -> set of (se, sh, cu)
        for (int i = 0; i < nwg; ++i) { const unsigned hw = h[2 * i], x = h[2 * i + 1] & 15; per[x].insert((hw >> 8) & 0xff); }
        int total = 0; for (auto& kv : per) total += (int)kv.second.size();
        printf("%-28s bits %3d -> CUs used %3d; per XCC:", p.name, bits, total);
        for (auto& kv : per) printf(" %u:%zu", kv.first, kv.second.size());
        printf("\n");
        // read bandwidth of stage-1-shaped tile walkers on this subset: 8 single-wave workgroups per CU, 19 KiB LDS each
        for (int per_cu : {8}) {
            const int wgs = total * per_cu; const int tiles = (int)(bytes / 16384 / wgs);
            float sum = 0; const int reps = 12;
            for (int r = 0; r < reps + 3; ++r) {
                const float4* src = d + (size_t)(r % nbuf) * (bytes / 16);
                hipEventRecord(a, st);
                hipLaunchKernelGGL((k_tiles<64, 16>), dim3(wgs), dim3(64), 19 * 1024, st, src, tiles, o);
                hipEventRecord(b, st); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (r >= 3) sum += ms;
            }
            const double moved = (double)wgs * tiles * 16384;
            printf("    tile walkers: %d WGs x %d tiles of 16 KiB: %7.1f us -> %6.0f GB/s\n", wgs, tiles, sum / reps * 1e3, moved / (sum / reps * 1e-3) / 1e9);
        }
        CK(hipStreamDestroy(st));
    }
    printf("alignment of the tile rows (full chip / half of every XCD), 16 sweeps of 1 KiB per tile:\n");
    for (int half = 0; half < 2; ++half) {
        std::vector<uint32_t> mask(8, 0);
        for (int i = 0; i < 256; ++i) if (!half || ((i / 8) % 2) == 0) mask[i / 32] |= 1u << (i % 32);
        hipStream_t st; CK(hipExtStreamCreateWithCUMask(&st, 8, mask.data()));
        const int cus = half ? 128 : 256, wgs = cus * 8, tiles = (int)(bytes / 16384 / wgs) - 1;
        for (int off : {0, 16, 32, 64, 96}) {
            float sum = 0; const int reps = 12;
            for (int r = 0; r < reps + 3; ++r) {
                const float4* src = d + (size_t)(r % nbuf) * (bytes / 16) + off / 16;
                hipEventRecord(a, st);
                hipLaunchKernelGGL((k_tiles<64, 16>), dim3(wgs), dim3(64), 19 * 1024, st, src, tiles, o);
                hipEventRecord(b, st); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (r >= 3) sum += ms;
            }
            const double moved = (double)wgs * tiles * 16384;
            printf("    %3d CUs, base + %2d B: %7.1f us -> %6.0f GB/s\n", cus, off, sum / reps * 1e3, moved / (sum / reps * 1e-3) / 1e9);
        }
        CK(hipStreamDestroy(st));
    }
    return 0;
}
