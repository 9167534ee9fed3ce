// Microbenchmark: achievable HBM read bandwidth on MI355X for the access shapes the stage-1 decimator could use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// (a) grid-stride float4 stream, U loads in flight per lane
template <int U>
__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ in, size_t n4, float* out)
{
    float acc = 0.f;
    size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (; i + 256 * (U - 1) < n4; i += stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = in[i + 256 * u];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 12345.678f) out[0] = acc;
}
// (b) one workgroup per contiguous 32 KiB tile (like the decimator), all loads issued up front, `tiles` tiles per WG with prefetch
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
    const size_t bytes = 512ull << 20;      // one push slab of the bench (1024 streams x 65536 cf32)
    const int nbuf = 4;                      // rotate over 2 GiB so nothing stays in the 256 MiB Infinity Cache
    float4* d; CK(hipMalloc(&d, bytes * nbuf)); CK(hipMemset(d, 1, bytes * nbuf));
    float* o; CK(hipMalloc(&o, 64));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const size_t n4 = bytes / 16;
    auto run = [&](const char* name, auto launch) {
        float best = 1e9f, sum = 0; const int reps = 20;
        for (int r = 0; r < reps + 3; ++r) {
            const float4* src = d + (size_t)(r % nbuf) * n4;
            hipEventRecord(a); launch(src); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (r >= 3) { best = ms < best ? ms : best; sum += ms; }
        }
        printf("%-58s avg %7.1f us  best %7.1f us  -> %6.0f GB/s (avg)\n", name, sum / reps * 1e3, best * 1e3, bytes / (sum / reps * 1e-3) / 1e9);
        return 0;
    };
    run("stream float4, 2048 WGs x 256, 4 loads in flight", [&](const float4* s) { hipLaunchKernelGGL((k_stream<4>), dim3(2048), dim3(256), 0, 0, s, n4, o); });
    run("stream float4, 2048 WGs x 256, 8 loads in flight", [&](const float4* s) { hipLaunchKernelGGL((k_stream<8>), dim3(2048), dim3(256), 0, 0, s, n4, o); });
    run("stream float4, 4096 WGs x 256, 8 loads in flight", [&](const float4* s) { hipLaunchKernelGGL((k_stream<8>), dim3(4096), dim3(256), 0, 0, s, n4, o); });
    run("stream float4, 8192 WGs x 256, 16 loads in flight", [&](const float4* s) { hipLaunchKernelGGL((k_stream<16>), dim3(8192), dim3(256), 0, 0, s, n4, o); });
    run("tiles 32 KiB, 128 thr x 16 loads, 1 tile/WG (16384 WGs)", [&](const float4* s) { hipLaunchKernelGGL((k_tiles<128, 16>), dim3(16384), dim3(128), 0, 0, s, 1, o); });
    run("tiles 32 KiB, 128 thr x 16 loads, 8 tiles/WG prefetch (2048)", [&](const float4* s) { hipLaunchKernelGGL((k_tiles<128, 16>), dim3(2048), dim3(128), 0, 0, s, 8, o); });
    run("tiles 32 KiB, 128 thr x 16 loads, 16 tiles/WG prefetch (1024)", [&](const float4* s) { hipLaunchKernelGGL((k_tiles<128, 16>), dim3(1024), dim3(128), 0, 0, s, 16, o); });
    run("tiles 16 KiB, 64 thr x 16 loads, 8 tiles/WG prefetch (4096)", [&](const float4* s) { hipLaunchKernelGGL((k_tiles<64, 16>), dim3(4096), dim3(64), 0, 0, s, 8, o); });
    run("tiles 32 KiB, 256 thr x 8 loads, 8 tiles/WG prefetch (2048)", [&](const float4* s) { hipLaunchKernelGGL((k_tiles<256, 8>), dim3(2048), dim3(256), 0, 0, s, 8, o); });
    // occupancy-limited like the stage-1 decimator: 19 KiB (or more) of dynamic LDS per single-wave workgroup
    for (int lds_kb : {19, 38, 9}) {
        char nm[128];
        snprintf(nm, sizeof nm, "tiles 16 KiB, 64 thr x 16 loads, 16 tiles/WG, %d KiB LDS/WG", lds_kb);
        run(nm, [&](const float4* s) { hipLaunchKernelGGL((k_tiles<64, 16>), dim3(2048), dim3(64), lds_kb * 1024, 0, s, 16, o); });
        snprintf(nm, sizeof nm, "tiles 32 KiB, 64 thr x 32 loads, 8 tiles/WG, %d KiB LDS/WG", lds_kb);
        run(nm, [&](const float4* s) { hipLaunchKernelGGL((k_tiles<64, 32>), dim3(2048), dim3(64), lds_kb * 1024, 0, s, 8, o); });
    }
    return 0;
}
