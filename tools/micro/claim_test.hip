// Stand-alone check of the per-XCD ticket counters used by the step kernel's stage-1 workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(64) void k(unsigned* ctr, unsigned* ctr_next, unsigned runs_per_xcd, unsigned* done_runs, unsigned* per_wg, unsigned* xcc_seen)
{
    extern __shared__ unsigned char lds[];
    unsigned xcd = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u;
    unsigned* my = ctr + xcd * 32, * nx = ctr_next + xcd * 32;
    unsigned n = 0;
    for (;;) {
        unsigned r = 0; if (threadIdx.x == 0) r = __hip_atomic_fetch_add(my, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r = (unsigned)__builtin_amdgcn_readfirstlane((int)r);
        if (r == runs_per_xcd && threadIdx.x == 0) (void)__hip_atomic_exchange(nx, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (r >= runs_per_xcd) break;
        ++n;
        unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < 200) { }     // 2 us of "work"
        if (threadIdx.x == 0) atomicAdd(&done_runs[xcd * 1024 + r], 1u);
    }
    if (threadIdx.x == 0) { per_wg[blockIdx.x] = n; atomicAdd(&xcc_seen[xcd], 1u); }
}
int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    unsigned* ctr; CK(hipMalloc(&ctr, 2 * 16 * 32 * 4)); CK(hipMemset(ctr, 0, 2 * 16 * 32 * 4));
    unsigned *done_runs, *per_wg, *seen; CK(hipMalloc(&done_runs, 16 * 1024 * 4)); CK(hipMalloc(&per_wg, 4096 * 4)); CK(hipMalloc(&seen, 64));
    for (int launch = 0; launch < 6; ++launch) {
        CK(hipMemset(done_runs, 0, 16 * 1024 * 4)); CK(hipMemset(per_wg, 0, 4096 * 4)); CK(hipMemset(seen, 0, 64));
        hipLaunchKernelGGL(k, dim3(2048), dim3(64), 19 * 1024, 0, ctr + (launch & 1) * 512, ctr + ((launch & 1) ^ 1) * 512, 1024u, done_runs, per_wg, seen);
        CK(hipDeviceSynchronize());
        std::vector<unsigned> d(16 * 1024), w(4096), s(16); std::vector<unsigned> c(1024);
        CK(hipMemcpy(d.data(), done_runs, 16 * 1024 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(w.data(), per_wg, 4096 * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(s.data(), seen, 64, hipMemcpyDeviceToHost)); CK(hipMemcpy(c.data(), ctr, 1024 * 4, hipMemcpyDeviceToHost));
        unsigned ok = 0, twice = 0; for (int i = 0; i < 8 * 1024; ++i) { ok += d[i] == 1; twice += d[i] > 1; }
        unsigned mx = 0, wg0 = 0; for (int i = 0; i < 2048; ++i) { mx = w[i] > mx ? w[i] : mx; wg0 += w[i] == 0; }
        printf("launch %d: runs done once %u / 8192, more than once %u; max runs per WG %u, idle WGs %u; WGs per XCD:", launch, ok, twice, mx, wg0);
        for (int x = 0; x < 8; ++x) printf(" %u", s[x]);
        printf("; counters after: this set"); for (int x = 0; x < 8; ++x) printf(" %u", c[(launch & 1) * 512 + x * 32]);
        printf(" | other set"); for (int x = 0; x < 8; ++x) printf(" %u", c[((launch & 1) ^ 1) * 512 + x * 32]);
        printf("\n");
    }
    return 0;
}
