// Is a line that XCD a holds clean in its L2 refreshed when XCD b rewrites it in an earlier kernel of the same queue?  (Does a kernel boundary make one
// XCD's stores visible to another XCD's later loads even when the reader still caches the old line?)  Three launches per round on one stream:
//   R1: the workgroups on XCD `a` read buffer X (and keep it in their L2);  W: the workgroups on XCD `b` overwrite X with the round number;
//   R2: XCD `a` reads X again and counts words that do not carry the round number.
// Build: hipcc --offload-arch=gfx950 -O2 tools/micro/xcd_coherence.hip -o tools/micro/xcd_coherence ; run: tools/micro/xcd_coherence [words=1024] [rounds=200]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ unsigned xcc() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u; }

__global__ void k_read(const unsigned* __restrict__ x, unsigned n, unsigned xcd, unsigned want, unsigned* bad, unsigned* sink)
{
    if (xcc() != xcd) return;
    unsigned acc = 0, nb = 0;
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x) { const unsigned v = x[i]; acc += v; if (v != want) ++nb; }
    if (nb) atomicAdd(bad, nb);
    if (acc == 0xdeadbeefu) *sink = acc;
}
__global__ void k_write(unsigned* x, unsigned n, unsigned xcd, unsigned val)
{
    if (xcc() != xcd) return;
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x) x[i] = val;
}

int main(int argc, char** argv)
{
    const unsigned n = argc > 1 ? atoi(argv[1]) : 1024, rounds = argc > 2 ? atoi(argv[2]) : 200;
    unsigned *x, *bad, *sink;
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&bad, 8)); CK(hipMalloc(&sink, 4));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (unsigned a = 0; a < 2; ++a) for (unsigned b = 0; b < 8; b += 3) {
        CK(hipMemsetAsync(x, 0, n * 4, st)); CK(hipMemsetAsync(bad, 0, 8, st));
        for (unsigned r = 1; r <= rounds; ++r) {
            // one workgroup per XCD is enough: 8 workgroups of a small grid land on the 8 XCDs round-robin; 64 to be sure every XCD gets some
            hipLaunchKernelGGL(k_read, dim3(64), dim3(256), 0, st, x, n, a, r - 1, bad + 1, sink);     // (bad[1]: the first read sees the previous round's value)
            hipLaunchKernelGGL(k_write, dim3(64), dim3(256), 0, st, x, n, b, r);
            hipLaunchKernelGGL(k_read, dim3(64), dim3(256), 0, st, x, n, a, r, bad, sink);
        }
        unsigned h[2]; CK(hipMemcpyAsync(h, bad, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
        printf("reader XCD %u, writer XCD %u, %u words, %u rounds: stale words seen after the write %u (before it: %u)\n", a, b, n, rounds, h[0], h[1]);
    }
    return 0;
}
