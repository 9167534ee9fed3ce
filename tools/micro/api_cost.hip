// Microbenchmark: host-side cost of the HIP runtime calls the engine issues per step (MI355X box, ROCm 7.2).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_empty(int* p) { if (p && threadIdx.x == 12345) *p = 1; }
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main()
{
    hipStream_t q, q2; CK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&q2, hipStreamNonBlocking));
    hipEvent_t ev, evt; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); CK(hipEventCreate(&evt));
    int* d; CK(hipMalloc(&d, 1 << 20)); int* h; CK(hipHostMalloc(&h, 1 << 20, hipHostMallocDefault));
    const int N = 2000;
    for (int w = 0; w < 2; ++w) {
        double t0 = now_us();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(1024, 4), dim3(256), 0, q, d);
        double t1 = now_us(); CK(hipStreamSynchronize(q)); double t2 = now_us();
        if (w) printf("kernel launch (4096 WGs, async):      %7.2f us/call   (+ drain %.0f us)\n", (t1 - t0) / N, t2 - t1);
        t0 = now_us(); for (int i = 0; i < N; ++i) CK(hipEventRecord(ev, q)); t1 = now_us(); CK(hipStreamSynchronize(q));
        if (w) printf("hipEventRecord (no timing):           %7.2f us/call\n", (t1 - t0) / N);
        t0 = now_us(); for (int i = 0; i < N; ++i) CK(hipEventRecord(evt, q)); t1 = now_us(); CK(hipStreamSynchronize(q));
        if (w) printf("hipEventRecord (timing):              %7.2f us/call\n", (t1 - t0) / N);
        t0 = now_us(); for (int i = 0; i < N; ++i) { CK(hipEventRecord(ev, q)); CK(hipStreamWaitEvent(q2, ev, 0)); } t1 = now_us(); CK(hipDeviceSynchronize());
        if (w) printf("record + cross-stream wait:           %7.2f us/pair\n", (t1 - t0) / N);
        t0 = now_us(); for (int i = 0; i < N; ++i) CK(hipMemcpyAsync(d, h, 64 * 1024, hipMemcpyHostToDevice, q)); t1 = now_us(); CK(hipStreamSynchronize(q));
        if (w) printf("hipMemcpyAsync H2D 64 KiB pinned:     %7.2f us/call\n", (t1 - t0) / N);
        t0 = now_us(); for (int i = 0; i < N; ++i) CK(hipMemcpyAsync(h, d, 48 * 1024, hipMemcpyDeviceToHost, q)); t1 = now_us(); CK(hipStreamSynchronize(q));
        if (w) printf("hipMemcpyAsync D2H 48 KiB pinned:     %7.2f us/call\n", (t1 - t0) / N);
        t0 = now_us(); for (int i = 0; i < 200; ++i) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, q, d); CK(hipStreamSynchronize(q)); } t1 = now_us();
        if (w) printf("launch + hipStreamSynchronize:        %7.2f us/round trip\n", (t1 - t0) / 200);
    }
    // graph of 10 kernels + H2D + D2H
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(q, hipStreamCaptureModeThreadLocal));
    CK(hipMemcpyAsync(d, h, 64 * 1024, hipMemcpyHostToDevice, q));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_empty, dim3(1024, 4), dim3(256), 0, q, d);
    CK(hipMemcpyAsync(h, d, 48 * 1024, hipMemcpyDeviceToHost, q));
    CK(hipStreamEndCapture(q, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 2; ++w) {
        double t0 = now_us(); for (int i = 0; i < 500; ++i) CK(hipGraphLaunch(ge, q)); double t1 = now_us(); CK(hipStreamSynchronize(q)); double t2 = now_us();
        if (w) printf("hipGraphLaunch (H2D + 10 kernels + D2H): %6.2f us/call host, %.2f us/graph end to end\n", (t1 - t0) / 500, (t2 - t0) / 500);
    }
    {   double t0 = now_us();
        for (int i = 0; i < 500; ++i) { CK(hipMemcpyAsync(d, h, 64 * 1024, hipMemcpyHostToDevice, q)); for (int j = 0; j < 10; ++j) hipLaunchKernelGGL(k_empty, dim3(1024, 4), dim3(256), 0, q, d); CK(hipMemcpyAsync(h, d, 48 * 1024, hipMemcpyDeviceToHost, q)); }
        double t1 = now_us(); CK(hipStreamSynchronize(q)); double t2 = now_us();
        printf("same sequence eager:                     %6.2f us/seq host, %.2f us/seq end to end\n", (t1 - t0) / 500, (t2 - t0) / 500);
    }
    return 0;
}
