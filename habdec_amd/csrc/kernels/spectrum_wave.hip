// The spectrum step as a launch of its own: one wave per stream (spectrum_wave.h has the body and the algorithm).
#include <hip/hip_runtime.h>

#include "spectrum_wave.h"

namespace hd {

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_spectrum_wave(const float2* __restrict__ fft_in, const float2* __restrict__ tw4096, float2* __restrict__ spec,
                                                        float* __restrict__ power, SpectrumStatsDev* __restrict__ stats,
                                                        const StreamCall* __restrict__ call, double rate, int bins_sep, const uint32_t seq)
{
    __shared__ float plane[64 * 65];
    const uint32_t s = blockIdx.x;
    if (!call[s].fft_run) return;
    spectrum_wave_body(fft_in, tw4096, spec, power, stats, s, rate, bins_sep, plane, seq);
}

void launch_spectrum_wave(hipStream_t st, uint32_t n_streams, const float2* fft_in, const float2* tw4096, float2* spec, float* power,
                          SpectrumStatsDev* stats, const StreamCall* call, double rate, int bins_sep, uint32_t seq)
{
    hipLaunchKernelGGL(k_spectrum_wave, dim3(n_streams), dim3(64), 0, st, fft_in, tw4096, spec, power, stats, call, rate, bins_sep, seq);
}

}  // namespace hd
