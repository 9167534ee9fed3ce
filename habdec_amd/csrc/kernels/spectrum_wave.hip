// The spectrum step as a launch of its own: one wave per stream (spectrum_wave.h has the body and the algorithm).
#include <hip/hip_runtime.h>

#include "spectrum_wave.h"

namespace hd {

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_spectrum_wave(const float2* __restrict__ fft_in, const float2* __restrict__ tw4096, float2* __restrict__ spec,
                                                        float* __restrict__ power, SpectrumStatsDev* __restrict__ stats,
                                                        const StreamCall* __restrict__ call, double rate, int bins_sep, const uint32_t seq,
                                                        const float2* __restrict__ chunk, const size_t chunk_stride, const uint32_t fir_hist_cap)
{
    __shared__ float plane[64 * 65];
    const uint32_t s = blockIdx.x;
    const uint32_t run = call[s].fft_run;
    if (!run) return;
    // fft_run == 2: this call's decimated chunk alone fills the buffer (Decoder.h:467-473 with an empty freq_in_ and >= 4096 new samples): its head is read where
    // the last decimation stage left it -- behind the low-pass history and the pending samples -- instead of from a second copy in fft_in
    const float2* x = run == 2u ? chunk + (size_t)s * chunk_stride + fir_hist_cap + call[s].pend_before : fft_in + (size_t)s * kFftBins;
    spectrum_wave_body(x, tw4096, spec, power, stats, s, rate, bins_sep, plane, seq);
}

void launch_spectrum_wave(hipStream_t st, uint32_t n_streams, const float2* fft_in, const float2* tw4096, float2* spec, float* power,
                          SpectrumStatsDev* stats, const StreamCall* call, double rate, int bins_sep, uint32_t seq, const float2* chunk, size_t chunk_stride, uint32_t fir_hist_cap)
{
    hipLaunchKernelGGL(k_spectrum_wave, dim3(n_streams), dim3(64), 0, st, fft_in, tw4096, spec, power, stats, call, rate, bins_sep, seq, chunk, chunk_stride, fir_hist_cap);
}

}  // namespace hd
