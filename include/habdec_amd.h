/*
 * habdec_amd.h -- C ABI of the MI355X-native RTTY demodulation engine (libhabdec_amd.so).
 *
 * This is the drop-in boundary underneath a source-compatible `habdec::Decoder<T>`
 * (habdec_amd/include/habdec/Decoder.h).  The reference has no FFI for this path -- its boundary is the
 * public surface of the header-only class template `habdec::Decoder<TReal>` (reference
 * code/Decoder/Decoder.h:65-141) -- so every entry point below names the reference member(s) it stands in
 * for.  Plain C types only: no C++, no torch, no HIP types cross this line.
 *
 * Model: one *engine* per GPU owns S independent *streams* (S = 1 for the websocket server's single
 * decoder, thousands for batch decoding).  All streams of an engine share the input sampling rate and
 * the decimation plan; baud / framing / low-pass settings are per stream.  One `hd_process_*` call is one
 * `pushSamples()+operator()()` round (reference code/websocketServer/main.cpp:240-245) for every stream.
 *
 * Threading: an engine is driven by ONE thread at a time for hd_process_* (like the reference's
 * DECODER_THREAD); getters/setters may be called from other threads and are serialised by an internal
 * mutex (reference Decoder.h:209,229,240,423).  Callbacks fire synchronously inside hd_process_*, in
 * stream order, on the calling thread (reference Decoder.h:604-606,625-626).
 *
 * Errors: every function returning `int` returns HD_OK (0) or a negative HD_ERR_*; hd_last_error() gives
 * the message of the last failure on the calling thread.  The reference itself reports problems by printing
 * and carrying on (Decoder.h:272-276, FirFilter.h:121-137); conditions it tolerates are tolerated here too,
 * conditions that are undefined behaviour there (an input shorter than a stage's history, Decimator.h:140-143)
 * are rejected with HD_ERR_UNSUPPORTED.
 */
#ifndef HABDEC_AMD_H
#define HABDEC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HD_OK 0
#define HD_ERR_INVALID (-1)      /* bad argument */
#define HD_ERR_DEVICE (-2)       /* HIP / rocFFT failure, or no gfx950 device */
#define HD_ERR_UNSUPPORTED (-3)  /* input the reference has undefined behaviour on */
#define HD_ERR_CAPACITY (-4)     /* more samples than the engine was sized for */

#define HD_FFT_BINS 4096         /* reference Decoder.h:162 fft_bins_cnt_ */

typedef struct hd_engine hd_engine;

typedef struct hd_engine_config {
    int32_t  device;            /* HIP device ordinal */
    uint32_t n_streams;         /* S >= 1 */
    uint32_t max_chunk;         /* most IQ samples one stream hands to one hd_process_* call (65536 in main.cpp:235) */
    double   sampling_rate;     /* IQVector::samplingRate() latched by the first pushSamples (Decoder.h:215-216) */
    uint32_t decimation;        /* total factor 1,2,4,...,256 = setupDecimationStagesFactor(2^dec) (Decoder.h:268-332) */
    /* per-stream defaults; reference defaults in websocketServer/GLOBALS.h:87-100 and Decoder.h:172-173 */
    double   baud;              /* Decoder::baud()           */
    uint32_t rtty_bits;         /* Decoder::rtty_bits()  7|8 */
    float    rtty_stops;        /* Decoder::rtty_stops() 1|2 */
    float    lowpass_bw_hz;     /* Decoder::lowpass_bw()     */
    float    lowpass_trans;     /* Decoder::lowpass_trans()  */
    int32_t  dc_remove;         /* Decoder::dc_remove()      */
    /* How the reference's unqualified sin/cos/abs calls resolved in the translation unit that compiled it
     * (DESIGN.md "lookup context"): 1 = <math.h> context (float trig in the tap design, float |.| in the
     * flip-point weights; default), 0 = <cmath>-only context (double trig, integer abs). */
    int32_t  lookup_mode;
    int32_t  enable_spectrum;   /* 1 = run the 4096-bin spectrum + AFC like the reference, 0 = skip (stage benchmarks) */
    /* 0 = keep the reference's decode gate: above 160 kHz decimated rate only decimation/FFT/AFC run
     * (Decoder.h:522-527).  1 = stage-level mode: FIR/demod/symbols run at any rate (BASELINE config 3). */
    int32_t  ungated;
    int32_t  keep_filtered;     /* 1 = also store the FIR output so hd_stream_filtered() works (parity tests) */
    /* 0 = hd_process_* returns after THIS call's text has been delivered (what Decoder::operator() does).
     * 1 = pipelined batch mode: a call enqueues its GPU work and delivers the PREVIOUS call's text, so the host
     *     text stage and the next call's decimation overlap the current call's symbol kernels; hd_flush() drains.
     * 2 = the same with one more call in flight where the launches of consecutive calls are strictly ordered on one queue (equally
     *     sized pushes through the step kernel): text arrives two calls late, and a host thread that is held up for the length of a
     *     launch does not leave the GPU idle.  Elsewhere 2 behaves like 1. */
    int32_t  pipeline;
    /* Arithmetic of the FIR sums (decimator stages Decimator.h:128-138, low-pass FirFilter.h:155-161).
     * 0 = HD_ARITH_EXACT (default): separately rounded multiply and add in ascending tap order, one accumulator per output -- every decimated, filtered
     *     and demodulated float is bit-identical to the reference's CPU arithmetic.
     * 1 = HD_ARITH_FAST: fused multiply-add (half the vector instructions of every FIR in the chain).  Intermediate floats agree with the exact mode to
     *     <= 1e-5 of the signal's peak (measured <= 3e-7); the discriminator and the symbol extractor are the exact mode's code, and on every stream of
     *     the parity suite the decoded characters and sentences are identical to the reference's.  Not bit-identical floats: choose it for throughput. */
    int32_t  arith;
} hd_engine_config;
#define HD_ARITH_EXACT 0
#define HD_ARITH_FAST 1

/* Fill `cfg` with the reference defaults (dec 64, 300 baud 8N2, low-pass 1500 Hz / 0.025, spectrum on). */
void hd_engine_config_default(hd_engine_config* cfg);

/* Decoder construction + setupDecimationStagesFactor (Decoder.h:268-332) for S streams on one GPU. */
int  hd_engine_create(const hd_engine_config* cfg, hd_engine** out);
void hd_engine_destroy(hd_engine* e);
const char* hd_last_error(void);
/* Smallest non-empty per-stream sample count hd_process_* accepts at a decimation factor (a multiple of the factor that covers every stage's
 * history: shorter inputs are undefined behaviour in the reference's Decimator.h:140-143); 0 = unsupported factor.  A caller that drains a queue
 * (Decoder::process, Decoder.h:426-436) leaves fewer samples queued until more arrive. */
uint32_t hd_min_chunk(uint32_t decimation);
uint32_t hd_engine_streams(const hd_engine* e);
/* getDecimationFactor / getDecimatedSamplingRate (Decoder.h:716-735) */
uint32_t hd_engine_decimation(const hd_engine* e);
double   hd_engine_decimated_rate(const hd_engine* e);

/* ---- per-stream control plane: the reference setters of the same names (Decoder.h:238-257, 655-712) ---- */
int hd_stream_set_baud(hd_engine* e, uint32_t stream, double baud);
int hd_stream_set_rtty(hd_engine* e, uint32_t stream, uint32_t bits, float stops);
int hd_stream_set_lowpass_bw(hd_engine* e, uint32_t stream, float hz);
int hd_stream_set_lowpass_trans(hd_engine* e, uint32_t stream, float trans);
int hd_stream_set_dc_remove(hd_engine* e, uint32_t stream, int on);
/* resetFrequencyCorrection (Decoder.h:806-809 -> AFC.h:187-194) */
int hd_stream_reset_frequency_correction(hd_engine* e, uint32_t stream, double correction);

/* ---- callbacks: Decoder::sentence_callback_ / character_callback_ (Decoder.h:135-138) ---- */
typedef void (*hd_sentence_cb)(void* user, uint32_t stream, const char* callsign, const char* data, const char* crc);
typedef void (*hd_chars_cb)(void* user, uint32_t stream, const char* chars, size_t n);
void hd_set_sentence_callback(hd_engine* e, hd_sentence_cb cb, void* user);   /* fires only on CRC match */
/* every sentence the scan finds, CRC-valid or not, in the order found and before the sentence callback of the same sentence: the point at which
 * the reference prints "<sentence> OK|ERR" and its running tally (Decoder.h:601 -> print_habhub_sentence.cpp:33-62) */
typedef void (*hd_match_cb)(void* user, uint32_t stream, const char* callsign, const char* data, const char* crc, int crc_ok);
void hd_set_match_callback(hd_engine* e, hd_match_cb cb, void* user);
void hd_set_chars_callback(hd_engine* e, hd_chars_cb cb, void* user);         /* printable chars of this call */

/* ---- data path: pushSamples() + operator()() (Decoder.h:206-219, 416-638) for all S streams ----
 * Stream s reads `n` cf32 samples (interleaved I,Q float32: the IQSource_File layout, IQSource_File.h:156-157)
 * starting at `iq + 2*s*stream_stride` floats.  `n_per_stream` (S entries) overrides the uniform `n` when not
 * NULL.  Every n must be <= max_chunk and a multiple of the decimation factor (the Decoder facade keeps the
 * remainder queued on the host exactly like Decoder.h:429-435).  Returns after the decoded text of this
 * call has been delivered (callbacks fired, getters updated). */
int hd_process_host(hd_engine* e, const float* iq, size_t stream_stride, const uint32_t* n_per_stream, uint32_t n);
/* Page-locked host memory the GPU addresses in place.  An IQ buffer that lives in such memory and is handed to hd_process_host of a SYNCHRONOUS engine
 * (pipeline = 0: the call returns when its kernels are done) is read by the first decimation stage straight over PCIe -- no staging copy inside the call
 * (one 65536-sample push of one stream: 512 KiB in ~10 us instead of a ~40 us pageable copy in front of the kernels).  The Decoder facade keeps its input
 * queue (the reference's iq_in_buffer_, Decoder.h:206-219) in it.  NULL when no HIP device is present or the allocation fails: use ordinary memory then. */
void* hd_pinned_alloc(size_t bytes);
void  hd_pinned_free(void* p);
/* Same, but `d_iq` is DEVICE memory on the engine's GPU (HBM-resident batches; base 16-byte aligned,
 * stream_stride even). */
int hd_process_device(hd_engine* e, const void* d_iq, size_t stream_stride, const uint32_t* n_per_stream, uint32_t n);
/* Pipelined mode: wait for the call in flight and deliver its results (no-op otherwise).  Getters that read
 * device buffers flush implicitly. */
int hd_flush(hd_engine* e);

/* Batched file ingest: pump `src` (habdec_amd_host.h, hd_host_iqfiles_*; one file per stream, opened
 * with chunk <= max_chunk and granule = the engine's decimation factor) through the engine until the files are exhausted
 * or `max_rounds` rounds were pushed.  A reader thread fills pinned push slabs (four, so that a slab is never refilled
 * while a pipelined call may still copy from it) while the calling thread runs hd_process_host on the previous one --
 * file reads, the H2D copy and the GPU work overlap.  Text is delivered as by hd_process_host; ends with hd_flush.
 * `samples_done` (optional) receives the IQ samples consumed over all streams. */
struct hd_host_iqfiles;
int hd_ingest_run(hd_engine* e, struct hd_host_iqfiles* src, uint64_t max_rounds, uint64_t* samples_done);

/* ---- results: getRTTY / getLastSentence (Decoder.h:641-652) and the callback streams ---- */
size_t hd_stream_rtty(hd_engine* e, uint32_t stream, char* buf, size_t cap);
size_t hd_stream_last_sentence(hd_engine* e, uint32_t stream, char* buf, size_t cap);
/* drain the log of CRC-valid sentences ("callsign,data*crc\n" each) / of all regex matches / of printable chars */
size_t hd_stream_take_sentences(hd_engine* e, uint32_t stream, char* buf, size_t cap);
size_t hd_stream_take_matches(hd_engine* e, uint32_t stream, char* buf, size_t cap);
size_t hd_stream_take_chars(hd_engine* e, uint32_t stream, char* buf, size_t cap);
uint64_t hd_engine_sentences_ok(const hd_engine* e);   /* CRC-valid sentences over all streams so far */

/* ---- GUI data: getFFT / getDemodulated / getPowerSpectrum / getPeaks / getNoiseFloor / getShift /
 *      getFrequencyCorrection (Decoder.h:751-803) ---- */
typedef struct hd_afc_info {
    double frequency_correction, shift_hz, noise_floor, noise_variance;
    int32_t peak_left, peak_right;      /* sign encodes validity like AFC::getPeaks (AFC.h:146-162) */
    uint64_t spectra;                    /* number of 4096-bin spectra computed so far */
} hd_afc_info;
int    hd_stream_afc(hd_engine* e, uint32_t stream, hd_afc_info* out);
size_t hd_stream_spectrum(hd_engine* e, uint32_t stream, float* iq, size_t cap_complex);   /* freq_out_, fftshifted */
size_t hd_stream_power(hd_engine* e, uint32_t stream, float* p, size_t cap);
size_t hd_stream_demodulated(hd_engine* e, uint32_t stream, float* v, size_t cap);         /* last call */

/* ---- parity taps (not in the reference API; the intermediates its process() holds in members) ---- */
size_t hd_stream_decimated(hd_engine* e, uint32_t stream, float* iq, size_t cap_complex);  /* iq_samples_temp_ of the last call */
size_t hd_stream_filtered(hd_engine* e, uint32_t stream, float* iq, size_t cap_complex);   /* iq_samples_filtered_ (keep_filtered=1) */
size_t hd_stream_bits(hd_engine* e, uint32_t stream, uint8_t* bits, size_t cap);           /* symbols of the last call */
size_t hd_stream_flips(hd_engine* e, uint32_t stream, uint32_t* flips, size_t cap);        /* flip points of the last call */
size_t hd_stream_fir_taps(hd_engine* e, uint32_t stream, float* taps, size_t cap);
uint32_t hd_stream_symbol_backlog(hd_engine* e, uint32_t stream);                          /* samples held by the symbol extractor */
/* Checksum of the discriminator output (getDemodulated(), Decoder.h:131) of the call DELIVERED last for this stream, without flushing the
 * pipeline: ck[0] = sum of the samples' bit patterns, ck[1] = sum of (i + 1) * bit pattern, mod 2^32, over n samples; *call_index counts
 * hd_process_* calls from 0.  n = 0xFFFFFFFF when the launch path that served the call does not compute it (the fused back end k_backend; the
 * stream tail and the separate kernels do). */
int hd_stream_demod_checksum(hd_engine* e, uint32_t stream, uint64_t* call_index, uint32_t* n, uint32_t ck[2]);
/* ... and every call delivered so far folded into one word, so that a free-running batch can be compared with a CPU run call by call without reading a
 * sample: hash = fold over the delivered calls, in order, of (n, ck[0], ck[1]) with h = (h ^ x) * 0x100000001B3 starting from 0xCBF29CE484222325 (FNV-1a over
 * 32-bit words); *calls = calls folded in, *calls_without = delivered calls whose launch path left no checksum (not folded in). */
int hd_stream_demod_checksum_total(hd_engine* e, uint32_t stream, uint64_t* calls, uint64_t* calls_without, uint64_t* hash);
uint64_t hd_stream_bits_total(hd_engine* e, uint32_t stream);                              /* symbols produced since the engine was created (delivered calls) */
uint64_t hd_stream_flip_list_full(hd_engine* e, uint32_t stream);                          /* delivered calls in which the device's flip list (512 flip points per call) filled up: the symbol search
                                                                                             * stopped there and went on in the next call -- the same bits as SymbolExtractor::operator() (SymbolExtractor.h:129-158,
                                                                                             * which has no such bound), delivered a call later */

/* ---- measurement ---- */
typedef struct hd_timing {
    double ms_total;        /* HIP-event time of the whole kernel sequence of the last hd_process_* call */
    double ms_front;        /* of the kernel that touches full-rate IQ: the first-stage decimator, or the step kernel that contains it (see path) */
    uint64_t front_bytes;   /* algorithmic bytes of that launch: 8 B per input sample + 8 B per output sample */
    uint64_t samples;       /* input samples consumed by the last call over all streams */
    /* host side of the most recent hd_process_* call, microseconds */
    double host_enqueue_us; /* size bookkeeping + uploads + kernel launches */
    double host_wait_us;    /* blocked on the GPU for the results being delivered */
    double host_text_us;    /* AFC state machines, RTTY framing, sentence scan, callbacks */
    uint64_t timed_calls;   /* how many calls carried the HIP-event timing so far (ms_* are those of the latest one) */
    uint32_t path;          /* how the most recent call was launched: 0 separate kernels, 1 fused back end (k_backend), 2 stream tail kernel
                             * (k_tail), 3 step kernel (k_step: stage 1 + the previous call's stream tails in one launch; ms_front is ITS duration) */
    uint32_t step_variant;  /* 1 = the kernel that touches full-rate IQ is one workgroup per CU with LDS-DMA loader waves (stage1_ring.h): k_step_cu on path 3,
                             * k_stage1_cu (stage 1 alone) on paths 0-2; 0 = single-wave / classic workgroups (k_step, k_decimate) */
    uint64_t host_calls_in_place;   /* hd_process_host calls so far whose IQ was read in place from page-locked memory (hd_pinned_alloc): no staging copy */
    uint64_t lowpass_fft_calls;     /* calls so far whose low-pass ran through transforms (fast mode, >= 1024 taps: configs[4]'s 4097-tap filter) */
} hd_timing;
int hd_engine_timing(hd_engine* e, hd_timing* out);
/* Bracket the kernels with HIP events on every `every`-th call (default 8; 0 = never; 1 = every call).  Each event record is
 * a barrier packet that costs a few microseconds of queue time, so per-call timing slows a pipelined batch by ~8 %;
 * hd_engine_timing() reports the most recent timed call. */
void hd_engine_set_timing(hd_engine* e, int every);

#ifdef __cplusplus
}
#endif
#endif /* HABDEC_AMD_H */
