/*
 * habdec_oracle.h -- C ABI of the CPU ORACLE for the RTTY demodulation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This library restates, on the CPU and in strict sequential
 * non-FMA float32 arithmetic, what the reference's code/Decoder chain computes.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * (habdec_amd/) never does.
 *
 * Parity pinning (see oracle/README.md and DESIGN.md):
 *   - every stage below is checked bit-for-bit against the reference's own stage classes
 *     compiled from /root/reference by oracle/Makefile into oracle/_ref/libhabdec_ref.so
 *     (tests/test_oracle_vs_ref.py) and against the fixtures in tests/golden/ generated from
 *     that build (tools/gen_golden.py);
 *   - the spectrum FFT is UNPINNED at the FFTW boundary (FFTW3f is not vendored by the
 *     reference and is absent from this image): orc_fft4096 is a double-precision DFT rounded
 *     to float, i.e. the mathematical definition FFTW approximates;
 *   - Decoder::process() sequencing cannot be compiled from the reference without stand-ins for
 *     fftw3.h / ssdv.h, so it is restated here from Decoder.h:416-638 and cross-checked against
 *     a composition of the compiled reference stage classes (oracle/ref_harness.cpp).
 */
#ifndef HABDEC_ORACLE_H
#define HABDEC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- decimation tables (reference filtercoef.h) and stage plan (Decoder.h:268-332) ---- */
/* stage plan for a total factor in {1,2,4,...,256}: returns number of stages (0..2) or -1. */
int orc_decim_plan(int total_factor, int stage_ratio[2], int stage_total_name[2]);
/* coefficient table used for (table named total /N, ratio /R); returns length or 0. */
size_t orc_decim_taps(int total_name, int ratio, const float** taps);

/* ---- one decimation stage (Decimator.h:69-146) ---- */
typedef struct orc_decimator orc_decimator;
orc_decimator* orc_decimator_new(int factor, const float* taps, size_t ntaps);
void orc_decimator_free(orc_decimator*);
/* In-place like the reference caller (Decoder.h:443-444): iq[0..n) cf32 interleaved is overwritten
 * by n/factor outputs.  Returns the output count, or (size_t)-1 when the reference would read
 * before its buffer (n < ntaps-1). */
size_t orc_decimator_run(orc_decimator*, float* iq, size_t n);

/* ---- DC removal (Decoder.h:450-459) ---- */
void orc_dc_remove(float* iq, size_t n);

/* ---- spectrum: 4096-point forward DFT + half swap (FFT.cpp:77-99), double precision inside ---- */
void orc_fft_shifted(const float* in_iq, float* out_iq, size_t n /* power of two */);

/* ---- AFC (AFC.h:92-194, Average.h) ---- */
typedef struct orc_afc orc_afc;
orc_afc* orc_afc_new(void);
void orc_afc_free(orc_afc*);
void orc_afc_set_spectrum(orc_afc*, const float* fft_iq, size_t nbins, double sampling_rate);
double orc_afc_process(orc_afc*);
void orc_afc_reset_correction(orc_afc*, double correction);
size_t orc_afc_power(orc_afc*, const float** p);
void orc_afc_get(orc_afc*, double* correction, double* shift_hz, double* noise_floor,
                 double* noise_var, int* gui_peak_l, int* gui_peak_r);

/* ---- low-pass FIR (FirFilter.h:117-209, habdec_windows.h:27-53) ---- */
typedef struct orc_fir orc_fir;
orc_fir* orc_fir_new(void);
void orc_fir_free(orc_fir*);
void orc_fir_set_input_size(orc_fir*, size_t n);              /* setInput() bookkeeping only */
void orc_fir_design(orc_fir*, float rel_width, float trans);   /* LP_BlackmanHarris */
/* 1 (default) = sin/cos resolve to the float overloads, 0 = to the double ones: see Q9b in the .cpp */
void orc_fir_design_mode(orc_fir*, int float_trig);
size_t orc_fir_taps(orc_fir*, const float** taps);
/* returns 0 ok, 1 "no taps", 2 "more taps than samples" (output untouched, as in the reference) */
int orc_fir_run(orc_fir*, const float* in_iq, size_t n, float* out_iq);

/* ---- FSK polar discriminator (FSK2_Demod.h:29-42), last sample kept per handle ---- */
typedef struct orc_demod orc_demod;
orc_demod* orc_demod_new(void);
void orc_demod_free(orc_demod*);
void orc_demod_run(orc_demod*, const float* iq, size_t n, float* out);

/* ---- symbol extractor (SymbolExtractor.h:108-255) ---- */
typedef struct orc_symex orc_symex;
orc_symex* orc_symex_new(void);
void orc_symex_free(orc_symex*);
void orc_symex_rates(orc_symex*, double sampling_rate, double symbol_rate);
void orc_symex_abs_mode(orc_symex*, int float_abs);   /* 1 default; see Q15b in the .cpp */
void orc_symex_push(orc_symex*, const float* v, size_t n);
void orc_symex_run(orc_symex*);
size_t orc_symex_held(orc_symex*);
size_t orc_symex_get(orc_symex*, uint8_t* bits, size_t cap);       /* drains like get() */
size_t orc_symex_last_flips(orc_symex*, const size_t** flips);    /* flip points of last run */

/* ---- RTTY framing (RTTY.h:59-137) ---- */
typedef struct orc_rtty orc_rtty;
orc_rtty* orc_rtty_new(size_t nbits, float nstops);
void orc_rtty_free(orc_rtty*);
void orc_rtty_push(orc_rtty*, const uint8_t* bits, size_t n);
size_t orc_rtty_run(orc_rtty*);
size_t orc_rtty_get(orc_rtty*, char* chars, size_t cap);
size_t orc_rtty_pending_bits(orc_rtty*);

/* ---- text stage (CRC.cpp:21-47, sentence_extract.cpp:58-98) ---- */
void orc_crc16(const char* s, size_t n, char out4[5]);
/* returns 1 on match; writes callsign/data/crc and the remaining stream (all NUL terminated,
 * caller provides cap >= n+1 each). */
int orc_extract_sentence(const char* stream, size_t n, char* callsign, char* data, char* crc,
                         char* rest, size_t cap);

/* ---- the whole Decoder (Decoder.h:206-638) ---- */
typedef struct orc_decoder orc_decoder;
orc_decoder* orc_decoder_new(void);
void orc_decoder_free(orc_decoder*);
int orc_decoder_setup_factor(orc_decoder*, size_t factor);       /* setupDecimationStagesFactor */
void orc_decoder_baud(orc_decoder*, double baud);
void orc_decoder_rtty(orc_decoder*, size_t bits, float stops);
void orc_decoder_lowpass_bw(orc_decoder*, float hz);
void orc_decoder_lowpass_trans(orc_decoder*, float trans);
void orc_decoder_dc_remove(orc_decoder*, int on);
/* 1 (default) = reference compiled in a <math.h> context, 0 = <cmath>-only context (Q9b/Q15b) */
void orc_decoder_lookup_mode(orc_decoder*, int mathh_context);
void orc_decoder_with_fft(orc_decoder*, int on);   /* 0 = skip spectrum/AFC (timing studies only) */
void orc_decoder_ungated(orc_decoder*, int on);    /* 1 = run FIR/demod/symbols above the 160 kHz gate too (stage-level parity for BASELINE config 3) */
void orc_decoder_push(orc_decoder*, const float* iq, size_t n, double sampling_rate);
void orc_decoder_process(orc_decoder*);
void orc_decoder_reset_correction(orc_decoder*, double correction);
/* results */
size_t orc_decoder_rtty_stream(orc_decoder*, const char** s);       /* getRTTY() */
size_t orc_decoder_last_sentence(orc_decoder*, const char** s);     /* getLastSentence() */
size_t orc_decoder_sentence_log(orc_decoder*, const char** s);      /* every sentence_callback_ firing: "call,data*crc\n" */
size_t orc_decoder_match_log(orc_decoder*, const char** s);         /* every regex match incl. bad CRC */
size_t orc_decoder_chars_log(orc_decoder*, const char** s);         /* all printable chars ever appended */
/* intermediates of the LAST process() call (empty when the stage did not run) */
size_t orc_decoder_last_decimated(orc_decoder*, const float** iq);
size_t orc_decoder_last_filtered(orc_decoder*, const float** iq);
size_t orc_decoder_last_demod(orc_decoder*, const float** v);
size_t orc_decoder_last_bits(orc_decoder*, const uint8_t** bits);
size_t orc_decoder_spectrum(orc_decoder*, const float** iq);        /* freq_out_ */
size_t orc_decoder_power(orc_decoder*, const float** p);
void orc_decoder_afc(orc_decoder*, double* correction, double* shift_hz, double* noise_floor,
                     double* noise_var, int* gui_peak_l, int* gui_peak_r);
size_t orc_decoder_fir_taps(orc_decoder*, const float** taps);
size_t orc_decoder_symex_held(orc_decoder*);
uint64_t orc_decoder_fft_count(orc_decoder*);

/* ---- multi-threaded CPU baseline driver (bench.py cpu_baseline leg) ---- */
typedef struct orc_bench_cfg {
    double sampling_rate, baud;
    size_t factor, bits;
    float stops, lowpass_bw, lowpass_trans;
    int mathh_context, ungated, with_fft;
} orc_bench_cfg;
/* libm cross-check of the discriminator restatement the oracle uses (orc_atan2f.h): number of differing results; report only */
size_t orc_atan2f_libm_mismatches(uint64_t seed, size_t n);
double orc_bench_run(const orc_bench_cfg* cfg, const float* const* iq_per_thread, const uint32_t* chunk_idx, size_t nchunks,
                     size_t chunk, int repeats, int nthreads, char* sentences, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
