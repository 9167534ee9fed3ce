/*
 * habdec_amd_host.h -- C ABI of the HOST-side pieces of the path (no GPU needed to call these).
 *
 * They are the exact routines the engine uses around its kernels: the decimation plan and coefficient tables,
 * the low-pass design, RTTY framing, sentence extraction + CRC, the AFC state machine, and the float-only
 * atan2f restatement the discriminator kernel runs.  Exported so integrators and the CPU test-suite can drive
 * them directly; each names the reference routine it stands in for.
 */
#ifndef HABDEC_AMD_HOST_H
#define HABDEC_AMD_HOST_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Decoder::setupDecimationStagesFactor's stage choice (Decoder.h:286-320): returns the stage count (0..2) or -1;
 * ratio[i], ntaps[i] describe stage i. */
int hd_host_decim_plan(unsigned total_factor, int ratio[2], unsigned ntaps[2]);
/* coefficients of stage `stage` of that plan (filtercoef.h tables); returns the count copied */
size_t hd_host_decim_taps(unsigned total_factor, int stage, float* taps, size_t cap);

/* FirFilter::LP_BlackmanHarris (FirFilter.h:173-209): design for `batch` input samples; `prev_ntaps` is the tap
 * count currently in use (a design of equal length is skipped -> returns 0).  float_trig: DESIGN.md lookup context. */
size_t hd_host_lowpass_design(float cutoff_rel, float transition, size_t batch, size_t prev_ntaps, int float_trig,
                              float* taps, size_t cap);

/* RTTY<bool> (RTTY.h:59-137) */
typedef struct hd_host_rtty hd_host_rtty;
hd_host_rtty* hd_host_rtty_new(size_t nbits, float nstops);
void hd_host_rtty_free(hd_host_rtty*);
size_t hd_host_rtty_push_run(hd_host_rtty*, const uint8_t* bits, size_t n, char* out, size_t cap);
size_t hd_host_rtty_pending(const hd_host_rtty*);   /* unframed bits the framer still holds (bounded: 2^18 + what can still frame) */

/* CRC (CRC.cpp:21-47) and extractSentence (sentence_extract.cpp:58-98) */
void hd_host_crc16(const char* s, size_t n, char out4[5]);
int hd_host_extract_sentence(const char* stream, size_t n, char* callsign, char* data, char* crc, char* rest, size_t cap);

/* Decoder text stage (Decoder.h:568-637) on raw framed bytes */
typedef struct hd_host_text hd_host_text;
hd_host_text* hd_host_text_new(size_t nbits, float nstops);
void hd_host_text_free(hd_host_text*);
void hd_host_text_push_bits(hd_host_text*, const uint8_t* bits, size_t n);
size_t hd_host_text_get(hd_host_text*, int which /*0 rtty stream,1 last sentence,2 ok log,3 match log,4 chars log*/, char* buf, size_t cap);

/* AFC::process after the spectrum reductions (AFC.h:108-184) + resetFrequencyCorrection (AFC.h:187-194) */
typedef struct hd_host_afc hd_host_afc;
hd_host_afc* hd_host_afc_new(void);
void hd_host_afc_free(hd_host_afc*);
void hd_host_afc_step(hd_host_afc*, int have_spectrum, int valid, int peak1, int peak2, float power1, float power2,
                      double mean, double sigma, size_t bins, double rate);
void hd_host_afc_reset(hd_host_afc*, double correction, size_t bins, double rate);
void hd_host_afc_get(hd_host_afc*, double* correction, double* shift_hz, double* noise_floor, double* noise_sigma,
                     int* peak_l, int* peak_r);

/* the discriminator's arithmetic, compiled for the host: bit-identical to glibc atan2f (FSK2_Demod.h:38) */
void hd_host_atan2f(const float* y, const float* x, float* out, size_t n);
void hd_host_discriminate(const float* iq, size_t n, float prev_re, float prev_im, float* out);

/* ---- GUI payloads: the websocket server's binary "PWR_" / "DEM_" bodies, per stream ----
 * hd_host_spectrum_payload = SpectrumToStream (habdec_ws_protocol.cpp:355-405): zoom to the centre bins (zoom clamped to
 * [0.01, 0.99]), thin to `resolution` bins (ShrinkVector, :338-351), quantise to type_size 1/2/4 bytes per bin against the
 * slice's min/max (CompressedVector.cpp:75-118), behind the 52-byte SpectrumInfoHeader (NetTransport.h:29-47).  `bins` is
 * hd_stream_power(), the scalars are hd_stream_afc() (peak_left/right signed like AFC::getPeaks) and the decimated rate.
 * hd_host_demod_payload = DemodToStream + the 20-byte DemodHeader (:408-429, NetTransport.h:50-57) for hd_stream_demodulated().
 * Both return the payload size in bytes (written to `out` when cap suffices) and the number of values sent. */
size_t hd_host_spectrum_payload(const float* bins, size_t n, double noise_floor, double noise_variance, double sampling_rate, double shift,
                                int peak_left, int peak_right, float zoom, int resolution, int type_size, uint8_t* out, size_t cap,
                                size_t* bins_sent);
size_t hd_host_demod_payload(const float* trace, size_t n, int resolution, int type_size, uint8_t* out, size_t cap, size_t* values_sent);

/* ---- post-decode telemetry ----
 * Return convention of the parsers: 1 = parsed, 0 = the reference returns "nothing" (std::nullopt / no GPS fix / fewer than
 * six fields), -1 = an input on which the reference lets std::stoi / std::stof / std::string::at throw. */
typedef struct hd_host_telemetry {
    char callsign[64];            /* leading '$' run removed (sentence_parse.cpp:156-163) */
    int32_t frame, hour, minute;
    float second, lat, lon, alt;  /* decimal degrees; NMEA ddmm.mmmm inputs converted (sentence_parse.cpp:106-143) */
} hd_host_telemetry;
int hd_host_parse_time(const char* text, int* hour, int* minute, float* second);             /* parse_sentence_time, :47-68 */
int hd_host_parse_gps_pos(const char* text, float* out);                                      /* parse_gps_pos, :106-143 */
int hd_host_parse_sentence(const char* sentence_without_crc, hd_host_telemetry* out);         /* parse_sentence, :146-196, without the clock */
/* timestamp_from_HMS (:73-100) with the clock passed in (seconds since the epoch, UTC): "YYYY-MM-DDTHH:MM:SSZ" */
size_t hd_host_timestamp_from_hms(int64_t now_unix, int hour, int minute, float second, char* buf, size_t cap);
/* CalcGpsDistance (GpsDistance.cpp:21-84): out = {line distance m, great-circle distance m, angle rad, elevation deg, bearing deg} */
void hd_host_gps_distance(double lat1, double lon1, double alt1, double lat2, double lon2, double alt2, double out[5]);

/* ---- sondehub upload batch (last hop): the JSON body of SondeHubUploader::upload (sondehub_uploader.cpp:31-69) ----
 * Any number of threads push (one record per CRC-valid sentence: what SentenceCallback does, websocketServer/main.cpp:286-306 --
 * habdec::parse_sentence, time_received = utc_now_iso()); one thread takes the batch: a JSON array with the reference's eleven fields per
 * record, serialised byte for byte like its nlohmann::json (keys sorted, compact, Grisu2 shortest floats, alt as int).  The HTTP PUT
 * itself is the application's.  Clocks are passed in (nanoseconds since the epoch, UTC), so results are reproducible. */
typedef struct hd_host_sondehub hd_host_sondehub;
hd_host_sondehub* hd_host_sondehub_new(const char* uploader_callsign, const char* software_version /* first 7 characters are used */);
void hd_host_sondehub_free(hd_host_sondehub*);
/* 1 = queued; 0 = dropped like the reference does (fewer than six fields, no GPS fix, bad time); -1 = the reference would throw (stoi / stof) */
int hd_host_sondehub_push_sentence(hd_host_sondehub*, uint32_t stream, const char* callsign, const char* data, int64_t now_unix_ns);
int hd_host_sondehub_push(hd_host_sondehub*, const char* payload_callsign, const char* time_received, const char* datetime, int frame,
                          float lat, float lon, float alt);                                   /* SondeHubUploader::push of a ready MinTelemetry */
size_t hd_host_sondehub_size(const hd_host_sondehub*);
/* Body size in bytes (0 = queue empty); written NUL-terminated to out when cap > size, and only then is the batch consumed. */
size_t hd_host_sondehub_take(hd_host_sondehub*, int64_t now_unix_ns, char* out, size_t cap, size_t* n_records);
size_t hd_host_utc_iso(int64_t unix_ns, char* buf, size_t cap);                              /* utc_now_iso() (common/utc_now_iso.cpp:7-22) at a given instant */
size_t hd_host_json_number(double v, char* buf, size_t cap);                                 /* how the body prints a float field */

/* ---- batched cf32 file ingest: S IQ files -> one push slab per round ----
 * Each file is an IQSource_File<float> (IQSource_File.h:124-172): raw interleaved float32 I,Q, no header; a read returns
 * what is left, the end of file is noticed by the read that runs into it, and the NEXT read rewinds when `loop` (else
 * returns 0).  `chunk` = samples requested per stream and round (the reference asks for 65536, main.cpp:235);
 * `granule` = the decoder's total decimation factor: every stream delivers a whole multiple of it per round and the
 * remainder is carried in front of its next round (what Decoder::process does with its queue, Decoder.h:429-435).
 * `realtime_rate` > 0 throttles each round like IQSource_File.h:165-169; 0 = as fast as the files can be read. */
typedef struct hd_host_iqfiles hd_host_iqfiles;
hd_host_iqfiles* hd_host_iqfiles_open(const char* const* paths, uint32_t n_files, int loop, uint32_t chunk, uint32_t granule,
                                      double realtime_rate);
void hd_host_iqfiles_close(hd_host_iqfiles*);
uint32_t hd_host_iqfiles_streams(const hd_host_iqfiles*);
uint64_t hd_host_iqfiles_count(const hd_host_iqfiles*, uint32_t stream);     /* IQSource_File::count(): samples in the file */
uint64_t hd_host_iqfiles_rewinds(const hd_host_iqfiles*, uint32_t stream);
/* One round: slab[s*stride .. ] (stride in complex samples, >= chunk) receives stream s's samples, n_per_stream[s] how many
 * (a multiple of granule).  Returns the number of streams that read anything (0: all files exhausted and not looping). */
uint32_t hd_host_iqfiles_next(hd_host_iqfiles*, float* slab, size_t stride, uint32_t* n_per_stream);

#ifdef __cplusplus
}
#endif
#endif
