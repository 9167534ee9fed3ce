// Drives habdec::Decoder<float> (the MI355X facade) exactly like the reference's DECODER_THREAD
// (code/websocketServer/main.cpp:203-283): read 65536 cf32 samples from a raw IQ file (the IQSource_File layout),
// pushSamples(), operator()(), poll the getters the websocket layer polls.  Prints one line per callback so the test
// can compare with the CPU oracle.
//   usage: decoder_thread_demo <iqfile> <sampling_rate> <dec_exponent> <baud> <bits> <stops> [lowpass_hz]
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "habdec/Decoder.h"

typedef float TReal;
typedef habdec::Decoder<TReal> TDecoder;

int main(int argc, char** argv)
{
    if (argc < 7) { std::fprintf(stderr, "usage\n"); return 2; }
    std::ifstream f(argv[1], std::ios::binary);
    const double fs = std::atof(argv[2]);
    TDecoder DECODER;
    // configuration order of websocketServer/main.cpp:544-554
    DECODER.baud(std::atof(argv[4]));
    DECODER.rtty_bits(std::atoi(argv[5]));
    DECODER.rtty_stops((float)std::atof(argv[6]));
    DECODER.livePrint(false);
    DECODER.dc_remove(false);
    DECODER.lowpass_bw(argc > 7 ? (float)std::atof(argv[7]) : 1500.0f);
    DECODER.lowpass_trans(0.025f);
    DECODER.setupDecimationStagesFactor(1u << std::atoi(argv[3]));
    size_t n_sent = 0, reentered = 0;
    DECODER.sentence_callback_ = [&](std::string callsign, std::string data, std::string crc) {
        std::printf("SENTENCE %s,%s*%s\n", callsign.c_str(), data.c_str(), crc.c_str());
        ++n_sent;
        // what a GUI does from this callback: ask the decoder for its text and its state (no lock is held here, as in the reference)
        const std::string last = DECODER.getLastSentence();
        if (last == callsign + "," + data + "*" + crc && !DECODER.getRTTY().empty() && DECODER.getDecimationFactor() > 0) ++reentered;
        (void)DECODER.getFrequencyCorrection();
    };
    std::string chars;
    DECODER.character_callback_ = [&](std::string c) { chars += c; };

    habdec::IQVector<TReal> samples;
    samples.resize(256 * 256);
    samples.samplingRate(fs);
    size_t total = 0;
    for (;;) {
        f.read(reinterpret_cast<char*>(samples.data()), samples.size() * sizeof(std::complex<TReal>));
        const size_t count = (size_t)f.gcount() / sizeof(std::complex<TReal>);
        if (!count) break;
        samples.resize(count);
        DECODER.pushSamples(samples);
        DECODER();
        total += count;
        (void)DECODER.getFrequencyCorrection();
        (void)DECODER.getDemodulated();
    }
    const auto info = DECODER.getSpectrumInfo();
    std::printf("RTTY %s\n", DECODER.getRTTY().c_str());
    std::printf("LAST %s\n", DECODER.getLastSentence().c_str());
    std::printf("INFO dec=%d fsd=%.3f bins=%zu spectrum=%zu peak_l=%d peak_r=%d samples=%zu sentences=%zu reentered=%zu\n", DECODER.getDecimationFactor(),
                DECODER.getDecimatedSamplingRate(), DECODER.getBinsCount(), info.size(), info.peak_left_, info.peak_right_, total, n_sent, reentered);
    // setupDecimationStagesBW (Decoder.h:336-412): smallest power-of-two division that reaches the rate; more than /256 is refused
    std::printf("BW %zu %zu %zu\n", DECODER.setupDecimationStagesBW(fs / 10.0), DECODER.setupDecimationStagesBW(fs * 2.0), DECODER.setupDecimationStagesBW(fs / 1000.0));
    return 0;
}
