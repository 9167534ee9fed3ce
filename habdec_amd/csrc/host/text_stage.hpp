// Per-stream host text stage: framed bytes -> printable stream -> sentences (+ logs for the C ABI).
//
// Follows what Decoder::process() does after the symbol extractor (reference code/Decoder/Decoder.h:568-637):
// keep printable characters and '\n', append to the RTTY text stream, and while more than 20 characters are
// buffered scan for `$$callsign,data*CRC`; a match always updates "last sentence", the sentence callback fires
// only when the CRC16 matches; finally trim a >1000 character stream to its last '$'.
#pragma once
#include <cstdint>
#include <functional>
#include <string>

#include "rtty_framer.hpp"
#include "sentence.hpp"

namespace hd {

struct TextStage {
    RttyFramer framer;
    std::string stream;          // getRTTY()
    std::string last_sentence;   // getLastSentence()
    std::string ok_log, match_log, char_log;
    uint64_t ok_count = 0;

    // `bits` of this call have already been pushed into `framer`.  Returns the printable chars of this call.
    // on_match: every sentence the scan finds, with the verdict of its CRC (what the reference prints, Decoder.h:601); on_sentence: those whose
    // CRC16 matches (Decoder.h:604-606).
    template <typename OnSentence, typename OnMatch>
    std::string run(bool had_bits, OnSentence&& on_sentence, OnMatch&& on_match)
    {
        std::string raw;
        if (had_bits) framer.frame(raw);
        if (raw.empty()) return {};
        std::string printable;
        for (char c : raw)
            if ((c >= 0x20 && c <= 0x7e) || c == '\n') printable.push_back(c);
        stream += printable;
        char_log += printable;
        if (stream.size() > 20) {
            SentenceMatch m;
            while (extract_sentence(stream, m)) {
                stream = m.rest;
                last_sentence = m.callsign + "," + m.data + "*" + m.crc;
                match_log += last_sentence + "\n";
                const bool ok = m.crc == crc16_ccitt_hex(m.callsign + "," + m.data);
                on_match(m, ok);
                if (ok) {
                    ok_log += last_sentence + "\n";
                    ++ok_count;
                    on_sentence(m);
                }
            }
        }
        if (stream.size() > 1000) stream.erase(0, stream.rfind('$'));
        return printable;
    }
    template <typename OnSentence>
    std::string run(bool had_bits, OnSentence&& on_sentence) { return run(had_bits, on_sentence, [](const SentenceMatch&, bool) {}); }
};

}  // namespace hd
