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
    bool scanned_clean = false;  // `stream` is known to hold no sentence: a full scan said so, and it has only been appended to or cut at the front since

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
        // The scan is a backtracking search over the whole stream (up to ~1000 characters: tens of microseconds on noise), run by the reference after every
        // push that brought characters.  A stream that held no sentence can only hold one now if the new characters complete a sentence's ending -- a
        // terminator ('*' or '$') and four word characters -- or bring the '*' the reference insists on (sentence_extract.cpp:70); otherwise the scan's
        // answer is known.  (Without this the text stage of 1024 streams, an eighth of them decoding noise, grew from 50 us to 800 us per call within a minute.)
        bool scan = stream.size() > 20;
        if (scan && scanned_clean) {
            scan = false;
            const size_t n = stream.size(), first = n - printable.size();
            for (size_t p = first; p < n && !scan; ++p) {
                const char* c = stream.data() + p;
                scan = *c == '*' || (p >= 4 && detail::is_word(c[0]) && detail::is_word(c[-1]) && detail::is_word(c[-2]) && detail::is_word(c[-3]) && (c[-4] == '*' || c[-4] == '$'));
            }
        }
        if (stream.size() <= 20) scanned_clean = false;      // (not scanned at this length: unknown)
        if (scan) {
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
            scanned_clean = true;
        }
        if (stream.size() > 1000) stream.erase(0, stream.rfind('$'));
        return printable;
    }
    template <typename OnSentence>
    std::string run(bool had_bits, OnSentence&& on_sentence) { return run(had_bits, on_sentence, [](const SentenceMatch&, bool) {}); }
};

}  // namespace hd
