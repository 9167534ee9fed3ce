// Telemetry sentence extraction and CRC on the host.
//
// The reference scans its character stream with std::regex
//     .*?(\$+)([\w,\-,\s]+?),(.+?)(\*|\$)(\w\w\w\w).*        (code/Decoder/sentence_extract.cpp:30)
// as a FULL match on a copy with '\n' -> ' ' (:66), then cuts the stream 4 characters after the
// terminator (:84-85).  std::regex costs tens of microseconds per call, which would make the host the
// bottleneck when thousands of streams deliver characters, so the same leftmost, lazy/greedy backtracking
// order is spelled out here as loops (equivalence is fuzzed against the oracle's std::regex in
// tests/test_host_logic.py).  The stream only ever holds printable ASCII and '\n' (Decoder.h:575-578), so
// `.` matches every character after the newline replacement.
#pragma once
#include <cstddef>
#include <string>

namespace hd {

inline std::string crc16_ccitt_hex(const std::string& s)   // reference code/Decoder/CRC.cpp:21-47
{
    unsigned crc = 0xFFFFu;
    for (unsigned char ch : s) {
        // the reference widens a (signed) char: bytes >= 0x80 would smear into the high half, which is
        // masked off again when printing; printable input never gets there.
        crc ^= static_cast<unsigned>(static_cast<int>(static_cast<signed char>(ch))) << 8;
        for (int b = 0; b < 8; ++b) crc = (crc & 0x8000u) ? (crc << 1) ^ 0x1021u : crc << 1;
    }
    static const char digits[] = "0123456789ABCDEF";
    std::string r(4, '0');
    r[0] = digits[(crc >> 12) & 15]; r[1] = digits[(crc >> 8) & 15]; r[2] = digits[(crc >> 4) & 15]; r[3] = digits[crc & 15];
    return r;
}

struct SentenceMatch {
    std::string callsign, data, crc, rest;
};

namespace detail {
inline bool is_word(char c) { return (c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || c == '_'; }
inline bool is_space(char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }
inline bool in_callsign_class(char c) { return is_word(c) || c == ',' || c == '-' || is_space(c); }
}  // namespace detail

// Returns true and fills `m` when the stream contains a sentence.
inline bool extract_sentence(const std::string& raw, SentenceMatch& m)
{
    using namespace detail;
    std::string s = raw;
    for (char& c : s) if (c == '\n') c = ' ';
    if (s.find('*') == std::string::npos) return false;          // reference: find("*") < npos-4
    const size_t n = s.size();
    // `.*?` : shortest prefix first  ->  dollar run start ascending
    for (size_t d0 = 0; d0 < n; ++d0) {
        if (s[d0] != '$') continue;
        size_t run = 0;
        while (d0 + run < n && s[d0 + run] == '$') ++run;
        // `(\$+)` greedy: longest run first
        for (size_t k = run; k >= 1; --k) {
            const size_t c0 = d0 + k;                             // callsign starts here
            // `([\w,\-,\s]+?),` lazy: shortest callsign first, must be followed by ','
            for (size_t c1 = c0 + 1; c1 < n && in_callsign_class(s[c1 - 1]); ++c1) {
                if (s[c1] != ',') continue;
                const size_t p0 = c1 + 1;                         // data starts here
                // `(.+?)(\*|\$)(\w\w\w\w)` lazy: shortest data first
                for (size_t t = p0 + 1; t + 4 < n; ++t) {
                    if (s[t] != '*' && s[t] != '$') continue;
                    if (!(is_word(s[t + 1]) && is_word(s[t + 2]) && is_word(s[t + 3]) && is_word(s[t + 4]))) continue;
                    m.callsign = s.substr(c0, c1 - c0);
                    m.data = s.substr(p0, t - p0);
                    m.crc = s.substr(t + 1, 4);
                    m.rest = s.substr(t + 4);                     // leaves the CRC's last character behind
                    return true;
                }
            }
        }
    }
    return false;
}

}  // namespace hd
