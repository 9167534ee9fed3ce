// Asynchronous-serial (RTTY) character framing on the host: bits -> bytes.
//
// Behaviour of the reference's RTTY<bool> (code/Decoder/RTTY.h:77-137): a frame is a 0 start bit, `nbits`
// data bits LSB first and `nstops` 1 stop bits; every framed byte is emitted (printable or not); after a
// scan everything up to the last stop bit of the last frame is dropped, an unframed tail is kept.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace hd {

class RttyFramer {
public:
    size_t nbits = 0;
    float nstops = 0;

    void push_packed(const uint32_t* words, uint32_t count)
    {
        for (uint32_t i = 0; i < count; ++i) fifo_.push_back((words[i >> 5] >> (i & 31)) & 1u);
    }
    void push_bits(const uint8_t* b, size_t n) { fifo_.insert(fifo_.end(), b, b + n); }
    size_t pending() const { return fifo_.size(); }

    // Appends framed bytes to `out`; returns how many.
    size_t frame(std::string& out)
    {
        if (!nbits && !nstops) return 0;
        const float need = 1 + nbits + nstops;
        if (fifo_.size() < need) return 0;
        size_t made = 0, consumed_to = 0;
        size_t i = 0;
        while (i < fifo_.size()) {
            bool ok = fifo_[i] == 0 && (i + need) <= fifo_.size();
            for (size_t s = 0; ok && s < nstops; ++s) ok = fifo_[i + 1 + nbits + s] == 1;
            if (!ok) { ++i; continue; }
            char c = 0;
            for (size_t k = 0; k < nbits; ++k) c += fifo_[i + 1 + k] << k;
            out.push_back(c);
            ++made;
            i = static_cast<size_t>(i + 1 + nbits + nstops);   // float arithmetic like the reference's `i += nstops_`
            consumed_to = i;
        }
        if (consumed_to > 1) fifo_.erase(fifo_.begin(), fifo_.begin() + consumed_to);
        return made;
    }

private:
    std::vector<uint8_t> fifo_;
};

}  // namespace hd
