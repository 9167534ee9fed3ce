// Asynchronous-serial (RTTY) character framing on the host: bits -> bytes.
//
// Behaviour of the reference's RTTY<bool> (code/Decoder/RTTY.h:77-137): a frame is a 0 start bit, `nbits`
// data bits LSB first and `nstops` 1 stop bits; every framed byte is emitted (printable or not); after a
// scan everything up to the last stop bit of the last frame is dropped, an unframed tail is kept -- here only as far as it can still matter (frame()).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <algorithm>
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
        // The reference drops what it has framed and keeps the rest (RTTY.h:134) -- for good -- and scans all of it again on every push: on a stream that
        // never frames (a carrier far off tune demodulates to one level) that is every bit ever pushed, 50 us per call and stream after a minute.  A position
        // that failed with its whole frame present fails identically for as long as the character format stays the same, so the scan starts at the first
        // position that was still WAITING for bits last time (scan_from_); a change of format scans everything once more, as the reference would.
        if (nbits != seen_nbits_ || nstops != seen_nstops_) { seen_nbits_ = nbits; seen_nstops_ = nstops; scan_from_ = 0; }
        size_t made = 0, consumed_to = 0;
        size_t i = scan_from_;
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
        if (consumed_to > 1) { fifo_.erase(fifo_.begin(), fifo_.begin() + consumed_to); scan_from_ = 0; }
        // where the next scan starts: the first start bit whose frame is still incomplete (only the last few positions can be)
        size_t live = fifo_.size();
        const size_t span = static_cast<size_t>(need) + 1;
        for (size_t k = fifo_.size() > span ? fifo_.size() - span : 0; k < fifo_.size(); ++k)
            if (k >= scan_from_ && fifo_[k] == 0 && (k + need) > fifo_.size()) { live = k; break; }
        scan_from_ = live;
        // ... and the one departure from the reference: beyond kKeep bits the oldest bits that can no longer frame in this format are forgotten (they could
        // only matter if the format changed after more than kKeep unframed bits -- the reference would then frame hours-old noise)
        if (fifo_.size() > kKeep && scan_from_ > 0) {
            const size_t n = std::min(fifo_.size() - kKeep, scan_from_);
            fifo_.erase(fifo_.begin(), fifo_.begin() + n);
            scan_from_ -= n;
        }
        return made;
    }

    static constexpr size_t kKeep = size_t(1) << 18;   // unframed bits kept for a later change of format (1.5 hours of a 50-baud stream)

private:
    std::vector<uint8_t> fifo_;
    size_t scan_from_ = 0;                   // positions in front of it are known not to frame in the format below
    size_t seen_nbits_ = 0;
    float seen_nstops_ = 0;
};

}  // namespace hd
