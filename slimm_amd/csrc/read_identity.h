// The reference's read key, in canonical form (quirk Q18).
//
// analyze_alignments keys its reads by a STRING: qName + ".1" (first in pair) / ".2" (else last in pair) / nothing
// (reference src/slimm.hpp:204-208).  A first-in-pair record of read "N" and an unflagged record of a read literally
// named "N.1" are therefore ONE read.  The identity this library works with -- (name key, mate) -- is made a bijection
// with those strings by shortening such a name:
//   (base, mate) = (name, 1) if flag & 0x40;  (name, 2) else if flag & 0x80;
//                  else (name minus its last two bytes, 1 / 2) if the name ends in ".1" / ".2";  else (name, 0).
// Keys, check words and adjacent-name compares all work on the base; the flag handed on carries the base's mate bit.
#pragma once
#include <cstddef>
#include <cstdint>

namespace slimm {

// shortens n to the base's length; returns the flag with the mate bit (0x40 / 0x80) the base carries
inline uint16_t canonical_read(const char* s, size_t& n, uint16_t flag) {
    if (flag & 0xC0u) return flag;
    if (n >= 2 && s[n - 2] == '.' && (s[n - 1] == '1' || s[n - 1] == '2')) {
        flag = static_cast<uint16_t>(flag | (s[n - 1] == '1' ? 0x40u : 0x80u));
        n -= 2;
    }
    return flag;
}

}  // namespace slimm
