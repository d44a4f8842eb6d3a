// fast_inflate: one raw DEFLATE stream (a BGZF block's payload) into a buffer of known size.
#pragma once
#include <cstddef>
#include <cstdint>

namespace bdhost {

// in[0, in_len) is the compressed payload; up to 32 bytes behind it may be READ (a BGZF payload is followed by its 8-byte
// footer and the next block) -- callers that cannot guarantee that use zlib.  Exactly out_len bytes must come out; up to
// out_slack bytes behind out + out_len may be overwritten with scratch (match copies run eight bytes at a time).
// Returns false on anything unexpected; the output is then undefined and the caller should let zlib judge the block.
bool fast_inflate(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_len, size_t out_slack);

// CRC-32 (the gzip / BGZF polynomial, as zlib's crc32 with an initial 0) of buf[0, n): sixteen table look-ups per sixteen bytes --
// about four times zlib 1.2.11's rate, so that checking every inflated block costs the host reader a few percent, not half its time.
uint32_t crc32_fast(const uint8_t* buf, size_t n);

}  // namespace bdhost
