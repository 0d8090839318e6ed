// util/snappy-codec.h (MI355X facade) -- the raw Snappy block format, the default page codec of
// the reference's Parquet writer (hdfs-parquet-table-writer.cc: codec_ defaults to SNAPPY;
// SnappyDecompressor / SnappyCompressor behind Codec::ProcessBlock32, hdfs-parquet-scanner.cc:
// 830-836, 866-876).  The snappy library is not in this image, so the format is restated from its
// published description (format_description.txt):
//   preamble  = uncompressed length, little-endian base-128 varint (at most 32 bits);
//   elements  = tag byte, low two bits select the kind:
//     00 literal     length-1 in the upper six bits (0..59), or 60..63 = length-1 follows in 1..4
//                    little-endian bytes; then the literal bytes;
//     01 copy        length 4..11 = 4 + bits 2..4; offset = bits 5..7 << 8 | next byte (11 bits);
//     10 copy        length 1..64 = 1 + upper six bits; offset = next two bytes, little-endian;
//     11 copy        length 1..64 = 1 + upper six bits; offset = next four bytes, little-endian.
//   A copy reads `length` bytes starting `offset` bytes back in the output; it may overlap its own
//   output (offset < length repeats a pattern); offset 0 and offsets beyond the output so far are
//   corrupt.
// Host-side only.  The decompressor accepts every valid stream; the compressor is a plain greedy
// matcher (4-byte hash, offsets < 65536, copies split into pieces of at most 64 bytes) -- valid
// output, not byte-identical to libsnappy's, which no reader requires.
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

namespace impala {
namespace snappy {

// uncompressed length of a stream, -1 when the preamble is malformed; *header_len = its bytes
inline int64_t UncompressedLength(const uint8_t* in, int64_t in_len, int* header_len) {
  uint64_t v = 0;
  for (int i = 0; i < 5 && i < in_len; ++i) {
    v |= (uint64_t)(in[i] & 0x7F) << (7 * i);
    if (!(in[i] & 0x80)) {
      if (v > 0xFFFFFFFFull) return -1;
      *header_len = i + 1;
      return (int64_t)v;
    }
  }
  return -1;
}

// in -> out (out_len bytes, which must equal the preamble's length); false on any corrupt input
inline bool Uncompress(const uint8_t* in, int64_t in_len, uint8_t* out, int64_t out_len) {
  int hdr = 0;
  if (UncompressedLength(in, in_len, &hdr) != out_len) return false;
  int64_t ip = hdr, op = 0;
  while (ip < in_len) {
    const uint8_t tag = in[ip++];
    int64_t len, offset;
    switch (tag & 3) {
      case 0: {
        len = (tag >> 2) + 1;
        if (len > 60) {
          const int extra = (int)len - 60;  // 1..4 length bytes
          if (ip + extra > in_len) return false;
          uint32_t l = 0;
          for (int i = 0; i < extra; ++i) l |= (uint32_t)in[ip + i] << (8 * i);
          ip += extra;
          len = (int64_t)l + 1;
        }
        if (len > in_len - ip || len > out_len - op) return false;
        memcpy(out + op, in + ip, (size_t)len);
        ip += len;
        op += len;
        continue;
      }
      case 1:
        if (ip + 1 > in_len) return false;
        len = 4 + ((tag >> 2) & 7);
        offset = ((int64_t)(tag >> 5) << 8) | in[ip];
        ip += 1;
        break;
      case 2:
        if (ip + 2 > in_len) return false;
        len = (tag >> 2) + 1;
        offset = in[ip] | ((int64_t)in[ip + 1] << 8);
        ip += 2;
        break;
      default:
        if (ip + 4 > in_len) return false;
        len = (tag >> 2) + 1;
        offset = in[ip] | ((int64_t)in[ip + 1] << 8) | ((int64_t)in[ip + 2] << 16) | ((int64_t)in[ip + 3] << 24);
        ip += 4;
        break;
    }
    if (offset == 0 || offset > op || len > out_len - op) return false;
    if (offset >= len) {
      memcpy(out + op, out + op - offset, (size_t)len);
    } else {
      for (int64_t i = 0; i < len; ++i) out[op + i] = out[op - offset + i];  // pattern repeat
    }
    op += len;
  }
  return op == out_len;
}

namespace detail {
inline void EmitLiteral(const uint8_t* p, int64_t len, std::vector<uint8_t>* out) {
  if (len <= 0) return;
  const uint32_t n = (uint32_t)(len - 1);
  if (n < 60) {
    out->push_back((uint8_t)(n << 2));
  } else {
    const int bytes = n < (1u << 8) ? 1 : n < (1u << 16) ? 2 : n < (1u << 24) ? 3 : 4;
    out->push_back((uint8_t)((59 + bytes) << 2));
    for (int i = 0; i < bytes; ++i) out->push_back((uint8_t)(n >> (8 * i)));
  }
  out->insert(out->end(), p, p + len);
}
inline void EmitCopy(int64_t offset, int64_t len, std::vector<uint8_t>* out) {
  while (len > 0) {
    // leave no tail shorter than 4 behind a 64-byte piece, so the short form stays usable
    int64_t piece = len > 64 ? (len - 64 < 4 ? 60 : 64) : len;
    if (piece >= 4 && piece <= 11 && offset < 2048) {
      out->push_back((uint8_t)(1 | ((piece - 4) << 2) | ((offset >> 8) << 5)));
      out->push_back((uint8_t)offset);
    } else {
      out->push_back((uint8_t)(2 | ((piece - 1) << 2)));
      out->push_back((uint8_t)offset);
      out->push_back((uint8_t)(offset >> 8));
    }
    len -= piece;
  }
}
inline uint32_t Load32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
}  // namespace detail

inline void Compress(const uint8_t* in, int64_t in_len, std::vector<uint8_t>* out) {
  out->clear();
  out->reserve((size_t)(in_len + in_len / 6 + 32));
  for (uint64_t v = (uint64_t)in_len;;) {  // preamble
    const uint8_t b = v & 0x7F;
    v >>= 7;
    out->push_back(v ? (b | 0x80) : b);
    if (!v) break;
  }
  constexpr int kHashBits = 14;
  std::vector<int64_t> table((size_t)1 << kHashBits, -1);
  int64_t ip = 0, lit = 0;
  while (ip + 4 <= in_len) {
    const uint32_t h = (detail::Load32(in + ip) * 0x1E35A7BDu) >> (32 - kHashBits);
    const int64_t cand = table[h];
    table[h] = ip;
    if (cand >= 0 && ip - cand < 65536 && detail::Load32(in + cand) == detail::Load32(in + ip)) {
      int64_t len = 4;
      while (ip + len < in_len && in[cand + len] == in[ip + len]) ++len;
      detail::EmitLiteral(in + lit, ip - lit, out);
      detail::EmitCopy(ip - cand, len, out);
      ip += len;
      lit = ip;
    } else {
      ++ip;
    }
  }
  detail::EmitLiteral(in + lit, in_len - lit, out);
}

}  // namespace snappy
}  // namespace impala
