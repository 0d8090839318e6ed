// util/fle-encoding.h (MI355X facade) -- impala::FleDecoder / impala::FleEncoder with the public
// interface of the reference's util/fle-encoding.h (class FleDecoder :35-240, FleEncoder :242-342),
// implemented over the C-ABI of libips_hip.so instead of AVX2 intrinsics.
//
// What runs where
//   ctor           page bytes are uploaded to HBM once (the decoder still does not own the host
//                  buffer: "must be maintained by the caller", dict-encoding.h:179-181)
//   Get / Skip     the page is decoded in ONE ips_fle_decode launch on first use and values are
//                  then served from a host copy; the row cursor has the reference's semantics
//                  (fle-encoding.h:344-402, 404-567): Get(v) reads row p and advances,
//                  Get(v, s) reads row p+s and leaves p = p+s+1, Skip(s) adds s; all return
//                  false once the row they need lies at or beyond the end of the buffer
//   Eq..In         non-advancing, append num_rows bits for rows p.. (fle-encoding.h:7962-8313).
//                  The reference is called once per 1024-row batch (hdfs-parquet-scanner.cc:1838);
//                  the facade evaluates the predicate for the WHOLE page in one ips_fle_pred
//                  launch the first time a given (op, constants) is seen on the page and serves
//                  every later batch from that bitmap (a small cache keyed by (op, constants):
//                  a BETWEEN is Ge then Le per batch, simple-predicates.h:145-153) -- the
//                  concatenation of batch bitmaps is identical.
// Differences kept on purpose: a trailing partial block (buffer_len not a multiple of
// 8*bit_width) is ignored rather than over-read.
#pragma once
#include <algorithm>
#include <memory>
#include <vector>

#include "../ips/runtime.h"

namespace impala {

using ips::SkipBitset;

class FleDecoder {
 public:
  FleDecoder(uint8_t* buffer, int buffer_len, int bit_width) : s_(std::make_shared<State>()) {
    s_->bw = bit_width;
    const int64_t block_bytes = (int64_t)bit_width * 8;
    s_->blocks = block_bytes > 0 ? buffer_len / block_bytes : 0;
    s_->rows = s_->blocks * 64;
    s_->usable = bit_width >= 1 && bit_width <= 32;
    if (s_->usable && s_->blocks > 0)
      s_->usable = s_->enc.upload(buffer, (size_t)(s_->blocks * block_bytes));
  }
  FleDecoder() {}

  template <typename T>
  bool Get(T* val) { return Fetch(val, 0); }

  template <typename T>
  bool Get(T* val, int skip_rows) { return Fetch(val, skip_rows); }

  bool Skip(int skip_rows) {
    if (!s_) return false;
    s_->cursor += skip_rows;
    return s_->cursor < s_->rows;  // the landing block would have to be unpacked
  }

  void Eq(int64_t num_rows, SkipBitset& skip_bitset, uint64_t value) { Pred(IPS_OP_EQ, num_rows, skip_bitset, &value, 1); }
  void Lt(int64_t num_rows, SkipBitset& skip_bitset, uint64_t value) { Pred(IPS_OP_LT, num_rows, skip_bitset, &value, 1); }
  void Le(int64_t num_rows, SkipBitset& skip_bitset, uint64_t value) { Pred(IPS_OP_LE, num_rows, skip_bitset, &value, 1); }
  void Gt(int64_t num_rows, SkipBitset& skip_bitset, uint64_t value) { Pred(IPS_OP_GT, num_rows, skip_bitset, &value, 1); }
  void Ge(int64_t num_rows, SkipBitset& skip_bitset, uint64_t value) { Pred(IPS_OP_GE, num_rows, skip_bitset, &value, 1); }
  void In(int64_t num_rows, SkipBitset& skip_bitset, std::vector<uint64_t>& values) {
    Pred(IPS_OP_IN, num_rows, skip_bitset, values.data(), (int)values.size());
  }
  void Clear() {}

  int bit_width() { return s_ ? s_->bw : 0; }

  // ---- facade extras (not in the reference) ----
  int64_t cursor() const { return s_ ? s_->cursor : 0; }
  int64_t rows_in_buffer() const { return s_ ? s_->rows : 0; }
  const void* device_blocks() const { return s_ ? s_->enc.get() : nullptr; }
  bool usable() const { return s_ && s_->usable; }

 private:
  struct State {
    int bw = 0;
    int64_t blocks = 0, rows = 0, cursor = 0;
    bool usable = false;
    ips::DeviceBuffer enc;
    bool decoded = false;
    std::vector<uint32_t> values;       // whole page, filled by one ips_fle_decode
    ips::PredCache preds;               // whole-page bitmaps per (op, constants)
  };

  bool EnsureDecoded() {
    if (s_->decoded) return true;
    if (!s_->usable) return false;
    s_->values.resize((size_t)s_->rows);
    if (s_->rows > 0) {
      ips::DeviceBuffer out((size_t)s_->rows * 4);
      ++ips::stats().decode_launches;
      if (!ips::ok(ips_fle_decode(s_->enc.get(), s_->rows, s_->bw, out.get(), 4, nullptr), "ips_fle_decode") ||
          !out.download(s_->values.data(), (size_t)s_->rows * 4))
        return false;
    }
    s_->decoded = true;
    return true;
  }

  template <typename T>
  bool Fetch(T* val, int skip_rows) {
    if (!s_) return false;
    s_->cursor += skip_rows;
    if (s_->cursor >= s_->rows || !EnsureDecoded()) return false;
    *val = (T)s_->values[(size_t)s_->cursor];
    ++s_->cursor;
    return true;
  }

  void Pred(ips_op op, int64_t num_rows, SkipBitset& out, const uint64_t* consts, int n) {
    if (!s_ || num_rows <= 0) return;
    const std::vector<uint64_t>* words = s_->preds.find((int)op, consts, (size_t)n * 8);
    if (!words) {
      std::vector<uint64_t>* w = s_->preds.insert((int)op, consts, (size_t)n * 8);
      w->assign((size_t)((s_->rows + 63) / 64), 0);
      if (s_->usable && s_->rows > 0) {
        ++ips::stats().pred_launches;
        ips::DeviceBuffer bm((size_t)w->size() * 8);
        bool done = false;
        if (op == IPS_OP_IN && n > IPS_MAX_IN_LIST) {
          // In() takes a vector of any length (fle-encoding.h:8236-8313): beyond what a kernel argument
          // holds the list becomes a resident set for this one whole-page evaluation
          ips_inset* set = nullptr;
          done = ips::ok(ips_inset_open(consts, n, &set), "ips_inset_open") &&
                 ips::ok(ips_fle_pred_inset(s_->enc.get(), s_->rows, s_->bw, set, bm.as<uint64_t>(), nullptr),
                         "ips_fle_pred_inset");
          if (done) ips::ok(ips_stream_synchronize(nullptr), "ips_stream_synchronize");
          ips_inset_close(set);
        } else {
          done = ips::ok(ips_fle_pred(s_->enc.get(), s_->rows, s_->bw, op, consts, n, bm.as<uint64_t>(), nullptr),
                         "ips_fle_pred");
        }
        if (done) bm.download(w->data(), w->size() * 8);
      }
      words = w;
    }
    ips::append_bits(out, *words, s_->cursor, num_rows, s_->rows);
  }

  std::shared_ptr<State> s_;
};

// FleEncoder(buffer, buffer_len, bit_width); Put(value); Flush() -> bytes; fle-encoding.h:242-342,
// 8315-8365, 9806-9812.  Values are staged on the host and bit-sliced by one ips_fle_encode
// launch at Flush().  Padding rows of the last block are zero (undefined in the reference).
class FleEncoder {
 public:
  FleEncoder(uint8_t* buffer, int buffer_len, int bit_width)
      : buffer_(buffer), capacity_(buffer_len), bit_width_(bit_width) {}

  bool Put(uint64_t value) {
    // the reference refuses a value once the next block would not fit (fle-encoding.h:8318)
    const int64_t blocks_after = ((int64_t)staged_.size() + 1 + 63) / 64;
    if (blocks_after * bit_width_ * 8 > capacity_) return false;
    staged_.push_back((uint32_t)value);
    return true;
  }

  int Flush() {
    const int64_t n = (int64_t)staged_.size();
    len_ = (int)ips_fle_encoded_bytes(n, bit_width_);
    if (n == 0) return len_;
    ips::DeviceBuffer vals, enc((size_t)len_);
    if (vals.upload(staged_.data(), (size_t)n * 4) &&
        ips::ok(ips_fle_encode(vals.get(), 4, n, bit_width_, enc.get(), nullptr), "ips_fle_encode"))
      enc.download(buffer_, (size_t)len_);
    return len_;
  }

  void Clear() { staged_.clear(); len_ = 0; }
  uint8_t* buffer() { return buffer_; }
  int len() { return len_; }

 private:
  uint8_t* buffer_;
  int capacity_;
  int bit_width_;
  int len_ = 0;
  std::vector<uint32_t> staged_;
};

}  // namespace impala
