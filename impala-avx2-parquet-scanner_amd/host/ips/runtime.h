// ips/runtime.h -- small host-side helpers of the C++ facade: status handling, device buffers
// and the bitmap container used in the facade signatures.  Pure C++17 over the C-ABI (include/
// ips.h); nothing here includes HIP.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "../../../include/ips.h"

#if defined(IPS_USE_BOOST_DYNAMIC_BITSET)
#include <boost/dynamic_bitset.hpp>
#endif

namespace ips {

// The reference never throws on the scan path (bool returns + Status parse_status_,
// hdfs-parquet-scanner.h:157).  The facade records the first failing status per thread; callers
// that return bool fold it into their result, the scanner facade exposes it as parse_status().
inline ips_status& sticky_status() {
  static thread_local ips_status s = IPS_OK;
  return s;
}
inline bool ok(ips_status s, const char* where) {
  if (s == IPS_OK) return true;
  if (sticky_status() == IPS_OK) sticky_status() = s;
  fprintf(stderr, "[ips] %s failed: status %d: %s\n", where, (int)s, ips_last_error());
  return false;
}

// Device work issued by the facade, for tests of its call pattern (the reference evaluates every
// leaf once per 1024-row batch, hdfs-parquet-scanner.cc:1838; the facade evaluates a page once per
// distinct predicate and serves the batches from that bitmap).
struct FacadeStats {
  long pred_launches = 0;   // whole-page predicate evaluations (FLE / dictionary / PLAIN)
  long page_uploads = 0;    // host -> device copies of page payloads
  long decode_launches = 0;
};
inline FacadeStats& stats() {
  static thread_local FacadeStats s;
  return s;
}

// Whole-page bitmaps of the predicates seen on a page, keyed by (op, constants): a BETWEEN is
// Ge then Le per batch (simple-predicates.h:145-153), a conjunct list alternates even more leaves,
// so one slot would evict itself on every call.  Small and linear: a page sees a handful of leaves.
class PredCache {
 public:
  // words of the entry for (op, key bytes), or NULL
  const std::vector<uint64_t>* find(int op, const void* key, size_t key_len) const {
    for (const Entry& e : entries_)
      if (e.op == op && e.key.size() == key_len && memcmp(e.key.data(), key, key_len) == 0) return &e.words;
    return nullptr;
  }
  std::vector<uint64_t>* insert(int op, const void* key, size_t key_len) {
    if (entries_.size() >= kMaxEntries) entries_.erase(entries_.begin());  // oldest out
    entries_.emplace_back();
    Entry& e = entries_.back();
    e.op = op;
    e.key.assign((const uint8_t*)key, (const uint8_t*)key + key_len);
    return &e.words;
  }
  void clear() { entries_.clear(); }

 private:
  static const size_t kMaxEntries = 16;
  struct Entry { int op; std::vector<uint8_t> key; std::vector<uint64_t> words; };
  std::vector<Entry> entries_;
};

// Owns one device allocation.
class DeviceBuffer {
 public:
  DeviceBuffer() = default;
  explicit DeviceBuffer(size_t bytes) { resize(bytes); }
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  DeviceBuffer(DeviceBuffer&& o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
  DeviceBuffer& operator=(DeviceBuffer&& o) noexcept {
    if (this != &o) { release(); p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; }
    return *this;
  }
  ~DeviceBuffer() { release(); }
  bool resize(size_t bytes) {
    if (bytes <= n_) return true;
    release();
    if (!ok(ips_malloc(&p_, bytes + 16), "ips_malloc")) { p_ = nullptr; return false; }
    n_ = bytes;
    return true;
  }
  void release() {
    if (p_) ips_free(p_);
    p_ = nullptr;
    n_ = 0;
  }
  void* get() const { return p_; }
  template <typename T> T* as() const { return reinterpret_cast<T*>(p_); }
  size_t size() const { return n_; }
  bool upload(const void* h, size_t bytes) {
    ++stats().page_uploads;
    return resize(bytes) && (bytes == 0 || ok(ips_memcpy_h2d(p_, h, bytes, nullptr), "ips_memcpy_h2d"));
  }
  bool download(void* h, size_t bytes) const {
    return bytes == 0 || (ok(ips_memcpy_d2h(h, p_, bytes, nullptr), "ips_memcpy_d2h") &&
                          ok(ips_stream_synchronize(nullptr), "ips_stream_synchronize"));
  }

 private:
  void* p_ = nullptr;
  size_t n_ = 0;
};

// The part of boost::dynamic_bitset<>'s contract the scan path uses (call sites:
// fle-encoding.h:7975,8007; dict-encoding.h:466; hdfs-parquet-scanner.cc:329,344,1125,1139;
// simple-predicates.h:152,162): LSB-first 64-bit blocks, append(block) places bit k of the block
// at index size()+k.  Define IPS_USE_BOOST_DYNAMIC_BITSET to use Boost's own class instead.
class DynamicBitset {
 public:
  typedef uint64_t block_type;
  void push_back(bool bit) {
    if ((n_ & 63) == 0) w_.push_back(0);
    if (bit) w_[n_ >> 6] |= 1ull << (n_ & 63);
    ++n_;
  }
  void append(block_type block) {
    const int sh = (int)(n_ & 63);
    if (sh == 0) {
      w_.push_back(block);
    } else {
      w_.back() |= block << sh;
      w_.push_back(block >> (64 - sh));
    }
    n_ += 64;
  }
  void resize(size_t n, bool value = false) {
    const size_t old = n_;
    w_.resize((n + 63) / 64, value ? ~0ull : 0ull);
    if (n > old && value && (old & 63)) w_[old >> 6] |= ~0ull << (old & 63);
    n_ = n;
    trim();
  }
  void clear() { w_.clear(); n_ = 0; }
  size_t size() const { return n_; }
  size_t count() const {
    size_t c = 0;
    for (uint64_t x : w_) c += (size_t)__builtin_popcountll(x);
    return c;
  }
  bool operator[](size_t i) const { return (w_[i >> 6] >> (i & 63)) & 1; }
  bool test(size_t i) const { return (*this)[i]; }
  void set(size_t i, bool v) {
    if (v) w_[i >> 6] |= 1ull << (i & 63); else w_[i >> 6] &= ~(1ull << (i & 63));
  }
  DynamicBitset& operator&=(const DynamicBitset& o) {
    for (size_t i = 0; i < w_.size() && i < o.w_.size(); ++i) w_[i] &= o.w_[i];
    return *this;
  }
  DynamicBitset& operator|=(const DynamicBitset& o) {
    for (size_t i = 0; i < w_.size() && i < o.w_.size(); ++i) w_[i] |= o.w_[i];
    return *this;
  }
  const std::vector<uint64_t>& words() const { return w_; }

 private:
  void trim() {
    if (n_ & 63) w_.back() &= (1ull << (n_ & 63)) - 1ull;
  }
  std::vector<uint64_t> w_;
  size_t n_ = 0;
};

#if defined(IPS_USE_BOOST_DYNAMIC_BITSET)
typedef boost::dynamic_bitset<> SkipBitset;
#else
typedef DynamicBitset SkipBitset;
#endif

// Append bits [first, first + n) of an LSB-first word array to a bitset (rows beyond 'limit' bits
// of the source are appended as 0: the reference reads out of bounds there).
template <typename Bitset>
inline void append_bits(Bitset& dst, const std::vector<uint64_t>& src, int64_t first, int64_t n,
                        int64_t limit) {
  int64_t i = 0;
  auto word_at = [&](int64_t bit) -> uint64_t {  // 64 source bits starting at 'bit'
    if (bit >= limit) return 0;
    const int64_t w = bit >> 6;
    const int sh = (int)(bit & 63);
    uint64_t x = src[(size_t)w] >> sh;
    if (sh && (size_t)(w + 1) < src.size()) x |= src[(size_t)w + 1] << (64 - sh);
    const int64_t valid = limit - bit;
    if (valid < 64) x &= (1ull << valid) - 1ull;
    return x;
  };
  for (; i + 64 <= n; i += 64) dst.append(word_at(first + i));
  if (i < n) {
    uint64_t x = word_at(first + i);
    for (; i < n; ++i, x >>= 1) dst.push_back(x & 1ull);
  }
}

}  // namespace ips
