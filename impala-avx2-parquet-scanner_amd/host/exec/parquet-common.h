// exec/parquet-common.h (MI355X facade) -- ParquetPlainEncoder for the fixed-width types on the
// scan path (int8/16/32/64, float, double), parquet-common.h:84-267, 304-449.
//   ByteSize / Encode / Decode / Skip   host, as in the reference
//   Eq / Lt / Le / Gt / Ge / In         one ips_plain_pred launch over the page rows handed in
// Operand order: IPS_PLAIN_SEMANTICS selects IPS_SEM_REFERENCE (default: bit = val OP x, what the
// reference computes, parquet-common.h:203-247) or IPS_SEM_SQL (bit = x OP val).  In() has an
// empty body in the reference (:252-255); here it evaluates a real IN under SQL semantics and is
// a no-op under REFERENCE semantics.
// Out of scope (not in north_star): strings, decimals, timestamps.
#pragma once
#include <string.h>

#include <vector>

#include "../ips/runtime.h"

#ifndef IPS_PLAIN_SEMANTICS
#define IPS_PLAIN_SEMANTICS IPS_SEM_REFERENCE
#endif

namespace impala {

using ips::SkipBitset;

template <typename T> struct IpsTypeOf;
template <> struct IpsTypeOf<int8_t> { static const ips_type value = IPS_T_INT8; };
template <> struct IpsTypeOf<int16_t> { static const ips_type value = IPS_T_INT16; };
template <> struct IpsTypeOf<int32_t> { static const ips_type value = IPS_T_INT32; };
template <> struct IpsTypeOf<int64_t> { static const ips_type value = IPS_T_INT64; };
template <> struct IpsTypeOf<float> { static const ips_type value = IPS_T_FLOAT; };
template <> struct IpsTypeOf<double> { static const ips_type value = IPS_T_DOUBLE; };

class ParquetPlainEncoder {
 public:
  // int8/int16 occupy a 4-byte slot: "Parquet doesn't have 8-bit or 16-bit ints"
  template <typename T>
  static int ByteSize(const T&) { return ips_plain_stride(IpsTypeOf<T>::value); }

  template <typename T>
  static int Encode(uint8_t* buffer, int /*fixed_len_size*/, const T& t) {
    const int n = ByteSize(t);
    memset(buffer, 0, (size_t)n);
    if (sizeof(T) < 4) { int32_t wide = t; memcpy(buffer, &wide, 4); }
    else memcpy(buffer, &t, sizeof(T));
    return n;
  }
  template <typename T>
  static int Decode(uint8_t* buffer, int /*fixed_len_size*/, T* v) {
    memcpy(v, buffer, sizeof(T));
    return ByteSize(*v);
  }
  template <typename T>
  static int Decode(uint8_t* buffer, int /*fixed_len_size*/, T* v, int skip_rows) {
    const int skip_bytes = ByteSize(*v) * skip_rows;
    memcpy(v, buffer + skip_bytes, sizeof(T));
    return ByteSize(*v) + skip_bytes;
  }
  template <typename T>
  static int Skip(uint8_t* /*buffer*/, int /*fixed_len_size*/, T* v, int skip_rows) {
    return ByteSize(*v) * skip_rows;
  }

  template <typename T>
  static void Eq(uint8_t* b, int f, int64_t n, SkipBitset& s, T& val) { Pred(IPS_OP_EQ, b, f, n, s, &val, 1); }
  template <typename T>
  static void Lt(uint8_t* b, int f, int64_t n, SkipBitset& s, T& val) { Pred(IPS_OP_LT, b, f, n, s, &val, 1); }
  template <typename T>
  static void Le(uint8_t* b, int f, int64_t n, SkipBitset& s, T& val) { Pred(IPS_OP_LE, b, f, n, s, &val, 1); }
  template <typename T>
  static void Gt(uint8_t* b, int f, int64_t n, SkipBitset& s, T& val) { Pred(IPS_OP_GT, b, f, n, s, &val, 1); }
  template <typename T>
  static void Ge(uint8_t* b, int f, int64_t n, SkipBitset& s, T& val) { Pred(IPS_OP_GE, b, f, n, s, &val, 1); }
  template <typename T>
  static void In(uint8_t* b, int f, int64_t n, SkipBitset& s, std::vector<T>& val) {
    if (IPS_PLAIN_SEMANTICS == IPS_SEM_REFERENCE || val.empty()) return;  // reference: empty body
    Pred(IPS_OP_IN, b, f, n, s, val.data(), (int)val.size());
  }

 private:
  template <typename T>
  static void Pred(ips_op op, uint8_t* buffer, int /*fixed_len_size*/, int64_t num_rows,
                   SkipBitset& out, const T* lits, int n_lits) {
    if (num_rows <= 0) return;
    const ips_type t = IpsTypeOf<T>::value;
    // the static interface has nowhere to keep a page (the scanner's column reader does: it uploads
    // a page once per InitDataPage and caches whole-page bitmaps, hdfs-parquet-scanner.h); the device
    // scratch at least lives as long as the calling thread: no device allocation per call
    static thread_local ips::DeviceBuffer page, bm;
    if (!bm.resize((size_t)((num_rows + 63) / 64) * 8)) return;
    std::vector<uint64_t> words((size_t)((num_rows + 63) / 64), 0);
    if (page.upload(buffer, (size_t)num_rows * ips_plain_stride(t)) &&
        ips::ok(ips_plain_pred(page.get(), num_rows, t, op, lits, n_lits,
                               (ips_semantics)IPS_PLAIN_SEMANTICS, bm.as<uint64_t>(), nullptr),
                "ips_plain_pred"))
      bm.download(words.data(), words.size() * 8);
    ips::append_bits(out, words, 0, num_rows, num_rows);
  }
};

}  // namespace impala
