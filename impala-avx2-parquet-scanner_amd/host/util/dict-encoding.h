// util/dict-encoding.h (MI355X facade) -- DictEncoder<T> / DictDecoder<T> with the public
// interface of the reference (dict-encoding.h:32-232) for the numeric types on the scan path.
//
//   DictEncoder<T>   host hash/sort logic (the dictionary is tiny, <= 40000 entries, :157);
//                    WriteDict sorts ascending (:393-406), WriteData remaps the buffered indices
//                    and bit-slices them with one ips_fle_encode launch (:408-423)
//   DictDecoder<T>   ips_dict_open keeps the sorted dictionary on the host (literal -> code
//                    translation, :461-541, is a binary search there) and in HBM (gathers);
//                    SetData strips the 1-byte code width and hands the FLE blocks to the
//                    FleDecoder facade (:185-192)
#pragma once
#include <algorithm>
#include <map>
#include <memory>
#include <vector>

#include "../exec/parquet-common.h"
#include "bit-util.h"
#include "fle-encoding.h"

namespace impala {

class DictEncoderBase {
 public:
  virtual ~DictEncoderBase() {}
  virtual void WriteDict(uint8_t* buffer) = 0;
  virtual int num_entries() const = 0;
  void ClearIndices() { buffered_indices_.clear(); }
  int dict_encoded_size() { return dict_encoded_size_; }

  // minimum width for the buffered indices: 0 / 1 / ceil(log2 D), dict-encoding.h:76-80
  int bit_width() const { return ips_dict_bit_width(num_entries()); }

  // [uint8 bit_width][FLE blocks]; returns bytes written or -1, dict-encoding.h:408-423
  int WriteData(uint8_t* buffer, int buffer_len) { return WriteData(buffer, buffer_len, buffered_indices_); }

  // One data page of a chunk whose dictionary is complete: the page's insertion-order indices
  // (the table writer keeps them per page until Flush(), hdfs-parquet-table-writer.cc:434-464,
  // because the sorted codes exist only after WriteDict), dict-encoding.h:425-447.  The header
  // byte is the width the codes are really packed with -- the whole dictionary's -- where the
  // reference stores the page's own Log2(max index + 1).
  int WriteData(uint8_t* buffer, int buffer_len, const std::vector<int>& node_indices) {
    const int bw = bit_width();
    const int64_t need = 1 + ips_fle_encoded_bytes((int64_t)node_indices.size(), bw);
    if (need > buffer_len) return -1;
    *buffer = (uint8_t)bw;
    if (node_indices.empty() || bw == 0) return 1;
    std::vector<uint8_t> blocks((size_t)need - 1);
    FleEncoder encoder(blocks.data(), (int)blocks.size(), bw);
    for (int index : node_indices)
      if (!encoder.Put((uint64_t)to_sorted_indice_[(size_t)index])) return -1;
    const int len = encoder.Flush();
    memcpy(buffer + 1, blocks.data(), (size_t)len);  // page payload starts at an odd address
    return 1 + len;
  }
  const std::vector<int>& buffered_indices() const { return buffered_indices_; }

 protected:
  DictEncoderBase() : dict_encoded_size_(0) {}
  std::vector<int> buffered_indices_;   // insertion-order indices
  std::vector<int> to_sorted_indice_;   // insertion index -> sorted code (filled by WriteDict)
  int dict_encoded_size_;
};

template <typename T>
class DictEncoder : public DictEncoderBase {
 public:
  DictEncoder(void* /*MemPool*/ = nullptr, int encoded_value_size = -1)
      : encoded_value_size_(encoded_value_size) {}

  // returns the bytes added to the dictionary page, or -1 when the 40000-entry cap is hit
  int Put(const T& value) {
    auto it = index_of_.find(value);
    if (it != index_of_.end()) {
      buffered_indices_.push_back(it->second);
      return 0;
    }
    if (values_.size() >= 40000) return -1;  // Node::INVALID_INDEX, dict-encoding.h:157
    const int idx = (int)values_.size();
    index_of_.emplace(value, idx);
    values_.push_back(value);
    buffered_indices_.push_back(idx);
    const int added = ParquetPlainEncoder::ByteSize(value);
    dict_encoded_size_ += added;
    return added;
  }

  virtual void WriteDict(uint8_t* buffer) {
    std::vector<int> order(values_.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return values_[(size_t)a] < values_[(size_t)b]; });
    to_sorted_indice_.resize(values_.size());
    for (size_t i = 0; i < order.size(); ++i) {
      to_sorted_indice_[(size_t)order[i]] = (int)i;
      buffer += ParquetPlainEncoder::Encode(buffer, encoded_value_size_, values_[(size_t)order[i]]);
    }
  }

  virtual int num_entries() const { return (int)values_.size(); }

  // Bulk form of Put x n + WriteDict + WriteData for a column chunk that is already on the device
  // (n PLAIN slots): hashing, remap and bit-slicing run on the GPU (ips_dict_encode).  dict_page
  // and data_page are resized to the exact page sizes; false = over the 40000-entry cap, in
  // which case the writer falls back to PLAIN as it does when Put() returns -1.
  static bool EncodeColumn(const void* d_values, int64_t n, std::vector<uint8_t>* dict_page,
                           std::vector<uint8_t>* data_page, ips_stream stream = nullptr) {
    const ips_type type = IpsTypeOf<T>::value;
    dict_page->assign((size_t)40000 * (size_t)ips_plain_stride(type), 0);
    ips::DeviceBuffer blocks((size_t)ips_fle_encoded_bytes(n, 16) + 16);
    ips::DeviceBuffer workspace(ips_dict_encode_workspace_bytes(n));
    int64_t dict_len = 0;
    int bw = 0;
    const ips_status st = ips_dict_encode(d_values, n, type, dict_page->data(), (int64_t)dict_page->size(),
                                          &dict_len, &bw, blocks.get(), workspace.get(), stream);
    if (st == IPS_ERR_UNSUPPORTED) return false;
    if (!ips::ok(st, "ips_dict_encode")) return false;
    dict_page->resize((size_t)dict_len);
    const int64_t len = ips_fle_encoded_bytes(n, bw);
    data_page->assign((size_t)len + 1, 0);
    (*data_page)[0] = (uint8_t)bw;
    if (len > 0 && bw > 0) {
      std::vector<uint8_t> tmp((size_t)len);
      if (!ips::ok(ips_memcpy_d2h(tmp.data(), blocks.get(), (size_t)len, stream), "d2h")) return false;
      if (!ips::ok(ips_stream_synchronize(stream), "sync")) return false;
      memcpy(data_page->data() + 1, tmp.data(), (size_t)len);
    }
    return true;
  }

 private:
  std::vector<T> values_;
  std::map<T, int> index_of_;
  int encoded_value_size_;
};

class DictDecoderBase {
 public:
  // first byte = code bit width, rest = FLE blocks of the codes, dict-encoding.h:185-192
  void SetData(uint8_t* buffer, int buffer_len) {
    bit_width_ = *buffer;
    data_decoder_.reset(new FleDecoder(buffer + 1, buffer_len - 1, bit_width_));
  }
  virtual ~DictDecoderBase() {}
  virtual int num_entries() const = 0;

 protected:
  std::unique_ptr<FleDecoder> data_decoder_;
  int bit_width_ = 0;
};

template <typename T>
class DictDecoder : public DictDecoderBase {
 public:
  DictDecoder(uint8_t* dict_buffer, int dict_len, int /*fixed_len_size*/) : dict_(nullptr) {
    ips::ok(ips_dict_open(dict_buffer, dict_len, IpsTypeOf<T>::value, &dict_), "ips_dict_open");
    const int slot = ips_plain_stride(IpsTypeOf<T>::value);
    for (int off = 0; off + slot <= dict_len; off += slot) {  // PLAIN-decode, :449-459
      T v;
      ParquetPlainEncoder::Decode(dict_buffer + off, -1, &v);
      values_.push_back(v);
    }
  }
  ~DictDecoder() { ips_dict_close(dict_); }
  DictDecoder(const DictDecoder&) = delete;
  DictDecoder& operator=(const DictDecoder&) = delete;

  virtual int num_entries() const { return (int)values_.size(); }

  bool GetValue(T* value) {
    int index;
    if (!data_decoder_ || !data_decoder_->Get(&index)) return false;
    if (index < 0 || (size_t)index >= values_.size()) return false;  // :316
    *value = values_[(size_t)index];
    return true;
  }
  bool GetValue(T* value, int skip_rows) {
    int index;
    if (!data_decoder_ || !data_decoder_->Get(&index, skip_rows)) return false;
    if (index < 0 || (size_t)index >= values_.size()) return false;
    *value = values_[(size_t)index];
    return true;
  }
  bool SkipValue(int skip_rows) { return data_decoder_ && data_decoder_->Skip(skip_rows); }

  void Eq(int64_t n, SkipBitset& b, T& val) { Pred(IPS_OP_EQ, n, b, &val, 1); }
  void Gt(int64_t n, SkipBitset& b, T& val) { Pred(IPS_OP_GT, n, b, &val, 1); }
  void Lt(int64_t n, SkipBitset& b, T& val) { Pred(IPS_OP_LT, n, b, &val, 1); }
  void Ge(int64_t n, SkipBitset& b, T& val) { Pred(IPS_OP_GE, n, b, &val, 1); }
  void Le(int64_t n, SkipBitset& b, T& val) { Pred(IPS_OP_LE, n, b, &val, 1); }
  void In(int64_t n, SkipBitset& b, std::vector<T>& vals) {
    if (vals.empty()) { b.resize(b.size() + (size_t)n, false); return; }
    Pred(IPS_OP_IN, n, b, vals.data(), (int)vals.size());
  }

  // facade extras
  const ips_dict* handle() const { return dict_; }
  FleDecoder* codes() { return data_decoder_.get(); }
  int code_bit_width() const { return bit_width_; }

 private:
  // literal -> code (host binary search inside the library), then the FLE predicate on codes;
  // constant answers use resize(n, value) exactly like dict-encoding.h:466,476,478
  void Pred(ips_op op, int64_t num_rows, SkipBitset& out, const T* lits, int n_lits) {
    ips_xl_kind kind = IPS_XL_ALL_FALSE;
    ips_op fle_op = op;
    int n_codes = 0;
    std::vector<uint64_t> codes((size_t)n_lits);
    if (!dict_ || !ips::ok(ips_dict_translate(dict_, op, lits, n_lits, &kind, &fle_op, codes.data(),
                                              &n_codes), "ips_dict_translate"))
      kind = IPS_XL_ALL_FALSE;
    if (kind == IPS_XL_ALL_FALSE) { out.resize(out.size() + (size_t)num_rows, false); return; }
    if (kind == IPS_XL_ALL_TRUE) { out.resize(out.size() + (size_t)num_rows, true); return; }
    codes.resize((size_t)n_codes);
    switch (fle_op) {
      case IPS_OP_EQ: data_decoder_->Eq(num_rows, out, codes[0]); break;
      case IPS_OP_LT: data_decoder_->Lt(num_rows, out, codes[0]); break;
      case IPS_OP_LE: data_decoder_->Le(num_rows, out, codes[0]); break;
      case IPS_OP_GT: data_decoder_->Gt(num_rows, out, codes[0]); break;
      case IPS_OP_GE: data_decoder_->Ge(num_rows, out, codes[0]); break;
      default: data_decoder_->In(num_rows, out, codes); break;
    }
  }

  ips_dict* dict_;
  std::vector<T> values_;
};

}  // namespace impala
