// exec/hdfs-parquet-scanner.h (MI355X facade) -- the slice of HdfsParquetScanner that sits on the
// hot path (hdfs-parquet-scanner.h:91-102, .cc:305-567, 1006-1038, 1078-1182, 1825-1907):
//   Eq/Lt/Le/Gt/Ge/In<T>(col_idx, num_rows, bitset, literal)   -> ColumnReader<T> dispatch
//   CreateSimplePredicates / EvalSimplePredicates               1024-row batches, conjunct AND
//   bitmap -> skip list, ReadValue(skip) / SkipValue            late materialisation
// Everything the reference obtains from the absent Impala runtime (footer, thrift page headers,
// DiskIoMgr streams, decompression, tuples) is replaced by AddDictionaryColumn / AddPlainColumn,
// which take the uncompressed page payloads exactly as ReadDataPage would hand them to
// InitDataPage (.cc:882-916): [int32 n_def_bytes][FLE def levels] (OPTIONAL columns only) followed
// by [uint8 code_width][FLE codes] or PLAIN values.
#pragma once
#include <string.h>

#include <memory>
#include <type_traits>
#include <vector>

#include <string>

#include "../exprs/simple-predicates.h"
#include "../util/dict-encoding.h"
#include "parquet-common.h"
#include "parquet-page-header.h"
#include "parquet-column-chunk.h"

namespace impala {

class HdfsParquetScanner {
 public:
  class BaseColumnReader {
   public:
    virtual ~BaseColumnReader() {}
    virtual bool SkipValue(int skip_rows) = 0;
    virtual bool LowerLeaf(int op, const void* lits, int n_lits, ips_column* col, ips_node* node) = 0;
    // facade extra (AssembleRowsFused): materialise the rows selected by d_selection (num_rows rows
    // of the current page) on the device and describe them as one ips_tuple_column.  'keep' holds
    // the device buffers until the tuples have been assembled.
    struct Materialised {
      std::vector<std::unique_ptr<ips::DeviceBuffer>> keep;
      ips::DeviceBuffer& add(size_t bytes) {
        keep.emplace_back(new ips::DeviceBuffer(bytes));
        return *keep.back();
      }
      ips::DeviceBuffer counts;  // batch counts of the selection (shared by all REQUIRED columns)
      std::vector<const int64_t*> bad_index;  // device flags: a selected code outside its dictionary
    };
    virtual bool SelectInto(int64_t num_rows, const uint64_t* d_selection, ips_tuple_column* col,
                            Materialised* m) = 0;
    // facade extra (EvalSimplePredicatesChunks): the whole column chunk -- the current page and the
    // queued ones -- as an ips_chunk; 'keep' owns the device copies of the queued pages
    struct ChunkHolder {
      ips_chunk* chunk = nullptr;
      std::vector<std::unique_ptr<ips::DeviceBuffer>> keep;
      ChunkHolder() {}
      ChunkHolder(const ChunkHolder&) = delete;
      ChunkHolder& operator=(const ChunkHolder&) = delete;
      ~ChunkHolder() { if (chunk) ips_chunk_close(chunk); }
      const void* upload(const void* h, size_t bytes) {
        keep.emplace_back(new ips::DeviceBuffer());
        return keep.back()->upload(h, bytes) ? keep.back()->get() : nullptr;
      }
    };
    virtual bool OpenChunk(ChunkHolder* h) = 0;
    virtual const ips_dict* dict_handle() const = 0;  // NULL: a PLAIN column
    virtual int slot_width() const = 0;               // bytes of a materialised value: 4 or 8
    // rows of the current page (DataPageHeader.num_values): what the device buffers of the page hold
    int64_t page_rows() const { return page_rows_; }
    int64_t num_buffered_values() const { return num_buffered_values_; }
    void consume(int64_t n) { num_buffered_values_ -= n; }
    int max_def_level() const { return max_def_level_; }
    // further data pages of the column chunk, consumed in order when the current one is used up
    // (ReadDataPage / InitDataPage, .cc:766-916)
    struct PendingPage { uint8_t* data; int len; int64_t num_values; };
    void AddPendingPage(uint8_t* data, int len, int64_t num_values) {
      pending_pages_.push_back(PendingPage{data, len, num_values});
    }
    bool NextPage() {
      if (pending_pages_.empty()) return false;
      const PendingPage pg = pending_pages_.front();
      pending_pages_.erase(pending_pages_.begin());
      return InitDataPage(pg.data, pg.len, pg.num_values);
    }
    // Data-page payload framing (InitDataPage, .cc:882-916): OPTIONAL columns start with
    // [int32 n_def_bytes][FLE definition levels], then [uint8 code_width][FLE codes].  The
    // reference bounds-checks the 4-byte read (ReadWriteUtil::Read); a short or corrupt page must
    // not make the decoders read outside the buffer.
    static bool SplitDataPage(uint8_t* data, int len, int max_def_level, uint8_t** def_levels,
                              int* n_def_bytes, uint8_t** codes, int* codes_len) {
      return parquet::SplitDataPage(data, len, max_def_level, def_levels, n_def_bytes, codes, codes_len);
    }

   protected:
    virtual bool InitDataPage(uint8_t* data, int len, int64_t num_values) = 0;
    int64_t num_buffered_values_ = 0;
    int64_t page_rows_ = 0;
    int max_def_level_ = 0;
    std::vector<PendingPage> pending_pages_;
    // decompressed pages / the dictionary of a chunk read through AddColumnChunk (the reference's
    // dictionary_pool_ and decompressed_data_pool_, .cc:831-840, 868)
    std::vector<std::unique_ptr<std::vector<uint8_t>>> owned_pages_;
    friend class HdfsParquetScanner;
  };

  // Per column type reader, hdfs-parquet-scanner.cc:305-567
  template <typename T>
  class ColumnReader : public BaseColumnReader {
   public:
    ~ColumnReader() { if (in_set_) ips_inset_close(in_set_); }
    // ColumnReader::IntersectBitset, .cc:326-331 (the device twin is ips_bitmap_expand)
    static void IntersectBitset(SkipBitset& root_bitset, SkipBitset& sub_bitset) {
      int64_t j = -1;
      for (size_t i = 0; i < root_bitset.size(); ++i)
        if (root_bitset[i]) root_bitset.set(i, sub_bitset[(size_t)++j]);
    }

    // one body for Eq..In (.cc:333-445): dictionary pages go through DictDecoder<T>, nullable
    // ones through def_levels->Eq(max_def) + IntersectBitset; PLAIN pages through the encoder
    template <typename L>
    void Pred(int op, int64_t num_rows, SkipBitset& skip_bitset, L& lit) {
      if (dict_decoder_) {
        if (max_def_level_ == 0) { Call(op, *dict_decoder_, num_rows, skip_bitset, lit); return; }
        fle_def_levels_->Eq(num_rows, skip_bitset, (uint64_t)max_def_level_);
        SkipBitset data_bitset;
        Call(op, *dict_decoder_, (int64_t)skip_bitset.count(), data_bitset, lit);
        IntersectBitset(skip_bitset, data_bitset);
      } else {
        PlainCall(op, num_rows, skip_bitset, lit);
      }
    }

    // ReadValue(pool, tuple, skip_rows) -> ReadSlot(skip) (.cc:1006-1027, 515-531).  *is_null
    // is set for a NULL row of an OPTIONAL column (ReadDefinitionLevel, .cc:927-979).
    bool ReadValue(T* slot, int skip_rows, bool* is_null = nullptr) {
      if (is_null) *is_null = false;
      if (max_def_level_ > 0) {
        int data_skip = 0;
        for (int i = 0; i < skip_rows; ++i) {
          int def;
          if (!fle_def_levels_->Get(&def)) return false;
          if (def == max_def_level_) ++data_skip;
        }
        int def;
        if (!fle_def_levels_->Get(&def)) return false;
        if (def != max_def_level_) {
          if (is_null) *is_null = true;
          return data_skip == 0 || SkipData(data_skip);
        }
        return ReadData(slot, data_skip);
      }
      return ReadData(slot, skip_rows);
    }

    virtual bool SkipValue(int skip_rows) {  // .cc:1029-1038, 533-547
      if (max_def_level_ > 0) {
        int data_skip = 0;
        for (int i = 0; i < skip_rows; ++i) {
          int def;
          if (!fle_def_levels_->Get(&def)) return false;
          if (def == max_def_level_) ++data_skip;
        }
        return data_skip == 0 || SkipData(data_skip);
      }
      return SkipData(skip_rows);
    }

    virtual bool LowerLeaf(int op, const void* lits, int n_lits, ips_column* col, ips_node* node) {
      memset(col, 0, sizeof(*col));
      memset(node, 0, sizeof(*node));
      node->kind = IPS_NODE_LEAF;
      if (max_def_level_ > 0 && !dict_decoder_) return false;  // the PLAIN branch has no NULL handling (.cc:346-348)
      if (dict_decoder_ && op == IPS_OP_IN && n_lits > 16) {
        // InOperate keeps a vector of any length (simple-predicates.h:195-205): the codes of the
        // literals that are dictionary entries (dict-encoding.h:523-541) become a resident set, kept
        // with the reader for as long as the same literals come back
        const size_t bytes = (size_t)n_lits * sizeof(T);
        if (!in_set_ || in_set_key_.size() != bytes || memcmp(in_set_key_.data(), lits, bytes) != 0) {
          if (in_set_) ips_inset_close(in_set_);
          in_set_ = nullptr;
          if (!ips::ok(ips_dict_inset_open(dict_decoder_->handle(), lits, n_lits, &in_set_), "ips_dict_inset_open")) return false;
          in_set_key_.assign((const uint8_t*)lits, (const uint8_t*)lits + bytes);
        }
        col->encoding = IPS_COL_FLE;
        col->bit_width = dict_decoder_->code_bit_width();
        col->d_data = dict_decoder_->codes()->device_blocks();
        if (max_def_level_ > 0) {
          if (!fle_def_levels_ || !fle_def_levels_->usable()) return false;
          col->max_def_level = max_def_level_;
          col->d_def_levels = fle_def_levels_->device_blocks();
          col->def_bit_width = fle_def_levels_->bit_width();
          col->n_data_rows = dict_decoder_->codes()->rows_in_buffer();
        }
        node->op = IPS_OP_IN;
        node->n_consts = 0;
        node->inset = in_set_;
        return true;
      }
      if (n_lits > 16) return false;
      if (dict_decoder_) {
        ips_xl_kind kind; ips_op fle_op; int n_codes = 0;
        uint64_t codes[16];
        if (ips_dict_translate(dict_decoder_->handle(), (ips_op)op, lits, n_lits, &kind, &fle_op,
                               codes, &n_codes) != IPS_OK)
          return false;
        col->encoding = IPS_COL_FLE;
        col->bit_width = dict_decoder_->code_bit_width();
        col->d_data = dict_decoder_->codes()->device_blocks();
        if (max_def_level_ > 0) {  // OPTIONAL: levels + data rows, evaluated by the nullable leaf
          if (!fle_def_levels_ || !fle_def_levels_->usable()) return false;
          col->max_def_level = max_def_level_;
          col->d_def_levels = fle_def_levels_->device_blocks();
          col->def_bit_width = fle_def_levels_->bit_width();
          col->n_data_rows = dict_decoder_->codes()->rows_in_buffer();
        }
        // constant answers become range predicates that are always false / true on codes
        if (kind == IPS_XL_ALL_FALSE) { node->op = IPS_OP_LT; node->n_consts = 1; node->consts[0] = 0; return true; }
        if (kind == IPS_XL_ALL_TRUE) { node->op = IPS_OP_GE; node->n_consts = 1; node->consts[0] = 0; return true; }
        node->op = fle_op;
        node->n_consts = n_codes;
        for (int i = 0; i < n_codes; ++i) node->consts[i] = codes[i];
        return true;
      }
      col->encoding = IPS_COL_PLAIN;
      col->type = IpsTypeOf<T>::value;
      if (!EnsurePlainResident()) return false;
      col->d_data = plain_dev_.get();
      if (IPS_PLAIN_SEMANTICS == IPS_SEM_REFERENCE) {  // literal OP x == x OP' literal (quirk Q1)
        if (op == IPS_OP_LT) op = IPS_OP_GT; else if (op == IPS_OP_GT) op = IPS_OP_LT;
        else if (op == IPS_OP_LE) op = IPS_OP_GE; else if (op == IPS_OP_GE) op = IPS_OP_LE;
        else if (op == IPS_OP_IN) return false;  // no reference behaviour (empty body)
      }
      node->op = op;
      node->n_consts = n_lits;
      for (int i = 0; i < n_lits; ++i) memcpy(&node->consts[i], (const T*)lits + i, sizeof(T));
      return true;
    }

    // ReadValue(skip) of every selected row at once (.cc:1151-1181, 1006-1027): dictionary pages
    // through ips_dict_select / ips_dict_select_nullable, PLAIN pages through ips_plain_select.
    virtual bool SelectInto(int64_t num_rows, const uint64_t* d_selection, ips_tuple_column* col,
                            Materialised* m) {
      memset(col, 0, sizeof(*col));
      const ips_type t = IpsTypeOf<T>::value;
      const int vw = ips_plain_stride(t);
      col->value_width = vw;
      const int64_t nb = (num_rows + IPS_BATCH_ROWS - 1) / IPS_BATCH_ROWS;
      if (dict_decoder_ && max_def_level_ > 0) {
        if (!fle_def_levels_ || !fle_def_levels_->usable()) return false;
        const int64_t n_data = dict_decoder_->codes()->rows_in_buffer();
        ips::DeviceBuffer& dense = m->add((size_t)std::max<int64_t>(n_data, 16) * vw);
        ips::DeviceBuffer& flags = m->add((size_t)((num_rows + 63) / 64 + 2) * 8);
        ips::DeviceBuffer& cnt = m->add(32);
        m->bad_index.push_back(cnt.as<int64_t>() + 2);
        ips::DeviceBuffer& ws = m->add(ips_select_nullable_workspace_bytes(num_rows, n_data, vw));
        if (!ips::ok(ips_dict_select_nullable(dict_decoder_->handle(), fle_def_levels_->device_blocks(),
                                              fle_def_levels_->bit_width(), max_def_level_, num_rows,
                                              dict_decoder_->codes()->device_blocks(), n_data,
                                              dict_decoder_->code_bit_width(), d_selection, dense.get(),
                                              flags.as<uint64_t>(), cnt.as<int64_t>(), ws.get(), nullptr),
                     "ips_dict_select_nullable"))
          return false;
        col->d_dense_values = dense.get();
        col->d_nonnull_flags = flags.as<uint64_t>();
        return true;
      }
      ips::DeviceBuffer& values = m->add((size_t)nb * IPS_BATCH_ROWS * vw);
      if (!m->counts.get() && !m->counts.resize((size_t)nb * 4)) return false;
      ips_status st;
      if (dict_decoder_) {
        st = ips_dict_select(dict_decoder_->handle(), dict_decoder_->codes()->device_blocks(), num_rows,
                             dict_decoder_->code_bit_width(), d_selection, values.get(),
                             m->counts.as<uint32_t>(), nullptr);
      } else {
        if (!EnsurePlainResident()) return false;
        st = ips_plain_select(plain_dev_.get(), num_rows, t, d_selection, values.get(), m->counts.as<uint32_t>(), nullptr);
      }
      if (!ips::ok(st, "select")) return false;
      col->d_batch_values = values.get();
      return true;
    }

    virtual const ips_dict* dict_handle() const { return dict_decoder_ ? dict_decoder_->handle() : nullptr; }
    virtual int slot_width() const { return ips_plain_stride(IpsTypeOf<T>::value); }

    // the current page (already resident) followed by the queued ones (uploaded here)
    virtual bool OpenChunk(ChunkHolder* h) {
      std::vector<ips_chunk_page> pages;
      const int stride = ips_plain_stride(IpsTypeOf<T>::value);
      if (dict_decoder_) {
        if (max_def_level_ > 1) return false;  // flat schemas only (hdfs-parquet-scanner.cc:338-345)
        ips_chunk_page pg;
        memset(&pg, 0, sizeof(pg));
        pg.d_data = dict_decoder_->codes()->device_blocks();
        pg.n_rows = page_rows_;
        pg.bit_width = dict_decoder_->code_bit_width();
        if (max_def_level_ > 0) {
          if (!fle_def_levels_ || !fle_def_levels_->usable()) return false;
          pg.d_def_levels = fle_def_levels_->device_blocks();
          pg.n_data_rows = dict_decoder_->codes()->rows_in_buffer();
        }
        if (page_rows_ > 0) pages.push_back(pg);
        for (const PendingPage& pp : pending_pages_) {
          uint8_t *def = nullptr, *codes = nullptr;
          int n_def_bytes = 0, codes_len = 0;
          if (!SplitDataPage(pp.data, pp.len, max_def_level_, &def, &n_def_bytes, &codes, &codes_len) ||
              parquet::CheckDictDataPage(def, n_def_bytes, codes, codes_len, max_def_level_, pp.num_values))
            return ips::ok(IPS_ERR_INVALID_ARG, "OpenChunk: truncated or corrupt data page");
          if (pp.num_values == 0) continue;
          memset(&pg, 0, sizeof(pg));
          pg.n_rows = pp.num_values;
          pg.bit_width = codes[0];
          const int64_t blocks = (codes_len - 1) / (8 * pg.bit_width);
          pg.d_data = h->upload(codes + 1, (size_t)(blocks * 8 * pg.bit_width));
          if (!pg.d_data) return false;
          if (max_def_level_ > 0) {
            pg.d_def_levels = h->upload(def, (size_t)(((pp.num_values + 63) / 64) * 8));
            if (!pg.d_def_levels) return false;
            pg.n_data_rows = blocks * 64;
          }
          pages.push_back(pg);
        }
        return ips::ok(ips_chunk_open(pages.data(), (int)pages.size(), IPS_COL_FLE, IpsTypeOf<T>::value, max_def_level_,
                                      &h->chunk), "ips_chunk_open");
      }
      if (max_def_level_ > 0) return false;  // the PLAIN branch has no NULL handling (.cc:346-348)
      ips_chunk_page pg;
      memset(&pg, 0, sizeof(pg));
      if (plain_rows_ > 0) {
        if (!EnsurePlainResident()) return false;
        pg.d_data = plain_dev_.get();
        pg.n_rows = plain_rows_;
        pages.push_back(pg);
      }
      for (const PendingPage& pp : pending_pages_) {
        if (pp.num_values == 0) continue;
        if ((int64_t)pp.len < pp.num_values * stride) return ips::ok(IPS_ERR_INVALID_ARG, "OpenChunk: truncated PLAIN page");
        memset(&pg, 0, sizeof(pg));
        pg.n_rows = pp.num_values;
        pg.d_data = h->upload(pp.data, (size_t)(pp.num_values * stride));
        if (!pg.d_data) return false;
        pages.push_back(pg);
      }
      return ips::ok(ips_chunk_open(pages.data(), (int)pages.size(), IPS_COL_PLAIN, IpsTypeOf<T>::value, 0, &h->chunk),
                     "ips_chunk_open");
    }

   protected:
    // [int32 n_def_bytes][def levels]? [uint8 width][codes] for dictionary columns, raw slots for
    // PLAIN ones; the dictionary stays (one dictionary page per column chunk)
    virtual bool InitDataPage(uint8_t* data, int len, int64_t num_values) {
      num_buffered_values_ = num_values;
      if (dict_decoder_) {
        uint8_t *def = nullptr, *codes = nullptr;
        int n_def_bytes = 0, codes_len = 0;
        page_rows_ = num_values;
        if (!SplitDataPage(data, len, max_def_level_, &def, &n_def_bytes, &codes, &codes_len) ||
            parquet::CheckDictDataPage(def, n_def_bytes, codes, codes_len, max_def_level_, num_values)) {
          ips::ok(IPS_ERR_INVALID_ARG, "InitDataPage: truncated or corrupt data page");
          num_buffered_values_ = page_rows_ = 0;
          return false;
        }
        if (max_def_level_ > 0)  // .cc:882-901
          fle_def_levels_.reset(new FleDecoder(def, n_def_bytes, BitUtil::Log2((uint64_t)max_def_level_ + 1)));
        dict_decoder_->SetData(codes, codes_len);
        return true;
      }
      data_ = plain_begin_ = data;
      plain_rows_ = page_rows_ = num_values;
      T dummy;
      data_end_ = data + num_values * ParquetPlainEncoder::ByteSize(dummy);
      plain_dev_.release();
      plain_preds_.clear();
      return true;
    }

   private:
    friend class HdfsParquetScanner;
    template <typename D, typename L>
    static void Call(int op, D& d, int64_t n, SkipBitset& b, L& lit);
    template <typename L>
    void PlainCall(int op, int64_t n, SkipBitset& b, L& lit);

    bool ReadData(T* slot, int skip) {
      if (dict_decoder_) return dict_decoder_->GetValue(slot, skip);
      data_ += ParquetPlainEncoder::Decode(data_, -1, slot, skip);
      return data_ <= data_end_;
    }
    bool SkipData(int skip) {
      if (dict_decoder_) return dict_decoder_->SkipValue(skip);
      T dummy;
      data_ += ParquetPlainEncoder::Skip(data_, -1, &dummy, skip);
      return data_ <= data_end_;
    }

    std::unique_ptr<DictDecoder<T>> dict_decoder_;
    std::unique_ptr<FleDecoder> fle_def_levels_;
    ips_inset* in_set_ = nullptr;        // the long IN list last lowered on this column (LowerLeaf)
    std::vector<uint8_t> in_set_key_;
    uint8_t* data_ = nullptr;        // PLAIN: current position (the predicates start here)
    uint8_t* data_end_ = nullptr;
    uint8_t* plain_begin_ = nullptr;
    int64_t plain_rows_ = 0;
    ips::DeviceBuffer plain_dev_;     // the PLAIN page, uploaded once per InitDataPage
    ips::PredCache plain_preds_;      // whole-page bitmaps per (op, literals)
    bool EnsurePlainResident() {
      return plain_dev_.get() ||
             plain_dev_.upload(plain_begin_, (size_t)plain_rows_ * ips_plain_stride(IpsTypeOf<T>::value));
    }
  };

  HdfsParquetScanner() {}
  ~HdfsParquetScanner() { for (SimplePredicate* p : owned_) delete p; }

  // ---- page plumbing (stands in for ReadDataPage / InitDataPage) ----
  // data_page: [int32 n_def_bytes][def levels]? [uint8 width][codes]; max_def_level 0 = REQUIRED
  template <typename T>
  int AddDictionaryColumn(uint8_t* dict_page, int dict_len, uint8_t* data_page, int data_len,
                          int64_t num_values, int max_def_level = 0) {
    uint8_t *def = nullptr, *codes = nullptr;
    int n_def_bytes = 0, codes_len = 0;
    if (!BaseColumnReader::SplitDataPage(data_page, data_len, max_def_level, &def, &n_def_bytes, &codes,
                                         &codes_len) ||
        parquet::CheckDictDataPage(def, n_def_bytes, codes, codes_len, max_def_level, num_values)) {
      ips::ok(IPS_ERR_INVALID_ARG, "AddDictionaryColumn: truncated or corrupt data page");
      return -1;
    }
    auto* r = new ColumnReader<T>();
    r->num_buffered_values_ = r->page_rows_ = num_values;
    r->max_def_level_ = max_def_level;
    if (max_def_level > 0)  // .cc:882-901
      r->fle_def_levels_.reset(new FleDecoder(def, n_def_bytes, BitUtil::Log2((uint64_t)max_def_level + 1)));
    r->dict_decoder_.reset(new DictDecoder<T>(dict_page, dict_len, -1));
    r->dict_decoder_->SetData(codes, codes_len);
    column_readers_.emplace_back(r);
    return (int)column_readers_.size() - 1;
  }

  template <typename T>
  int AddPlainColumn(uint8_t* page, int64_t num_values) {
    auto* r = new ColumnReader<T>();
    r->num_buffered_values_ = r->page_rows_ = num_values;
    r->data_ = r->plain_begin_ = page;
    r->plain_rows_ = num_values;
    T dummy;
    r->data_end_ = page + num_values * ParquetPlainEncoder::ByteSize(dummy);
    column_readers_.emplace_back(r);
    return (int)column_readers_.size() - 1;
  }

  // ---- page container: BaseColumnReader::ReadDataPage, .cc:730-924 -------------------------
  // A column chunk exactly as it sits in the file: a run of [thrift PageHeader][page body] with
  // the bodies compressed by the chunk's codec (metadata_->codec); num_values is the chunk's
  // ColumnMetaData.num_values.  The walk does what ReadDataPage does per page -- deserialize the
  // header, skip page types it does not know, decompress, take the one dictionary page (PLAIN /
  // PLAIN_DICTIONARY / FLE_DICTIONARY, entry count checked against the header), check the data
  // page's encodings -- but for the whole chunk at once: the first data page initialises the
  // column reader, the others are queued (AddDataPage).  Returns the column index, or -1 with
  // *error set (the reference's Status / parse_status_).
  template <typename T>
  int AddColumnChunk(const uint8_t* chunk, int64_t chunk_len, int64_t num_values, int codec,
                     int max_def_level, std::string* error) {
    auto fail = [&](const char* msg) {
      if (error) *error = msg;
      ips::ok(IPS_ERR_INVALID_ARG, msg);
      return -1;
    };
    // the host-only part (headers, codecs, page bookkeeping) lives in parquet-column-chunk.h
    parquet::ColumnChunkPages pages;
    if (const char* msg = parquet::WalkColumnChunk(chunk, chunk_len, num_values, codec, max_def_level,
                                                   ips_plain_stride(IpsTypeOf<T>::value), &pages))
      return fail(msg);
    auto& owned = pages.owned;
    std::vector<uint8_t>* dict_values = pages.dict_values;
    auto& data_pages = pages.data_pages;
    using DataPage = parquet::ChunkDataPage;
    const bool dict_coded = pages.dict_coded;
    int idx;
    if (dict_coded) {
      if (!dict_values) return fail("File corrupt. Missing dictionary page.");  // .cc:458-460
      idx = AddDictionaryColumn<T>(dict_values->data(), (int)dict_values->size(), data_pages[0].bytes->data(),
                                   (int)data_pages[0].bytes->size(), data_pages[0].num_values, max_def_level);
      if (idx < 0) return fail("truncated or corrupt data page");
      for (size_t i = 1; i < data_pages.size(); ++i)
        AddDataPage(idx, data_pages[i].bytes->data(), (int)data_pages[i].bytes->size(), data_pages[i].num_values);
    } else {
      // PLAIN pages: the values follow the level bytes (data_ += num_definition_bytes, .cc:916-917);
      // the PLAIN branch of the predicates ignores the levels (quirk Q3)
      // (an OPTIONAL PLAIN page stores its non-NULL values only; with a NULL in it the vectorised
      // PLAIN path, which never looks at the levels (.cc:346-348, quirk Q3), would read other rows)
      bool has_nulls = false;
      auto values_of = [&](const DataPage& pg, uint8_t** v) -> bool {
        int64_t stored = 0;
        if (!parquet::PlainPageValues(pg.bytes->data(), (int64_t)pg.bytes->size(), max_def_level, pg.num_values,
                                      ips_plain_stride(IpsTypeOf<T>::value), v, &stored))
          return false;
        has_nulls = has_nulls || stored != pg.num_values;
        return true;
      };
      uint8_t* v = nullptr;
      for (size_t i = 0; i < data_pages.size(); ++i)
        if (!values_of(data_pages[i], &v)) return fail("truncated PLAIN data page");
      if (has_nulls)
        return fail("OPTIONAL PLAIN pages with NULLs need the row-at-a-time path (the vectorised PLAIN branch ignores definition levels)");
      if (!values_of(data_pages[0], &v)) return fail("truncated PLAIN data page");
      idx = AddPlainColumn<T>(v, data_pages[0].num_values);
      for (size_t i = 1; i < data_pages.size(); ++i) {
        if (!values_of(data_pages[i], &v)) return fail("truncated PLAIN data page");
        AddDataPage(idx, v, (int)(data_pages[i].num_values * ips_plain_stride(IpsTypeOf<T>::value)), data_pages[i].num_values);
      }
    }
    for (auto& b : owned) column_readers_[(size_t)idx]->owned_pages_.push_back(std::move(b));
    return idx;
  }

  // a further data page of column idx (same dictionary); pages are consumed in the order added
  void AddDataPage(int idx, uint8_t* data_page, int data_len, int64_t num_values) {
    column_readers_[(size_t)idx]->AddPendingPage(data_page, data_len, num_values);
  }

  // ---- hdfs-parquet-scanner.h:91-102 ----
  template <typename T> void Eq(int idx, int64_t n, SkipBitset& b, T& val) { reader<T>(idx)->Pred(IPS_OP_EQ, n, b, val); }
  template <typename T> void Lt(int idx, int64_t n, SkipBitset& b, T& val) { reader<T>(idx)->Pred(IPS_OP_LT, n, b, val); }
  template <typename T> void Le(int idx, int64_t n, SkipBitset& b, T& val) { reader<T>(idx)->Pred(IPS_OP_LE, n, b, val); }
  template <typename T> void Gt(int idx, int64_t n, SkipBitset& b, T& val) { reader<T>(idx)->Pred(IPS_OP_GT, n, b, val); }
  template <typename T> void Ge(int idx, int64_t n, SkipBitset& b, T& val) { reader<T>(idx)->Pred(IPS_OP_GE, n, b, val); }
  template <typename T> void In(int idx, int64_t n, SkipBitset& b, std::vector<T>& val) { reader<T>(idx)->Pred(IPS_OP_IN, n, b, val); }

  // conjunct list, .cc:1825-1835.  Nodes live as long as the scanner: every node is handed to
  // Own() once (RuntimeState::obj_pool() in the reference, scalar-fn-call.cc:947).
  void AddSimplePredicate(SimplePredicate* root) { if (root) simple_predicates_.push_back(root); }
  void ClearSimplePredicates() { simple_predicates_.clear(); }
  SimplePredicate* Own(SimplePredicate* p) { owned_.push_back(p); return p; }
  size_t num_simple_predicates() const { return simple_predicates_.size(); }

  // .cc:1837-1865: batch = min(1024, rows left in every column's page); AND over the conjuncts
  bool EvalSimplePredicates(SkipBitset& skip_bitset) {
    int64_t limit_rows = 1024;
    for (auto& c : column_readers_) {
      if (c->num_buffered_values() == 0 && !c->NextPage()) return false;  // ReadDataPage, .cc:1846-1850
      if (c->num_buffered_values() < limit_rows) limit_rows = c->num_buffered_values();
    }
    for (auto& c : column_readers_) c->consume(limit_rows);
    simple_predicates_[0]->GetBitset(this, limit_rows, skip_bitset);
    for (size_t i = 1; i < simple_predicates_.size(); ++i) {
      SkipBitset tmp_bitset;
      simple_predicates_[i]->GetBitset(this, limit_rows, tmp_bitset);
      skip_bitset &= tmp_bitset;
    }
    return ips::sticky_status() == IPS_OK && (int64_t)skip_bitset.size() == limit_rows;
  }

  // bitmap -> skip list, .cc:1134-1148: skip_rows[j] = zeros before the j-th set bit
  static void BitsetToSkipList(const SkipBitset& skip_bitset, std::vector<int>* skip_rows,
                               int* last_skip_rows) {
    skip_rows->clear();
    int start = 0;
    const int n = (int)skip_bitset.size();
    for (int i = 0; i < n; ++i)
      if (skip_bitset[(size_t)i]) { skip_rows->push_back(i - start); start = i + 1; }
    *last_skip_rows = n - start;
  }

  template <typename T>
  bool ReadValue(int idx, T* slot, int skip_rows, bool* is_null = nullptr) {
    return reader<T>(idx)->ReadValue(slot, skip_rows, is_null);
  }
  bool SkipValue(int idx, int skip_rows) { return column_readers_[(size_t)idx]->SkipValue(skip_rows); }

  // facade extra: the whole conjunct list over all rows of the pages in one ips_eval_program
  // call (REQUIRED and OPTIONAL dictionary columns, PLAIN columns).  bitmap_words:
  // ceil(num_rows/64) LSB-first words.
  // the fused calls hand num_rows straight to the device: never more than every column's current
  // page holds (a page header may lie about num_values; the per-batch path is bounded by
  // append_bits, the page checks of InitDataPage bound what was uploaded)
  bool RowsResident(int64_t num_rows) const {
    if (num_rows < 0) return false;
    for (auto& c : column_readers_)
      if (num_rows > c->page_rows()) return ips::ok(IPS_ERR_INVALID_ARG, "fused call: num_rows exceeds the rows of a column's current page");
    return true;
  }

  bool EvalSimplePredicatesFused(int64_t num_rows, std::vector<uint64_t>* bitmap_words) {
    if (!RowsResident(num_rows)) return false;
    lower_cols_.clear();
    std::vector<ips_node> program;
    for (size_t i = 0; i < simple_predicates_.size(); ++i) {
      if (!simple_predicates_[i]->Lower(this, &program)) return false;
      if (i > 0) { ips_node n; memset(&n, 0, sizeof(n)); n.kind = IPS_NODE_AND; program.push_back(n); }
    }
    bitmap_words->assign((size_t)((num_rows + 63) / 64), 0);
    ips::DeviceBuffer bm(bitmap_words->size() * 8);
    // temporaries of trees that keep several bitmaps alive come from the caller (no hidden
    // allocation inside the library); kept across calls
    const size_t ws_bytes = ips_program_workspace_bytes(program.data(), (int)program.size(), lower_cols_.data(),
                                                        (int)lower_cols_.size(), num_rows);
    if (ws_bytes > 0 && !program_workspace_.resize(ws_bytes)) return false;
    return ips::ok(ips_eval_program(program.data(), (int)program.size(), lower_cols_.data(),
                                    (int)lower_cols_.size(), num_rows, bm.as<uint64_t>(),
                                    ws_bytes ? program_workspace_.get() : nullptr, nullptr),
                   "ips_eval_program") &&
           bm.download(bitmap_words->data(), bitmap_words->size() * 8);
  }

  // facade extra: the conjunct list over EVERY page of the column chunks -- each reader's current
  // page and the ones queued behind it (AddColumnChunk / AddDataPage) -- in one
  // ips_eval_program_chunks call: the page loop of EvalSimplePredicates (.cc:1837-1855, batches cut at
  // every column's page end) runs inside the launches, pages of different columns need not align.
  // *num_rows = rows of the chunks; bitmap_words: ceil(rows / 64) LSB-first words.
  bool EvalSimplePredicatesChunks(std::vector<uint64_t>* bitmap_words, int64_t* num_rows) {
    lower_cols_.clear();
    chunk_readers_.clear();
    lower_chunks_ = true;
    std::vector<ips_node> program;
    bool lowered = true;
    for (size_t i = 0; i < simple_predicates_.size() && lowered; ++i) {
      lowered = simple_predicates_[i]->Lower(this, &program);
      if (lowered && i > 0) { ips_node n; memset(&n, 0, sizeof(n)); n.kind = IPS_NODE_AND; program.push_back(n); }
    }
    lower_chunks_ = false;
    if (!lowered || program.empty()) return false;
    std::vector<std::unique_ptr<BaseColumnReader::ChunkHolder>> holders;
    std::vector<const ips_chunk*> chunks;
    for (int idx : chunk_readers_) {
      holders.emplace_back(new BaseColumnReader::ChunkHolder());
      if (!column_readers_[(size_t)idx]->OpenChunk(holders.back().get())) return false;
      chunks.push_back(holders.back()->chunk);
    }
    const int64_t n = ips_chunk_num_rows(chunks[0]);
    for (const ips_chunk* c : chunks)
      if (ips_chunk_num_rows(c) != n) return ips::ok(IPS_ERR_INVALID_ARG, "EvalSimplePredicatesChunks: the column chunks hold different row counts");
    *num_rows = n;
    bitmap_words->assign((size_t)((n + 63) / 64), 0);
    if (n == 0) return true;
    ips::DeviceBuffer bm(bitmap_words->size() * 8);
    const size_t ws_bytes = ips_chunk_program_workspace_bytes(program.data(), (int)program.size(), chunks.data(), (int)chunks.size());
    if (ws_bytes > 0 && !program_workspace_.resize(ws_bytes)) return false;
    return ips::ok(ips_eval_program_chunks(program.data(), (int)program.size(), chunks.data(), (int)chunks.size(),
                                           bm.as<uint64_t>(), ws_bytes ? program_workspace_.get() : nullptr, nullptr),
                   "ips_eval_program_chunks") &&
           bm.download(bitmap_words->data(), bitmap_words->size() * 8);
  }

  struct SlotDesc {
    int col_idx;            // column reader
    int tuple_offset;       // SlotDescriptor::tuple_offset()
    int null_byte_offset;   // null_indicator_offset().byte_offset (OPTIONAL columns)
    int null_bit_mask;      // null_indicator_offset().bit_mask
  };
  // facade extra: AssembleRows' vector path (.cc:1101-1182) over EVERY page of the column chunks: the
  // conjunct list (ips_eval_program_chunks), every slot's late materialisation against the resulting
  // selection (ips_chunk_select: batches cut at the column's own page ends -> ips_batches_compact ->
  // one dense array per slot) and the row-major tuples (ips_assemble_tuples with dense REQUIRED
  // columns next to the selection's own batch counts).  Slots on OPTIONAL dictionary / FLE columns go through
  // ips_chunk_select_nullable (values of the selected NOT-NULL rows + NOT-NULL flags, NULL-indicator bits in the
  // tuples); an OPTIONAL PLAIN slot makes it return false.
  bool AssembleRowsChunks(int tuple_size, const uint8_t* template_tuple, const std::vector<SlotDesc>& slots,
                          std::vector<uint8_t>* tuples, int64_t* num_tuples) {
    if (slots.empty() || slots.size() > IPS_TUPLE_MAX_COLS) return false;
    lower_cols_.clear();
    chunk_readers_.clear();
    lower_chunks_ = true;
    std::vector<ips_node> program;
    bool lowered = true;
    for (size_t i = 0; i < simple_predicates_.size() && lowered; ++i) {
      lowered = simple_predicates_[i]->Lower(this, &program);
      if (lowered && i > 0) { ips_node n; memset(&n, 0, sizeof(n)); n.kind = IPS_NODE_AND; program.push_back(n); }
    }
    lower_chunks_ = false;
    if (!lowered) return false;
    // a chunk per reader that the predicates or the slots touch
    std::vector<std::unique_ptr<BaseColumnReader::ChunkHolder>> holders(column_readers_.size());
    auto chunk_of = [&](int idx) -> const ips_chunk* {
      if (!holders[(size_t)idx]) {
        holders[(size_t)idx].reset(new BaseColumnReader::ChunkHolder());
        if (!column_readers_[(size_t)idx]->OpenChunk(holders[(size_t)idx].get())) return nullptr;
      }
      return holders[(size_t)idx]->chunk;
    };
    std::vector<const ips_chunk*> pred_chunks;
    for (int idx : chunk_readers_) {
      const ips_chunk* c = chunk_of(idx);
      if (!c) return false;
      pred_chunks.push_back(c);
    }
    const ips_chunk* first = chunk_of(slots[0].col_idx);
    if (!first) return false;
    const int64_t n = ips_chunk_num_rows(first);
    for (const ips_chunk* c : pred_chunks)
      if (ips_chunk_num_rows(c) != n) return ips::ok(IPS_ERR_INVALID_ARG, "AssembleRowsChunks: the column chunks hold different row counts");
    *num_tuples = 0;
    tuples->clear();
    if (n == 0) return true;
    ips::DeviceBuffer bm((size_t)((n + 63) / 64 + 2) * 8);
    if (program.empty()) {
      if (!ips::ok(ips_bitmap_fill(bm.as<uint64_t>(), n, 1, nullptr), "ips_bitmap_fill")) return false;
    } else {
      const size_t ws_bytes = ips_chunk_program_workspace_bytes(program.data(), (int)program.size(), pred_chunks.data(), (int)pred_chunks.size());
      if (ws_bytes > 0 && !program_workspace_.resize(ws_bytes)) return false;
      if (!ips::ok(ips_eval_program_chunks(program.data(), (int)program.size(), pred_chunks.data(), (int)pred_chunks.size(),
                                           bm.as<uint64_t>(), ws_bytes ? program_workspace_.get() : nullptr, nullptr),
                   "ips_eval_program_chunks"))
        return false;
    }
    const int64_t nb = (n + IPS_BATCH_ROWS - 1) / IPS_BATCH_ROWS;
    ips::DeviceBuffer counts((size_t)nb * 4), d_count(16);
    int64_t count = 0;
    if (!ips::ok(ips_bitmap_batch_counts(bm.as<uint64_t>(), n, counts.as<uint32_t>(), nullptr), "ips_bitmap_batch_counts") ||
        !ips::ok(ips_bitmap_count(bm.as<uint64_t>(), n, d_count.as<int64_t>(), nullptr), "ips_bitmap_count") ||
        !d_count.download(&count, 8))
      return false;
    std::vector<std::unique_ptr<ips::DeviceBuffer>> keep;
    std::vector<ips_tuple_column> cols(slots.size());
    int n_optional = 0;
    for (size_t i = 0; i < slots.size(); ++i) {
      BaseColumnReader* r = column_readers_[(size_t)slots[i].col_idx].get();
      const ips_chunk* c = chunk_of(slots[i].col_idx);
      if (!c || ips_chunk_num_rows(c) != n) return false;
      const int vw = r->slot_width();
      if (r->max_def_level() > 0) {
        // an OPTIONAL column: values of the selected NOT-NULL rows + one NOT-NULL flag per selected row, across
        // the chunk's pages in four launches (ips_chunk_select_nullable); NULL rows keep the template's bytes and
        // get their NULL-indicator bit (descriptors.h:60-71)
        const size_t need = ips_chunk_select_nullable_workspace_bytes(c);
        if (need == 0) return ips::ok(IPS_ERR_UNSUPPORTED, "AssembleRowsChunks: an OPTIONAL slot needs FLE / dictionary pages");
        keep.emplace_back(new ips::DeviceBuffer(need + 64));
        ips::DeviceBuffer& sws = *keep.back();
        keep.emplace_back(new ips::DeviceBuffer((size_t)std::max<int64_t>(count, 1) * vw + 64));
        ips::DeviceBuffer& dense = *keep.back();
        keep.emplace_back(new ips::DeviceBuffer((size_t)((n + 63) / 64 + 2) * 8));
        ips::DeviceBuffer& flags = *keep.back();
        ips::DeviceBuffer d_counts(32);
        int64_t sel_counts[3] = {0, 0, 0};
        if (!ips::ok(ips_chunk_select_nullable(c, r->dict_handle(), bm.as<uint64_t>(), dense.get(), flags.as<uint64_t>(),
                                               d_counts.as<int64_t>(), sws.get(), nullptr), "ips_chunk_select_nullable") ||
            !d_counts.download(sel_counts, 24))
          return false;
        if (sel_counts[2]) return ips::ok(IPS_ERR_BAD_INDEX, "AssembleRowsChunks: a dictionary code outside the dictionary");
        if (sel_counts[0] != count) return ips::ok(IPS_ERR_HIP, "AssembleRowsChunks: a column saw another number of selected rows than the selection holds");
        memset(&cols[i], 0, sizeof(cols[i]));
        cols[i].value_width = vw;
        cols[i].tuple_offset = slots[i].tuple_offset;
        cols[i].d_dense_values = dense.get();
        cols[i].d_nonnull_flags = flags.as<uint64_t>();
        cols[i].null_byte_offset = slots[i].null_byte_offset;
        cols[i].null_bit_mask = slots[i].null_bit_mask;
        ++n_optional;
        continue;
      }
      const int64_t cb = ips_chunk_num_batches(c);
      keep.emplace_back(new ips::DeviceBuffer((size_t)cb * IPS_BATCH_ROWS * vw));
      ips::DeviceBuffer& values = *keep.back();
      keep.emplace_back(new ips::DeviceBuffer((size_t)cb * 4 + 16));
      ips::DeviceBuffer& cnts = *keep.back();
      keep.emplace_back(new ips::DeviceBuffer((size_t)std::max<int64_t>(count, 1) * vw + 64));
      ips::DeviceBuffer& dense = *keep.back();
      keep.emplace_back(new ips::DeviceBuffer(ips_batches_workspace_bytes(cb * IPS_BATCH_ROWS) + 64));
      ips::DeviceBuffer& ws = *keep.back();
      ips::DeviceBuffer total(16);
      if (!ips::ok(ips_chunk_select(c, r->dict_handle(), bm.as<uint64_t>(), values.get(), cnts.as<uint32_t>(), nullptr), "ips_chunk_select") ||
          !ips::ok(ips_batches_compact(values.get(), cnts.as<uint32_t>(), cb * IPS_BATCH_ROWS, vw, dense.get(), total.as<int64_t>(),
                                       ws.get(), nullptr), "ips_batches_compact"))
        return false;
      int64_t got = 0;
      if (!total.download(&got, 8) || got != count) return ips::ok(IPS_ERR_HIP, "AssembleRowsChunks: a column materialised another number of rows than the selection holds");
      memset(&cols[i], 0, sizeof(cols[i]));
      cols[i].value_width = vw;
      cols[i].tuple_offset = slots[i].tuple_offset;
      cols[i].d_dense_values = dense.get();
    }
    ips::DeviceBuffer d_tuples((size_t)std::max<int64_t>(count, 1) * tuple_size + 64), d_total(16);
    ips::DeviceBuffer ws(ips_assemble_workspace_bytes(n, n_optional) + 64);
    if (!ips::ok(ips_assemble_tuples(cols.data(), (int)cols.size(), counts.as<uint32_t>(), n, tuple_size, template_tuple,
                                     d_tuples.get(), d_total.as<int64_t>(), ws.get(), nullptr), "ips_assemble_tuples"))
      return false;
    int64_t total = 0;
    if (!d_total.download(&total, 8) || total != count) return false;
    tuples->assign((size_t)total * tuple_size, 0);
    *num_tuples = total;
    return total == 0 || d_tuples.download(tuples->data(), tuples->size());
  }

  // facade extra: AssembleRows' vector path (.cc:1101-1182) for num_rows rows of the current pages
  // in a handful of launches -- the conjunct list (ips_eval_program), every slot's late
  // materialisation against the resulting bitmap, and the row-major tuples (ips_assemble_tuples:
  // InitTuple(template_tuple_) + each column's ReadValue at slot_desc->tuple_offset(), NULL
  // indicator bits for OPTIONAL columns, descriptors.h:60-95).  Slots: 4- and 8-byte types.
  bool AssembleRowsFused(int64_t num_rows, int tuple_size, const uint8_t* template_tuple,
                         const std::vector<SlotDesc>& slots, std::vector<uint8_t>* tuples,
                         int64_t* num_tuples) {
    if (slots.empty() || slots.size() > IPS_TUPLE_MAX_COLS || !RowsResident(num_rows)) return false;
    std::vector<uint64_t> words;
    lower_cols_.clear();
    std::vector<ips_node> program;
    for (size_t i = 0; i < simple_predicates_.size(); ++i) {
      if (!simple_predicates_[i]->Lower(this, &program)) return false;
      if (i > 0) { ips_node n; memset(&n, 0, sizeof(n)); n.kind = IPS_NODE_AND; program.push_back(n); }
    }
    ips::DeviceBuffer bm((size_t)((num_rows + 63) / 64 + 2) * 8);
    if (program.empty()) {
      if (!ips::ok(ips_bitmap_fill(bm.as<uint64_t>(), num_rows, 1, nullptr), "ips_bitmap_fill")) return false;
    } else {
      const size_t ws_bytes = ips_program_workspace_bytes(program.data(), (int)program.size(), lower_cols_.data(),
                                                          (int)lower_cols_.size(), num_rows);
      if (ws_bytes > 0 && !program_workspace_.resize(ws_bytes)) return false;
      if (!ips::ok(ips_eval_program(program.data(), (int)program.size(), lower_cols_.data(), (int)lower_cols_.size(),
                                    num_rows, bm.as<uint64_t>(), ws_bytes ? program_workspace_.get() : nullptr, nullptr),
                   "ips_eval_program"))
        return false;
    }
    BaseColumnReader::Materialised mat;
    std::vector<ips_tuple_column> cols(slots.size());
    int n_optional = 0;
    for (size_t i = 0; i < slots.size(); ++i) {
      BaseColumnReader* r = column_readers_[(size_t)slots[i].col_idx].get();
      if (!r->SelectInto(num_rows, bm.as<uint64_t>(), &cols[i], &mat)) return false;
      cols[i].tuple_offset = slots[i].tuple_offset;
      if (cols[i].d_nonnull_flags) {
        cols[i].null_byte_offset = slots[i].null_byte_offset;
        cols[i].null_bit_mask = slots[i].null_bit_mask;
        ++n_optional;
      }
    }
    if (!mat.counts.get()) {  // only OPTIONAL slots: the batch counts come from the bitmap itself
      const int64_t nb = (num_rows + IPS_BATCH_ROWS - 1) / IPS_BATCH_ROWS;
      if (!mat.counts.resize((size_t)nb * 4) ||
          !ips::ok(ips_bitmap_batch_counts(bm.as<uint64_t>(), num_rows, mat.counts.as<uint32_t>(), nullptr),
                   "ips_bitmap_batch_counts"))
        return false;
    }
    int64_t count = 0;
    ips::DeviceBuffer d_count(16);
    if (!ips::ok(ips_bitmap_count(bm.as<uint64_t>(), num_rows, d_count.as<int64_t>(), nullptr), "ips_bitmap_count") ||
        !d_count.download(&count, 8))
      return false;
    ips::DeviceBuffer d_tuples((size_t)std::max<int64_t>(count, 1) * tuple_size + 64), d_total(16);
    ips::DeviceBuffer ws(ips_assemble_workspace_bytes(num_rows, n_optional) + 64);
    if (!ips::ok(ips_assemble_tuples(cols.data(), (int)cols.size(), mat.counts.as<uint32_t>(), num_rows, tuple_size,
                                     template_tuple, d_tuples.get(), d_total.as<int64_t>(), ws.get(), nullptr),
                 "ips_assemble_tuples"))
      return false;
    int64_t total = 0;
    if (!d_total.download(&total, 8) || total != count) return false;
    for (const int64_t* flag : mat.bad_index) {  // DictDecoder::GetValue returning false (dict-encoding.h:316)
      int64_t bad = 0;
      if (!ips::ok(ips_memcpy_d2h(&bad, flag, 8, nullptr), "ips_memcpy_d2h") ||
          !ips::ok(ips_stream_synchronize(nullptr), "ips_stream_synchronize"))
        return false;
      if (bad) return ips::ok(IPS_ERR_BAD_INDEX, "AssembleRowsFused: a dictionary code outside the dictionary");
    }
    tuples->assign((size_t)total * tuple_size, 0);
    *num_tuples = total;
    return total == 0 || d_tuples.download(tuples->data(), tuples->size());
  }

  // used by LeafOperate::Lower: returns the column slot of the program for reader idx
  bool LowerLeaf(int idx, int op, const void* lits, int n_lits, ips_node* node) {
    ips_column col;
    if (lower_chunks_) {  // leaf.column = the slot of reader idx among the chunks of this evaluation
      if (!column_readers_[(size_t)idx]->LowerLeaf(op, lits, n_lits, &col, node)) return false;
      for (size_t i = 0; i < chunk_readers_.size(); ++i)
        if (chunk_readers_[i] == idx) { node->column = (int)i; return true; }
      if (chunk_readers_.size() >= IPS_PROGRAM_MAX_COLS) return false;
      chunk_readers_.push_back(idx);
      node->column = (int)chunk_readers_.size() - 1;
      return true;
    }
    if (lower_cols_.size() >= IPS_PROGRAM_MAX_COLS) return false;
    if (!column_readers_[(size_t)idx]->LowerLeaf(op, lits, n_lits, &col, node)) return false;
    for (size_t i = 0; i < lower_cols_.size(); ++i)
      if (lower_cols_[i].d_data == col.d_data) { node->column = (int)i; return true; }
    lower_cols_.push_back(col);
    node->column = (int)lower_cols_.size() - 1;
    return true;
  }

  ips_status parse_status() const { return ips::sticky_status(); }

 private:
  // the reference reinterpret_casts on the literal's type (.cc:1870, quirk Q9); the facade
  // requires the column to have been added with the same T
  template <typename T>
  ColumnReader<T>* reader(int idx) { return static_cast<ColumnReader<T>*>(column_readers_[(size_t)idx].get()); }

  std::vector<std::unique_ptr<BaseColumnReader>> column_readers_;
  std::vector<SimplePredicate*> simple_predicates_;
  std::vector<SimplePredicate*> owned_;
  std::vector<ips_column> lower_cols_;
  std::vector<int> chunk_readers_;  // EvalSimplePredicatesChunks: reader index of every chunk slot
  bool lower_chunks_ = false;
  ips::DeviceBuffer program_workspace_;
};

// ---- ColumnReader dispatch helpers ----
template <typename T>
template <typename D, typename L>
void HdfsParquetScanner::ColumnReader<T>::Call(int op, D& d, int64_t n, SkipBitset& b, L& lit) {
  if constexpr (std::is_same<L, std::vector<T>>::value) {
    d.In(n, b, lit);
  } else {
    switch (op) {
      case IPS_OP_EQ: d.Eq(n, b, lit); break;
      case IPS_OP_LT: d.Lt(n, b, lit); break;
      case IPS_OP_LE: d.Le(n, b, lit); break;
      case IPS_OP_GT: d.Gt(n, b, lit); break;
      default: d.Ge(n, b, lit); break;
    }
  }
}

// PLAIN pages (ParquetPlainEncoder::Eq..In(data_, ..), parquet-common.h:197-255): the reference
// walks num_rows slots from data_ on every call.  Here the page is uploaded once per InitDataPage,
// each distinct (op, literals) is evaluated over the WHOLE page in one ips_plain_pred launch, and
// the 1024-row batches are served from that bitmap at the rows data_ stands at.
template <typename T>
template <typename L>
void HdfsParquetScanner::ColumnReader<T>::PlainCall(int op, int64_t n, SkipBitset& b, L& lit) {
  const T* lits;
  int n_lits;
  if constexpr (std::is_same<L, std::vector<T>>::value) {
    if (IPS_PLAIN_SEMANTICS == IPS_SEM_REFERENCE || lit.empty()) return;  // reference: empty body (:252-255)
    lits = lit.data();
    n_lits = (int)lit.size();
  } else {
    lits = &lit;
    n_lits = 1;
  }
  if (n <= 0) return;
  const ips_type t = IpsTypeOf<T>::value;
  const std::vector<uint64_t>* words = plain_preds_.find(op, lits, (size_t)n_lits * sizeof(T));
  if (!words) {
    std::vector<uint64_t>* w = plain_preds_.insert(op, lits, (size_t)n_lits * sizeof(T));
    w->assign((size_t)((plain_rows_ + 63) / 64), 0);
    if (plain_rows_ > 0 && EnsurePlainResident()) {
      ++ips::stats().pred_launches;
      ips::DeviceBuffer bm(w->size() * 8);
      if (ips::ok(ips_plain_pred(plain_dev_.get(), plain_rows_, t, (ips_op)op, lits, n_lits,
                                 (ips_semantics)IPS_PLAIN_SEMANTICS, bm.as<uint64_t>(), nullptr),
                  "ips_plain_pred"))
        bm.download(w->data(), w->size() * 8);
    }
    words = w;
  }
  const int64_t first = (int64_t)(data_ - plain_begin_) / ips_plain_stride(t);
  ips::append_bits(b, *words, first, n, plain_rows_);
}

// ---- leaves of the predicate tree, simple-predicates.h:165-205 ----
template <typename T, int OP>
void LeafOperate<T, OP>::GetBitset(HdfsParquetScanner* scanner, int64_t num_rows, SkipBitset& skip_bitset) {
  switch (OP) {
    case IPS_OP_EQ: scanner->Eq(idx_, num_rows, skip_bitset, vals_[0]); break;
    case IPS_OP_LT: scanner->Lt(idx_, num_rows, skip_bitset, vals_[0]); break;
    case IPS_OP_LE: scanner->Le(idx_, num_rows, skip_bitset, vals_[0]); break;
    case IPS_OP_GT: scanner->Gt(idx_, num_rows, skip_bitset, vals_[0]); break;
    case IPS_OP_GE: scanner->Ge(idx_, num_rows, skip_bitset, vals_[0]); break;
    default: scanner->In(idx_, num_rows, skip_bitset, vals_); break;
  }
}

template <typename T, int OP>
bool LeafOperate<T, OP>::Lower(HdfsParquetScanner* scanner, std::vector<ips_node>* program) {
  ips_node n;
  if (!scanner->LowerLeaf(idx_, OP, vals_.data(), (int)vals_.size(), &n)) return false;
  program->push_back(n);
  return true;
}

}  // namespace impala
