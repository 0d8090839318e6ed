// exec/parquet-column-chunk.h (MI355X facade) -- the host-only half of BaseColumnReader::ReadDataPage
// (hdfs-parquet-scanner.cc:730-924): walking a column chunk's bytes page by page (thrift headers,
// codecs, dictionary page, encodings, value counts) and the framing of a data page's payload
// (InitDataPage, :882-916).  No device, no libips_hip.so: HdfsParquetScanner::AddColumnChunk uploads
// what this returns, and tests/host_column_chunk_test.cpp fuzzes it under ASan + UBSan.
#pragma once
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <memory>
#include <vector>

#include "parquet-page-header.h"

namespace impala {
namespace parquet {

struct ChunkDataPage { std::vector<uint8_t>* bytes; int64_t num_values; int encoding; };
struct ColumnChunkPages {
  std::vector<std::unique_ptr<std::vector<uint8_t>>> owned;  // decompressed page bodies
  std::vector<uint8_t>* dict_values = nullptr;               // the dictionary page's PLAIN entries
  std::vector<ChunkDataPage> data_pages;
  bool dict_coded = false;
};

// Data-page payload framing (InitDataPage, .cc:882-916): OPTIONAL columns start with
// [int32 n_def_bytes][FLE definition levels], then [uint8 code_width][FLE codes].  The
// reference bounds-checks the 4-byte read (ReadWriteUtil::Read); a short or corrupt page must
// not make the decoders read outside the buffer.
inline bool SplitDataPage(uint8_t* data, int len, int max_def_level, uint8_t** def_levels,
                          int* n_def_bytes, uint8_t** codes, int* codes_len) {
  uint8_t* p = data;
  int left = len;
  *def_levels = nullptr;
  *n_def_bytes = 0;
  if (data == nullptr || len < 0) return false;
  if (max_def_level > 0) {
    if (left < 4) return false;
    int32_t nb;
    memcpy(&nb, p, 4);
    p += 4; left -= 4;
    if (nb < 0 || nb > left) return false;
    *def_levels = p;
    *n_def_bytes = nb;
    p += nb; left -= nb;
  }
  if (left < 1) return false;  // the code-width byte (DictDecoderBase::SetData)
  if (*p < 1 || *p > 32) return false;
  *codes = p;
  *codes_len = left;
  return true;
}

// rows of a page whose level equals max_def_level: the values the page stores.  Levels are an FLE
// column of width Log2(max_def_level + 1) over num_values rows (writer :387-399): value k of block b
// is bit 63 - k of the block's words (fle-encoding.h:8338-8340).  -1: the bytes do not cover the rows.
inline int64_t CountNonNull(const uint8_t* levels, int64_t n_bytes, int max_def_level, int64_t num_values) {
  int w = 0;
  for (uint64_t x = (uint64_t)max_def_level; x; x >>= 1) ++w;  // Log2(max + 1) for max >= 1
  if (w < 1 || num_values < 0) return -1;
  const int64_t blocks = (num_values + 63) / 64;
  if (blocks * 8 * w > n_bytes) return -1;
  int64_t count = 0;
  for (int64_t b = 0; b < blocks; ++b) {
    uint64_t eq = ~0ull;  // rows whose level equals max_def_level
    for (int i = 0; i < w; ++i) {
      uint64_t plane;
      memcpy(&plane, levels + (b * w + i) * 8, 8);
      eq &= ((uint64_t)max_def_level >> i) & 1 ? plane : ~plane;
    }
    const int64_t rows = num_values - b * 64;
    if (rows < 64) eq &= ~0ull << (64 - rows);  // row k sits at bit 63 - k
    count += __builtin_popcountll(eq);
  }
  return count;
}

// What AddColumnChunk's pages promise must be there before the device reads it (the reference's
// decoders check every Get against buffer_end_guard_, fle-encoding.h:350,407; its predicates do not,
// quirk Q7 -- a header that claims more num_values than the level or code blocks hold would send the
// fused kernels past the device allocations).  nullptr = the page holds its rows.
inline const char* CheckDictDataPage(const uint8_t* def_levels, int n_def_bytes, const uint8_t* codes, int codes_len,
                                     int max_def_level, int64_t num_values) {
  if (codes == nullptr || codes_len < 1) return "data page without a code-width byte";
  const int bw = codes[0];
  if (bw < 1 || bw > 32) return "code width outside 1..32";
  int64_t data_rows = num_values;
  if (max_def_level > 0) {
    data_rows = CountNonNull(def_levels, n_def_bytes, max_def_level, num_values);
    if (data_rows < 0) return "definition levels hold fewer rows than the page header states";
  }
  if (((data_rows + 63) / 64) * 8 * bw > (int64_t)codes_len - 1)
    return "code blocks hold fewer rows than the page header states";
  return nullptr;
}

// PLAIN pages: the values follow the level bytes (data_ += num_definition_bytes, .cc:916-917).  An
// OPTIONAL page stores its non-NULL values only: *n_stored = the rows the levels say are there.
inline bool PlainPageValues(uint8_t* data, int64_t len, int max_def_level, int64_t num_values,
                            int value_stride, uint8_t** values, int64_t* n_stored = nullptr) {
  uint8_t* p = data;
  int64_t left = len;
  if (data == nullptr || len < 0 || num_values < 0) return false;
  int64_t stored = num_values;
  if (max_def_level > 0) {
    if (left < 4) return false;
    int32_t nb;
    memcpy(&nb, p, 4);
    if (nb < 0 || nb > left - 4) return false;
    stored = CountNonNull(p + 4, nb, max_def_level, num_values);
    if (stored < 0) return false;
    p += 4 + nb; left -= 4 + nb;
  }
  if (left < stored * value_stride) return false;
  *values = p;
  if (n_stored) *n_stored = stored;
  return true;
}

// nullptr on success, else what ReadDataPage would put into parse_status_
inline const char* WalkColumnChunk(const uint8_t* chunk, int64_t chunk_len, int64_t num_values, int codec,
                                   int max_def_level, int value_stride, ColumnChunkPages* out) {
    if (!CodecSupported(codec)) return ("compression codec not supported (UNCOMPRESSED, SNAPPY and GZIP are)");
    if (chunk == nullptr || chunk_len < 0 || num_values < 0) return ("bad column chunk");
    auto& owned = out->owned;
    std::vector<uint8_t>*& dict_values = out->dict_values;
    using DataPage = ChunkDataPage;
    auto& data_pages = out->data_pages;
    int64_t pos = 0, num_values_read = 0;
    while (num_values_read < num_values) {  // .cc:741-748
      if (pos >= chunk_len) return ("column metadata states more values than the pages hold");  // PARQUET_COLUMN_METADATA_INVALID
      PageHeader header;
      uint32_t header_size = (uint32_t)std::min<int64_t>(chunk_len - pos, 1 << 20);
      if (!DeserializeThriftMsg(chunk + pos, &header_size, true, &header))
        return ("could not read the page header");  // .cc:762-798
      pos += header_size;
      const int64_t data_size = header.compressed_page_size;
      const int64_t uncompressed_size = header.uncompressed_page_size;
      if (data_size < 0 || uncompressed_size < 0 || pos + data_size > chunk_len) return ("page runs past the column chunk");
      if (header.type == PageType::DICTIONARY_PAGE) {  // .cc:806-853
        if (dict_values) return ("Column chunk should not contain two dictionary pages.");
        if (!header.__isset.dictionary_page_header) return ("Dictionary page does not have dictionary header set.");
        const int e = header.dictionary_page_header.encoding;
        if (e != Encoding::PLAIN && e != Encoding::PLAIN_DICTIONARY && e != Encoding::FLE_DICTIONARY)
          return ("Only PLAIN and PLAIN_DICTIONARY encodings are supported for dictionary pages.");
        owned.emplace_back(new std::vector<uint8_t>());
        if (!Decompress(codec, chunk + pos, data_size, uncompressed_size, owned.back().get()))
          return ("dictionary page does not decompress to its stated size");
        dict_values = owned.back().get();
        if ((int64_t)dict_values->size() != (int64_t)header.dictionary_page_header.num_values * value_stride)
          return ("Invalid dictionary. Entry count differs from the dictionary page header");  // .cc:845-850
        pos += data_size;
        continue;
      }
      if (header.type != PageType::DATA_PAGE || !header.__isset.data_page_header) {
        pos += data_size;  // "We can safely skip non-data pages", .cc:855-859
        continue;
      }
      const DataPageHeader& dh = header.data_page_header;
      if (dh.num_values < 0) return ("negative value count in a data page header");
      if (!IsEncodingSupported(dh.encoding)) return ("unsupported data page encoding");  // .cc:1637-1649
      // the vectorised predicates read the levels with fle_def_levels_ whatever the header says
      // (.cc:342 vs :885-912): refuse instead of dereferencing NULL
      if (max_def_level > 0 && dh.definition_level_encoding != Encoding::FLE)
        return ("definition levels are not FLE encoded: this page needs the row-at-a-time path");
      owned.emplace_back(new std::vector<uint8_t>());
      if (!Decompress(codec, chunk + pos, data_size, uncompressed_size, owned.back().get()))
        return ("data page does not decompress to its stated size");
      pos += data_size;
      num_values_read += dh.num_values;
      if (dh.num_values == 0) continue;
      data_pages.push_back(DataPage{owned.back().get(), dh.num_values, dh.encoding});
    }
    if (num_values_read != num_values) return ("pages hold more values than the column metadata states");
    if (data_pages.empty()) return ("column chunk without data pages");
    const bool dict_coded = out->dict_coded = data_pages[0].encoding == Encoding::PLAIN_DICTIONARY ||
                            data_pages[0].encoding == Encoding::FLE_DICTIONARY;
    for (const DataPage& pg : data_pages) {
      const bool d = pg.encoding == Encoding::PLAIN_DICTIONARY || pg.encoding == Encoding::FLE_DICTIONARY;
      if (d != dict_coded || (!d && pg.encoding != Encoding::PLAIN))
        return ("mixed page encodings inside one column chunk are not supported");
    }
    return nullptr;
}

}  // namespace parquet
}  // namespace impala
