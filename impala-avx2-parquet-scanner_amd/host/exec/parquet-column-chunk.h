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

// PLAIN pages: the values follow the level bytes (data_ += num_definition_bytes, .cc:916-917)
inline bool PlainPageValues(uint8_t* data, int64_t len, int max_def_level, int64_t num_values,
                            int value_stride, uint8_t** values) {
  uint8_t* p = data;
  int64_t left = len;
  if (data == nullptr || len < 0 || num_values < 0) return false;
  if (max_def_level > 0) {
    if (left < 4) return false;
    int32_t nb;
    memcpy(&nb, p, 4);
    if (nb < 0 || nb > left - 4) return false;
    p += 4 + nb; left -= 4 + nb;
  }
  if (left < num_values * value_stride) return false;
  *values = p;
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
