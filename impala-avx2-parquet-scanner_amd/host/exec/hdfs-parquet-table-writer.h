// exec/hdfs-parquet-table-writer.h (MI355X facade) -- the slice of HdfsParquetTableWriter::
// BaseColumnWriter that produces the bytes of ONE column chunk (hdfs-parquet-table-writer.cc:
// 266-299 AppendRow, 434-464 WriteDictDataPage, 466-620 Flush, 700-735 NewPage): the dictionary page
// behind its thrift PageHeader, then the data pages, each
//     [PageHeader][ [int32 n_def_bytes][FLE definition levels]  [uint8 code width][FLE codes] ]
// with the page bodies run through the chunk's codec.  It exists so that the scanner facade can be
// fed -- and tested with -- the byte stream the reference's writer produces, not to
// write files: footer, row groups and statistics stay out of scope.
#pragma once
#include <string.h>

#include <vector>

#include "../util/dict-encoding.h"
#include "parquet-page-header.h"

namespace impala {

template <typename T>
class ColumnChunkWriter {
 public:
  // max_def_level 0 = REQUIRED column (no definition levels are written), 1 = OPTIONAL
  ColumnChunkWriter(int codec, int max_def_level, int64_t page_rows)
      : codec_(codec), max_def_level_(max_def_level), page_rows_(page_rows) { NewPage(); }

  // BaseColumnWriter::AppendRow, :266-299: value == NULL appends a NULL row
  bool AppendRow(const T* value) {
    if (value == nullptr && max_def_level_ == 0) return false;
    Page& pg = pages_.back();
    pg.def_levels.push_back(value ? (uint32_t)max_def_level_ : 0u);
    if (value) {
      if (dict_.Put(*value) < 0) return false;  // over the 40000-entry cap: the reference falls back to PLAIN
      pg.node_indices.push_back(dict_.buffered_indices().back());
    }
    if ((int64_t)pg.def_levels.size() == page_rows_) NewPage();
    return true;
  }

  int64_t num_values() const {
    int64_t n = 0;
    for (const Page& pg : pages_) n += (int64_t)pg.def_levels.size();
    return n;
  }

  // BaseColumnWriter::Flush, :466-620: dictionary page first, then every non-empty data page
  bool Flush(std::vector<uint8_t>* out) {
    {
      std::vector<uint8_t> dict_buffer((size_t)dict_.dict_encoded_size() + 8);
      dict_.WriteDict(dict_buffer.data());
      dict_buffer.resize((size_t)dict_.dict_encoded_size());
      parquet::PageHeader header;
      header.type = parquet::PageType::DICTIONARY_PAGE;
      header.uncompressed_page_size = (int32_t)dict_buffer.size();
      header.dictionary_page_header.num_values = dict_.num_entries();
      header.dictionary_page_header.encoding = parquet::Encoding::FLE_DICTIONARY;  // :491
      header.__isset.dictionary_page_header = true;
      if (!AppendPage(header, dict_buffer, out)) return false;
    }
    for (const Page& pg : pages_) {
      if (pg.def_levels.empty()) continue;  // "Last page might be empty", :542-547
      std::vector<uint8_t> body;
      if (max_def_level_ > 0) {
        const int def_bw = BitUtil::Log2((uint64_t)max_def_level_ + 1);
        std::vector<uint8_t> defs((size_t)ips_fle_encoded_bytes((int64_t)pg.def_levels.size(), def_bw));
        FleEncoder def_levels(defs.data(), (int)defs.size(), def_bw);
        for (uint32_t d : pg.def_levels)
          if (!def_levels.Put(d)) return false;
        const int32_t num_def_level_bytes = def_levels.Flush();
        body.resize(4 + (size_t)num_def_level_bytes);  // buffer.Append(num_def_level_bytes); Append(levels), :567-575
        memcpy(body.data(), &num_def_level_bytes, 4);
        memcpy(body.data() + 4, defs.data(), (size_t)num_def_level_bytes);
      }
      std::vector<uint8_t> values((size_t)(1 + ips_fle_encoded_bytes((int64_t)pg.node_indices.size(), 32)));
      const int len = dict_.WriteData(values.data(), (int)values.size(), pg.node_indices);
      if (len < 0) return false;
      body.insert(body.end(), values.begin(), values.begin() + len);
      parquet::PageHeader header;
      header.type = parquet::PageType::DATA_PAGE;
      header.uncompressed_page_size = (int32_t)body.size();
      header.data_page_header.num_values = (int32_t)pg.def_levels.size();
      header.data_page_header.encoding = parquet::Encoding::FLE_DICTIONARY;
      header.data_page_header.definition_level_encoding = parquet::Encoding::FLE;          // :728
      header.data_page_header.repetition_level_encoding = parquet::Encoding::BIT_PACKED;   // :729
      header.__isset.data_page_header = true;
      if (!AppendPage(header, body, out)) return false;
    }
    return true;
  }

 private:
  struct Page {
    std::vector<uint32_t> def_levels;
    std::vector<int> node_indices;  // insertion-order dictionary indices of the non-NULL rows
  };
  void NewPage() { pages_.emplace_back(); }

  bool AppendPage(parquet::PageHeader header, const std::vector<uint8_t>& body, std::vector<uint8_t>* out) {
    std::vector<uint8_t> compressed;
    if (!parquet::Compress(codec_, body.data(), (int64_t)body.size(), &compressed)) return false;
    header.compressed_page_size = (int32_t)compressed.size();
    parquet::SerializePageHeader(header, out);
    out->insert(out->end(), compressed.begin(), compressed.end());
    return true;
  }

  int codec_;
  int max_def_level_;
  int64_t page_rows_;
  DictEncoder<T> dict_;
  std::vector<Page> pages_;
};

}  // namespace impala
