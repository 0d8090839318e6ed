// CPU test of the host-only half of the page container (host/exec/parquet-column-chunk.h): the
// column-chunk walk and the data-page framing on well-formed chunks and, under ASan + UBSan, on
// thousands of corrupted ones.  No GPU, no libips_hip.so.
#include <stdio.h>
#include <stdlib.h>

#include <string>

#include "../impala-avx2-parquet-scanner_amd/host/exec/parquet-column-chunk.h"

using namespace impala::parquet;

static int g_fail = 0;
#define CHECK(c) do { if (!(c)) { ++g_fail; fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); } } while (0)

static uint64_t g_x = 0x2545F4914F6CDD1Dull;
static uint64_t rnd() { g_x ^= g_x << 13; g_x ^= g_x >> 7; g_x ^= g_x << 17; return g_x; }

static void append_page(std::vector<uint8_t>* chunk, const PageHeader& proto, const std::vector<uint8_t>& body, int codec) {
  std::vector<uint8_t> comp, hdr;
  Compress(codec, body.data(), (int64_t)body.size(), &comp);
  PageHeader h = proto;
  h.uncompressed_page_size = (int32_t)body.size();
  h.compressed_page_size = (int32_t)comp.size();
  SerializePageHeader(h, &hdr);
  chunk->insert(chunk->end(), hdr.begin(), hdr.end());
  chunk->insert(chunk->end(), comp.begin(), comp.end());
}

// a chunk of n_pages data pages of rows_per_page rows; dict: a dictionary page of 37 int32 entries
// in front and [width byte][blocks] payloads, else PLAIN int32 values; optional: level section
static std::vector<uint8_t> make_chunk(int codec, bool dict, bool optional, int n_pages, int rows_per_page) {
  std::vector<uint8_t> chunk;
  if (dict) {
    PageHeader dh;
    dh.type = PageType::DICTIONARY_PAGE;
    dh.__isset.dictionary_page_header = true;
    dh.dictionary_page_header.num_values = 37;
    dh.dictionary_page_header.encoding = Encoding::PLAIN_DICTIONARY;
    std::vector<uint8_t> entries(37 * 4);
    for (auto& b : entries) b = (uint8_t)rnd();
    append_page(&chunk, dh, entries, codec);
  }
  for (int pg = 0; pg < n_pages; ++pg) {
    PageHeader h;
    h.type = PageType::DATA_PAGE;
    h.__isset.data_page_header = true;
    h.data_page_header.num_values = rows_per_page;
    h.data_page_header.encoding = dict ? Encoding::FLE_DICTIONARY : Encoding::PLAIN;
    h.data_page_header.definition_level_encoding = Encoding::FLE;
    h.data_page_header.repetition_level_encoding = Encoding::BIT_PACKED;
    std::vector<uint8_t> body;
    if (optional) {
      const int32_t nb = (rows_per_page + 63) / 64 * 8;
      body.resize(4 + (size_t)nb);
      memcpy(body.data(), &nb, 4);
      for (int i = 0; i < nb; ++i) body[4 + (size_t)i] = (uint8_t)rnd();
    }
    if (dict) {
      body.push_back(6);  // code width
      const size_t blocks = (size_t)(rows_per_page + 63) / 64 * 6 * 8;
      for (size_t i = 0; i < blocks; ++i) body.push_back((uint8_t)rnd());
    } else {
      for (int i = 0; i < rows_per_page * 4; ++i) body.push_back((uint8_t)(i * 7));
    }
    append_page(&chunk, h, body, codec);
  }
  return chunk;
}

// touch every byte the framing functions hand out (ASan checks the bounds)
static uint64_t consume(const ColumnChunkPages& pages, int max_def_level) {
  uint64_t sum = 0;
  if (pages.dict_values) for (uint8_t b : *pages.dict_values) sum += b;
  for (const ChunkDataPage& pg : pages.data_pages) {
    if (pages.dict_coded) {
      uint8_t *def = nullptr, *codes = nullptr;
      int nb = 0, cl = 0;
      if (SplitDataPage(pg.bytes->data(), (int)pg.bytes->size(), max_def_level, &def, &nb, &codes, &cl)) {
        for (int i = 0; i < nb; ++i) sum += def[i];
        for (int i = 0; i < cl; ++i) sum += codes[i];
        // the row check the scanner runs before it uploads the page: whatever it answers, it reads
        // inside the page; when it accepts, the blocks the device will read are inside it too
        if (CheckDictDataPage(def, nb, codes, cl, max_def_level, pg.num_values) == nullptr) {
          const int bw = codes[0];
          int64_t rows = pg.num_values;
          if (max_def_level > 0) {
            rows = CountNonNull(def, nb, max_def_level, pg.num_values);
            for (int64_t i = 0; i < ((pg.num_values + 63) / 64) * 8; ++i) sum += def[i];
          }
          for (int64_t i = 0; i < ((rows + 63) / 64) * 8 * bw; ++i) sum += codes[1 + i];
        }
      }
    } else {
      uint8_t* v = nullptr;
      int64_t stored = 0;  // an OPTIONAL PLAIN page stores its non-NULL values only
      if (PlainPageValues(pg.bytes->data(), (int64_t)pg.bytes->size(), max_def_level, pg.num_values, 4, &v, &stored))
        for (int64_t i = 0; i < stored * 4; ++i) sum += v[i];
    }
  }
  return sum;
}

int main() {
  uint64_t sink = 0;
  int accepted = 0, refused = 0;
  for (int codec : {(int)CompressionCodec::UNCOMPRESSED, (int)CompressionCodec::SNAPPY, (int)CompressionCodec::GZIP}) {
    for (int dict = 0; dict < 2; ++dict) {
      for (int optional = 0; optional < 2; ++optional) {
        const int n_pages = 3, rows = 700;
        const std::vector<uint8_t> chunk = make_chunk(codec, dict != 0, optional != 0, n_pages, rows);
        ColumnChunkPages pages;
        const char* err = WalkColumnChunk(chunk.data(), (int64_t)chunk.size(), (int64_t)n_pages * rows, codec, optional, 4, &pages);
        CHECK(err == nullptr);
        CHECK((int)pages.data_pages.size() == n_pages && pages.dict_coded == (dict != 0));
        CHECK((pages.dict_values != nullptr) == (dict != 0));
        sink += consume(pages, optional);
        // what ReadDataPage refuses
        ColumnChunkPages p2;
        CHECK(WalkColumnChunk(chunk.data(), (int64_t)chunk.size() / 2, (int64_t)n_pages * rows, codec, optional, 4, &p2) != nullptr);
        ColumnChunkPages p3;
        CHECK(WalkColumnChunk(chunk.data(), (int64_t)chunk.size(), (int64_t)n_pages * rows + 1, codec, optional, 4, &p3) != nullptr);
        ColumnChunkPages p4;
        CHECK(WalkColumnChunk(chunk.data(), (int64_t)chunk.size(), (int64_t)n_pages * rows, CompressionCodec::LZO, optional, 4, &p4) != nullptr);
        // corrupted chunks: a few bytes changed, sometimes cut short, sometimes the wrong row
        // count or codec -- any answer is fine, out-of-bounds accesses are not
        for (int it = 0; it < 1500; ++it) {
          std::vector<uint8_t> bad = chunk;
          const int flips = 1 + (int)(rnd() % 4);
          for (int k = 0; k < flips; ++k) bad[rnd() % bad.size()] = (uint8_t)rnd();
          if (rnd() % 5 == 0) bad.resize(1 + rnd() % bad.size());
          const int64_t nv = (rnd() % 7 == 0) ? (int64_t)(rnd() % 5000) : (int64_t)n_pages * rows;
          const int cd = (rnd() % 11 == 0) ? (int)(rnd() % 4) : codec;
          ColumnChunkPages p;
          if (WalkColumnChunk(bad.data(), (int64_t)bad.size(), nv, cd, optional, 4, &p) == nullptr) {
            ++accepted;
            sink += consume(p, optional);
          } else {
            ++refused;
          }
        }
      }
    }
  }
  CHECK(accepted > 0 && refused > 0);
  printf("host_column_chunk_test: %d failed (%d corrupted chunks accepted, %d refused, %llu)\n", g_fail, accepted,
         refused, (unsigned long long)(sink & 1));
  return g_fail ? 1 : 0;
}
