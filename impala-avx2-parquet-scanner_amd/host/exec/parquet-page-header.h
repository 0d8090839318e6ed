// exec/parquet-page-header.h (MI355X facade) -- the page container of a Parquet column chunk as
// the reference reads and writes it: thrift PageHeader / DataPageHeader / DictionaryPageHeader
// (parquet.thrift:163-331) in the compact protocol (what DeserializeThriftMsg(buf, &len, true, ..)
// parses in BaseColumnReader::ReadDataPage, hdfs-parquet-scanner.cc:762-770, and what
// thrift_serializer_->Serialize writes in BaseColumnWriter::Flush, hdfs-parquet-table-writer.cc:
// 519-522, 605-608), plus the page codecs: UNCOMPRESSED, GZIP (zlib) and SNAPPY (the reference
// writer's default; util/snappy-codec.h restates the raw block format).  LZO is refused, as the
// reference's own Parquet scanner does.
//
// No thrift dependency: the compact protocol is a handful of varints.  A struct is a sequence of
// fields, each introduced by one byte (id delta << 4 | type) or, for deltas outside 1..15, by
// (0 << 4 | type) followed by the zigzag-varint field id; 0x00 ends the struct.  i32 / enum
// values are zigzag varints; unknown fields (e.g. DataPageHeader.statistics) are skipped by type.
// Host-side only: nothing here touches the device.
#pragma once
#include <stdint.h>
#include <string.h>
#include <zlib.h>

#include <vector>

#include "../util/snappy-codec.h"

namespace impala {
namespace parquet {

// parquet.thrift:163-225
struct Encoding {
  enum type { PLAIN = 0, PLAIN_DICTIONARY = 2, RLE = 3, BIT_PACKED = 4, DELTA_BINARY_PACKED = 5,
              DELTA_LENGTH_BYTE_ARRAY = 6, DELTA_BYTE_ARRAY = 7, RLE_DICTIONARY = 8, FLE = 9,
              FLE_DICTIONARY = 10 };
};
// parquet.thrift:230-242
struct CompressionCodec { enum type { UNCOMPRESSED = 0, SNAPPY = 1, GZIP = 2, LZO = 3 }; };
struct PageType { enum type { DATA_PAGE = 0, INDEX_PAGE = 1, DICTIONARY_PAGE = 2, DATA_PAGE_V2 = 3 }; };

struct DataPageHeader {  // parquet.thrift:245-260
  int32_t num_values = 0;
  int32_t encoding = 0;
  int32_t definition_level_encoding = 0;
  int32_t repetition_level_encoding = 0;
};
struct DictionaryPageHeader {  // parquet.thrift:266-275
  int32_t num_values = 0;
  int32_t encoding = 0;
  bool is_sorted = false;
};
struct PageHeader {  // parquet.thrift:311-331
  int32_t type = 0;
  int32_t uncompressed_page_size = 0;
  int32_t compressed_page_size = 0;
  int32_t crc = 0;
  DataPageHeader data_page_header;
  DictionaryPageHeader dictionary_page_header;
  struct { bool crc = false, data_page_header = false, dictionary_page_header = false; } __isset;
};

// hdfs-parquet-scanner.cc:1637-1649
inline bool IsEncodingSupported(int e) {
  switch (e) {
    case Encoding::PLAIN: case Encoding::PLAIN_DICTIONARY: case Encoding::BIT_PACKED:
    case Encoding::RLE: case Encoding::FLE: case Encoding::FLE_DICTIONARY: return true;
    default: return false;
  }
}

namespace compact {
enum { T_STOP = 0, T_TRUE = 1, T_FALSE = 2, T_BYTE = 3, T_I16 = 4, T_I32 = 5, T_I64 = 6, T_DOUBLE = 7,
       T_BINARY = 8, T_LIST = 9, T_SET = 10, T_MAP = 11, T_STRUCT = 12 };

struct Reader {
  const uint8_t* p;
  const uint8_t* end;
  bool ok = true;
  int depth = 0;
  Reader(const uint8_t* b, size_t n) : p(b), end(b + n) {}
  uint8_t byte() {
    if (p >= end) { ok = false; return 0; }
    return *p++;
  }
  uint64_t varint() {
    uint64_t v = 0;
    for (int shift = 0; shift < 64; shift += 7) {
      const uint8_t b = byte();
      if (!ok) return 0;
      v |= (uint64_t)(b & 0x7F) << shift;
      if (!(b & 0x80)) return v;
    }
    ok = false;
    return 0;
  }
  int64_t zigzag() {
    const uint64_t v = varint();
    return (int64_t)(v >> 1) ^ -(int64_t)(v & 1);
  }
  void skip_bytes(uint64_t n) {
    if ((uint64_t)(end - p) < n) { ok = false; return; }
    p += n;
  }
  // next field of the current struct: false at STOP.  *last_id carries the id context.
  bool field(int* last_id, int* id, int* type) {
    const uint8_t h = byte();
    if (!ok || h == T_STOP) return false;
    *type = h & 0x0F;
    const int delta = h >> 4;
    *id = delta ? *last_id + delta : (int)zigzag();
    *last_id = *id;
    return ok;
  }
  void skip_element(int type) {
    if (type == T_TRUE || type == T_FALSE) byte(); else skip(type);
  }
  void skip(int type) {
    if (!ok) return;
    if (++depth > 16) { ok = false; return; }
    switch (type) {
      case T_TRUE: case T_FALSE: break;  // the value lives in the field header
      case T_BYTE: byte(); break;
      case T_I16: case T_I32: case T_I64: varint(); break;
      case T_DOUBLE: skip_bytes(8); break;
      case T_BINARY: skip_bytes(varint()); break;
      // containers: every element takes at least one byte (a bool inside a container is a byte of
      // its own, unlike a bool field), so a count beyond the bytes left is corrupt -- without this
      // bound a map<bool, bool> with a 2^60 count consumed nothing per element and never ended
      case T_LIST: case T_SET: {
        const uint8_t h = byte();
        uint64_t n = h >> 4;
        if (n == 15) n = varint();
        if (n > (uint64_t)(end - p)) { ok = false; break; }
        const int et = h & 0x0F;
        for (uint64_t i = 0; i < n && ok; ++i) skip_element(et);
        break;
      }
      case T_MAP: {
        const uint64_t n = varint();
        if (n > (uint64_t)(end - p)) { ok = false; break; }
        if (n) {
          const uint8_t kv = byte();
          for (uint64_t i = 0; i < n && ok; ++i) { skip_element(kv >> 4); skip_element(kv & 0x0F); }
        }
        break;
      }
      case T_STRUCT: {
        int last = 0, id, t;
        while (field(&last, &id, &t)) skip(t);
        break;
      }
      default: ok = false;
    }
    --depth;
  }
  int32_t i32(int type) {
    if (type != T_I32) { ok = false; return 0; }
    return (int32_t)zigzag();
  }
};

struct Writer {
  std::vector<uint8_t>* out;
  explicit Writer(std::vector<uint8_t>* o) : out(o) {}
  void varint(uint64_t v) {
    while (v >= 0x80) { out->push_back((uint8_t)(v | 0x80)); v >>= 7; }
    out->push_back((uint8_t)v);
  }
  void zigzag(int64_t v) { varint(((uint64_t)v << 1) ^ (uint64_t)(v >> 63)); }
  void field(int* last_id, int id, int type) {
    const int delta = id - *last_id;
    if (delta > 0 && delta <= 15) {
      out->push_back((uint8_t)((delta << 4) | type));
    } else {
      out->push_back((uint8_t)type);
      zigzag(id);
    }
    *last_id = id;
  }
  void i32(int* last_id, int id, int32_t v) { field(last_id, id, T_I32); zigzag(v); }
  void stop() { out->push_back(0); }
};
}  // namespace compact

// DeserializeThriftMsg(buffer, &len, /*compact=*/true, &header): *len holds the bytes available
// on entry and the bytes consumed on success.  false: the buffer ends inside the header (the
// caller reads more and retries, hdfs-parquet-scanner.cc:762-798) or the bytes are not a PageHeader.
inline bool DeserializeThriftMsg(const uint8_t* buffer, uint32_t* len, bool compact_protocol,
                                 PageHeader* header) {
  if (!compact_protocol || buffer == nullptr) return false;
  compact::Reader r(buffer, *len);
  *header = PageHeader();
  bool have_type = false, have_usize = false, have_csize = false;
  int last = 0, id, t;
  while (r.field(&last, &id, &t)) {
    switch (id) {
      case 1: header->type = r.i32(t); have_type = true; break;
      case 2: header->uncompressed_page_size = r.i32(t); have_usize = true; break;
      case 3: header->compressed_page_size = r.i32(t); have_csize = true; break;
      case 4: header->crc = r.i32(t); header->__isset.crc = true; break;
      case 5: {
        if (t != compact::T_STRUCT) { r.ok = false; break; }
        DataPageHeader& d = header->data_page_header;
        int l2 = 0, id2, t2, seen = 0;
        while (r.field(&l2, &id2, &t2)) {
          switch (id2) {
            case 1: d.num_values = r.i32(t2); seen |= 1; break;
            case 2: d.encoding = r.i32(t2); seen |= 2; break;
            case 3: d.definition_level_encoding = r.i32(t2); seen |= 4; break;
            case 4: d.repetition_level_encoding = r.i32(t2); seen |= 8; break;
            default: r.skip(t2);  // 5: optional Statistics
          }
        }
        if (seen != 15) r.ok = false;  // four required fields
        header->__isset.data_page_header = true;
        break;
      }
      case 7: {
        if (t != compact::T_STRUCT) { r.ok = false; break; }
        DictionaryPageHeader& d = header->dictionary_page_header;
        int l2 = 0, id2, t2, seen = 0;
        while (r.field(&l2, &id2, &t2)) {
          switch (id2) {
            case 1: d.num_values = r.i32(t2); seen |= 1; break;
            case 2: d.encoding = r.i32(t2); seen |= 2; break;
            case 3: d.is_sorted = t2 == compact::T_TRUE; if (t2 != compact::T_TRUE && t2 != compact::T_FALSE) r.ok = false; break;
            default: r.skip(t2);
          }
        }
        if (seen != 3) r.ok = false;
        header->__isset.dictionary_page_header = true;
        break;
      }
      default: r.skip(t);  // 6 index_page_header, 8 data_page_header_v2, anything newer
    }
    if (!r.ok) return false;
  }
  if (!r.ok || !have_type || !have_usize || !have_csize) return false;
  *len = (uint32_t)(r.p - buffer);
  return true;
}

// ThriftSerializer(/*compact=*/true)::Serialize(&header, ..): the bytes the table writer puts in
// front of every page (hdfs-parquet-table-writer.cc:519-522, 605-608).
inline void SerializePageHeader(const PageHeader& h, std::vector<uint8_t>* out) {
  compact::Writer w(out);
  int last = 0;
  w.i32(&last, 1, h.type);
  w.i32(&last, 2, h.uncompressed_page_size);
  w.i32(&last, 3, h.compressed_page_size);
  if (h.__isset.crc) w.i32(&last, 4, h.crc);
  if (h.__isset.data_page_header) {
    w.field(&last, 5, compact::T_STRUCT);
    int l2 = 0;
    w.i32(&l2, 1, h.data_page_header.num_values);
    w.i32(&l2, 2, h.data_page_header.encoding);
    w.i32(&l2, 3, h.data_page_header.definition_level_encoding);
    w.i32(&l2, 4, h.data_page_header.repetition_level_encoding);
    w.stop();
  }
  if (h.__isset.dictionary_page_header) {
    w.field(&last, 7, compact::T_STRUCT);
    int l2 = 0;
    w.i32(&l2, 1, h.dictionary_page_header.num_values);
    w.i32(&l2, 2, h.dictionary_page_header.encoding);
    w.field(&l2, 3, h.dictionary_page_header.is_sorted ? compact::T_TRUE : compact::T_FALSE);
    w.stop();
  }
  w.stop();
}

// ---- page codecs (Codec::ProcessBlock32 of the decompressor_ / compressor_ the reference creates
// from metadata_->codec, hdfs-parquet-scanner.cc:830-836, 866-876) -------------------------------
inline bool CodecSupported(int codec) {
  return codec == CompressionCodec::UNCOMPRESSED || codec == CompressionCodec::GZIP ||
         codec == CompressionCodec::SNAPPY;
}

// input -> exactly uncompressed_size bytes of output; false on corrupt data or a size mismatch
inline bool Decompress(int codec, const uint8_t* in, int64_t in_len, int64_t uncompressed_size,
                       std::vector<uint8_t>* out) {
  if (in_len < 0 || uncompressed_size < 0) return false;
  // a corrupt header must not turn into a multi-gigabyte allocation: pages are megabytes; the
  // sizes are int32 in the thrift struct, half of that range is refused outright
  constexpr int64_t kMaxPageBytes = 1ll << 30;
  if (uncompressed_size > kMaxPageBytes) return false;
  if (codec == CompressionCodec::UNCOMPRESSED) {
    if (in_len != uncompressed_size) return false;
    out->assign(in, in + in_len);
    return true;
  }
  if (codec == CompressionCodec::SNAPPY) {
    int hdr = 0;
    if (snappy::UncompressedLength(in, in_len, &hdr) != uncompressed_size) return false;  // before allocating
    out->assign((size_t)uncompressed_size, 0);
    return snappy::Uncompress(in, in_len, out->data(), uncompressed_size);
  }
  if (codec != CompressionCodec::GZIP) return false;  // LZO
  out->assign((size_t)uncompressed_size + 1, 0);
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  if (inflateInit2(&zs, 15 + 32) != Z_OK) return false;  // zlib or gzip framing, auto-detected
  zs.next_in = const_cast<Bytef*>(in);
  zs.avail_in = (uInt)in_len;
  zs.next_out = out->data();
  zs.avail_out = (uInt)out->size();
  const int rc = inflate(&zs, Z_FINISH);
  const uint64_t produced = zs.total_out;
  inflateEnd(&zs);
  if (rc != Z_STREAM_END || (int64_t)produced != uncompressed_size) return false;
  out->resize((size_t)uncompressed_size);
  return true;
}

inline bool Compress(int codec, const uint8_t* in, int64_t in_len, std::vector<uint8_t>* out) {
  if (codec == CompressionCodec::UNCOMPRESSED) { out->assign(in, in + in_len); return true; }
  if (codec == CompressionCodec::SNAPPY) { snappy::Compress(in, in_len, out); return true; }
  if (codec != CompressionCodec::GZIP) return false;
  uLongf cap = compressBound((uLong)in_len) + 32;
  out->assign((size_t)cap, 0);
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
  zs.next_in = const_cast<Bytef*>(in);
  zs.avail_in = (uInt)in_len;
  zs.next_out = out->data();
  zs.avail_out = (uInt)out->size();
  const int rc = deflate(&zs, Z_FINISH);
  const uint64_t produced = zs.total_out;
  deflateEnd(&zs);
  if (rc != Z_STREAM_END) return false;
  out->resize((size_t)produced);
  return true;
}

}  // namespace parquet
}  // namespace impala
