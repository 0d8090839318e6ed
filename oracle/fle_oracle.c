/*
 * oracle/fle_oracle.c -- TEST INFRASTRUCTURE ONLY (see fle_oracle.h for the rules and the
 * parity status: the reference is unbuildable in this image; this restatement is pinned by the
 * reference's own test data and by an independent row-at-a-time model in tests/).
 *
 * Plain C restatement of the algorithms of the reference hot path.  Citations are
 * reference-relative file:line.
 */
#include "fle_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* dynamic_bitset behaviour (Boost's documented contract: LSB-first blocks; append(block) puts */
/* bit k of the block at index size()+k).  Call sites: fle-encoding.h:7975,8007,8063;           */
/* dict-encoding.h:466,476; hdfs-parquet-scanner.cc:329,344,1125,1139; simple-predicates.h:152. */
/* ------------------------------------------------------------------------------------------ */
static void bitset_reserve(orc_bitset* b, int64_t nbits) {
  int64_t need = (nbits + 63) / 64 + 1;
  if (need <= b->cap_words) return;
  int64_t cap = b->cap_words ? b->cap_words : 16;
  while (cap < need) cap *= 2;
  b->words = (uint64_t*)realloc(b->words, (size_t)cap * 8);
  memset(b->words + b->cap_words, 0, (size_t)(cap - b->cap_words) * 8);
  b->cap_words = cap;
}

void orc_bitset_init(orc_bitset* b) { b->words = NULL; b->nbits = 0; b->cap_words = 0; }
void orc_bitset_free(orc_bitset* b) { free(b->words); orc_bitset_init(b); }
void orc_bitset_clear(orc_bitset* b) {
  if (b->words) memset(b->words, 0, (size_t)b->cap_words * 8);
  b->nbits = 0;
}
void orc_bitset_push_back(orc_bitset* b, int bit) {
  bitset_reserve(b, b->nbits + 1);
  if (bit) b->words[b->nbits >> 6] |= 1ULL << (b->nbits & 63);
  b->nbits += 1;
}
void orc_bitset_append(orc_bitset* b, uint64_t block) {
  bitset_reserve(b, b->nbits + 64);
  int sh = (int)(b->nbits & 63);
  int64_t w = b->nbits >> 6;
  b->words[w] |= block << sh;
  if (sh) b->words[w + 1] |= block >> (64 - sh);
  b->nbits += 64;
}
void orc_bitset_resize(orc_bitset* b, int64_t nbits, int value) {
  if (nbits > b->nbits) {
    bitset_reserve(b, nbits);
    if (value) {
      for (int64_t i = b->nbits; i < nbits; ++i) b->words[i >> 6] |= 1ULL << (i & 63);
    }
  } else {
    for (int64_t i = nbits; i < b->nbits; ++i) b->words[i >> 6] &= ~(1ULL << (i & 63));
  }
  b->nbits = nbits;
}
int64_t orc_bitset_count(const orc_bitset* b) {
  int64_t c = 0;
  for (int64_t w = 0; w < (b->nbits + 63) / 64; ++w) c += __builtin_popcountll(b->words[w]);
  return c;
}
int orc_bitset_test(const orc_bitset* b, int64_t i) { return (int)((b->words[i >> 6] >> (i & 63)) & 1); }
void orc_bitset_and(orc_bitset* a, const orc_bitset* b) {
  for (int64_t w = 0; w < (a->nbits + 63) / 64; ++w) a->words[w] &= b->words[w];
}
void orc_bitset_or(orc_bitset* a, const orc_bitset* b) {
  for (int64_t w = 0; w < (a->nbits + 63) / 64; ++w) a->words[w] |= b->words[w];
}

/* ------------------------------------------------------------------------------------------ */
/* FLE layout                                                                                  */
/* ------------------------------------------------------------------------------------------ */

/* FleEncoder::Flush length, fle-encoding.h:9806-9812; asserted by fle-test.cc:219,224,238. */
int64_t orc_fle_encoded_bytes(int64_t n_rows, int bw) { return ((n_rows + 63) / 64) * (int64_t)bw * 8; }

/* BitUtil::Log2 = ceil(log2 x), bit-util.h:128-140. */
int orc_log2_ceil(uint64_t x) {
  if (x <= 1) return 0;
  --x;
  int result = 1;
  while (x >>= 1) ++result;
  return result;
}

/* DictEncoderBase::bit_width, dict-encoding.h:76-80. */
int orc_bit_width_for_entries(int64_t num_entries) {
  if (num_entries == 0) return 0;
  if (num_entries == 1) return 1;
  return orc_log2_ceil((uint64_t)num_entries);
}

/* Layout statement: fle-encoding.h:8338-8340 (commented Put), fle-benchmark.cc:395-411.
 * Word i of block b holds bit i of the 64 values; value k sits at bit 63-k.  Padding rows of the
 * last block are zero here (the reference leaves them undefined). */
void orc_fle_encode(const uint32_t* values, int64_t n, int bw, uint64_t* enc) {
  int64_t blocks = (n + 63) / 64;
  memset(enc, 0, (size_t)(blocks * bw) * 8);
  for (int64_t r = 0; r < n; ++r) {
    uint64_t* blk = enc + (r >> 6) * bw;
    int k = (int)(r & 63);
    for (int l = 0; l < bw; ++l) blk[l] |= (uint64_t)((values[r] >> l) & 1u) << (63 - k);
  }
}

/* One block, scalar: fle-encoding.h:425-430 (commented Get body), fle-benchmark.cc:458-474. */
static void unpack_block(const uint64_t* blk, int bw, uint32_t* out64) {
  for (int k = 0; k < 64; ++k) {
    uint32_t v = 0;
    for (int i = 0; i < bw; ++i) v |= (uint32_t)((blk[i] >> (63 - k)) & 1ULL) << i;
    out64[k] = v;
  }
}

void orc_fle_decode(const uint64_t* enc, int64_t n, int bw, uint32_t* out) {
  uint32_t tmp[64];
  for (int64_t b = 0; b * 64 < n; ++b) {
    unpack_block(enc + b * bw, bw, tmp);
    int64_t m = n - b * 64 < 64 ? n - b * 64 : 64;
    memcpy(out + b * 64, tmp, (size_t)m * 4);
  }
}

/* ------------------------------------------------------------------------------------------ */
/* Stateful decoder: cursor semantics of fle-encoding.h:344-402 (Get(val,skip), Skip) and      */
/* :404-567 (Get).                                                                             */
/* ------------------------------------------------------------------------------------------ */
void orc_fle_decoder_init(orc_fle_decoder* d, const uint8_t* buffer, int64_t buffer_len, int bw) {
  /* ctor fle-encoding.h:37-48: count_ = 64 forces an unpack on first Get. */
  d->buffer = (const uint64_t*)buffer;
  d->buffer_end = d->buffer;
  d->buffer_guard = (const uint64_t*)(buffer + buffer_len);
  d->bit_width = bw;
  d->count = 64;
  memset(d->current, 0, sizeof(d->current));
}

static void refill(orc_fle_decoder* d) {
  unpack_block(d->buffer_end, d->bit_width, d->current);
  d->buffer_end += d->bit_width;
}

/* fle-encoding.h:404-567 */
int orc_fle_get(orc_fle_decoder* d, uint64_t* val) {
  if (d->count == 64) {
    if (d->buffer_end >= d->buffer_guard) return 0;
    d->count = 0;
    refill(d);
  }
  *val = d->current[d->count];
  ++d->count;
  return 1;
}

/* shared cursor advance of fle-encoding.h:346-363 and :382-399 */
static int advance(orc_fle_decoder* d, int skip_rows) {
  if (d->count == 64) {
    d->count = skip_rows & 63;
    int skip_64 = (skip_rows & ~63) >> 6;
    d->buffer_end += (int64_t)d->bit_width * skip_64;
    if (d->buffer_end >= d->buffer_guard) return 0;
    refill(d);
  } else {
    skip_rows += d->count;
    d->count = skip_rows & 63;
    if (skip_rows >= 64) {
      int skip_64 = ((skip_rows & ~63) >> 6) - 1;
      d->buffer_end += (int64_t)d->bit_width * skip_64;
      if (d->buffer_end >= d->buffer_guard) return 0;
      refill(d);
    }
  }
  return 1;
}

/* fle-encoding.h:344-379 */
int orc_fle_get_skip(orc_fle_decoder* d, uint64_t* val, int skip_rows) {
  if (!advance(d, skip_rows)) return 0;
  *val = d->current[d->count];
  ++d->count;
  return 1;
}

/* fle-encoding.h:381-402 */
int orc_fle_skip(orc_fle_decoder* d, int skip_rows) { return advance(d, skip_rows); }

/* 64-bit reversal, the six swap steps of fle-encoding.h:8045-8054. */
static uint64_t bitrev64(uint64_t m) {
  m = ((m >> 1) & 0x5555555555555555ULL) | ((m & 0x5555555555555555ULL) << 1);
  m = ((m >> 2) & 0x3333333333333333ULL) | ((m & 0x3333333333333333ULL) << 2);
  m = ((m >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((m & 0x0F0F0F0F0F0F0F0FULL) << 4);
  m = ((m >> 8) & 0x00FF00FF00FF00FFULL) | ((m & 0x00FF00FF00FF00FFULL) << 8);
  m = ((m >> 16) & 0x0000FFFF0000FFFFULL) | ((m & 0x0000FFFF0000FFFFULL) << 16);
  return (m >> 32) | (m << 32);
}

/* C[i] of fle-encoding.h:8015-8021: "value & 0x01 << i" is an int shift, so for i == 31 the mask
 * sign-extends (quirk Q6); restated literally. */
static uint64_t const_plane(uint64_t value, int i) {
  uint64_t m = (uint64_t)(int64_t)(int32_t)(1u << i);
  return (value & m) ? ~0ULL : 0ULL;
}

/* One block -> (Mlt, Meq, Mgt) MSB->LSB recurrence, fle-encoding.h:8035-8042 (Lt),
 * :8147-8154 (Gt), :7985-7988 (Eq). Row k is at bit 63-k. */
static void block_masks(const uint64_t* blk, int bw, uint64_t value, uint64_t* mlt, uint64_t* meq,
                        uint64_t* mgt) {
  uint64_t lt = 0, gt = 0, eq = ~0ULL;
  for (int i = bw - 1; i >= 0; --i) {
    uint64_t c = const_plane(value, i);
    uint64_t x = blk[i];
    lt |= eq & c & ~x;
    gt |= eq & ~c & x;
    eq &= ~(x ^ c);
  }
  *mlt = lt; *meq = eq; *mgt = gt;
}

static int scalar_cmp(int op, uint64_t x, const uint64_t* values, int n_values) {
  switch (op) {
    case ORC_OP_EQ: return x == values[0];
    case ORC_OP_LT: return x < values[0];
    case ORC_OP_LE: return x <= values[0];
    case ORC_OP_GT: return x > values[0];
    case ORC_OP_GE: return x >= values[0];
    default: {
      int f = 0;
      for (int j = 0; j < n_values; ++j) f = f || (x == values[j]);
      return f;
    }
  }
}

/* FleDecoder::Eq/Lt/Le/Gt/Ge/In, fle-encoding.h:7962-8313: prefix = leftover of the unpacked
 * block compared value-wise (:8023-8031), then whole blocks from buffer_end_ on the encoded
 * words, reversed and appended; ragged tail pushes the first num_rows bits (:8056-8062).
 * Non-advancing: the decoder is not modified. */
void orc_fle_pred(const orc_fle_decoder* d, int op, int64_t num_rows, orc_bitset* out,
                  const uint64_t* values, int n_values) {
  for (int i = d->count; i != 64 && num_rows > 0; ++i, --num_rows)
    orc_bitset_push_back(out, scalar_cmp(op, d->current[i], values, n_values));

  const uint64_t* blk = d->buffer_end;
  while (num_rows > 0) {
    uint64_t m;
    if (op == ORC_OP_IN) {
      m = 0;
      for (int v = 0; v < n_values; ++v) {
        uint64_t lt, eq, gt;
        block_masks(blk, d->bit_width, values[v], &lt, &eq, &gt);
        m |= eq;
      }
    } else {
      uint64_t lt, eq, gt;
      block_masks(blk, d->bit_width, values[0], &lt, &eq, &gt);
      switch (op) {
        case ORC_OP_EQ: m = eq; break;
        case ORC_OP_LT: m = lt; break;
        case ORC_OP_LE: m = lt | eq; break;
        case ORC_OP_GT: m = gt; break;
        default: m = gt | eq; break;
      }
    }
    blk += d->bit_width;
    m = bitrev64(m);
    if (num_rows < 64) {
      for (int i = 0; i < num_rows; ++i) orc_bitset_push_back(out, (int)((m >> i) & 1));
      break;
    }
    orc_bitset_append(out, m);
    num_rows -= 64;
  }
}

void orc_fle_pred_words(const uint64_t* enc, int64_t n, int bw, int op, const uint64_t* values,
                        int n_values, uint64_t* bitmap_words) {
  orc_fle_decoder d;
  orc_fle_decoder_init(&d, (const uint8_t*)enc, orc_fle_encoded_bytes(n, bw), bw);
  orc_bitset bs;
  orc_bitset_init(&bs);
  orc_fle_pred(&d, op, n, &bs, values, n_values);
  int64_t nw = (n + 63) / 64;
  memset(bitmap_words, 0, (size_t)nw * 8);
  if (bs.words) memcpy(bitmap_words, bs.words, (size_t)nw * 8);
  orc_bitset_free(&bs);
}

/* ------------------------------------------------------------------------------------------ */
/* Dictionary codec                                                                            */
/* ------------------------------------------------------------------------------------------ */
static int type_size(int type) {
  switch (type) {
    case ORC_T_INT8: return 1;
    case ORC_T_INT16: return 2;
    case ORC_T_INT32: case ORC_T_FLOAT: return 4;
    default: return 8;
  }
}

/* operator< of T (signed ints, IEEE floats), dict-encoding.h:370-372. */
static int elem_less(int type, const void* a, const void* b) {
  switch (type) {
    case ORC_T_INT8: return *(const int8_t*)a < *(const int8_t*)b;
    case ORC_T_INT16: return *(const int16_t*)a < *(const int16_t*)b;
    case ORC_T_INT32: return *(const int32_t*)a < *(const int32_t*)b;
    case ORC_T_INT64: return *(const int64_t*)a < *(const int64_t*)b;
    case ORC_T_FLOAT: return *(const float*)a < *(const float*)b;
    default: return *(const double*)a < *(const double*)b;
  }
}

static __thread int g_sort_type;
static int qsort_cmp(const void* a, const void* b) {
  if (elem_less(g_sort_type, a, b)) return -1;
  if (elem_less(g_sort_type, b, a)) return 1;
  return 0;
}

/* std::lower_bound / upper_bound as used by dict-encoding.h:463,480,493,506,518,528. */
static int64_t lower_bound(const uint8_t* dict, int64_t n, int type, const void* v) {
  int sz = type_size(type);
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    int64_t mid = lo + (hi - lo) / 2;
    if (elem_less(type, dict + mid * sz, v)) lo = mid + 1; else hi = mid;
  }
  return lo;
}
static int64_t upper_bound(const uint8_t* dict, int64_t n, int type, const void* v) {
  int sz = type_size(type);
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    int64_t mid = lo + (hi - lo) / 2;
    if (!elem_less(type, v, dict + mid * sz)) lo = mid + 1; else hi = mid;
  }
  return lo;
}

/* DictEncoder::Put + WriteDict (sort, dict-encoding.h:393-406) + index remap (:408-411,434).
 * The dictionary page is PLAIN: sizeof(T) bytes per entry (parquet-common.h:87-88,169-173);
 * int8/int16 dictionary entries are written with ByteSize == 4 (parquet-common.h:304-306). */
int64_t orc_dict_build(const void* values, int64_t n, int type, void* dict_page, uint32_t* codes) {
  int sz = type_size(type);
  uint8_t* sorted = (uint8_t*)malloc((size_t)(n > 0 ? n : 1) * sz);
  memcpy(sorted, values, (size_t)n * sz);
  g_sort_type = type;
  qsort(sorted, (size_t)n, (size_t)sz, qsort_cmp);
  int64_t d = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (d == 0 || elem_less(type, sorted + (d - 1) * sz, sorted + i * sz)) {
      memmove(sorted + d * sz, sorted + i * sz, (size_t)sz);
      ++d;
    }
  }
  if (d > 40000) { free(sorted); return -1; } /* dict-encoding.h:157 */
  for (int64_t i = 0; i < n; ++i)
    codes[i] = (uint32_t)lower_bound(sorted, d, type, (const uint8_t*)values + i * sz);
  int page_sz = (type == ORC_T_INT8 || type == ORC_T_INT16) ? 4 : sz;
  uint8_t* page = (uint8_t*)dict_page;
  for (int64_t i = 0; i < d; ++i) {
    /* parquet-common.h:170-173 copies ByteSize(t) = 4 bytes out of a 1- or 2-byte object, so the
     * upper bytes of a narrow slot are unspecified in the reference (Decode, :319-322, reads the
     * low byte(s) only); the restatement writes the value as the int32 Parquet declares. */
    if (type == ORC_T_INT8) { int32_t wide = *(const int8_t*)(sorted + i * sz); memcpy(page + i * 4, &wide, 4); }
    else if (type == ORC_T_INT16) { int32_t wide; int16_t v; memcpy(&v, sorted + i * sz, 2); wide = v; memcpy(page + i * 4, &wide, 4); }
    else memcpy(page + i * page_sz, sorted + i * sz, (size_t)sz);
  }
  free(sorted);
  return d;
}

/* DictEncoderBase::WriteData, dict-encoding.h:408-423: [uint8 bit_width][FLE blocks]. */
int64_t orc_dict_write_data(const uint32_t* codes, int64_t n, int64_t num_entries, uint8_t* page) {
  int bw = orc_bit_width_for_entries(num_entries);
  page[0] = (uint8_t)bw;
  int64_t bytes = orc_fle_encoded_bytes(n, bw);
  if (bytes > 0) {
    uint64_t* tmp = (uint64_t*)malloc((size_t)bytes);
    orc_fle_encode(codes, n, bw, tmp);
    memcpy(page + 1, tmp, (size_t)bytes);
    free(tmp);
  }
  return 1 + bytes;
}

/* DictDecoder<T>::Eq/Lt/Le/Gt/Ge/In literal -> code translation, dict-encoding.h:461-541.
 * 'dict' is the decoded dictionary: num_entries elements of sizeof(T). */
int orc_dict_translate(const void* dict, int64_t D, int type, int op, const void* literals,
                       int n_literals, int* fle_op, uint64_t* codes, int* n_codes) {
  const uint8_t* dd = (const uint8_t*)dict;
  int sz = type_size(type);
  const void* v = literals;
  *n_codes = 0;
  if (D == 0) return ORC_XL_ALL_FALSE; /* guard for quirk Q12 */
  const void* first = dd;
  const void* last = dd + (D - 1) * sz;
  switch (op) {
    case ORC_OP_EQ: { /* :461-470 */
      int64_t lb = lower_bound(dd, D, type, v);
      if (lb == D || elem_less(type, v, dd + lb * sz)) return ORC_XL_ALL_FALSE;
      *fle_op = ORC_OP_EQ; codes[0] = (uint64_t)lb; *n_codes = 1;
      return ORC_XL_FLE;
    }
    case ORC_OP_GT: /* :472-483 */
      if (!elem_less(type, v, last)) return ORC_XL_ALL_FALSE;      /* back() <= val */
      if (elem_less(type, v, first)) return ORC_XL_ALL_TRUE;       /* dict[0] > val */
      *fle_op = ORC_OP_GE; codes[0] = (uint64_t)upper_bound(dd, D, type, v); *n_codes = 1;
      return ORC_XL_FLE;
    case ORC_OP_LT: /* :485-496 */
      if (!elem_less(type, first, v)) return ORC_XL_ALL_FALSE;     /* dict[0] >= val */
      if (elem_less(type, last, v)) return ORC_XL_ALL_TRUE;        /* back() < val */
      *fle_op = ORC_OP_LT; codes[0] = (uint64_t)lower_bound(dd, D, type, v); *n_codes = 1;
      return ORC_XL_FLE;
    case ORC_OP_GE: /* :498-509 */
      if (elem_less(type, last, v)) return ORC_XL_ALL_FALSE;       /* back() < val */
      if (!elem_less(type, first, v)) return ORC_XL_ALL_TRUE;      /* dict[0] >= val */
      *fle_op = ORC_OP_GE; codes[0] = (uint64_t)lower_bound(dd, D, type, v); *n_codes = 1;
      return ORC_XL_FLE;
    case ORC_OP_LE: /* :511-521 */
      if (elem_less(type, v, first)) return ORC_XL_ALL_FALSE;      /* dict[0] > val */
      if (!elem_less(type, v, last)) return ORC_XL_ALL_TRUE;       /* back() <= val */
      *fle_op = ORC_OP_LT; codes[0] = (uint64_t)upper_bound(dd, D, type, v); *n_codes = 1;
      return ORC_XL_FLE;
    default: { /* In, :523-541 */
      const uint8_t* lv = (const uint8_t*)literals;
      for (int i = 0; i < n_literals; ++i) {
        int64_t lb = lower_bound(dd, D, type, lv + i * sz);
        if (lb == D || elem_less(type, lv + i * sz, dd + lb * sz)) continue;
        codes[(*n_codes)++] = (uint64_t)lb;
      }
      if (*n_codes == 0) return ORC_XL_ALL_FALSE;
      *fle_op = ORC_OP_IN;
      return ORC_XL_FLE;
    }
  }
}

/* DictDecoderBase::SetData (dict-encoding.h:185-192): first byte = code width, rest = blocks. */
static uint64_t* page_blocks_aligned(const uint8_t* data_page, int64_t page_len, int* bw) {
  *bw = data_page[0];
  int64_t bytes = page_len - 1;
  uint64_t* buf = (uint64_t*)malloc((size_t)(bytes > 0 ? bytes : 8));
  if (bytes > 0) memcpy(buf, data_page + 1, (size_t)bytes);
  return buf;
}

void orc_dict_pred_words(const void* dict, int64_t D, int type, const uint8_t* data_page,
                         int64_t page_len, int64_t n, int op, const void* literals, int n_literals,
                         uint64_t* bitmap_words) {
  int64_t nw = (n + 63) / 64;
  uint64_t codes[64];
  uint64_t* many = n_literals > 64 ? (uint64_t*)malloc((size_t)n_literals * 8) : codes;
  int fle_op = 0, n_codes = 0;
  int kind = orc_dict_translate(dict, D, type, op, literals, n_literals, &fle_op, many, &n_codes);
  if (kind == ORC_XL_ALL_FALSE) {
    memset(bitmap_words, 0, (size_t)nw * 8); /* resize(num_rows, false) */
  } else if (kind == ORC_XL_ALL_TRUE) {
    memset(bitmap_words, 0, (size_t)nw * 8); /* resize(num_rows, true) */
    for (int64_t i = 0; i < n; ++i) bitmap_words[i >> 6] |= 1ULL << (i & 63);
  } else {
    int bw;
    uint64_t* blocks = page_blocks_aligned(data_page, page_len, &bw);
    orc_fle_pred_words(blocks, n, bw, fle_op, many, n_codes, bitmap_words);
    free(blocks);
  }
  if (many != codes) free(many);
}

/* DictDecoder<T>::GetValue, dict-encoding.h:310-319. */
int orc_dict_decode(const void* dict, int64_t D, int type, const uint8_t* data_page,
                    int64_t page_len, int64_t n, void* out) {
  int bw;
  uint64_t* blocks = page_blocks_aligned(data_page, page_len, &bw);
  int sz = type_size(type);
  orc_fle_decoder d;
  orc_fle_decoder_init(&d, (const uint8_t*)blocks, page_len - 1, bw);
  int ok = 1;
  for (int64_t r = 0; r < n && ok; ++r) {
    uint64_t idx;
    if (!orc_fle_get(&d, &idx) || (int64_t)idx >= D) { ok = 0; break; }
    memcpy((uint8_t*)out + r * sz, (const uint8_t*)dict + idx * sz, (size_t)sz);
  }
  free(blocks);
  return ok;
}

/* ------------------------------------------------------------------------------------------ */
/* PLAIN fixed-width pages                                                                     */
/* ------------------------------------------------------------------------------------------ */
/* ParquetPlainEncoder::ByteSize, parquet-common.h:92-117,304-306. */
int orc_plain_stride(int type) {
  switch (type) {
    case ORC_T_INT8: case ORC_T_INT16: case ORC_T_INT32: case ORC_T_FLOAT: return 4;
    default: return 8;
  }
}

#define PLAIN_LOOP(T)                                                                        \
  do {                                                                                       \
    const T* lit = (const T*)literals;                                                       \
    for (int64_t r = 0; r < n; ++r) {                                                        \
      T x;                                                                                   \
      memcpy(&x, page + r * stride, sizeof(T));                                              \
      int bit = 0;                                                                           \
      T a = ref ? lit[0] : x; /* REFERENCE: val OP x (quirk Q1); SQL: x OP val */            \
      T b = ref ? x : lit[0];                                                                \
      switch (op) {                                                                          \
        case ORC_OP_EQ: bit = a == b; break;                                                 \
        case ORC_OP_LT: bit = a < b; break;                                                  \
        case ORC_OP_LE: bit = a <= b; break;                                                 \
        case ORC_OP_GT: bit = a > b; break;                                                  \
        case ORC_OP_GE: bit = a >= b; break;                                                 \
        default:                                                                             \
          for (int j = 0; j < n_literals; ++j) bit = bit || (x == lit[j]);                   \
      }                                                                                      \
      if (bit) bitmap_words[r >> 6] |= 1ULL << (r & 63);                                     \
    }                                                                                        \
  } while (0)

/* ParquetPlainEncoder::Eq/Lt/Le/Gt/Ge: generic parquet-common.h:197-250, int8 :335-383,
 * int16 :400-449 (only the low 1/2 bytes of the 4-byte slot are compared).  In: the reference
 * body is empty (:252-255, quirk Q2) -- in REFERENCE semantics nothing is produced (words are
 * zeroed here); SQL semantics implements a real IN. */
void orc_plain_pred_words(const uint8_t* page, int64_t n, int type, int op, const void* literals,
                          int n_literals, int semantics, uint64_t* bitmap_words) {
  int64_t nw = (n + 63) / 64;
  memset(bitmap_words, 0, (size_t)nw * 8);
  int ref = semantics == ORC_SEM_REFERENCE;
  if (op == ORC_OP_IN && ref) return;
  int stride = orc_plain_stride(type);
  switch (type) {
    case ORC_T_INT8: PLAIN_LOOP(int8_t); break;
    case ORC_T_INT16: PLAIN_LOOP(int16_t); break;
    case ORC_T_INT32: PLAIN_LOOP(int32_t); break;
    case ORC_T_INT64: PLAIN_LOOP(int64_t); break;
    case ORC_T_FLOAT: PLAIN_LOOP(float); break;
    default: PLAIN_LOOP(double); break;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* Scanner-level bitmap logic                                                                  */
/* ------------------------------------------------------------------------------------------ */
/* AndOperate / OrOperate, simple-predicates.h:145-163; conjunct AND hdfs-parquet-scanner.cc:1861 */
void orc_bitmap_and(uint64_t* a, const uint64_t* b, int64_t n_words) {
  for (int64_t i = 0; i < n_words; ++i) a[i] &= b[i];
}
void orc_bitmap_or(uint64_t* a, const uint64_t* b, int64_t n_words) {
  for (int64_t i = 0; i < n_words; ++i) a[i] |= b[i];
}

/* ColumnReader::IntersectBitset, hdfs-parquet-scanner.cc:326-331. */
void orc_bitmap_expand(const uint64_t* root, const uint64_t* sub, int64_t n_rows, uint64_t* out) {
  int64_t j = -1;
  memset(out, 0, (size_t)((n_rows + 63) / 64) * 8);
  for (int64_t i = 0; i < n_rows; ++i) {
    if ((root[i >> 6] >> (i & 63)) & 1) {
      ++j;
      if ((sub[j >> 6] >> (j & 63)) & 1) out[i >> 6] |= 1ULL << (i & 63);
    }
  }
}

/* bitmap -> skip list, hdfs-parquet-scanner.cc:1134-1148. */
int64_t orc_skip_list(const uint64_t* bitmap, int64_t n_rows, int32_t* skip_rows,
                      int64_t* last_skip_rows) {
  int64_t start = 0, cnt = 0;
  for (int64_t i = 0; i < n_rows; ++i) {
    if ((bitmap[i >> 6] >> (i & 63)) & 1) {
      skip_rows[cnt++] = (int32_t)(i - start);
      start = i + 1;
    }
  }
  *last_skip_rows = n_rows - start;
  return cnt;
}

/* Late materialisation of one column: ReadValue(skip) -> Get(val, skip) per selected row, then
 * SkipValue(last_skip): hdfs-parquet-scanner.cc:1151-1181, :1006-1038; fle-encoding.h:344-402. */
int64_t orc_fle_select(const uint64_t* enc, int64_t enc_bytes, int64_t n, int bw,
                       const uint64_t* bitmap, uint32_t* out) {
  orc_fle_decoder d;
  orc_fle_decoder_init(&d, (const uint8_t*)enc, enc_bytes, bw);
  int64_t start = 0, cnt = 0;
  for (int64_t i = 0; i < n; ++i) {
    if ((bitmap[i >> 6] >> (i & 63)) & 1) {
      uint64_t v;
      if (!orc_fle_get_skip(&d, &v, (int)(i - start))) return -1;
      out[cnt++] = (uint32_t)v;
      start = i + 1;
    }
  }
  if (n - start > 0) orc_fle_skip(&d, (int)(n - start));
  return cnt;
}
