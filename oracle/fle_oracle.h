/*
 * oracle/fle_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference's bit-sliced "FLE" codec, sorted-dictionary
 * codec, PLAIN-page predicates and the scanner's bitmap logic.  It is the checker for the
 * HIP path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product (libips_hip.so and the host facade) never links, loads or calls anything here.
 *
 * Parity status: the reference itself is UNBUILDABLE in this image (fle-encoding.h needs Boost
 * bind/function/dynamic_bitset and Impala's common/, util/ headers, none of which exist here, and
 * stand-ins are not allowed), so this restatement is pinned by the reference's own test data:
 *   - fle-test.cc:202-275  round trips + encoded lengths (16 B @w=1, 32 B @w=2 for 100 values),
 *   - fle-test.cc:216-223,235-236  commented golden words (0x3fff, 0xfffffffff0000000,
 *     0x5555555555555555, 0x5555555550000000) on their defined bits,
 *   - dict-test.cc:32-157  dictionary round trips + entry counts,
 *   - the known-answer words recorded in SURVEY.md section 0.
 * The predicates Eq/Lt/Le/Gt/Ge/In have NO test in the reference: "parity unpinned" by reference
 * fixtures; they are pinned here against an independent row-at-a-time model (tests/).
 *
 * Every function cites the reference file:line it follows (paths relative to the reference root).
 */
#ifndef IPS_ORACLE_FLE_ORACLE_H
#define IPS_ORACLE_FLE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same numbering as include/ips.h (kept separate on purpose: the oracle shares no code with it). */
enum { ORC_OP_EQ = 0, ORC_OP_LT = 1, ORC_OP_LE = 2, ORC_OP_GT = 3, ORC_OP_GE = 4, ORC_OP_IN = 5 };
enum { ORC_T_INT8 = 0, ORC_T_INT16 = 1, ORC_T_INT32 = 2, ORC_T_INT64 = 3, ORC_T_FLOAT = 4,
       ORC_T_DOUBLE = 5 };
enum { ORC_SEM_REFERENCE = 0, ORC_SEM_SQL = 1 };
/* result kind of a dictionary literal->code translation */
enum { ORC_XL_ALL_FALSE = 0, ORC_XL_ALL_TRUE = 1, ORC_XL_FLE = 2 };

/* ---- boost::dynamic_bitset<> behaviour the reference relies on (LSB-first 64-bit blocks) ---- */
typedef struct {
  uint64_t* words;
  int64_t nbits;
  int64_t cap_words;
} orc_bitset;

void orc_bitset_init(orc_bitset* b);
void orc_bitset_free(orc_bitset* b);
void orc_bitset_clear(orc_bitset* b);
void orc_bitset_push_back(orc_bitset* b, int bit);
void orc_bitset_append(orc_bitset* b, uint64_t block);
void orc_bitset_resize(orc_bitset* b, int64_t nbits, int value);
int64_t orc_bitset_count(const orc_bitset* b);
int orc_bitset_test(const orc_bitset* b, int64_t i);
void orc_bitset_and(orc_bitset* a, const orc_bitset* b);
void orc_bitset_or(orc_bitset* a, const orc_bitset* b);

/* ---- FLE codec ---- */
int64_t orc_fle_encoded_bytes(int64_t n_rows, int bw);
int orc_bit_width_for_entries(int64_t num_entries);
int orc_log2_ceil(uint64_t x);
void orc_fle_encode(const uint32_t* values, int64_t n, int bw, uint64_t* enc);
void orc_fle_decode(const uint64_t* enc, int64_t n, int bw, uint32_t* out);

/* Stateful decoder with the reference's cursor semantics. */
typedef struct {
  const uint64_t* buffer;
  const uint64_t* buffer_end;   /* next block to unpack */
  const uint64_t* buffer_guard; /* buffer + buffer_len */
  int bit_width;
  int count;                    /* position inside the unpacked block; 64 = none unpacked */
  uint32_t current[64];
} orc_fle_decoder;

void orc_fle_decoder_init(orc_fle_decoder* d, const uint8_t* buffer, int64_t buffer_len, int bw);
int orc_fle_get(orc_fle_decoder* d, uint64_t* val);
int orc_fle_get_skip(orc_fle_decoder* d, uint64_t* val, int skip_rows);
int orc_fle_skip(orc_fle_decoder* d, int skip_rows);
/* Non-advancing predicates; append num_rows bits to 'out'. */
void orc_fle_pred(const orc_fle_decoder* d, int op, int64_t num_rows, orc_bitset* out,
                  const uint64_t* values, int n_values);

/* Stateless convenience: rows [0,n) of a fresh decoder -> ceil(n/64) words, tail bits zero. */
void orc_fle_pred_words(const uint64_t* enc, int64_t n, int bw, int op, const uint64_t* values,
                        int n_values, uint64_t* bitmap_words);

/* ---- dictionary codec ---- */
/* Writer side: sort + remap.  'values' are n raw elements of 'type'; returns number of entries,
 * writes the sorted dictionary page (PLAIN) into dict_page (capacity n elements) and the
 * remapped codes into codes[n].  Returns -1 when the 40000-entry cap is exceeded. */
int64_t orc_dict_build(const void* values, int64_t n, int type, void* dict_page, uint32_t* codes);
/* Data page = [uint8 bw][FLE(codes)]; returns total bytes written. */
int64_t orc_dict_write_data(const uint32_t* codes, int64_t n, int64_t num_entries, uint8_t* page);
/* literal -> code translation; returns ORC_XL_*; fle_op / codes valid for ORC_XL_FLE. */
int orc_dict_translate(const void* dict, int64_t num_entries, int type, int op,
                       const void* literals, int n_literals, int* fle_op, uint64_t* codes,
                       int* n_codes);
/* Full predicate on a REQUIRED dictionary data page (payload after def levels). */
void orc_dict_pred_words(const void* dict, int64_t num_entries, int type, const uint8_t* data_page,
                         int64_t page_len, int64_t n, int op, const void* literals, int n_literals,
                         uint64_t* bitmap_words);
/* value_r = dict[code_r]; returns 0 on an out-of-range code. */
int orc_dict_decode(const void* dict, int64_t num_entries, int type, const uint8_t* data_page,
                    int64_t page_len, int64_t n, void* out);

/* ---- PLAIN fixed-width pages ---- */
int orc_plain_stride(int type);
void orc_plain_pred_words(const uint8_t* page, int64_t n, int type, int op, const void* literals,
                          int n_literals, int semantics, uint64_t* bitmap_words);

/* ---- scanner-level bitmap logic ---- */
void orc_bitmap_and(uint64_t* a, const uint64_t* b, int64_t n_words);
void orc_bitmap_or(uint64_t* a, const uint64_t* b, int64_t n_words);
/* IntersectBitset: j-th set bit of root takes sub[j]. */
void orc_bitmap_expand(const uint64_t* root, const uint64_t* sub, int64_t n_rows, uint64_t* out);
/* bitmap -> skip list; returns number of selected rows. */
int64_t orc_skip_list(const uint64_t* bitmap, int64_t n_rows, int32_t* skip_rows,
                      int64_t* last_skip_rows);
/* Late materialisation of one FLE column through Get(val, skip) / Skip(last). */
int64_t orc_fle_select(const uint64_t* enc, int64_t enc_bytes, int64_t n, int bw,
                       const uint64_t* bitmap, uint32_t* out);

/* ---- timed CPU baseline (bench.py cpu_baseline leg only) ---- */
/* Fused workload on 'threads' host threads, one 64-row-aligned stripe per thread:
 * predicate -> bitmap, then decode of the selected rows (blocks with >=1 selected row are
 * unpacked, as the reference's Get(val, skip) does).  mode 0 = one call per stripe,
 * mode 1 = reference-shaped 1024-row batches.  Returns selected-row count. */
int64_t orc_bench_fused(const uint64_t* enc, int64_t n, int bw, int op, uint64_t value,
                        int threads, int mode, uint64_t* bitmap, uint32_t* sel_out);
/* best-of-reps seconds of orc_bench_fused after a warm-up pass, buffers pre-touched by the caller */
double orc_bench_fused_best(const uint64_t* enc, int64_t n, int bw, int op, uint64_t value,
                            int threads, int mode, uint64_t* bitmap, uint32_t* sel_out, int reps,
                            int64_t* n_sel);
void orc_fast_unpack_block(const uint64_t* blk, int bw, uint32_t* out64);
void orc_swar_unpack_block(const uint64_t* blk, int bw, uint32_t* out64);
int orc_hw_threads(void);
int orc_has_avx2(void);

#ifdef __cplusplus
}
#endif
#endif
