/*
 * include/ips.h -- C-ABI of libips_hip.so: MI355X (gfx950) implementation of the bit-sliced
 * "FLE" / sorted-dictionary decode + vectorised predicate -> selection-bitmap hot path of
 * zuowang/Impala-avx2-parquet-scanner.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch/HIP types.  Every entry
 * point names the reference interface it replaces (file:line relative to the reference root).
 * INTEGRATION.md shows the reference-side binding for each one.
 *
 * Conventions
 *  - "d_" pointers are device (HBM) pointers, 16-byte aligned; "h_" / unprefixed small arrays
 *    are host pointers.  The caller owns every buffer; the library owns only the opaque handles
 *    it returns (the reference decoders likewise never own page bytes, dict-encoding.h:179-181).
 *  - Bitmaps are arrays of little-endian uint64 words, bit (r % 64) of word (r / 64) <-> row r,
 *    1 = row passes (boost::dynamic_bitset<> block order; "skip_bitset" in the reference is a
 *    misnomer, hdfs-parquet-scanner.cc:1139-1143).  ceil(n_rows/64) words are written; bits
 *    >= n_rows of the last word are zero.
 *  - FLE layout (fle-encoding.h:8338-8340, 425-430): blocks of 64 values; a block of bit width w
 *    is w consecutive uint64 words; word i holds bit i of all 64 values, value k at bit 63-k.
 *    Encoded size is ceil(n/64)*w*8 bytes (fle-encoding.h:9806-9812); kernels never read past it.
 *  - All functions return IPS_OK (0) or an error code; ips_last_error() gives the message of
 *    the calling thread's last failure.  No exceptions cross the ABI.  Calls are asynchronous on
 *    'stream' (a hipStream_t passed as void*; NULL = default stream) unless stated.
 *  - Entry points are thread-safe; handles are thread-compatible (one thread at a time), like
 *    the reference's per-scanner-thread decoders (expr-context.h:39-42).
 */
#ifndef IPS_H
#define IPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IPS_VERSION 301 /* 0.3.1: column chunks as page lists (ips_chunk_*), ips_set_program_strategy; .1: ips_chunk_select_nullable */

typedef enum {
  IPS_OK = 0,
  IPS_ERR_INVALID_ARG = 1, /* bad bit width / op / type / NULL / misaligned pointer */
  IPS_ERR_UNSUPPORTED = 2,
  IPS_ERR_HIP = 3,         /* a HIP runtime call failed (no device, launch failure, ...) */
  IPS_ERR_NOMEM = 4,
  IPS_ERR_OUT_OF_DATA = 5, /* FleDecoder::Get returning false, fle-encoding.h:350,407 */
  IPS_ERR_BAD_INDEX = 6    /* dictionary code >= num_entries, dict-encoding.h:316 */
} ips_status;

/* FleDecoder::{Eq,Lt,Le,Gt,Ge,In}, fle-encoding.h:150-155; SimplePredicate leaves,
 * simple-predicates.h:68-143; fn names eq/lt/le/gt/ge/in_set_lookup, scalar-fn-call.cc:945-962 */
typedef enum { IPS_OP_EQ = 0, IPS_OP_LT = 1, IPS_OP_LE = 2, IPS_OP_GT = 3, IPS_OP_GE = 4,
               IPS_OP_IN = 5 } ips_op;

/* Column value types on the path (explicit instantiations hdfs-parquet-scanner.cc:1909-2058,
 * minus string/decimal/timestamp which are out of scope). */
typedef enum { IPS_T_INT8 = 0, IPS_T_INT16 = 1, IPS_T_INT32 = 2, IPS_T_INT64 = 3,
               IPS_T_FLOAT = 4, IPS_T_DOUBLE = 5 } ips_type;

/* PLAIN-page predicate operand order.  REFERENCE reproduces parquet-common.h:203,214,225,236,247
 * (bit = literal OP x, operands reversed); SQL is bit = x OP literal. */
typedef enum { IPS_SEM_REFERENCE = 0, IPS_SEM_SQL = 1 } ips_semantics;

/* Result of a dictionary literal -> code translation (dict-encoding.h:461-541). */
typedef enum { IPS_XL_ALL_FALSE = 0, IPS_XL_ALL_TRUE = 1, IPS_XL_FLE = 2 } ips_xl_kind;

typedef void* ips_stream;
typedef struct ips_dict ips_dict;
typedef struct ips_inset ips_inset; /* an IN list of any length, resident on the device (below) */

/* Rows per selection batch of the fused scan: each batch's selected values are written densely,
 * in row order, at d_batch_values + batch*IPS_BATCH_ROWS (the scanner hands row batches upstream,
 * hdfs-parquet-scanner.cc:1092-1182; the reference batch is 1024 rows, :1838). */
#define IPS_BATCH_ROWS 2048
#define IPS_MAX_IN_LIST 256

/* ---- library / device ---------------------------------------------------------------------- */
int ips_version(void);
const char* ips_last_error(void);
ips_status ips_device_count(int* count);
ips_status ips_set_device(int device);
/* name, CU count, HBM bytes; any out pointer may be NULL */
ips_status ips_device_info(char* name, int name_len, int* compute_units, int64_t* hbm_bytes);

/* Device memory / stream plumbing for hosts that do not link HIP themselves (the C++ facade). */
ips_status ips_malloc(void** d_ptr, size_t bytes);
ips_status ips_free(void* d_ptr);
ips_status ips_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes, ips_stream stream);
ips_status ips_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes, ips_stream stream);
ips_status ips_memset(void* d_dst, int value, size_t bytes, ips_stream stream);
ips_status ips_stream_create(ips_stream* stream);
ips_status ips_stream_destroy(ips_stream stream);
ips_status ips_stream_synchronize(ips_stream stream);

/* ---- FLE codec ----------------------------------------------------------------------------- */
/* FleEncoder::Flush() return value, fle-encoding.h:9806-9812 (fle-test.cc:219,224,238). */
int64_t ips_fle_encoded_bytes(int64_t n_rows, int bit_width);

/* FleEncoder::Put x n + Flush, fle-encoding.h:8315-8365, 9806-9812 (Pack_w :8367-9803).
 * d_values: n_rows unsigned values of in_width bytes (1, 2 or 4), each < 2^bit_width.
 * Padding rows of the last block are written as zero (the reference leaves them undefined). */
ips_status ips_fle_encode(const void* d_values, int in_width, int64_t n_rows, int bit_width,
                          void* d_enc, ips_stream stream);

/* FleDecoder::Get x n (batch form), fle-encoding.h:404-567 via Unpack_w :569-7329.
 * out_width: bytes per decoded value: 1 (bit_width<=8), 2 (<=16) or 4; the reference's staging
 * widths (fle-encoding.h:365-371) are the minimum legal widths. */
ips_status ips_fle_decode(const void* d_enc, int64_t n_rows, int bit_width, void* d_out,
                          int out_width, ips_stream stream);

/* FleDecoder::Eq/Lt/Le/Gt/Ge/In on rows [0, n_rows), fle-encoding.h:7962-8313.
 * consts: n_consts (1, or 1..IPS_MAX_IN_LIST for IN) unsigned constants < 2^bit_width (host).
 * Evaluated directly on the encoded bit-planes; nothing is decoded. */
ips_status ips_fle_pred(const void* d_enc, int64_t n_rows, int bit_width, ips_op op,
                        const uint64_t* consts, int n_consts, uint64_t* d_bitmap,
                        ips_stream stream);

/* Predicate on an OPTIONAL (nullable) column: ColumnReader<T>::Eq..In's dictionary branch,
 * hdfs-parquet-scanner.cc:338-345 -- fle_def_levels_->Eq(n, bits, max_def_level), the data predicate
 * over the bits.count() non-NULL values, IntersectBitset (:326-331) -- without the host ever
 * learning the non-NULL count.  bit r of d_bitmap = row r is not NULL and its value passes.
 *   d_def_levels   FLE blocks of the n_rows definition levels, width def_bit_width
 *                  (= Log2(max_def_level + 1), writer :387-399); width 1 / max_def 1 is read as the
 *                  NOT-NULL bits directly
 *   d_data_enc     FLE blocks of the non-NULL rows' values only; n_data_rows = rows this buffer
 *                  holds (any upper bound of the non-NULL count that the buffer covers, e.g.
 *                  64 * blocks; rows beyond it count as not selected)
 *   d_workspace    ips_nullable_workspace_bytes(n_rows) bytes, 16-byte aligned
 * Launch-only (4 launches at most: [levels == max_def], tile ranks, data predicate, expand). */
size_t ips_nullable_workspace_bytes(int64_t n_rows);
ips_status ips_fle_pred_nullable(const void* d_def_levels, int def_bit_width, int max_def_level,
                                 int64_t n_rows, const void* d_data_enc, int64_t n_data_rows,
                                 int bit_width, ips_op op, const uint64_t* consts, int n_consts,
                                 uint64_t* d_bitmap, void* d_workspace, ips_stream stream);

/* Fused scan of one FLE column chunk: predicate -> bitmap, plus late materialisation of the
 * selected rows in the same pass over the encoded bytes (EvalSimplePredicates + ReadValue(skip),
 * hdfs-parquet-scanner.cc:1837-1865, 1134-1181, 1006-1027; FleDecoder::Get(val, skip)
 * fle-encoding.h:344-379).
 *   d_bitmap        ceil(n/64) words (may not be NULL)
 *   d_batch_values  ceil(n/IPS_BATCH_ROWS) batches of IPS_BATCH_ROWS slots of 4 bytes; batch b
 *                   holds its d_batch_counts[b] selected values first, in row order
 *   d_batch_counts  uint32 per batch */
ips_status ips_fle_scan(const void* d_enc, int64_t n_rows, int bit_width, ips_op op,
                        const uint64_t* consts, int n_consts, uint64_t* d_bitmap,
                        uint32_t* d_batch_values, uint32_t* d_batch_counts, ips_stream stream);

/* ips_fle_scan over MANY data pages (separate buffers) with one launch per 64 pages: the scanner
 * holds a column chunk as a list of pages (ReadDataPage / InitDataPage per page,
 * hdfs-parquet-scanner.cc:882-916; pages and row groups are independent, :1056-1060), and a
 * 2^20-row page is 0.8 us of HBM time -- far below a launch.  Every page has its own outputs,
 * laid out as ips_fle_scan lays them out; all pages share bit_width and the predicate (EQ..GE).
 * The page descriptors travel in the kernel argument: no device-side table, no copy, capturable. */
typedef struct {
  const void* d_enc;        /* FLE blocks of the page, 16-byte aligned */
  int64_t n_rows;
  uint64_t* d_bitmap;       /* ceil(n_rows/64) words */
  uint32_t* d_batch_values; /* ceil(n_rows/IPS_BATCH_ROWS) batches of IPS_BATCH_ROWS slots */
  uint32_t* d_batch_counts; /* ceil(n_rows/IPS_BATCH_ROWS) counts */
} ips_page_scan;
ips_status ips_fle_scan_pages(const ips_page_scan* h_pages, int n_pages, int bit_width, ips_op op,
                              const uint64_t* consts, int n_consts, ips_stream stream);

/* Late materialisation against an existing bitmap (e.g. the AND of several columns' predicates):
 * per batch, the values of the rows whose bitmap bit is set (ReadValue(skip) per selected row,
 * hdfs-parquet-scanner.cc:1151-1181).  Same output layout as ips_fle_scan. */
ips_status ips_fle_select(const void* d_enc, int64_t n_rows, int bit_width,
                          const uint64_t* d_bitmap, uint32_t* d_batch_values,
                          uint32_t* d_batch_counts, ips_stream stream);

/* Concatenate the batches of ips_fle_scan / ips_fle_select / ips_dict_scan into one dense array
 * (tuple order of AssembleRows, hdfs-parquet-scanner.cc:1151-1181).  value_width 4 or 8.
 * d_total receives the number of values (int64).  Workspace: ips_batches_workspace_bytes(). */
size_t ips_batches_workspace_bytes(int64_t n_rows);
ips_status ips_batches_compact(const void* d_batch_values, const uint32_t* d_batch_counts,
                               int64_t n_rows, int value_width, void* d_dense, int64_t* d_total,
                               void* d_workspace, ips_stream stream);

/* Multi-column late materialisation into row-major tuples (AssembleRows' vector path,
 * hdfs-parquet-scanner.cc:1151-1181: per selected row, every column's ReadValue writes its slot
 * at tuple + slot_desc->tuple_offset(); descriptors.h:60-71, 75-95).  All columns were
 * materialised against the SAME bitmap (ips_fle_select / ips_dict_scan / ips_fle_scan), so they
 * share d_batch_counts.  Tuple i (i-th selected row, row order) is written at
 * d_tuples + i*tuple_size and starts as a copy of h_template_tuple (tuple_size bytes on the host:
 * the scanner's template_tuple_ that InitTuple() copies first, hdfs-scanner.h; NULL = all zero), so
 * every byte of every tuple is defined.  d_total receives the tuple count.
 * Workspace: ips_assemble_workspace_bytes(n_rows, number of OPTIONAL columns). */
typedef struct {
  const void* d_batch_values; /* batches of IPS_BATCH_ROWS slots of value_width bytes */
  int32_t value_width;        /* 4 or 8 */
  int32_t tuple_offset;       /* byte offset of the slot inside the tuple */
  /* OPTIONAL columns (all zero / NULL for REQUIRED ones): d_dense_values holds the values of the
   * selected NON-NULL rows only, densely, in row order (ips_dict_select over the data rows chosen
   * by ips_bitmap_compress(nonnull, selection) + ips_batches_compact); d_nonnull_flags has one bit
   * per selected row (ips_bitmap_compress(selection, nonnull)).  A NULL row gets
   * tuple[null_byte_offset] |= null_bit_mask (SlotDescriptor::null_indicator_offset(),
   * descriptors.h:60-71) and its slot keeps the template's bytes. */
  const void* d_dense_values;
  const uint64_t* d_nonnull_flags; /* NULL with d_dense_values set: a REQUIRED column given as ONE dense array
                                      (tuple i takes value i) instead of batches -- what ips_chunk_select +
                                      ips_batches_compact produce over a page list, whose batches are cut at
                                      the column's own page ends and cannot share d_batch_counts */
  int32_t null_byte_offset;
  int32_t null_bit_mask;
} ips_tuple_column;
#define IPS_TUPLE_MAX_COLS 16
size_t ips_assemble_workspace_bytes(int64_t n_rows, int n_optional_cols);
ips_status ips_assemble_tuples(const ips_tuple_column* cols, int n_cols,
                               const uint32_t* d_batch_counts, int64_t n_rows, int tuple_size,
                               const void* h_template_tuple, void* d_tuples, int64_t* d_total,
                               void* d_workspace, ips_stream stream);

/* ---- sorted dictionary codec ---------------------------------------------------------------- */
/* DictDecoder<T>::DictDecoder(dict_buffer, dict_len, fixed_len_size), dict-encoding.h:449-459:
 * PLAIN-decodes the (ascending) dictionary page; keeps a host copy for literal translation and a
 * device copy for gathers.  Synchronous. */
ips_status ips_dict_open(const void* h_dict_page, int64_t dict_len, ips_type type,
                         ips_dict** dict);
ips_status ips_dict_close(ips_dict* dict);
/* DictDecoder::num_entries(), dict-encoding.h:214 */
int64_t ips_dict_num_entries(const ips_dict* dict);
/* DictEncoderBase::bit_width(), dict-encoding.h:76-80 with BitUtil::Log2, bit-util.h:128-140 */
int ips_dict_bit_width(int64_t num_entries);

/* Dictionary-encode a column chunk on the GPU: DictEncoder<T>::Put x n, WriteDict (entries sorted
 * ascending, dict-encoding.h:393-406) and WriteData (indices remapped to sorted codes and
 * bit-sliced with width ceil(log2 D), :408-423).  d_values: n_rows PLAIN slots (4 or 8 bytes,
 * ips_plain_stride).  h_dict_page receives the PLAIN dictionary page (*dict_len bytes),
 * *bit_width the code width (the byte WriteData puts in front of the blocks), d_codes_enc the FLE
 * blocks (ips_fle_encoded_bytes(n_rows, *bit_width) bytes; reserve width 16).  More than 40000
 * distinct values (dict-encoding.h:157) -> IPS_ERR_UNSUPPORTED, the caller falls back to PLAIN
 * like the reference's writer.  Synchronous (the <= 40000 entries are sorted on the host).
 * int8/int16 values are the low 1/2 bytes of their slot (what Decode reads) and are written to
 * the page sign-extended to int32; floats are keyed by bit pattern (+0.0 and -0.0 are two entries);
 * a NaN in a FLOAT/DOUBLE column -> IPS_ERR_UNSUPPORTED (operator< gives the dictionary no order,
 * dict-encoding.h:370-372: fall back to PLAIN).  All temporaries live in d_workspace
 * (ips_dict_encode_workspace_bytes(n_rows) bytes, 16-byte aligned): nothing is allocated. */
size_t ips_dict_encode_workspace_bytes(int64_t n_rows);
ips_status ips_dict_encode(const void* d_values, int64_t n_rows, ips_type type, void* h_dict_page,
                           int64_t dict_page_capacity, int64_t* dict_len, int* bit_width,
                           void* d_codes_enc, void* d_workspace, ips_stream stream);

/* Literal -> code translation of DictDecoder<T>::Eq/Lt/Le/Gt/Ge/In, dict-encoding.h:461-541.
 * literals: n_literals values of the dictionary's type (host), any number of them for IN.  codes
 * must hold n_literals. */
ips_status ips_dict_translate(const ips_dict* dict, ips_op op, const void* literals,
                              int n_literals, ips_xl_kind* kind, ips_op* fle_op,
                              uint64_t* codes, int* n_codes);

/* ---- IN lists of any length ------------------------------------------------------------------- */
/* FleDecoder::In / DictDecoder::In take a vector of any length (fle-encoding.h:8236-8313,
 * dict-encoding.h:523-541; a dictionary holds up to 40000 codes).  The lists of up to
 * IPS_MAX_IN_LIST constants that the calls above take travel in the kernel arguments; a longer one
 * (or one that is used again and again: InOperate keeps its literals for the whole scan,
 * simple-predicates.h:195-205) is made resident once as an ips_inset: the membership table of the
 * members below 2^16 (codes of up to 16 bits look themselves up: one LDS read per row whatever the
 * list's length) and the ascending member list (wider columns compare against it member by member;
 * a column of w bits ignores members >= 2^w).  ips_inset_open / ips_dict_inset_open are synchronous
 * (they upload); duplicates are dropped; an empty set selects nothing. */
ips_status ips_inset_open(const uint64_t* consts, int64_t n_consts, ips_inset** set);
/* the set of CODES of the literals that are dictionary entries: DictDecoder<T>::In's translation,
 * dict-encoding.h:523-541, done once */
ips_status ips_dict_inset_open(const ips_dict* dict, const void* literals, int64_t n_literals,
                               ips_inset** set);
ips_status ips_inset_close(ips_inset* set);
int64_t ips_inset_size(const ips_inset* set); /* distinct members */
/* ips_fle_pred / ips_fle_scan / ips_dict_scan with op IN over a set (a dictionary column's predicate
 * alone is ips_fle_pred_inset on its codes) */
ips_status ips_fle_pred_inset(const void* d_enc, int64_t n_rows, int bit_width, const ips_inset* set,
                              uint64_t* d_bitmap, ips_stream stream);
ips_status ips_fle_scan_inset(const void* d_enc, int64_t n_rows, int bit_width, const ips_inset* set,
                              uint64_t* d_bitmap, uint32_t* d_batch_values, uint32_t* d_batch_counts,
                              ips_stream stream);
ips_status ips_dict_scan_inset(const ips_dict* dict, const void* d_codes_enc, int64_t n_rows,
                               int bit_width, const ips_inset* set, uint64_t* d_bitmap,
                               void* d_batch_values, uint32_t* d_batch_counts, ips_stream stream);

/* DictDecoder<T>::Eq/Lt/Le/Gt/Ge/In on a REQUIRED column's data page, dict-encoding.h:461-541.
 * d_codes_enc: the FLE blocks of the page, i.e. the payload AFTER the 1-byte bit-width header
 * that DictDecoderBase::SetData strips (dict-encoding.h:185-192); bit_width is that header byte. */
ips_status ips_dict_pred(const ips_dict* dict, const void* d_codes_enc, int64_t n_rows,
                         int bit_width, ips_op op, const void* literals, int n_literals,
                         uint64_t* d_bitmap, ips_stream stream);

/* ips_dict_pred on an OPTIONAL column (see ips_fle_pred_nullable): literals are translated to
 * codes first; a translation that says "all rows" selects every NON-NULL row. */
ips_status ips_dict_pred_nullable(const ips_dict* dict, const void* d_def_levels,
                                  int def_bit_width, int max_def_level, int64_t n_rows,
                                  const void* d_codes_enc, int64_t n_data_rows, int bit_width,
                                  ips_op op, const void* literals, int n_literals,
                                  uint64_t* d_bitmap, void* d_workspace, ips_stream stream);

/* DictDecoder<T>::GetValue x n, dict-encoding.h:310-319: d_out[r] = dict[code_r] (sizeof(T)
 * bytes each; int8/int16 as 1/2 bytes).  *d_bad_index (int32, may be NULL) is set non-zero if a
 * code >= num_entries was met (the reference returns false, :316). */
ips_status ips_dict_decode(const ips_dict* dict, const void* d_codes_enc, int64_t n_rows,
                           int bit_width, void* d_out, int32_t* d_bad_index, ips_stream stream);

/* Fused dictionary scan: ips_dict_pred + gather of the selected rows' values in one pass.
 * Batch slots are sizeof(T) bytes (T = dictionary type, int8/int16 widened to 4). */
ips_status ips_dict_scan(const ips_dict* dict, const void* d_codes_enc, int64_t n_rows,
                         int bit_width, ips_op op, const void* literals, int n_literals,
                         uint64_t* d_bitmap, void* d_batch_values, uint32_t* d_batch_counts,
                         ips_stream stream);

/* Late materialisation of a dictionary column against an existing bitmap: per batch, dict[code]
 * of the rows whose bit is set (ReadValue(skip) -> DictDecoder::GetValue(value, skip),
 * hdfs-parquet-scanner.cc:515-531, dict-encoding.h:321-330).  Slots as in ips_dict_scan. */
ips_status ips_dict_select(const ips_dict* dict, const void* d_codes_enc, int64_t n_rows,
                           int bit_width, const uint64_t* d_bitmap, void* d_batch_values,
                           uint32_t* d_batch_counts, ips_stream stream);

/* Late materialisation of an OPTIONAL column against a selection bitmap, in one call: the
 * ReadDefinitionLevel walk + GetValue(skip) of the selected non-NULL rows that ReadValue(skip) does
 * per row (hdfs-parquet-scanner.cc:1006-1038, 927-979).  Produces exactly what an OPTIONAL
 * ips_tuple_column takes:
 *   d_dense_values    the values of the selected NON-NULL rows, dense, in row order (dictionary
 *                     entries of ips slot width when dict != NULL, else the raw 4-byte FLE values)
 *   d_nonnull_flags   one bit per SELECTED row, 1 = not NULL (ceil(n_rows/64) words reserved)
 *   d_counts          int64[3]: [0] selected rows, [1] selected non-NULL rows, [2] non-zero when a
 *                     selected code lies outside the dictionary (DictDecoder::GetValue returns false,
 *                     dict-encoding.h:316; the slot of such a row is left unwritten)
 * d_def_levels / n_data_rows as in ips_fle_pred_nullable; d_selection: ceil(n_rows/64) words.
 * Workspace: ips_select_nullable_workspace_bytes(n_rows, n_data_rows, value_width 4 | 8). */
size_t ips_select_nullable_workspace_bytes(int64_t n_rows, int64_t n_data_rows, int value_width);
ips_status ips_dict_select_nullable(const ips_dict* dict, const void* d_def_levels, int def_bit_width,
                                    int max_def_level, int64_t n_rows, const void* d_codes_enc,
                                    int64_t n_data_rows, int bit_width, const uint64_t* d_selection,
                                    void* d_dense_values, uint64_t* d_nonnull_flags, int64_t* d_counts,
                                    void* d_workspace, ips_stream stream);

/* ---- PLAIN fixed-width pages ---------------------------------------------------------------- */
/* ParquetPlainEncoder::ByteSize(ColumnType), parquet-common.h:92-117: 4 or 8 */
int ips_plain_stride(ips_type type);
/* ParquetPlainEncoder::Eq/Lt/Le/Gt/Ge<T>, parquet-common.h:197-250 (int8 :335-383, int16
 * :400-449).  IN: the reference body is empty (parquet-common.h:252-255); REFERENCE semantics
 * returns IPS_ERR_UNSUPPORTED for it, SQL semantics evaluates it. */
ips_status ips_plain_pred(const void* d_page, int64_t n_rows, ips_type type, ips_op op,
                          const void* literals, int n_literals, ips_semantics semantics,
                          uint64_t* d_bitmap, ips_stream stream);

/* ips_plain_pred (SQL semantics) on an OPTIONAL PLAIN page: d_page holds the page's n_data_rows stored
 * (non-NULL) values, d_def_levels the FLE levels of all n_rows rows; bit r = row r is not NULL and
 * its value passes.  The reference's PLAIN branch ignores the levels altogether (hdfs-parquet-
 * scanner.cc:346-348: an OPTIONAL column that fell back from dictionary to PLAIN pages, dict-encoding.h:
 * 157, is compared against other rows' values there), so there is no REFERENCE-semantics form of this.
 * Workspace: ips_nullable_workspace_bytes(n_rows). */
ips_status ips_plain_pred_nullable(const void* d_def_levels, int def_bit_width, int max_def_level,
                                   int64_t n_rows, const void* d_page, int64_t n_data_rows,
                                   ips_type type, ips_op op, const void* literals, int n_literals,
                                   uint64_t* d_bitmap, void* d_workspace, ips_stream stream);

/* Fused scan of a PLAIN page: bitmap of (x op literal) [AND (x op2 literal2) when literal2 != NULL:
 * a BETWEEN as And(Ge, Le), simple-predicates.h:145-153] plus the selected rows' slots, one pass over
 * the page -- EvalSimplePredicates + ReadValue(skip) on the same PLAIN column
 * (hdfs-parquet-scanner.cc:1837-1865, 1006-1027; parquet-common.h:186-250).  Outputs as
 * ips_plain_select / ips_fle_scan lay them out (slots of ips_plain_stride(type) bytes). */
ips_status ips_plain_scan(const void* d_page, int64_t n_rows, ips_type type, ips_op op,
                          const void* literals, int n_literals, ips_op op2, const void* literal2,
                          ips_semantics semantics, uint64_t* d_bitmap, void* d_batch_values,
                          uint32_t* d_batch_counts, ips_stream stream);

/* Late materialisation on a PLAIN page against an existing bitmap: ReadValue(skip) ->
 * ParquetPlainEncoder::Decode(buffer, size, &val, skip_rows) per selected row
 * (parquet-common.h:186-190, hdfs-parquet-scanner.cc:1006-1027).  The selected rows' slots
 * (ips_plain_stride(type) bytes each: int8/int16 stay 4-byte slots) are written per batch exactly
 * like ips_fle_select writes them: batch b holds d_batch_counts[b] slots at
 * d_batch_values + b * IPS_BATCH_ROWS * stride, row order kept. */
ips_status ips_plain_select(const void* d_page, int64_t n_rows, ips_type type,
                            const uint64_t* d_bitmap, void* d_batch_values,
                            uint32_t* d_batch_counts, ips_stream stream);

/* ---- bitmap algebra (SimplePredicate tree, simple-predicates.h:145-163) --------------------- */
ips_status ips_bitmap_and(uint64_t* d_a, const uint64_t* d_b, int64_t n_rows, ips_stream stream);
ips_status ips_bitmap_or(uint64_t* d_a, const uint64_t* d_b, int64_t n_rows, ips_stream stream);
/* dynamic_bitset::resize(n, value) as used by dict-encoding.h:466,476,478 */
ips_status ips_bitmap_fill(uint64_t* d_a, int64_t n_rows, int value, ips_stream stream);
/* dynamic_bitset::count(), hdfs-parquet-scanner.cc:344,1125; d_count is int64 */
ips_status ips_bitmap_count(const uint64_t* d_a, int64_t n_rows, int64_t* d_count,
                            ips_stream stream);
/* The selected rows of every IPS_BATCH_ROWS-row batch of a selection (uint32 per batch): what the
 * fused scans write next to their values, for selections that come from elsewhere (the skip-list
 * walk of hdfs-parquet-scanner.cc:1134-1148 counts them row by row). */
ips_status ips_bitmap_batch_counts(const uint64_t* d_a, int64_t n_rows, uint32_t* d_batch_counts,
                                   ips_stream stream);
/* ColumnReader::IntersectBitset, hdfs-parquet-scanner.cc:326-331: the j-th set bit of d_root
 * takes bit j of d_sub; cleared bits stay 0.  Workspace: ips_expand_workspace_bytes(n_rows). */
size_t ips_expand_workspace_bytes(int64_t n_rows);
ips_status ips_bitmap_expand(const uint64_t* d_root, const uint64_t* d_sub, int64_t n_rows,
                             uint64_t* d_out, void* d_workspace, ips_stream stream);

/* The inverse of ips_bitmap_expand: d_out bit j = d_src bit (position of the j-th set bit of
 * d_mask); popcount(mask) bits are produced, the rest of the last word is zero.  With
 * mask = nonnull, src = selection it yields the selection over a nullable column's DATA rows (the
 * rows ReadValue really decodes once ReadDefinitionLevel said non-NULL,
 * hdfs-parquet-scanner.cc:1009-1014); with mask = selection, src = nonnull it yields the
 * non-NULL flag of every selected row (the NULL indicator bit, :1022-1026).
 * d_out must hold ceil(n_rows/64) words; d_n_out (int64, may be NULL) receives popcount(mask).
 * Workspace: ips_expand_workspace_bytes(n_rows). */
ips_status ips_bitmap_compress(const uint64_t* d_mask, const uint64_t* d_src, int64_t n_rows,
                               uint64_t* d_out, int64_t* d_n_out, void* d_workspace,
                               ips_stream stream);

/* ---- fused predicate program (EvalSimplePredicates, hdfs-parquet-scanner.cc:1837-1865) ------ */
typedef enum { IPS_NODE_LEAF = 0, IPS_NODE_AND = 1, IPS_NODE_OR = 2 } ips_node_kind;
typedef enum { IPS_COL_FLE = 0, IPS_COL_PLAIN = 1 } ips_col_encoding;

typedef struct {
  int32_t encoding;         /* ips_col_encoding */
  int32_t bit_width;        /* FLE: 1..32 */
  int32_t type;             /* PLAIN: ips_type */
  int32_t max_def_level;    /* 0 = REQUIRED column; > 0 = OPTIONAL (FLE: the reader's nullable branch,
                               hdfs-parquet-scanner.cc:338-345; PLAIN: SQL meaning -- NULL rows fail, the stored
                               values are compared -- where the reference ignores the levels, :346-348) */
  const void* d_data;       /* FLE blocks or PLAIN page (OPTIONAL: of the non-NULL rows only) */
  const void* d_def_levels; /* OPTIONAL: FLE blocks of the n_rows definition levels */
  int32_t def_bit_width;    /* OPTIONAL: Log2(max_def_level + 1) */
  int32_t reserved;
  int64_t n_data_rows;      /* OPTIONAL: rows d_data holds (see ips_fle_pred_nullable) */
} ips_column;

/* Postfix program: leaves push a bitmap, AND/OR pop two and push one (AndOperate / OrOperate,
 * simple-predicates.h:145-163); a conjunct list is a chain of ANDs (:1857-1862). */
typedef struct {
  int32_t kind;          /* ips_node_kind */
  int32_t column;        /* leaf: index into cols */
  int32_t op;            /* leaf: ips_op (FLE: on codes/values; PLAIN: SQL semantics) */
  int32_t n_consts;      /* leaf: 1, or 1..16 for IN */
  uint64_t consts[16];   /* leaf: FLE constants, or PLAIN literal bit patterns */
  const ips_inset* inset; /* leaf, op IPS_OP_IN on an FLE column: the list as a set of any length
                            (ips_inset_open); n_consts / consts are then ignored.  NULL otherwise */
} ips_node;

/* How ips_eval_program evaluates a tree.  PER_OPERAND: one launch of a stand-alone predicate kernel
 * per operand (a leaf, or two leaves on one column such as BETWEEN, in one pass) writing / AND-ing /
 * OR-ing into a bitmap -- those kernels run at 70-80 % of the HBM roofline.  ONE_PASS: a left-deep
 * conjunct / disjunct chain of two to six operands on REQUIRED FLE columns (comparisons, pairs, IN
 * lists of up to 16 constants; at most 16 KiB of planes per 2048 rows) as ONE kernel that reads every
 * column once and writes the bitmap once -- over page lists too: chunks cut at the same rows page by page,
 * chunks cut at different rows segment by segment (ips_eval_program_chunks; the latter only under ONE_PASS:
 * it ties the per-operand launches); other trees fall back to PER_OPERAND.  AUTO (the default):
 * ONE_PASS for such chains unless one of them compares a 32-bit column (its stand-alone kernel skips
 * the low planes of decided sub-tiles), PER_OPERAND otherwise.  ONE_LAUNCH: the whole tree as one
 * stack-machine kernel (REQUIRED columns only; 2-3x slower, for callers that must have a single
 * launch).  Process-wide; every strategy produces identical bitmaps. */
typedef enum { IPS_PROGRAM_AUTO = 0, IPS_PROGRAM_PER_OPERAND = 1, IPS_PROGRAM_ONE_PASS = 2,
               IPS_PROGRAM_ONE_LAUNCH = 3 } ips_program_strategy;
ips_status ips_set_program_strategy(int strategy);

#define IPS_PROGRAM_MAX_NODES 32
#define IPS_PROGRAM_MAX_COLS 8
/* Trees that keep more than one bitmap alive (an OR of ANDs) and leaves on OPTIONAL columns (rank
 * tables, the data rows' bitmap) park their temporaries in d_workspace:
 * ips_program_workspace_bytes(...) bytes, 16-byte aligned; 0 for a conjunct chain over REQUIRED
 * columns, and then d_workspace may be NULL.  The call allocates nothing and only launches, so it
 * can be captured into a hipGraph together with its workspace. */
size_t ips_program_workspace_bytes(const ips_node* nodes, int n_nodes, const ips_column* cols,
                                   int n_cols, int64_t n_rows);
ips_status ips_eval_program(const ips_node* nodes, int n_nodes, const ips_column* cols, int n_cols,
                            int64_t n_rows, uint64_t* d_bitmap, void* d_workspace,
                            ips_stream stream);

/* ---- column chunks as lists of pages --------------------------------------------------------- */
/* A column reader holds its chunk as data pages (ReadDataPage / InitDataPage per page,
 * hdfs-parquet-scanner.cc:730-924) whose row counts are the writer's choice: pages of different
 * columns end at different rows and EvalSimplePredicates cuts every batch at the page ends of all
 * columns (:1837-1855).  An ips_chunk is that page list resident in HBM: the pages stay separate
 * device buffers (no concatenation, no copy), every page keeps its own FLE block geometry (blocks
 * of 64 rows start at the page's first row) and its own code width (the dictionary writer puts the
 * width of each data page in its first byte and it grows with the dictionary, dict-encoding.h:
 * 425-447).  Calls over a chunk cost one launch per run of equally wide pages, whatever the number
 * of pages, and produce bitmaps over the chunk's rows: bit r <-> row r of the chunk = row
 * r - row0(p) of its page p.  Rows need not be aligned to anything; a page may have any size >= 0.
 * ips_chunk_open is synchronous (it uploads the page table); the calls over chunks only launch. */
typedef struct {
  const void* d_data;        /* FLE blocks (after the width byte SetData strips) or PLAIN slots of the page,
                                16-byte aligned; OPTIONAL chunks: of the page's non-NULL rows only */
  int64_t n_rows;            /* rows of the page, NULLs included (DataPageHeader.num_values) */
  int32_t bit_width;         /* FLE: the page's width, 1..32 */
  int32_t reserved;
  const void* d_def_levels;  /* OPTIONAL chunks: FLE blocks (width 1) of the page's n_rows definition levels */
  int64_t n_data_rows;       /* OPTIONAL chunks: rows d_data holds (see ips_fle_pred_nullable) */
} ips_chunk_page;
typedef struct ips_chunk ips_chunk;
/* encoding: ips_col_encoding; type: PLAIN chunks; max_def_level: 0 REQUIRED, 1 OPTIONAL (FLE /
 * dictionary chunks of a flat schema -- what the reference's vectorised path handles, :338-348;
 * anything else is IPS_ERR_UNSUPPORTED).  The page buffers must outlive the handle's use. */
ips_status ips_chunk_open(const ips_chunk_page* h_pages, int n_pages, int encoding, ips_type type,
                          int max_def_level, ips_chunk** chunk);
ips_status ips_chunk_close(ips_chunk* chunk);
int64_t ips_chunk_num_rows(const ips_chunk* chunk);
/* batch slots of the chunk's fused scans: the sum over the pages of ceil(rows / IPS_BATCH_ROWS) */
int64_t ips_chunk_num_batches(const ips_chunk* chunk);
int ips_chunk_num_pages(const ips_chunk* chunk); /* the non-empty pages */

/* ips_eval_program over chunks: leaf.column indexes 'chunks', which all hold the same number of rows
 * (the column chunks of one row group).  FLE constants that do not fit a page's width are folded per
 * page (always false / always true on that page); PLAIN leaves have SQL semantics, as in
 * ips_eval_program.  d_bitmap: ceil(rows / 64) words.  Workspace: ips_chunk_program_workspace_bytes(). */
size_t ips_chunk_program_workspace_bytes(const ips_node* nodes, int n_nodes,
                                         const ips_chunk* const* chunks, int n_chunks);
ips_status ips_eval_program_chunks(const ips_node* nodes, int n_nodes, const ips_chunk* const* chunks,
                                   int n_chunks, uint64_t* d_bitmap, void* d_workspace,
                                   ips_stream stream);

/* The fused scans over a REQUIRED chunk: ips_fle_scan / ips_dict_scan / ips_plain_scan with the
 * page loop inside the launch.  d_bitmap is the chunk's; the batch list is the concatenation of the
 * pages' batches (a page's last batch is partial): ips_chunk_num_batches() batches of
 * IPS_BATCH_ROWS slots and as many counts, in row order, which ips_batches_compact turns into one
 * dense array (pass ips_chunk_num_batches() * IPS_BATCH_ROWS as its n_rows). */
ips_status ips_chunk_fle_scan(const ips_chunk* chunk, ips_op op, const uint64_t* consts, int n_consts,
                              uint64_t* d_bitmap, uint32_t* d_batch_values, uint32_t* d_batch_counts,
                              ips_stream stream);
ips_status ips_chunk_dict_scan(const ips_chunk* chunk, const ips_dict* dict, ips_op op,
                               const void* literals, int n_literals, uint64_t* d_bitmap,
                               void* d_batch_values, uint32_t* d_batch_counts, ips_stream stream);
ips_status ips_chunk_plain_scan(const ips_chunk* chunk, ips_op op, const void* literals, int n_literals,
                                ips_op op2, const void* literal2, ips_semantics semantics,
                                uint64_t* d_bitmap, void* d_batch_values, uint32_t* d_batch_counts,
                                ips_stream stream);
/* Late materialisation of a REQUIRED chunk against a selection over the chunk's rows: ips_fle_select /
 * ips_dict_select / ips_plain_select with the page loop inside (dict NULL: the FLE values / the PLAIN
 * slots of ips_plain_stride bytes).  The batches are cut at the column's own page ends:
 * ips_batches_compact turns them into the dense array that ips_assemble_tuples takes as a dense
 * REQUIRED column (ips_tuple_column.d_dense_values with d_nonnull_flags NULL), next to batch counts
 * of the selection itself (ips_bitmap_batch_counts). */
ips_status ips_chunk_select(const ips_chunk* chunk, const ips_dict* dict, const uint64_t* d_bitmap,
                            void* d_batch_values, uint32_t* d_batch_counts, ips_stream stream);
/* The same for an OPTIONAL chunk (FLE values or dictionary codes; flat schema, width-1 levels): ReadValue(skip)
 * over a whole selection across ReadDataPage boundaries (hdfs-parquet-scanner.cc:1006-1038 + 927-979:
 * ReadDefinitionLevel says which selected rows are NULL and which data row of ITS page a NOT-NULL one decodes) --
 * ips_dict_select_nullable with the page loop inside.  d_selection: a bitmap over the chunk's rows.  Outputs as
 * there: d_dense_values = the values of the selected NOT-NULL rows, densely, in row order (4-byte FLE values, or
 * dictionary entries of the dictionary's slot size); d_nonnull_flags = one bit per selected row
 * (ceil(rows / 64) words); d_counts[0] = selected rows, [1] = values, [2] != 0: a selected code lies outside
 * the dictionary.  Both feed ips_assemble_tuples as an OPTIONAL column.  Four launches whatever the number of
 * pages (one more per further run of pages of another code width); no synchronisation, nothing allocated. */
size_t ips_chunk_select_nullable_workspace_bytes(const ips_chunk* chunk);
ips_status ips_chunk_select_nullable(const ips_chunk* chunk, const ips_dict* dict, const uint64_t* d_selection,
                                     void* d_dense_values, uint64_t* d_nonnull_flags, int64_t* d_counts,
                                     void* d_workspace, ips_stream stream);

/* ---- multi-GPU exchange (one process per GPU) ------------------------------------------------ */
/* The path shards by row stripes (blocks of 64 rows are independent, hdfs-parquet-scanner.cc:
 * 1056-1060); the only exchange is an all-gather of the stripes' bitmap words: RCCL over xGMI
 * (ncclAllGather, ncclUint64).  librccl is loaded on first use.  Rank 0 creates the id, the
 * host's bootstrap distributes its IPS_COMM_ID_BYTES bytes, every rank calls ips_comm_init.
 * d_all_words receives nranks * n_words words, rank r's slice at r * n_words. */
#define IPS_COMM_ID_BYTES 128
typedef struct ips_comm ips_comm;
ips_status ips_comm_unique_id(void* id_bytes, int len);
ips_status ips_comm_init(const void* id_bytes, int nranks, int rank, ips_comm** comm);
ips_status ips_comm_destroy(ips_comm* comm);
ips_status ips_allgather_bitmap(ips_comm* comm, const uint64_t* d_local_words, int64_t n_words,
                                uint64_t* d_all_words, ips_stream stream);

/* One step of the sharded fused scan with the exchange overlapped INSIDE the step:
 * the rank's rows are n_chunks pieces of n_rows / n_chunks rows (a multiple of IPS_BATCH_ROWS)
 * in block-cyclic order -- piece i of rank r is piece i * nranks + r of the whole column -- so the
 * all-gather of chunk i fills words [i * nranks * w, (i + 1) * nranks * w) of d_all_bitmap
 * (w = piece words) in natural row order.  ONE launch on 'stream' scans all pieces (piece after
 * piece in dispatch order, each signalling its completion); piece i is gathered on the
 * communicator's own stream as soon as it is complete, while the later pieces are being scanned.
 * Outputs d_local_bitmap / d_batch_values / d_batch_counts as ips_fle_scan over the n_rows local
 * rows.  The call first makes 'stream' wait for the gathers of earlier calls (they may still read
 * d_local_bitmap), so the buffers can be reused step after step; ips_comm_join makes any stream
 * wait for all gathers issued so far (before d_all_bitmap is read). */
ips_status ips_fle_scan_allgather(ips_comm* comm, const void* d_enc, int64_t n_rows, int bit_width,
                                  ips_op op, const uint64_t* consts, int n_consts, int n_chunks,
                                  uint64_t* d_local_bitmap, uint32_t* d_batch_values,
                                  uint32_t* d_batch_counts, uint64_t* d_all_bitmap, ips_stream stream);
ips_status ips_comm_join(ips_comm* comm, ips_stream stream);
/* The sharded step for a predicate tree over several columns (BASELINE configs[4]: the three Q6
 * columns over the ranks): 'chunks' are the rank's column chunks, all cut alike into the exchange
 * pieces -- the same number of pages, every page the same whole number of 64-row words on every rank
 * -- in block-cyclic order as above.  ONE call issues the plan's launches over all pieces and the
 * all-gather of every piece; the launches of the plan's last operand signal piece after piece, so the
 * exchange of the early pieces overlaps the evaluation of the later ones.  d_local_bitmap: the rank's
 * rows; workspace as ips_eval_program_chunks. */
ips_status ips_eval_program_chunks_allgather(ips_comm* comm, const ips_node* nodes, int n_nodes,
                                             const ips_chunk* const* chunks, int n_chunks,
                                             uint64_t* d_local_bitmap, uint64_t* d_all_bitmap,
                                             void* d_workspace, ips_stream stream);
/* Synchronises with the communicator's stream; IPS_ERR_HIP if a piece of a sharded step was waited
 * for in vain (its waiter gives up after about two seconds rather than spin for ever). */
ips_status ips_comm_check(ips_comm* comm);

/* ---- synthetic data (bench / tests) --------------------------------------------------------- */
/* d_out[i] = splitmix64(seed + i) & mask, as uint32 (SURVEY 8d generator). */
ips_status ips_synth_splitmix_u32(uint64_t seed, int64_t n, uint32_t mask, uint32_t* d_out,
                                  ips_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* IPS_H */
