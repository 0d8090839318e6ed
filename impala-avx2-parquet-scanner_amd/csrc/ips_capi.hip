// ips_capi.hip -- the extern "C" boundary of libips_hip.so (declared in include/ips.h):
// argument validation, dispatch on bit width / type, dictionary handles.  No kernel lives here.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "ips_fle_kernels.h"
#include "ips_chunk_host.h"
#include "ips_host.h"

namespace ips {

// ---- error text -----------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

ips_status hip_fail(hipError_t e, const char* what) {
  set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
  (void)hipGetLastError();  // clear the sticky error so later calls report their own
  return IPS_ERR_HIP;
}

// ---- device facts / grid sizing -------------------------------------------------------------
static std::mutex g_mu;
static std::unordered_map<int, int> g_cus;                 // device -> CU count
static std::unordered_map<const void*, int> g_occupancy;   // kernel -> blocks per CU

int device_cus() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_cus.find(dev);
  if (it != g_cus.end()) return it->second;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    cus = 0;
  g_cus[dev] = cus;
  return cus;
}

// Blocks launched per resident block slot.  A grid of exactly the resident size finishes on its
// slowest CU; smaller work shares let the dispatcher even that out (round 1: 4x, +3..6 % on the
// w=32 scan and predicate kernels; end of round 2, with the narrow scans at 8 waves per SIMD: 8x is
// neutral at w=32 and 3-5 % faster at w <= 16; 12x and more cost the early-pruning predicate of that
// build its prefetch -- today's, with buffer-resource loads, is 5 % faster at 32x).  Round 3: the predicate-only kernels and the one-pass chain, whose waves spend a few
// hundred instructions per sub-tile, run best with shares of one or two sub-tiles per wave (kind 1 /
// 2 below; tools/ab/grid_mult_kinds.py, tools/kernel_tour.py with IPS_TOUR_MULTS).  Dev builds read
// IPS_GRID_MULT / _PRED / _CHAIN / _DECODE / _DICT_DECODE / _SCAN_WIDE on every call.
int grid_mult(int kind) {
  static const char* const names[] = {"IPS_GRID_MULT", "IPS_GRID_MULT_PRED", "IPS_GRID_MULT_CHAIN", "IPS_GRID_MULT_DECODE",
                                      "IPS_GRID_MULT_DICT_DECODE", "IPS_GRID_MULT_SCAN_WIDE"};
  static const int defaults[] = {8, IPS_GRID_MULT_PRED, IPS_GRID_MULT_CHAIN, IPS_GRID_MULT_DECODE, IPS_GRID_MULT_DICT_DECODE,
                                 IPS_GRID_MULT_SCAN_WIDE};
  if (kind < 0 || kind > kGridScanWide) kind = kGridScan;
  const char* e = dev_env(names[kind]);
  const int v = e ? atoi(e) : 0;
  return v > 0 ? v : defaults[kind];
}

int grid_for_tiles(const void* kernel, int64_t tiles, int kind) {
  int cus = device_cus();
  if (cus <= 0) {
    set_error("no HIP device");
    return 0;
  }
  int per_cu = 0;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_occupancy.find(kernel);
    if (it != g_occupancy.end()) per_cu = it->second;
  }
  if (per_cu == 0) {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kThreads, 0) != hipSuccess ||
        per_cu <= 0)
      per_cu = 2;
    std::lock_guard<std::mutex> lk(g_mu);
    g_occupancy[kernel] = per_cu;
  }
  int64_t want = (tiles + kWavesPerBlock - 1) / kWavesPerBlock;
  if (want < 1) want = 1;
  int64_t cap = (int64_t)cus * per_cu * grid_mult(kind);
  if (want <= cap) return (int)want;
  // Even shares: with 'cap' blocks and want = 1.33 * cap, a third of the blocks would make two
  // rounds and the rest one -- the launch lasts two rounds for 1.33 rounds of work (a 2^25-row
  // chunk of the headline column: 40 us instead of 30).  Give every block the same number of
  // rounds k = ceil(want / cap) instead.
  const int64_t k = (want + cap - 1) / cap;
  return (int)((want + k - 1) / k);
}

// ---- launchers living in the other translation units ----------------------------------------
#define IPS_DECL_PARTS(P)                                                                       \
  ips_status launch_fle_scan_part_##P(int, int, int, const uint64_t*, int64_t, const PredArgs&, \
                                      uint32_t*, const uint32_t*, void*, uint32_t*, const void*, \
                                      uint32_t, int32_t*, hipStream_t);                         \
  ips_status launch_fle_decode_part_##P(int, int, int, const uint64_t*, int64_t, void*,         \
                                        const void*, uint32_t, int32_t*, hipStream_t);          \
  ips_status launch_fle_encode_part_##P(int, int, const void*, int64_t, uint64_t*, hipStream_t); \
  ips_status launch_fle_pred_part_##P(int, const uint64_t*, int64_t, const PredArgs&, uint32_t*, \
                                      hipStream_t);                                             \
  ips_status launch_fle_scan_pages_part_##P(int, const PageBatch&, int, int64_t,                \
                                            const PredArgs&, hipStream_t);                      \
  ips_status launch_fle_leaf_part_##P(int, const uint64_t*, int64_t, const PredArgs&, uint64_t*, \
                                      bool*, hipStream_t);                                       \
  ips_status launch_fle_selnull_part_##P(int, int, const uint64_t*, int64_t, const SelNullArgs&,  \
                                         void*, const void*, uint32_t, int64_t*, hipStream_t);
IPS_DECL_PARTS(a) IPS_DECL_PARTS(b) IPS_DECL_PARTS(c) IPS_DECL_PARTS(d)

ips_status launch_bitmap_binop(int op, uint64_t* a, const uint64_t* b, int64_t n_words, hipStream_t s);
ips_status launch_bitmap_fill(uint64_t* a, int64_t n_rows, int value, hipStream_t s);
ips_status launch_bitmap_count(const uint64_t* a, int64_t n_rows, int64_t* count, hipStream_t s);
ips_status launch_bitmap_batch_counts(const uint64_t* bitmap, int64_t n_rows, uint32_t* counts, hipStream_t s);
ips_status launch_bitmap_expand(const uint64_t* root, const uint64_t* sub, int64_t n_rows,
                                uint64_t* out, void* workspace, hipStream_t s);
ips_status launch_batches_compact(const void* batch_values, const uint32_t* counts,
                                  int64_t n_batches, int value_width, void* dense, int64_t* total,
                                  void* workspace, hipStream_t s);
ips_status launch_assemble_tuples(const ips_tuple_column* cols, int n_cols, const uint32_t* counts,
                                  int64_t n_batches, int tuple_size, const void* h_template,
                                  void* tuples, int64_t* total, void* workspace, hipStream_t s);
ips_status launch_synth(uint64_t seed, int64_t n, uint32_t mask, uint32_t* out, hipStream_t s);
size_t scan_workspace_bytes(int64_t items);
size_t assemble_workspace_bytes(int64_t n_batches, int n_optional);
size_t batches_workspace_bytes(int64_t n_batches);
ips_status launch_bitmap_compress(const uint64_t* mask, const uint64_t* src, int64_t n_rows,
                                  uint64_t* out, int64_t* n_out, void* workspace, hipStream_t s);

ips_status launch_fle_scan(int w, int mode, int gather, const uint64_t* enc, int64_t n_rows,
                           const PredArgs& args, uint32_t* bitmap32, const uint32_t* given32,
                           void* batch_values, uint32_t* batch_counts, const void* dict,
                           uint32_t dict_entries, int32_t* bad_index, hipStream_t s) {
#define IPS_A w, mode, gather, enc, n_rows, args, bitmap32, given32, batch_values, batch_counts, \
              dict, dict_entries, bad_index, s
  if (w <= 8) return launch_fle_scan_part_a(IPS_A);
  if (w <= 16) return launch_fle_scan_part_b(IPS_A);
  if (w <= 24) return launch_fle_scan_part_c(IPS_A);
  return launch_fle_scan_part_d(IPS_A);
#undef IPS_A
}

ips_status launch_fle_decode(int w, int out_width, int gather, const uint64_t* enc, int64_t n_rows,
                             void* out, const void* dict, uint32_t dict_entries,
                             int32_t* bad_index, hipStream_t s) {
#define IPS_A w, out_width, gather, enc, n_rows, out, dict, dict_entries, bad_index, s
  if (w <= 8) return launch_fle_decode_part_a(IPS_A);
  if (w <= 16) return launch_fle_decode_part_b(IPS_A);
  if (w <= 24) return launch_fle_decode_part_c(IPS_A);
  return launch_fle_decode_part_d(IPS_A);
#undef IPS_A
}

ips_status launch_fle_pred(int w, const uint64_t* enc, int64_t n_rows, const PredArgs& args,
                           uint32_t* bitmap32, hipStream_t s) {
  if (w <= 8) return launch_fle_pred_part_a(w, enc, n_rows, args, bitmap32, s);
  if (w <= 16) return launch_fle_pred_part_b(w, enc, n_rows, args, bitmap32, s);
  if (w <= 24) return launch_fle_pred_part_c(w, enc, n_rows, args, bitmap32, s);
  return launch_fle_pred_part_d(w, enc, n_rows, args, bitmap32, s);
}

bool fused_leaf_enabled() {  // dev switch for A/B runs
  static const bool on = dev_env("IPS_NO_FUSED_LEAF") == nullptr;
  return on;
}

// The nullable leaf in one launch (fle_leaf_kernel); the tile counts of the root must be complete
// on the stream.  *taken = false: the shape keeps the three-launch route, nothing was launched.
ips_status launch_fle_leaf(int w, int root_kind, const uint64_t* root, int64_t n_rows,
                           const uint32_t* tile_counts, const uint64_t* enc, int64_t n_sub,
                           const PredArgs& pred, uint64_t* out, int combine, bool* taken,
                           hipStream_t s) {
  *taken = false;
  if (!fused_leaf_enabled() || n_sub <= 0 || n_rows <= 0) return IPS_OK;
  if (n_rows >= (1ll << 40)) {
    set_error("nullable leaf: %lld rows, the rank arithmetic holds 2^40", (long long)n_rows);
    return IPS_ERR_INVALID_ARG;
  }
  PredArgs args = pred;
  args.combine = combine;
  args.aux_blocks = 0;
  args.aux_kind = root_kind;
  args.aux_root = root;
  args.aux_rows = n_rows;
  args.aux_counts = const_cast<uint32_t*>(tile_counts);
  if (w <= 8) return launch_fle_leaf_part_a(w, enc, n_sub, args, out, taken, s);
  if (w <= 16) return launch_fle_leaf_part_b(w, enc, n_sub, args, out, taken, s);
  if (w <= 24) return launch_fle_leaf_part_c(w, enc, n_sub, args, out, taken, s);
  return launch_fle_leaf_part_d(w, enc, n_sub, args, out, taken, s);
}

ips_status launch_fle_selnull(int w, int gather, const uint64_t* enc, int64_t n_data, const SelNullArgs& a,
                              void* dense, const void* dict, uint32_t dict_entries, int64_t* n_values,
                              hipStream_t s) {
  if (w <= 8) return launch_fle_selnull_part_a(w, gather, enc, n_data, a, dense, dict, dict_entries, n_values, s);
  if (w <= 16) return launch_fle_selnull_part_b(w, gather, enc, n_data, a, dense, dict, dict_entries, n_values, s);
  if (w <= 24) return launch_fle_selnull_part_c(w, gather, enc, n_data, a, dense, dict, dict_entries, n_values, s);
  return launch_fle_selnull_part_d(w, gather, enc, n_data, a, dense, dict, dict_entries, n_values, s);
}

ips_status launch_fle_encode(int w, int in_width, const void* values, int64_t n_rows,
                             uint64_t* enc, hipStream_t s) {
  if (w <= 8) return launch_fle_encode_part_a(w, in_width, values, n_rows, enc, s);
  if (w <= 16) return launch_fle_encode_part_b(w, in_width, values, n_rows, enc, s);
  if (w <= 24) return launch_fle_encode_part_c(w, in_width, values, n_rows, enc, s);
  return launch_fle_encode_part_d(w, in_width, values, n_rows, enc, s);
}

// ---- shared argument handling ---------------------------------------------------------------
static inline hipStream_t S(ips_stream s) { return reinterpret_cast<hipStream_t>(s); }

static bool check_fle_common(const void* d_enc, int64_t n_rows, int bw, const char* fn) {
  if (n_rows < 0) { set_error("%s: n_rows < 0", fn); return false; }
  if (bw < 1 || bw > 32) { set_error("%s: bit_width %d not in 1..32", fn, bw); return false; }
  if (n_rows > 0 && d_enc == nullptr) { set_error("%s: NULL encoded buffer", fn); return false; }
  if (!aligned16(d_enc)) { set_error("%s: encoded buffer not 16-byte aligned", fn); return false; }
  return true;
}

// Outcome of comparing against constants that do not fit in bw bits (the
// reference is inconsistent there; the build defines it by the unsigned SQL meaning).
ips_status build_pred_args(int bw, ips_op op, const uint64_t* consts, int n_consts,
                                  PredArgs* args, ConstKind* kind, const char* fn) {
  if (op < IPS_OP_EQ || op > IPS_OP_IN) { set_error("%s: bad op %d", fn, (int)op); return IPS_ERR_INVALID_ARG; }
  if (consts == nullptr || n_consts < 1) { set_error("%s: no constants", fn); return IPS_ERR_INVALID_ARG; }
  if (op != IPS_OP_IN && n_consts != 1) { set_error("%s: op takes exactly one constant", fn); return IPS_ERR_INVALID_ARG; }
  if (n_consts > IPS_MAX_IN_LIST) { set_error("%s: IN list longer than %d", fn, IPS_MAX_IN_LIST); return IPS_ERR_INVALID_ARG; }
  const uint64_t limit = bw == 32 ? 0xFFFFFFFFull : ((1ull << bw) - 1ull);
  memset(args, 0, sizeof(*args));
  args->op = (int32_t)op;
  *kind = kEvaluate;
  if (op == IPS_OP_IN) {
    int n = 0;
    for (int i = 0; i < n_consts; ++i)
      if (consts[i] <= limit) args->consts[n++] = (uint32_t)consts[i];
    args->n_consts = n;
    if (n == 0) *kind = kAllFalse;
    return IPS_OK;
  }
  args->n_consts = 1;
  if (consts[0] > limit) {
    *kind = (op == IPS_OP_LT || op == IPS_OP_LE) ? kAllTrue : kAllFalse;
    return IPS_OK;
  }
  args->consts[0] = (uint32_t)consts[0];
  return IPS_OK;
}

static int64_t n_batches_of(int64_t n_rows) { return (n_rows + IPS_BATCH_ROWS - 1) / IPS_BATCH_ROWS; }

}  // namespace ips

using namespace ips;

// =============================================================================================
extern "C" {

int ips_version(void) { return IPS_VERSION; }
const char* ips_last_error(void) { return g_err; }

ips_status ips_device_count(int* count) {
  IPS_REQUIRE(count != nullptr, "ips_device_count: NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *count = 0; return hip_fail(e, "hipGetDeviceCount"); }
  *count = n;
  return IPS_OK;
}

ips_status ips_set_device(int device) {
  IPS_HIP_TRY(hipSetDevice(device));
  return IPS_OK;
}

ips_status ips_device_info(char* name, int name_len, int* compute_units, int64_t* hbm_bytes) {
  int dev = 0;
  IPS_HIP_TRY(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  IPS_HIP_TRY(hipGetDeviceProperties(&prop, dev));
  if (name && name_len > 0) snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName);
  if (compute_units) *compute_units = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
  return IPS_OK;
}

ips_status ips_malloc(void** d_ptr, size_t bytes) {
  IPS_REQUIRE(d_ptr != nullptr, "ips_malloc: NULL");
  hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 16);
  if (e == hipErrorOutOfMemory) { set_error("ips_malloc: out of device memory (%zu bytes)", bytes); (void)hipGetLastError(); return IPS_ERR_NOMEM; }
  if (e != hipSuccess) return hip_fail(e, "hipMalloc");
  return IPS_OK;
}
ips_status ips_free(void* d_ptr) { IPS_HIP_TRY(hipFree(d_ptr)); return IPS_OK; }
ips_status ips_memcpy_h2d(void* d, const void* h, size_t bytes, ips_stream s) {
  IPS_HIP_TRY(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, S(s)));
  return IPS_OK;
}
ips_status ips_memcpy_d2h(void* h, const void* d, size_t bytes, ips_stream s) {
  IPS_HIP_TRY(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, S(s)));
  return IPS_OK;
}
ips_status ips_memset(void* d, int value, size_t bytes, ips_stream s) {
  IPS_HIP_TRY(hipMemsetAsync(d, value, bytes, S(s)));
  return IPS_OK;
}
ips_status ips_stream_create(ips_stream* s) {
  IPS_REQUIRE(s != nullptr, "ips_stream_create: NULL");
  hipStream_t hs;
  IPS_HIP_TRY(hipStreamCreateWithFlags(&hs, hipStreamNonBlocking));
  *s = reinterpret_cast<ips_stream>(hs);
  return IPS_OK;
}
ips_status ips_stream_destroy(ips_stream s) { IPS_HIP_TRY(hipStreamDestroy(S(s))); return IPS_OK; }
ips_status ips_stream_synchronize(ips_stream s) { IPS_HIP_TRY(hipStreamSynchronize(S(s))); return IPS_OK; }

// ---- FLE ------------------------------------------------------------------------------------
int64_t ips_fle_encoded_bytes(int64_t n_rows, int bit_width) {
  if (n_rows < 0 || bit_width < 0) return -1;
  return ((n_rows + 63) / 64) * (int64_t)bit_width * 8;
}

ips_status ips_fle_encode(const void* d_values, int in_width, int64_t n_rows, int bit_width,
                          void* d_enc, ips_stream stream) {
  if (!check_fle_common(d_enc, n_rows, bit_width, "ips_fle_encode")) return IPS_ERR_INVALID_ARG;
  IPS_REQUIRE(in_width == 1 || in_width == 2 || in_width == 4, "ips_fle_encode: in_width %d", in_width);
  IPS_REQUIRE(n_rows == 0 || (d_values && aligned16(d_values)), "ips_fle_encode: values NULL or misaligned");
  if (n_rows == 0) return IPS_OK;
  return launch_fle_encode(bit_width, in_width, d_values, n_rows, reinterpret_cast<uint64_t*>(d_enc), S(stream));
}

ips_status ips_fle_decode(const void* d_enc, int64_t n_rows, int bit_width, void* d_out,
                          int out_width, ips_stream stream) {
  if (!check_fle_common(d_enc, n_rows, bit_width, "ips_fle_decode")) return IPS_ERR_INVALID_ARG;
  IPS_REQUIRE(out_width == 1 || out_width == 2 || out_width == 4, "ips_fle_decode: out_width %d", out_width);
  IPS_REQUIRE(n_rows == 0 || (d_out && aligned16(d_out)), "ips_fle_decode: output NULL or misaligned");
  if (n_rows == 0) return IPS_OK;
  return launch_fle_decode(bit_width, out_width, 0, reinterpret_cast<const uint64_t*>(d_enc), n_rows,
                           d_out, nullptr, 0, nullptr, S(stream));
}

ips_status ips_fle_pred(const void* d_enc, int64_t n_rows, int bit_width, ips_op op,
                        const uint64_t* consts, int n_consts, uint64_t* d_bitmap,
                        ips_stream stream) {
  if (!check_fle_common(d_enc, n_rows, bit_width, "ips_fle_pred")) return IPS_ERR_INVALID_ARG;
  PredArgs args;
  ConstKind kind;
  ips_status st = build_pred_args(bit_width, op, consts, n_consts, &args, &kind, "ips_fle_pred");
  if (st != IPS_OK) return st;
  IPS_REQUIRE(n_rows == 0 || (d_bitmap && aligned16(d_bitmap)), "ips_fle_pred: bitmap NULL or misaligned");
  if (n_rows == 0) return IPS_OK;
  if (kind != kEvaluate) return launch_bitmap_fill(d_bitmap, n_rows, kind == kAllTrue, S(stream));
  return launch_fle_pred(bit_width, reinterpret_cast<const uint64_t*>(d_enc), n_rows, args,
                         reinterpret_cast<uint32_t*>(d_bitmap), S(stream));
}

static ips_status scan_common(const void* d_enc, int64_t n_rows, int bw, const PredArgs& args,
                              ConstKind kind, uint64_t* d_bitmap, void* d_batch_values,
                              uint32_t* d_batch_counts, int gather, const void* d_dict,
                              uint32_t dict_entries, int32_t* d_bad, hipStream_t s) {
  const uint64_t* enc = reinterpret_cast<const uint64_t*>(d_enc);
  uint32_t* bm32 = reinterpret_cast<uint32_t*>(d_bitmap);
  if (kind == kAllFalse) {
    ips_status st = launch_bitmap_fill(d_bitmap, n_rows, 0, s);
    if (st != IPS_OK) return st;
    IPS_HIP_TRY(hipMemsetAsync(d_batch_counts, 0, (size_t)n_batches_of(n_rows) * 4, s));
    return IPS_OK;
  }
  if (kind == kAllTrue) {
    ips_status st = launch_bitmap_fill(d_bitmap, n_rows, 1, s);
    if (st != IPS_OK) return st;
    return launch_fle_scan(bw, kScanGivenBitmap, gather, enc, n_rows, args, nullptr, bm32,
                           d_batch_values, d_batch_counts, d_dict, dict_entries, d_bad, s);
  }
  int mode = args.op == IPS_OP_IN ? kScanInList : kScanPredicate;
  return launch_fle_scan(bw, mode, gather, enc, n_rows, args, bm32, nullptr, d_batch_values,
                         d_batch_counts, d_dict, dict_entries, d_bad, s);
}

size_t ips_nullable_workspace_bytes(int64_t n_rows) { return nullable_workspace_bytes(n_rows < 0 ? 0 : n_rows); }

static ips_status check_nullable(const void* d_def_levels, int def_bit_width, int max_def_level,
                                 int64_t n_rows, int64_t n_data_rows, const void* d_bitmap,
                                 const void* d_workspace, const char* fn) {
  IPS_REQUIRE(n_rows >= 0 && n_data_rows >= 0, "%s: negative row count", fn);
  IPS_REQUIRE(def_bit_width >= 1 && def_bit_width <= 32, "%s: definition-level width %d", fn, def_bit_width);
  IPS_REQUIRE(max_def_level >= 1 && (def_bit_width == 32 || (uint64_t)max_def_level < (1ull << def_bit_width)),
              "%s: max_def_level %d does not fit %d bits", fn, max_def_level, def_bit_width);
  IPS_REQUIRE(n_rows == 0 || (d_def_levels && aligned16(d_def_levels)), "%s: definition levels NULL or misaligned", fn);
  IPS_REQUIRE(n_rows == 0 || (d_bitmap && aligned16(d_bitmap)), "%s: bitmap NULL or misaligned", fn);
  IPS_REQUIRE(n_rows == 0 || (d_workspace && aligned16(d_workspace)),
              "%s: pass a workspace of ips_nullable_workspace_bytes() bytes", fn);
  return IPS_OK;
}

// data predicate over the data rows -> ws.sub, then expand into the NOT-NULL positions
static ips_status nullable_leaf(const void* d_def_levels, int def_bit_width, int max_def_level,
                                int64_t n_rows, const void* d_data_enc, int64_t n_data_rows,
                                int bit_width, const PredArgs& args, ConstKind kind,
                                uint64_t* d_bitmap, void* d_workspace, hipStream_t s) {
  if (kind == kAllFalse) return launch_bitmap_fill(d_bitmap, n_rows, 0, s);
  const NullableWs ws = nullable_workspace(d_workspace, n_rows);
  int root_kind = 0;
  const uint64_t* root = nullptr;
  ips_status st = nullable_prepare_root(d_def_levels, def_bit_width, max_def_level, n_rows, ws,
                                        &root_kind, &root, s, /*count_tiles=*/false);
  if (st != IPS_OK) return st;
  int64_t n_sub = n_data_rows < n_rows ? n_data_rows : n_rows;  // a page holds no more data rows than rows
  const uint64_t* enc = reinterpret_cast<const uint64_t*>(d_data_enc);
  if (kind == kAllTrue || n_sub <= 0) {
    st = launch_rank_tile_counts(root_kind, root, n_rows, ws.tile_counts, s);
    if (st == IPS_OK && kind == kAllTrue) {
      n_sub = n_rows;
      st = launch_bitmap_fill(ws.sub, n_rows, 1, s);
    }
  } else if (fused_leaf_enabled()) {
    // counts, then predicate + IntersectBitset in one kernel (fle_leaf_kernel); the shapes it
    // does not take (long IN lists) run predicate and expand as separate launches
    st = launch_rank_tile_counts(root_kind, root, n_rows, ws.tile_counts, s);
    if (st != IPS_OK) return st;
    bool taken = false;
    st = launch_fle_leaf(bit_width, root_kind, root, n_rows, ws.tile_counts, enc, n_sub, args, d_bitmap, 0, &taken, s);
    if (st != IPS_OK || taken) return st;
    st = launch_fle_pred(bit_width, enc, n_sub, args, reinterpret_cast<uint32_t*>(ws.sub), s);
  } else {
    PredArgs with_counts = args;  // the tile counts ride on the data predicate's launch
    attach_rank_counts(&with_counts, root_kind, root, n_rows, ws.tile_counts);
    st = launch_fle_pred(bit_width, enc, n_sub, with_counts, reinterpret_cast<uint32_t*>(ws.sub), s);
  }
  if (st != IPS_OK) return st;
  return launch_expand(root_kind, root, ws.sub, n_rows, n_sub, ws.tile_counts, d_bitmap, 0, s);
}

ips_status ips_fle_pred_nullable(const void* d_def_levels, int def_bit_width, int max_def_level,
                                 int64_t n_rows, const void* d_data_enc, int64_t n_data_rows,
                                 int bit_width, ips_op op, const uint64_t* consts, int n_consts,
                                 uint64_t* d_bitmap, void* d_workspace, ips_stream stream) {
  ips_status st = check_nullable(d_def_levels, def_bit_width, max_def_level, n_rows, n_data_rows,
                                 d_bitmap, d_workspace, "ips_fle_pred_nullable");
  if (st != IPS_OK) return st;
  if (!check_fle_common(d_data_enc, n_data_rows, bit_width, "ips_fle_pred_nullable")) return IPS_ERR_INVALID_ARG;
  PredArgs args;
  ConstKind kind;
  st = build_pred_args(bit_width, op, consts, n_consts, &args, &kind, "ips_fle_pred_nullable");
  if (st != IPS_OK) return st;
  if (n_rows == 0) return IPS_OK;
  return nullable_leaf(d_def_levels, def_bit_width, max_def_level, n_rows, d_data_enc, n_data_rows,
                       bit_width, args, kind, d_bitmap, d_workspace, S(stream));
}

ips_status ips_fle_scan(const void* d_enc, int64_t n_rows, int bit_width, ips_op op,
                        const uint64_t* consts, int n_consts, uint64_t* d_bitmap,
                        uint32_t* d_batch_values, uint32_t* d_batch_counts, ips_stream stream) {
  if (!check_fle_common(d_enc, n_rows, bit_width, "ips_fle_scan")) return IPS_ERR_INVALID_ARG;
  PredArgs args;
  ConstKind kind;
  ips_status st = build_pred_args(bit_width, op, consts, n_consts, &args, &kind, "ips_fle_scan");
  if (st != IPS_OK) return st;
  IPS_REQUIRE(n_rows == 0 || (d_bitmap && aligned16(d_bitmap) && d_batch_values &&
                              aligned16(d_batch_values) && d_batch_counts),
              "ips_fle_scan: output NULL or misaligned");
  if (n_rows == 0) return IPS_OK;
  return scan_common(d_enc, n_rows, bit_width, args, kind, d_bitmap, d_batch_values,
                     d_batch_counts, 0, nullptr, 0, nullptr, S(stream));
}

ips_status ips_fle_select(const void* d_enc, int64_t n_rows, int bit_width,
                          const uint64_t* d_bitmap, uint32_t* d_batch_values,
                          uint32_t* d_batch_counts, ips_stream stream) {
  if (!check_fle_common(d_enc, n_rows, bit_width, "ips_fle_select")) return IPS_ERR_INVALID_ARG;
  IPS_REQUIRE(n_rows == 0 || (d_bitmap && d_batch_values && aligned16(d_batch_values) && d_batch_counts),
              "ips_fle_select: NULL or misaligned argument");
  if (n_rows == 0) return IPS_OK;
  PredArgs args;
  memset(&args, 0, sizeof(args));
  return launch_fle_scan(bit_width, kScanGivenBitmap, 0, reinterpret_cast<const uint64_t*>(d_enc),
                         n_rows, args, nullptr, reinterpret_cast<const uint32_t*>(d_bitmap),
                         d_batch_values, d_batch_counts, nullptr, 0, nullptr, S(stream));
}

ips_status ips_fle_scan_pages(const ips_page_scan* h_pages, int n_pages, int bit_width, ips_op op,
                              const uint64_t* consts, int n_consts, ips_stream stream) {
  IPS_REQUIRE(n_pages >= 0, "ips_fle_scan_pages: n_pages < 0");
  IPS_REQUIRE(op != IPS_OP_IN, "ips_fle_scan_pages: IN lists go page by page through ips_fle_scan");
  PredArgs args;
  ConstKind kind;
  ips_status st = build_pred_args(bit_width, op, consts, n_consts, &args, &kind, "ips_fle_scan_pages");
  if (st != IPS_OK) return st;
  if (n_pages == 0) return IPS_OK;
  IPS_REQUIRE(h_pages != nullptr, "ips_fle_scan_pages: NULL page list");
  for (int i = 0; i < n_pages; ++i) {
    const ips_page_scan& pg = h_pages[i];
    if (!check_fle_common(pg.d_enc, pg.n_rows, bit_width, "ips_fle_scan_pages")) return IPS_ERR_INVALID_ARG;
    IPS_REQUIRE(pg.n_rows == 0 || (pg.d_bitmap && aligned16(pg.d_bitmap) && pg.d_batch_values &&
                                   aligned16(pg.d_batch_values) && pg.d_batch_counts),
                "ips_fle_scan_pages: page %d: NULL or misaligned output", i);
  }
  hipStream_t s = S(stream);
  if (kind != kEvaluate) {  // a constant outside the domain: the answer does not depend on the data
    for (int i = 0; i < n_pages; ++i) {
      const ips_page_scan& pg = h_pages[i];
      if (pg.n_rows == 0) continue;
      st = scan_common(pg.d_enc, pg.n_rows, bit_width, args, kind, pg.d_bitmap, pg.d_batch_values,
                       pg.d_batch_counts, 0, nullptr, 0, nullptr, s);
      if (st != IPS_OK) return st;
    }
    return IPS_OK;
  }
  for (int first = 0; first < n_pages; first += kPagesPerLaunch) {
    PageBatch batch;
    memset(&batch, 0, sizeof(batch));
    int n = 0;
    int64_t max_rows = 0;
    for (int i = first; i < n_pages && i < first + kPagesPerLaunch; ++i) {
      const ips_page_scan& pg = h_pages[i];
      if (pg.n_rows == 0) continue;
      batch.pages[n++] = PageScan{reinterpret_cast<const uint64_t*>(pg.d_enc), pg.n_rows,
                                  reinterpret_cast<uint32_t*>(pg.d_bitmap), pg.d_batch_values,
                                  pg.d_batch_counts};
      if (pg.n_rows > max_rows) max_rows = pg.n_rows;
    }
    if (n == 0) continue;
    if (bit_width <= 8) st = launch_fle_scan_pages_part_a(bit_width, batch, n, max_rows, args, s);
    else if (bit_width <= 16) st = launch_fle_scan_pages_part_b(bit_width, batch, n, max_rows, args, s);
    else if (bit_width <= 24) st = launch_fle_scan_pages_part_c(bit_width, batch, n, max_rows, args, s);
    else st = launch_fle_scan_pages_part_d(bit_width, batch, n, max_rows, args, s);
    if (st != IPS_OK) return st;
  }
  return IPS_OK;
}

size_t ips_batches_workspace_bytes(int64_t n_rows) { return batches_workspace_bytes(n_batches_of(n_rows)); }

ips_status ips_batches_compact(const void* d_batch_values, const uint32_t* d_batch_counts,
                               int64_t n_rows, int value_width, void* d_dense, int64_t* d_total,
                               void* d_workspace, ips_stream stream) {
  IPS_REQUIRE(n_rows >= 0, "ips_batches_compact: n_rows < 0");
  IPS_REQUIRE(value_width == 4 || value_width == 8, "ips_batches_compact: value_width %d", value_width);
  IPS_REQUIRE(d_total && d_workspace, "ips_batches_compact: NULL total/workspace");
  IPS_REQUIRE(n_rows == 0 || (d_batch_values && d_batch_counts && d_dense), "ips_batches_compact: NULL argument");
  return launch_batches_compact(d_batch_values, d_batch_counts, n_batches_of(n_rows), value_width,
                                d_dense, d_total, d_workspace, S(stream));
}

ips_status ips_assemble_tuples(const ips_tuple_column* cols, int n_cols,
                               const uint32_t* d_batch_counts, int64_t n_rows, int tuple_size,
                               const void* h_template_tuple, void* d_tuples, int64_t* d_total,
                               void* d_workspace, ips_stream stream) {
  IPS_REQUIRE(cols && n_cols >= 1 && n_cols <= IPS_TUPLE_MAX_COLS, "ips_assemble_tuples: n_cols %d not in 1..%d", n_cols, IPS_TUPLE_MAX_COLS);
  IPS_REQUIRE(n_rows >= 0 && tuple_size > 0, "ips_assemble_tuples: bad n_rows / tuple_size");
  IPS_REQUIRE(d_total && d_workspace, "ips_assemble_tuples: NULL total/workspace");
  IPS_REQUIRE(n_rows == 0 || (d_batch_counts && d_tuples), "ips_assemble_tuples: NULL argument");
  for (int i = 0; i < n_cols; ++i) {
    IPS_REQUIRE(cols[i].value_width == 4 || cols[i].value_width == 8, "ips_assemble_tuples: column %d: value_width", i);
    IPS_REQUIRE(cols[i].tuple_offset >= 0 && cols[i].tuple_offset + cols[i].value_width <= tuple_size,
                "ips_assemble_tuples: column %d: slot outside the tuple", i);
    if (cols[i].d_nonnull_flags) {
      IPS_REQUIRE(cols[i].d_dense_values, "ips_assemble_tuples: column %d: OPTIONAL column without dense values", i);
      IPS_REQUIRE(cols[i].null_byte_offset >= 0 && cols[i].null_byte_offset < tuple_size &&
                  cols[i].null_bit_mask > 0 && cols[i].null_bit_mask < 256,
                  "ips_assemble_tuples: column %d: bad NULL indicator", i);
    } else {
      IPS_REQUIRE(n_rows == 0 || cols[i].d_batch_values || cols[i].d_dense_values, "ips_assemble_tuples: column %d: NULL values", i);
    }
  }
  return launch_assemble_tuples(cols, n_cols, d_batch_counts, n_batches_of(n_rows), tuple_size,
                                h_template_tuple, d_tuples, d_total, d_workspace, S(stream));
}

size_t ips_assemble_workspace_bytes(int64_t n_rows, int n_optional_cols) {
  return assemble_workspace_bytes(n_batches_of(n_rows), n_optional_cols < 0 ? 0 : n_optional_cols);
}

// ---- dictionary -----------------------------------------------------------------------------
}  // extern "C"


namespace ips {
static int type_elem(ips_type t) {
  switch (t) {
    case IPS_T_INT8: return 1;
    case IPS_T_INT16: return 2;
    case IPS_T_INT32: case IPS_T_FLOAT: return 4;
    default: return 8;
  }
}
static bool valid_type(int t) { return t >= IPS_T_INT8 && t <= IPS_T_DOUBLE; }

template <typename T>
static void translate_t(const ips_dict* d, ips_op op, const void* literals, int n_literals,
                        ips_xl_kind* kind, ips_op* fle_op, uint64_t* codes, int* n_codes) {
  // DictDecoder<T>::Eq/Gt/Lt/Ge/Le/In, dict-encoding.h:461-541 (empty-dictionary guard added,
  // the reference dereferences dict_[0]/back() unconditionally: SURVEY quirk Q12)
  const T* first = reinterpret_cast<const T*>(d->host.data());
  const T* last = first + d->n;
  const T* lit = reinterpret_cast<const T*>(literals);
  *n_codes = 0;
  *fle_op = op;
  if (d->n == 0) { *kind = IPS_XL_ALL_FALSE; return; }
  const T v = lit[0];
  // a NaN literal compares false with everything (IEEE); it must not reach lower_bound either
  if (op != IPS_OP_IN && v != v) { *kind = IPS_XL_ALL_FALSE; return; }
  switch (op) {
    case IPS_OP_EQ: {
      const T* it = std::lower_bound(first, last, v);
      if (it == last || v < *it) { *kind = IPS_XL_ALL_FALSE; return; }
      codes[0] = (uint64_t)(it - first); *n_codes = 1; *fle_op = IPS_OP_EQ; *kind = IPS_XL_FLE;
      return;
    }
    case IPS_OP_GT:
      if (last[-1] <= v) { *kind = IPS_XL_ALL_FALSE; return; }
      if (first[0] > v) { *kind = IPS_XL_ALL_TRUE; return; }
      codes[0] = (uint64_t)(std::upper_bound(first, last, v) - first);
      *n_codes = 1; *fle_op = IPS_OP_GE; *kind = IPS_XL_FLE;
      return;
    case IPS_OP_LT:
      if (first[0] >= v) { *kind = IPS_XL_ALL_FALSE; return; }
      if (last[-1] < v) { *kind = IPS_XL_ALL_TRUE; return; }
      codes[0] = (uint64_t)(std::lower_bound(first, last, v) - first);
      *n_codes = 1; *fle_op = IPS_OP_LT; *kind = IPS_XL_FLE;
      return;
    case IPS_OP_GE:
      if (last[-1] < v) { *kind = IPS_XL_ALL_FALSE; return; }
      if (first[0] >= v) { *kind = IPS_XL_ALL_TRUE; return; }
      codes[0] = (uint64_t)(std::lower_bound(first, last, v) - first);
      *n_codes = 1; *fle_op = IPS_OP_GE; *kind = IPS_XL_FLE;
      return;
    case IPS_OP_LE:
      if (first[0] > v) { *kind = IPS_XL_ALL_FALSE; return; }
      if (last[-1] <= v) { *kind = IPS_XL_ALL_TRUE; return; }
      codes[0] = (uint64_t)(std::upper_bound(first, last, v) - first);
      *n_codes = 1; *fle_op = IPS_OP_LT; *kind = IPS_XL_FLE;
      return;
    default: {
      for (int i = 0; i < n_literals; ++i) {
        if (lit[i] != lit[i]) continue;  // NaN
        const T* it = std::lower_bound(first, last, lit[i]);
        if (it == last || lit[i] < *it) continue;
        codes[(*n_codes)++] = (uint64_t)(it - first);
      }
      *fle_op = IPS_OP_IN;
      *kind = *n_codes ? IPS_XL_FLE : IPS_XL_ALL_FALSE;
      return;
    }
  }
}

ips_status translate(const ips_dict* d, ips_op op, const void* literals, int n_literals,
                     ips_xl_kind* kind, ips_op* fle_op, uint64_t* codes, int* n_codes) {
  switch (d->type) {
    case IPS_T_INT8: translate_t<int8_t>(d, op, literals, n_literals, kind, fle_op, codes, n_codes); break;
    case IPS_T_INT16: translate_t<int16_t>(d, op, literals, n_literals, kind, fle_op, codes, n_codes); break;
    case IPS_T_INT32: translate_t<int32_t>(d, op, literals, n_literals, kind, fle_op, codes, n_codes); break;
    case IPS_T_INT64: translate_t<int64_t>(d, op, literals, n_literals, kind, fle_op, codes, n_codes); break;
    case IPS_T_FLOAT: translate_t<float>(d, op, literals, n_literals, kind, fle_op, codes, n_codes); break;
    default: translate_t<double>(d, op, literals, n_literals, kind, fle_op, codes, n_codes); break;
  }
  return IPS_OK;
}

ips_status check_dict_call(const ips_dict* dict, ips_op op, const void* literals,
                           int n_literals, const char* fn) {
  IPS_REQUIRE(dict != nullptr, "%s: NULL dictionary", fn);
  IPS_REQUIRE(op >= IPS_OP_EQ && op <= IPS_OP_IN, "%s: bad op %d", fn, (int)op);
  IPS_REQUIRE(literals != nullptr && n_literals >= 1, "%s: no literals", fn);
  IPS_REQUIRE(op == IPS_OP_IN || n_literals == 1, "%s: op takes exactly one literal", fn);
  IPS_REQUIRE(n_literals <= IPS_MAX_IN_LIST, "%s: IN list longer than %d", fn, IPS_MAX_IN_LIST);
  return IPS_OK;
}
}  // namespace ips

extern "C" {

ips_status ips_dict_open(const void* h_dict_page, int64_t dict_len, ips_type type,
                         ips_dict** dict) {
  IPS_REQUIRE(dict != nullptr, "ips_dict_open: NULL out pointer");
  IPS_REQUIRE(valid_type(type), "ips_dict_open: bad type %d", (int)type);
  IPS_REQUIRE(dict_len >= 0 && (dict_len == 0 || h_dict_page), "ips_dict_open: bad page");
  const int slot = ips_plain_stride(type);
  IPS_REQUIRE(dict_len % slot == 0, "ips_dict_open: dict_len %lld not a multiple of %d", (long long)dict_len, slot);
  ips_dict* d = new ips_dict();
  d->type = type;
  d->n = dict_len / slot;
  d->elem = type_elem(type);
  d->slot = slot;
  d->d_entries = nullptr;
  d->host.resize((size_t)d->n * d->elem);
  std::vector<uint8_t> dev((size_t)d->n * slot);
  const uint8_t* page = reinterpret_cast<const uint8_t*>(h_dict_page);
  for (int64_t i = 0; i < d->n; ++i) {
    // ParquetPlainEncoder::Decode, parquet-common.h:179-183 (int8 :319-322, int16 :385-388)
    memcpy(&d->host[(size_t)i * d->elem], page + i * slot, (size_t)d->elem);
    if (type == IPS_T_INT8) { int32_t x = *(int8_t*)&d->host[(size_t)i]; memcpy(&dev[(size_t)i * 4], &x, 4); }
    else if (type == IPS_T_INT16) { int16_t h; memcpy(&h, &d->host[(size_t)i * 2], 2); int32_t x = h; memcpy(&dev[(size_t)i * 4], &x, 4); }
    else memcpy(&dev[(size_t)i * slot], page + i * slot, (size_t)slot);
  }
  // the literal translation binary-searches with T's operator<: the page must be ascending, which
  // rules out NaN entries (no strict weak order)
  bool ordered = true;
  for (int64_t i = 0; i < d->n && ordered; ++i) {
    if (type == IPS_T_FLOAT) { float x; memcpy(&x, &d->host[(size_t)i * 4], 4); ordered = x == x; }
    else if (type == IPS_T_DOUBLE) { double x; memcpy(&x, &d->host[(size_t)i * 8], 8); ordered = x == x; }
  }
  if (!ordered) {
    delete d;
    set_error("ips_dict_open: NaN entry in a FLOAT/DOUBLE dictionary (order undefined, dict-encoding.h:370-372)");
    return IPS_ERR_INVALID_ARG;
  }
  hipError_t e = hipMalloc(&d->d_entries, dev.size() ? dev.size() : 16);
  if (e != hipSuccess) { delete d; return hip_fail(e, "hipMalloc(dictionary)"); }
  if (!dev.empty()) {
    e = hipMemcpy(d->d_entries, dev.data(), dev.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d->d_entries); delete d; return hip_fail(e, "hipMemcpy(dictionary)"); }
  }
  *dict = d;
  return IPS_OK;
}

ips_status ips_dict_close(ips_dict* dict) {
  if (!dict) return IPS_OK;
  if (dict->d_entries) (void)hipFree(dict->d_entries);
  delete dict;
  return IPS_OK;
}

int64_t ips_dict_num_entries(const ips_dict* dict) { return dict ? dict->n : -1; }

int ips_dict_bit_width(int64_t num_entries) {
  if (num_entries <= 0) return 0;
  if (num_entries == 1) return 1;
  uint64_t x = (uint64_t)num_entries - 1;  // BitUtil::Log2 = ceil(log2), bit-util.h:128-140
  int result = 1;
  while (x >>= 1) ++result;
  return result;
}

ips_status ips_dict_translate(const ips_dict* dict, ips_op op, const void* literals,
                              int n_literals, ips_xl_kind* kind, ips_op* fle_op, uint64_t* codes,
                              int* n_codes) {
  // (host-side only: any number of literals; the calls that put the codes into a kernel argument take
  // up to IPS_MAX_IN_LIST, longer lists go through ips_dict_inset_open)
  IPS_REQUIRE(dict != nullptr, "ips_dict_translate: NULL dictionary");
  IPS_REQUIRE(op >= IPS_OP_EQ && op <= IPS_OP_IN, "ips_dict_translate: bad op %d", (int)op);
  IPS_REQUIRE(literals != nullptr && n_literals >= 1, "ips_dict_translate: no literals");
  IPS_REQUIRE(op == IPS_OP_IN || n_literals == 1, "ips_dict_translate: op takes exactly one literal");
  IPS_REQUIRE(kind && fle_op && codes && n_codes, "ips_dict_translate: NULL out pointer");
  return translate(dict, op, literals, n_literals, kind, fle_op, codes, n_codes);
}

// ---- IN sets of any length --------------------------------------------------------------------
}  // extern "C"
namespace ips {
void inset_pred_args(const ips_inset* set, int bw, PredArgs* args, bool* always_false) {
  memset(args, 0, sizeof(*args));
  const uint64_t limit = bw >= 32 ? 0xFFFFFFFFull : ((1ull << bw) - 1ull);
  const int64_t fit = std::upper_bound(set->members.begin(), set->members.end(), (uint32_t)limit) - set->members.begin();
  *always_false = fit == 0;
  args->op = IPS_OP_IN;
  args->n_consts = 1 << 30;  // (the launchers pick the membership table for codes of up to 16 bits)
  args->in_table = set->d_table;
  args->in_list = set->d_list;
  args->in_list_n = (int32_t)fit;
}
static ips_status inset_create(std::vector<uint32_t>& members, ips_inset** out) {
  std::sort(members.begin(), members.end());
  members.erase(std::unique(members.begin(), members.end()), members.end());
  ips_inset* s = new ips_inset();
  s->members.swap(members);
  s->d_table = nullptr;
  s->d_list = nullptr;
  std::vector<uint32_t> table(2048, 0u);
  for (uint32_t c : s->members)
    if (c < 65536u) table[c >> 5] |= 1u << (c & 31u);
  const size_t list_bytes = (s->members.empty() ? 1 : s->members.size()) * sizeof(uint32_t);
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&s->d_table), 2048 * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->d_list), list_bytes);
  if (e == hipSuccess) e = hipMemcpy(s->d_table, table.data(), 2048 * sizeof(uint32_t), hipMemcpyHostToDevice);
  if (e == hipSuccess && !s->members.empty())
    e = hipMemcpy(s->d_list, s->members.data(), s->members.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    if (s->d_table) (void)hipFree(s->d_table);
    if (s->d_list) (void)hipFree(s->d_list);
    delete s;
    return hip_fail(e, "ips_inset_open");
  }
  *out = s;
  return IPS_OK;
}
}  // namespace ips
extern "C" {

ips_status ips_inset_open(const uint64_t* consts, int64_t n_consts, ips_inset** set) {
  IPS_REQUIRE(set != nullptr && n_consts >= 0 && (n_consts == 0 || consts), "ips_inset_open: bad argument");
  std::vector<uint32_t> members;
  members.reserve((size_t)n_consts);
  for (int64_t i = 0; i < n_consts; ++i)
    if (consts[i] <= 0xFFFFFFFFull) members.push_back((uint32_t)consts[i]);  // (an FLE value has at most 32 bits)
  return inset_create(members, set);
}

ips_status ips_dict_inset_open(const ips_dict* dict, const void* literals, int64_t n_literals, ips_inset** set) {
  IPS_REQUIRE(dict != nullptr && set != nullptr && n_literals >= 0 && (n_literals == 0 || literals),
              "ips_dict_inset_open: bad argument");
  // DictDecoder<T>::In's translation (dict-encoding.h:523-541): the codes of the literals that are entries
  std::vector<uint32_t> members;
  const int64_t kPiece = 4096;
  std::vector<uint64_t> codes((size_t)kPiece);
  const uint8_t* lit = reinterpret_cast<const uint8_t*>(literals);
  for (int64_t i = 0; i < n_literals; i += kPiece) {
    const int n = (int)std::min<int64_t>(kPiece, n_literals - i);
    ips_xl_kind kind; ips_op fle_op; int n_codes = 0;
    translate(dict, IPS_OP_IN, lit + (size_t)i * (size_t)dict->elem, n, &kind, &fle_op, codes.data(), &n_codes);
    for (int j = 0; j < n_codes; ++j) members.push_back((uint32_t)codes[(size_t)j]);
  }
  return inset_create(members, set);
}

ips_status ips_inset_close(ips_inset* set) {
  if (!set) return IPS_OK;
  if (set->d_table) (void)hipFree(set->d_table);
  if (set->d_list) (void)hipFree(set->d_list);
  delete set;
  return IPS_OK;
}

int64_t ips_inset_size(const ips_inset* set) { return set ? (int64_t)set->members.size() : -1; }

ips_status ips_fle_pred_inset(const void* d_enc, int64_t n_rows, int bit_width, const ips_inset* set,
                              uint64_t* d_bitmap, ips_stream stream) {
  if (!check_fle_common(d_enc, n_rows, bit_width, "ips_fle_pred_inset")) return IPS_ERR_INVALID_ARG;
  IPS_REQUIRE(set != nullptr, "ips_fle_pred_inset: NULL set");
  IPS_REQUIRE(n_rows == 0 || (d_bitmap && aligned16(d_bitmap)), "ips_fle_pred_inset: bitmap NULL or misaligned");
  if (n_rows == 0) return IPS_OK;
  PredArgs args;
  bool none = false;
  inset_pred_args(set, bit_width, &args, &none);
  if (none) return launch_bitmap_fill(d_bitmap, n_rows, 0, S(stream));
  return launch_fle_pred(bit_width, reinterpret_cast<const uint64_t*>(d_enc), n_rows, args,
                         reinterpret_cast<uint32_t*>(d_bitmap), S(stream));
}

ips_status ips_fle_scan_inset(const void* d_enc, int64_t n_rows, int bit_width, const ips_inset* set,
                              uint64_t* d_bitmap, uint32_t* d_batch_values, uint32_t* d_batch_counts,
                              ips_stream stream) {
  if (!check_fle_common(d_enc, n_rows, bit_width, "ips_fle_scan_inset")) return IPS_ERR_INVALID_ARG;
  IPS_REQUIRE(set != nullptr, "ips_fle_scan_inset: NULL set");
  IPS_REQUIRE(n_rows == 0 || (d_bitmap && aligned16(d_bitmap) && d_batch_values && aligned16(d_batch_values) && d_batch_counts),
              "ips_fle_scan_inset: output NULL or misaligned");
  if (n_rows == 0) return IPS_OK;
  PredArgs args;
  bool none = false;
  inset_pred_args(set, bit_width, &args, &none);
  return scan_common(d_enc, n_rows, bit_width, args, none ? kAllFalse : kEvaluate, d_bitmap, d_batch_values,
                     d_batch_counts, 0, nullptr, 0, nullptr, S(stream));
}

ips_status ips_dict_scan_inset(const ips_dict* dict, const void* d_codes_enc, int64_t n_rows, int bit_width,
                               const ips_inset* set, uint64_t* d_bitmap, void* d_batch_values,
                               uint32_t* d_batch_counts, ips_stream stream) {
  IPS_REQUIRE(dict != nullptr && set != nullptr, "ips_dict_scan_inset: NULL dictionary / set");
  if (!check_fle_common(d_codes_enc, n_rows, bit_width, "ips_dict_scan_inset")) return IPS_ERR_INVALID_ARG;
  IPS_REQUIRE(bit_width <= 16, "ips_dict_scan_inset: code width %d > 16", bit_width);
  IPS_REQUIRE(n_rows == 0 || (d_bitmap && aligned16(d_bitmap) && d_batch_values && aligned16(d_batch_values) && d_batch_counts),
              "ips_dict_scan_inset: output NULL or misaligned");
  if (n_rows == 0) return IPS_OK;
  PredArgs args;
  bool none = false;
  inset_pred_args(set, bit_width, &args, &none);
  return scan_common(d_codes_enc, n_rows, bit_width, args, none ? kAllFalse : kEvaluate, d_bitmap, d_batch_values,
                     d_batch_counts, dict->slot, dict->d_entries, (uint32_t)dict->n, nullptr, S(stream));
}

ips_status ips_dict_pred(const ips_dict* dict, const void* d_codes_enc, int64_t n_rows,
                         int bit_width, ips_op op, const void* literals, int n_literals,
                         uint64_t* d_bitmap, ips_stream stream) {
  ips_status st = check_dict_call(dict, op, literals, n_literals, "ips_dict_pred");
  if (st != IPS_OK) return st;
  IPS_REQUIRE(n_rows >= 0, "ips_dict_pred: n_rows < 0");
  IPS_REQUIRE(n_rows == 0 || (d_bitmap && aligned16(d_bitmap)), "ips_dict_pred: bitmap NULL or misaligned");
  if (n_rows == 0) return IPS_OK;
  ips_xl_kind kind; ips_op fle_op; int n_codes = 0;
  uint64_t codes[IPS_MAX_IN_LIST];
  translate(dict, op, literals, n_literals, &kind, &fle_op, codes, &n_codes);
  if (kind != IPS_XL_FLE) return launch_bitmap_fill(d_bitmap, n_rows, kind == IPS_XL_ALL_TRUE, S(stream));
  return ips_fle_pred(d_codes_enc, n_rows, bit_width, fle_op, codes, n_codes, d_bitmap, stream);
}

ips_status ips_dict_pred_nullable(const ips_dict* dict, const void* d_def_levels,
                                  int def_bit_width, int max_def_level, int64_t n_rows,
                                  const void* d_codes_enc, int64_t n_data_rows, int bit_width,
                                  ips_op op, const void* literals, int n_literals,
                                  uint64_t* d_bitmap, void* d_workspace, ips_stream stream) {
  ips_status st = check_dict_call(dict, op, literals, n_literals, "ips_dict_pred_nullable");
  if (st != IPS_OK) return st;
  st = check_nullable(d_def_levels, def_bit_width, max_def_level, n_rows, n_data_rows, d_bitmap,
                      d_workspace, "ips_dict_pred_nullable");
  if (st != IPS_OK) return st;
  if (!check_fle_common(d_codes_enc, n_data_rows, bit_width, "ips_dict_pred_nullable")) return IPS_ERR_INVALID_ARG;
  if (n_rows == 0) return IPS_OK;
  ips_xl_kind kind; ips_op fle_op; int n_codes = 0;
  uint64_t codes[IPS_MAX_IN_LIST];
  translate(dict, op, literals, n_literals, &kind, &fle_op, codes, &n_codes);
  PredArgs args;
  ConstKind ck = kEvaluate;
  if (kind == IPS_XL_FLE) {
    st = build_pred_args(bit_width, fle_op, codes, n_codes, &args, &ck, "ips_dict_pred_nullable");
    if (st != IPS_OK) return st;
  } else {
    memset(&args, 0, sizeof(args));
    ck = kind == IPS_XL_ALL_TRUE ? kAllTrue : kAllFalse;  // ALL_TRUE = every NON-NULL row (:338-345)
  }
  return nullable_leaf(d_def_levels, def_bit_width, max_def_level, n_rows, d_codes_enc, n_data_rows,
                       bit_width, args, ck, d_bitmap, d_workspace, S(stream));
}

ips_status ips_dict_decode(const ips_dict* dict, const void* d_codes_enc, int64_t n_rows,
                           int bit_width, void* d_out, int32_t* d_bad_index, ips_stream stream) {
  IPS_REQUIRE(dict != nullptr, "ips_dict_decode: NULL dictionary");
  if (!check_fle_common(d_codes_enc, n_rows, bit_width, "ips_dict_decode")) return IPS_ERR_INVALID_ARG;
  IPS_REQUIRE(bit_width <= 16, "ips_dict_decode: code width %d > 16 (dictionaries hold <= 40000 entries)", bit_width);
  IPS_REQUIRE(n_rows == 0 || (d_out && aligned16(d_out)), "ips_dict_decode: output NULL or misaligned");
  if (d_bad_index) IPS_HIP_TRY(hipMemsetAsync(d_bad_index, 0, 4, S(stream)));
  if (n_rows == 0) return IPS_OK;
  return launch_fle_decode(bit_width, 4, dict->slot, reinterpret_cast<const uint64_t*>(d_codes_enc),
                           n_rows, d_out, dict->d_entries, (uint32_t)dict->n, d_bad_index, S(stream));
}

ips_status ips_dict_scan(const ips_dict* dict, const void* d_codes_enc, int64_t n_rows,
                         int bit_width, ips_op op, const void* literals, int n_literals,
                         uint64_t* d_bitmap, void* d_batch_values, uint32_t* d_batch_counts,
                         ips_stream stream) {
  ips_status st = check_dict_call(dict, op, literals, n_literals, "ips_dict_scan");
  if (st != IPS_OK) return st;
  if (!check_fle_common(d_codes_enc, n_rows, bit_width, "ips_dict_scan")) return IPS_ERR_INVALID_ARG;
  IPS_REQUIRE(bit_width <= 16, "ips_dict_scan: code width %d > 16", bit_width);
  IPS_REQUIRE(n_rows == 0 || (d_bitmap && aligned16(d_bitmap) && d_batch_values &&
                              aligned16(d_batch_values) && d_batch_counts),
              "ips_dict_scan: output NULL or misaligned");
  if (n_rows == 0) return IPS_OK;
  ips_xl_kind kind; ips_op fle_op; int n_codes = 0;
  uint64_t codes[IPS_MAX_IN_LIST];
  translate(dict, op, literals, n_literals, &kind, &fle_op, codes, &n_codes);
  PredArgs args;
  ConstKind ck = kEvaluate;
  if (kind == IPS_XL_FLE) {
    st = build_pred_args(bit_width, fle_op, codes, n_codes, &args, &ck, "ips_dict_scan");
    if (st != IPS_OK) return st;
  } else {
    memset(&args, 0, sizeof(args));
    ck = kind == IPS_XL_ALL_TRUE ? kAllTrue : kAllFalse;
  }
  return scan_common(d_codes_enc, n_rows, bit_width, args, ck, d_bitmap, d_batch_values,
                     d_batch_counts, dict->slot, dict->d_entries, (uint32_t)dict->n, nullptr,
                     S(stream));
}

ips_status ips_dict_select(const ips_dict* dict, const void* d_codes_enc, int64_t n_rows,
                           int bit_width, const uint64_t* d_bitmap, void* d_batch_values,
                           uint32_t* d_batch_counts, ips_stream stream) {
  IPS_REQUIRE(dict != nullptr, "ips_dict_select: NULL dictionary");
  if (!check_fle_common(d_codes_enc, n_rows, bit_width, "ips_dict_select")) return IPS_ERR_INVALID_ARG;
  IPS_REQUIRE(bit_width <= 16, "ips_dict_select: code width %d > 16", bit_width);
  IPS_REQUIRE(n_rows == 0 || (d_bitmap && d_batch_values && aligned16(d_batch_values) && d_batch_counts),
              "ips_dict_select: NULL or misaligned argument");
  if (n_rows == 0) return IPS_OK;
  PredArgs args;
  memset(&args, 0, sizeof(args));
  return launch_fle_scan(bit_width, kScanGivenBitmap, dict->slot,
                         reinterpret_cast<const uint64_t*>(d_codes_enc), n_rows, args, nullptr,
                         reinterpret_cast<const uint32_t*>(d_bitmap), d_batch_values,
                         d_batch_counts, dict->d_entries, (uint32_t)dict->n, nullptr, S(stream));
}

// ---- OPTIONAL column: late materialisation in one call ---------------------------------------
namespace {
struct SelNullWs {
  uint32_t* rank;        // tile counts (used by one compress at a time; one-pass route: NOT-NULL rows)
  uint32_t* rank_s;      // one-pass route: tile counts of the selection
  uint32_t* rank_rs;     //                 and of the selected NOT-NULL rows
  uint64_t* nonnull;     // NOT-NULL bitmap when the levels are wider than a bit
  uint64_t* data_sel;    // the selection over the data rows
  uint8_t* batch_values; // per-batch values of the selected data rows
  uint32_t* batch_counts;
  uint8_t* compact_ws;
  size_t total;
};
size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
bool select_nullable_one_pass() {  // dev switch for A/B runs: IPS_SELECT_NULLABLE_STEPS=1 composes the call from ten launches
  static const bool on = dev_env("IPS_SELECT_NULLABLE_STEPS") == nullptr;
  return on;
}
SelNullWs sel_null_ws(void* base, int64_t n_rows, int64_t n_data_rows, int value_width) {
  const size_t bm = align256((size_t)((n_rows + 63) / 64) * 8);
  const int64_t nb = n_batches_of(n_data_rows > 0 ? n_data_rows : 1);
  uint8_t* p = reinterpret_cast<uint8_t*>(base);
  SelNullWs w;
  size_t off = 0;
  w.rank = reinterpret_cast<uint32_t*>(p + off); off += rank_workspace_bytes(n_rows);
  w.rank_s = reinterpret_cast<uint32_t*>(p + off); off += rank_workspace_bytes(n_rows);
  w.rank_rs = reinterpret_cast<uint32_t*>(p + off); off += rank_workspace_bytes(n_rows);
  w.nonnull = reinterpret_cast<uint64_t*>(p + off); off += bm;
  w.data_sel = reinterpret_cast<uint64_t*>(p + off); off += bm;
  if (!select_nullable_one_pass()) {  // per-batch buffers of the step-by-step route only
    w.batch_values = p + off; off += align256((size_t)nb * IPS_BATCH_ROWS * (size_t)value_width);
    w.batch_counts = reinterpret_cast<uint32_t*>(p + off); off += align256((size_t)nb * 4);
    w.compact_ws = p + off; off += align256(batches_workspace_bytes(nb));
  } else {
    w.batch_values = nullptr;
    w.batch_counts = nullptr;
    w.compact_ws = nullptr;
  }
  w.total = off;
  return w;
}
}  // namespace

size_t ips_select_nullable_workspace_bytes(int64_t n_rows, int64_t n_data_rows, int value_width) {
  if (n_rows < 0 || n_data_rows < 0 || (value_width != 4 && value_width != 8)) return 0;
  return sel_null_ws(nullptr, n_rows, n_data_rows, value_width).total;
}

ips_status ips_dict_select_nullable(const ips_dict* dict, const void* d_def_levels, int def_bit_width,
                                    int max_def_level, int64_t n_rows, const void* d_codes_enc,
                                    int64_t n_data_rows, int bit_width, const uint64_t* d_selection,
                                    void* d_dense_values, uint64_t* d_nonnull_flags, int64_t* d_counts,
                                    void* d_workspace, ips_stream stream) {
  ips_status st = check_nullable(d_def_levels, def_bit_width, max_def_level, n_rows, n_data_rows,
                                 d_nonnull_flags, d_workspace, "ips_dict_select_nullable");
  if (st != IPS_OK) return st;
  if (!check_fle_common(d_codes_enc, n_data_rows, bit_width, "ips_dict_select_nullable")) return IPS_ERR_INVALID_ARG;
  IPS_REQUIRE(!dict || bit_width <= 16, "ips_dict_select_nullable: code width %d > 16", bit_width);
  IPS_REQUIRE(d_counts != nullptr, "ips_dict_select_nullable: NULL counts");
  IPS_REQUIRE(n_rows == 0 || (d_selection && aligned16(d_selection) && d_dense_values && aligned16(d_dense_values)),
              "ips_dict_select_nullable: NULL or misaligned argument");
  hipStream_t s = S(stream);
  IPS_HIP_TRY(hipMemsetAsync(d_counts, 0, 24, s));  // (also the bad-index flag, set by the kernels)
  if (n_rows == 0) return IPS_OK;
  const int vw = dict ? dict->slot : 4;
  const SelNullWs w = sel_null_ws(d_workspace, n_rows, n_data_rows, vw);
  NullableWs nws;
  nws.tile_counts = w.rank;
  nws.sub = w.data_sel;
  nws.nonnull = w.nonnull;
  int root_kind = 0;
  const uint64_t* root = nullptr;
  const bool one_pass = select_nullable_one_pass();
  st = nullable_prepare_root(d_def_levels, def_bit_width, max_def_level, n_rows, nws, &root_kind, &root, s,
                             /*count_tiles=*/false);
  if (st != IPS_OK) return st;
  if (one_pass) {
    // counts of NOT-NULL / selected / selected NOT-NULL rows per rank tile (and the flag words
    // cleared), then fle_select_nullable_kernel: values, the NOT-NULL flag of every selected row
    // (the NULL indicator bit, hdfs-parquet-scanner.cc:1022-1026) and both counts
    st = launch_rank3_counts(root_kind, root, d_selection, n_rows, w.rank, w.rank_s, w.rank_rs, d_nonnull_flags, s);
    if (st != IPS_OK) return st;
    const int64_t n_data = n_data_rows < n_rows ? n_data_rows : n_rows;
    if (n_data > 0) {
      SelNullArgs a;
      memset(&a, 0, sizeof(a));  // (pages = NULL: one buffer per column)
      a.root = reinterpret_cast<const unsigned long long*>(root);
      a.sel = reinterpret_cast<const unsigned long long*>(d_selection);
      a.c_r = w.rank;
      a.c_s = w.rank_s;
      a.c_rs = w.rank_rs;
      a.flags = reinterpret_cast<unsigned long long*>(d_nonnull_flags);
      a.n_selected = d_counts;
      a.bad_index = d_counts + 2;
      a.n_rows = n_rows;
      a.root_kind = root_kind;
      st = launch_fle_selnull(bit_width, dict ? dict->slot : 0, reinterpret_cast<const uint64_t*>(d_codes_enc), n_data,
                              a, d_dense_values, dict ? dict->d_entries : nullptr, dict ? (uint32_t)dict->n : 0u,
                              d_counts + 1, s);
      return st;  // (the kernel also wrote the flags and the number of selected rows)
    }
    IPS_HIP_TRY(hipMemsetAsync(d_counts + 1, 0, 8, s));
    return launch_compress_counted(0, d_selection, root_kind, root, n_rows, d_nonnull_flags, d_counts, w.rank_s, s);
  }
  // 1. the selection over the DATA rows (the rows ReadValue decodes once ReadDefinitionLevel said
  //    non-NULL, hdfs-parquet-scanner.cc:1009-1014)
  st = launch_compress(root_kind, root, 0, d_selection, n_rows, w.data_sel, nullptr, w.rank, s);
  if (st != IPS_OK) return st;
  // 2. their values, per batch, then dense in row order
  const int64_t n_data = n_data_rows < n_rows ? n_data_rows : n_rows;
  if (n_data > 0) {
    PredArgs args;
    memset(&args, 0, sizeof(args));
    st = launch_fle_scan(bit_width, kScanGivenBitmap, dict ? dict->slot : 0,
                         reinterpret_cast<const uint64_t*>(d_codes_enc), n_data, args, nullptr,
                         reinterpret_cast<const uint32_t*>(w.data_sel), w.batch_values, w.batch_counts,
                         dict ? dict->d_entries : nullptr, dict ? (uint32_t)dict->n : 0u,
                         reinterpret_cast<int32_t*>(d_counts + 2), s);
    if (st != IPS_OK) return st;
    st = launch_batches_compact(w.batch_values, w.batch_counts, n_batches_of(n_data), vw, d_dense_values,
                                d_counts + 1, w.compact_ws, s);
    if (st != IPS_OK) return st;
  } else {
    IPS_HIP_TRY(hipMemsetAsync(d_counts + 1, 0, 8, s));
  }
  // 3. the NOT-NULL flag of every selected row (the NULL indicator bit, :1022-1026)
  return launch_compress(0, d_selection, root_kind, root, n_rows, d_nonnull_flags, d_counts, w.rank, s);
}

// The same over an OPTIONAL column chunk held as a page list: four launches whatever the number of pages
// (+ one more per further run of pages of another code width).
namespace {
struct ChunkSelNullWs {
  uint64_t* sel_copy;
  uint32_t *c_r, *c_s, *c_rs;
  uint64_t *page_s, *page_rs;
  size_t total;
};
ChunkSelNullWs chunk_sel_null_ws(void* base, const ips_chunk* c) {
  uint8_t* p = reinterpret_cast<uint8_t*>(base);
  ChunkSelNullWs w;
  size_t off = 0;
  w.sel_copy = reinterpret_cast<uint64_t*>(p + off); off += align256((size_t)c->n_batches * (IPS_BATCH_ROWS / 64) * 8 + 16);
  w.c_r = reinterpret_cast<uint32_t*>(p + off); off += align256((size_t)c->rank_entries * 4);
  w.c_s = reinterpret_cast<uint32_t*>(p + off); off += align256((size_t)c->rank_entries * 4);
  w.c_rs = reinterpret_cast<uint32_t*>(p + off); off += align256((size_t)c->rank_entries * 4);
  w.page_s = reinterpret_cast<uint64_t*>(p + off); off += align256(c->pages.size() * 8);
  w.page_rs = reinterpret_cast<uint64_t*>(p + off); off += align256(c->pages.size() * 8);
  w.total = off;
  return w;
}
}  // namespace

size_t ips_chunk_select_nullable_workspace_bytes(const ips_chunk* chunk) {
  if (!chunk || chunk->max_def_level <= 0) return 0;
  return chunk_sel_null_ws(nullptr, chunk).total;
}

ips_status ips_chunk_select_nullable(const ips_chunk* chunk, const ips_dict* dict, const uint64_t* d_selection,
                                     void* d_dense_values, uint64_t* d_nonnull_flags, int64_t* d_counts,
                                     void* d_workspace, ips_stream stream) {
  IPS_REQUIRE(chunk != nullptr, "ips_chunk_select_nullable: NULL chunk");
  IPS_REQUIRE(chunk->encoding == IPS_COL_FLE && chunk->max_def_level > 0,
              "ips_chunk_select_nullable: an OPTIONAL FLE / dictionary chunk (REQUIRED chunks: ips_chunk_select)");
  IPS_REQUIRE(d_counts != nullptr, "ips_chunk_select_nullable: NULL counts");
  hipStream_t s = S(stream);
  IPS_HIP_TRY(hipMemsetAsync(d_counts, 0, 24, s));  // (also the bad-index flag, set by the kernels)
  if (chunk->n_rows == 0 || chunk->pages.empty()) return IPS_OK;
  IPS_REQUIRE(d_selection && aligned16(d_selection) && d_dense_values && aligned16(d_dense_values) && d_nonnull_flags &&
                  aligned16(d_nonnull_flags) && d_workspace && aligned16(d_workspace),
              "ips_chunk_select_nullable: NULL or misaligned argument");
  for (const ips_chunk::Run& run : chunk->runs)
    IPS_REQUIRE(!dict || run.bit_width <= 16, "ips_chunk_select_nullable: code width %d > 16", run.bit_width);
  const ChunkSelNullWs w = chunk_sel_null_ws(d_workspace, chunk);
  // the flag words (one bit per selected row, at most one per row): the select kernel ORs its segment ends in
  IPS_HIP_TRY(hipMemsetAsync(d_nonnull_flags, 0, (size_t)((chunk->n_rows + 63) / 64) * 8, s));
  int64_t max_rows = 0;
  for (const ips_chunk::Run& run : chunk->runs) max_rows = run.max_rows > max_rows ? run.max_rows : max_rows;
  ips_status st = launch_selnull_pages_prepare(chunk->d_pages, (int)chunk->pages.size(), max_rows, d_selection, chunk->n_rows,
                                               w.sel_copy, w.c_r, w.c_s, w.c_rs, w.page_s, w.page_rs, d_counts, s);
  if (st != IPS_OK) return st;
  for (const ips_chunk::Run& run : chunk->runs) {
    if (run.max_data <= 0) continue;  // (pages of NULLs only: their flags are zero already)
    SelNullArgs a;
    memset(&a, 0, sizeof(a));
    a.c_r = w.c_r;
    a.c_s = w.c_s;
    a.c_rs = w.c_rs;
    a.flags = reinterpret_cast<unsigned long long*>(d_nonnull_flags);
    a.bad_index = d_counts + 2;
    a.n_rows = run.max_rows;
    a.root_kind = kRootLevels1;
    a.pages = chunk->d_pages + run.first;
    a.n_pages = run.count;
    a.sel_copy = reinterpret_cast<const unsigned long long*>(w.sel_copy);
    a.page_s = reinterpret_cast<const unsigned long long*>(w.page_s) + run.first;
    a.page_rs = reinterpret_cast<const unsigned long long*>(w.page_rs) + run.first;
    st = launch_fle_selnull(run.bit_width, dict ? dict->slot : 0, nullptr, 0, a, d_dense_values,
                            dict ? dict->d_entries : nullptr, dict ? (uint32_t)dict->n : 0u, nullptr, s);
    if (st != IPS_OK) return st;
  }
  return IPS_OK;
}

// ---- PLAIN ----------------------------------------------------------------------------------
int ips_plain_stride(ips_type type) {
  switch (type) {
    case IPS_T_INT8: case IPS_T_INT16: case IPS_T_INT32: case IPS_T_FLOAT: return 4;
    case IPS_T_INT64: case IPS_T_DOUBLE: return 8;
  }
  return -1;
}

ips_status ips_plain_pred(const void* d_page, int64_t n_rows, ips_type type, ips_op op,
                          const void* literals, int n_literals, ips_semantics semantics,
                          uint64_t* d_bitmap, ips_stream stream) {
  IPS_REQUIRE(valid_type(type), "ips_plain_pred: bad type %d", (int)type);
  IPS_REQUIRE(op >= IPS_OP_EQ && op <= IPS_OP_IN, "ips_plain_pred: bad op %d", (int)op);
  IPS_REQUIRE(n_rows >= 0, "ips_plain_pred: n_rows < 0");
  IPS_REQUIRE(literals && n_literals >= 1, "ips_plain_pred: no literals");
  IPS_REQUIRE(op == IPS_OP_IN || n_literals == 1, "ips_plain_pred: op takes exactly one literal");
  IPS_REQUIRE(n_literals <= 16, "ips_plain_pred: IN list longer than 16");
  IPS_REQUIRE(semantics == IPS_SEM_REFERENCE || semantics == IPS_SEM_SQL, "ips_plain_pred: bad semantics");
  if (op == IPS_OP_IN && semantics == IPS_SEM_REFERENCE) {
    set_error("ips_plain_pred: IN has no reference behaviour on PLAIN pages (empty body, parquet-common.h:252-255)");
    return IPS_ERR_UNSUPPORTED;
  }
  IPS_REQUIRE(n_rows == 0 || (d_page && aligned16(d_page) && d_bitmap), "ips_plain_pred: NULL or misaligned argument");
  if (n_rows == 0) return IPS_OK;
  int eff = op;
  if (semantics == IPS_SEM_REFERENCE) {  // literal OP x  ==  x OP' literal
    if (op == IPS_OP_LT) eff = IPS_OP_GT; else if (op == IPS_OP_GT) eff = IPS_OP_LT;
    else if (op == IPS_OP_LE) eff = IPS_OP_GE; else if (op == IPS_OP_GE) eff = IPS_OP_LE;
  }
  return launch_plain_pred(type, d_page, n_rows, eff, literals, n_literals, d_bitmap, S(stream));
}

// A predicate on an OPTIONAL PLAIN page under SQL semantics.  The reference's PLAIN branch never looks
// at the definition levels (hdfs-parquet-scanner.cc:346-348, quirk Q3): a column that overflowed the
// 40000-entry dictionary (dict-encoding.h:157) and is OPTIONAL is compared against other rows' values
// there.  Here: the comparison over the page's stored (non-NULL) values, then IntersectBitset into the
// NOT-NULL positions (:326-331) -- three launches (tile counts, predicate, expand).
ips_status ips_plain_pred_nullable(const void* d_def_levels, int def_bit_width, int max_def_level, int64_t n_rows,
                                   const void* d_page, int64_t n_data_rows, ips_type type, ips_op op,
                                   const void* literals, int n_literals, uint64_t* d_bitmap, void* d_workspace,
                                   ips_stream stream) {
  ips_status st = check_nullable(d_def_levels, def_bit_width, max_def_level, n_rows, n_data_rows, d_bitmap,
                                 d_workspace, "ips_plain_pred_nullable");
  if (st != IPS_OK) return st;
  IPS_REQUIRE(valid_type(type), "ips_plain_pred_nullable: bad type %d", (int)type);
  IPS_REQUIRE(op >= IPS_OP_EQ && op <= IPS_OP_IN, "ips_plain_pred_nullable: bad op %d", (int)op);
  IPS_REQUIRE(literals && n_literals >= 1 && n_literals <= 16 && (op == IPS_OP_IN || n_literals == 1),
              "ips_plain_pred_nullable: 1..16 literals, one unless IN");
  IPS_REQUIRE(n_data_rows == 0 || (d_page && aligned16(d_page)), "ips_plain_pred_nullable: page NULL or misaligned");
  if (n_rows == 0) return IPS_OK;
  hipStream_t s = S(stream);
  const NullableWs ws = nullable_workspace(d_workspace, n_rows);
  int root_kind = 0;
  const uint64_t* root = nullptr;
  st = nullable_prepare_root(d_def_levels, def_bit_width, max_def_level, n_rows, ws, &root_kind, &root, s, /*count_tiles=*/true);
  if (st != IPS_OK) return st;
  const int64_t n_sub = n_data_rows < n_rows ? n_data_rows : n_rows;
  if (n_sub > 0) {
    st = launch_plain_pred(type, d_page, n_sub, op, literals, n_literals, ws.sub, s);
    if (st != IPS_OK) return st;
  }
  return launch_expand(root_kind, root, ws.sub, n_rows, n_sub, ws.tile_counts, d_bitmap, 0, s);
}

ips_status ips_plain_scan(const void* d_page, int64_t n_rows, ips_type type, ips_op op,
                          const void* literals, int n_literals, ips_op op2, const void* literal2,
                          ips_semantics semantics, uint64_t* d_bitmap, void* d_batch_values,
                          uint32_t* d_batch_counts, ips_stream stream) {
  IPS_REQUIRE(valid_type(type), "ips_plain_scan: bad type %d", (int)type);
  IPS_REQUIRE(op >= IPS_OP_EQ && op <= IPS_OP_IN, "ips_plain_scan: bad op %d", (int)op);
  IPS_REQUIRE(n_rows >= 0, "ips_plain_scan: n_rows < 0");
  IPS_REQUIRE(literals && n_literals >= 1 && n_literals <= 16, "ips_plain_scan: 1..16 literals");
  IPS_REQUIRE(op == IPS_OP_IN || n_literals == 1, "ips_plain_scan: op takes exactly one literal");
  IPS_REQUIRE(!literal2 || (op != IPS_OP_IN && op2 >= IPS_OP_EQ && op2 <= IPS_OP_GE),
              "ips_plain_scan: the second comparison must be EQ..GE on a non-IN first one");
  IPS_REQUIRE(semantics == IPS_SEM_REFERENCE || semantics == IPS_SEM_SQL, "ips_plain_scan: bad semantics");
  if (op == IPS_OP_IN && semantics == IPS_SEM_REFERENCE) {
    set_error("ips_plain_scan: IN has no reference behaviour on PLAIN pages (empty body, parquet-common.h:252-255)");
    return IPS_ERR_UNSUPPORTED;
  }
  IPS_REQUIRE(n_rows == 0 || (d_page && aligned16(d_page) && d_bitmap && aligned16(d_bitmap) &&
                              d_batch_values && aligned16(d_batch_values) && d_batch_counts),
              "ips_plain_scan: NULL or misaligned argument");
  if (n_rows == 0) return IPS_OK;
  auto flip = [&](int o) {  // REFERENCE: literal OP x  ==  x OP' literal
    if (semantics != IPS_SEM_REFERENCE) return o;
    if (o == IPS_OP_LT) return (int)IPS_OP_GT;
    if (o == IPS_OP_GT) return (int)IPS_OP_LT;
    if (o == IPS_OP_LE) return (int)IPS_OP_GE;
    if (o == IPS_OP_GE) return (int)IPS_OP_LE;
    return o;
  };
  return launch_plain_scan(type, d_page, n_rows, flip(op), literals, n_literals, literal2 ? 1 : 0,
                           literal2 ? flip(op2) : 0, literal2, d_bitmap, d_batch_values,
                           d_batch_counts, S(stream));
}

ips_status ips_plain_select(const void* d_page, int64_t n_rows, ips_type type,
                            const uint64_t* d_bitmap, void* d_batch_values,
                            uint32_t* d_batch_counts, ips_stream stream) {
  IPS_REQUIRE(valid_type(type), "ips_plain_select: bad type %d", (int)type);
  IPS_REQUIRE(n_rows >= 0, "ips_plain_select: n_rows < 0");
  IPS_REQUIRE(n_rows == 0 || (d_page && aligned16(d_page) && d_bitmap && aligned16(d_bitmap) &&
                              d_batch_values && aligned16(d_batch_values) && d_batch_counts),
              "ips_plain_select: NULL or misaligned argument");
  if (n_rows == 0) return IPS_OK;
  return launch_plain_select(ips_plain_stride(type), d_page, n_rows, d_bitmap, d_batch_values,
                             d_batch_counts, S(stream));
}

// ---- bitmap algebra -------------------------------------------------------------------------
ips_status ips_bitmap_and(uint64_t* d_a, const uint64_t* d_b, int64_t n_rows, ips_stream stream) {
  IPS_REQUIRE(n_rows >= 0 && (n_rows == 0 || (d_a && d_b)), "ips_bitmap_and: bad argument");
  return launch_bitmap_binop(0, d_a, d_b, (n_rows + 63) / 64, S(stream));
}
ips_status ips_bitmap_or(uint64_t* d_a, const uint64_t* d_b, int64_t n_rows, ips_stream stream) {
  IPS_REQUIRE(n_rows >= 0 && (n_rows == 0 || (d_a && d_b)), "ips_bitmap_or: bad argument");
  return launch_bitmap_binop(1, d_a, d_b, (n_rows + 63) / 64, S(stream));
}
ips_status ips_bitmap_fill(uint64_t* d_a, int64_t n_rows, int value, ips_stream stream) {
  IPS_REQUIRE(n_rows >= 0 && (n_rows == 0 || d_a), "ips_bitmap_fill: bad argument");
  return launch_bitmap_fill(d_a, n_rows, value, S(stream));
}
ips_status ips_bitmap_count(const uint64_t* d_a, int64_t n_rows, int64_t* d_count, ips_stream stream) {
  IPS_REQUIRE(n_rows >= 0 && d_count && (n_rows == 0 || d_a), "ips_bitmap_count: bad argument");
  return launch_bitmap_count(d_a, n_rows, d_count, S(stream));
}
ips_status ips_bitmap_batch_counts(const uint64_t* d_a, int64_t n_rows, uint32_t* d_batch_counts, ips_stream stream) {
  IPS_REQUIRE(n_rows >= 0 && (n_rows == 0 || (d_a && d_batch_counts)), "ips_bitmap_batch_counts: bad argument");
  return launch_bitmap_batch_counts(d_a, n_rows, d_batch_counts, S(stream));
}
size_t ips_expand_workspace_bytes(int64_t n_rows) {  // shared by ips_bitmap_expand and ips_bitmap_compress
  const size_t a = scan_workspace_bytes((n_rows + 63) / 64), b = rank_workspace_bytes(n_rows < 0 ? 0 : n_rows);
  return a > b ? a : b;
}
ips_status ips_bitmap_compress(const uint64_t* d_mask, const uint64_t* d_src, int64_t n_rows,
                               uint64_t* d_out, int64_t* d_n_out, void* d_workspace,
                               ips_stream stream) {
  IPS_REQUIRE(n_rows >= 0, "ips_bitmap_compress: n_rows < 0");
  IPS_REQUIRE(n_rows == 0 || (d_mask && d_src && d_out && d_workspace), "ips_bitmap_compress: NULL argument");
  return launch_bitmap_compress(d_mask, d_src, n_rows, d_out, d_n_out, d_workspace, S(stream));
}
ips_status ips_bitmap_expand(const uint64_t* d_root, const uint64_t* d_sub, int64_t n_rows,
                             uint64_t* d_out, void* d_workspace, ips_stream stream) {
  IPS_REQUIRE(n_rows >= 0, "ips_bitmap_expand: n_rows < 0");
  IPS_REQUIRE(n_rows == 0 || (d_root && d_sub && d_out && d_workspace), "ips_bitmap_expand: NULL argument");
  return launch_bitmap_expand(d_root, d_sub, n_rows, d_out, d_workspace, S(stream));
}

ips_status ips_synth_splitmix_u32(uint64_t seed, int64_t n, uint32_t mask, uint32_t* d_out,
                                  ips_stream stream) {
  IPS_REQUIRE(n >= 0 && (n == 0 || d_out), "ips_synth_splitmix_u32: bad argument");
  return launch_synth(seed, n, mask, d_out, S(stream));
}

}  // extern "C"
