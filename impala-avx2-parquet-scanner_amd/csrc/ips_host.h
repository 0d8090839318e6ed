// ips_host.h -- host-side launch plumbing shared by the translation units of libips_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ips.h"
#include "ips_device.h"

#include <vector>

// DictDecoder<T>'s state (dict-encoding.h:449-459): the sorted entries on the host (literal
// translation) and on the device (gathers)
struct ips_dict {
  ips_type type;
  int64_t n;
  int elem;                     // sizeof(T)
  int slot;                     // PLAIN slot / device entry bytes: 4 or 8
  std::vector<uint8_t> host;    // n elements of sizeof(T), ascending
  void* d_entries;              // n entries of 'slot' bytes (int8/int16 sign-extended to int32)
};

// An IN list of any length (FleDecoder::In / DictDecoder::In take a vector, fle-encoding.h:8236-8313,
// dict-encoding.h:523-541; a dictionary holds up to 40000 codes): resident on the device as the
// 65536-bit membership table of the members below 2^16 and as the ascending list of all members.
struct ips_inset {
  std::vector<uint32_t> members;  // ascending, distinct (host copy: how many fit a width)
  uint32_t* d_table;              // 2048 dwords
  uint32_t* d_list;               // members.size() dwords (at least one)
};

namespace ips {

// the IN predicate over 'set' on a column of bw bits: *always_false when no member fits the width
void inset_pred_args(const ips_inset* set, int bw, PredArgs* args, bool* always_false);

// Outcome of comparing against constants that do not fit in bw bits (the
// reference is inconsistent there; the build defines it by the unsigned SQL meaning).
enum ConstKind { kEvaluate = 0, kAllFalse = 1, kAllTrue = 2 };
ips_status build_pred_args(int bw, ips_op op, const uint64_t* consts, int n_consts, PredArgs* args,
                           ConstKind* kind, const char* fn);
// DictDecoder<T>::Eq..In's literal -> code translation (dict-encoding.h:461-541)
ips_status translate(const ips_dict* d, ips_op op, const void* literals, int n_literals, ips_xl_kind* kind,
                     ips_op* fle_op, uint64_t* codes, int* n_codes);
ips_status check_dict_call(const ips_dict* dict, ips_op op, const void* literals, int n_literals, const char* fn);

// thread-local error text behind ips_last_error()
void set_error(const char* fmt, ...);
ips_status hip_fail(hipError_t e, const char* what);

#define IPS_HIP_TRY(expr)                                   \
  do {                                                      \
    hipError_t _e = (expr);                                 \
    if (_e != hipSuccess) return ::ips::hip_fail(_e, #expr); \
  } while (0)

#define IPS_REQUIRE(cond, ...)          \
  do {                                  \
    if (!(cond)) {                      \
      ::ips::set_error(__VA_ARGS__);    \
      return IPS_ERR_INVALID_ARG;       \
    }                                   \
  } while (0)

// Blocks to launch for a grid-stride kernel over 'tiles' wave sub-tiles: enough to fill every CU
// at the kernel's occupancy, never more than the work.  Occupancy is queried once per kernel.
enum GridKind { kGridScan = 0, kGridPred = 1, kGridChain = 2, kGridDecode = 3, kGridDictDecode = 4, kGridScanWide = 5 };  // grid_mult
int grid_for_tiles(const void* kernel, int64_t tiles, int kind = kGridScan);
int device_cus();
int grid_mult(int kind = kGridScan);

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- launchers implemented in the per-kernel translation units ----
// mode: 0 single-constant predicate, 1 given bitmap, 2 IN list.  gather: 0 / 4 / 8 bytes per dictionary entry.
ips_status launch_fle_scan(int w, int mode, int gather, const uint64_t* enc, int64_t n_rows,
                           const PredArgs& args, uint32_t* bitmap32, const uint32_t* given32,
                           void* batch_values, uint32_t* batch_counts, const void* dict,
                           uint32_t dict_entries, int32_t* bad_index, hipStream_t s);
ips_status launch_fle_decode(int w, int out_width, int gather, const uint64_t* enc, int64_t n_rows,
                             void* out, const void* dict, uint32_t dict_entries,
                             int32_t* bad_index, hipStream_t s);
ips_status launch_fle_encode(int w, int in_width, const void* values, int64_t n_rows,
                             uint64_t* enc, hipStream_t s);
ips_status launch_fle_pred(int w, const uint64_t* enc, int64_t n_rows, const PredArgs& args,
                           uint32_t* bitmap32, hipStream_t s);
// combine: 0 set / 1 and-into / 2 or-into the bitmap; join/op2/literal2: second predicate on the
// same column evaluated in the same pass (0 = none)
ips_status launch_plain_scan(int type, const void* page, int64_t n_rows, int op, const void* literals,
                             int n_literals, int join, int op2, const void* literal2,
                             uint64_t* bitmap, void* batch_values, uint32_t* batch_counts,
                             hipStream_t s);
ips_status launch_plain_select(int stride_bytes, const void* page, int64_t n_rows,
                               const uint64_t* bitmap, void* batch_values, uint32_t* batch_counts,
                               hipStream_t s);
ips_status launch_bitmap_binop(int op, uint64_t* a, const uint64_t* b, int64_t n_words, hipStream_t s);  // 0 and, 1 or
ips_status launch_plain_pred(int type, const void* page, int64_t n_rows, int op,
                             const void* literals, int n_literals, uint64_t* bitmap, hipStream_t s,
                             int combine = 0, int join = 0, int op2 = 0,
                             const void* literal2 = nullptr);

// rank-based kernels (ips_rank.hip).  root_kind 0: bitmap words, 1: width-1 definition levels
// with max_def_level 1 (the level words are the NOT-NULL bits, MSB first)
int64_t rank_tiles(int64_t n_rows);
size_t rank_workspace_bytes(int64_t n_rows);
ips_status launch_rank_tile_counts(int root_kind, const uint64_t* root, int64_t n_rows,
                                   uint32_t* tile_counts, hipStream_t s);
ips_status launch_expand(int root_kind, const uint64_t* root, const uint64_t* sub, int64_t n_rows,
                         int64_t n_sub_bits, const uint32_t* tile_counts, uint64_t* out,
                         int combine, hipStream_t s);
bool fused_leaf_enabled();
ips_status launch_fle_leaf(int w, int root_kind, const uint64_t* root, int64_t n_rows,
                           const uint32_t* tile_counts, const uint64_t* enc, int64_t n_sub,
                           const PredArgs& pred, uint64_t* out, int combine, bool* taken,
                           hipStream_t s);
ips_status launch_bitmap_fill(uint64_t* a, int64_t n_rows, int value, hipStream_t s);
ips_status launch_rank3_counts(int root_kind, const uint64_t* root, const uint64_t* sel, int64_t n_rows,
                               uint32_t* c_r, uint32_t* c_s, uint32_t* c_rs, uint64_t* zero_out,
                               hipStream_t s);
ips_status launch_compress_counted(int mask_kind, const uint64_t* mask, int src_kind, const uint64_t* src,
                                   int64_t n_rows, uint64_t* out, int64_t* n_out,
                                   const uint32_t* tile_counts, hipStream_t s);
ips_status launch_compress(int mask_kind, const uint64_t* mask, int src_kind, const uint64_t* src,
                           int64_t n_rows, uint64_t* out, int64_t* n_out, uint32_t* tile_counts,
                           hipStream_t s);

// Workspace of a nullable predicate leaf: tile counts | data-row bitmap | NOT-NULL bitmap (the
// last only when the definition levels are wider than one bit).
struct NullableWs {
  uint32_t* tile_counts;
  uint64_t* sub;
  uint64_t* nonnull;
};
size_t nullable_workspace_bytes(int64_t n_rows);
NullableWs nullable_workspace(void* d_workspace, int64_t n_rows);
// NOT-NULL root of an OPTIONAL column: returns the root kind and pointer after (if the levels are
// wider than a bit) evaluating def == max_def into ws.nonnull; then counts the tiles, unless the
// caller lets the counts ride on its next predicate launch (count_tiles = false, then
// attach_rank_counts on that launch's arguments).
ips_status nullable_prepare_root(const void* d_def_levels, int def_bit_width, int max_def_level,
                                 int64_t n_rows, const NullableWs& ws, int* root_kind,
                                 const uint64_t** root, hipStream_t s, bool count_tiles = true);
void attach_rank_counts(PredArgs* args, int root_kind, const uint64_t* root, int64_t n_rows,
                        uint32_t* tile_counts);

}  // namespace ips
