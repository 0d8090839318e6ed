// ips_fle_scan.hip -- instantiations + launchers of the fused scan / select kernels for the bit
// widths [IPS_WLO, IPS_WHI] (the file is compiled four times so the build parallelises).
#include <stdlib.h>

#include "ips_fle_kernels.h"
#include "ips_chunk_host.h"

#ifndef IPS_WLO
#error "compile with -DIPS_WLO=.. -DIPS_PART=.."
#endif
#define IPS_CAT2(a, b) a##b
#define IPS_CAT(a, b) IPS_CAT2(a, b)

namespace ips {

// IN lists of at least this many constants take the membership-table path (dev override:
// IPS_IN_TABLE_MIN=<K> for every width)
static int in_table_forced() {
  static const int forced = [] { const char* e = dev_env("IPS_IN_TABLE_MIN"); return e ? atoi(e) : 0; }();
  return forced;
}
static int in_table_min(int w) { return in_table_forced() > 0 ? in_table_forced() : in_table_min_consts(w); }
static int in_table_min_pred(int w) { return in_table_forced() > 0 ? in_table_forced() : in_table_min_consts_pred(w); }

template <int W, int MODE, int G>
static ips_status launch_one(const uint64_t* enc, int64_t n_rows, const PredArgs& args,
                             uint32_t* bitmap32, const uint32_t* given32, void* batch_values,
                             uint32_t* batch_counts, const void* dict, uint32_t dict_entries,
                             int32_t* bad_index, hipStream_t s) {
  using GT = typename GatherT<G>::type;
  auto kern = fle_scan_kernel<W, MODE, G>;
  const int64_t tiles = (n_rows + kRowsPerTile - 1) / kRowsPerTile;
  int grid = grid_for_tiles(reinterpret_cast<const void*>(kern), tiles, W >= 16 ? kGridScanWide : kGridScan);
  if (grid <= 0) return IPS_ERR_HIP;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), 0, s, enc, n_rows, args, bitmap32, given32,
                     reinterpret_cast<GT*>(batch_values), batch_counts,
                     reinterpret_cast<const GT*>(dict), dict_entries, bad_index);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

template <int W>
static ips_status launch_w(int mode, int gather, const uint64_t* enc, int64_t n_rows,
                           const PredArgs& args, uint32_t* bitmap32, const uint32_t* given32,
                           void* batch_values, uint32_t* batch_counts, const void* dict,
                           uint32_t dict_entries, int32_t* bad_index, hipStream_t s) {
#define IPS_ARGS enc, n_rows, args, bitmap32, given32, batch_values, batch_counts, dict, \
                 dict_entries, bad_index, s
  if (gather == 0) {
    if (mode == kScanPredicate) return launch_one<W, kScanPredicate, 0>(IPS_ARGS);
    if (mode == kScanInList) {
      if constexpr (W <= 16) {
        if (args.n_consts >= in_table_min(W)) return launch_one<W, kScanInTable, 0>(IPS_ARGS);
      }
      return launch_one<W, kScanInList, 0>(IPS_ARGS);
    }
    return launch_one<W, kScanGivenBitmap, 0>(IPS_ARGS);
  }
  if constexpr (W <= 16) {  // dictionaries hold <= 40000 entries: codes are <= 16 bits wide
    if (mode == kScanPredicate) {
      if (gather == 4) return launch_one<W, kScanPredicate, 4>(IPS_ARGS);
      if (gather == 8) return launch_one<W, kScanPredicate, 8>(IPS_ARGS);
    }
    if (mode == kScanInList && args.n_consts >= in_table_min(W)) {
      if (gather == 4) return launch_one<W, kScanInTable, 4>(IPS_ARGS);
      if (gather == 8) return launch_one<W, kScanInTable, 8>(IPS_ARGS);
    }
    if (mode == kScanInList) {
      if (gather == 4) return launch_one<W, kScanInList, 4>(IPS_ARGS);
      if (gather == 8) return launch_one<W, kScanInList, 8>(IPS_ARGS);
    }
    if (mode == kScanGivenBitmap) {
      if (gather == 4) return launch_one<W, kScanGivenBitmap, 4>(IPS_ARGS);
      if (gather == 8) return launch_one<W, kScanGivenBitmap, 8>(IPS_ARGS);
    }
  }
#undef IPS_ARGS
  set_error("fused dictionary scan: unsupported bit width %d / gather %d / mode %d", W, gather,
            mode);
  return IPS_ERR_UNSUPPORTED;
}

ips_status IPS_CAT(launch_fle_scan_part_, IPS_PART)(
    int w, int mode, int gather, const uint64_t* enc, int64_t n_rows, const PredArgs& args,
    uint32_t* bitmap32, const uint32_t* given32, void* batch_values, uint32_t* batch_counts,
    const void* dict, uint32_t dict_entries, int32_t* bad_index, hipStream_t s) {
#define IPS_CASE(N)                                                                          \
  case IPS_WLO + N:                                                                          \
    return launch_w<IPS_WLO + N>(mode, gather, enc, n_rows, args, bitmap32, given32,         \
                                 batch_values, batch_counts, dict, dict_entries, bad_index, s);
  switch (w) {
    IPS_CASE(0) IPS_CASE(1) IPS_CASE(2) IPS_CASE(3) IPS_CASE(4) IPS_CASE(5) IPS_CASE(6) IPS_CASE(7)
  }
#undef IPS_CASE
  set_error("bit width %d outside part starting at %d", w, IPS_WLO);
  return IPS_ERR_INVALID_ARG;
}

template <int W>
static ips_status launch_pages_w(const PageBatch& batch, int n_pages, int64_t max_rows,
                                 const PredArgs& args, hipStream_t s) {
  auto kern = fle_scan_pages_kernel<W>;
  const int64_t tiles = (max_rows + kRowsPerTile - 1) / kRowsPerTile;  // of the largest page
  // x: shares of one page's sub-tiles, so that x * n_pages fills the device about grid_mult() times
  const int total = grid_for_tiles(reinterpret_cast<const void*>(kern), tiles * (int64_t)n_pages);
  int64_t gx = (total + n_pages - 1) / n_pages;
  const int64_t gx_max = (tiles + kWavesPerBlock - 1) / kWavesPerBlock;
  if (gx > gx_max) gx = gx_max;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)n_pages), dim3(kThreads), 0, s, batch, args);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

ips_status IPS_CAT(launch_fle_scan_pages_part_, IPS_PART)(int w, const PageBatch& batch,
                                                          int n_pages, int64_t max_rows,
                                                          const PredArgs& args, hipStream_t s) {
#define IPS_CASE(N) \
  case IPS_WLO + N: return launch_pages_w<IPS_WLO + N>(batch, n_pages, max_rows, args, s);
  switch (w) {
    IPS_CASE(0) IPS_CASE(1) IPS_CASE(2) IPS_CASE(3) IPS_CASE(4) IPS_CASE(5) IPS_CASE(6) IPS_CASE(7)
  }
#undef IPS_CASE
  set_error("bit width %d outside part starting at %d", w, IPS_WLO);
  return IPS_ERR_INVALID_ARG;
}

template <int W, int KIND>
static ips_status launch_pred_wk(const uint64_t* enc, int64_t n_rows, const PredArgs& args,
                                 uint32_t* bitmap32, hipStream_t s) {
  auto kern = fle_pred_w_kernel<W, KIND>;
  const int64_t tiles = (n_rows + kRowsPerTile - 1) / kRowsPerTile;
  // (w <= 4: a sub-tile is 1 KiB, shares of one sub-tile per wave cost more in dispatches than they even out:
  // 29 us at 8x, 33 us at 16x and beyond; from w = 6 on 32x is 3-7 % faster than 8x)
  int grid = grid_for_tiles(reinterpret_cast<const void*>(kern), tiles, W >= 6 ? kGridPred : kGridScan);
  if (grid <= 0) return IPS_ERR_HIP;
  hipLaunchKernelGGL(kern, dim3(grid + args.aux_blocks), dim3(kThreads), 0, s, enc, n_rows, args, bitmap32);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

template <int W>
static ips_status launch_pred_w(const uint64_t* enc, int64_t n_rows, const PredArgs& args,
                                uint32_t* bitmap32, hipStream_t s) {
  if constexpr (W == 32) {
    static const bool early = dev_env("IPS_NO_EARLY_PRUNE") == nullptr;
    if (early && args.op != 5) {
      const int64_t tiles = (n_rows + kRowsPerTile - 1) / kRowsPerTile;
      auto kern = args.join != 0 ? fle_pred32_early_kernel<32, true> : fle_pred32_early_kernel<32, false>;
      int grid = grid_for_tiles(reinterpret_cast<const void*>(kern), tiles, kGridPred);  // (8x: 98 us, 32x: 92.5)
      if (grid <= 0) return IPS_ERR_HIP;
      hipLaunchKernelGGL(kern, dim3(grid + args.aux_blocks), dim3(kThreads), 0, s, enc, n_rows, args, bitmap32);
      IPS_HIP_TRY(hipGetLastError());
      return IPS_OK;
    }
  }
  if (args.join != 0) return launch_pred_wk<W, kPredPair>(enc, n_rows, args, bitmap32, s);
  if (args.op == 5) {
    if constexpr (W <= 16) {
      if (args.n_consts >= in_table_min_pred(W)) return launch_pred_wk<W, kPredInTable>(enc, n_rows, args, bitmap32, s);
    }
    return launch_pred_wk<W, kPredInList>(enc, n_rows, args, bitmap32, s);
  }
  return launch_pred_wk<W, kPredSingle>(enc, n_rows, args, bitmap32, s);
}

// The fused nullable leaf (fle_leaf_kernel): one workgroup per quarter rank tile of output words.
// *taken = false (nothing launched) for the shapes that keep predicate and expand as separate
// launches: comparisons at w = 32 with IPS_NO_EARLY_PRUNE set.
template <int W>
static ips_status launch_leaf_w(const uint64_t* enc, int64_t n_sub, const PredArgs& args, uint64_t* out,
                                bool* taken, hipStream_t s) {
  const int64_t n_words = (args.aux_rows + 63) / 64;
  const dim3 grid((unsigned)((n_words + kExpWordsPerBlock - 1) / kExpWordsPerBlock));
  unsigned long long* o = reinterpret_cast<unsigned long long*>(out);
  *taken = true;
  if constexpr (W == 32) {
    // comparisons of full-width columns: the kernel prunes early (high planes first); with the
    // dev switch that turns pruning off they take predicate + expand
    static const bool early = dev_env("IPS_NO_EARLY_PRUNE") == nullptr;
    if (!early && args.op != 5) {
      *taken = false;
      return IPS_OK;
    }
  }
  if (args.join != 0) {
    hipLaunchKernelGGL((fle_leaf_kernel<W, kPredPair>), grid, dim3(kThreads), 0, s, enc, n_sub, args, o);
  } else if (args.op == 5) {
    if constexpr (W <= 16) {
      if (args.n_consts >= in_table_min_pred(W)) {
        hipLaunchKernelGGL((fle_leaf_kernel<W, kPredInTable>), grid, dim3(kThreads), 0, s, enc, n_sub, args, o);
        IPS_HIP_TRY(hipGetLastError());
        return IPS_OK;
      }
    }
    hipLaunchKernelGGL((fle_leaf_kernel<W, kPredInList>), grid, dim3(kThreads), 0, s, enc, n_sub, args, o);
  } else {
    hipLaunchKernelGGL((fle_leaf_kernel<W, kPredSingle>), grid, dim3(kThreads), 0, s, enc, n_sub, args, o);
  }
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

ips_status IPS_CAT(launch_fle_leaf_part_, IPS_PART)(int w, const uint64_t* enc, int64_t n_sub,
                                                    const PredArgs& args, uint64_t* out, bool* taken,
                                                    hipStream_t s) {
#define IPS_CASE(N) \
  case IPS_WLO + N: return launch_leaf_w<IPS_WLO + N>(enc, n_sub, args, out, taken, s);
  switch (w) {
    IPS_CASE(0) IPS_CASE(1) IPS_CASE(2) IPS_CASE(3) IPS_CASE(4) IPS_CASE(5) IPS_CASE(6) IPS_CASE(7)
  }
#undef IPS_CASE
  set_error("bit width %d outside part starting at %d", w, IPS_WLO);
  return IPS_ERR_INVALID_ARG;
}

// Late materialisation of an OPTIONAL column (fle_select_nullable_kernel): gather 0 = the FLE
// values themselves (32-bit), 4 / 8 = through a dictionary of 4- / 8-byte entries (codes <= 16 bits).
template <int W>
static ips_status launch_selnull_w(int gather, const uint64_t* enc, int64_t n_data, const SelNullArgs& a,
                                   void* dense, const void* dict, uint32_t dict_entries, int64_t* n_values,
                                   hipStream_t s) {
  const int64_t n_words = (a.n_rows + 63) / 64;
  const dim3 grid((unsigned)((n_words + kExpWordsPerBlock - 1) / kExpWordsPerBlock), a.pages ? (unsigned)a.n_pages : 1u);
  if (gather == 0) {
    hipLaunchKernelGGL((fle_select_nullable_kernel<W, 0>), grid, dim3(kThreads), 0, s, enc, n_data, a,
                       reinterpret_cast<uint32_t*>(dense), (const uint32_t*)nullptr, 0u, n_values);
  } else {
    if constexpr (W <= 16) {
      if (gather == 4)
        hipLaunchKernelGGL((fle_select_nullable_kernel<W, 4>), grid, dim3(kThreads), 0, s, enc, n_data, a,
                           reinterpret_cast<uint32_t*>(dense), reinterpret_cast<const uint32_t*>(dict), dict_entries, n_values);
      else
        hipLaunchKernelGGL((fle_select_nullable_kernel<W, 8>), grid, dim3(kThreads), 0, s, enc, n_data, a,
                           reinterpret_cast<uint64_t*>(dense), reinterpret_cast<const uint64_t*>(dict), dict_entries, n_values);
    } else {
      set_error("dictionary codes wider than 16 bits");
      return IPS_ERR_INVALID_ARG;
    }
  }
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

ips_status IPS_CAT(launch_fle_selnull_part_, IPS_PART)(int w, int gather, const uint64_t* enc, int64_t n_data,
                                                       const SelNullArgs& a, void* dense, const void* dict,
                                                       uint32_t dict_entries, int64_t* n_values, hipStream_t s) {
#define IPS_CASE(N) \
  case IPS_WLO + N: return launch_selnull_w<IPS_WLO + N>(gather, enc, n_data, a, dense, dict, dict_entries, n_values, s);
  switch (w) {
    IPS_CASE(0) IPS_CASE(1) IPS_CASE(2) IPS_CASE(3) IPS_CASE(4) IPS_CASE(5) IPS_CASE(6) IPS_CASE(7)
  }
#undef IPS_CASE
  set_error("bit width %d outside part starting at %d", w, IPS_WLO);
  return IPS_ERR_INVALID_ARG;
}

// ---- the paged kernels: one launch over a run of pages of one bit width (blockIdx.y = page) ----
template <int W, int KIND>
static ips_status launch_pred_pages_wk(const ChunkPage* d_pages, int n_pages, int64_t max_rows, int64_t chunk_rows,
                                       const PredArgs& args, uint32_t* bitmap32, hipStream_t s) {
  auto kern = fle_pred_pages_kernel<W, KIND>;
  const int64_t tiles = (max_rows + kRowsPerTile - 1) / kRowsPerTile;
  const int gx = paged_grid_x(reinterpret_cast<const void*>(kern), tiles, n_pages);
  if (gx <= 0) return IPS_ERR_HIP;
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)n_pages), dim3(kThreads), 0, s, d_pages, chunk_rows, args, bitmap32);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

template <int W>
static ips_status launch_pred_pages_w(const ChunkPage* d_pages, int n_pages, int64_t max_rows, int64_t chunk_rows,
                                      const PredArgs& args, uint32_t* bitmap32, hipStream_t s) {
  if constexpr (W == 32) {
    static const bool early = dev_env("IPS_NO_EARLY_PRUNE") == nullptr;
    if (early && args.op != 5) {
      const int64_t tiles = (max_rows + kRowsPerTile - 1) / kRowsPerTile;
      auto kern = args.join != 0 ? fle_pred32_early_pages_kernel<32, true> : fle_pred32_early_pages_kernel<32, false>;
      const int gx = paged_grid_x(reinterpret_cast<const void*>(kern), tiles, n_pages);
      if (gx <= 0) return IPS_ERR_HIP;
      hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)n_pages), dim3(kThreads), 0, s, d_pages, chunk_rows, args, bitmap32);
      IPS_HIP_TRY(hipGetLastError());
      return IPS_OK;
    }
  }
#define IPS_P(KIND) launch_pred_pages_wk<W, KIND>(d_pages, n_pages, max_rows, chunk_rows, args, bitmap32, s)
  if (args.join != 0) return IPS_P(kPredPair);
  if (args.op == 5) {
    if constexpr (W <= 16) {
      if (args.n_consts >= in_table_min_pred(W)) return IPS_P(kPredInTable);
    }
    return IPS_P(kPredInList);
  }
  return IPS_P(kPredSingle);
#undef IPS_P
}

ips_status IPS_CAT(launch_fle_pred_pages_part_, IPS_PART)(int w, const ChunkPage* d_pages, int n_pages, int64_t max_rows,
                                                          int64_t chunk_rows, const PredArgs& args, uint32_t* bitmap32,
                                                          hipStream_t s) {
#define IPS_CASE(N) \
  case IPS_WLO + N: return launch_pred_pages_w<IPS_WLO + N>(d_pages, n_pages, max_rows, chunk_rows, args, bitmap32, s);
  switch (w) {
    IPS_CASE(0) IPS_CASE(1) IPS_CASE(2) IPS_CASE(3) IPS_CASE(4) IPS_CASE(5) IPS_CASE(6) IPS_CASE(7)
  }
#undef IPS_CASE
  set_error("bit width %d outside part starting at %d", w, IPS_WLO);
  return IPS_ERR_INVALID_ARG;
}

// the nullable leaf per page: grid.x = quarter rank tiles of the largest page
template <int W>
static ips_status launch_leaf_pages_w(const ChunkPage* d_pages, int n_pages, int64_t max_rows, int64_t chunk_rows,
                                      const PredArgs& args, uint32_t* bitmap32, hipStream_t s) {
  const int64_t n_words = (max_rows + 63) / 64;
  const dim3 grid((unsigned)((n_words + kExpWordsPerBlock - 1) / kExpWordsPerBlock), (unsigned)n_pages);
#define IPS_L(KIND) \
  hipLaunchKernelGGL((fle_leaf_pages_kernel<W, KIND>), grid, dim3(kThreads), 0, s, d_pages, chunk_rows, args, bitmap32)
  if (args.join != 0) {
    IPS_L(kPredPair);
  } else if (args.op == 5) {
    bool table = false;
    if constexpr (W <= 16) {
      if (args.n_consts >= in_table_min_pred(W)) {
        IPS_L(kPredInTable);
        table = true;
      }
    }
    if (!table) IPS_L(kPredInList);
  } else {
    IPS_L(kPredSingle);
  }
#undef IPS_L
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

ips_status IPS_CAT(launch_fle_leaf_pages_part_, IPS_PART)(int w, const ChunkPage* d_pages, int n_pages, int64_t max_rows,
                                                          int64_t chunk_rows, const PredArgs& args, uint32_t* bitmap32,
                                                          hipStream_t s) {
#define IPS_CASE(N) \
  case IPS_WLO + N: return launch_leaf_pages_w<IPS_WLO + N>(d_pages, n_pages, max_rows, chunk_rows, args, bitmap32, s);
  switch (w) {
    IPS_CASE(0) IPS_CASE(1) IPS_CASE(2) IPS_CASE(3) IPS_CASE(4) IPS_CASE(5) IPS_CASE(6) IPS_CASE(7)
  }
#undef IPS_CASE
  set_error("bit width %d outside part starting at %d", w, IPS_WLO);
  return IPS_ERR_INVALID_ARG;
}

// fused scan / late materialisation over the pages of a chunk
template <int W, int MODE, int G>
static ips_status launch_scan_chunk_one(const ChunkPage* d_pages, int n_pages, int64_t max_rows, int64_t chunk_rows,
                                        const PredArgs& args, uint32_t* bitmap32, const uint32_t* given32,
                                        void* batch_values, uint32_t* batch_counts, const void* dict,
                                        uint32_t dict_entries, int32_t* bad_index, hipStream_t s) {
  using GT = typename GatherT<G>::type;
  auto kern = fle_scan_chunk_kernel<W, MODE, G>;
  const int64_t tiles = (max_rows + kRowsPerTile - 1) / kRowsPerTile;
  const int gx = paged_grid_x(reinterpret_cast<const void*>(kern), tiles, n_pages);
  if (gx <= 0) return IPS_ERR_HIP;
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)n_pages), dim3(kThreads), 0, s, d_pages, chunk_rows, args,
                     bitmap32, given32, reinterpret_cast<GT*>(batch_values), batch_counts,
                     reinterpret_cast<const GT*>(dict), dict_entries, bad_index);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

template <int W>
static ips_status launch_scan_chunk_w(int mode, int gather, const ChunkPage* d_pages, int n_pages, int64_t max_rows,
                                      int64_t chunk_rows, const PredArgs& args, uint32_t* bitmap32,
                                      const uint32_t* given32, void* batch_values, uint32_t* batch_counts,
                                      const void* dict, uint32_t dict_entries, int32_t* bad_index, hipStream_t s) {
#define IPS_ARGS d_pages, n_pages, max_rows, chunk_rows, args, bitmap32, given32, batch_values, batch_counts, dict, \
                 dict_entries, bad_index, s
  if (gather == 0) {
    if (mode == kScanPredicate) return launch_scan_chunk_one<W, kScanPredicate, 0>(IPS_ARGS);
    if (mode == kScanInList) {
      if constexpr (W <= 16) {
        if (args.n_consts >= in_table_min(W)) return launch_scan_chunk_one<W, kScanInTable, 0>(IPS_ARGS);
      }
      return launch_scan_chunk_one<W, kScanInList, 0>(IPS_ARGS);
    }
    return launch_scan_chunk_one<W, kScanGivenBitmap, 0>(IPS_ARGS);
  }
  if constexpr (W <= 16) {
    if (mode == kScanPredicate) {
      if (gather == 4) return launch_scan_chunk_one<W, kScanPredicate, 4>(IPS_ARGS);
      if (gather == 8) return launch_scan_chunk_one<W, kScanPredicate, 8>(IPS_ARGS);
    }
    if (mode == kScanInList && args.n_consts >= in_table_min(W)) {
      if (gather == 4) return launch_scan_chunk_one<W, kScanInTable, 4>(IPS_ARGS);
      if (gather == 8) return launch_scan_chunk_one<W, kScanInTable, 8>(IPS_ARGS);
    }
    if (mode == kScanInList) {
      if (gather == 4) return launch_scan_chunk_one<W, kScanInList, 4>(IPS_ARGS);
      if (gather == 8) return launch_scan_chunk_one<W, kScanInList, 8>(IPS_ARGS);
    }
    if (mode == kScanGivenBitmap) {
      if (gather == 4) return launch_scan_chunk_one<W, kScanGivenBitmap, 4>(IPS_ARGS);
      if (gather == 8) return launch_scan_chunk_one<W, kScanGivenBitmap, 8>(IPS_ARGS);
    }
  }
#undef IPS_ARGS
  set_error("fused dictionary scan over pages: unsupported bit width %d / gather %d / mode %d", W, gather, mode);
  return IPS_ERR_UNSUPPORTED;
}

ips_status IPS_CAT(launch_fle_scan_chunk_part_, IPS_PART)(
    int w, int mode, int gather, const ChunkPage* d_pages, int n_pages, int64_t max_rows, int64_t chunk_rows,
    const PredArgs& args, uint32_t* bitmap32, const uint32_t* given32, void* batch_values, uint32_t* batch_counts,
    const void* dict, uint32_t dict_entries, int32_t* bad_index, hipStream_t s) {
#define IPS_CASE(N)                                                                                              \
  case IPS_WLO + N:                                                                                              \
    return launch_scan_chunk_w<IPS_WLO + N>(mode, gather, d_pages, n_pages, max_rows, chunk_rows, args, bitmap32, \
                                            given32, batch_values, batch_counts, dict, dict_entries, bad_index, s);
  switch (w) {
    IPS_CASE(0) IPS_CASE(1) IPS_CASE(2) IPS_CASE(3) IPS_CASE(4) IPS_CASE(5) IPS_CASE(6) IPS_CASE(7)
  }
#undef IPS_CASE
  set_error("bit width %d outside part starting at %d", w, IPS_WLO);
  return IPS_ERR_INVALID_ARG;
}

ips_status IPS_CAT(launch_fle_pred_part_, IPS_PART)(int w, const uint64_t* enc, int64_t n_rows,
                                                    const PredArgs& args, uint32_t* bitmap32,
                                                    hipStream_t s) {
#define IPS_CASE(N) \
  case IPS_WLO + N: return launch_pred_w<IPS_WLO + N>(enc, n_rows, args, bitmap32, s);
  switch (w) {
    IPS_CASE(0) IPS_CASE(1) IPS_CASE(2) IPS_CASE(3) IPS_CASE(4) IPS_CASE(5) IPS_CASE(6) IPS_CASE(7)
  }
#undef IPS_CASE
  set_error("bit width %d outside part starting at %d", w, IPS_WLO);
  return IPS_ERR_INVALID_ARG;
}

}  // namespace ips
