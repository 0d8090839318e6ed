// ips_chunk_host.h -- the host side of a column chunk held as a list of pages (ips_chunk,
// include/ips.h) and the launchers of the paged kernels, shared by the translation units.
#pragma once
#include <vector>

#include "ips_chunk_device.h"
#include "ips_host.h"

// Pages of one bit width form a run that one launch covers (blockIdx.y = page of the run): the
// dictionary writer stores the width of each data page in its first byte, and the width grows with
// the dictionary (dict-encoding.h:425-447), so the pages of one chunk may differ.
struct ips_chunk {
  int encoding;        // ips_col_encoding
  int type;            // PLAIN: ips_type
  int max_def_level;   // 0 REQUIRED; 1 OPTIONAL (flat schema: width-1 levels)
  int64_t n_rows;      // rows of all pages
  int64_t n_batches;   // sum over the pages of ceil(rows / 2048)
  struct Run { int bit_width; int first; int count; int64_t max_rows; int64_t max_data; };
  std::vector<ips::ChunkPage> pages;  // the non-empty pages, in row order
  std::vector<Run> runs;
  ips::ChunkPage* d_pages;            // the same on the device
  uint32_t rank_entries;              // OPTIONAL: entries (uint32) of all pages' tile-count tables
  // pages that start inside a bitmap dword: 4 dwords per batch slot for the end dwords of the sub-tiles
  // (edge mode of the paged kernels, ips_chunk_device.h); NULL when every page starts on a multiple of
  // 32 rows.  One evaluation at a time may use it (handles are thread-compatible).
  uint32_t* d_edges;
};

namespace ips {

// grid.x of a paged launch: shares of one page's sub-tiles such that x * pages fills the device
// about grid_mult() times (never more than the largest page's tiles need)
int paged_grid_x(const void* kernel, int64_t max_tiles, int n_pages);

// per-width launchers (ips_fle_scan.hip): pages = first page of the run (device)
ips_status launch_fle_pred_pages(int w, const ChunkPage* d_pages, int n_pages, int64_t max_rows, int64_t chunk_rows,
                                 const PredArgs& args, uint32_t* bitmap32, hipStream_t s);
ips_status launch_fle_leaf_pages(int w, const ChunkPage* d_pages, int n_pages, int64_t max_rows, int64_t chunk_rows,
                                 const PredArgs& args, uint32_t* bitmap32, hipStream_t s);
ips_status launch_fle_scan_chunk(int w, int mode, int gather, const ChunkPage* d_pages, int n_pages, int64_t max_rows,
                                 int64_t chunk_rows, const PredArgs& args, uint32_t* bitmap32, const uint32_t* given32,
                                 void* batch_values, uint32_t* batch_counts, const void* dict, uint32_t dict_entries,
                                 int32_t* bad_index, hipStream_t s);
// PLAIN pages (ips_plain.hip)
// (edges: the chunk's edge slots or NULL; 4-byte slots use them and run the fix-up themselves)
ips_status launch_plain_pred_pages(int type, const ChunkPage* d_pages, int n_pages, int64_t max_rows, int64_t chunk_rows,
                                   int op, const void* literals, int n_literals, uint64_t* bitmap, hipStream_t s,
                                   int combine, int join, int op2, const void* literal2, uint32_t* edges);
ips_status launch_plain_scan_pages(int type, const ChunkPage* d_pages, int n_pages, int64_t max_rows, int64_t chunk_rows,
                                   int op, const void* literals, int n_literals, int join, int op2, const void* literal2,
                                   uint64_t* bitmap, void* batch_values, uint32_t* batch_counts, hipStream_t s,
                                   uint32_t* edges);
ips_status launch_plain_select_pages(int stride_bytes, const ChunkPage* d_pages, int n_pages, int64_t max_rows,
                                     int64_t chunk_rows, const uint64_t* bitmap, void* batch_values,
                                     uint32_t* batch_counts, hipStream_t s);
// edge mode's second launch over the same pages (ips_chunk.hip)
ips_status launch_window_fixup(const ChunkPage* d_pages, int n_pages, int64_t max_rows, int64_t chunk_rows,
                               uint32_t* bitmap32, const uint32_t* edges, int combine, hipStream_t s, int run_dwords = 64);
// ips_chunk_select_nullable's first three launches (ips_rank.hip): the pages' selections as aligned bitmaps, the
// three count tables per page, the pages' first selected / first selected NOT-NULL row and the totals
ips_status launch_selnull_pages_prepare(const ChunkPage* d_pages, int n_pages, int64_t max_rows, const uint64_t* d_sel,
                                        int64_t chunk_rows, uint64_t* sel_copy, uint32_t* c_r, uint32_t* c_s,
                                        uint32_t* c_rs, uint64_t* page_s, uint64_t* page_rs, int64_t* counts,
                                        hipStream_t s);
// tile counts of every page's definition levels (ips_rank.hip): page p's table at counts + page.rank0
ips_status launch_rank_counts_pages(const ChunkPage* d_pages, int n_pages, int64_t max_rows, uint32_t* counts,
                                    hipStream_t s);

}  // namespace ips
