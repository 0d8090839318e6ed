// ips_chunk.hip -- a column chunk held as a LIST OF PAGES (ips_chunk, include/ips.h): the handle, and
// the fused scan / late materialisation over it.  The scanner's column reader works through its
// chunk page by page (ReadDataPage / InitDataPage, hdfs-parquet-scanner.cc:730-924) and cuts every
// batch at the page ends of all columns (:1837-1855); here the pages stay separate device buffers
// and ONE launch per run of equally wide pages covers them (blockIdx.y = page, ips_chunk_device.h).
#include <string.h>

#include "ips_chunk_host.h"

namespace ips {

int paged_grid_x(const void* kernel, int64_t max_tiles, int n_pages) {
  if (n_pages <= 0) return 0;
  int total = grid_for_tiles(kernel, max_tiles * (int64_t)n_pages);
  if (total <= 0) return 0;
  // (dev: fewer, longer-lived workgroups -- 1/2: Q6 over 573 pages 365 -> 384 us, 1/4: 408 us)
  static const int div = [] { const char* e = dev_env("IPS_PAGED_GRID_DIV"); int v = e ? atoi(e) : 1; return v > 0 ? v : 1; }();
  total = (total + div - 1) / div;
  int64_t gx = (total + n_pages - 1) / n_pages;
  const int64_t gx_max = (max_tiles + kWavesPerBlock - 1) / kWavesPerBlock;
  if (gx > gx_max) gx = gx_max;
  return (int)(gx < 1 ? 1 : gx);
}

#define IPS_DECL_PAGED(P)                                                                                         \
  ips_status launch_fle_pred_pages_part_##P(int, const ChunkPage*, int, int64_t, int64_t, const PredArgs&,        \
                                            uint32_t*, hipStream_t);                                              \
  ips_status launch_fle_leaf_pages_part_##P(int, const ChunkPage*, int, int64_t, int64_t, const PredArgs&,        \
                                            uint32_t*, hipStream_t);                                              \
  ips_status launch_fle_scan_chunk_part_##P(int, int, int, const ChunkPage*, int, int64_t, int64_t,               \
                                            const PredArgs&, uint32_t*, const uint32_t*, void*, uint32_t*,        \
                                            const void*, uint32_t, int32_t*, hipStream_t);
IPS_DECL_PAGED(a) IPS_DECL_PAGED(b) IPS_DECL_PAGED(c) IPS_DECL_PAGED(d)

ips_status launch_fle_pred_pages(int w, const ChunkPage* d_pages, int n_pages, int64_t max_rows, int64_t chunk_rows,
                                 const PredArgs& args, uint32_t* bitmap32, hipStream_t s) {
#define IPS_A w, d_pages, n_pages, max_rows, chunk_rows, args, bitmap32, s
  if (w <= 8) return launch_fle_pred_pages_part_a(IPS_A);
  if (w <= 16) return launch_fle_pred_pages_part_b(IPS_A);
  if (w <= 24) return launch_fle_pred_pages_part_c(IPS_A);
  return launch_fle_pred_pages_part_d(IPS_A);
}

ips_status launch_fle_leaf_pages(int w, const ChunkPage* d_pages, int n_pages, int64_t max_rows, int64_t chunk_rows,
                                 const PredArgs& args, uint32_t* bitmap32, hipStream_t s) {
  if (w <= 8) return launch_fle_leaf_pages_part_a(IPS_A);
  if (w <= 16) return launch_fle_leaf_pages_part_b(IPS_A);
  if (w <= 24) return launch_fle_leaf_pages_part_c(IPS_A);
  return launch_fle_leaf_pages_part_d(IPS_A);
#undef IPS_A
}

ips_status launch_fle_scan_chunk(int w, int mode, int gather, const ChunkPage* d_pages, int n_pages, int64_t max_rows,
                                 int64_t chunk_rows, const PredArgs& args, uint32_t* bitmap32, const uint32_t* given32,
                                 void* batch_values, uint32_t* batch_counts, const void* dict, uint32_t dict_entries,
                                 int32_t* bad_index, hipStream_t s) {
#define IPS_A w, mode, gather, d_pages, n_pages, max_rows, chunk_rows, args, bitmap32, given32, batch_values, \
              batch_counts, dict, dict_entries, bad_index, s
  if (w <= 8) return launch_fle_scan_chunk_part_a(IPS_A);
  if (w <= 16) return launch_fle_scan_chunk_part_b(IPS_A);
  if (w <= 24) return launch_fle_scan_chunk_part_c(IPS_A);
  return launch_fle_scan_chunk_part_d(IPS_A);
#undef IPS_A
}

// Edge mode's second launch: thread t of page blockIdx.y owns the bitmap dword at the START of the
// page's sub-tile t (t = tiles: the dword behind the last one): the high part of sub-tile t - 1 and the
// low part of sub-tile t, which the paged kernel left in the edge slots.  Inside a page such a dword is
// complete (plain store / read-modify-write); at the page's two ends it is shared with the
// neighbouring pages and merged atomically -- two atomics per page.
// run_dwords: dwords per run of the kernel that filled the slots (64: sub-tiles of 2048 rows; 62: the segmented chain)
__global__ __launch_bounds__(256) void window_fixup_kernel(const ChunkPage* __restrict__ pages, int64_t chunk_rows,
                                                           uint32_t* __restrict__ bitmap32,
                                                           const uint32_t* __restrict__ edges, int combine, int run_dwords) {
  const ChunkPage pg = pages[blockIdx.y];
  if (pg.n_rows <= 0) return;  // (an empty segment of the segmented chain)
  const BitmapWindow w = bitmap_window(bitmap32, pg, chunk_rows);
  if (w.shift == 0u) return;
  const int64_t run_rows = (int64_t)run_dwords * 32;
  const int64_t tiles = (pg.n_rows + run_rows - 1) / run_rows;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t > tiles) return;
  const uint32_t s = w.shift, r = 32u - s;
  const uint32_t* e = edges + 4 * (size_t)pg.batch0;
  auto rows_mask = [&](int64_t dword) -> uint32_t {  // rows of the page inside page-relative dword 'dword'
    const int64_t valid = pg.n_rows - dword * 32;
    return valid >= 32 ? ~0u : valid <= 0 ? 0u : ((1u << valid) - 1u);
  };
  uint32_t val = 0u, mask = 0u;
  if (t < tiles) {
    val |= e[4 * t];
    mask |= rows_mask(run_dwords * t) << s;
  }
  if (t > 0) {
    val |= e[4 * (t - 1) + 1];
    mask |= rows_mask(run_dwords * t - 1) >> r;
  }
  const int64_t d = run_dwords * t;
  const bool tail = w.own_tail && d == w.tail_dword && mask != 0u;
  if (tail) mask |= w.tail_mask;
  window_put(w.base + d, val & mask, mask, combine, 0u);
  if (tail && w.tail_extra && combine != 2) IPS_BITMAP_STORE(w.base + d + 1, 0u);
}

ips_status launch_window_fixup(const ChunkPage* d_pages, int n_pages, int64_t max_rows, int64_t chunk_rows,
                               uint32_t* bitmap32, const uint32_t* edges, int combine, hipStream_t s, int run_dwords) {
  if (n_pages <= 0 || edges == nullptr) return IPS_OK;
  const int64_t tiles = (max_rows + (int64_t)run_dwords * 32 - 1) / ((int64_t)run_dwords * 32);
  hipLaunchKernelGGL(window_fixup_kernel, dim3((unsigned)((tiles + 1 + 255) / 256), (unsigned)n_pages), dim3(256), 0, s,
                     d_pages, chunk_rows, bitmap32, edges, combine, run_dwords);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

// The predicate of one operand on a run of pages of width bw.  A constant that does not fit the
// run's width (a code of a later, larger dictionary state) makes that comparison
// constant on the run: always true becomes GE 0, always false LT 0 -- evaluated by the same kernels,
// so an OPTIONAL column still selects exactly its NOT-NULL rows.
void run_pred_args(int bw, int op, const uint64_t* consts, int n_consts, int join, int op2, uint64_t const2,
                   int combine, PredArgs* a) {
  memset(a, 0, sizeof(*a));
  const uint64_t limit = bw >= 32 ? 0xFFFFFFFFull : ((1ull << bw) - 1ull);
  auto fold = [&](int o, uint64_t c, int32_t* out_op, uint32_t* out_c) {
    if (c <= limit) { *out_op = o; *out_c = (uint32_t)c; return; }
    const bool always = o == IPS_OP_LT || o == IPS_OP_LE;
    *out_op = always ? IPS_OP_GE : IPS_OP_LT;
    *out_c = 0u;
  };
  a->combine = combine;
  if (op == IPS_OP_IN) {
    int n = 0;
    for (int i = 0; i < n_consts && n < IPS_MAX_IN_LIST; ++i)
      if (consts[i] <= limit) a->consts[n++] = (uint32_t)consts[i];
    a->op = n ? IPS_OP_IN : IPS_OP_LT;
    a->n_consts = n ? n : 1;
    if (!n) a->consts[0] = 0u;
    return;
  }
  a->n_consts = 1;
  fold(op, consts[0], &a->op, &a->consts[0]);
  if (join != 0) {
    a->join = join;
    fold(op2, const2, &a->op2, &a->const2);
  }
}

}  // namespace ips

using namespace ips;

static inline hipStream_t S(ips_stream s) { return reinterpret_cast<hipStream_t>(s); }

extern "C" {

ips_status ips_chunk_open(const ips_chunk_page* h_pages, int n_pages, int encoding, ips_type type,
                          int max_def_level, ips_chunk** chunk) {
  IPS_REQUIRE(chunk != nullptr, "ips_chunk_open: NULL out pointer");
  IPS_REQUIRE(n_pages >= 0 && (n_pages == 0 || h_pages), "ips_chunk_open: bad page list");
  IPS_REQUIRE(encoding == IPS_COL_FLE || encoding == IPS_COL_PLAIN, "ips_chunk_open: bad encoding %d", encoding);
  IPS_REQUIRE(encoding != IPS_COL_PLAIN || (type >= IPS_T_INT8 && type <= IPS_T_DOUBLE), "ips_chunk_open: bad type %d", (int)type);
  IPS_REQUIRE(max_def_level >= 0, "ips_chunk_open: max_def_level < 0");
  if (max_def_level > 0) {
    // the vectorised path of the reference knows flat schemas only (width-1 levels, max level 1:
    // hdfs-parquet-scanner.cc:338-345 with the writer of hdfs-parquet-table-writer.cc:387-399); its PLAIN
    // branch ignores the levels altogether (:346-348)
    if (max_def_level != 1 || encoding != IPS_COL_FLE) {
      set_error("ips_chunk_open: OPTIONAL chunks are FLE / dictionary coded with max_def_level 1");
      return IPS_ERR_UNSUPPORTED;
    }
  }
  ips_chunk* c = new ips_chunk();
  c->encoding = encoding;
  c->type = (int)type;
  c->max_def_level = max_def_level;
  c->n_rows = 0;
  c->n_batches = 0;
  c->d_pages = nullptr;
  c->d_edges = nullptr;
  c->rank_entries = 0;
  auto fail = [&](const char* msg, int i) {
    set_error("ips_chunk_open: page %d: %s", i, msg);
    delete c;
    return IPS_ERR_INVALID_ARG;
  };
  for (int i = 0; i < n_pages; ++i) {
    const ips_chunk_page& hp = h_pages[i];
    if (hp.n_rows < 0) return fail("n_rows < 0", i);
    if (hp.n_rows == 0) continue;
    if (!hp.d_data && !(max_def_level > 0 && hp.n_data_rows == 0)) return fail("NULL data", i);
    if (!aligned16(hp.d_data)) return fail("data not 16-byte aligned", i);
    if (encoding == IPS_COL_FLE && (hp.bit_width < 1 || hp.bit_width > 32)) return fail("bit width not in 1..32", i);
    ChunkPage pg;
    memset(&pg, 0, sizeof(pg));
    pg.data = reinterpret_cast<const uint64_t*>(hp.d_data);
    pg.n_rows = hp.n_rows;
    pg.n_data = hp.n_rows;
    pg.row0 = c->n_rows;
    if (max_def_level > 0) {
      if (!hp.d_def_levels || !aligned16(hp.d_def_levels)) return fail("definition levels NULL or misaligned", i);
      if (hp.n_data_rows < 0) return fail("n_data_rows < 0", i);
      pg.levels = reinterpret_cast<const uint64_t*>(hp.d_def_levels);
      pg.n_data = hp.n_data_rows < hp.n_rows ? hp.n_data_rows : hp.n_rows;  // a page holds no more data rows than rows
      pg.rank0 = c->rank_entries;
      c->rank_entries += (uint32_t)(rank_tiles(hp.n_rows) * (1 + kWavesPerBlock));
    }
    const int64_t batches = (hp.n_rows + IPS_BATCH_ROWS - 1) / IPS_BATCH_ROWS;
    if (c->n_batches + batches >= (1ll << 32)) return fail("more than 2^32 batches in the chunk", i);
    pg.batch0 = (uint32_t)c->n_batches;
    c->n_batches += batches;
    c->n_rows += hp.n_rows;
    if (c->n_rows >= (1ll << 40)) return fail("the chunk holds 2^40 rows or more", i);
    const int bw = encoding == IPS_COL_FLE ? hp.bit_width : 0;
    if (c->runs.empty() || c->runs.back().bit_width != bw)
      c->runs.push_back(ips_chunk::Run{bw, (int)c->pages.size(), 0, 0, 0});
    ips_chunk::Run& run = c->runs.back();
    run.count += 1;
    if (pg.n_rows > run.max_rows) run.max_rows = pg.n_rows;
    if (pg.n_data > run.max_data) run.max_data = pg.n_data;
    c->pages.push_back(pg);
  }
  if (!c->pages.empty()) {
    c->pages.back().flags |= kPageLast;
    const size_t bytes = c->pages.size() * sizeof(ChunkPage);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->d_pages), bytes);
    if (e != hipSuccess) { delete c; return hip_fail(e, "hipMalloc(page table)"); }
    e = hipMemcpy(c->d_pages, c->pages.data(), bytes, hipMemcpyHostToDevice);
    bool shifted = false;
    for (const ChunkPage& pg : c->pages) shifted = shifted || (pg.row0 & 31) != 0;
    if (e == hipSuccess && shifted && max_def_level == 0)  // (OPTIONAL chunks go through the leaf kernel: no edge mode)
      e = hipMalloc(reinterpret_cast<void**>(&c->d_edges), (size_t)c->n_batches * 16);
    if (e != hipSuccess) {
      (void)hipFree(c->d_pages);
      if (c->d_edges) (void)hipFree(c->d_edges);
      delete c;
      return hip_fail(e, "ips_chunk_open(page table)");
    }
  }
  *chunk = c;
  return IPS_OK;
}

ips_status ips_chunk_close(ips_chunk* chunk) {
  if (!chunk) return IPS_OK;
  if (chunk->d_pages) (void)hipFree(chunk->d_pages);
  if (chunk->d_edges) (void)hipFree(chunk->d_edges);
  delete chunk;
  return IPS_OK;
}

int64_t ips_chunk_num_rows(const ips_chunk* chunk) { return chunk ? chunk->n_rows : -1; }
int64_t ips_chunk_num_batches(const ips_chunk* chunk) { return chunk ? chunk->n_batches : -1; }
int ips_chunk_num_pages(const ips_chunk* chunk) { return chunk ? (int)chunk->pages.size() : -1; }

}  // extern "C"

namespace {

ips_status check_scan_outputs(const ips_chunk* chunk, const void* d_bitmap, const void* d_batch_values,
                              const void* d_batch_counts, const char* fn) {
  IPS_REQUIRE(chunk != nullptr, "%s: NULL chunk", fn);
  IPS_REQUIRE(chunk->n_rows == 0 || (d_bitmap && aligned16(d_bitmap) && d_batch_values && aligned16(d_batch_values) &&
                                     d_batch_counts),
              "%s: output NULL or misaligned", fn);
  return IPS_OK;
}

// one logical predicate (op / constants) over every run of an FLE chunk
ips_status scan_runs(const ips_chunk* chunk, int mode_hint, int op, const uint64_t* consts, int n_consts, int gather,
                     const void* d_dict, uint32_t dict_entries, uint64_t* d_bitmap, const uint64_t* d_given,
                     void* d_batch_values, uint32_t* d_batch_counts, hipStream_t s) {
  for (const ips_chunk::Run& run : chunk->runs) {
    PredArgs args;
    int mode = mode_hint;
    if (mode_hint == kScanGivenBitmap) {
      memset(&args, 0, sizeof(args));
    } else {
      run_pred_args(run.bit_width, op, consts, n_consts, 0, 0, 0, 0, &args);
      mode = args.op == IPS_OP_IN ? kScanInList : kScanPredicate;
    }
    if (gather != 0 && run.bit_width > 16) {
      set_error("dictionary codes wider than 16 bits (page run of width %d)", run.bit_width);
      return IPS_ERR_INVALID_ARG;
    }
    const bool emits = mode_hint != kScanGivenBitmap;
    args.edges = emits ? chunk->d_edges : nullptr;
    ips_status st = launch_fle_scan_chunk(run.bit_width, mode, gather, chunk->d_pages + run.first, run.count,
                                          run.max_rows, chunk->n_rows, args, reinterpret_cast<uint32_t*>(d_bitmap),
                                          reinterpret_cast<const uint32_t*>(d_given), d_batch_values, d_batch_counts,
                                          d_dict, dict_entries, nullptr, s);
    if (st != IPS_OK) return st;
    if (emits && chunk->d_edges) {
      st = launch_window_fixup(chunk->d_pages + run.first, run.count, run.max_rows, chunk->n_rows,
                               reinterpret_cast<uint32_t*>(d_bitmap), chunk->d_edges, 0, s);
      if (st != IPS_OK) return st;
    }
  }
  return IPS_OK;
}

}  // namespace

extern "C" {

ips_status ips_chunk_fle_scan(const ips_chunk* chunk, ips_op op, const uint64_t* consts, int n_consts,
                              uint64_t* d_bitmap, uint32_t* d_batch_values, uint32_t* d_batch_counts,
                              ips_stream stream) {
  ips_status st = check_scan_outputs(chunk, d_bitmap, d_batch_values, d_batch_counts, "ips_chunk_fle_scan");
  if (st != IPS_OK) return st;
  IPS_REQUIRE(chunk->encoding == IPS_COL_FLE && chunk->max_def_level == 0, "ips_chunk_fle_scan: a REQUIRED FLE chunk");
  IPS_REQUIRE(op >= IPS_OP_EQ && op <= IPS_OP_IN, "ips_chunk_fle_scan: bad op %d", (int)op);
  IPS_REQUIRE(consts && n_consts >= 1 && n_consts <= IPS_MAX_IN_LIST && (op == IPS_OP_IN || n_consts == 1),
              "ips_chunk_fle_scan: bad constant list");
  return scan_runs(chunk, kScanPredicate, op, consts, n_consts, 0, nullptr, 0, d_bitmap, nullptr, d_batch_values,
                   d_batch_counts, S(stream));
}

ips_status ips_chunk_dict_scan(const ips_chunk* chunk, const ips_dict* dict, ips_op op, const void* literals,
                               int n_literals, uint64_t* d_bitmap, void* d_batch_values, uint32_t* d_batch_counts,
                               ips_stream stream) {
  ips_status st = check_scan_outputs(chunk, d_bitmap, d_batch_values, d_batch_counts, "ips_chunk_dict_scan");
  if (st != IPS_OK) return st;
  IPS_REQUIRE(chunk->encoding == IPS_COL_FLE && chunk->max_def_level == 0, "ips_chunk_dict_scan: a REQUIRED dictionary chunk");
  st = check_dict_call(dict, op, literals, n_literals, "ips_chunk_dict_scan");
  if (st != IPS_OK) return st;
  ips_xl_kind kind; ips_op fle_op; int n_codes = 0;
  uint64_t codes[IPS_MAX_IN_LIST];
  translate(dict, op, literals, n_literals, &kind, &fle_op, codes, &n_codes);
  if (kind == IPS_XL_ALL_FALSE) { fle_op = IPS_OP_LT; codes[0] = 0; n_codes = 1; }  // constant on every page
  if (kind == IPS_XL_ALL_TRUE) { fle_op = IPS_OP_GE; codes[0] = 0; n_codes = 1; }
  return scan_runs(chunk, kScanPredicate, fle_op, codes, n_codes, dict->slot, dict->d_entries, (uint32_t)dict->n,
                   d_bitmap, nullptr, d_batch_values, d_batch_counts, S(stream));
}

ips_status ips_chunk_select(const ips_chunk* chunk, const ips_dict* dict, const uint64_t* d_bitmap,
                            void* d_batch_values, uint32_t* d_batch_counts, ips_stream stream) {
  ips_status st = check_scan_outputs(chunk, d_bitmap, d_batch_values, d_batch_counts, "ips_chunk_select");
  if (st != IPS_OK) return st;
  IPS_REQUIRE(chunk->max_def_level == 0, "ips_chunk_select: a REQUIRED chunk (OPTIONAL pages: ips_dict_select_nullable page by page)");
  if (chunk->encoding == IPS_COL_PLAIN) {  // ReadValue(skip) -> ParquetPlainEncoder::Decode(.., skip_rows), parquet-common.h:186-190
    IPS_REQUIRE(dict == nullptr, "ips_chunk_select: a PLAIN chunk has no dictionary");
    if (chunk->pages.empty()) return IPS_OK;
    return launch_plain_select_pages(ips_plain_stride((ips_type)chunk->type), chunk->d_pages, (int)chunk->pages.size(),
                                     chunk->runs[0].max_rows, chunk->n_rows, d_bitmap, d_batch_values, d_batch_counts,
                                     S(stream));
  }
  return scan_runs(chunk, kScanGivenBitmap, 0, nullptr, 0, dict ? dict->slot : 0, dict ? dict->d_entries : nullptr,
                   dict ? (uint32_t)dict->n : 0u, nullptr, d_bitmap, d_batch_values, d_batch_counts, S(stream));
}

ips_status ips_chunk_plain_scan(const ips_chunk* chunk, ips_op op, const void* literals, int n_literals, ips_op op2,
                                const void* literal2, ips_semantics semantics, uint64_t* d_bitmap,
                                void* d_batch_values, uint32_t* d_batch_counts, ips_stream stream) {
  ips_status st = check_scan_outputs(chunk, d_bitmap, d_batch_values, d_batch_counts, "ips_chunk_plain_scan");
  if (st != IPS_OK) return st;
  IPS_REQUIRE(chunk->encoding == IPS_COL_PLAIN, "ips_chunk_plain_scan: a PLAIN chunk");
  IPS_REQUIRE(op >= IPS_OP_EQ && op <= IPS_OP_IN, "ips_chunk_plain_scan: bad op %d", (int)op);
  IPS_REQUIRE(literals && n_literals >= 1 && n_literals <= 16 && (op == IPS_OP_IN || n_literals == 1),
              "ips_chunk_plain_scan: 1..16 literals");
  IPS_REQUIRE(!literal2 || (op != IPS_OP_IN && op2 >= IPS_OP_EQ && op2 <= IPS_OP_GE),
              "ips_chunk_plain_scan: the second comparison must be EQ..GE on a non-IN first one");
  IPS_REQUIRE(semantics == IPS_SEM_REFERENCE || semantics == IPS_SEM_SQL, "ips_chunk_plain_scan: bad semantics");
  if (op == IPS_OP_IN && semantics == IPS_SEM_REFERENCE) {
    set_error("ips_chunk_plain_scan: IN has no reference behaviour on PLAIN pages (empty body, parquet-common.h:252-255)");
    return IPS_ERR_UNSUPPORTED;
  }
  if (chunk->pages.empty()) return IPS_OK;
  auto flip = [&](int o) {  // REFERENCE: literal OP x  ==  x OP' literal
    if (semantics != IPS_SEM_REFERENCE) return o;
    if (o == IPS_OP_LT) return (int)IPS_OP_GT;
    if (o == IPS_OP_GT) return (int)IPS_OP_LT;
    if (o == IPS_OP_LE) return (int)IPS_OP_GE;
    if (o == IPS_OP_GE) return (int)IPS_OP_LE;
    return o;
  };
  const ips_chunk::Run& run = chunk->runs[0];  // PLAIN pages form one run
  return launch_plain_scan_pages(chunk->type, chunk->d_pages, (int)chunk->pages.size(), run.max_rows, chunk->n_rows,
                                 flip(op), literals, n_literals, literal2 ? 1 : 0, literal2 ? flip(op2) : 0, literal2,
                                 d_bitmap, d_batch_values, d_batch_counts, S(stream), chunk->d_edges);
}

}  // extern "C"
