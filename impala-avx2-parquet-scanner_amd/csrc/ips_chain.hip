// ips_chain.hip -- a conjunct chain over REQUIRED FLE columns in ONE pass.
//
// EvalSimplePredicates' conjunct list (hdfs-parquet-scanner.cc:1857-1862) and any left-deep AND / OR
// chain of operands: every operand is a comparison, a pair of comparisons on one column (BETWEEN,
// simple-predicates.h:145-153) or a short IN list (FleDecoder::In, fle-encoding.h:8283-8290).  The
// per-operand plan runs one launch per operand and folds the results through the bitmap (2 bits of
// read-modify-write per row and operand: on the Q6 shape 2.03 GB moved for 1.73 GB of algorithmic
// bytes).  Here a wave keeps a 2048-row stripe's result in a register while it walks the operands:
// every column is read once and the bitmap is written once.
//
// The sub-tiles of all operands for the same 2048 rows form one STRIPE; its 16-byte chunks are dealt
// to LOAD SLOTS of 64 chunks, every slot inside one operand (w = 12: three slots, w = 6: two, w = 4:
// one).  A slot's descriptor -- a buffer resource over the operand's column prepared by the host, the
// slot's place in the sub-tile and in LDS -- is wave-uniform and arrives as one 32-byte scalar load,
// the loads and the staging into the operands' plane images take the width at run time (a handful of
// vector ops per slot), and the whole NEXT stripe is in flight in the slot registers while the
// current one is evaluated.  The evaluation of an operand is reached through a wave-uniform switch on
// its width: compile-time-width code, the building blocks of the stand-alone predicate kernels.
//
// Round 2's kernel (fle_chain_kernel) evaluated with run-time widths and re-derived resources per
// slot: 519 scalar instructions per stripe, a tie with the three launches.
#include "ips_device.h"
#include "ips_chunk_host.h"
#include "ips_chain.h"

#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

namespace ips {

// f(std::integral_constant<int, W>) for W == w; widths beyond MAXW are not compiled into the kernel
template <int MAXW, typename F>
__device__ __forceinline__ void width_switch(int w, F&& f) {
  switch (w) {
#define IPS_CASE(W)                                               \
  case W:                                                         \
    if constexpr (W <= MAXW) f(std::integral_constant<int, W>{}); \
    break;
    IPS_CASE(1) IPS_CASE(2) IPS_CASE(3) IPS_CASE(4) IPS_CASE(5) IPS_CASE(6) IPS_CASE(7) IPS_CASE(8)
    IPS_CASE(9) IPS_CASE(10) IPS_CASE(11) IPS_CASE(12) IPS_CASE(13) IPS_CASE(14) IPS_CASE(15) IPS_CASE(16)
    IPS_CASE(17) IPS_CASE(18) IPS_CASE(19) IPS_CASE(20) IPS_CASE(21) IPS_CASE(22) IPS_CASE(23) IPS_CASE(24)
    IPS_CASE(25) IPS_CASE(26) IPS_CASE(27) IPS_CASE(28) IPS_CASE(29) IPS_CASE(30) IPS_CASE(31) IPS_CASE(32)
#undef IPS_CASE
    default: break;
  }
}

// The descriptors are read from the kernel-argument segment through a constant-address-space pointer:
// scalar loads at run-time indices.  (Indexing the by-value argument itself made the compiler copy
// the whole structure to scratch.)
typedef const ChainArgsW __attribute__((address_space(4))) * ChainArgsK;
typedef const uint32_t __attribute__((address_space(4))) * ConstsK;
struct ChainStep {  // one operand, in scalar registers
  int w, kind, op, op2, join, combine, n_in, img_dw;
  uint32_t c1, c2;
  ConstsK in_consts, m1, m2;
};
__device__ __forceinline__ ChainStep chain_step(ChainArgsK a, int i) {
  ChainStep o;
  o.w = a->ops[i].w;
  o.kind = a->ops[i].kind;
  o.op = a->ops[i].op;
  o.op2 = a->ops[i].op2;
  o.join = a->ops[i].join;
  o.combine = a->ops[i].combine;
  o.n_in = a->ops[i].n_in;
  o.img_dw = a->ops[i].img_dw;
  o.c1 = a->ops[i].c1;
  o.c2 = a->ops[i].c2;
  o.in_consts = a->ops[i].in_consts;
  o.m1 = a->ops[i].m1;
  o.m2 = a->ops[i].m2;
  return o;
}

// the operand's rows of this lane (MSB-first, like every predicate building block).  Operands of up
// to 16 bits take their constants as plane masks from the descriptor (ChainOpW::m1 / m2).
template <int W>
__device__ __forceinline__ uint32_t chain_eval(const uint32_t* img, int lane, const ChainStep& o) {
  if (o.kind == kChainPair) {  // wave-uniform
    if constexpr (W > 16) {
      uint32_t r1, r2;
      pred_pair_from_lds(img, W, lane, o.op, o.c1, o.op2, o.c2, &r1, &r2);
      return o.join == 1 ? (r1 & r2) : (r1 | r2);
    } else {
      uint32_t p[W];
      planes_from_lds<W>(img, lane, p);
      uint32_t r1, r2;
      if (o.op != 0 && o.op2 != 0) {  // two borrow chains (BETWEEN and friends)
        uint32_t a1 = borrow_init(o.op), a2 = borrow_init(o.op2);
#pragma unroll
        for (int k = 0; k < W; ++k) {
          a1 = borrow_step(a1, p[k], o.m1[k]);
          a2 = borrow_step(a2, p[k], o.m2[k]);
        }
        r1 = borrow_select(a1, o.op);
        r2 = borrow_select(a2, o.op2);
      } else {
        const bool eq1 = o.op == 0, eq2 = o.op2 == 0;
        uint32_t a1 = eq1 ? ~0u : borrow_init(o.op), a2 = eq2 ? ~0u : borrow_init(o.op2);
#pragma unroll
        for (int k = 0; k < W; ++k) {
          a1 = eq1 ? eq_step(a1, p[k], o.m1[k]) : borrow_step(a1, p[k], o.m1[k]);
          a2 = eq2 ? eq_step(a2, p[k], o.m2[k]) : borrow_step(a2, p[k], o.m2[k]);
        }
        r1 = eq1 ? a1 : borrow_select(a1, o.op);
        r2 = eq2 ? a2 : borrow_select(a2, o.op2);
      }
      return o.join == 1 ? (r1 & r2) : (r1 | r2);
    }
  }
  if (o.kind == kChainIn) {
    if constexpr (W <= 16) {
      uint32_t p[W];
      planes_from_lds<W>(img, lane, p);
      return pred_in_from_regs<W>(p, o.in_consts, o.n_in);
    } else {
      return pred_in_from_lds(img, W, lane, o.in_consts, o.n_in);
    }
  }
  if constexpr (W > 16) {  // streamed from LDS eight planes at a time: no 32 plane registers next to the slots
    return pred_single_from_lds(img, W, lane, o.op, o.c1);
  } else {
    uint32_t p[W];
    planes_from_lds<W>(img, lane, p);
    if (o.op == 0) {
      uint32_t eq = ~0u;
#pragma unroll
      for (int k = 0; k < W; ++k) eq = eq_step(eq, p[k], o.m1[k]);
      return eq;
    }
    uint32_t b = borrow_init(o.op);
#pragma unroll
    for (int k = 0; k < W; ++k) b = borrow_step(b, p[k], o.m1[k]);
    return borrow_select(b, o.op);
  }
}

// Instantiated per (load slots, widest operand): the slot registers and the widest compiled-in
// predicate body set the register allocation, i.e. the waves per SIMD
constexpr int chain_min_waves(int ltot, int maxw) {
  return maxw <= 16 ? (ltot <= 4 ? 8 : ltot <= 6 ? 5 : ltot <= 8 ? 4 : 3) : (ltot <= 8 ? 4 : 3);
}

typedef const ChainPagedArgsW __attribute__((address_space(4))) * ChainPagedArgsK;

// a buffer resource whose four words the host prepared (no scalar ops in the kernel)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t host_rsrc(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
  const u32x4 words = {w0, w1, w2, w3};
  __amdgpu_buffer_rsrc_t rsrc;
  __builtin_memcpy(&rsrc, &words, 16);
  return rsrc;
}

// PAGED: blockIdx.y = page; the page's rows in every operand's own block geometry (FLE blocks restart at
// the page's first row, which is the same row in every operand's chunk), the stripe's bitmap dwords through
// the page's window into the chunk-wide bitmap.
template <int LTOT, int MAXW, bool PAGED>
__device__ __forceinline__ void chain_body(ChainArgsK a, ChainPagedArgsK pa, int64_t n_rows, uint32_t* __restrict__ bitmap32,
                                           uint32_t* lds_all) {
  const int lane = lane_id();
  const int wave = wave_id();
  uint32_t* lds32 = lds_all + wave * a->image_dwords;
  [[maybe_unused]] ChunkPage pg;
  [[maybe_unused]] BitmapWindow win;
  // PAGED: the page's bytes in every slot's operand, fetched side by side before anything depends on them
  // (slot by slot inside the loads, the two scalar loads per slot formed a chain of twelve)
  [[maybe_unused]] const uint64_t* slot_base[LTOT];
  if constexpr (PAGED) {
    pg = reinterpret_cast<const ChunkPage*>(pa->pg.slot_pages[0])[blockIdx.y];
#pragma unroll
    for (int i = 0; i < LTOT; ++i) slot_base[i] = reinterpret_cast<const ChunkPage*>(pa->pg.slot_pages[i])[blockIdx.y].data;
    n_rows = pg.n_data;
    win = bitmap_window(bitmap32, pg, pa->pg.chunk_rows, pa->pg.done, pa->pg.edges);
  }
  int64_t tiles = (n_rows + kRowsPerTile - 1) / kRowsPerTile;
  int64_t stride = (int64_t)gridDim.x * kWavesPerBlock;
  const int64_t bm_dwords = bitmap_dwords(n_rows);
  const int n_ops = a->n_ops;
  const uint32_t lane_byte = (uint32_t)lane * 16u;
  int64_t tile = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  [[maybe_unused]] WindowCarry carry;
  if constexpr (PAGED) {
    const TileShare sh = tile_share(win, tiles, tile, stride);
    tile = sh.first;
    stride = sh.step;
    tiles = sh.end;
  }

  u32x4 r[LTOT];
  // slot i of stripe t: chunks beyond the sub-tile's 16 w take an offset outside every column (zeros, no
  // traffic); bytes beyond the column's end are outside the resource
  auto load_slot = [&](int i, int64_t t) {
    const uint32_t first = a->slots[i].first_byte, tile_bytes = a->slots[i].tile_bytes;
    const uint32_t in_tile = first + lane_byte;
    const uint32_t off = in_tile < tile_bytes ? (uint32_t)t * tile_bytes + in_tile : 0xFFFFFFF0u;
    if constexpr (PAGED) {  // over the page's bytes in this operand's chunk
      const int64_t bytes = ((n_rows + 63) / 64) * (int64_t)(tile_bytes >> 5);
      r[i] = buffer_load16<true>(__builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t*>(slot_base[i]), 0,
                                                                   tile_bytes ? (int)bytes : 0, kBufferRsrcDword3), off);
    } else {
      r[i] = buffer_load16<true>(host_rsrc(a->slots[i].rsrc[0], a->slots[i].rsrc[1], a->slots[i].rsrc[2], a->slots[i].rsrc[3]), off);
    }
  };
  auto stage_slot = [&](int i) {
    const uint32_t first = a->slots[i].first_byte, tile_bytes = a->slots[i].tile_bytes;
    const uint32_t in_tile = first + lane_byte;
    if (in_tile < tile_bytes) {
      const uint32_t w = tile_bytes >> 8;
      const uint32_t wi = in_tile >> 3;  // first of the chunk's two words
      const uint32_t blk = __umulhi(wi, a->slots[i].inv_w);
      uint32_t* dst = lds32 + a->slots[i].img_dw + 2 * (blk * (w | 1u) + (wi - blk * w));
      const u32x2 lo = {r[i].x, r[i].y}, hi = {r[i].z, r[i].w};
      *reinterpret_cast<u32x2*>(dst) = lo;
      *reinterpret_cast<u32x2*>(dst + 2) = hi;
    }
  };
  // (slots beyond n_slots have tile_bytes = 0: no lane stages them, their loads are out of range -- no
  // wave-uniform branch per slot, so the descriptors' scalar loads are batched)
  if (tile < tiles) {
#pragma unroll
    for (int i = 0; i < LTOT; ++i) load_slot(i, tile);
  }
  while (tile < tiles) {
    const int64_t next = tile + stride;
    // all slots are staged before the first load of the next stripe is issued: a load in between would
    // sit in front of the older ones in vmcnt's order and every staging step would wait for it
#pragma unroll
    for (int i = 0; i < LTOT; ++i) stage_slot(i);
    if (next < tiles) {  // the slot registers are free: the whole next stripe
#pragma unroll
      for (int i = 0; i < LTOT; ++i) load_slot(i, next);
    }
    wave_lds_fence();
    uint32_t acc = 0u;
#pragma unroll 1
    for (int i = 0; i < n_ops; ++i) {
      const ChainStep o = chain_step(a, i);
      uint32_t sel = 0u;
      width_switch<MAXW>(o.w, [&](auto W) { sel = chain_eval<decltype(W)::value>(lds32 + o.img_dw, lane, o); });
      acc = o.combine == 0 ? sel : o.combine == 1 ? (acc & sel) : (acc | sel);
    }
    const uint32_t bm = finish_bitmap_dword(acc, tile, lane, n_rows);
    const int64_t d = tile * 64 + lane;
    if constexpr (PAGED) {
      window_emit(win, carry, d, bm, 0);
    } else if (d < bm_dwords) {
      IPS_BITMAP_STORE(bitmap32 + d, bm);
    }
    wave_lds_fence();  // the images are rewritten by the next stripe
    tile = next;
  }
  if constexpr (PAGED) {
    window_flush(win, carry, 0);
    page_done(pa->pg.done, pa->pg.done_page0, pa->pg.done_epoch);
  }
}

// (the kernels' first parameter is read through the kernarg segment pointer: it sits at offset 0)
template <int LTOT, int MAXW>
__global__ __launch_bounds__(kThreads, chain_min_waves(LTOT, MAXW)) void fle_chain_w_kernel(
    ChainArgsW a_by_value, int64_t n_rows, uint32_t* __restrict__ bitmap32) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_all[];
  (void)a_by_value;
  chain_body<LTOT, MAXW, false>((ChainArgsK)__builtin_amdgcn_kernarg_segment_ptr(), nullptr, n_rows, bitmap32, lds_all);
}

template <int LTOT, int MAXW>
__global__ __launch_bounds__(kThreads, chain_min_waves(LTOT, MAXW)) void fle_chain_w_pages_kernel(
    ChainPagedArgsW a_by_value, uint32_t* __restrict__ bitmap32) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_all[];
  (void)a_by_value;
  const ChainPagedArgsK pa = (ChainPagedArgsK)__builtin_amdgcn_kernarg_segment_ptr();
  chain_body<LTOT, MAXW, true>((ChainArgsK)pa, pa, 0, bitmap32, lds_all);
}

// ---------------------------------------------------------------------------------------------
// The chain over chunks whose pages end at DIFFERENT rows -- the usual case: every column writer cuts
// its pages by its own byte count, and EvalSimplePredicates cuts its batches at every column's page
// end (hdfs-parquet-scanner.cc:1837-1855).  The page starts of all operands cut the chunk's rows into
// SEGMENTS inside which every operand stays in one page.  A segment starts at an arbitrary row r of
// an operand's page: block r / 64, bit r % 64.  A stripe is therefore 62 bitmap dwords = 1984 rows:
// whatever the bit offset (0..63), the 32 blocks from the stripe's first block on hold all of its rows,
// so every operand still takes ONE ordinary 32-block sub-tile per stripe (consecutive stripes re-read one
// block), stripes stay independent -- one per wave, any order -- and the operand's result dwords are
// shifted into place with two DPP moves and a funnel shift.  The stripe's 62 dwords leave through the
// segment's window into the chunk-wide bitmap (a segment is a "page" of the bitmap: ips_chunk_device.h;
// segments that start inside a bitmap dword use edge slots + the fix-up launch).
// chain_segments_kernel (one workgroup) merges the page starts on the device, so the call stays
// asynchronous and allocation-free: ranks by binary search in an LDS copy of the starts; it also deals the
// chain's workgroups (four stripes each) to the segments, so that no workgroup is without work.
// Measured on the Q6 shape (pages of 2^20 / 2^20 - 37 / 700 001 rows: 2002 segments of 37 .. 700 001 rows):
// the chain kernel 345 us + the merge 50 us + the fix-up 10 us = 397-405 us against 367 us for the three
// per-operand launches (417-429 us before their 62-dword stripes) -- which is why only
// IPS_PROGRAM_ONE_PASS selects it.  The
// wave of a stripe sits behind three dependent rounds of loads (its segment's number, the segment's tables, the
// data; the contiguous chain: one, the chain over common pages: two).  Tried: blockIdx.y = segment with a
// looping wave (41 % of the wave slots idle, segments of 1 to 353 stripes: 352 us), a grid sized for the longest
// segment (500 us: the workgroups without work), all operands' images resident (6 waves per SIMD: 352 us; now one
// image per wave, 8 waves: 345 us).
// ---------------------------------------------------------------------------------------------
typedef const ChainSegmentedArgsW __attribute__((address_space(4))) * ChainSegArgsK;

// upper bound of the chain's workgroups (four stripes each): every segment wastes less than one
static int64_t chain_seg_max_wgs(int n_bounds, int64_t chunk_rows) {
  return (chunk_rows / kChainSegRows + n_bounds) / kWavesPerBlock + n_bounds + 1;
}

__global__ __launch_bounds__(1024) void chain_segments_kernel(ChainSegArgs sg, int n_ops, ChunkPage* __restrict__ seg_pages,
                                                              ChainSegSlot* __restrict__ seg_slots,
                                                              uint32_t* __restrict__ seg_bits, uint32_t* __restrict__ wg_seg,
                                                              uint32_t* __restrict__ seg_wg0, uint32_t* __restrict__ n_wgs) {
  __shared__ int64_t starts[kChainSegMaxBounds];   // operand after operand
  __shared__ int64_t merged[kChainSegMaxBounds];
  int first[kChainWMaxOps + 1];
  first[0] = 0;
#pragma unroll
  for (int i = 0; i < kChainWMaxOps; ++i) first[i + 1] = first[i] + (i < n_ops ? sg.op_n_pages[i] : 0);
  const int n_bounds = first[kChainWMaxOps];
#pragma unroll
  for (int i = 0; i < kChainWMaxOps; ++i) {
    if (i < n_ops) {
      const ChunkPage* pages = reinterpret_cast<const ChunkPage*>(sg.op_pages[i]);
      for (int p = threadIdx.x; p < sg.op_n_pages[i]; p += blockDim.x) starts[first[i] + p] = pages[p].row0;
    }
  }
  __syncthreads();
  // entries of operand j's list below x (strict) or not above it
  auto count = [&](int j, int64_t x, bool or_equal) -> int {
    int lo = first[j], hi = first[j + 1];
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      const int64_t v = starts[mid];
      if (v < x || (or_equal && v == x)) lo = mid + 1; else hi = mid;
    }
    return lo - first[j];
  };
  for (int g = threadIdx.x; g < n_bounds; g += blockDim.x) {
    int op = 0;
#pragma unroll
    for (int i = 1; i < kChainWMaxOps; ++i) op += (i < n_ops && g >= first[i]) ? 1 : 0;
    const int64_t x = starts[g];
    int rank = g - first[op];  // (an operand's own starts are strictly increasing)
#pragma unroll
    for (int j = 0; j < kChainWMaxOps; ++j)
      if (j < n_ops && j != op) rank += count(j, x, j < op);  // ties: the lower operand first
    merged[rank] = x;
  }
  __syncthreads();
  for (int r = threadIdx.x; r < n_bounds; r += blockDim.x) {
    const int64_t start = merged[r];
    const int64_t end = r + 1 < n_bounds ? merged[r + 1] : sg.chunk_rows;
    ChunkPage sp;
    sp.data = nullptr;
    sp.levels = nullptr;
    sp.n_rows = end - start;
    sp.n_data = end - start;
    sp.row0 = start;
    sp.batch0 = (uint32_t)(start / kChainSegRows + r);  // first edge slot: unique and increasing over the segments
    sp.flags = r + 1 == n_bounds ? kPageLast : 0u;
    sp.rank0 = 0u;
    sp.reserved = 0u;
    seg_pages[r] = sp;
    // where the segment starts inside every operand's page
#pragma unroll
    for (int j = 0; j < kChainWMaxOps; ++j) {
      ChainSegSlot sl;
      sl.base = nullptr;
      sl.blocks = 0;
      uint32_t bit = 0u;
      if (j < n_ops) {
        const ChunkPage* pg = reinterpret_cast<const ChunkPage*>(sg.op_pages[j]) + (count(j, start, true) - 1);
        const int64_t in_page = start - pg->row0;
        const int64_t b0 = in_page >> 6;
        sl.base = pg->data + b0 * sg.op_w[j];
        sl.blocks = (pg->n_rows + 63) / 64 - b0;
        bit = (uint32_t)(in_page & 63);
      }
      seg_slots[(size_t)r * 8 + j] = sl;
      seg_bits[r * 8 + j] = bit;
    }
  }
  // the chain's workgroups: ceil(stripes / 4) per segment, dealt in segment order (exclusive prefix sums over
  // the segments: four consecutive segments per thread, then a scan of the 1024 partial sums)
  __syncthreads();  // (the starts are not needed any more: their array holds the scan)
  uint32_t* scan = reinterpret_cast<uint32_t*>(starts);
  auto wgs_of = [&](int r) -> uint32_t {
    if (r >= n_bounds) return 0u;
    const int64_t rows = (r + 1 < n_bounds ? merged[r + 1] : sg.chunk_rows) - merged[r];
    const int64_t stripes = (rows + kChainSegRows - 1) / kChainSegRows;
    return (uint32_t)((stripes + kWavesPerBlock - 1) / kWavesPerBlock);
  };
  const int r0 = (int)threadIdx.x * 4;
  const uint32_t c0 = wgs_of(r0), c1 = wgs_of(r0 + 1), c2 = wgs_of(r0 + 2), c3 = wgs_of(r0 + 3);
  scan[threadIdx.x] = c0 + c1 + c2 + c3;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const uint32_t add = (int)threadIdx.x >= d ? scan[threadIdx.x - d] : 0u;
    __syncthreads();
    scan[threadIdx.x] += add;
    __syncthreads();
  }
  uint32_t at = scan[threadIdx.x] - (c0 + c1 + c2 + c3);  // workgroups in front of segment r0
  const uint32_t cnt[4] = {c0, c1, c2, c3};
  uint32_t* first_wg = scan + 1024;   // [segment]: its first workgroup and its count, for the fill below
  uint32_t* n_of_seg = scan + 1024 + kChainSegMaxBounds;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (r0 + k < n_bounds) {
      seg_wg0[r0 + k] = at;
      first_wg[r0 + k] = at;
      n_of_seg[r0 + k] = cnt[k];
      at += cnt[k];
    }
  }
  if (threadIdx.x == 1023) *n_wgs = scan[1023];
  __syncthreads();
  // a wave per segment, a lane per workgroup: coalesced stores (a thread per segment wrote up to 89 entries
  // one after the other: 20 us of the kernel)
  const int lane = (int)(threadIdx.x & 63), wv = (int)(threadIdx.x >> 6);
  for (int r = wv; r < n_bounds; r += 16) {
    const uint32_t w0 = first_wg[r], cn = n_of_seg[r];
    for (uint32_t g = (uint32_t)lane; g < cn; g += 64u) wg_seg[w0 + g] = (uint32_t)r;
  }
}

template <int LTOT, int MAXW>
__global__ __launch_bounds__(kThreads, chain_min_waves(LTOT, MAXW)) void fle_chain_w_segments_kernel(
    ChainSegmentedArgsW a_by_value, uint32_t* __restrict__ bitmap32) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_all[];
  (void)a_by_value;
  const ChainSegArgsK sa = (ChainSegArgsK)__builtin_amdgcn_kernarg_segment_ptr();
  const ChainArgsK a = (ChainArgsK)sa;
  if (blockIdx.x >= *sa->sg.n_wgs) return;  // (the grid is the host's upper bound)
  const uint32_t si = sa->sg.wg_seg[blockIdx.x];
  const ChunkPage seg = reinterpret_cast<const ChunkPage*>(sa->sg.seg_pages)[si];
  const int lane = lane_id();
  const int wave = wave_id();
  // one stripe per wave, four consecutive ones per workgroup
  const int64_t t = (int64_t)(blockIdx.x - sa->sg.seg_wg0[si]) * kWavesPerBlock + wave;
  if (t * kChainSegRows >= seg.n_rows) return;
  uint32_t* lds32 = lds_all + wave * sa->sg.image_dwords;
  const BitmapWindow win = bitmap_window(bitmap32, seg, sa->sg.chunk_rows, nullptr, sa->sg.edges);
  const int n_ops = a->n_ops;
  const uint32_t lane_byte = (uint32_t)lane * 16u;
  // per operand: the first block of the segment in its page, the blocks the page holds from there on and the rows
  // of that block in front of the segment (prepared by chain_segments_kernel: nothing here depends on another load
  // but the segment's number)
  const ChainSegSlot* slots_of_seg = sa->sg.seg_slots + (size_t)si * 8;
  const uint32_t* bits_of_seg = sa->sg.seg_bits + (size_t)si * 8;
  uint32_t op_bit[kChainWMaxOps];
#pragma unroll
  for (int i = 0; i < kChainWMaxOps; ++i) op_bit[i] = bits_of_seg[i];
  u32x4 r[LTOT];
  // slot i: the 32 blocks from block 31 t of the segment on (the page's bytes bound the loads)
#pragma unroll
  for (int i = 0; i < LTOT; ++i) {
    const uint32_t first = a->slots[i].first_byte, tile_bytes = a->slots[i].tile_bytes;
    const uint32_t w = tile_bytes >> 8;
    const ChainSegSlot sl = slots_of_seg[sa->sg.slot_op[i]];
    int64_t left = (sl.blocks - 31 * t) * (int64_t)w * 8;
    left = left < 0 ? 0 : (left > (int64_t)tile_bytes ? (int64_t)tile_bytes : left);
    const uint32_t in_tile = first + lane_byte;
    r[i] = buffer_load16<true>(__builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t*>(sl.base + 31 * t * (int64_t)w), 0,
                                                                 tile_bytes ? (int)left : 0, kBufferRsrcDword3),
                               in_tile < tile_bytes ? in_tile : 0xFFFFFFF0u);
  }
  // ONE plane image per wave, the widest operand's: the operands are staged from the slot registers one after the
  // other (the LDS a wave holds bounds the resident waves, and a stripe's wave lives through three rounds of loads)
  uint32_t acc = 0u;
#pragma unroll 1
  for (int i = 0; i < n_ops; ++i) {
    const ChainStep o = chain_step(a, i);
#pragma unroll
    for (int k = 0; k < LTOT; ++k) {
      if (sa->sg.slot_op[k] == i) {  // wave-uniform
        const uint32_t first = a->slots[k].first_byte, tile_bytes = a->slots[k].tile_bytes;
        const uint32_t in_tile = first + lane_byte;
        if (in_tile < tile_bytes) {
          const uint32_t w = tile_bytes >> 8;
          const uint32_t wi = in_tile >> 3;
          const uint32_t blk = __umulhi(wi, a->slots[k].inv_w);
          uint32_t* dst = lds32 + 2 * (blk * (w | 1u) + (wi - blk * w));
          const u32x2 lo = {r[k].x, r[k].y}, hi = {r[k].z, r[k].w};
          *reinterpret_cast<u32x2*>(dst) = lo;
          *reinterpret_cast<u32x2*>(dst + 2) = hi;
        }
      }
    }
    wave_lds_fence();
    uint32_t sel = 0u;
    width_switch<MAXW>(o.w, [&](auto W) { sel = chain_eval<decltype(W)::value>(lds32, lane, o); });
    wave_lds_fence();  // the image is rewritten by the next operand
    // rows 32 l .. 32 l + 31 of the sub-tile -> rows of the stripe: the segment starts 'bit' rows into the block
    uint32_t bit = op_bit[0];
#pragma unroll
    for (int j = 1; j < kChainWMaxOps; ++j) bit = i == j ? op_bit[j] : bit;  // (wave-uniform selects)
    uint32_t lo = bitrev32(sel);
    if (bit & 32u) lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x130, 0xF, 0xF, true);  // wave_shl:1
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x130, 0xF, 0xF, true);
    sel = __builtin_amdgcn_alignbit(hi, lo, bit & 31u);
    acc = o.combine == 0 ? sel : o.combine == 1 ? (acc & sel) : (acc | sel);
  }
  WindowCarry carry;
  window_emit(win, carry, t * kChainSegRunDwords + lane, acc, 0, kChainSegRunDwords - 1, false, 0u, t);
  window_flush(win, carry, 0);
}

// resident workgroups per CU of a kernel with 'lds' bytes of dynamic image (not known to the occupancy cache
// of grid_for_tiles)
static int chain_resident(const void* kern, size_t lds) {
  static std::mutex mu;
  static std::map<std::pair<const void*, size_t>, int> resident;
  const auto key = std::make_pair(kern, lds);
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = resident.find(key);
    if (it != resident.end()) return it->second;
  }
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, kThreads, lds) != hipSuccess || per_cu <= 0) per_cu = 2;
  std::lock_guard<std::mutex> lk(mu);
  resident[key] = per_cu;
  return per_cu;
}

template <int LTOT, int MAXW>
static ips_status launch_chain_w_class(const ChainArgsW& a, int64_t n_rows, uint32_t* bitmap32, hipStream_t s) {
  const int64_t tiles = (n_rows + kRowsPerTile - 1) / kRowsPerTile;
  auto kern = fle_chain_w_kernel<LTOT, MAXW>;
  const size_t lds = (size_t)kWavesPerBlock * a.image_dwords * 4;
  const int per_cu = chain_resident(reinterpret_cast<const void*>(kern), lds);
  const int64_t want = (tiles + kWavesPerBlock - 1) / kWavesPerBlock;
  const int64_t cap = (int64_t)device_cus() * per_cu * grid_mult(kGridChain);
  const int64_t rounds = (want + cap - 1) / cap;
  const int grid = (int)(want <= cap ? want : (want + rounds - 1) / rounds);
  if (grid <= 0) return IPS_ERR_HIP;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, s, a, n_rows, bitmap32);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

template <int LTOT, int MAXW>
static ips_status launch_chain_w_pages_class(const ChainPagedArgsW& pa, int n_pages, int64_t max_rows,
                                             uint32_t* bitmap32, hipStream_t s) {
  const int64_t tiles = (max_rows + kRowsPerTile - 1) / kRowsPerTile;  // of the largest page
  auto kern = fle_chain_w_pages_kernel<LTOT, MAXW>;
  const size_t lds = (size_t)kWavesPerBlock * pa.chain.image_dwords * 4;
  const int per_cu = chain_resident(reinterpret_cast<const void*>(kern), lds);
  // x: shares of one page's stripes, so that x * pages fills the device about grid_mult times
  const int64_t total = (int64_t)device_cus() * per_cu * grid_mult(kGridChain);
  int64_t gx = (total + n_pages - 1) / n_pages;
  const int64_t gx_max = (tiles + kWavesPerBlock - 1) / kWavesPerBlock;
  gx = gx > gx_max ? gx_max : gx;
  gx = gx < 1 ? 1 : gx;
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)n_pages), dim3(kThreads), lds, s, pa, bitmap32);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

// slots, images and plane masks of a chain; enc = NULL: paged (the slot carries its operand, the resource
// is made per page)
static ips_status chain_layout(ChainArgsW& a, const void* const* enc, int64_t n_blocks, int* max_w) {
  int slots = 0, img = 0, maxw = 0;
  for (int i = 0; i < a.n_ops; ++i) {
    const int w = a.ops[i].w;
    maxw = w > maxw ? w : maxw;
    const int64_t bytes = n_blocks * w * 8;
    // (32-bit offsets: the last sub-tile's chunks beyond the column must stay below 2^32 to be range-checked away)
    if (bytes >= 0xFFFF0000ll) return IPS_ERR_UNSUPPORTED;
    a.ops[i].img_dw = img;
    for (int k = 0; k < 16; ++k) {
      a.ops[i].m1[k] = ((a.ops[i].c1 >> k) & 1u) ? ~0u : 0u;
      a.ops[i].m2[k] = ((a.ops[i].c2 >> k) & 1u) ? ~0u : 0u;
    }
    for (int c = 0; c < 16 * w; c += kWave) {
      if (slots == kChainWMaxSlots) return IPS_ERR_UNSUPPORTED;
      ChainSlot& sl = a.slots[slots++];
      if (enc) {
        const uint64_t base = reinterpret_cast<uint64_t>(enc[i]);
        sl.rsrc[0] = (uint32_t)base;
        sl.rsrc[1] = (uint32_t)(base >> 32) & 0xFFFFu;
        sl.rsrc[2] = (uint32_t)bytes;
        sl.rsrc[3] = kBufferRsrcDword3;
      } else {
        sl.rsrc[0] = (uint32_t)i;
      }
      sl.first_byte = (uint32_t)c * 16u;
      sl.tile_bytes = 256u * (uint32_t)w;
      sl.inv_w = (uint32_t)(0x100000000ull / (uint64_t)w) + 1u;
      sl.img_dw = img;
    }
    img += plane_tile_bytes(w) / 4;
  }
  a.n_slots = slots;
  a.image_dwords = img;
  if ((size_t)kWavesPerBlock * img * 4 > 64 * 1024) return IPS_ERR_UNSUPPORTED;
  *max_w = maxw;
  return IPS_OK;
}

template <int LTOT, int MAXW>
static ips_status launch_chain_w_segments_class(const ChainSegmentedArgsW& sa, int64_t chunk_rows, uint32_t* bitmap32,
                                                hipStream_t s) {
  auto kern = fle_chain_w_segments_kernel<LTOT, MAXW>;
  const size_t lds = (size_t)kWavesPerBlock * sa.sg.image_dwords * 4;
  hipLaunchKernelGGL(kern, dim3((unsigned)chain_seg_max_wgs(sa.sg.n_bounds, chunk_rows)), dim3(kThreads), lds, s, sa, bitmap32);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

#define IPS_CHAIN_DISPATCH(FN, SLOTS, MAXW, ...)                                              \
  do {                                                                                        \
    if ((SLOTS) <= 2) return (MAXW) <= 16 ? FN<2, 16>(__VA_ARGS__) : FN<2, 32>(__VA_ARGS__);   \
    if ((SLOTS) <= 4) return (MAXW) <= 16 ? FN<4, 16>(__VA_ARGS__) : FN<4, 32>(__VA_ARGS__);   \
    if ((SLOTS) <= 6) return (MAXW) <= 16 ? FN<6, 16>(__VA_ARGS__) : FN<6, 32>(__VA_ARGS__);   \
    if ((SLOTS) <= 8) return (MAXW) <= 16 ? FN<8, 16>(__VA_ARGS__) : FN<8, 32>(__VA_ARGS__);   \
    if ((SLOTS) <= 12) return (MAXW) <= 16 ? FN<12, 16>(__VA_ARGS__) : FN<12, 32>(__VA_ARGS__); \
    return (MAXW) <= 16 ? FN<16, 16>(__VA_ARGS__) : FN<16, 32>(__VA_ARGS__);                   \
  } while (0)

// a.ops[0..n_ops) filled by the caller (w, kind, operators, constants, combine); enc[i] = operand i's column
ips_status launch_chain_w(ChainArgsW& a, const void* const* enc, int64_t n_rows, uint32_t* bitmap32, hipStream_t s) {
  int maxw = 0;
  const ips_status st = chain_layout(a, enc, (n_rows + 63) / 64, &maxw);
  if (st != IPS_OK) return st;
  IPS_CHAIN_DISPATCH(launch_chain_w_class, a.n_slots, maxw, a, n_rows, bitmap32, s);
}

ips_status launch_chain_w_pages(ChainPagedArgsW& pa, const void* const* op_pages, int n_pages, int64_t max_rows,
                                uint32_t* bitmap32, hipStream_t s) {
  int maxw = 0;
  const ips_status st = chain_layout(pa.chain, nullptr, (max_rows + 63) / 64, &maxw);
  if (st != IPS_OK) return st;
  for (int i = 0; i < kChainWMaxSlots; ++i)
    pa.pg.slot_pages[i] = op_pages[i < pa.chain.n_slots ? (int)pa.chain.slots[i].rsrc[0] : 0];
  IPS_CHAIN_DISPATCH(launch_chain_w_pages_class, pa.chain.n_slots, maxw, pa, n_pages, max_rows, bitmap32, s);
}

namespace {
struct ChainSegWs {
  ChunkPage* seg_pages;
  ChainSegSlot* seg_slots;
  uint32_t* seg_bits;
  uint32_t* wg_seg;
  uint32_t* seg_wg0;
  uint32_t* n_wgs;
  uint32_t* edges;
  size_t total;
};
size_t seg_align(size_t x) { return (x + 255) & ~(size_t)255; }
ChainSegWs chain_seg_ws(void* base, int n_bounds, int64_t chunk_rows) {
  uint8_t* p = reinterpret_cast<uint8_t*>(base);
  ChainSegWs w;
  size_t off = 0;
  w.seg_pages = reinterpret_cast<ChunkPage*>(p + off); off += seg_align((size_t)n_bounds * sizeof(ChunkPage));
  w.seg_slots = reinterpret_cast<ChainSegSlot*>(p + off); off += seg_align((size_t)n_bounds * 8 * sizeof(ChainSegSlot));
  w.seg_bits = reinterpret_cast<uint32_t*>(p + off); off += seg_align((size_t)n_bounds * 8 * 4);
  w.wg_seg = reinterpret_cast<uint32_t*>(p + off); off += seg_align((size_t)chain_seg_max_wgs(n_bounds, chunk_rows) * 4);
  w.seg_wg0 = reinterpret_cast<uint32_t*>(p + off); off += seg_align((size_t)n_bounds * 4);
  w.n_wgs = reinterpret_cast<uint32_t*>(p + off); off += 256;
  // edge slots: 4 dwords per stripe; slot of a stripe = first row / 1984 + segment index (+ stripe in the segment)
  w.edges = reinterpret_cast<uint32_t*>(p + off); off += seg_align(((size_t)(chunk_rows / kChainSegRows) + (size_t)n_bounds + 4) * 16);
  w.total = off;
  return w;
}
}  // namespace

size_t chain_segments_workspace_bytes(int n_bounds, int64_t chunk_rows) {
  return chain_seg_ws(nullptr, n_bounds, chunk_rows).total;
}

ips_status launch_chain_w_segments(ChainSegmentedArgsW& sa, const void* const* op_pages, const int* op_n_pages,
                                   int64_t max_rows, int64_t chunk_rows, uint32_t* bitmap32, void* workspace,
                                   hipStream_t s) {
  int maxw = 0, n_bounds = 0;
  const ips_status st = chain_layout(sa.chain, nullptr, (max_rows + 63) / 64 + 64, &maxw);
  if (st != IPS_OK) return st;
  for (int i = 0; i < sa.chain.n_ops; ++i) n_bounds += op_n_pages[i];
  if (n_bounds <= 0 || n_bounds > kChainSegMaxBounds) return IPS_ERR_UNSUPPORTED;
  const ChainSegWs w = chain_seg_ws(workspace, n_bounds, chunk_rows);
  for (int i = 0; i < kChainWMaxOps; ++i) {
    sa.sg.op_pages[i] = op_pages[i < sa.chain.n_ops ? i : 0];
    sa.sg.op_n_pages[i] = i < sa.chain.n_ops ? op_n_pages[i] : 0;
    sa.sg.op_w[i] = i < sa.chain.n_ops ? sa.chain.ops[i].w : 0;
  }
  sa.sg.image_dwords = plane_tile_bytes(maxw) / 4;  // one image per wave, the widest operand's
  for (int i = 0; i < kChainWMaxSlots; ++i) {
    sa.sg.slot_op[i] = i < sa.chain.n_slots ? (int)sa.chain.slots[i].rsrc[0] : 0;
  }
  sa.sg.seg_pages = w.seg_pages;
  sa.sg.seg_slots = w.seg_slots;
  sa.sg.seg_bits = w.seg_bits;
  sa.sg.wg_seg = w.wg_seg;
  sa.sg.seg_wg0 = w.seg_wg0;
  sa.sg.n_wgs = w.n_wgs;
  sa.sg.chunk_rows = chunk_rows;
  sa.sg.edges = w.edges;
  sa.sg.n_bounds = n_bounds;
  hipLaunchKernelGGL(chain_segments_kernel, dim3(1), dim3(1024), 0, s, sa.sg, sa.chain.n_ops, w.seg_pages, w.seg_slots,
                     w.seg_bits, w.wg_seg, w.seg_wg0, w.n_wgs);
  IPS_HIP_TRY(hipGetLastError());
  auto launch = [&]() -> ips_status {
    IPS_CHAIN_DISPATCH(launch_chain_w_segments_class, sa.chain.n_slots, maxw, sa, chunk_rows, bitmap32, s);
  };
  const ips_status st2 = launch();
  if (st2 != IPS_OK) return st2;
  return launch_window_fixup(w.seg_pages, n_bounds, max_rows, chunk_rows, bitmap32, w.edges, 0, s, kChainSegRunDwords);
}

}  // namespace ips
