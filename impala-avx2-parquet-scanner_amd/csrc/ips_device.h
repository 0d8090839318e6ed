// ips_device.h -- device-side building blocks shared by the FLE kernels (gfx950 only).
//
// Execution model ("one wavefront per row-group stripe"): a wave owns whole SUB-TILES of 32 FLE
// blocks (2048 rows = IPS_BATCH_ROWS) and never synchronises with another wave; a workgroup is
// just 4 such waves sharing a CU.  Per sub-tile a wave
//   1. pulls the encoded bytes HBM -> VGPR with 16-byte coalesced loads (the next sub-tile's
//      loads are issued before the current one is consumed: register prefetch),
//   2. scatters them into its private LDS region in a bank-conflict-free (odd stride) layout,
//   3. re-reads them lane-per-half-block (lane l <-> block l/2, rows 32*(l&1)..+31),
//   4. runs the predicate recurrence / the 32x32 bit transposes in registers,
//   5. exchanges results through the same LDS region to get coalesced, row-ordered stores.
// LDS operations of one wave execute in order, so steps 2->3 and 4->5 need only a compiler-level
// wavefront fence, never s_barrier.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ips_bitops.h"
#include "ips_knobs.h"

namespace ips {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kThreads = kWave * kWavesPerBlock;

// what a fused scan kernel evaluates (ips_fle_kernels.h)
enum ScanMode { kScanPredicate = 0, kScanGivenBitmap = 1, kScanInList = 2, kScanInTable = 3 };

// Predicate parameters travel in the kernarg segment: wave-uniform, read with scalar loads.
struct PredArgs {
  int32_t op;        // ips_op
  int32_t n_consts;  // 1, or 1..256 for IN
  // predicate-only kernels (used by the per-operand plan of ips_eval_program):
  int32_t combine;   // 0: bitmap = result; 1: bitmap &= result; 2: bitmap |= result
  int32_t join;      // 0: single predicate; 1 / 2: result = pred(op, consts[0]) AND / OR
  int32_t op2;       //    pred(op2, const2), both evaluated in ONE pass over the planes
  uint32_t const2;   //    (BETWEEN = Ge AND Le arrives this way)
  // tile counts of a NOT-NULL root that ride on this launch (nullable leaf, ips_rank_device.h):
  // workgroups [0, aux_blocks) count tile blockIdx.x of aux_root and exit, the predicate runs on
  // the workgroups behind them.  aux_blocks = 0: none.
  int32_t aux_blocks;
  int32_t aux_kind;  // RootKind
  const void* aux_root;
  int64_t aux_rows;
  uint32_t* aux_counts;
  // paged launches of a sharded step (ips_comm.hip): every wave adds one to done[blockIdx.y] when the
  // results of its share of page blockIdx.y are visible device-wide; NULL = nobody is waiting
  uint32_t* done;
  int32_t done_page0;  // page index of blockIdx.y == 0 in 'done' (the first page of the launch's run)
  uint32_t done_epoch; // the value a complete page's flag takes in this step
  // IN over an ips_inset (any number of members): the 65536-bit membership table of the members
  // < 2^16 (codes of up to 16 bits look themselves up) and, for wider columns, the ascending member
  // list, of which the first in_list_n fit the column's width.  NULL: the list in consts[]
  const uint32_t* in_table;
  const uint32_t* in_list;
  int32_t in_list_n;
  int32_t reserved2;
  // paged launches over a chunk with pages that start inside a bitmap dword: the two end dwords of
  // every sub-tile go to the chunk's edge slots and a fix-up launch merges them (ips_chunk_device.h)
  uint32_t* edges;
  uint32_t consts[256];
};

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
// wave index inside the workgroup, as a scalar (SGPR) value so that everything derived from it
// (tile numbers, bounds, branches) stays on the scalar unit
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }

// 0 or ~0 from bit k of c (one s_bfe_i32)
__device__ __forceinline__ uint32_t bit_mask(uint32_t c, int k) {
  return (uint32_t)(((int32_t)(c << (31 - k))) >> 31);
}

// ---- HBM -> VGPR ---------------------------------------------------------------------------
// A full sub-tile is 32*W words = 16*W chunks of 16 bytes; lane l takes chunks l, l+64, ...
// MAXLOADS = ceil(16*W/64).  'total_words' bounds the encoded buffer: nothing beyond
// ceil(n/64)*W words is ever read (the reference's unpackers over-read 32 bytes, quirk Q5).
// 16-byte load of data that is read exactly once (column pages): the nt hint keeps the stream from
// displacing the kernel's own dirty output lines in L2, so those leave for HBM in larger bursts.
// Measured on the headline scan: 252 -> 224 us, predicate only: 204 -> 180 us (2^28 rows, w=32).
// (With plain stores, kernels whose output is as large as their input measured 4 % slower with it.)
template <bool NT = true>
__device__ __forceinline__ u32x4 stream_load(const u32x4* p) {
#ifdef IPS_NO_NT_LOADS
  return *p;
#else
  return NT ? __builtin_nontemporal_load(p) : *p;
#endif
}

// The bitmap a kernel produces is not read again by that kernel: storing it with the nt hint keeps
// it from occupying L2 next to the page stream (predicate kernels -5...-10 %, fused scan neutral).
#ifdef IPS_NO_NT_BITMAP_STORE
#define IPS_BITMAP_STORE(p, v) (*(p) = (v))
#else
#define IPS_BITMAP_STORE(p, v) __builtin_nontemporal_store((v), (p))
#endif

// Full 16-byte-per-lane output streams (decode, encode): written once, never re-read by the kernel
// (decode -2...-4 %, encode -3...-5 %).  Scattered 4-byte value stores of the scan are the opposite
// case: they need L2 to merge them, nt makes them 5-15 % slower.
#ifdef IPS_NO_NT_STREAM_STORE
#define IPS_STREAM_STORE16(p, v) (*reinterpret_cast<u32x4*>(p) = (v))
#else
#define IPS_STREAM_STORE16(p, v) __builtin_nontemporal_store((v), reinterpret_cast<u32x4*>(p))
#endif

// The sub-tile is read through a BUFFER resource that spans exactly its bytes inside the encoded
// buffer (base = first word of the sub-tile, num_records = min(256 W, bytes left)): the hardware's
// range check returns zeros for every dword outside it, so the last, partial sub-tile of a column,
// a sub-tile past the end and the lanes beyond chunk 16 W all take the same instructions as a
// whole sub-tile.  (The first version branched to a per-chunk bounds path for the last sub-tile:
// never executed twice, but its 64-bit addresses and conditions cost every FLE kernel about 34
// VGPRs of allocation -- fle_scan_kernel<32>: 150 -> 116 = 4 waves per SIMD instead of 3.)
constexpr unsigned kBufferRsrcDword3 = 0x00020000u;  // gfx9 family: DATA_FORMAT_32, raw (unswizzled) buffer

template <bool NT = true>
__device__ __forceinline__ u32x4 buffer_load16(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_offset) {
#ifdef IPS_NO_NT_LOADS
  return __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_offset, 0, 0);
#else
  return __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_offset, 0, NT ? 2 : 0);  // aux bit 1: nt
#endif
}

// resource over the bytes of sub-tile 'tile' that exist (none: an empty resource, every load 0)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const uint64_t* __restrict__ enc, int64_t tile,
                                                            int w, int64_t total_words) {
  const int64_t w0 = tile * (int64_t)(kBlocksPerTile * w);
  int64_t left = total_words - w0;  // words of the buffer from the sub-tile's first word on
  left = left < 0 ? 0 : (left > kBlocksPerTile * w ? kBlocksPerTile * w : left);
  const uint64_t* base = left > 0 ? enc + w0 : enc;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t*>(base), 0, (int)(left * 8), kBufferRsrcDword3);
}

template <int MAXLOADS, bool NT = true>
__device__ __forceinline__ void tile_load(const uint64_t* __restrict__ enc, int64_t tile, int w,
                                          int64_t total_words, int lane, u32x4 (&r)[MAXLOADS]) {
  const __amdgpu_buffer_rsrc_t rsrc = tile_rsrc(enc, tile, w, total_words);
#pragma unroll
  for (int i = 0; i < MAXLOADS; ++i) r[i] = buffer_load16<NT>(rsrc, (uint32_t)(i * kWave + lane) * 16u);
}

// Late materialisation against a given bitmap touches only the blocks that hold a selected row --
// what FleDecoder::Get(val, skip) / Skip() do with pointer arithmetic (fle-encoding.h:344-402:
// whole blocks are stepped over, never unpacked).  need_blocks: bit 2b set <-> block b of the
// sub-tile is needed; lanes whose 16-byte chunk lies in unneeded blocks issue no load, so whole
// cache lines of such blocks are never fetched.  Registers of skipped chunks keep stale data: it
// only reaches lanes that have no selected row.
template <int MAXLOADS, int W>
__device__ __forceinline__ void tile_load_needed(const uint64_t* __restrict__ enc, int64_t tile,
                                                 int64_t total_words, int lane,
                                                 uint64_t need_blocks, u32x4 (&r)[MAXLOADS]) {
  const __amdgpu_buffer_rsrc_t rsrc = tile_rsrc(enc, tile, W, total_words);
#pragma unroll
  for (int i = 0; i < MAXLOADS; ++i) {
    const int c = i * kWave + lane;
    if (c < 16 * W) {
      const int b0 = (2 * c) / W, b1 = (2 * c + 1) / W;
      if (((need_blocks >> (2 * b0)) | (need_blocks >> (2 * b1))) & 1ull)
        r[i] = buffer_load16<true>(rsrc, (uint32_t)c * 16u);
    }
  }
}

// ---- VGPR -> LDS (odd word stride per block) ------------------------------------------------
// inv_w = floor(2^32 / w) + 1 when the caller has it (w changes at run time inside the kernel):
// word / w == umulhi(word, inv_w) exactly for word < 2^16; 0 = divide.
template <int MAXLOADS>
__device__ __forceinline__ void tile_to_lds(uint32_t* lds32, int w, int lane,
                                            const u32x4 (&r)[MAXLOADS], uint32_t inv_w = 0u) {
  const int chunks = 16 * w;
  const int stride = w | 1;
  if (w & 1) {  // stride == w: the LDS image is linear, one 16-byte store per chunk
#pragma unroll
    for (int i = 0; i < MAXLOADS; ++i) {
      int c = i * kWave + lane;
      if (c < chunks) *reinterpret_cast<u32x4*>(lds32 + 4 * c) = r[i];
    }
  } else {  // even w: both words of a chunk belong to one block; padded rows are only 8-aligned
#pragma unroll
    for (int i = 0; i < MAXLOADS; ++i) {
      int c = i * kWave + lane;
      if (c < chunks) {
        int wi = 2 * c;
        int blk = inv_w ? (int)__umulhi((uint32_t)wi, inv_w) : wi / w;
        int k = wi - blk * w;
        uint32_t* dst = lds32 + 2 * (blk * stride + k);
        u32x2 lo = {r[i].x, r[i].y}, hi = {r[i].z, r[i].w};
        *reinterpret_cast<u32x2*>(dst) = lo;
        *reinterpret_cast<u32x2*>(dst + 2) = hi;
      }
    }
  }
}

// dword index of plane k's half for this lane: lane l <-> block l>>1; q = l&1 selects rows
// 32q..32q+31, which live in the HIGH dword (q=0) or LOW dword (q=1) of each plane word.
__device__ __forceinline__ int plane_base_dw(int w, int lane) {
  return 2 * ((lane >> 1) * (w | 1)) + (1 - (lane & 1));
}

// ---- predicate on the encoded planes, streaming from LDS (any w, nothing kept in VGPRs) -----
// One v_bitop3_b32 per plane (ips_bitops.h: borrow_step / eq_step), planes taken LSB -> MSB.
__device__ __forceinline__ uint32_t pred_single_from_lds(const uint32_t* lds32, int w, int lane,
                                                         int op, uint32_t c) {
  const uint32_t* p = lds32 + plane_base_dw(w, lane);
  const bool is_eq = op == 0;  // wave-uniform
  uint32_t acc = is_eq ? ~0u : borrow_init(op);
  int k = 0;
  for (; k + 8 <= w; k += 8) {  // 8 independent LDS reads in flight per round trip
    uint32_t x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = p[2 * (k + e)];
    if (is_eq) {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc = eq_step(acc, x[e], bit_mask(c, k + e));
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc = borrow_step(acc, x[e], bit_mask(c, k + e));
    }
  }
  for (; k < w; ++k) {
    const uint32_t x = p[2 * k];
    acc = is_eq ? eq_step(acc, x, bit_mask(c, k)) : borrow_step(acc, x, bit_mask(c, k));
  }
  return is_eq ? acc : borrow_select(acc, op);
}

// Two comparisons against the same column in one pass over its planes (a BETWEEN): two chains,
// two ops per plane (an EQ member takes the generic per-plane select).
__device__ __forceinline__ void pred_pair_from_lds(const uint32_t* lds32, int w, int lane, int op1,
                                                   uint32_t c1, int op2, uint32_t c2,
                                                   uint32_t* r1, uint32_t* r2) {
  const uint32_t* p = lds32 + plane_base_dw(w, lane);
  const bool eq1 = op1 == 0, eq2 = op2 == 0;
  uint32_t a1 = eq1 ? ~0u : borrow_init(op1), a2 = eq2 ? ~0u : borrow_init(op2);
  if (!eq1 && !eq2) {
    int k = 0;
    for (; k + 4 <= w; k += 4) {
      const uint32_t x0 = p[2 * k], x1 = p[2 * k + 2], x2 = p[2 * k + 4], x3 = p[2 * k + 6];
      a1 = borrow_step(a1, x0, bit_mask(c1, k));     a2 = borrow_step(a2, x0, bit_mask(c2, k));
      a1 = borrow_step(a1, x1, bit_mask(c1, k + 1)); a2 = borrow_step(a2, x1, bit_mask(c2, k + 1));
      a1 = borrow_step(a1, x2, bit_mask(c1, k + 2)); a2 = borrow_step(a2, x2, bit_mask(c2, k + 2));
      a1 = borrow_step(a1, x3, bit_mask(c1, k + 3)); a2 = borrow_step(a2, x3, bit_mask(c2, k + 3));
    }
    for (; k < w; ++k) {
      const uint32_t x = p[2 * k];
      a1 = borrow_step(a1, x, bit_mask(c1, k));
      a2 = borrow_step(a2, x, bit_mask(c2, k));
    }
  } else {
    for (int k = 0; k < w; ++k) {
      const uint32_t x = p[2 * k];
      a1 = eq1 ? eq_step(a1, x, bit_mask(c1, k)) : borrow_step(a1, x, bit_mask(c1, k));
      a2 = eq2 ? eq_step(a2, x, bit_mask(c2, k)) : borrow_step(a2, x, bit_mask(c2, k));
    }
  }
  *r1 = eq1 ? a1 : borrow_select(a1, op1);
  *r2 = eq2 ? a2 : borrow_select(a2, op2);
}

// IN: the planes are re-read from LDS once per constant, never from HBM (the reference makes K
// full passes over the block's words as well, fle-encoding.h:8283-8290).
template <typename ConstsPtrT>
__device__ __forceinline__ uint32_t pred_in_from_lds(const uint32_t* lds32, int w, int lane,
                                                     ConstsPtrT consts, int n_consts) {
  const uint32_t* p = lds32 + plane_base_dw(w, lane);
  uint32_t any = 0u;
#pragma unroll 1
  for (int j = 0; j < n_consts; ++j) {
    const uint32_t c = (uint32_t)consts[j];
    uint32_t ne = 0u;
    int k = w - 1;
    for (; k >= 3; k -= 4) {
      uint32_t x3 = p[2 * k], x2 = p[2 * k - 2], x1 = p[2 * k - 4], x0 = p[2 * k - 6];
      ne = ne_step(ne, x3, bit_mask(c, k));
      ne = ne_step(ne, x2, bit_mask(c, k - 1));
      ne = ne_step(ne, x1, bit_mask(c, k - 2));
      ne = ne_step(ne, x0, bit_mask(c, k - 3));
    }
    for (; k >= 0; --k) ne = ne_step(ne, p[2 * k], bit_mask(c, k));
    any |= ~ne;
  }
  return any;
}

__device__ __forceinline__ uint32_t pred_from_lds(const uint32_t* lds32, int w, int lane,
                                                  const PredArgs& a) {
  if (a.join != 0) {
    uint32_t r1, r2;
    pred_pair_from_lds(lds32, w, lane, a.op, a.consts[0], a.op2, a.const2, &r1, &r2);
    return a.join == 1 ? (r1 & r2) : (r1 | r2);
  }
  if (a.op != 5) return pred_single_from_lds(lds32, w, lane, a.op, a.consts[0]);
  if (a.in_list) return pred_in_from_lds(lds32, w, lane, a.in_list, a.in_list_n);
  return pred_in_from_lds(lds32, w, lane, a.consts, a.n_consts);
}

// ---- predicate on planes already in registers (compile-time W), single constant ------------
// (IN lists go through pred_from_lds: a K-deep loop around 32 live plane registers costs the
// kernel a third of its occupancy.)
template <int W>
__device__ __forceinline__ uint32_t pred_from_regs(const uint32_t (&p)[W], const PredArgs& a) {
  const uint32_t c = a.consts[0];
  if (a.op == 0) {  // wave-uniform
    uint32_t eq = ~0u;
#pragma unroll
    for (int k = 0; k < W; ++k) eq = eq_step(eq, p[k], bit_mask(c, k));
    return eq;
  }
  uint32_t b = borrow_init(a.op);
#pragma unroll
  for (int k = 0; k < W; ++k) b = borrow_step(b, p[k], bit_mask(c, k));
  return borrow_select(b, a.op);
}

// IN on planes in registers (W <= 16: every dictionary code width): one bit-select op per plane
// and constant, four constants per round so the scalar loads of the list are batched.
// not-equal mask of the half-block against one constant: ne | (plane ^ constant bit), one
// v_bitop3_b32 per plane
template <int W>
__device__ __forceinline__ uint32_t ne_const(const uint32_t (&p)[W], uint32_t c) {
  uint32_t ne = 0u;
#pragma unroll
  for (int k = 0; k < W; ++k) ne = ne_step(ne, p[k], bit_mask(c, k));
  return ne;
}

template <int W, typename ConstsPtrT>
__device__ __forceinline__ uint32_t pred_in_from_regs(const uint32_t (&p)[W], ConstsPtrT consts,
                                                      int n_consts) {
  uint32_t none = ~0u;  // rows equal to none of the constants so far
  int j = 0;
#pragma unroll 1
  for (; j + 4 <= n_consts; j += 4) {
    const uint32_t c0 = (uint32_t)consts[j], c1 = (uint32_t)consts[j + 1];
    const uint32_t c2 = (uint32_t)consts[j + 2], c3 = (uint32_t)consts[j + 3];
    const uint32_t n0 = ne_const<W>(p, c0), n1 = ne_const<W>(p, c1);
    const uint32_t n2 = ne_const<W>(p, c2), n3 = ne_const<W>(p, c3);
    none &= (n0 & n1) & (n2 & n3);
  }
#pragma unroll 1
  for (; j < n_consts; ++j) none &= ne_const<W>(p, (uint32_t)consts[j]);
  return ~none;
}

// Bitmap dword of this lane: bit j <-> row 32q+j of the block; rows >= n_rows are cleared.
__device__ __forceinline__ uint32_t finish_bitmap_dword(uint32_t sel_msb_first, int64_t tile,
                                                        int lane, int64_t n_rows) {
  uint32_t r = bitrev32(sel_msb_first);
  int64_t row0 = tile * kRowsPerTile + (int64_t)lane * 32;
  int64_t valid = n_rows - row0;
  if (valid < 32) r = valid <= 0 ? 0u : (r & ((1u << valid) - 1u));
  return r;
}

// Number of bitmap dwords that exist: 2 * ceil(n/64).
__device__ __forceinline__ int64_t bitmap_dwords(int64_t n_rows) { return 2 * ((n_rows + 63) / 64); }

// ---- order-preserving compaction of the selected rows of a sub-tile -------------------------
// Wave-wide inclusive prefix sum in 6 DPP adds (row_shr 1,2,4,8 inside each 16-lane row, then
// row_bcast:15 into rows 1,3 and row_bcast:31 into rows 2,3).
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x) {
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);  // row_shr:1
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);  // row_shr:2
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);  // row_shr:4
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);  // row_shr:8
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false); // row_bcast:15
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false); // row_bcast:31
  return x;
}

// Wave-wide maximum (same DPP steps as the scan; the result is wave-uniform).
__device__ __forceinline__ uint32_t wave_max(uint32_t x) {
  auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
  x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true));   // row_shr:1
  x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true));   // row_shr:2
  x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true));   // row_shr:4
  x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true));   // row_shr:8
  x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false));  // row_bcast:15
  x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false));  // row_bcast:31
  return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}

// Dense compaction image: element P of the sub-tile's selected rows lives at dword P + P/32 of the
// wave's LDS region.  The one-dword pad per 32 elements keeps the per-lane scatter below
// conflict-free up to 100 % selectivity (lane l writes element 32*l + j in step j: bank
// (j + l) mod 32).
__device__ __forceinline__ uint32_t compact_dw(uint32_t P) { return P + (P >> 5); }

// Dense path: each lane appends the selected ones of its own 32 rows (v[j] <-> bit j of bm) behind
// those of all lower lanes (P = exclusive prefix of the per-lane counts).  32 predicated LDS
// stores; the caller fences before (the region still holds the planes) and after.
__device__ __forceinline__ void compact_lane_values(uint32_t* lds32, uint32_t bm, uint32_t P,
                                                    const uint32_t (&v)[32]) {
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const bool sel = (bm & (1u << j)) != 0u;
    if (sel) lds32[P + (P >> 5)] = v[j];
    P += sel ? 1u : 0u;
  }
}

// compacted LDS image [0, count) -> dst[0, count), coalesced 16-byte stores; dst 16-byte aligned.
__device__ __forceinline__ void store_compacted(const uint32_t* lds32, uint32_t count,
                                                uint32_t* __restrict__ dst, int lane) {
  for (uint32_t p = 4u * lane; p < count; p += 4u * kWave) {
    const uint32_t* src = lds32 + compact_dw(p);  // 4 elements never straddle a pad
    if (p + 4 <= count) {
      u32x4 t = {src[0], src[1], src[2], src[3]};
      IPS_STREAM_STORE16(dst + p, t);  // dense path: full lines, written once (sel 50 %: -15 %)
    } else {
      for (uint32_t e = p; e < count; ++e) dst[e] = src[e - p];
    }
  }
}

// Write this lane's 32 row values into the padded row tile (8 x 16-byte LDS stores).
__device__ __forceinline__ void values_to_row_tile(uint32_t* lds32, int lane,
                                                   const uint32_t (&v)[32]) {
  uint32_t* dst = lds32 + lane * kRowTileStrideDw;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    u32x4 t = {v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
    *reinterpret_cast<u32x4*>(dst + 4 * i) = t;
  }
}

template <int W>
__device__ __forceinline__ void planes_from_lds(const uint32_t* lds32, int lane,
                                                uint32_t (&p)[W]) {
  const uint32_t* src = lds32 + plane_base_dw(W, lane);
#pragma unroll
  for (int k = 0; k < W; ++k) p[k] = src[2 * k];
}

}  // namespace ips
