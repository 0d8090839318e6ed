// ips_program.hip -- fused evaluation of a whole SimplePredicate tree over several columns in
// ONE pass: every referenced column chunk is read exactly once and a single bitmap is written.
//
// Replaces HdfsParquetScanner::EvalSimplePredicates (hdfs-parquet-scanner.cc:1837-1865) with the
// AndOperate / OrOperate / {Eq..In}Operate nodes of simple-predicates.h:145-205.  The reference
// materialises one dynamic_bitset per node per 1024-row batch; here the tree is a postfix program
// in the kernarg segment and the per-node bitmaps are 32-bit lane registers parked in LDS.
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <utility>

#include <atomic>

#include "ips_chunk_host.h"
#include "ips_chain.h"

namespace ips {

void run_pred_args(int bw, int op, const uint64_t* consts, int n_consts, int join, int op2, uint64_t const2,
                   int combine, PredArgs* a);  // ips_chunk.hip

// how ips_eval_program evaluates a tree (ips_set_program_strategy; process-wide, AUTO by default)
static std::atomic<int> g_program_strategy{IPS_PROGRAM_AUTO};

constexpr int kMaxLeaves = 16;
constexpr int kStackDepth = 8;
constexpr int kMaxJobs = 2 * kMaxLeaves;  // an 8-byte PLAIN leaf stages its sub-tile in two halves

// One 16-byte descriptor per program node: a single broadcast ds_read_b128 per node at run time.
struct NodeDesc {
  int8_t kind;       // ips_node_kind
  int8_t encoding;   // leaf: ips_col_encoding
  int8_t type;       // leaf: ips_type (PLAIN)
  int8_t n_stage;    // leaf: staging jobs to run before evaluating (0 = the previous leaf staged
                     //       the same column, e.g. BETWEEN = And(Ge, Le); 2 = 8-byte PLAIN halves)
  int8_t op;
  int8_t n_consts;
  int8_t bit_width;
  int8_t leaf_fuse;  // bits 0..3: index into consts[][]; bits 4..5: the AND/OR node that follows
                     // this leaf and has been folded into it (0 none, 1 AND, 2 OR)
  uint32_t const_lo, const_hi;  // consts[leaf][0]
};

// One staging job = one <= 8 KiB piece of one column's sub-tile: 8 x 16 bytes per lane.
enum JobKind { kJobFle = 0, kJobPlain4 = 1, kJobPlain8Lo = 2, kJobPlain8Hi = 3 };
struct JobDesc {
  const void* data;
  int32_t kind_width;  // kind | bit_width << 8
  uint32_t inv_width;  // floor(2^32 / bit_width) + 1 (FLE): block index without a division
};

struct Program {
  int32_t n_nodes;
  int32_t n_jobs;
  int32_t pad[2];
  NodeDesc nodes[IPS_PROGRAM_MAX_NODES];
  JobDesc jobs[kMaxJobs];
  uint64_t consts[kMaxLeaves][16];
};
static_assert(sizeof(NodeDesc) == 16 && sizeof(JobDesc) == 16, "descriptor layout");
static_assert(sizeof(Program) % 16 == 0 && sizeof(Program) <= 3968, "kernarg budget");

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// ---- staging: HBM -> VGPR (issued one job ahead) and VGPR -> LDS ------------------------------
__device__ __forceinline__ void job_load(const void* data, int kind, int w, int64_t tile,
                                         int64_t n_rows, int lane, u32x4 (&r)[8]) {
  if (kind == kJobFle) {
    tile_load<8>(reinterpret_cast<const uint64_t*>(data), tile, w, ((n_rows + 63) / 64) * w, lane, r);
  } else if (kind == kJobPlain4) {
    const uint32_t* page = reinterpret_cast<const uint32_t*>(data);
    const int64_t row_base = tile * kRowsPerTile;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rho = 4 * (i * kWave + lane);
      const int64_t valid = n_rows - (row_base + rho);
      u32x4 t = {0u, 0u, 0u, 0u};
      if (valid >= 4) {
        t = stream_load(reinterpret_cast<const u32x4*>(page + row_base + rho));
      } else {
        if (valid > 0) t.x = page[row_base + rho];
        if (valid > 1) t.y = page[row_base + rho + 1];
        if (valid > 2) t.z = page[row_base + rho + 2];
      }
      r[i] = t;
    }
  } else {
    const uint64_t* page = reinterpret_cast<const uint64_t*>(data);
    const int64_t row_base = tile * kRowsPerTile + (kind == kJobPlain8Hi ? 1024 : 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rr = 2 * (i * kWave + lane);
      const int64_t valid = n_rows - (row_base + rr);
      u32x4 t = {0u, 0u, 0u, 0u};
      if (valid >= 2) {
        t = stream_load(reinterpret_cast<const u32x4*>(page + row_base + rr));
      } else if (valid > 0) {
        u32x2 q = *reinterpret_cast<const u32x2*>(page + row_base + rr);
        t.x = q.x; t.y = q.y;
      }
      r[i] = t;
    }
  }
}

__device__ __forceinline__ void job_to_lds(int kind, int w, uint32_t inv_w, uint32_t* lds32,
                                           int lane, const u32x4 (&r)[8]) {
  if (kind == kJobFle) {
    tile_to_lds<8>(lds32, w, lane, r, inv_w);
  } else if (kind == kJobPlain4) {  // padded row tile: lane l later reads its 32 rows
#pragma unroll
    for (int i = 0; i < 8; ++i)
      *reinterpret_cast<u32x4*>(lds32 + row_tile_dw(4 * (i * kWave + lane))) = r[i];
  } else {  // 1024 8-byte rows: row rr at dword (rr>>4)*36 + (rr&15)*2
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rr = 2 * (i * kWave + lane);
      *reinterpret_cast<u32x4*>(lds32 + (rr >> 4) * kRowTileStrideDw + (rr & 15) * 2) = r[i];
    }
  }
}

// ---- PLAIN leaves: compare this lane's 32 consecutive rows -----------------------------------
template <typename T>
__device__ __forceinline__ T as_literal(uint64_t raw) {
  T lit;
  __builtin_memcpy(&lit, &raw, sizeof(T));
  return lit;
}

struct LeafHdr {  // wave-uniform
  int op;
  int n_consts;
  uint64_t const0;
  const uint64_t* consts;  // LDS copy of the leaf's constants (IN lists)
};

template <typename T>
__device__ __forceinline__ bool leaf_cmp(T x, const LeafHdr& lf) {
  const T lit = as_literal<T>(lf.const0);
  switch (lf.op) {
    case 0: return x == lit;
    case 1: return x < lit;
    case 2: return x <= lit;
    case 3: return x > lit;
    case 4: return x >= lit;
    default: {
      bool f = false;
      for (int j = 0; j < lf.n_consts; ++j) f = f || (x == as_literal<T>(lf.consts[j]));
      return f;
    }
  }
}

template <typename T>
__device__ __noinline__ uint32_t plain4_eval(const LeafHdr lf, const uint32_t* lds32, int lane) {
  uint32_t bm = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    u32x4 t = *reinterpret_cast<const u32x4*>(lds32 + lane * kRowTileStrideDw + 4 * i);
    uint32_t raw[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      T x;
      if constexpr (sizeof(T) == 4) __builtin_memcpy(&x, &raw[e], 4);
      else x = (T)(int32_t)raw[e];
      if (leaf_cmp<T>(x, lf)) bm |= 1u << (4 * i + e);
    }
  }
  return bm;
}

// half h of the sub-tile is staged: lanes 32h..32h+31 own its rows (two 16-row groups each)
template <typename T>
__device__ __noinline__ uint32_t plain8_eval(const LeafHdr lf, const uint32_t* lds32, int lane,
                                             int h) {
  uint32_t bm = 0;
  if ((lane >> 5) == h) {
    const int g0 = 2 * (lane & 31);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        u32x4 t = *reinterpret_cast<const u32x4*>(lds32 + (g0 + g) * kRowTileStrideDw + 4 * i);
        uint64_t raw[2] = {((uint64_t)t.y << 32) | t.x, ((uint64_t)t.w << 32) | t.z};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          T x;
          __builtin_memcpy(&x, &raw[e], 8);
          if (leaf_cmp<T>(x, lf)) bm |= 1u << (16 * g + 2 * i + e);
        }
      }
    }
  }
  return bm;
}

__device__ __forceinline__ uint32_t mask_rows(uint32_t bm, int64_t tile, int lane, int64_t n_rows) {
  int64_t valid = n_rows - (tile * kRowsPerTile + (int64_t)lane * 32);
  if (valid < 32) bm = valid <= 0 ? 0u : (bm & ((1u << valid) - 1u));
  return bm;
}

// Reading a by-value kernel argument with run-time indices makes the compiler copy all of it to
// scratch, and scalar loads from the kernarg segment cost a dependent ~1 us chain per leaf; so the
// workgroup copies the program once into LDS (through the constant-address-space kernarg pointer)
// and every node is one broadcast ds_read_b128 + v_readfirstlane.
#define IPS_KARG __attribute__((address_space(4)))

__global__ __launch_bounds__(kThreads) void program_kernel(Program prog, int64_t n_rows,
                                                           uint32_t* __restrict__ bitmap32) {
  // per wave: one staging region (row tile, also large enough for any plane tile) + node stack
  __shared__ __attribute__((aligned(16)))
      uint32_t lds_all[kWavesPerBlock * (kRowTileBytes / 4 + kStackDepth * kWave) +
                       sizeof(Program) / 4];
#if defined(__HIP_DEVICE_COMPILE__)
  (void)prog;
  const int lane = lane_id();
  const int wave = wave_id();
  uint32_t* lds32 = lds_all + wave * (kRowTileBytes / 4 + kStackDepth * kWave);
  uint32_t* stack = lds32 + kRowTileBytes / 4;
  uint32_t* prog32 = lds_all + kWavesPerBlock * (kRowTileBytes / 4 + kStackDepth * kWave);
  {
    const IPS_KARG uint32_t* karg = (const IPS_KARG uint32_t*)__builtin_amdgcn_kernarg_segment_ptr();
    for (int i = threadIdx.x; i < (int)(sizeof(Program) / 4); i += kThreads) prog32[i] = karg[i];
    __syncthreads();
  }
  const Program* P = reinterpret_cast<const Program*>(prog32);

  const int64_t tiles = (n_rows + kRowsPerTile - 1) / kRowsPerTile;
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock;
  const int64_t bm_dwords = bitmap_dwords(n_rows);
  const int n_jobs = (int)uni((uint32_t)P->n_jobs);
  const int n_nodes = (int)uni((uint32_t)P->n_nodes);

  // wave-uniform fetch of job j's descriptor
  auto job_desc = [&](int j, const void*& data, int& kind, int& w, uint32_t& inv_w) {
    u32x4 d = *reinterpret_cast<const u32x4*>(&P->jobs[j]);
    data = reinterpret_cast<const void*>(((uint64_t)uni(d.y) << 32) | uni(d.x));
    const uint32_t kw = uni(d.z);
    kind = (int)(kw & 0xFF);
    w = (int)(kw >> 8);
    inv_w = uni(d.w);
  };

  int64_t tile = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  u32x4 r[8];  // the NEXT staging job's bytes, always one job ahead of the evaluation
  const void* jdata; int jkind, jw; uint32_t jinv;  // descriptor of the job whose bytes are in r
  job_desc(0, jdata, jkind, jw, jinv);
  if (tile < tiles) job_load(jdata, jkind, jw, tile, n_rows, lane, r);

  // Stack machine with the top of stack in a register.  A leaf that is immediately followed by
  // AND/OR (every leaf of a conjunct chain, the second leaf of every BETWEEN) is folded with that
  // node on the host: top = top OP leaf, no stack traffic.  Only real pushes / pops go through the
  // eight spill registers, selected by the wave-uniform depth.
  uint32_t st[kStackDepth];
#pragma unroll
  for (int i = 0; i < kStackDepth; ++i) st[i] = 0u;
  (void)stack;
  constexpr int kNodeSkip = 3;  // an AND/OR node folded into the preceding leaf

  for (; tile < tiles; tile += stride) {
    int depth = 0;  // elements on the stack, the top one lives in 'top'
    uint32_t top = 0u;
    int job = 0;
    const uint32_t row_mask = mask_rows(~0u, tile, lane, n_rows);  // rows >= n_rows of this lane
#pragma unroll 1
    for (int n = 0; n < n_nodes; ++n) {
      const u32x4 nd = *reinterpret_cast<const u32x4*>(&P->nodes[n]);
      const uint32_t w0 = uni(nd.x), w1 = uni(nd.y);
      const int kind = (int)(w0 & 0xFF);
      if (kind == kNodeSkip) continue;
      if (kind == IPS_NODE_LEAF) {
        const int encoding = (int)((w0 >> 8) & 0xFF);
        const int type = (int)((w0 >> 16) & 0xFF);
        const int n_stage = (int)((w0 >> 24) & 0xFF);
        const int bit_width = (int)((w1 >> 16) & 0xFF);
        const int leaf = (int)((w1 >> 24) & 0xF);
        const int fuse = (int)((w1 >> 28) & 0x3);
        const LeafHdr lf{(int)(w1 & 0xFF), (int)((w1 >> 8) & 0xFF),
                         ((uint64_t)uni(nd.w) << 32) | uni(nd.z), &P->consts[leaf][0]};
        uint32_t bm = 0u;
#pragma unroll 1
        for (int h = 0; h < (n_stage ? n_stage : 1); ++h) {
          if (n_stage) {
            // bring the prefetched job from VGPRs into LDS and put the following one (possibly
            // of this wave's next sub-tile) in flight
            wave_lds_fence();  // earlier readers of the region are done
            job_to_lds(jkind, jw, jinv, lds32, lane, r);
            ++job;
            const bool wrap = job == n_jobs;  // next job belongs to this wave's next sub-tile
            const int64_t next_tile = wrap ? tile + stride : tile;
            job_desc(wrap ? 0 : job, jdata, jkind, jw, jinv);
            if (next_tile < tiles) job_load(jdata, jkind, jw, next_tile, n_rows, lane, r);
            wave_lds_fence();
          }
          if (encoding == IPS_COL_FLE) {
            uint32_t sel;
            if (lf.op != 5) {
              sel = pred_single_from_lds(lds32, bit_width, lane, lf.op, (uint32_t)lf.const0);
            } else {
              sel = 0u;
              const uint32_t* pl = lds32 + plane_base_dw(bit_width, lane);
#pragma unroll 1
              for (int j = 0; j < lf.n_consts; ++j) {
                const uint64_t cj = lf.consts[j];
                const uint32_t c = uni((uint32_t)cj);
                uint32_t eq = ~0u;
                for (int k = bit_width - 1; k >= 0; --k) eq &= ~(pl[2 * k] ^ bit_mask(c, k));
                sel |= eq;
              }
            }
            bm = bitrev32(sel);
          } else if (type == IPS_T_INT64 || type == IPS_T_DOUBLE) {
            bm |= type == IPS_T_INT64 ? plain8_eval<int64_t>(lf, lds32, lane, h)
                                      : plain8_eval<double>(lf, lds32, lane, h);
          } else {
            switch (type) {
              case IPS_T_INT8: bm = plain4_eval<int8_t>(lf, lds32, lane); break;
              case IPS_T_INT16: bm = plain4_eval<int16_t>(lf, lds32, lane); break;
              case IPS_T_INT32: bm = plain4_eval<int32_t>(lf, lds32, lane); break;
              default: bm = plain4_eval<float>(lf, lds32, lane); break;
            }
          }
        }
        bm &= row_mask;
        if (fuse == 1) {
          top &= bm;
        } else if (fuse == 2) {
          top |= bm;
        } else {  // push
          if (depth > 0) {
#pragma unroll
            for (int i = 0; i < kStackDepth; ++i)
              if (depth - 1 == i) st[i] = top;
          }
          top = bm;
          ++depth;
        }
      } else {  // AND / OR of the two topmost elements
        uint32_t below = 0u;
#pragma unroll
        for (int i = 0; i < kStackDepth; ++i)
          if (depth - 2 == i) below = st[i];
        top = kind == IPS_NODE_AND ? (below & top) : (below | top);
        --depth;
      }
    }
    const int64_t d = tile * 64 + lane;
    if (d < bm_dwords) IPS_BITMAP_STORE(bitmap32 + d, top);
  }
#endif
}

ips_status launch_program(const Program& prog, int64_t n_rows, uint32_t* bitmap32, hipStream_t s) {
  const int64_t tiles = (n_rows + kRowsPerTile - 1) / kRowsPerTile;
  int grid = grid_for_tiles(reinterpret_cast<const void*>(program_kernel), tiles);
  if (grid <= 0) return IPS_ERR_HIP;
  hipLaunchKernelGGL(program_kernel, dim3(grid), dim3(kThreads), 0, s, prog, n_rows, bitmap32);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

}  // namespace ips

using namespace ips;

// ---------------------------------------------------------------------------------------------
// Per-operand strategy (the default).  Every operand -- a single leaf, or two leaves on one
// column (BETWEEN = And(Ge, Le), simple-predicates.h:145-153, hdfs-parquet-scanner.cc:1857-1862)
// evaluated in the same pass -- is one launch of a stand-alone predicate kernel that writes, ANDs
// or ORs its result into a bitmap.  Those kernels run at 70-78 % of the HBM roofline (the
// measured read ceiling of the part), the single-launch program_kernel at ~20 %, so the extra
// read-modify-write of the bitmap (2 bits per row and operand) is cheap.  program_kernel stays
// as the one-launch alternative (IPS_PROGRAM_ONE_LAUNCH).
// ---------------------------------------------------------------------------------------------
namespace {

struct ChainItem {
  bool acc;             // the accumulator (already materialised in the output bitmap)
  const ips_node* a;    // leaf
  const ips_node* b;    // second leaf on the same column (range) or NULL
  int join;             // 1 AND / 2 OR between a and b
};

// Temporaries of the OPTIONAL columns of one call: per column the rank table (+ NOT-NULL bitmap
// when the levels are wider than a bit), prepared once; one data-row bitmap shared by all leaves
// (work on a stream is ordered).
struct NullableCol {
  int root_kind = 0;
  const uint64_t* root = nullptr;
  uint32_t* tile_counts = nullptr;
  bool counted = false;  // the tile counts ride on the column's first data-predicate launch
};
struct NullableCtx {
  NullableCol col[IPS_PROGRAM_MAX_COLS];
  uint64_t* sub = nullptr;
};

// carry: an OPTIONAL column whose tile counts ride on this (REQUIRED FLE) leaf's launch, or NULL
ips_status emit_item(const ChainItem& it, int combine, const ips_column* cols, int64_t n_rows,
                     uint64_t* d_bitmap, NullableCtx& nctx, hipStream_t s, NullableCol* carry = nullptr) {
  const ips_column& c = cols[it.a->column];
  if (c.encoding == IPS_COL_FLE) {
    PredArgs args;
    memset(&args, 0, sizeof(args));
    args.op = it.a->op;
    args.n_consts = it.a->n_consts;
    for (int j = 0; j < it.a->n_consts; ++j) args.consts[j] = (uint32_t)it.a->consts[j];
    if (it.a->inset) {  // an IN list of any length (ips_inset)
      bool none = false;
      inset_pred_args(it.a->inset, c.bit_width, &args, &none);
      if (none) { args.op = IPS_OP_LT; args.n_consts = 1; args.consts[0] = 0u; args.in_table = nullptr; args.in_list = nullptr; }
    }
    args.combine = combine;
    if (it.b) {
      args.join = it.join;
      args.op2 = it.b->op;
      args.const2 = (uint32_t)it.b->consts[0];
    }
    if (c.max_def_level > 0) {
      // ColumnReader's nullable branch (hdfs-parquet-scanner.cc:338-345): the predicate over the
      // data rows, then IntersectBitset into the NOT-NULL positions, combined into the target
      NullableCol& nc = nctx.col[it.a->column];
      const int64_t n_sub = c.n_data_rows < n_rows ? c.n_data_rows : n_rows;
      args.combine = 0;
      const uint64_t* enc = reinterpret_cast<const uint64_t*>(c.d_data);
      const bool fused = fused_leaf_enabled() && n_sub > 0;
      if (!nc.counted) {
        nc.counted = true;
        if (n_sub > 0 && !fused) {
          attach_rank_counts(&args, nc.root_kind, nc.root, n_rows, nc.tile_counts);
        } else {
          ips_status st = launch_rank_tile_counts(nc.root_kind, nc.root, n_rows, nc.tile_counts, s);
          if (st != IPS_OK) return st;
        }
      }
      if (fused) {  // predicate + IntersectBitset + combine in one kernel
        bool taken = false;
        ips_status st = launch_fle_leaf(c.bit_width, nc.root_kind, nc.root, n_rows, nc.tile_counts, enc,
                                        n_sub, args, d_bitmap, combine, &taken, s);
        if (st != IPS_OK || taken) return st;
      }
      if (n_sub > 0) {
        ips_status st = launch_fle_pred(c.bit_width, enc, n_sub, args, reinterpret_cast<uint32_t*>(nctx.sub), s);
        if (st != IPS_OK) return st;
      }
      return launch_expand(nc.root_kind, nc.root, nctx.sub, n_rows, n_sub, nc.tile_counts, d_bitmap,
                           combine, s);
    }
    if (carry && !carry->counted) {  // the counting workgroups of a later nullable leaf, for free
      attach_rank_counts(&args, carry->root_kind, carry->root, n_rows, carry->tile_counts);
      carry->counted = true;
    }
    return launch_fle_pred(c.bit_width, reinterpret_cast<const uint64_t*>(c.d_data), n_rows, args,
                           reinterpret_cast<uint32_t*>(d_bitmap), s);
  }
  // PLAIN: the constants carry the literals' bit patterns
  const int sz = (c.type == IPS_T_INT8) ? 1 : (c.type == IPS_T_INT16) ? 2
               : (c.type == IPS_T_INT64 || c.type == IPS_T_DOUBLE) ? 8 : 4;
  uint8_t lits[16 * 8], lit2[8];
  for (int j = 0; j < it.a->n_consts; ++j) memcpy(lits + j * sz, &it.a->consts[j], (size_t)sz);
  if (it.b) memcpy(lit2, &it.b->consts[0], (size_t)sz);
  if (c.max_def_level > 0) {
    // an OPTIONAL PLAIN column (SQL meaning; the reference ignores the levels here, quirk Q3): the
    // comparison over the stored values into the shared data-row bitmap, then IntersectBitset into the
    // NOT-NULL positions with the operand's combine mode
    NullableCol& nc = nctx.col[it.a->column];
    const int64_t n_sub = c.n_data_rows < n_rows ? c.n_data_rows : n_rows;
    if (!nc.counted) {
      nc.counted = true;
      ips_status st = launch_rank_tile_counts(nc.root_kind, nc.root, n_rows, nc.tile_counts, s);
      if (st != IPS_OK) return st;
    }
    if (n_sub > 0) {
      ips_status st = launch_plain_pred(c.type, c.d_data, n_sub, it.a->op, lits, it.a->n_consts, nctx.sub, s, 0,
                                        it.b ? it.join : 0, it.b ? it.b->op : 0, it.b ? lit2 : nullptr);
      if (st != IPS_OK) return st;
    }
    return launch_expand(nc.root_kind, nc.root, nctx.sub, n_rows, n_sub, nc.tile_counts, d_bitmap, combine, s);
  }
  return launch_plain_pred(c.type, c.d_data, n_rows, it.a->op, lits, it.a->n_consts, d_bitmap, s,
                           combine, it.b ? it.join : 0, it.b ? it.b->op : 0, it.b ? lit2 : nullptr);
}

// Plans the whole tree on a stack of bitmaps.  Operands are folded into an existing bitmap
// whenever one side of an AND / OR already is one (both commute); two leaf operands open a new
// bitmap; two bitmaps are merged by ips_bitmap_and / or's kernel.  The bitmap the root ends up in
// is mapped onto d_bitmap, the others (a left-deep conjunct chain has none) live in the CALLER's
// workspace (ips_program_workspace_bytes): the entry point allocates nothing, frees nothing and
// never synchronises, so it can be captured into a hipGraph like every other launch.
struct Item { int slot; const ips_node* a; const ips_node* b; int join; };  // slot < 0: leaf / pair
struct Step { int kind; Item item; int combine; int dst; int src; };         // 0 pred, 1 merge
struct Plan {
  Step steps[2 * IPS_PROGRAM_MAX_NODES];
  int n_steps = 0;
  int n_slots = 0;
  int root = 0;
};

bool make_plan(const ips_node* nodes, int n_nodes, Plan* pl) {
  Item stack[IPS_PROGRAM_MAX_NODES];
  int sp = 0;
  for (int i = 0; i < n_nodes; ++i) {
    const ips_node& nd = nodes[i];
    if (nd.kind == IPS_NODE_LEAF) {
      if (sp >= IPS_PROGRAM_MAX_NODES) return false;
      stack[sp++] = Item{-1, &nd, nullptr, 0};
      continue;
    }
    if (sp < 2) return false;
    const int op = nd.kind == IPS_NODE_AND ? 1 : 2;
    Item y = stack[--sp];
    Item x = stack[--sp];
    if (x.slot < 0 && y.slot < 0 && !x.b && !y.b && x.a->column == y.a->column &&
        x.a->op != IPS_OP_IN && y.a->op != IPS_OP_IN) {
      stack[sp++] = Item{-1, x.a, y.a, op};  // two leaves on one column: one pass
    } else if (x.slot >= 0 && y.slot >= 0) {
      pl->steps[pl->n_steps++] = Step{1, Item{}, op, x.slot, y.slot};
      stack[sp++] = x;
    } else if (x.slot >= 0 || y.slot >= 0) {
      const Item& bm = x.slot >= 0 ? x : y;
      pl->steps[pl->n_steps++] = Step{0, x.slot >= 0 ? y : x, op, bm.slot, -1};
      stack[sp++] = bm;
    } else {
      const int slot = pl->n_slots++;
      pl->steps[pl->n_steps++] = Step{0, x, 0, slot, -1};
      pl->steps[pl->n_steps++] = Step{0, y, op, slot, -1};
      stack[sp++] = Item{slot, nullptr, nullptr, 0};
    }
  }
  if (sp != 1) return false;
  if (stack[0].slot < 0) {
    pl->steps[pl->n_steps++] = Step{0, stack[0], 0, pl->n_slots, -1};
    stack[0].slot = pl->n_slots++;
  }
  pl->root = stack[0].slot;
  return true;
}

size_t plan_slot_bytes(int64_t n_rows) {
  const size_t bitmap_bytes = (size_t)((n_rows + 63) / 64) * 8;
  return (bitmap_bytes + 255) & ~(size_t)255;
}

// workspace: [plan slots beyond the root] [shared data-row bitmap] [per OPTIONAL column: rank table,
// NOT-NULL bitmap]
bool column_used(const ips_node* nodes, int n_nodes, int col) {
  for (int i = 0; i < n_nodes; ++i)
    if (nodes[i].kind == IPS_NODE_LEAF && nodes[i].column == col) return true;
  return false;
}

size_t nullable_part_bytes(const ips_node* nodes, int n_nodes, const ips_column* cols, int n_cols,
                           int64_t n_rows) {
  size_t bytes = 0;
  for (int c = 0; c < n_cols; ++c)
    if (cols[c].max_def_level > 0 && column_used(nodes, n_nodes, c))
      bytes += rank_workspace_bytes(n_rows) + plan_slot_bytes(n_rows);
  return bytes ? bytes + plan_slot_bytes(n_rows) : 0;
}

ips_status run_plan(const Plan& pl, const ips_node* nodes, int n_nodes, const ips_column* cols,
                    int n_cols, int64_t n_rows, uint64_t* d_bitmap, uint8_t* temp, hipStream_t s) {
  const size_t slot_bytes = plan_slot_bytes(n_rows);
  NullableCtx nctx;
  {
    uint8_t* p = temp + slot_bytes * (size_t)(pl.n_slots > 1 ? pl.n_slots - 1 : 0);
    bool any = false;
    for (int c = 0; c < n_cols; ++c) any = any || (cols[c].max_def_level > 0 && column_used(nodes, n_nodes, c));
    if (any) {
      nctx.sub = reinterpret_cast<uint64_t*>(p);
      p += slot_bytes;
      for (int c = 0; c < n_cols; ++c) {
        if (cols[c].max_def_level <= 0 || !column_used(nodes, n_nodes, c)) continue;
        NullableWs ws;
        ws.tile_counts = reinterpret_cast<uint32_t*>(p);
        ws.sub = nctx.sub;
        ws.nonnull = reinterpret_cast<uint64_t*>(p + rank_workspace_bytes(n_rows));
        p += rank_workspace_bytes(n_rows) + slot_bytes;
        ips_status st = nullable_prepare_root(cols[c].d_def_levels, cols[c].def_bit_width,
                                              cols[c].max_def_level, n_rows, ws, &nctx.col[c].root_kind,
                                              &nctx.col[c].root, s, /*count_tiles=*/false);
        if (st != IPS_OK) return st;
        nctx.col[c].tile_counts = ws.tile_counts;
      }
    }
  }
  auto slot_ptr = [&](int slot) -> uint64_t* {
    if (slot == pl.root) return d_bitmap;
    return reinterpret_cast<uint64_t*>(temp + slot_bytes * (size_t)(slot < pl.root ? slot : slot - 1));
  };
  // The tile counts of an OPTIONAL column are needed by its first leaf; if a leaf on a REQUIRED FLE
  // column runs before it, the counting workgroups ride on that launch (PredArgs::aux_*) instead of
  // being a launch of their own in front of the nullable leaf.
  int carry_col[2 * IPS_PROGRAM_MAX_NODES];
  for (int i = 0; i < pl.n_steps; ++i) carry_col[i] = -1;
  static const bool carry_off = dev_env("IPS_NO_COUNT_CARRY") != nullptr;  // dev switch for A/B runs
  for (int c = 0; c < n_cols && !carry_off; ++c) {
    if (cols[c].max_def_level <= 0) continue;
    int first = -1;
    for (int i = 0; i < pl.n_steps && first < 0; ++i)
      if (pl.steps[i].kind == 0 && pl.steps[i].item.a->column == c) first = i;
    for (int j = first - 1; j >= 0; --j) {
      const Step& q = pl.steps[j];
      if (q.kind != 0 || carry_col[j] >= 0) continue;
      const ips_column& qc = cols[q.item.a->column];
      if (qc.encoding == IPS_COL_FLE && qc.max_def_level == 0) {
        carry_col[j] = c;
        break;
      }
    }
  }
  ips_status st = IPS_OK;
  for (int i = 0; i < pl.n_steps && st == IPS_OK; ++i) {
    const Step& p = pl.steps[i];
    if (p.kind == 0) st = emit_item(ChainItem{false, p.item.a, p.item.b, p.item.join}, p.combine, cols, n_rows, slot_ptr(p.dst), nctx, s,
                                    carry_col[i] >= 0 ? &nctx.col[carry_col[i]] : nullptr);
    else st = launch_bitmap_binop(p.combine == 1 ? 0 : 1, slot_ptr(p.dst), slot_ptr(p.src), (n_rows + 63) / 64, s);
  }
  return st;
}

}  // namespace

extern "C" ips_status ips_set_program_strategy(int strategy) {
  IPS_REQUIRE(strategy >= IPS_PROGRAM_AUTO && strategy <= IPS_PROGRAM_ONE_LAUNCH,
              "ips_set_program_strategy: %d is not an ips_program_strategy", strategy);
  g_program_strategy.store(strategy, std::memory_order_relaxed);
  return IPS_OK;
}

extern "C" size_t ips_program_workspace_bytes(const ips_node* nodes, int n_nodes,
                                              const ips_column* cols, int n_cols, int64_t n_rows) {
  if (!nodes || n_nodes < 1 || n_nodes > IPS_PROGRAM_MAX_NODES || n_rows <= 0) return 0;
  for (int i = 0; i < n_nodes; ++i)
    if (nodes[i].kind == IPS_NODE_LEAF && (nodes[i].column < 0 || nodes[i].column >= n_cols)) return 0;
  Plan pl;
  if (!make_plan(nodes, n_nodes, &pl)) return 0;
  size_t bytes = pl.n_slots > 1 ? plan_slot_bytes(n_rows) * (size_t)(pl.n_slots - 1) : 0;
  if (cols && n_cols >= 1 && n_cols <= IPS_PROGRAM_MAX_COLS)
    bytes += nullable_part_bytes(nodes, n_nodes, cols, n_cols, n_rows);
  return bytes;
}

extern "C" ips_status ips_eval_program(const ips_node* nodes, int n_nodes, const ips_column* cols,
                                       int n_cols, int64_t n_rows, uint64_t* d_bitmap,
                                       void* d_workspace, ips_stream stream) {
  IPS_REQUIRE(nodes && n_nodes >= 1 && n_nodes <= IPS_PROGRAM_MAX_NODES,
              "ips_eval_program: n_nodes %d not in 1..%d", n_nodes, IPS_PROGRAM_MAX_NODES);
  IPS_REQUIRE(cols && n_cols >= 1 && n_cols <= IPS_PROGRAM_MAX_COLS,
              "ips_eval_program: n_cols %d not in 1..%d", n_cols, IPS_PROGRAM_MAX_COLS);
  IPS_REQUIRE(n_rows >= 0, "ips_eval_program: n_rows < 0");
  IPS_REQUIRE(n_rows == 0 || (d_bitmap && aligned16(d_bitmap)), "ips_eval_program: bitmap NULL or misaligned");
  Program prog;
  memset(&prog, 0, sizeof(prog));
  prog.n_nodes = n_nodes;
  int depth = 0, max_depth = 0, n_leaves = 0;
  bool any_nullable = false, any_inset = false;
  const void* last_staged = nullptr;
  for (int i = 0; i < n_nodes; ++i) {
    const ips_node& nd = nodes[i];
    NodeDesc& desc = prog.nodes[i];
    desc.kind = (int8_t)nd.kind;
    if (nd.kind == IPS_NODE_LEAF) {
      IPS_REQUIRE(n_leaves < kMaxLeaves, "ips_eval_program: more than %d leaves", kMaxLeaves);
      IPS_REQUIRE(nd.column >= 0 && nd.column < n_cols, "ips_eval_program: node %d: bad column", i);
      IPS_REQUIRE(nd.op >= IPS_OP_EQ && nd.op <= IPS_OP_IN, "ips_eval_program: node %d: bad op", i);
      IPS_REQUIRE(nd.inset || (nd.n_consts >= 1 && nd.n_consts <= 16 && (nd.op == IPS_OP_IN || nd.n_consts == 1)),
                  "ips_eval_program: node %d: bad constant count", i);
      IPS_REQUIRE(!nd.inset || (nd.op == IPS_OP_IN && cols[nd.column].encoding == IPS_COL_FLE && nd.n_consts >= 0 && nd.n_consts <= 16),
                  "ips_eval_program: node %d: a set belongs to an IN leaf on an FLE column", i);
      any_inset = any_inset || nd.inset != nullptr;
      const ips_column& c = cols[nd.column];
      IPS_REQUIRE(n_rows == 0 || (c.d_data && aligned16(c.d_data)),
                  "ips_eval_program: column %d: data NULL or misaligned", nd.column);
      IPS_REQUIRE(c.max_def_level >= 0, "ips_eval_program: column %d: max_def_level < 0", nd.column);
      if (c.max_def_level > 0) {
        IPS_REQUIRE(c.def_bit_width >= 1 && c.def_bit_width <= 32 &&
                    (c.def_bit_width == 32 || (uint64_t)c.max_def_level < (1ull << c.def_bit_width)),
                    "ips_eval_program: column %d: max_def_level does not fit the level width", nd.column);
        IPS_REQUIRE(n_rows == 0 || (c.d_def_levels && aligned16(c.d_def_levels)),
                    "ips_eval_program: column %d: definition levels NULL or misaligned", nd.column);
        IPS_REQUIRE(c.n_data_rows >= 0, "ips_eval_program: column %d: n_data_rows < 0", nd.column);
        any_nullable = true;
      }
      if (c.encoding == IPS_COL_FLE) {
        IPS_REQUIRE(c.bit_width >= 1 && c.bit_width <= 32, "ips_eval_program: column %d: bit width", nd.column);
        const uint64_t limit = c.bit_width == 32 ? 0xFFFFFFFFull : ((1ull << c.bit_width) - 1ull);
        for (int j = 0; j < nd.n_consts && !nd.inset; ++j)
          IPS_REQUIRE(nd.consts[j] <= limit, "ips_eval_program: node %d: constant does not fit the bit width", i);
      } else {
        IPS_REQUIRE(c.encoding == IPS_COL_PLAIN, "ips_eval_program: column %d: bad encoding", nd.column);
        IPS_REQUIRE(c.type >= IPS_T_INT8 && c.type <= IPS_T_DOUBLE, "ips_eval_program: column %d: bad type", nd.column);
      }
      desc.encoding = (int8_t)c.encoding;
      desc.type = (int8_t)c.type;
      desc.op = (int8_t)nd.op;
      desc.n_consts = (int8_t)nd.n_consts;
      desc.bit_width = (int8_t)c.bit_width;
      desc.leaf_fuse = (int8_t)n_leaves;
      if (depth >= 1 && i + 1 < n_nodes &&
          (nodes[i + 1].kind == IPS_NODE_AND || nodes[i + 1].kind == IPS_NODE_OR))
        desc.leaf_fuse = (int8_t)(n_leaves | ((nodes[i + 1].kind == IPS_NODE_AND ? 1 : 2) << 4));
      desc.const_lo = (uint32_t)nd.consts[0];
      desc.const_hi = (uint32_t)(nd.consts[0] >> 32);
      for (int j = 0; j < nd.n_consts; ++j) prog.consts[n_leaves][j] = nd.consts[j];
      // staging jobs: a leaf on the column the previous leaf staged re-uses the LDS image
      const bool wide = c.encoding == IPS_COL_PLAIN && (c.type == IPS_T_INT64 || c.type == IPS_T_DOUBLE);
      if (!wide && last_staged == c.d_data) {
        desc.n_stage = 0;
      } else {
        desc.n_stage = wide ? 2 : 1;
        JobDesc& jb = prog.jobs[prog.n_jobs++];
        jb.data = c.d_data;
        const int jkind = c.encoding == IPS_COL_FLE ? kJobFle : wide ? kJobPlain8Lo : kJobPlain4;
        jb.kind_width = jkind | (c.bit_width << 8);
        jb.inv_width = c.encoding == IPS_COL_FLE ? (uint32_t)(0x100000000ull / (uint64_t)c.bit_width) + 1u : 0u;
        if (wide) {
          JobDesc& hi = prog.jobs[prog.n_jobs++];
          hi = jb;
          hi.kind_width = kJobPlain8Hi;
        }
        last_staged = wide ? nullptr : c.d_data;
      }
      ++n_leaves;
      ++depth;
    } else {
      IPS_REQUIRE(nd.kind == IPS_NODE_AND || nd.kind == IPS_NODE_OR, "ips_eval_program: node %d: bad kind", i);
      IPS_REQUIRE(depth >= 2, "ips_eval_program: node %d: stack underflow", i);
      --depth;
      // folded into the leaf right before it?  (the kernel then skips this node)
      if (i > 0 && nodes[i - 1].kind == IPS_NODE_LEAF && (prog.nodes[i - 1].leaf_fuse >> 4) != 0)
        desc.kind = 3;
    }
    if (depth > max_depth) max_depth = depth;
  }
  IPS_REQUIRE(depth == 1, "ips_eval_program: program leaves %d bitmaps on the stack", depth);
  IPS_REQUIRE(max_depth <= kStackDepth, "ips_eval_program: tree deeper than %d", kStackDepth);
  if (n_rows == 0) return IPS_OK;
  const int strategy = g_program_strategy.load(std::memory_order_relaxed);
  if (strategy != IPS_PROGRAM_ONE_LAUNCH) {
    Plan pl;
    if (make_plan(nodes, n_nodes, &pl)) {
      // a pure chain over REQUIRED FLE columns runs as one pass with one bitmap write (ips_chain.hip):
      // what AUTO picks, and what IPS_PROGRAM_ONE_PASS asks for; IPS_PROGRAM_PER_OPERAND keeps the launches
      if (pl.n_slots == 1 && !any_nullable && pl.n_steps >= 2 && pl.n_steps <= kChainWMaxOps &&
          (strategy == IPS_PROGRAM_ONE_PASS || strategy == IPS_PROGRAM_AUTO)) {
        ChainArgsW ca;
        memset(&ca, 0, sizeof(ca));
        const void* enc[kChainWMaxOps];
        bool ok = true;
        for (int i = 0; i < pl.n_steps && ok; ++i) {
          const Step& p = pl.steps[i];
          const ips_node* la = p.item.a;
          const ips_node* lb = p.item.b;
          const ips_column& c = cols[la->column];
          ok = p.kind == 0 && c.encoding == IPS_COL_FLE && (i == 0 ? p.combine == 0 : p.combine != 0) && !la->inset;
          // AUTO leaves 32-bit comparisons to their stand-alone kernel, which reads the low planes only of
          // the sub-tiles the high planes leave undecided ((32, 8): 153 us per operand, 233 us in one pass)
          if (strategy == IPS_PROGRAM_AUTO && c.bit_width == 32 && la->op != IPS_OP_IN) ok = false;
          if (!ok) break;
          ChainOpW& o = ca.ops[i];
          enc[i] = c.d_data;
          o.w = c.bit_width;
          o.op = la->op;
          o.c1 = (uint32_t)la->consts[0];
          o.combine = p.combine;
          if (la->op == IPS_OP_IN) {
            o.kind = kChainIn;
            o.n_in = la->n_consts;
            for (int j = 0; j < la->n_consts; ++j) o.in_consts[j] = (uint32_t)la->consts[j];
          } else if (lb) {
            o.kind = kChainPair;
            o.join = p.item.join;
            o.op2 = lb->op;
            o.c2 = (uint32_t)lb->consts[0];
          } else {
            o.kind = kChainSingle;
          }
        }
        if (ok) {
          ca.n_ops = pl.n_steps;
          const ips_status st = launch_chain_w(ca, enc, n_rows, reinterpret_cast<uint32_t*>(d_bitmap),
                                               reinterpret_cast<hipStream_t>(stream));
          if (st != IPS_ERR_UNSUPPORTED) return st;  // (too many load slots, a column of 4 GiB: the plan below)
        }
      }
      IPS_REQUIRE((pl.n_slots <= 1 && !any_nullable) || (d_workspace && aligned16(d_workspace)),
                  "ips_eval_program: this tree needs temporaries (%d bitmaps alive%s): pass a workspace of "
                  "ips_program_workspace_bytes() bytes", pl.n_slots, any_nullable ? ", OPTIONAL columns" : "");
      return run_plan(pl, nodes, n_nodes, cols, n_cols, n_rows, d_bitmap,
                      reinterpret_cast<uint8_t*>(d_workspace), reinterpret_cast<hipStream_t>(stream));
    }
  }
  if (any_nullable || any_inset) {
    set_error("ips_eval_program: OPTIONAL columns and IN sets are evaluated by the per-operand plan only");
    return IPS_ERR_UNSUPPORTED;
  }
  return launch_program(prog, n_rows, reinterpret_cast<uint32_t*>(d_bitmap),
                        reinterpret_cast<hipStream_t>(stream));
}

// ---------------------------------------------------------------------------------------------
// The same tree over column chunks held as LISTS OF PAGES (ips_chunk): what EvalSimplePredicates
// does across ReadDataPage boundaries (hdfs-parquet-scanner.cc:1837-1855: every batch ends where
// the first column's page ends; pages of different columns need not align).  Per operand and per
// run of equally wide pages ONE launch (blockIdx.y = page); every page is evaluated in its own
// block geometry and writes / ANDs / ORs its rows' bits at its row offset of the bitmap
// (ips_chunk_device.h), so the concatenation of the reference's per-batch bitsets comes out
// whatever the page boundaries are.  OPTIONAL chunks go through the paged nullable leaf; their
// tile counts are one extra launch per column and call.
// ---------------------------------------------------------------------------------------------
namespace {

size_t align256z(size_t x) { return (x + 255) & ~(size_t)255; }

struct ChunkCtx {
  uint32_t* rank[IPS_PROGRAM_MAX_COLS] = {};
  bool counted[IPS_PROGRAM_MAX_COLS] = {};
};

// done: the launches count their waves per page (the last operand of a sharded step), or NULL;
// *signalled = false when this operand has no counting kernel
ips_status emit_item_chunk(const ChainItem& it, int combine, const ips_chunk* const* chunks, int64_t n_rows,
                           uint64_t* d_bitmap, ChunkCtx& ctx, hipStream_t s, uint32_t* done = nullptr,
                           bool* signalled = nullptr, uint32_t done_epoch = 0) {
  const int col = it.a->column;
  const ips_chunk* c = chunks[col];
  if (c->pages.empty()) return IPS_OK;
  uint32_t* bm32 = reinterpret_cast<uint32_t*>(d_bitmap);
  if (c->encoding == IPS_COL_FLE) {
    if (c->max_def_level > 0 && !ctx.counted[col]) {
      int64_t max_rows = 0;
      for (const ips_chunk::Run& run : c->runs) max_rows = run.max_rows > max_rows ? run.max_rows : max_rows;
      ips_status st = launch_rank_counts_pages(c->d_pages, (int)c->pages.size(), max_rows, ctx.rank[col], s);
      if (st != IPS_OK) return st;
      ctx.counted[col] = true;
    }
    for (const ips_chunk::Run& run : c->runs) {
      PredArgs args;
      run_pred_args(run.bit_width, it.a->op, it.a->consts, it.a->n_consts, it.b ? it.join : 0, it.b ? it.b->op : 0,
                    it.b ? it.b->consts[0] : 0, combine, &args);
      if (it.a->inset) {  // an IN list of any length (ips_inset); no member below 2^width: LT 0 = nothing
        bool none = false;
        inset_pred_args(it.a->inset, run.bit_width, &args, &none);
        args.combine = combine;
        if (none) { args.op = IPS_OP_LT; args.n_consts = 1; args.consts[0] = 0u; args.in_table = nullptr; args.in_list = nullptr; }
      }
      ips_status st;
      if (done) {
        args.done = done;
        args.done_page0 = run.first;
        args.done_epoch = done_epoch;
        *signalled = true;
      }
      if (c->max_def_level > 0) {
        args.aux_counts = ctx.rank[col];
        st = launch_fle_leaf_pages(run.bit_width, c->d_pages + run.first, run.count, run.max_rows, n_rows, args, bm32, s);
      } else {
        // pages that start inside a bitmap dword: fle_pred_body walks them in stripes of 62 whole dwords (no shared
        // dwords but the page's two ends); only the early-pruning w = 32 kernel still uses the edge slots + fix-up
        const bool early32 = run.bit_width == 32 && args.op != IPS_OP_IN;
        args.edges = (done || !early32) ? nullptr : c->d_edges;  // (a signalling launch merges its shared dwords itself)
        st = launch_fle_pred_pages(run.bit_width, c->d_pages + run.first, run.count, run.max_rows, n_rows, args, bm32, s);
        if (st == IPS_OK && args.edges)
          st = launch_window_fixup(c->d_pages + run.first, run.count, run.max_rows, n_rows, bm32, c->d_edges, combine, s);
      }
      if (st != IPS_OK) return st;
    }
    return IPS_OK;
  }
  const int sz = (c->type == IPS_T_INT8) ? 1 : (c->type == IPS_T_INT16) ? 2
               : (c->type == IPS_T_INT64 || c->type == IPS_T_DOUBLE) ? 8 : 4;
  uint8_t lits[16 * 8], lit2[8];
  for (int j = 0; j < it.a->n_consts; ++j) memcpy(lits + j * sz, &it.a->consts[j], (size_t)sz);
  if (it.b) memcpy(lit2, &it.b->consts[0], (size_t)sz);
  return launch_plain_pred_pages(c->type, c->d_pages, (int)c->pages.size(), c->runs[0].max_rows, n_rows, it.a->op,
                                 lits, it.a->n_consts, d_bitmap, s, combine, it.b ? it.join : 0, it.b ? it.b->op : 0,
                                 it.b ? lit2 : nullptr, c->d_edges);
}

ips_status check_chunk_program(const ips_node* nodes, int n_nodes, const ips_chunk* const* chunks, int n_chunks,
                               int64_t* n_rows) {
  IPS_REQUIRE(nodes && n_nodes >= 1 && n_nodes <= IPS_PROGRAM_MAX_NODES,
              "ips_eval_program_chunks: n_nodes %d not in 1..%d", n_nodes, IPS_PROGRAM_MAX_NODES);
  IPS_REQUIRE(chunks && n_chunks >= 1 && n_chunks <= IPS_PROGRAM_MAX_COLS,
              "ips_eval_program_chunks: n_chunks %d not in 1..%d", n_chunks, IPS_PROGRAM_MAX_COLS);
  for (int c = 0; c < n_chunks; ++c) {
    IPS_REQUIRE(chunks[c] != nullptr, "ips_eval_program_chunks: chunk %d is NULL", c);
    IPS_REQUIRE(chunks[c]->n_rows == chunks[0]->n_rows,
                "ips_eval_program_chunks: chunk %d holds %lld rows, chunk 0 %lld (the columns of one row group)", c,
                (long long)chunks[c]->n_rows, (long long)chunks[0]->n_rows);
  }
  *n_rows = chunks[0]->n_rows;
  for (int i = 0; i < n_nodes; ++i) {
    const ips_node& nd = nodes[i];
    if (nd.kind == IPS_NODE_LEAF) {
      IPS_REQUIRE(nd.column >= 0 && nd.column < n_chunks, "ips_eval_program_chunks: node %d: bad column", i);
      IPS_REQUIRE(nd.op >= IPS_OP_EQ && nd.op <= IPS_OP_IN, "ips_eval_program_chunks: node %d: bad op", i);
      IPS_REQUIRE(nd.inset || (nd.n_consts >= 1 && nd.n_consts <= 16 && (nd.op == IPS_OP_IN || nd.n_consts == 1)),
                  "ips_eval_program_chunks: node %d: bad constant count", i);
      IPS_REQUIRE(!nd.inset || (nd.op == IPS_OP_IN && chunks[nd.column]->encoding == IPS_COL_FLE && nd.n_consts >= 0 && nd.n_consts <= 16),
                  "ips_eval_program_chunks: node %d: a set belongs to an IN leaf on an FLE column", i);
    } else {
      IPS_REQUIRE(nd.kind == IPS_NODE_AND || nd.kind == IPS_NODE_OR, "ips_eval_program_chunks: node %d: bad kind", i);
    }
  }
  return IPS_OK;
}

}  // namespace

namespace {
// A plan that the one-pass chain can take over page lists: its operands, their page tables, and whether all
// chunks are cut at the same rows
struct ChunkChain {
  ChainOpW ops[kChainWMaxOps];
  const void* op_pages[kChainWMaxOps];
  int op_n_pages[kChainWMaxOps];
  const ips_chunk* c0;
  bool co_paged;
  int n_bounds;       // page starts of all operands
  int64_t max_rows;   // of the largest page of any operand
};
bool chunk_chain(const Plan& pl, const ips_chunk* const* chunks, ChunkChain* cc) {
  const int strategy = g_program_strategy.load(std::memory_order_relaxed);
  if (!(pl.n_slots == 1 && pl.n_steps >= 2 && pl.n_steps <= kChainWMaxOps &&
        (strategy == IPS_PROGRAM_AUTO || strategy == IPS_PROGRAM_ONE_PASS)))
    return false;
  memset(cc->ops, 0, sizeof(cc->ops));
  cc->c0 = chunks[pl.steps[0].item.a->column];
  cc->co_paged = true;
  cc->n_bounds = 0;
  cc->max_rows = 0;
  for (int i = 0; i < pl.n_steps; ++i) {
    const Step& p = pl.steps[i];
    if (p.kind != 0) return false;
    const ips_node* la = p.item.a;
    const ips_node* lb = p.item.b;
    const ips_chunk* c = chunks[la->column];
    if (!(c->encoding == IPS_COL_FLE && c->max_def_level == 0 && c->runs.size() == 1 && !la->inset &&
          (i == 0 ? p.combine == 0 : p.combine != 0)))
      return false;
    if (strategy == IPS_PROGRAM_AUTO && c->runs[0].bit_width == 32 && la->op != IPS_OP_IN) return false;
    if (c->pages.size() != cc->c0->pages.size()) cc->co_paged = false;
    for (size_t k = 0; cc->co_paged && k < c->pages.size(); ++k) cc->co_paged = c->pages[k].n_rows == cc->c0->pages[k].n_rows;
    PredArgs folded;  // constants that do not fit the chunk's width make their comparison constant
    run_pred_args(c->runs[0].bit_width, la->op, la->consts, la->n_consts, lb ? p.item.join : 0, lb ? lb->op : 0,
                  lb ? lb->consts[0] : 0, p.combine, &folded);
    ChainOpW& o = cc->ops[i];
    o.w = c->runs[0].bit_width;
    o.op = folded.op;
    o.c1 = folded.consts[0];
    o.combine = p.combine;
    if (folded.op == IPS_OP_IN) {
      if (folded.n_consts > 16) return false;
      o.kind = kChainIn;
      o.n_in = folded.n_consts;
      for (int j = 0; j < folded.n_consts; ++j) o.in_consts[j] = folded.consts[j];
    } else if (lb) {
      o.kind = kChainPair;
      o.join = folded.join;
      o.op2 = folded.op2;
      o.c2 = folded.const2;
    } else {
      o.kind = kChainSingle;
    }
    cc->op_pages[i] = c->d_pages;
    cc->op_n_pages[i] = (int)c->pages.size();
    cc->n_bounds += (int)c->pages.size();
    cc->max_rows = c->runs[0].max_rows > cc->max_rows ? c->runs[0].max_rows : cc->max_rows;
  }
  return true;
}
}  // namespace

extern "C" size_t ips_chunk_program_workspace_bytes(const ips_node* nodes, int n_nodes,
                                                    const ips_chunk* const* chunks, int n_chunks) {
  int64_t n_rows = 0;
  if (check_chunk_program(nodes, n_nodes, chunks, n_chunks, &n_rows) != IPS_OK) return 0;
  Plan pl;
  if (!make_plan(nodes, n_nodes, &pl)) return 0;
  size_t bytes = pl.n_slots > 1 ? plan_slot_bytes(n_rows) * (size_t)(pl.n_slots - 1) : 0;
  for (int c = 0; c < n_chunks; ++c)
    if (chunks[c]->max_def_level > 0 && column_used(nodes, n_nodes, c)) bytes += align256z((size_t)chunks[c]->rank_entries * 4);
  ChunkChain cc;  // a one-pass chain over chunks cut at different rows: the segment tables and edge slots
  if (n_rows > 0 && g_program_strategy.load(std::memory_order_relaxed) == IPS_PROGRAM_ONE_PASS &&
      chunk_chain(pl, chunks, &cc) && !cc.co_paged && cc.n_bounds <= kChainSegMaxBounds)
    bytes += chain_segments_workspace_bytes(cc.n_bounds, n_rows);
  return bytes;
}

namespace ips {
ips_status eval_program_chunks_signalled(const ips_node* nodes, int n_nodes, const ips_chunk* const* chunks, int n_chunks,
                                         uint64_t* d_bitmap, void* d_workspace, uint32_t* done, uint32_t done_epoch,
                                         bool* signalled, hipStream_t s) {
  if (signalled) *signalled = false;
  int64_t n_rows = 0;
  ips_status st = check_chunk_program(nodes, n_nodes, chunks, n_chunks, &n_rows);
  if (st != IPS_OK) return st;
  IPS_REQUIRE(n_rows == 0 || (d_bitmap && aligned16(d_bitmap)), "ips_eval_program_chunks: bitmap NULL or misaligned");
  Plan pl;
  if (!make_plan(nodes, n_nodes, &pl)) {
    set_error("ips_eval_program_chunks: the nodes are not a postfix program that leaves one bitmap");
    return IPS_ERR_INVALID_ARG;
  }
  if (n_rows == 0) return IPS_OK;
  // (the segment tables of a one-pass chain over differently cut chunks are the one optional part of the workspace:
  // without it such a plan takes the per-operand launches)
  ChunkChain cc;
  const bool chain = chunk_chain(pl, chunks, &cc);
  // (chunks cut at different rows: the segmented chain is what IPS_PROGRAM_ONE_PASS asks for; AUTO keeps the
  // per-operand launches, which measure better -- 367 us against 397-405 on the Q6 shape, ips_chain.hip)
  const bool segmented = chain && !cc.co_paged && cc.n_bounds <= kChainSegMaxBounds &&
                         g_program_strategy.load(std::memory_order_relaxed) == IPS_PROGRAM_ONE_PASS;
  const size_t need = ips_chunk_program_workspace_bytes(nodes, n_nodes, chunks, n_chunks);
  IPS_REQUIRE(need == 0 || segmented || (d_workspace && aligned16(d_workspace)),
              "ips_eval_program_chunks: pass a workspace of ips_chunk_program_workspace_bytes() bytes");
  IPS_REQUIRE(!d_workspace || aligned16(d_workspace), "ips_eval_program_chunks: misaligned workspace");
  uint8_t* temp = reinterpret_cast<uint8_t*>(d_workspace);
  const size_t slot_bytes = plan_slot_bytes(n_rows);
  ChunkCtx ctx;
  {
    uint8_t* p = temp + slot_bytes * (size_t)(pl.n_slots > 1 ? pl.n_slots - 1 : 0);
    for (int c = 0; c < n_chunks; ++c) {
      if (chunks[c]->max_def_level <= 0 || !column_used(nodes, n_nodes, c)) continue;
      ctx.rank[c] = reinterpret_cast<uint32_t*>(p);
      p += align256z((size_t)chunks[c]->rank_entries * 4);
    }
  }
  auto slot_ptr = [&](int slot) -> uint64_t* {
    if (slot == pl.root) return d_bitmap;
    return reinterpret_cast<uint64_t*>(temp + slot_bytes * (size_t)(slot < pl.root ? slot : slot - 1));
  };
  // A chain over REQUIRED FLE chunks runs as ONE pass (ips_chain.hip): pages that hold the same rows in every
  // chunk with blockIdx.y = page; pages cut differently per column with blockIdx.y = segment (the rows between
  // two neighbouring page starts of any operand).  OPTIONAL chunks, dictionary widths that grow inside a chunk and
  // 32-bit comparisons (AUTO) take the per-operand launches below.
  if (chain) {
    uint32_t* bm32 = reinterpret_cast<uint32_t*>(d_bitmap);
    if (cc.co_paged) {
      ChainPagedArgsW pa;
      memset(&pa, 0, sizeof(pa));
      memcpy(pa.chain.ops, cc.ops, sizeof(cc.ops));
      pa.chain.n_ops = pl.n_steps;
      pa.pg.chunk_rows = n_rows;
      pa.pg.edges = done ? nullptr : cc.c0->d_edges;  // (a signalling launch merges its shared dwords itself)
      pa.pg.done = done;
      pa.pg.done_page0 = 0;
      pa.pg.done_epoch = done_epoch;
      st = launch_chain_w_pages(pa, cc.op_pages, (int)cc.c0->pages.size(), cc.c0->runs[0].max_rows, bm32, s);
      if (st == IPS_OK && pa.pg.edges)
        st = launch_window_fixup(cc.c0->d_pages, (int)cc.c0->pages.size(), cc.c0->runs[0].max_rows, n_rows, bm32, cc.c0->d_edges, 0, s);
      if (st == IPS_OK && done && signalled) *signalled = true;
      if (st != IPS_ERR_UNSUPPORTED) return st;
      st = IPS_OK;
    } else if (!done && segmented && temp != nullptr) {
      ChainSegmentedArgsW sa;
      memset(&sa, 0, sizeof(sa));
      memcpy(sa.chain.ops, cc.ops, sizeof(cc.ops));
      sa.chain.n_ops = pl.n_steps;
      st = launch_chain_w_segments(sa, cc.op_pages, cc.op_n_pages, cc.max_rows, n_rows, bm32, temp, s);
      if (st != IPS_ERR_UNSUPPORTED) return st;
      st = IPS_OK;
    }
  }
  for (int i = 0; i < pl.n_steps && st == IPS_OK; ++i) {
    const Step& p = pl.steps[i];
    const bool last = i + 1 == pl.n_steps;
    if (p.kind == 0)
      st = emit_item_chunk(ChainItem{false, p.item.a, p.item.b, p.item.join}, p.combine, chunks, n_rows, slot_ptr(p.dst), ctx, s,
                           last ? done : nullptr, signalled, done_epoch);
    else
      st = launch_bitmap_binop(p.combine == 1 ? 0 : 1, slot_ptr(p.dst), slot_ptr(p.src), (n_rows + 63) / 64, s);
  }
  return st;
}
}  // namespace ips

extern "C" ips_status ips_eval_program_chunks(const ips_node* nodes, int n_nodes, const ips_chunk* const* chunks,
                                              int n_chunks, uint64_t* d_bitmap, void* d_workspace,
                                              ips_stream stream) {
  return eval_program_chunks_signalled(nodes, n_nodes, chunks, n_chunks, d_bitmap, d_workspace, nullptr, 0, nullptr,
                                       reinterpret_cast<hipStream_t>(stream));
}
