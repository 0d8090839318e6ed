// ips_program.hip -- fused evaluation of a whole SimplePredicate tree over several columns in
// ONE pass: every referenced column chunk is read exactly once and a single bitmap is written.
//
// Replaces HdfsParquetScanner::EvalSimplePredicates (hdfs-parquet-scanner.cc:1837-1865) with the
// AndOperate / OrOperate / {Eq..In}Operate nodes of simple-predicates.h:145-205.  The reference
// materialises one dynamic_bitset per node per 1024-row batch; here the tree is a postfix program
// in the kernarg segment and the per-node bitmaps are 32-bit lane registers parked in LDS.
#include <string.h>

#include "ips_host.h"

namespace ips {

constexpr int kMaxLeaves = 16;
constexpr int kStackDepth = 8;

struct ProgLeaf {
  const void* data;
  int32_t encoding;  // ips_col_encoding
  int32_t bit_width;
  int32_t type;      // ips_type (PLAIN)
  int32_t op;
  int32_t n_consts;
  int32_t pad;
  uint64_t consts[16];
};

struct Program {
  int32_t n_nodes;
  int32_t n_leaves;
  int8_t kind[IPS_PROGRAM_MAX_NODES];   // ips_node_kind
  int8_t leaf[IPS_PROGRAM_MAX_NODES];   // index into leaves for kind == LEAF
  ProgLeaf leaves[kMaxLeaves];
};

// ---- PLAIN leaf: compare this lane's 32 consecutive rows (values staged in the padded row tile)
template <typename T>
__device__ __forceinline__ bool leaf_cmp(T x, int op, const ProgLeaf& lf) {
  T lit;
  __builtin_memcpy(&lit, &lf.consts[0], sizeof(T));
  switch (op) {
    case 0: return x == lit;
    case 1: return x < lit;
    case 2: return x <= lit;
    case 3: return x > lit;
    case 4: return x >= lit;
    default: {
      bool f = false;
      for (int j = 0; j < lf.n_consts; ++j) {
        T l2;
        __builtin_memcpy(&l2, &lf.consts[j], sizeof(T));
        f = f || (x == l2);
      }
      return f;
    }
  }
}

// 4-byte slots: the wave stages 2048 slots (8 KiB) in the row-tile layout, lane reads its 32.
template <typename T>
__device__ __forceinline__ uint32_t plain_leaf_4(const ProgLeaf& lf, uint32_t* lds32, int64_t tile,
                                                 int64_t n_rows, int lane) {
  const int64_t row_base = tile * kRowsPerTile;
  const uint32_t* page = reinterpret_cast<const uint32_t*>(lf.data);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int rho = 4 * (i * kWave + lane);
    int64_t valid = n_rows - (row_base + rho);
    u32x4 t = {0u, 0u, 0u, 0u};
    if (valid >= 4) {
      t = *reinterpret_cast<const u32x4*>(page + row_base + rho);
    } else {
      if (valid > 0) t.x = page[row_base + rho];
      if (valid > 1) t.y = page[row_base + rho + 1];
      if (valid > 2) t.z = page[row_base + rho + 2];
    }
    *reinterpret_cast<u32x4*>(lds32 + row_tile_dw(rho)) = t;
  }
  wave_lds_fence();
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
      if (leaf_cmp<T>(x, lf.op, lf)) bm |= 1u << (4 * i + e);
    }
  }
  wave_lds_fence();
  return bm;
}

// 8-byte slots: two half-tiles of 1024 rows; in half h lanes 32h..32h+31 own the staged rows.
template <typename T>
__device__ __forceinline__ uint32_t plain_leaf_8(const ProgLeaf& lf, uint32_t* lds32, int64_t tile,
                                                 int64_t n_rows, int lane) {
  const uint64_t* page = reinterpret_cast<const uint64_t*>(lf.data);
  uint32_t bm = 0;
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    const int64_t row_base = tile * kRowsPerTile + h * 1024;
    // stage: 1024 slots = 512 chunks of 16 bytes; row r at dword (r>>4)*36 + (r&15)*2
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int r = 2 * (i * kWave + lane);
      int64_t valid = n_rows - (row_base + r);
      u32x4 t = {0u, 0u, 0u, 0u};
      if (valid >= 2) {
        t = *reinterpret_cast<const u32x4*>(page + row_base + r);
      } else if (valid > 0) {
        u32x2 q = *reinterpret_cast<const u32x2*>(page + row_base + r);
        t.x = q.x; t.y = q.y;
      }
      *reinterpret_cast<u32x4*>(lds32 + (r >> 4) * kRowTileStrideDw + (r & 15) * 2) = t;
    }
    wave_lds_fence();
    if ((lane >> 5) == h) {
      const int g0 = 2 * (lane & 31);  // this lane's two 16-row groups
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
            if (leaf_cmp<T>(x, lf.op, lf)) bm |= 1u << (16 * g + 2 * i + e);
          }
        }
      }
    }
    wave_lds_fence();
  }
  return bm;
}

__device__ __forceinline__ uint32_t mask_rows(uint32_t bm, int64_t tile, int lane, int64_t n_rows) {
  int64_t valid = n_rows - (tile * kRowsPerTile + (int64_t)lane * 32);
  if (valid < 32) bm = valid <= 0 ? 0u : (bm & ((1u << valid) - 1u));
  return bm;
}

__global__ __launch_bounds__(kThreads) void program_kernel(Program prog, int64_t n_rows,
                                                           uint32_t* __restrict__ bitmap32) {
  // per wave: one tile region (row tile, also large enough for any plane tile) + the node stack
  __shared__ __attribute__((aligned(16)))
      uint32_t lds_all[kWavesPerBlock * (kRowTileBytes / 4 + kStackDepth * kWave)];
  const int lane = lane_id();
  const int wave = wave_id();
  uint32_t* lds32 = lds_all + wave * (kRowTileBytes / 4 + kStackDepth * kWave);
  uint32_t* stack = lds32 + kRowTileBytes / 4;

  const int64_t tiles = (n_rows + kRowsPerTile - 1) / kRowsPerTile;
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock;
  const int64_t bm_dwords = bitmap_dwords(n_rows);

  for (int64_t tile = (int64_t)blockIdx.x * kWavesPerBlock + wave; tile < tiles; tile += stride) {
    int sp = 0;
#pragma unroll 1
    for (int n = 0; n < prog.n_nodes; ++n) {
      const int kind = prog.kind[n];
      if (kind == IPS_NODE_LEAF) {
        const ProgLeaf& lf = prog.leaves[prog.leaf[n]];
        uint32_t bm;
        if (lf.encoding == IPS_COL_FLE) {
          const int w = lf.bit_width;
          const int64_t total_words = ((n_rows + 63) / 64) * w;
          u32x4 r[8];
          tile_load<8>(reinterpret_cast<const uint64_t*>(lf.data), tile, w, total_words, lane, r);
          tile_to_lds<8>(lds32, w, lane, r);
          wave_lds_fence();
          uint32_t sel;
          if (lf.op != 5) sel = pred_single_from_lds(lds32, w, lane, lf.op, (uint32_t)lf.consts[0]);
          else sel = pred_in_from_lds(lds32, w, lane, lf.consts, lf.n_consts);
          bm = finish_bitmap_dword(sel, tile, lane, n_rows);
          wave_lds_fence();
        } else {
          switch (lf.type) {
            case IPS_T_INT8: bm = plain_leaf_4<int8_t>(lf, lds32, tile, n_rows, lane); break;
            case IPS_T_INT16: bm = plain_leaf_4<int16_t>(lf, lds32, tile, n_rows, lane); break;
            case IPS_T_INT32: bm = plain_leaf_4<int32_t>(lf, lds32, tile, n_rows, lane); break;
            case IPS_T_FLOAT: bm = plain_leaf_4<float>(lf, lds32, tile, n_rows, lane); break;
            case IPS_T_INT64: bm = plain_leaf_8<int64_t>(lf, lds32, tile, n_rows, lane); break;
            default: bm = plain_leaf_8<double>(lf, lds32, tile, n_rows, lane); break;
          }
          bm = mask_rows(bm, tile, lane, n_rows);
        }
        stack[sp * kWave + lane] = bm;
        ++sp;
      } else {
        uint32_t b = stack[(sp - 1) * kWave + lane];
        uint32_t a = stack[(sp - 2) * kWave + lane];
        stack[(sp - 2) * kWave + lane] = kind == IPS_NODE_AND ? (a & b) : (a | b);
        --sp;
      }
    }
    const int64_t d = tile * 64 + lane;
    if (d < bm_dwords) bitmap32[d] = stack[lane];
  }
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

extern "C" ips_status ips_eval_program(const ips_node* nodes, int n_nodes, const ips_column* cols,
                                       int n_cols, int64_t n_rows, uint64_t* d_bitmap,
                                       ips_stream stream) {
  IPS_REQUIRE(nodes && n_nodes >= 1 && n_nodes <= IPS_PROGRAM_MAX_NODES,
              "ips_eval_program: n_nodes %d not in 1..%d", n_nodes, IPS_PROGRAM_MAX_NODES);
  IPS_REQUIRE(cols && n_cols >= 1 && n_cols <= IPS_PROGRAM_MAX_COLS,
              "ips_eval_program: n_cols %d not in 1..%d", n_cols, IPS_PROGRAM_MAX_COLS);
  IPS_REQUIRE(n_rows >= 0, "ips_eval_program: n_rows < 0");
  IPS_REQUIRE(n_rows == 0 || (d_bitmap && aligned16(d_bitmap)), "ips_eval_program: bitmap NULL or misaligned");
  Program prog;
  memset(&prog, 0, sizeof(prog));
  prog.n_nodes = n_nodes;
  int depth = 0, max_depth = 0;
  for (int i = 0; i < n_nodes; ++i) {
    const ips_node& nd = nodes[i];
    prog.kind[i] = (int8_t)nd.kind;
    if (nd.kind == IPS_NODE_LEAF) {
      IPS_REQUIRE(prog.n_leaves < kMaxLeaves, "ips_eval_program: more than %d leaves", kMaxLeaves);
      IPS_REQUIRE(nd.column >= 0 && nd.column < n_cols, "ips_eval_program: node %d: bad column", i);
      IPS_REQUIRE(nd.op >= IPS_OP_EQ && nd.op <= IPS_OP_IN, "ips_eval_program: node %d: bad op", i);
      IPS_REQUIRE(nd.n_consts >= 1 && nd.n_consts <= 16 && (nd.op == IPS_OP_IN || nd.n_consts == 1),
                  "ips_eval_program: node %d: bad constant count", i);
      const ips_column& c = cols[nd.column];
      IPS_REQUIRE(n_rows == 0 || (c.d_data && aligned16(c.d_data)),
                  "ips_eval_program: column %d: data NULL or misaligned", nd.column);
      ProgLeaf& lf = prog.leaves[prog.n_leaves];
      lf.data = c.d_data;
      lf.encoding = c.encoding;
      lf.bit_width = c.bit_width;
      lf.type = c.type;
      lf.op = nd.op;
      lf.n_consts = nd.n_consts;
      if (c.encoding == IPS_COL_FLE) {
        IPS_REQUIRE(c.bit_width >= 1 && c.bit_width <= 32, "ips_eval_program: column %d: bit width", nd.column);
        const uint64_t limit = c.bit_width == 32 ? 0xFFFFFFFFull : ((1ull << c.bit_width) - 1ull);
        for (int j = 0; j < nd.n_consts; ++j)
          IPS_REQUIRE(nd.consts[j] <= limit, "ips_eval_program: node %d: constant does not fit the bit width", i);
      } else {
        IPS_REQUIRE(c.encoding == IPS_COL_PLAIN, "ips_eval_program: column %d: bad encoding", nd.column);
        IPS_REQUIRE(c.type >= IPS_T_INT8 && c.type <= IPS_T_DOUBLE, "ips_eval_program: column %d: bad type", nd.column);
      }
      for (int j = 0; j < nd.n_consts; ++j) lf.consts[j] = nd.consts[j];
      prog.leaf[i] = (int8_t)prog.n_leaves++;
      ++depth;
    } else {
      IPS_REQUIRE(nd.kind == IPS_NODE_AND || nd.kind == IPS_NODE_OR, "ips_eval_program: node %d: bad kind", i);
      IPS_REQUIRE(depth >= 2, "ips_eval_program: node %d: stack underflow", i);
      --depth;
    }
    if (depth > max_depth) max_depth = depth;
  }
  IPS_REQUIRE(depth == 1, "ips_eval_program: program leaves %d bitmaps on the stack", depth);
  IPS_REQUIRE(max_depth <= kStackDepth, "ips_eval_program: tree deeper than %d", kStackDepth);
  if (n_rows == 0) return IPS_OK;
  return launch_program(prog, n_rows, reinterpret_cast<uint32_t*>(d_bitmap),
                        reinterpret_cast<hipStream_t>(stream));
}
