// ips_misc.hip -- the kernels that are not templated on the bit width: predicate-only FLE scan
// (any w at run time), PLAIN-page predicates, bitmap algebra, IntersectBitset expand, batch
// concatenation and the synthetic-column generator.  All HBM-bound streaming kernels.
#include <string.h>

#include "ips_host.h"

namespace ips {

// read-once column slots and the 16-byte tuple stream of assemble_tuples (232 -> 207 us); dev knob
// IPS_AUX_NT=0 turns the hints off
#ifndef IPS_AUX_NT
#define IPS_AUX_NT 1
#endif
template <typename T>
__device__ __forceinline__ T aux_load(const T* p) {
  return IPS_AUX_NT ? __builtin_nontemporal_load(p) : *p;
}
template <typename T>
__device__ __forceinline__ void aux_store(T* p, T v) {
  if (IPS_AUX_NT) __builtin_nontemporal_store(v, p); else *p = v;
}

// =============================================================================================
// PLAIN fixed-width pages: ParquetPlainEncoder::Eq/Lt/Le/Gt/Ge (parquet-common.h:197-250, int8
// :335-383, int16 :400-449).  bit = x OP literal (SQL order; the REFERENCE order is obtained by
// the caller swapping LT<->GT, LE<->GE).  Lane loads 16 bytes = RPL rows; the RPL ballots are
// re-interleaved into row order with a second round of ballots.
// =============================================================================================
template <typename T>
struct PlainLit {
  T v[16];
  int32_t n;
  int32_t combine;  // 0 set, 1 and-into, 2 or-into the bitmap
  int32_t join;     // 0 none; 1 / 2: AND / OR with (x op2 v2) in the same pass
  int32_t op2;
  T v2;
};

template <typename T>
__device__ __forceinline__ bool plain_cmp(T x, int op, const PlainLit<T>& lit) {
  switch (op) {
    case 0: return x == lit.v[0];
    case 1: return x < lit.v[0];
    case 2: return x <= lit.v[0];
    case 3: return x > lit.v[0];
    case 4: return x >= lit.v[0];
    default: {
      bool f = false;
      for (int j = 0; j < lit.n; ++j) f = f || (x == lit.v[j]);
      return f;
    }
  }
}

// T = compared type, S = slot type (int32_t for 4-byte slots, int64_t for 8-byte slots)
template <typename T, typename S>
__device__ __forceinline__ T slot_value(S raw) {
  if constexpr (sizeof(T) == sizeof(S)) {
    T t;
    __builtin_memcpy(&t, &raw, sizeof(T));
    return t;
  } else {
    return (T)raw;  // int8/int16: low bytes of the 4-byte slot, sign-extended by the cast
  }
}

// A wave takes 8 consecutive chunks (8 KiB of the page) per iteration: the eight 16-byte loads of a
// lane are issued back to back, and the 8*RPL bitmap words they produce leave as one contiguous
// store (128 or 256 bytes) instead of eight small ones.
constexpr int kPlainChunksPerTile = 8;

template <typename T, typename S>
__global__ __launch_bounds__(kThreads) void plain_pred_kernel(const S* __restrict__ page,
                                                              int64_t n_rows, int op,
                                                              PlainLit<T> lit,
                                                              uint64_t* __restrict__ bitmap) {
  constexpr int RPL = 16 / sizeof(S);  // rows per lane per load: 4 or 2
  constexpr int U = kPlainChunksPerTile;
  const int lane = lane_id();
  const int64_t wave_g = (int64_t)blockIdx.x * kWavesPerBlock + wave_id();
  const int64_t waves = (int64_t)gridDim.x * kWavesPerBlock;
  const int64_t rows_per_chunk = 64 * RPL;
  const int64_t rows_per_tile = rows_per_chunk * U;
  const int64_t tiles = (n_rows + rows_per_tile - 1) / rows_per_tile;
  const int64_t n_words = (n_rows + 63) / 64;

  for (int64_t tile = wave_g; tile < tiles; tile += waves) {
    const int64_t tile_row0 = tile * rows_per_tile;
    S raw[U][RPL];
    if (tile_row0 + rows_per_tile <= n_rows) {  // wave-uniform: full tile
#pragma unroll
      for (int u = 0; u < U; ++u) {
        u32x4 t = stream_load(reinterpret_cast<const u32x4*>(page + tile_row0 + u * rows_per_chunk + lane * RPL));
        __builtin_memcpy(raw[u], &t, 16);
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t row0 = tile_row0 + u * rows_per_chunk + (int64_t)lane * RPL;
#pragma unroll
        for (int e = 0; e < RPL; ++e) raw[u][e] = row0 + e < n_rows ? page[row0 + e] : (S)0;
      }
    }
    uint64_t mine = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row0 = tile_row0 + u * rows_per_chunk + (int64_t)lane * RPL;
      uint64_t m[RPL];
#pragma unroll
      for (int e = 0; e < RPL; ++e) {
        const T x = slot_value<T, S>(raw[u][e]);
        bool b = plain_cmp<T>(x, op, lit);
        if (lit.join != 0) {
          PlainLit<T> l2;
          l2.v[0] = lit.v2;
          l2.n = 1;
          const bool b2 = plain_cmp<T>(x, lit.op2, l2);
          b = lit.join == 1 ? (b && b2) : (b || b2);
        }
        b = b && (row0 + e < n_rows);
        m[e] = __builtin_amdgcn_ballot_w64(b);  // bit l <-> row RPL*l + e of the chunk
      }
      // row 64*j + t of the chunk sits in ballot (t % RPL) at bit (64*j + t) / RPL
      uint64_t sel;
      if (RPL == 4) sel = (lane & 2) ? ((lane & 1) ? m[3] : m[2]) : ((lane & 1) ? m[1] : m[0]);
      else sel = (lane & 1) ? m[1] : m[0];
#pragma unroll
      for (int j = 0; j < RPL; ++j) {
        int src = (64 * j + lane) / RPL;
        uint64_t word = __builtin_amdgcn_ballot_w64(((sel >> src) & 1ull) != 0ull);
        if (lane == u * RPL + j) mine = word;
      }
    }
    const int64_t wi = tile * (U * RPL) + lane;
    if (lane < U * RPL && wi < n_words) {
      if (lit.combine == 1) mine &= bitmap[wi];
      else if (lit.combine == 2) mine |= bitmap[wi];
      IPS_BITMAP_STORE(bitmap + wi, mine);
    }
  }
}

// Fused scan of a PLAIN page: the predicate's bitmap AND the selected rows' slots in one pass over
// the page (EvalSimplePredicates + ReadValue(skip) on the same column, hdfs-parquet-scanner.cc:
// 1837-1865, 1006-1027; parquet-common.h:186-250).  A wave takes one 2048-row batch per iteration
// (8 or 16 chunks of 16 bytes per lane); per chunk the lanes' selected slots are ranked with a DPP
// prefix sum and stored straight to the batch, so row order is kept: chunk, lane, element.
template <typename T, typename S>
__global__ __launch_bounds__(kThreads) void plain_scan_kernel(const S* __restrict__ page,
                                                              int64_t n_rows, int op,
                                                              PlainLit<T> lit,
                                                              uint64_t* __restrict__ bitmap,
                                                              S* __restrict__ batch_values,
                                                              uint32_t* __restrict__ batch_counts) {
  constexpr int RPL = 16 / sizeof(S);           // rows per lane per load: 4 or 2
  constexpr int U = kRowsPerTile / (64 * RPL);  // chunks per batch: 8 or 16
  const int lane = lane_id();
  const int64_t waves = (int64_t)gridDim.x * kWavesPerBlock;
  const int64_t rows_per_chunk = 64 * RPL;
  const int64_t n_batches = (n_rows + kRowsPerTile - 1) / kRowsPerTile;
  const int64_t n_words = (n_rows + 63) / 64;

  for (int64_t batch = (int64_t)blockIdx.x * kWavesPerBlock + wave_id(); batch < n_batches;
       batch += waves) {
    const int64_t batch_row0 = batch * kRowsPerTile;
    uint64_t mine = 0;
    uint32_t base = 0;  // selected rows of the batch so far (wave-uniform)
    S* dst = batch_values + batch_row0;
    constexpr int UH = 8;  // chunks per round: 8 KiB in flight per wave, 32 page registers per lane
#pragma unroll 1
    for (int h = 0; h < U / UH; ++h) {
      const int64_t round_row0 = batch_row0 + (int64_t)h * UH * rows_per_chunk;
      S raw[UH][RPL];
      if (round_row0 + UH * rows_per_chunk <= n_rows) {  // wave-uniform: full round
#pragma unroll
        for (int u = 0; u < UH; ++u) {
          u32x4 t = stream_load(reinterpret_cast<const u32x4*>(page + round_row0 + u * rows_per_chunk + lane * RPL));
          __builtin_memcpy(raw[u], &t, 16);
        }
      } else {
#pragma unroll
        for (int u = 0; u < UH; ++u) {
          const int64_t row0 = round_row0 + u * rows_per_chunk + (int64_t)lane * RPL;
#pragma unroll
          for (int e = 0; e < RPL; ++e) raw[u][e] = row0 + e < n_rows ? page[row0 + e] : (S)0;
        }
      }
#pragma unroll
      for (int u = 0; u < UH; ++u) {
        const int64_t row0 = round_row0 + u * rows_per_chunk + (int64_t)lane * RPL;
        uint64_t m[RPL];
        bool sel_e[RPL];
#pragma unroll
        for (int e = 0; e < RPL; ++e) {
          const T x = slot_value<T, S>(raw[u][e]);
          bool b = plain_cmp<T>(x, op, lit);
          if (lit.join != 0) {
            PlainLit<T> l2;
            l2.v[0] = lit.v2;
            l2.n = 1;
            const bool b2 = plain_cmp<T>(x, lit.op2, l2);
            b = lit.join == 1 ? (b && b2) : (b || b2);
          }
          b = b && (row0 + e < n_rows);
          sel_e[e] = b;
          m[e] = __builtin_amdgcn_ballot_w64(b);  // bit l <-> row RPL*l + e of the chunk
        }
        // bitmap words of the chunk (row 64*j + t sits in ballot (t % RPL) at bit (64*j + t) / RPL)
        uint64_t sel;
        if (RPL == 4) sel = (lane & 2) ? ((lane & 1) ? m[3] : m[2]) : ((lane & 1) ? m[1] : m[0]);
        else sel = (lane & 1) ? m[1] : m[0];
#pragma unroll
        for (int j = 0; j < RPL; ++j) {
          int src = (64 * j + lane) / RPL;
          uint64_t word = __builtin_amdgcn_ballot_w64(((sel >> src) & 1ull) != 0ull);
          if (lane == (h * UH + u) * RPL + j) mine = word;
        }
        // selected slots of the chunk, in row order.  The ballots already hold everything a prefix
        // sum would compute: rows selected in lower lanes = mbcnt of each ballot, chunk total =
        // scalar popcounts.
        if ((m[0] | m[1] | (RPL == 4 ? (m[RPL - 2] | m[RPL - 1]) : 0ull)) != 0ull) {  // wave-uniform
          uint32_t P = base;
          uint32_t total = 0;
#pragma unroll
          for (int e = 0; e < RPL; ++e) {
            P += __builtin_amdgcn_mbcnt_hi((uint32_t)(m[e] >> 32),
                                           __builtin_amdgcn_mbcnt_lo((uint32_t)m[e], 0u));
            total += (uint32_t)__builtin_popcountll(m[e]);
          }
#pragma unroll
          for (int e = 0; e < RPL; ++e)
            if (sel_e[e]) dst[P++] = raw[u][e];
          base += total;
        }
      }
    }
    const int64_t wi = batch * 32 + lane;
    if (lane < 32 && wi < n_words) IPS_BITMAP_STORE(bitmap + wi, mine);
    if (lane == 0) batch_counts[batch] = base;
  }
}

template <typename T, typename S>
static ips_status launch_plain_scan_t(const void* page, int64_t n_rows, int op, const void* literals,
                                      int n_literals, int join, int op2, const void* literal2,
                                      uint64_t* bitmap, void* batch_values, uint32_t* batch_counts,
                                      hipStream_t s) {
  PlainLit<T> lit;
  lit.n = n_literals;
  lit.combine = 0;
  lit.join = join;
  lit.op2 = op2;
  lit.v2 = literal2 ? *reinterpret_cast<const T*>(literal2) : T();
  for (int i = 0; i < 16; ++i) lit.v[i] = i < n_literals ? reinterpret_cast<const T*>(literals)[i] : T();
  const int64_t n_batches = (n_rows + kRowsPerTile - 1) / kRowsPerTile;
  int64_t want = (n_batches + kWavesPerBlock - 1) / kWavesPerBlock;
  int64_t cap = (int64_t)device_cus() * 4 * grid_mult();
  int grid = (int)(want < cap ? want : cap);
  if (grid <= 0) return IPS_ERR_HIP;
  hipLaunchKernelGGL((plain_scan_kernel<T, S>), dim3(grid), dim3(kThreads), 0, s,
                     reinterpret_cast<const S*>(page), n_rows, op, lit, bitmap,
                     reinterpret_cast<S*>(batch_values), batch_counts);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

ips_status launch_plain_scan(int type, const void* page, int64_t n_rows, int op, const void* literals,
                             int n_literals, int join, int op2, const void* literal2,
                             uint64_t* bitmap, void* batch_values, uint32_t* batch_counts,
                             hipStream_t s) {
#define IPS_PS(T, S)                                                                              \
  return launch_plain_scan_t<T, S>(page, n_rows, op, literals, n_literals, join, op2, literal2, \
                                   bitmap, batch_values, batch_counts, s)
  switch (type) {
    case IPS_T_INT8: IPS_PS(int8_t, int32_t);
    case IPS_T_INT16: IPS_PS(int16_t, int32_t);
    case IPS_T_INT32: IPS_PS(int32_t, int32_t);
    case IPS_T_INT64: IPS_PS(int64_t, int64_t);
    case IPS_T_FLOAT: IPS_PS(float, int32_t);
    case IPS_T_DOUBLE: IPS_PS(double, int64_t);
  }
#undef IPS_PS
  set_error("plain_scan: bad type %d", type);
  return IPS_ERR_INVALID_ARG;
}

template <typename T, typename S>
static ips_status launch_plain_t(const void* page, int64_t n_rows, int op, const void* literals,
                                 int n_literals, uint64_t* bitmap, hipStream_t s, int combine,
                                 int join, int op2, const void* literal2) {
  PlainLit<T> lit;
  lit.n = n_literals;
  lit.combine = combine;
  lit.join = join;
  lit.op2 = op2;
  lit.v2 = literal2 ? *reinterpret_cast<const T*>(literal2) : T();
  for (int i = 0; i < 16; ++i) lit.v[i] = i < n_literals ? reinterpret_cast<const T*>(literals)[i] : T();
  constexpr int RPL = 16 / sizeof(S);
  int64_t tiles = (n_rows + 64 * RPL * kPlainChunksPerTile - 1) / (64 * RPL * kPlainChunksPerTile);
  int64_t want = (tiles + kWavesPerBlock - 1) / kWavesPerBlock;
  int64_t cap = (int64_t)device_cus() * 8 * grid_mult();
  int grid = (int)(want < cap ? want : cap);
  if (grid <= 0) return IPS_ERR_HIP;
  hipLaunchKernelGGL((plain_pred_kernel<T, S>), dim3(grid), dim3(kThreads), 0, s,
                     reinterpret_cast<const S*>(page), n_rows, op, lit, bitmap);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

ips_status launch_plain_pred(int type, const void* page, int64_t n_rows, int op,
                             const void* literals, int n_literals, uint64_t* bitmap,
                             hipStream_t s, int combine, int join, int op2, const void* literal2) {
#define IPS_PL(T, S) \
  return launch_plain_t<T, S>(page, n_rows, op, literals, n_literals, bitmap, s, combine, join, op2, literal2)
  switch (type) {
    case IPS_T_INT8: IPS_PL(int8_t, int32_t);
    case IPS_T_INT16: IPS_PL(int16_t, int32_t);
    case IPS_T_INT32: IPS_PL(int32_t, int32_t);
    case IPS_T_INT64: IPS_PL(int64_t, int64_t);
    case IPS_T_FLOAT: IPS_PL(float, int32_t);
    case IPS_T_DOUBLE: IPS_PL(double, int64_t);
  }
#undef IPS_PL
  set_error("plain_pred: bad type %d", type);
  return IPS_ERR_INVALID_ARG;
}

// =============================================================================================
// Late materialisation on a PLAIN page: ReadValue(skip) -> ParquetPlainEncoder::Decode(buffer,
// size, &val, skip_rows) per selected row (parquet-common.h:186-190, hdfs-parquet-scanner.cc:
// 1006-1027).  One wave per 2048-row batch: lane l owns rows 32l..32l+31 of the batch (one bitmap
// dword), a DPP prefix sum of the popcounts gives its first output slot, then it walks its set
// bits four at a time -- four independent slot loads in flight -- and stores them in row order.
// Same batch layout as ips_fle_select.
// =============================================================================================
constexpr uint32_t kPlainSelectStreamMin = 64;  // selected rows per 2048-row batch (3 %)

template <typename S>
__global__ __launch_bounds__(kThreads) void plain_select_kernel(
    const S* __restrict__ page, int64_t n_rows, const uint32_t* __restrict__ bitmap32,
    S* __restrict__ batch_values, uint32_t* __restrict__ batch_counts) {
  const int lane = lane_id();
  const int64_t n_batches = (n_rows + kRowsPerTile - 1) / kRowsPerTile;
  const int64_t bm_dwords = bitmap_dwords(n_rows);
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t batch = (int64_t)blockIdx.x * kWavesPerBlock + wave_id(); batch < n_batches;
       batch += stride) {
    const int64_t d = batch * 64 + lane;
    uint32_t m = d < bm_dwords ? bitmap32[d] : 0u;
    const int64_t row0 = batch * kRowsPerTile + (int64_t)lane * 32;
    const int64_t valid = n_rows - row0;
    if (valid < 32) m = valid <= 0 ? 0u : (m & ((1u << valid) - 1u));
    const uint32_t mine = (uint32_t)__builtin_popcount(m);
    const uint32_t incl = wave_inclusive_scan(mine);
    const uint32_t count = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    uint32_t P = incl - mine;
    const S* src = page + row0;
    S* dst = batch_values + batch * kRowsPerTile;
    // Above a few per cent selectivity nearly every 128-byte line of the batch holds a selected
    // row (10 %: 81 % of the lines of an 8-byte column), and fetching them through scattered
    // 8-byte loads costs far more than their bytes: stream the whole batch with coalesced 16-byte
    // loads instead and rank the selected slots with the ballots of their bitmap bits (the
    // materialisation half of plain_scan_kernel).  Wave-uniform choice per batch.
    constexpr int RPL = 16 / (int)sizeof(S);
    constexpr int U = kRowsPerTile / (64 * RPL);
    constexpr int UH = 8;
    const int64_t batch_row0 = batch * kRowsPerTile;
    if (count >= kPlainSelectStreamMin && batch_row0 + kRowsPerTile <= n_rows) {
      uint32_t base = 0;
#pragma unroll 1
      for (int h = 0; h < U / UH; ++h) {
        S raw[UH][RPL];
        uint32_t bits[UH];
#pragma unroll
        for (int u = 0; u < UH; ++u) {
          const int r = (h * UH + u) * 64 * RPL + lane * RPL;  // first row of this lane's load
          u32x4 t = stream_load(reinterpret_cast<const u32x4*>(page + batch_row0 + r));
          __builtin_memcpy(raw[u], &t, 16);
          bits[u] = bitmap32[batch * 64 + (r >> 5)] >> (r & 31);
        }
#pragma unroll
        for (int u = 0; u < UH; ++u) {
          uint64_t mm[RPL];
          bool sel_e[RPL];
#pragma unroll
          for (int e = 0; e < RPL; ++e) {
            sel_e[e] = ((bits[u] >> e) & 1u) != 0u;
            mm[e] = __builtin_amdgcn_ballot_w64(sel_e[e]);
          }
          uint64_t any = mm[0];
#pragma unroll
          for (int e = 1; e < RPL; ++e) any |= mm[e];
          if (any != 0ull) {  // wave-uniform
            uint32_t Q = base;
            uint32_t total = 0;
#pragma unroll
            for (int e = 0; e < RPL; ++e) {
              Q += __builtin_amdgcn_mbcnt_hi((uint32_t)(mm[e] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm[e], 0u));
              total += (uint32_t)__builtin_popcountll(mm[e]);
            }
#pragma unroll
            for (int e = 0; e < RPL; ++e)
              if (sel_e[e]) dst[Q++] = raw[u][e];
            base += total;
          }
        }
      }
      if (lane == 0) batch_counts[batch] = count;
      continue;
    }
    while (__builtin_amdgcn_ballot_w64(m != 0u) != 0ull) {
      S x[4];
      bool ok[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        ok[e] = m != 0u;
        x[e] = ok[e] ? src[__builtin_ctz(m)] : (S)0;
        m &= m - 1u;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (ok[e]) dst[P++] = x[e];
      }
    }
    if (lane == 0) batch_counts[batch] = count;
  }
}

ips_status launch_plain_select(int stride_bytes, const void* page, int64_t n_rows,
                               const uint64_t* bitmap, void* batch_values, uint32_t* batch_counts,
                               hipStream_t s) {
  const int64_t n_batches = (n_rows + kRowsPerTile - 1) / kRowsPerTile;
  if (n_batches <= 0) return IPS_OK;
  int64_t want = (n_batches + kWavesPerBlock - 1) / kWavesPerBlock;
  int64_t cap = (int64_t)device_cus() * 8 * grid_mult();
  const int grid = (int)(want < cap ? want : cap);
  if (stride_bytes == 4)
    hipLaunchKernelGGL((plain_select_kernel<uint32_t>), dim3(grid), dim3(kThreads), 0, s,
                       reinterpret_cast<const uint32_t*>(page), n_rows,
                       reinterpret_cast<const uint32_t*>(bitmap),
                       reinterpret_cast<uint32_t*>(batch_values), batch_counts);
  else
    hipLaunchKernelGGL((plain_select_kernel<uint64_t>), dim3(grid), dim3(kThreads), 0, s,
                       reinterpret_cast<const uint64_t*>(page), n_rows,
                       reinterpret_cast<const uint32_t*>(bitmap),
                       reinterpret_cast<uint64_t*>(batch_values), batch_counts);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

// =============================================================================================
// Bitmap algebra: AndOperate / OrOperate (simple-predicates.h:145-163), resize(n, value),
// count().
// =============================================================================================
template <int OP>
__global__ void bitmap_binop_kernel(uint64_t* __restrict__ a, const uint64_t* __restrict__ b,
                                    int64_t n_words) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words;
       i += (int64_t)gridDim.x * blockDim.x)
    a[i] = OP == 0 ? (a[i] & b[i]) : (a[i] | b[i]);
}

__global__ void bitmap_fill_kernel(uint64_t* __restrict__ a, int64_t n_rows, int value) {
  const int64_t n_words = (n_rows + 63) / 64;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words;
       i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t w = value ? ~0ull : 0ull;
    int64_t valid = n_rows - i * 64;
    if (valid < 64) w &= (1ull << valid) - 1ull;
    a[i] = w;
  }
}

__global__ __launch_bounds__(256) void bitmap_count_kernel(const uint64_t* __restrict__ a,
                                                           int64_t n_rows,
                                                           unsigned long long* __restrict__ count) {
  __shared__ unsigned long long part[4];
  const int64_t n_words = (n_rows + 63) / 64;
  const int64_t n_pairs = n_words / 2;  // 16-byte loads; the last (maybe partial) words go scalar
  unsigned long long c = 0;
  const ulonglong2* a2 = reinterpret_cast<const ulonglong2*>(a);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pairs;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (2 * i + 2 < n_words) {  // both words are full
      ulonglong2 w = a2[i];
      c += __builtin_popcountll(w.x) + __builtin_popcountll(w.y);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < 4) {  // the <= 3 trailing words, with the row mask
    const int64_t first = n_pairs >= 1 ? 2 * (n_pairs - 1) : 0;
    const int64_t i = first + threadIdx.x;
    if (i < n_words && (i >= 2 * n_pairs || 2 * (i / 2) + 2 >= n_words)) {
      uint64_t w = a[i];
      int64_t valid = n_rows - i * 64;
      if (valid < 64) w &= (1ull << valid) - 1ull;
      c += __builtin_popcountll(w);
    }
  }
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
  if (lane_id() == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    c = part[0] + part[1] + part[2] + part[3];
    if (c) atomicAdd(count, c);
  }
}

static int small_grid(int64_t items, int threads) {
  int64_t want = (items + threads - 1) / threads;
  int64_t cap = (int64_t)device_cus() * 8;
  if (want < 1) want = 1;
  return (int)(want < cap ? want : cap);
}

ips_status launch_bitmap_binop(int op, uint64_t* a, const uint64_t* b, int64_t n_words,
                               hipStream_t s) {
  if (n_words <= 0) return IPS_OK;
  int grid = small_grid(n_words, 256);
  if (op == 0) hipLaunchKernelGGL(bitmap_binop_kernel<0>, dim3(grid), dim3(256), 0, s, a, b, n_words);
  else hipLaunchKernelGGL(bitmap_binop_kernel<1>, dim3(grid), dim3(256), 0, s, a, b, n_words);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

ips_status launch_bitmap_fill(uint64_t* a, int64_t n_rows, int value, hipStream_t s) {
  if (n_rows <= 0) return IPS_OK;
  hipLaunchKernelGGL(bitmap_fill_kernel, dim3(small_grid((n_rows + 63) / 64, 256)), dim3(256), 0,
                     s, a, n_rows, value);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

ips_status launch_bitmap_count(const uint64_t* a, int64_t n_rows, int64_t* count, hipStream_t s) {
  IPS_HIP_TRY(hipMemsetAsync(count, 0, 8, s));
  if (n_rows <= 0) return IPS_OK;
  int grid = small_grid((n_rows + 127) / 128, 256);
  if (grid > device_cus() * 2) grid = device_cus() * 2;  // one atomic per block
  hipLaunchKernelGGL(bitmap_count_kernel, dim3(grid), dim3(256), 0, s, a, n_rows,
                     reinterpret_cast<unsigned long long*>(count));
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

// =============================================================================================
// Device-wide exclusive scan of small per-item counts (three launches), shared by
//  - IntersectBitset expand: items = bitmap words, count = popcount      (scanner.cc:326-331)
//  - batch concatenation:   items = batches, count = batch_counts[b]      (scanner.cc:1151-1181)
// A block handles kScanItems consecutive items; block totals are scanned by one block.
// =============================================================================================
constexpr int kScanThreads = 256;
constexpr int kScanPerThread = 4;
constexpr int kScanItems = kScanThreads * kScanPerThread;

struct PopcItems {
  const uint64_t* words;
  int64_t n_rows;
  __device__ __forceinline__ uint32_t operator()(int64_t i) const {
    uint64_t w = words[i];
    int64_t valid = n_rows - i * 64;
    if (valid < 64) w &= (1ull << valid) - 1ull;
    return __builtin_popcountll(w);
  }
};
struct ArrayItems {
  const uint32_t* counts;
  __device__ __forceinline__ uint32_t operator()(int64_t i) const { return counts[i]; }
};

// exclusive prefix of this thread's value inside the block + block total (all threads)
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* total) {
  __shared__ uint32_t wave_sums[kScanThreads / kWave];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  uint32_t incl = v;
  for (int off = 1; off < kWave; off <<= 1) {
    uint32_t t = __shfl_up(incl, off);
    if (lane >= off) incl += t;
  }
  __syncthreads();  // wave_sums may still be read from a previous call
  if (lane == kWave - 1) wave_sums[wave] = incl;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (int i = 0; i < kScanThreads / kWave; ++i) {
    if (i < wave) base += wave_sums[i];
    tot += wave_sums[i];
  }
  *total = tot;
  return base + incl - v;
}

template <typename Items>
__global__ __launch_bounds__(kScanThreads) void scan_block_totals_kernel(Items items, int64_t n,
                                                                         uint64_t* block_totals) {
  const int64_t base = (int64_t)blockIdx.x * kScanItems + (int64_t)threadIdx.x * kScanPerThread;
  uint32_t v = 0;
#pragma unroll
  for (int e = 0; e < kScanPerThread; ++e)
    if (base + e < n) v += items(base + e);
  uint32_t total;
  block_exclusive_scan(v, &total);
  if (threadIdx.x == 0) block_totals[blockIdx.x] = total;
}

// one block: block_totals[0..nb) -> exclusive prefixes in place; grand total -> *grand
__global__ __launch_bounds__(kScanThreads) void scan_of_totals_kernel(uint64_t* block_totals,
                                                                      int64_t nb,
                                                                      int64_t* grand) {
  __shared__ uint64_t part[kScanThreads];
  const int64_t per = (nb + kScanThreads - 1) / kScanThreads;
  const int64_t lo = (int64_t)threadIdx.x * per;
  const int64_t hi = lo + per < nb ? lo + per : nb;
  uint64_t sum = 0;
  for (int64_t i = lo; i < hi; ++i) sum += block_totals[i];
  part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint64_t run = 0;
    for (int i = 0; i < kScanThreads; ++i) {
      uint64_t t = part[i];
      part[i] = run;
      run += t;
    }
    if (grand) *grand = (int64_t)run;
  }
  __syncthreads();
  uint64_t run = part[threadIdx.x];
  for (int64_t i = lo; i < hi; ++i) {
    uint64_t t = block_totals[i];
    block_totals[i] = run;
    run += t;
  }
}

// (IntersectBitset -- expand -- and its inverse, compress, live in ips_rank.hip.)

// exclusive prefix popcount per bitmap word (rank support for the NULL flags of selected rows)
__global__ __launch_bounds__(kScanThreads) void word_prefix_kernel(
    const uint64_t* __restrict__ words, int64_t n_bits, const uint64_t* __restrict__ block_offsets,
    uint64_t* __restrict__ prefix) {
  const int64_t n_words = (n_bits + 63) / 64;
  const int64_t base = (int64_t)blockIdx.x * kScanItems + (int64_t)threadIdx.x * kScanPerThread;
  PopcItems items{words, n_bits};
  uint32_t pc[kScanPerThread];
  uint32_t v = 0;
#pragma unroll
  for (int e = 0; e < kScanPerThread; ++e) {
    pc[e] = base + e < n_words ? items(base + e) : 0u;
    v += pc[e];
  }
  uint32_t total;
  uint64_t off = block_offsets[blockIdx.x] + block_exclusive_scan(v, &total);
#pragma unroll
  for (int e = 0; e < kScanPerThread; ++e) {
    if (base + e < n_words) prefix[base + e] = off;
    off += pc[e];
  }
}

size_t scan_workspace_bytes(int64_t items) {
  return (size_t)((items + kScanItems - 1) / kScanItems + 1) * 8;
}

// ---- batch concatenation --------------------------------------------------------------------
// Exclusive offset of every batch among all selected rows (block scan of the counts on top of the
// scanned block totals); the per-batch kernels below then run one wave per batch.
__global__ __launch_bounds__(kScanThreads) void batch_offsets_kernel(
    const uint32_t* __restrict__ counts, int64_t n_batches,
    const uint64_t* __restrict__ block_offsets, uint64_t* __restrict__ batch_off) {
  const int64_t base = (int64_t)blockIdx.x * kScanItems + (int64_t)threadIdx.x * kScanPerThread;
  uint32_t c[kScanPerThread];
  uint32_t v = 0;
#pragma unroll
  for (int e = 0; e < kScanPerThread; ++e) {
    c[e] = base + e < n_batches ? counts[base + e] : 0u;
    v += c[e];
  }
  uint32_t total;
  uint64_t off = block_offsets[blockIdx.x] + block_exclusive_scan(v, &total);
#pragma unroll
  for (int e = 0; e < kScanPerThread; ++e) {
    if (base + e < n_batches) batch_off[base + e] = off;
    off += c[e];
  }
}

template <typename V>
__global__ __launch_bounds__(kThreads) void batches_compact_kernel(
    const V* __restrict__ batch_values, const uint32_t* __restrict__ counts, int64_t n_batches,
    const uint64_t* __restrict__ batch_off, V* __restrict__ dense) {
  const int lane = lane_id();
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t batch = (int64_t)blockIdx.x * kWavesPerBlock + wave_id(); batch < n_batches;
       batch += stride) {
    const uint32_t cnt = counts[batch];
    const V* src = batch_values + batch * kRowsPerTile;
    V* dst = dense + batch_off[batch];
    uint32_t i = lane;
    for (; i + 3 * kWave < cnt; i += 4 * kWave) {  // four independent loads in flight per lane
      V x0 = src[i], x1 = src[i + kWave], x2 = src[i + 2 * kWave], x3 = src[i + 3 * kWave];
      dst[i] = x0; dst[i + kWave] = x1; dst[i + 2 * kWave] = x2; dst[i + 3 * kWave] = x3;
    }
    for (; i < cnt; i += kWave) dst[i] = src[i];  // (nt here: 51 -> 88 us, the 4-byte stores need L2)
  }
}

size_t batches_workspace_bytes(int64_t n_batches) {
  return scan_workspace_bytes(n_batches) + (size_t)(n_batches > 0 ? n_batches : 0) * 8 + 64;
}

// totals + per-batch offsets into the workspace; *total = number of selected rows
static ips_status launch_batch_offsets(const uint32_t* counts, int64_t n_batches, int64_t* total,
                                       void* workspace, uint64_t** batch_off, hipStream_t s) {
  const int64_t nb = (n_batches + kScanItems - 1) / kScanItems;
  uint64_t* totals = reinterpret_cast<uint64_t*>(workspace);
  *batch_off = reinterpret_cast<uint64_t*>(reinterpret_cast<uint8_t*>(workspace) +
                                           ((scan_workspace_bytes(n_batches) + 63) & ~(size_t)63));
  hipLaunchKernelGGL((scan_block_totals_kernel<ArrayItems>), dim3((unsigned)nb),
                     dim3(kScanThreads), 0, s, ArrayItems{counts}, n_batches, totals);
  hipLaunchKernelGGL(scan_of_totals_kernel, dim3(1), dim3(kScanThreads), 0, s, totals, nb, total);
  hipLaunchKernelGGL(batch_offsets_kernel, dim3((unsigned)nb), dim3(kScanThreads), 0, s, counts,
                     n_batches, totals, *batch_off);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

static int per_batch_grid(int64_t n_batches) {
  int64_t want = (n_batches + kWavesPerBlock - 1) / kWavesPerBlock;
  int64_t cap = (int64_t)device_cus() * 10 * grid_mult();
  return (int)(want < cap ? want : cap);
}

ips_status launch_batches_compact(const void* batch_values, const uint32_t* counts,
                                  int64_t n_batches, int value_width, void* dense,
                                  int64_t* total, void* workspace, hipStream_t s) {
  if (n_batches <= 0) {
    IPS_HIP_TRY(hipMemsetAsync(total, 0, 8, s));
    return IPS_OK;
  }
  uint64_t* batch_off = nullptr;
  ips_status st = launch_batch_offsets(counts, n_batches, total, workspace, &batch_off, s);
  if (st != IPS_OK) return st;
  const int grid = per_batch_grid(n_batches);
  if (value_width == 4)
    hipLaunchKernelGGL((batches_compact_kernel<uint32_t>), dim3(grid), dim3(kThreads), 0, s,
                       reinterpret_cast<const uint32_t*>(batch_values), counts, n_batches,
                       batch_off, reinterpret_cast<uint32_t*>(dense));
  else
    hipLaunchKernelGGL((batches_compact_kernel<uint64_t>), dim3(grid), dim3(kThreads), 0, s,
                       reinterpret_cast<const uint64_t*>(batch_values), counts, n_batches,
                       batch_off, reinterpret_cast<uint64_t*>(dense));
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

// ---- tuple assembly -------------------------------------------------------------------------
struct TupleCols {
  const void* values[IPS_TUPLE_MAX_COLS];       // batches (REQUIRED) or dense values (OPTIONAL)
  const uint64_t* flags[IPS_TUPLE_MAX_COLS];    // OPTIONAL: non-NULL flag per selected row
  const uint64_t* prefix[IPS_TUPLE_MAX_COLS];   // OPTIONAL: set flags before each flag word
  int32_t width[IPS_TUPLE_MAX_COLS];
  int32_t offset[IPS_TUPLE_MAX_COLS];
  int32_t null_byte[IPS_TUPLE_MAX_COLS];
  int32_t null_mask[IPS_TUPLE_MAX_COLS];
  int32_t n_cols;
  int32_t tuple_size;
  const uint8_t* d_template;                    // wide tuples: device copy of the template tuple
  uint32_t tmpl[32];                            // tuples <= 128 bytes: the template itself
};

constexpr int kTupleImageMax = 128;  // tuples up to this size are staged through LDS

// Row-major tuples from per-column batches (AssembleRows, hdfs-parquet-scanner.cc:1151-1181).
// One wave per batch, 64 tuples per round: lane i builds tuple i of the round in the wave's LDS
// image (every column read is a coalesced 256/512-byte load of the batch), then the image -- 64
// consecutive tuples = one contiguous piece of the output -- leaves as dword stores of consecutive
// lanes.  Every tuple starts as a copy of the template tuple (InitTuple()).  IMAGE = false (tuples
// wider than 128 bytes or not a multiple of 4): each lane copies the template and writes its slots
// straight to HBM.
// FAST: every column REQUIRED, 4 bytes wide and 4-byte aligned in the tuple (the common layout):
// the column loop is unrolled with constant indices, all slot loads of a round are issued before
// the first one is consumed.
template <bool IMAGE, bool FAST>
__global__ __launch_bounds__(kThreads) void assemble_tuples_kernel(
    TupleCols tc, const uint32_t* __restrict__ counts, int64_t n_batches,
    const uint64_t* __restrict__ batch_off, uint8_t* __restrict__ tuples) {
  extern __shared__ __attribute__((aligned(16))) uint8_t image_all[];  // IMAGE: 4 waves x 64 tuples
  const int lane = lane_id();
  const int wave = wave_id();
  const int ts = tc.tuple_size;
  uint8_t* image = image_all + (IMAGE ? wave * kWave * ts : 0);
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t batch = (int64_t)blockIdx.x * kWavesPerBlock + wave; batch < n_batches;
       batch += stride) {
    const uint32_t cnt = counts[batch];
    const uint64_t first = batch_off[batch];
    for (uint32_t i0 = 0; i0 < cnt; i0 += kWave) {
      const uint32_t i = i0 + lane;
      const uint32_t in_round = cnt - i0 < (uint32_t)kWave ? cnt - i0 : (uint32_t)kWave;
      uint8_t* t = IMAGE ? image + lane * ts : tuples + (first + i) * (uint64_t)ts;
      if (IMAGE) {
#pragma unroll
        for (int d = 0; d < kTupleImageMax / 4; ++d)
          if (4 * d < ts) reinterpret_cast<uint32_t*>(t)[d] = tc.tmpl[d];
      } else if (i < cnt) {
        for (int d = 0; d < ts; ++d) t[d] = tc.d_template[d];
      }
      if (FAST) {
        uint32_t x[IPS_TUPLE_MAX_COLS];
        const uint64_t src = (uint64_t)batch * kRowsPerTile + i;
#pragma unroll
        for (int col = 0; col < IPS_TUPLE_MAX_COLS; ++col)
          x[col] = (col < tc.n_cols && i < cnt) ? aux_load(reinterpret_cast<const uint32_t*>(tc.values[col]) + src) : 0u;
        if (i < cnt) {
#pragma unroll
          for (int col = 0; col < IPS_TUPLE_MAX_COLS; ++col)
            if (col < tc.n_cols) *reinterpret_cast<uint32_t*>(t + tc.offset[col]) = x[col];
        }
      } else if (i < cnt) {
        const uint64_t gi = first + i;  // index of this tuple among all selected rows
        for (int col = 0; col < tc.n_cols; ++col) {
          uint64_t src = (uint64_t)batch * kRowsPerTile + i;
          if (tc.flags[col]) {  // OPTIONAL column: NULL bit, or the rank-th dense value
            const uint64_t fw = tc.flags[col][gi >> 6];
            if (!((fw >> (gi & 63)) & 1ull)) {
              t[tc.null_byte[col]] |= (uint8_t)tc.null_mask[col];
              continue;
            }
            src = tc.prefix[col][gi >> 6] + __builtin_popcountll(fw & ((1ull << (gi & 63)) - 1ull));
          }
          uint8_t* slot = t + tc.offset[col];
          const bool aligned = ((tc.offset[col] | ts) & 3) == 0;  // slots are naturally aligned in Impala tuples
          if (tc.width[col] == 4) {
            const uint32_t x = reinterpret_cast<const uint32_t*>(tc.values[col])[src];
            if (aligned) *reinterpret_cast<uint32_t*>(slot) = x;
            else __builtin_memcpy(slot, &x, 4);
          } else {
            const uint64_t x = reinterpret_cast<const uint64_t*>(tc.values[col])[src];
            if (aligned) {
              reinterpret_cast<uint32_t*>(slot)[0] = (uint32_t)x;
              reinterpret_cast<uint32_t*>(slot)[1] = (uint32_t)(x >> 32);
            } else {
              __builtin_memcpy(slot, &x, 8);
            }
          }
        }
      }
      if (IMAGE) {
        wave_lds_fence();
        uint8_t* dst = tuples + (first + i0) * (uint64_t)ts;
        const uint32_t bytes = in_round * (uint32_t)ts;
        if ((ts & 15) == 0) {  // 16-byte pieces: the destination is 16-byte aligned as well
          for (uint32_t o = lane * 16; o < bytes; o += kWave * 16)
            aux_store(reinterpret_cast<u32x4*>(dst + o), *reinterpret_cast<const u32x4*>(image + o));
        } else {
          for (uint32_t o = lane * 4; o < bytes; o += kWave * 4)
            *reinterpret_cast<uint32_t*>(dst + o) = *reinterpret_cast<const uint32_t*>(image + o);
        }
        wave_lds_fence();
      }
    }
  }
}

size_t scan_workspace_bytes(int64_t items);
size_t assemble_workspace_bytes(int64_t n_batches, int n_optional) {
  const int64_t flag_words = (n_batches * kRowsPerTile + 63) / 64;
  return batches_workspace_bytes(n_batches) +
         (size_t)n_optional * (scan_workspace_bytes(flag_words) + (size_t)flag_words * 8) + 64 +
         4096;  // + device copy of a wide template tuple
}

ips_status launch_assemble_tuples(const ips_tuple_column* cols, int n_cols, const uint32_t* counts,
                                  int64_t n_batches, int tuple_size, const void* h_template,
                                  void* tuples, int64_t* total, void* workspace, hipStream_t s) {
  if (n_batches <= 0) {
    IPS_HIP_TRY(hipMemsetAsync(total, 0, 8, s));
    return IPS_OK;
  }
  TupleCols tc;
  memset(&tc, 0, sizeof(tc));
  tc.n_cols = n_cols;
  tc.tuple_size = tuple_size;
  // workspace: [batch scan totals + batch offsets][per OPTIONAL column: flag-word scan totals +
  // per-word prefix]
  const int64_t n_rows_cap = n_batches * kRowsPerTile;
  const int64_t flag_words = (n_rows_cap + 63) / 64;
  uint8_t* ws = reinterpret_cast<uint8_t*>(workspace) + batches_workspace_bytes(n_batches);
  for (int i = 0; i < n_cols; ++i) {
    tc.width[i] = cols[i].value_width;
    tc.offset[i] = cols[i].tuple_offset;
    if (cols[i].d_nonnull_flags) {
      tc.values[i] = cols[i].d_dense_values;
      tc.flags[i] = cols[i].d_nonnull_flags;
      tc.null_byte[i] = cols[i].null_byte_offset;
      tc.null_mask[i] = cols[i].null_bit_mask;
      uint64_t* f_totals = reinterpret_cast<uint64_t*>(ws);
      uint64_t* f_prefix = reinterpret_cast<uint64_t*>(ws + scan_workspace_bytes(flag_words));
      ws += scan_workspace_bytes(flag_words) + (size_t)flag_words * 8;
      const int64_t fb = (flag_words + kScanItems - 1) / kScanItems;
      // the flags bitmap has one bit per selected row; bits beyond the last tuple are zero
      hipLaunchKernelGGL((scan_block_totals_kernel<PopcItems>), dim3((unsigned)fb),
                         dim3(kScanThreads), 0, s, PopcItems{tc.flags[i], n_rows_cap}, flag_words,
                         f_totals);
      hipLaunchKernelGGL(scan_of_totals_kernel, dim3(1), dim3(kScanThreads), 0, s, f_totals, fb,
                         (int64_t*)nullptr);
      hipLaunchKernelGGL(word_prefix_kernel, dim3((unsigned)fb), dim3(kScanThreads), 0, s,
                         tc.flags[i], n_rows_cap, f_totals, f_prefix);
      tc.prefix[i] = f_prefix;
    } else {
      tc.values[i] = cols[i].d_batch_values;
    }
  }
  const bool image = tuple_size <= kTupleImageMax && tuple_size % 4 == 0;
  if (image) {
    if (h_template) memcpy(tc.tmpl, h_template, (size_t)tuple_size);
  } else {  // ws now points behind the last OPTIONAL column's scratch
    if (tuple_size > 4096) {
      set_error("ips_assemble_tuples: tuple_size %d > 4096", tuple_size);
      return IPS_ERR_UNSUPPORTED;
    }
    if (h_template) IPS_HIP_TRY(hipMemcpyAsync(ws, h_template, (size_t)tuple_size, hipMemcpyHostToDevice, s));
    else IPS_HIP_TRY(hipMemsetAsync(ws, 0, (size_t)tuple_size, s));
    tc.d_template = ws;
  }
  uint64_t* batch_off = nullptr;
  ips_status st = launch_batch_offsets(counts, n_batches, total, workspace, &batch_off, s);
  if (st != IPS_OK) return st;
  const int grid = per_batch_grid(n_batches);
  bool fast = image;
  for (int i = 0; i < n_cols; ++i)
    fast = fast && !cols[i].d_nonnull_flags && cols[i].value_width == 4 && (cols[i].tuple_offset & 3) == 0;
  const size_t lds = image ? (size_t)kWavesPerBlock * kWave * tuple_size : 0;
  uint8_t* out = reinterpret_cast<uint8_t*>(tuples);
  if (fast)
    hipLaunchKernelGGL((assemble_tuples_kernel<true, true>), dim3(grid), dim3(kThreads), lds, s, tc,
                       counts, n_batches, batch_off, out);
  else if (image)
    hipLaunchKernelGGL((assemble_tuples_kernel<true, false>), dim3(grid), dim3(kThreads), lds, s,
                       tc, counts, n_batches, batch_off, out);
  else
    hipLaunchKernelGGL((assemble_tuples_kernel<false, false>), dim3(grid), dim3(kThreads), 0, s, tc,
                       counts, n_batches, batch_off, out);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

// =============================================================================================
// Synthetic column generator: x_i = splitmix64(seed + i), value = x_i & mask.
// =============================================================================================
__global__ void synth_kernel(uint64_t seed, int64_t n, uint32_t mask, uint32_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t z = seed + (uint64_t)i + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    out[i] = (uint32_t)z & mask;
  }
}

ips_status launch_synth(uint64_t seed, int64_t n, uint32_t mask, uint32_t* out, hipStream_t s) {
  if (n <= 0) return IPS_OK;
  hipLaunchKernelGGL(synth_kernel, dim3(small_grid(n, 256)), dim3(256), 0, s, seed, n, mask, out);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

}  // namespace ips
