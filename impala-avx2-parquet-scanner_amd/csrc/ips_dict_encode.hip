// ips_dict_encode.hip -- dictionary encoder of a column chunk on the GPU:
// DictEncoder<T>::Put x n (dict-encoding.h:234-274), WriteDict (sort ascending, :393-406) and
// WriteData (remap to sorted codes + FleEncoder, :408-423).
//
// The dictionary is tiny (<= 40000 entries, dict-encoding.h:157) while the column is not, so:
//   1. one pass over the values inserts their bit patterns into an open-addressing hash table in
//      HBM (one 64-bit CAS per probe; the reference chains nodes through a 64K-bucket table),
//   2. the <= 40000 distinct entries go to the host, are sorted with T's operator< (signed ints,
//      IEEE floats -- what NodeValLess does) and every occupied slot receives its sorted code,
//   3. a second pass looks every value up again and writes its code; ips_fle_encode bit-slices
//      the codes with the width ceil(log2 D).
#include <string.h>

#include <algorithm>
#include <vector>

#include "ips_host.h"

namespace ips {

constexpr uint32_t kTableSlots = 1u << 17;  // 131072: load factor <= 0.31 at the 40000 cap
constexpr uint64_t kEmpty = ~0ull;
constexpr int kMaxEntries = 40000;

__device__ __forceinline__ uint32_t hash64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return (uint32_t)x;
}

// Key of a PLAIN slot = its low KB bytes: the bytes ParquetPlainEncoder::Decode reads back
// (parquet-common.h:180-183, :319-322 for int8/int16, whose upper slot bytes are unspecified).
template <typename S, int KB>
__device__ __forceinline__ uint64_t slot_bits(const S* values, int64_t i) {
  const uint64_t raw = sizeof(S) == 4 ? (uint64_t)(uint32_t)values[i] : (uint64_t)values[i];
  return KB >= 8 ? raw : raw & ((1ull << (KB * 8 % 64)) - 1);
}

// key of an already loaded slot
template <typename S, int KB>
__device__ __forceinline__ uint64_t slot_key(S v) {
  const uint64_t raw = sizeof(S) == 4 ? (uint64_t)(uint32_t)v : (uint64_t)v;
  return KB >= 8 ? raw : raw & ((1ull << (KB * 8 % 64)) - 1);
}

// Rows per lane and iteration: one 16-byte load (4 four-byte or 2 eight-byte slots), the next
// iteration's load issued before the current slots are looked up.  (First version: one slot per
// lane and iteration, every iteration a full HBM round trip: 1.6 ms per pass over 2^28 values.)
template <typename S>
struct SlotVec { static constexpr int N = 16 / (int)sizeof(S); S v[16 / sizeof(S)]; };
template <typename S>
__device__ __forceinline__ SlotVec<S> load_slots(const S* values, int64_t first) {
  SlotVec<S> r;
  const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(values + first));
  __builtin_memcpy(r.v, &t, 16);
  return r;
}

// Both passes are random-access bound in HBM/L2 (one lane per clock through the texture path), so
// every workgroup keeps a small front table in LDS: 4096 slots, 4 probes.  A column with up to a
// few thousand distinct values is then served almost entirely from LDS.
constexpr uint32_t kFrontSlots = 4096;
constexpr int kFrontProbes = 4;

__device__ __forceinline__ uint32_t front_hash(uint64_t bits) {
  uint32_t x = (uint32_t)bits ^ ((uint32_t)(bits >> 32) * 0x9E3779B1u);
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13;
  return x;
}

// counters: [0] distinct entries, [1] the all-ones pattern occurred (it is the table's EMPTY mark)
template <typename S, int KB>
__global__ __launch_bounds__(256) void dict_insert_kernel(const S* __restrict__ values, int64_t n,
                                                          unsigned long long* __restrict__ table,
                                                          unsigned int* __restrict__ counters) {
  __shared__ unsigned long long front[kFrontSlots];  // keys already known to be in the table
  for (uint32_t i = threadIdx.x; i < kFrontSlots; i += blockDim.x) front[i] = kEmpty;
  __syncthreads();
  auto insert_one = [&](uint64_t bits) {
    if (bits == kEmpty) {
      counters[1] = 1u;
      return;
    }
    const uint32_t fh = front_hash(bits);
    bool known = false;
    int free_slot = -1;
#pragma unroll
    for (int pr = 0; pr < kFrontProbes; ++pr) {
      const uint32_t fs = (fh + pr) & (kFrontSlots - 1);
      const unsigned long long k = front[fs];
      if (k == bits) { known = true; break; }
      if (k == kEmpty) { free_slot = (int)fs; break; }
    }
    if (known) return;
    uint32_t h = hash64(bits) & (kTableSlots - 1);
    for (uint32_t probe = 0; probe < kTableSlots; ++probe) {
      unsigned long long seen = table[h];
      if (seen == bits) break;
      if (seen == kEmpty) {
        seen = atomicCAS(table + h, (unsigned long long)kEmpty, (unsigned long long)bits);
        if (seen == kEmpty) {
          atomicAdd(counters, 1u);
          break;
        }
        if (seen == bits) break;
      }
      h = (h + 1) & (kTableSlots - 1);
      if (counters[0] > (unsigned)kMaxEntries) break;  // over the cap: the caller gives up anyway
    }
    // remember it (losing the race for the slot to another key only costs a later global probe)
    if (free_slot >= 0) atomicCAS(&front[free_slot], (unsigned long long)kEmpty, (unsigned long long)bits);
  };
  constexpr int VN = SlotVec<S>::N;
  const int64_t n_vec = n / VN;  // whole 16-byte groups
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  SlotVec<S> cur;
  if (g < n_vec) cur = load_slots<S>(values, g * VN);
  while (g < n_vec) {
    const int64_t gn = g + step;
    SlotVec<S> nxt = cur;
    if (gn < n_vec) nxt = load_slots<S>(values, gn * VN);
#pragma unroll
    for (int e = 0; e < VN; ++e) insert_one(slot_key<S, KB>(cur.v[e]));
    cur = nxt;
    g = gn;
  }
  for (int64_t i = n_vec * VN + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step)
    insert_one(slot_bits<S, KB>(values, i));
}

// Front table of the lookup pass: key and code of a slot travel in ONE 64-bit word when the key
// has at most 4 bytes (code << 32 | key; codes are < 40000 so the all-ones word is free to mean
// empty).  8-byte keys go to the global table directly.
template <typename S, int KB>
__global__ __launch_bounds__(256) void dict_lookup_kernel(
    const S* __restrict__ values, int64_t n, const unsigned long long* __restrict__ table,
    const uint32_t* __restrict__ code_of_slot, uint32_t code_of_ones, uint32_t* __restrict__ codes) {
  constexpr bool kFront = KB <= 4;
  __shared__ unsigned long long front[kFront ? kFrontSlots : 1];
  if (kFront) {
    for (uint32_t i = threadIdx.x; i < kFrontSlots; i += blockDim.x) front[i] = kEmpty;
    __syncthreads();
  }
  auto lookup_one = [&](uint64_t bits) -> uint32_t {
    uint32_t code = code_of_ones;
    if (bits != kEmpty) {
      bool hit = false;
      int free_slot = -1;
      if (kFront) {
        const uint32_t fh = front_hash(bits);
#pragma unroll
        for (int pr = 0; pr < kFrontProbes; ++pr) {
          const uint32_t fs = (fh + pr) & (kFrontSlots - 1);
          const unsigned long long e = front[fs];
          if (e == kEmpty) { free_slot = (int)fs; break; }
          if ((uint32_t)e == (uint32_t)bits) { code = (uint32_t)(e >> 32); hit = true; break; }
        }
      }
      if (!hit) {
        uint32_t h = hash64(bits) & (kTableSlots - 1);
        while (table[h] != bits) h = (h + 1) & (kTableSlots - 1);  // present by construction
        code = code_of_slot[h];
        if (kFront && free_slot >= 0)
          atomicCAS(&front[free_slot], (unsigned long long)kEmpty,
                    ((unsigned long long)code << 32) | (unsigned long long)(uint32_t)bits);
      }
    }
    return code;
  };
  constexpr int VN = SlotVec<S>::N;
  const int64_t n_vec = n / VN;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  SlotVec<S> cur;
  if (g < n_vec) cur = load_slots<S>(values, g * VN);
  while (g < n_vec) {
    const int64_t gn = g + step;
    SlotVec<S> nxt = cur;
    if (gn < n_vec) nxt = load_slots<S>(values, gn * VN);
    uint32_t c[VN];
#pragma unroll
    for (int e = 0; e < VN; ++e) c[e] = lookup_one(slot_key<S, KB>(cur.v[e]));
    if constexpr (VN == 4) {
      const u32x4 o = {c[0], c[1], c[2], c[3]};
      *reinterpret_cast<u32x4*>(codes + g * VN) = o;       // 16-byte aligned: the workspace offset is
    } else {
      const u32x2 o = {c[0], c[1]};
      *reinterpret_cast<u32x2*>(codes + g * VN) = o;
    }
    cur = nxt;
    g = gn;
  }
  for (int64_t i = n_vec * VN + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step)
    codes[i] = lookup_one(slot_bits<S, KB>(values, i));
}

template <typename T>
static T from_bits(uint64_t bits) {
  T t;
  if (sizeof(T) <= 4) { uint32_t lo = (uint32_t)bits; memcpy(&t, &lo, sizeof(T)); }
  else memcpy(&t, &bits, sizeof(T));
  return t;
}

template <typename T>
static bool is_nan(T) { return false; }
template <>
bool is_nan<float>(float v) { return v != v; }
template <>
bool is_nan<double>(double v) { return v != v; }

// Workspace layout (all temporaries come from the caller: the entry point allocates and frees
// nothing): hash table | code of every slot | counters | one 4-byte code per row.
static size_t ws_table_off() { return 0; }
static size_t ws_code_of_slot_off() { return (size_t)kTableSlots * 8; }
static size_t ws_counters_off() { return ws_code_of_slot_off() + (size_t)kTableSlots * 4; }
static size_t ws_codes_off() { return ws_counters_off() + 256; }
size_t dict_encode_workspace_bytes(int64_t n_rows) {
  return ws_codes_off() + (size_t)(n_rows > 0 ? n_rows : 1) * 4 + 16;
}

template <typename T, typename S>
static ips_status dict_encode_t(const void* d_values, int64_t n, ips_type type, void* h_dict_page,
                                int64_t page_capacity, int64_t* dict_len, int* bit_width,
                                void* d_codes_enc, void* d_workspace, hipStream_t s) {
  uint8_t* ws = reinterpret_cast<uint8_t*>(d_workspace);
  unsigned long long* d_table = reinterpret_cast<unsigned long long*>(ws + ws_table_off());
  unsigned int* d_counters = reinterpret_cast<unsigned int*>(ws + ws_counters_off());
  uint32_t* d_code_of_slot = reinterpret_cast<uint32_t*>(ws + ws_code_of_slot_off());
  uint32_t* d_codes = reinterpret_cast<uint32_t*>(ws + ws_codes_off());
#define IPS_TRY_CLEAN(expr)                                     \
  do {                                                          \
    hipError_t _e = (expr);                                     \
    if (_e != hipSuccess) return hip_fail(_e, #expr);              \
  } while (0)
  IPS_TRY_CLEAN(hipMemsetAsync(d_table, 0xFF, (size_t)kTableSlots * 8, s));
  IPS_TRY_CLEAN(hipMemsetAsync(d_counters, 0, 8, s));
  const int grid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)device_cus() * 8);
  if (n > 0)
    hipLaunchKernelGGL((dict_insert_kernel<S, (int)sizeof(T)>), dim3(grid > 0 ? grid : 1), dim3(256), 0, s,
                       reinterpret_cast<const S*>(d_values), n, d_table, d_counters);
  unsigned int counters[2] = {0, 0};
  IPS_TRY_CLEAN(hipMemcpyAsync(counters, d_counters, 8, hipMemcpyDeviceToHost, s));
  IPS_TRY_CLEAN(hipStreamSynchronize(s));
  const int64_t entries = (int64_t)counters[0] + (counters[1] ? 1 : 0);
  if (entries > kMaxEntries) {
    set_error("ips_dict_encode: more than %d distinct values (dict-encoding.h:157): use PLAIN", kMaxEntries);
    return IPS_ERR_UNSUPPORTED;
  }
  const int slot = ips_plain_stride(type);
  if (entries * slot > page_capacity) {
    set_error("ips_dict_encode: dictionary page needs %lld bytes", (long long)(entries * slot));
    return IPS_ERR_INVALID_ARG;
  }
  std::vector<unsigned long long> table(kTableSlots);
  IPS_TRY_CLEAN(hipMemcpy(table.data(), d_table, (size_t)kTableSlots * 8, hipMemcpyDeviceToHost));
  struct Entry { T value; uint32_t slot; };
  std::vector<Entry> ents;
  ents.reserve((size_t)entries);
  for (uint32_t h = 0; h < kTableSlots; ++h)
    if (table[h] != kEmpty) ents.push_back(Entry{from_bits<T>(table[h]), h});
  if (counters[1]) ents.push_back(Entry{from_bits<T>(kEmpty), kTableSlots});
  // operator< is not a strict weak order once a NaN is present (std::sort / lower_bound would be
  // undefined behaviour): such a column is not dictionary-encoded here
  for (const Entry& e : ents) {
    if (is_nan<T>(e.value)) {
      set_error("ips_dict_encode: NaN in a FLOAT/DOUBLE column: dictionary order undefined "
                "(dict-encoding.h:370-372), use PLAIN");
      return IPS_ERR_UNSUPPORTED;
    }
  }
  std::sort(ents.begin(), ents.end(), [](const Entry& a, const Entry& b) { return a.value < b.value; });
  std::vector<uint32_t> code_of_slot(kTableSlots, 0);
  uint32_t code_of_ones = 0;
  uint8_t* page = reinterpret_cast<uint8_t*>(h_dict_page);
  for (size_t i = 0; i < ents.size(); ++i) {
    if (ents[i].slot == kTableSlots) code_of_ones = (uint32_t)i; else code_of_slot[ents[i].slot] = (uint32_t)i;
    // ParquetPlainEncoder::Encode: sizeof(T) bytes, int8/int16 widened to the 4-byte slot
    memset(page + i * slot, 0, (size_t)slot);
    if (sizeof(T) < 4) { int32_t wide = (int32_t)ents[i].value; memcpy(page + i * slot, &wide, 4); }
    else memcpy(page + i * slot, &ents[i].value, sizeof(T));
  }
  *dict_len = (int64_t)ents.size() * slot;
  const int bw = ips_dict_bit_width((int64_t)ents.size());
  *bit_width = bw;
  ips_status st = IPS_OK;
  if (n > 0 && bw > 0) {
    IPS_TRY_CLEAN(hipMemcpyAsync(d_code_of_slot, code_of_slot.data(), (size_t)kTableSlots * 4,
                                 hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL((dict_lookup_kernel<S, (int)sizeof(T)>), dim3(grid > 0 ? grid : 1), dim3(256), 0, s,
                       reinterpret_cast<const S*>(d_values), n, d_table, d_code_of_slot, code_of_ones,
                       d_codes);
    st = launch_fle_encode(bw, 4, d_codes, n, reinterpret_cast<uint64_t*>(d_codes_enc), s);
    IPS_TRY_CLEAN(hipStreamSynchronize(s));
  }
  return st;
#undef IPS_TRY_CLEAN
}

}  // namespace ips

using namespace ips;

extern "C" size_t ips_dict_encode_workspace_bytes(int64_t n_rows) {
  return dict_encode_workspace_bytes(n_rows);
}

extern "C" ips_status ips_dict_encode(const void* d_values, int64_t n_rows, ips_type type,
                                      void* h_dict_page, int64_t dict_page_capacity,
                                      int64_t* dict_len, int* bit_width, void* d_codes_enc,
                                      void* d_workspace, ips_stream stream) {
  IPS_REQUIRE(n_rows >= 0, "ips_dict_encode: n_rows < 0");
  IPS_REQUIRE(type >= IPS_T_INT8 && type <= IPS_T_DOUBLE, "ips_dict_encode: bad type %d", (int)type);
  IPS_REQUIRE(h_dict_page && dict_len && bit_width, "ips_dict_encode: NULL out pointer");
  IPS_REQUIRE(n_rows == 0 || (d_values && aligned16(d_values) && d_codes_enc && aligned16(d_codes_enc)),
              "ips_dict_encode: NULL or misaligned device pointer");
  IPS_REQUIRE(d_workspace && aligned16(d_workspace),
              "ips_dict_encode: pass a workspace of ips_dict_encode_workspace_bytes() bytes");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (type) {
    case IPS_T_INT8: return dict_encode_t<int8_t, int32_t>(d_values, n_rows, type, h_dict_page, dict_page_capacity, dict_len, bit_width, d_codes_enc, d_workspace, s);
    case IPS_T_INT16: return dict_encode_t<int16_t, int32_t>(d_values, n_rows, type, h_dict_page, dict_page_capacity, dict_len, bit_width, d_codes_enc, d_workspace, s);
    case IPS_T_INT32: return dict_encode_t<int32_t, int32_t>(d_values, n_rows, type, h_dict_page, dict_page_capacity, dict_len, bit_width, d_codes_enc, d_workspace, s);
    case IPS_T_INT64: return dict_encode_t<int64_t, int64_t>(d_values, n_rows, type, h_dict_page, dict_page_capacity, dict_len, bit_width, d_codes_enc, d_workspace, s);
    case IPS_T_FLOAT: return dict_encode_t<float, int32_t>(d_values, n_rows, type, h_dict_page, dict_page_capacity, dict_len, bit_width, d_codes_enc, d_workspace, s);
    default: return dict_encode_t<double, int64_t>(d_values, n_rows, type, h_dict_page, dict_page_capacity, dict_len, bit_width, d_codes_enc, d_workspace, s);
  }
}
