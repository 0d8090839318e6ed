// ips_fle_decode.hip -- instantiations + launchers of the full-decode and encode kernels for the
// bit widths [IPS_WLO, IPS_WLO+7] (compiled four times so the build parallelises).
#include <stdlib.h>

#include "ips_fle_kernels.h"
#include "ips_host.h"

#ifndef IPS_WLO
#error "compile with -DIPS_WLO=.. -DIPS_PART=.."
#endif
#define IPS_CAT2(a, b) a##b
#define IPS_CAT(a, b) IPS_CAT2(a, b)

namespace ips {

template <int W, int OW, int G>
static ips_status launch_decode_one(const uint64_t* enc, int64_t n_rows, void* out,
                                    const void* dict, uint32_t dict_entries, int32_t* bad_index,
                                    hipStream_t s) {
  using GT = typename GatherT<G>::type;
  auto kern = fle_decode_kernel<W, OW, G>;
  const int64_t tiles = (n_rows + kRowsPerTile - 1) / kRowsPerTile;
  int grid = grid_for_tiles(reinterpret_cast<const void*>(kern), tiles, G == 0 ? kGridDecode : kGridDictDecode);
  if (grid <= 0) return IPS_ERR_HIP;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), 0, s, enc, n_rows, out,
                     reinterpret_cast<const GT*>(dict), dict_entries, bad_index, 0u);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

// Dictionaries beyond the 32 KiB a workgroup copies for itself: one workgroup of BW waves per CU
// shares a copy in dynamic LDS -- the whole dictionary if it fits next to the waves' images, else
// its first lds_entries entries (the rest is gathered from L2 as before).
constexpr size_t kLdsPerCu = 160 * 1024;
template <int W, int G, int BW, bool TAIL = false>
static ips_status launch_decode_shared(const uint64_t* enc, int64_t n_rows, void* out, const void* dict,
                                       uint32_t dict_entries, int32_t* bad_index, hipStream_t s) {
  using GT = typename GatherT<G>::type;
  auto kern = fle_decode_kernel<W, 4, G, BW, true, TAIL>;
  const size_t room = kLdsPerCu - (size_t)BW * DecodeLds<W, 4, G>::kWaveBytes - 64;  // next to the static images
  const size_t fit = room / G;
  const uint32_t lds_entries = dict_entries < fit ? dict_entries : (uint32_t)fit;
  const size_t dyn = ((size_t)lds_entries * G + 15) & ~(size_t)15;
  static bool attr_set = false;  // (idempotent; a race sets it twice)
  if (!attr_set) {
    IPS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)room));
    attr_set = true;
  }
  const int64_t tiles = (n_rows + kRowsPerTile - 1) / kRowsPerTile;
  const int64_t want = (tiles + BW - 1) / BW;
  const int64_t cus = device_cus();
  const int grid = (int)(want < cus ? want : cus);  // persistent: one workgroup per CU
  if (grid <= 0) return IPS_ERR_HIP;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(BW * kWave), dyn > room ? room : dyn, s, enc, n_rows, out,
                     reinterpret_cast<const GT*>(dict), dict_entries, bad_index, lds_entries);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

// does the whole dictionary fit next to the images of bw waves?
template <int W, int G>
static bool shared_dict_fits(int bw, uint32_t dict_entries) {
  return (size_t)bw * DecodeLds<W, 4, G>::kWaveBytes + (size_t)dict_entries * G + 80 <= kLdsPerCu;
}

template <int W>
static ips_status launch_decode_w(int out_width, int gather, const uint64_t* enc, int64_t n_rows,
                                  void* out, const void* dict, uint32_t dict_entries,
                                  int32_t* bad_index, hipStream_t s) {
  if (gather == 0) {
    if (out_width == 4) return launch_decode_one<W, 4, 0>(enc, n_rows, out, nullptr, 0, nullptr, s);
    if constexpr (W <= 16) {
      if (out_width == 2)
        return launch_decode_one<W, 2, 0>(enc, n_rows, out, nullptr, 0, nullptr, s);
    }
    if constexpr (W <= 8) {
      if (out_width == 1)
        return launch_decode_one<W, 1, 0>(enc, n_rows, out, nullptr, 0, nullptr, s);
    }
    set_error("fle_decode: out_width %d too narrow for bit width %d", out_width, W);
    return IPS_ERR_INVALID_ARG;
  }
  if constexpr (W <= 16) {
    if constexpr (W >= 12) {  // dictionaries that may exceed the private 32 KiB copy
      static const bool shared_off = dev_env("IPS_NO_SHARED_DICT") != nullptr;  // dev switch for A/B runs
      if (!shared_off && (size_t)dict_entries * (size_t)gather > (size_t)kDecodeDictLdsBytes && n_rows >= (1 << 20)) {
        // as many waves as leave room for the whole dictionary; the largest ones (e.g. 40000 int32
        // entries = 160000 bytes) go with EIGHT waves, three quarters of the entries in LDS and the tail
        // gathered from L2 with all of a sub-tile's tail loads in flight together (D = 40000 int32:
        // 606 us with four waves and a tenth in L2, 559 with the batched tail loads, 391 with eight
        // waves, 570 with sixteen; int64: 957 -> 864 us)
        if (gather == 4) {
          if (shared_dict_fits<W, 4>(16, dict_entries)) return launch_decode_shared<W, 4, 16>(enc, n_rows, out, dict, dict_entries, bad_index, s);
          if (shared_dict_fits<W, 4>(8, dict_entries)) return launch_decode_shared<W, 4, 8>(enc, n_rows, out, dict, dict_entries, bad_index, s);
          if (shared_dict_fits<W, 4>(4, dict_entries)) return launch_decode_shared<W, 4, 4>(enc, n_rows, out, dict, dict_entries, bad_index, s);
          return launch_decode_shared<W, 4, 8, true>(enc, n_rows, out, dict, dict_entries, bad_index, s);
        } else if (gather == 8) {
          if (shared_dict_fits<W, 8>(16, dict_entries)) return launch_decode_shared<W, 8, 16>(enc, n_rows, out, dict, dict_entries, bad_index, s);
          if (shared_dict_fits<W, 8>(8, dict_entries)) return launch_decode_shared<W, 8, 8>(enc, n_rows, out, dict, dict_entries, bad_index, s);
          if (shared_dict_fits<W, 8>(4, dict_entries)) return launch_decode_shared<W, 8, 4>(enc, n_rows, out, dict, dict_entries, bad_index, s);
          return launch_decode_shared<W, 8, 8, true>(enc, n_rows, out, dict, dict_entries, bad_index, s);
        }
      }
    }
    if (gather == 4)
      return launch_decode_one<W, 4, 4>(enc, n_rows, out, dict, dict_entries, bad_index, s);
    if (gather == 8)
      return launch_decode_one<W, 4, 8>(enc, n_rows, out, dict, dict_entries, bad_index, s);
  }
  set_error("dictionary decode: unsupported bit width %d / entry size %d", W, gather);
  return IPS_ERR_UNSUPPORTED;
}

template <int W, int IW>
static ips_status launch_encode_one(const void* values, int64_t n_rows, uint64_t* enc,
                                    hipStream_t s) {
  auto kern = fle_encode_kernel<W, IW>;
  const int64_t tiles = (n_rows + kRowsPerTile - 1) / kRowsPerTile;
  int grid = grid_for_tiles(reinterpret_cast<const void*>(kern), tiles);
  if (grid <= 0) return IPS_ERR_HIP;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), 0, s, values, n_rows, enc);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
}

template <int W>
static ips_status launch_encode_w(int in_width, const void* values, int64_t n_rows, uint64_t* enc,
                                  hipStream_t s) {
  if (in_width == 4) return launch_encode_one<W, 4>(values, n_rows, enc, s);
  if constexpr (W <= 16) {
    if (in_width == 2) return launch_encode_one<W, 2>(values, n_rows, enc, s);
  }
  if constexpr (W <= 8) {
    if (in_width == 1) return launch_encode_one<W, 1>(values, n_rows, enc, s);
  }
  set_error("fle_encode: in_width %d too narrow for bit width %d", in_width, W);
  return IPS_ERR_INVALID_ARG;
}

ips_status IPS_CAT(launch_fle_decode_part_, IPS_PART)(int w, int out_width, int gather,
                                                      const uint64_t* enc, int64_t n_rows,
                                                      void* out, const void* dict,
                                                      uint32_t dict_entries, int32_t* bad_index,
                                                      hipStream_t s) {
#define IPS_CASE(N)      \
  case IPS_WLO + N:      \
    return launch_decode_w<IPS_WLO + N>(out_width, gather, enc, n_rows, out, dict, dict_entries, \
                                        bad_index, s);
  switch (w) {
    IPS_CASE(0) IPS_CASE(1) IPS_CASE(2) IPS_CASE(3) IPS_CASE(4) IPS_CASE(5) IPS_CASE(6) IPS_CASE(7)
  }
#undef IPS_CASE
  set_error("bit width %d outside part starting at %d", w, IPS_WLO);
  return IPS_ERR_INVALID_ARG;
}

ips_status IPS_CAT(launch_fle_encode_part_, IPS_PART)(int w, int in_width, const void* values,
                                                      int64_t n_rows, uint64_t* enc,
                                                      hipStream_t s) {
#define IPS_CASE(N) \
  case IPS_WLO + N: \
    return launch_encode_w<IPS_WLO + N>(in_width, values, n_rows, enc, s);
  switch (w) {
    IPS_CASE(0) IPS_CASE(1) IPS_CASE(2) IPS_CASE(3) IPS_CASE(4) IPS_CASE(5) IPS_CASE(6) IPS_CASE(7)
  }
#undef IPS_CASE
  set_error("bit width %d outside part starting at %d", w, IPS_WLO);
  return IPS_ERR_INVALID_ARG;
}

}  // namespace ips
