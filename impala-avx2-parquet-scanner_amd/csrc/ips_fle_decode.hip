// ips_fle_decode.hip -- instantiations + launchers of the full-decode and encode kernels for the
// bit widths [IPS_WLO, IPS_WLO+7] (compiled four times so the build parallelises).
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
  int grid = grid_for_tiles(reinterpret_cast<const void*>(kern), tiles);
  if (grid <= 0) return IPS_ERR_HIP;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), 0, s, enc, n_rows, out,
                     reinterpret_cast<const GT*>(dict), dict_entries, bad_index);
  IPS_HIP_TRY(hipGetLastError());
  return IPS_OK;
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
