// ips_chunk_device.h -- a column chunk as a LIST OF PAGES on the device, and how kernels that
// work page by page write into (and read from) bitmaps over the chunk's row space.
//
// The scanner holds a column chunk as data pages whose row counts are whatever the writer chose
// (ReadDataPage / InitDataPage per page, hdfs-parquet-scanner.cc:730-924); pages of different
// columns end at different rows, and EvalSimplePredicates cuts its batches at every column's page
// end (hdfs-parquet-scanner.cc:1837-1855).  Here every page is evaluated in ITS OWN block geometry
// (FLE blocks of 64 rows start at the page's first row; blockIdx.y = page, one launch per run of
// pages of one bit width) and the result is written at the page's row offset of the chunk-wide
// bitmap.  A lane's 32 rows then straddle two bitmap dwords whenever row0 % 32 != 0:
//   * inside the wave the neighbour's bits arrive by one DPP wave shift, and every dword all of
//     whose 32 bits come from this wave is an ordinary store / read-modify-write;
//   * a dword shared with the previous or next sub-tile, or with the neighbouring PAGE, is merged
//     with atomics on exactly the bits this wave owns -- clear-then-set (store), and-with-mask
//     (AND into) or or (OR into) -- so the result does not depend on the order in which the
//     sharers arrive and needs no zero-initialised bitmap;
//   * the last page also owns the bits behind the chunk's last row up to the end of the bitmap's
//     last 64-bit word (they are zero, as in every bitmap of this library).
// Pages that start on a multiple of 32 rows and hold a multiple of 32 rows take the same stores as
// the single-buffer kernels.
#pragma once
#include "ips_device.h"

namespace ips {

// One page as the kernels see it: 64 bytes, wave-uniform, read with scalar loads by blockIdx.y.
struct ChunkPage {
  const uint64_t* data;    // FLE blocks (codes / values; OPTIONAL: of the non-NULL rows) or PLAIN slots
  const uint64_t* levels;  // OPTIONAL: FLE blocks (width 1) of the page's definition levels, else NULL
  int64_t n_rows;          // rows of the page, NULLs included
  int64_t n_data;          // rows 'data' holds (REQUIRED: n_rows)
  int64_t row0;            // first row of the page within the chunk
  uint32_t batch0;         // 2048-row sub-tiles of all earlier pages: the page's first batch slot
  uint32_t flags;          // bit 0: last page of the chunk
  uint32_t rank0;          // OPTIONAL: first entry of the page's tile counts in the rank workspace
  uint32_t reserved;
};
static_assert(sizeof(ChunkPage) == 56 || sizeof(ChunkPage) == 64, "page descriptor layout");
constexpr uint32_t kPageLast = 1u;

// Where a page's lane dwords go inside the chunk-wide bitmap (all wave-uniform).
struct BitmapWindow {
  uint32_t* base;        // dword that holds the page's first row
  uint32_t shift;        // row0 % 32
  uint32_t own_tail;     // last page: owns the bits behind the chunk's last row (see above)
  int64_t n_rows;        // rows of the page
  int64_t tail_dword;    // last page: page-relative index of the dword that holds the chunk's last row
  uint32_t tail_mask;    //            its bits behind that row
  uint32_t tail_extra;   //            1: the bitmap's last word has one more dword behind it
  uint32_t* edges;       // the page's edge slots (4 dwords per sub-tile: [lo, hi, -, -]) or NULL, see window_emit
  uint32_t coherent;     // 1: the dwords are stored device-coherently (written through to memory: a sharded
                         //    step hands pages to the exchange while the launch is still running, page_done)
};

__device__ __forceinline__ void window_store(uint32_t* p, uint32_t v, uint32_t coherent) {
  if (coherent) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else IPS_BITMAP_STORE(p, v);
}

__device__ __forceinline__ BitmapWindow bitmap_window(uint32_t* bitmap32, const ChunkPage& pg, int64_t chunk_rows,
                                                      const void* done = nullptr, uint32_t* edges = nullptr) {
  BitmapWindow w;
  w.edges = edges ? edges + 4 * (size_t)pg.batch0 : nullptr;
  // (device-coherent dword stores were tried for pages that are handed on while the launch is still
  // running: 439 us instead of 224 for the w = 32 scan -- the bitmap stays nt stores, see page_done)
  w.coherent = 0u;
  (void)done;
  w.base = bitmap32 + (pg.row0 >> 5);
  w.shift = (uint32_t)(pg.row0 & 31);
  w.own_tail = pg.flags & kPageLast;
  w.n_rows = pg.n_rows;
  const int64_t last = chunk_rows - 1;  // (a chunk with pages has rows)
  w.tail_dword = (last >> 5) - (pg.row0 >> 5);
  const uint32_t b = (uint32_t)(last & 31);
  w.tail_mask = b == 31u ? 0u : ~((2u << b) - 1u);
  w.tail_extra = ((last >> 5) & 1) == 0 ? 1u : 0u;  // an even dword index is followed by the word's high half
  return w;
}

// one dword of the chunk-wide bitmap: 'mask' = the bits this lane owns, 'val' its values there
// (val & ~mask == 0).  combine: 0 store, 1 AND into, 2 OR into.
// have_old / old: the dword's present value when the caller has fetched it ahead (window_operand).
// (by value: a pointer to the caller's register made the compiler keep it in scratch)
__device__ __forceinline__ void window_put(uint32_t* p, uint32_t val, uint32_t mask, int combine, uint32_t coherent,
                                           bool have_old = false, uint32_t old = 0u) {
  if (mask == 0u) return;
  if (mask == ~0u) {
    if (combine == 1) val &= have_old ? old : *p;
    else if (combine == 2) val |= have_old ? old : *p;
    window_store(p, val, coherent);
    return;
  }
#ifdef IPS_WINDOW_ABLATE  // dev (timing only, results are wrong): no merging of shared dwords
  return;
#endif
  if (combine == 0) {
    atomicAnd(p, ~mask);
    if (val) atomicOr(p, val);
  } else if (combine == 1) {
    atomicAnd(p, val | ~mask);
  } else if (val) {
    atomicOr(p, val);
  }
}

// A wave works through CONSECUTIVE runs of page-relative bitmap dwords (sub-tile after sub-tile of
// its share of the page).  With a shifted window the high part of a run's last lane belongs to the
// first dword of the next run: it is carried there in two scalar registers instead of being merged
// through memory, so that only the two ends of a wave's whole share touch dwords that another wave
// also writes (atomics: two per share, not two per sub-tile -- which cost the unaligned Q6 plan
// 2.6x: 900 us against 344 us, profiles/round3_chunks.md).
struct WindowCarry {
  uint32_t bm = 0u, vm = 0u;  // the previous run's last lane: its dword and the rows it owns (unshifted)
  int64_t next = -1;          // page-relative dword that follows the previous run; -1: nothing carried
};

// what is still carried goes out (the end of a wave's share, or a gap in its runs)
__device__ __forceinline__ void window_flush(const BitmapWindow& w, WindowCarry& cy, int combine) {
  if (cy.next >= 0 && w.shift != 0u && cy.vm != 0u && (threadIdx.x & (kWave - 1)) == 0) {
    const uint32_t r = 32u - w.shift;
    uint32_t m2 = cy.vm >> r;
    const bool tail = w.own_tail && cy.next == w.tail_dword && m2 != 0u;
    if (tail) m2 |= w.tail_mask;
    window_put(w.base + cy.next, cy.bm >> r, m2, combine, w.coherent);
    if (tail && w.tail_extra && combine != 2) window_store(w.base + cy.next + 1, 0u, w.coherent);
  }
  cy.next = -1;
  cy.bm = cy.vm = 0u;
}

// One run: lane l holds the page-relative dword d = d0 + l ('bm': bit j <-> row 32 d + j of the page)
// for l <= last_lane (63, or 31 when only half the wave holds dwords); every lane of the wave calls
// it.  combine: 0 store, 1 AND into, 2 OR into.
// The dword an AND-into / OR-into launch will combine lane dword d with, fetched ahead (with the
// sub-tile's register prefetch) when the window is aligned; 'have' says whether it was.
__device__ __forceinline__ bool window_has_operand(const BitmapWindow& w, int combine) {
  return combine != 0 && w.shift == 0u;  // wave-uniform
}
__device__ __forceinline__ uint32_t window_operand(const BitmapWindow& w, int64_t d) {
  return d * 32 < w.n_rows ? w.base[d] : 0u;
}

// run >= 0: the run's index among the page's runs of (last_lane + 1) dwords = its edge slot (runs of 64: derived from d).
__device__ __forceinline__ void window_emit(const BitmapWindow& w, WindowCarry& cy, int64_t d, uint32_t bm, int combine,
                                            int last_lane = kWave - 1, bool have_old = false, uint32_t old = 0u,
                                            int64_t run = -1) {
  const int lane = (int)(threadIdx.x & (kWave - 1));
  const int64_t valid = w.n_rows - d * 32;
  const uint32_t vm = lane > last_lane ? 0u : valid >= 32 ? ~0u : valid <= 0 ? 0u : ((1u << valid) - 1u);
  bm &= vm;
  uint32_t val = bm, mask = vm;
  if (w.shift != 0u && w.edges != nullptr && (last_lane == kWave - 1 || run >= 0)) {  // wave-uniform
    // Edge mode (sub-tile runs of 64 dwords): the run's two end dwords -- the low part of lane 0, the
    // high part of lane 63 -- go to the sub-tile's edge slot with plain stores and window_fixup_kernel
    // merges neighbouring sub-tiles' parts afterwards: no atomics in this kernel, any sub-tile order.
    const uint32_t s = w.shift, r = 32u - s;
    const uint32_t up_bm = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bm, 0x138, 0xF, 0xF, true);  // wave_shr:1
    const uint32_t up_vm = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vm, 0x138, 0xF, 0xF, true);
    uint32_t* slot = w.edges + 4 * (run >= 0 ? run : ((d - lane) >> 6));
    if (lane == 0) {
      slot[0] = bm << s;
      return;
    }
    if (lane == last_lane) slot[1] = bm >> r;
    if (lane > last_lane) return;  // (the dword behind the run belongs to the next run's lane 0)
    val = (bm << s) | (up_bm >> r);
    mask = (vm << s) | (up_vm >> r);
  } else if (w.shift != 0u) {  // wave-uniform
    const int64_t d0 = d - lane;
    if (cy.next != d0) window_flush(w, cy, combine);  // (not the run that follows the carried one)
    const uint32_t s = w.shift, r = 32u - s;
    uint32_t up_bm = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bm, 0x138, 0xF, 0xF, true);  // wave_shr:1
    uint32_t up_vm = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vm, 0x138, 0xF, 0xF, true);
    if (lane == 0) { up_bm = cy.bm; up_vm = cy.vm; }
    val = (bm << s) | (up_bm >> r);
    mask = (vm << s) | (up_vm >> r);
    if (lane > last_lane) mask = 0u;  // (the last lane's high part travels in the carry)
    cy.bm = (uint32_t)__builtin_amdgcn_readlane((int)bm, last_lane);
    cy.vm = (uint32_t)__builtin_amdgcn_readlane((int)vm, last_lane);
    cy.next = d0 + last_lane + 1;
  }
  const bool tail = w.own_tail && d == w.tail_dword && mask != 0u;  // zeros behind the chunk's last row
  if (tail) mask |= w.tail_mask;
  window_put(w.base + d, val, mask, combine, w.coherent, have_old && w.shift == 0u, old);
  if (tail && w.tail_extra && combine != 2) window_store(w.base + d + 1, 0u, w.coherent);
}

// The same for a wave whose lanes hold FOUR consecutive dwords each (two 64-bit words per lane, the
// nullable leaf): in[k] = page-relative dword d0 + k, lane l + 1 continues where lane l ends.
__device__ __forceinline__ void window_emit_quad(const BitmapWindow& w, WindowCarry& cy, int64_t d0,
                                                 const uint32_t (&in)[4], int combine) {
  const int lane = (int)(threadIdx.x & (kWave - 1));
  uint32_t bm[4], vm[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t valid = w.n_rows - (d0 + k) * 32;
    vm[k] = valid >= 32 ? ~0u : valid <= 0 ? 0u : ((1u << valid) - 1u);
    bm[k] = in[k] & vm[k];
  }
  const uint32_t s = w.shift, r = 32u - s;
  uint32_t pb = 0u, pv = 0u;
  if (s != 0u) {  // wave-uniform
    const int64_t run0 = d0 - 4 * lane;
    if (cy.next != run0) window_flush(w, cy, combine);
    pb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bm[3], 0x138, 0xF, 0xF, true);  // wave_shr:1
    pv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vm[3], 0x138, 0xF, 0xF, true);
    if (lane == 0) { pb = cy.bm; pv = cy.vm; }
    pb >>= r;
    pv >>= r;
    cy.bm = (uint32_t)__builtin_amdgcn_readlane((int)bm[3], kWave - 1);
    cy.vm = (uint32_t)__builtin_amdgcn_readlane((int)vm[3], kWave - 1);
    cy.next = run0 + 4 * kWave;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t val = (bm[k] << s) | pb;
    uint32_t mask = (vm[k] << s) | pv;
    const bool tail = w.own_tail && d0 + k == w.tail_dword && mask != 0u;
    if (tail) mask |= w.tail_mask;
    window_put(w.base + d0 + k, val, mask, combine, w.coherent);
    if (tail && w.tail_extra && combine != 2) window_store(w.base + d0 + k + 1, 0u, w.coherent);
    pb = s != 0u ? bm[k] >> r : 0u;
    pv = s != 0u ? vm[k] >> r : 0u;
  }
}

// The sub-tiles a wave takes of a page with 'tiles' sub-tiles: every wave-th one when the window is
// dword aligned (neighbouring waves stream neighbouring bytes), else ONE contiguous share, so that
// the carry above covers all but the share's two ends.
struct TileShare { int64_t first, step, end; };
__device__ __forceinline__ TileShare tile_share(const BitmapWindow& w, int64_t tiles, int64_t wave_index, int64_t n_waves,
                                                int64_t min_share = IPS_MIN_SHARE) {
  if (w.shift == 0u || w.edges != nullptr) return TileShare{wave_index, n_waves, tiles};
  // at least kMinShare sub-tiles per share (the waves behind the last share find nothing to do):
  // the two shared dwords of a share cost four atomics, 2^20-row pages of a narrow column would
  // otherwise be cut into shares of five sub-tiles (Q6 over unaligned pages: 476 -> 4xx us)
  int64_t q = (tiles + n_waves - 1) / n_waves;
  q = q < min_share ? min_share : q;
  const int64_t first = wave_index * q;
  return TileShare{first, 1, first + q < tiles ? first + q : tiles};
}

// The other direction: the wave reads a run of consecutive page-relative dwords of a selection over
// the chunk's rows, lane l the dword d = d0 + l (bit j <-> row 32 d + j of the page; rows beyond the
// page cleared).  Every lane calls it; lanes with active = false get 0 (they still fetch: the lane
// before them takes its high part from them).  total_dwords: dwords of the bitmap.
__device__ __forceinline__ uint32_t window_fetch_at(const uint32_t* __restrict__ bitmap32, int64_t total_dwords,
                                                    const ChunkPage& pg, int64_t d, bool active) {
  const int lane = (int)(threadIdx.x & (kWave - 1));
  const int64_t g = (pg.row0 >> 5) + d;
  const uint32_t s = (uint32_t)(pg.row0 & 31);
  const int64_t valid = pg.n_rows - d * 32;
  if (__builtin_amdgcn_ballot_w64(valid > 0) == 0ull) return 0u;  // wave-uniform: nothing of the page here
  // (a dword right behind the page's last row of this lane may still hold rows of the lane before it)
  const uint32_t lo = (valid > -32 && g < total_dwords) ? bitmap32[g] : 0u;
  uint32_t x = lo;
  if (s != 0u) {  // wave-uniform
    uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x130, 0xF, 0xF, true);  // wave_shl:1: lane i <- lane i + 1
    if (lane == kWave - 1) hi = (valid > 0 && g + 1 < total_dwords) ? bitmap32[g + 1] : 0u;  // (the last lane has no neighbour)
    x = (lo >> s) | (hi << (32u - s));
  }
  if (!active) return 0u;
  return valid >= 32 ? x : valid <= 0 ? 0u : (x & ((1u << valid) - 1u));
}
__device__ __forceinline__ uint32_t window_fetch(const uint32_t* __restrict__ bitmap32, int64_t total_dwords,
                                                 const ChunkPage& pg, int64_t tile, int lane) {
  return window_fetch_at(bitmap32, total_dwords, pg, tile * 64 + lane, true);
}

// End of a wave's work on page blockIdx.y of a launch whose pages are being waited for one by one
// (the exchange of a sharded step starts on a page's bitmap words while later pages are still being
// scanned): the wave's stores become visible device-wide, then it counts itself; the wave that makes
// the count gridDim.x * waves-per-workgroup raises the page's flag, which is what the waiter polls.
// Layout of the completion words (uint32, device memory owned by the waiter's side):
//   sub[page][k]   k < kDoneSubs, each in its own 128-byte line: waves (index % kDoneSubs == k) finished
//   top[page]      sub-counters that are complete
//   flag[page]     the page is complete (what the waiter polls)
// Every wave of a 2^28-row launch adding to ONE word per page serialises 8192 device-scope atomics on
// one address: ~50 ns each, 0.42 ms for a 0.21 ms scan (profiles/round3_sharded_step.md).  Spread over
// kDoneSubs lines the chains are 64 atomics long and run side by side.
constexpr int kDonePages = 64;
constexpr int kDoneSubs = 64;
constexpr int kDoneLine = 32;  // uint32 per 128-byte line
constexpr int kDoneTop = kDonePages * kDoneSubs * kDoneLine;
constexpr int kDoneFlag = kDoneTop + kDonePages * kDoneLine;
constexpr int kDoneWords = kDoneFlag + kDonePages * kDoneLine;  // (+ 1: the waiters' timeout word)

// The counters clean up after themselves: the wave that completes a page -- every wave of the page
// has counted by then, and the next step's launch starts after this one -- zeroes them before it
// raises the flag, and the flag takes the step's EPOCH (a new value every step), so nothing has to
// be reset between steps (a 528-KB memset + an event per step: 24 us of a 0.21-ms scan).
__device__ __forceinline__ void page_done(uint32_t* done, int page0, uint32_t epoch) {
  if (done == nullptr) return;
  // The wave waits until its stores have been taken by its XCD's L2 and counts itself; nothing is
  // flushed here.  A device-scope release per wave writes back (and with __threadfence also
  // invalidates) the whole L2 under the feet of every wave still streaming: 1.66 ms / 3.2 ms for a
  // 0.21 ms scan; device-coherent bitmap stores: 0.44 ms.  The dirty bitmap lines reach memory when
  // the waiter kernel on the consumer's stream ENDS: a kernel's end is a release over all L2s, the
  // same flush a piece-by-piece launch sequence gets from its kernel boundaries.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);
  if ((threadIdx.x & (kWave - 1)) == 0) {
    const uint32_t waves_per_block = blockDim.x / kWave;
    const uint32_t all = gridDim.x * waves_per_block;
    const uint32_t idx = blockIdx.x * waves_per_block + (threadIdx.x / kWave);
    const uint32_t subs = all < (uint32_t)kDoneSubs ? all : (uint32_t)kDoneSubs;
    const uint32_t k = idx % subs;
    const uint32_t mine = (all - k + subs - 1u) / subs;  // waves that count on sub-counter k
    const int page = page0 + (int)blockIdx.y;
    uint32_t* sub = done + ((size_t)page * kDoneSubs + k) * kDoneLine;
    if (__hip_atomic_fetch_add(sub, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == mine) {
      uint32_t* top = done + kDoneTop + (size_t)page * kDoneLine;
      if (__hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == subs) {
        for (uint32_t j = 0; j < subs; ++j)
          __hip_atomic_store(done + ((size_t)page * kDoneSubs + j) * kDoneLine, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);
        __hip_atomic_store(done + kDoneFlag + (size_t)page * kDoneLine, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// the page a workgroup works on (blockIdx.y of a paged launch)
__device__ __forceinline__ ChunkPage load_page(const ChunkPage* __restrict__ pages) {
  return pages[blockIdx.y];
}

}  // namespace ips
