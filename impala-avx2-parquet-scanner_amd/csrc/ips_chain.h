// ips_chain.h -- arguments of the one-pass conjunct chain (ips_chain.hip), filled by ips_eval_program.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ips.h"

namespace ips {

constexpr int kChainWMaxOps = 6;
constexpr int kChainWMaxSlots = 16;  // 16-byte loads per lane and stripe
enum ChainKind { kChainSingle = 0, kChainPair = 1, kChainIn = 2 };

// one operand: a comparison, two comparisons on one column, or a short IN list
struct ChainOpW {
  int32_t w;             // 1..32
  int32_t kind;          // ChainKind
  int32_t op, op2;       // ips_op of the comparison(s)
  int32_t join;          // pair: 1 AND / 2 OR of the two comparisons
  int32_t combine;       // 0 first operand, 1 AND / 2 OR into the accumulator
  uint32_t c1, c2;
  int32_t n_in;          // IN: members in in_consts
  int32_t img_dw;        // dword offset of the operand's plane image in the wave's LDS region
  uint32_t in_consts[16];
  // plane masks of c1 / c2 (0 or ~0 per plane, w <= 16): they arrive in scalar registers with two wide
  // scalar loads instead of one scalar bit-field extract per plane and constant
  uint32_t m1[16], m2[16];
  uint32_t pad[6];       // 256 bytes per operand
};
static_assert(sizeof(ChainOpW) == 256, "operand descriptors are indexed by a shift");

// one load slot of a stripe: 64 consecutive 16-byte chunks of ONE operand's sub-tile, so that
// everything about a slot is wave-uniform and arrives with one scalar load
struct ChainSlot {
  uint32_t rsrc[4];      // buffer resource over the operand's whole column (its bytes bound every load)
  uint32_t first_byte;   // byte offset of the slot's first chunk inside the sub-tile
  uint32_t tile_bytes;   // 256 w: bytes of a sub-tile of this operand
  uint32_t inv_w;        // floor(2^32 / w) + 1
  int32_t img_dw;        // the operand's plane image
};

struct ChainArgsW {
  ChainSlot slots[kChainWMaxSlots];
  ChainOpW ops[kChainWMaxOps];
  int32_t n_ops, n_slots;
  int32_t image_dwords;  // per wave
  int32_t reserved;
};

// The same chain over column chunks held as page lists whose pages hold the same rows in every operand's
// chunk (blockIdx.y = page; ips_chunk_device.h): per operand the device page table (first page of the launch)
struct ChainPagesArgs {
  const void* slot_pages[kChainWMaxSlots];  // per load slot: its operand's ChunkPage table (unused slots: operand 0's)
  int64_t chunk_rows;
  uint32_t* edges;       // edge slots of the chunk-wide bitmap or NULL (ips_chunk_device.h)
  uint32_t* done;        // a sharded step's page counters or NULL (page_done)
  int32_t done_page0;
  uint32_t done_epoch;
};
struct ChainPagedArgsW {
  ChainArgsW chain;      // (the slots' resources are made per page)
  ChainPagesArgs pg;
};

// The chain over column chunks whose pages end at DIFFERENT rows (ips_chain.hip: segments).
constexpr int kChainSegRunDwords = 62;                      // bitmap dwords per stripe
constexpr int kChainSegRows = kChainSegRunDwords * 32;      // 1984 rows per stripe
constexpr int kChainSegMaxBounds = 4096;                    // page starts of all operands together (LDS of the merge)
struct ChainSegSlot {
  const uint64_t* base;   // the operand's page data, from the block that holds the segment's first row on
  int64_t blocks;         // blocks the page holds from there on
};
struct ChainSegArgs {
  const void* op_pages[kChainWMaxOps];      // per operand
  int32_t op_n_pages[kChainWMaxOps];
  int32_t slot_op[kChainWMaxSlots];         // the slot's operand
  int32_t op_w[kChainWMaxOps + 2];          // the operands' bit widths
  const void* seg_pages;                    // ChunkPage per segment: row0, n_rows, batch0 = first edge slot, flags
  // per segment, filled by chain_segments_kernel so that a wave of the chain needs ONE round of loads to start:
  const ChainSegSlot* seg_slots;            // [segment][8]: first block of the segment in the operand's page, per operand
  const uint32_t* seg_bits;                 // [segment][8]: rows of that block in front of the segment (0..63), per operand
  // the chain's workgroups (4 stripes each) dealt to the segments: no workgroup without a segment
  const uint32_t* wg_seg;                   // [workgroup]: its segment
  const uint32_t* seg_wg0;                  // [segment]: its first workgroup
  const uint32_t* n_wgs;                    // workgroups with work
  int64_t chunk_rows;
  uint32_t* edges;
  int32_t n_bounds;                         // segments (page starts of all operands, ties kept: some are empty)
  int32_t image_dwords;                     // per wave: ONE plane image, the widest operand's
};
struct ChainSegmentedArgsW {
  ChainArgsW chain;
  ChainSegArgs sg;
};
// bytes of workspace the segmented chain needs for n_bounds page starts over chunk_rows rows
size_t chain_segments_workspace_bytes(int n_bounds, int64_t chunk_rows);
// sa.chain.ops[0..n_ops) filled by the caller; op_pages / op_n_pages: every operand's device page table; max_rows: rows
// of the largest page of any operand; workspace of chain_segments_workspace_bytes() bytes.  Three launches + the fix-up.
ips_status launch_chain_w_segments(ChainSegmentedArgsW& sa, const void* const* op_pages, const int* op_n_pages,
                                   int64_t max_rows, int64_t chunk_rows, uint32_t* bitmap32, void* workspace,
                                   hipStream_t s);

// IPS_ERR_UNSUPPORTED: the chain does not fit the kernel (more than kChainWMaxSlots load slots, a
// column of 4 GiB or more) -- the caller falls back to the per-operand plan
ips_status launch_chain_w(ChainArgsW& a, const void* const* enc, int64_t n_rows, uint32_t* bitmap32, hipStream_t s);
// pa.chain.ops[0..n_ops) and pa.pg (but slot_pages) filled by the caller; op_pages[i] = operand i's device page
// table (first page of the launch); n_pages pages of at most max_rows rows
ips_status launch_chain_w_pages(ChainPagedArgsW& pa, const void* const* op_pages, int n_pages, int64_t max_rows,
                                uint32_t* bitmap32, hipStream_t s);

}  // namespace ips
