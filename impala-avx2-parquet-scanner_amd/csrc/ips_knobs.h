// ips_knobs.h -- every development switch of libips_hip.so, in one place.
//
// The shipped library is the build WITHOUT -DIPS_DEV_KNOBS: all switches below are compile-time
// constants at their measured defaults, defining one of them on the command line is a build error,
// and the library reads no environment variable (dev_env() is a constant nullptr), so a stray
// variable in a host process cannot change a plan.  A/B builds for profiling are made with
//   make -C csrc OUT=../libips_X.so BUILD=build_X EXTRA="-DIPS_DEV_KNOBS -DIPS_<SWITCH>=<value>"
// and only those builds look at the IPS_* environment variables listed at the bottom.
#pragma once
#include <stdlib.h>

#ifndef IPS_DEV_KNOBS
#if defined(IPS_MIN_WAVES_PER_EU) || defined(IPS_NO_NT_LOADS) || defined(IPS_NO_NT_BITMAP_STORE) ||     \
    defined(IPS_NO_NT_STREAM_STORE) || defined(IPS_DECODE_NT_LOADS) || defined(IPS_ENCODE_NT_LOADS) ||  \
    defined(IPS_SCAN_SMALL_LDS) || defined(IPS_SCAN_SMALL_LDS_MAX_W) || defined(IPS_ABLATE) ||          \
    defined(IPS_GATHER_WIDE) || defined(IPS_GATHER_MAX_4) || defined(IPS_QUADS) || defined(IPS_QUADS16) || \
    defined(IPS_PHASE_B_GROUP) || defined(IPS_NT_VALUE_STORE) || defined(IPS_DECODE_PACKED) ||          \
    defined(IPS_EXP_ROUNDS) || defined(IPS_AUX_NT) || defined(IPS_PLAIN_ABLATE) || defined(IPS_MIN_SHARE) || defined(IPS_PLAIN_DENSE8) ||  \
    defined(IPS_WINDOW_ABLATE) || defined(IPS_GRID_MULT_PRED) || defined(IPS_GRID_MULT_CHAIN) ||  \
    defined(IPS_GRID_MULT_DECODE) || defined(IPS_GRID_MULT_DICT_DECODE) || defined(IPS_GRID_MULT_SCAN_WIDE)
#error "development switches need -DIPS_DEV_KNOBS (the default library has none)"
#endif
#endif

// ---- occupancy ------------------------------------------------------------------------------
#ifndef IPS_MIN_WAVES_PER_EU
#define IPS_MIN_WAVES_PER_EU 3  // __launch_bounds__ second argument of the wide FLE kernels
#endif

// ---- cache hints (ips_device.h; each one measured, see the comments at its use) --------------
// IPS_NO_NT_LOADS / IPS_NO_NT_BITMAP_STORE / IPS_NO_NT_STREAM_STORE: defined = plain accesses
#ifndef IPS_DECODE_NT_LOADS
#define IPS_DECODE_NT_LOADS true
#endif
#ifndef IPS_ENCODE_NT_LOADS
#define IPS_ENCODE_NT_LOADS true
#endif
#ifndef IPS_NT_VALUE_STORE
#define IPS_NT_VALUE_STORE 0    // nt hint on the fused scan's coalesced value stores (neutral)
#endif
#ifndef IPS_AUX_NT
#define IPS_AUX_NT 1            // read-once slots / the tuple stream of assemble_tuples (232 -> 207 us)
#endif

// ---- fused scan (ips_fle_kernels.h) ----------------------------------------------------------
#ifndef IPS_SCAN_SMALL_LDS
#define IPS_SCAN_SMALL_LDS 1        // 4 / 6 KiB per-wave layouts of the narrow scans
#endif
#ifndef IPS_SCAN_SMALL_LDS_MAX_W
#define IPS_SCAN_SMALL_LDS_MAX_W 8
#endif
#ifndef IPS_ABLATE
#define IPS_ABLATE 0  // timing only, RESULTS ARE WRONG: 1 no phase A/B, 2 nothing after the bitmap store, 3 no phase B, 4 phase B without its stores
#endif
#ifndef IPS_GATHER_WIDE
#define IPS_GATHER_WIDE 0           // rows per lane up to which w > 16 takes the plane-gather path (measured: off)
#endif
#ifndef IPS_GATHER_MAX_4
#define IPS_GATHER_MAX_4 12         // ... for w <= 4 / 8 / 12 / 16
#define IPS_GATHER_MAX_8 5
#define IPS_GATHER_MAX_12 3
#define IPS_GATHER_MAX_16 2
#endif
#ifndef IPS_QUADS
#define IPS_QUADS 1                 // wide columns park half-transposed "quads"
#endif
#ifndef IPS_QUADS16
#define IPS_QUADS16 1               // w=16 / 12 / 10 LT @10 %: 134 -> 117 / 127 -> 111 / 114 -> 100 us
#endif
#ifndef IPS_PHASE_B_GROUP
#define IPS_PHASE_B_GROUP 4         // phase-B rounds whose LDS reads are issued together
#endif
#ifndef IPS_DECODE_PACKED
#define IPS_DECODE_PACKED 1         // dictionary decode of <= 16-bit codes from the lane-packed image
#endif
#ifndef IPS_PLAIN_DENSE8
#define IPS_PLAIN_DENSE8 16u        // 8-byte PLAIN tiles (1024 rows) take the dense compaction from rows / this on
#endif
#ifndef IPS_PLAIN_ABLATE
#define IPS_PLAIN_ABLATE 0  // timing only, RESULTS ARE WRONG: 1 drops the PLAIN scan's value stores, 2 the whole materialisation
#endif

// ---- page lists (ips_chunk_device.h) ---------------------------------------------------------
#ifndef IPS_MIN_SHARE
#define IPS_MIN_SHARE 8             // sub-tiles per contiguous share of a page whose window is shifted (1: Q6 unaligned 476 us, 16: 420 us but the w=32 scan 220 -> 263 us)
#endif
// IPS_WINDOW_ABLATE: defined = shared bitmap dwords are not merged (timing only, RESULTS ARE WRONG)

// ---- grid sizes (ips_capi.hip: grid_mult) ----------------------------------------------------
#ifndef IPS_GRID_MULT_PRED
#define IPS_GRID_MULT_PRED 32       // workgroups per resident slot: stand-alone FLE predicate kernels of w >= 6 (w = 12: 71 -> 67 us)
#endif
#ifndef IPS_GRID_MULT_SCAN_WIDE
#define IPS_GRID_MULT_SCAN_WIDE 16  // ... fused scans of w >= 16 (w = 32 @1 / 10 / 30 %: 192 / 209 / 265 us at 8x, 188 / 205 / 260 at 16x)
#endif
#ifndef IPS_GRID_MULT_DECODE
#define IPS_GRID_MULT_DECODE 16     // ... fle_decode (w = 32 / 16 / 8: 378 / 174 / 85 us at 8x, 362 / 169 / 83 at 16x)
#endif
#ifndef IPS_GRID_MULT_DICT_DECODE
#define IPS_GRID_MULT_DICT_DECODE 32  // ... dictionary decode with a per-workgroup dictionary copy (D = 4096: 277 -> 247 us)
#endif
#ifndef IPS_GRID_MULT_CHAIN
#define IPS_GRID_MULT_CHAIN 64      // ... the one-pass conjunct chain (Q6 shape: 8x 297 us, 16x 289, 32x 284, 64x = one stripe per wave 279)
#endif

// ---- rank tiles (ips_rank_device.h) ----------------------------------------------------------
#ifndef IPS_EXP_ROUNDS
#define IPS_EXP_ROUNDS 2            // 16-byte root loads per lane of an expand / leaf workgroup
#endif

// ---- run-time switches of -DIPS_DEV_KNOBS builds ---------------------------------------------
// IPS_GRID_MULT=<k>            blocks launched per resident block slot (default 8); IPS_GRID_MULT_PRED / _CHAIN: the
//                              predicate-only kernels / the one-pass chain
// IPS_IN_TABLE_MIN=<K>         IN lists of >= K constants take the membership table, every width
// IPS_NO_EARLY_PRUNE=1         w = 32 comparisons without the high-planes-first pruning
// IPS_NO_FUSED_LEAF=1          nullable leaf as predicate + expand launches
// IPS_SELECT_NULLABLE_STEPS=1  ips_dict_select_nullable composed from ten launches
// IPS_NO_SHARED_DICT=1         large dictionaries gathered from L2 instead of the shared LDS copy
// IPS_NO_COUNT_CARRY=1         tile counts of an OPTIONAL column as a launch of their own
// IPS_PAGED_GRID_DIV=<k>       paged launches with 1/k of the usual workgroups (measured: 2 and 4 are slower)
// IPS_SHARD_NO_EXCHANGE=1      ips_fle_scan_allgather launches its signalling scan and no waiter / all-gather
namespace ips {
inline const char* dev_env(const char* name) {
#ifdef IPS_DEV_KNOBS
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}
}  // namespace ips
