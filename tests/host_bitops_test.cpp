// CPU unit test of the per-lane bit arithmetic the HIP kernels run (csrc/ips_bitops.h is
// __host__ __device__): the 32x32 bit transposes, planes<->values for every width, and the
// predicate recurrence, emulating one wave (64 lanes = 32 blocks x 2 halves) against the scalar
// layout definition fle-encoding.h:8338-8340.  Test infrastructure only.
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../impala-avx2-parquet-scanner_amd/csrc/ips_bitops.h"

static int fails = 0;
#define CHECK(c) do { if (!(c)) { if (++fails < 10) fprintf(stderr, "fail %s:%d %s\n", __FILE__, __LINE__, #c); } } while (0)

static uint64_t st = 88172645463325252ull;
static uint64_t rnd() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; }

template <int W>
static void test_width() {
  const uint64_t mod = 1ull << W;
  // one block of 64 values -> W words by the layout definition
  uint32_t vals[64];
  uint64_t words[W];
  for (int i = 0; i < W; ++i) words[i] = 0;
  for (int k = 0; k < 64; ++k) {
    vals[k] = (uint32_t)(rnd() % mod);
    for (int i = 0; i < W; ++i) words[i] |= (uint64_t)((vals[k] >> i) & 1u) << (63 - k);
  }
  for (int q = 0; q < 2; ++q) {  // lane halves: q=0 rows 0..31 (high dword), q=1 rows 32..63
    uint32_t p[W], v[32], p2[W];
    for (int i = 0; i < W; ++i) p[i] = (uint32_t)(words[i] >> (q == 0 ? 32 : 0));
    ips::planes_to_values<W>(p, v);
    for (int j = 0; j < 32; ++j) CHECK(v[j] == vals[32 * q + j]);
    ips::values_to_planes<W>(v, p2);
    for (int i = 0; i < W; ++i) CHECK(p2[i] == p[i]);
    // predicate recurrence on the half-block
    uint32_t consts[4] = {0u, (uint32_t)(mod - 1), vals[5], (uint32_t)(mod / 10)};
    for (uint32_t c : consts) {
      ips::CmpState s{0u, ~0u};
      for (int k = W - 1; k >= 0; --k) ips::cmp_step(s, p[k], ((c >> k) & 1u) ? ~0u : 0u);
      for (int op = 0; op < 5; ++op) {
        uint32_t bm = ips::bitrev32(ips::cmp_select(s, op));
        for (int j = 0; j < 32; ++j) {
          uint32_t x = vals[32 * q + j];
          bool e = op == 0 ? x == c : op == 1 ? x < c : op == 2 ? x <= c : op == 3 ? x > c : x >= c;
          CHECK((bool)((bm >> j) & 1u) == e);
        }
      }
      // the one-op-per-plane forms the kernels use (LSB -> MSB borrow / equality chains), also
      // composed from two halves as the w=32 early-pruning kernel does
      for (int op = 0; op < 5; ++op) {
        uint32_t acc = op == 0 ? ~0u : ips::borrow_init(op);
        for (int k = 0; k < W; ++k) {
          const uint32_t cm = ((c >> k) & 1u) ? ~0u : 0u;
          acc = op == 0 ? ips::eq_step(acc, p[k], cm) : ips::borrow_step(acc, p[k], cm);
        }
        const uint32_t sel = op == 0 ? acc : ips::borrow_select(acc, op);
        CHECK(sel == ips::cmp_select(s, op));
        const int split = W / 2;
        uint32_t blo = ips::borrow_init(op), eqlo = ~0u, bhi = 0u, eqhi = ~0u;
        for (int k = 0; k < split; ++k) {
          const uint32_t cm = ((c >> k) & 1u) ? ~0u : 0u;
          blo = ips::borrow_step(blo, p[k], cm); eqlo = ips::eq_step(eqlo, p[k], cm);
        }
        for (int k = split; k < W; ++k) {
          const uint32_t cm = ((c >> k) & 1u) ? ~0u : 0u;
          bhi = ips::borrow_step(bhi, p[k], cm); eqhi = ips::eq_step(eqhi, p[k], cm);
        }
        const uint32_t composed = op == 0 ? (eqhi & eqlo) : ips::borrow_select(bhi | (eqhi & blo), op);
        CHECK(composed == ips::cmp_select(s, op));
      }
      uint32_t ne = 0u;
      for (int k = 0; k < W; ++k) ne = ips::ne_step(ne, p[k], ((c >> k) & 1u) ? ~0u : 0u);
      CHECK(~ne == s.eq);
    }
    // half-transposed forms of the scan's index-list path
    if (W > 16) {
      uint32_t t[32];
      ips::planes_to_quads<W>(p, t);
      for (int j = 0; j < 32; ++j) {
        const uint32_t b = 31u - (uint32_t)j;
        CHECK(ips::quads_value(t[b & ~3u], t[(b & ~3u) + 1], t[(b & ~3u) + 2], t[(b & ~3u) + 3], b & 3u) == v[j]);
      }
      ips::quads_to_values(t);
      for (int j = 0; j < 32; ++j) CHECK(t[31 - j] == v[j]);
    } else if (W > 8) {
      uint32_t t[32];
      ips::planes_to_lane_quads16<W>(p, t);
      for (int j = 0; j < 32; ++j) {
        const uint32_t b = 31u - (uint32_t)j;
        const uint32_t base = b & 12u;
        CHECK(ips::quads_value(t[base], t[base + 1], t[base + 2], t[base + 3], (b & 16u) | (b & 3u), 0x1111u) == v[j]);
      }
    }
    {
      uint32_t a[32];
      ips::planes_to_lanes<W>(p, a);
      constexpr int R = ips::LaneWidth<W>::R;
      for (int j = 0; j < 32; ++j) {
        const int pos = 31 - j;
        if (R == 32) CHECK(a[pos] == v[j]);
        else CHECK(((a[pos % R] >> (R * (pos / R))) & ((1u << R) - 1u)) == v[j]);
      }
    }
  }
}

template <int W>
struct Run { static void go() { for (int r = 0; r < 20; ++r) test_width<W>(); Run<W - 1>::go(); } };
template <>
struct Run<0> { static void go() {} };

int main() {
  uint32_t a[32], b[32];
  for (int i = 0; i < 32; ++i) a[i] = b[i] = (uint32_t)rnd();
  ips::transpose32(b);
  for (int r = 0; r < 32; ++r) for (int i = 0; i < 32; ++i) CHECK(((b[r] >> i) & 1u) == ((a[i] >> r) & 1u));
  Run<32>::go();
  // geometry used by the kernels
  CHECK(ips::plane_tile_bytes(32) == 8448 && ips::plane_tile_bytes(31) == 7936);
  CHECK(ips::row_tile_dw(0) == 0 && ips::row_tile_dw(32) == 36 && ips::row_tile_dw(2047) == 63 * 36 + 31);
  printf("host_bitops_test: %d failures\n", fails);
  return fails ? 1 : 0;
}
