/*
 * oracle/cpu_baseline.c -- TEST INFRASTRUCTURE ONLY: the timed CPU "port" baseline that
 * bench.py's cpu_baseline leg runs on the GPU box's host cores.  Never linked into the product.
 *
 * Workload shape follows the reference's vectorised scan of one column
 * (hdfs-parquet-scanner.cc:1101-1182): predicate on the encoded 64-row words
 * (fle-encoding.h:8012-8066, scalar uint64 ops exactly like the reference), then late
 * materialisation of the selected rows, where every block that holds a selected row is unpacked
 * once (what Get(val, skip) does, fle-encoding.h:344-379).  The block unpack here is a
 * from-scratch SWAR 8x8 bit-matrix transpose (the reference uses hand-unrolled AVX2
 * shuffle/cmpeq sequences, fle-encoding.h:569-7329); gcc vectorises the widening loops under
 * the avx2 target clone when the host supports it.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "fle_oracle.h"

/* Host threads this process can really use: min(online CPUs, affinity mask, cgroup CPU quota).
 * A GPU box advertises all host CPUs but the container gets a share of them. */
int orc_hw_threads(void) {
  long n = sysconf(_SC_NPROCESSORS_ONLN);
  if (n < 1) n = 1;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) {
    int c = CPU_COUNT(&set);
    if (c > 0 && c < n) n = c;
  }
  FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r"); /* cgroup v2: "<quota> <period>" or "max ..." */
  if (f) {
    char q[64];
    long period = 0;
    if (fscanf(f, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
      long quota = atol(q);
      long c = (quota + period - 1) / period;
      if (c > 0 && c < n) n = c;
    }
    fclose(f);
  } else {
    long quota = -1, period = 0; /* cgroup v1 */
    FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r");
    FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
    if (fq && fp && fscanf(fq, "%ld", &quota) == 1 && fscanf(fp, "%ld", &period) == 1 &&
        quota > 0 && period > 0) {
      long c = (quota + period - 1) / period;
      if (c > 0 && c < n) n = c;
    }
    if (fq) fclose(fq);
    if (fp) fclose(fp);
  }
  return (int)n;
}

int orc_has_avx2(void) {
#if defined(__x86_64__)
  __builtin_cpu_init();
  return __builtin_cpu_supports("avx2") ? 1 : 0;
#else
  return 0;
#endif
}

static inline uint64_t rev64(uint64_t m) {
  m = ((m >> 1) & 0x5555555555555555ULL) | ((m & 0x5555555555555555ULL) << 1);
  m = ((m >> 2) & 0x3333333333333333ULL) | ((m & 0x3333333333333333ULL) << 2);
  m = ((m >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((m & 0x0F0F0F0F0F0F0F0FULL) << 4);
  return __builtin_bswap64(m);
}

/* 8 planes (missing ones zero) of one block -> 64 bytes, byte k = bits of row k. */
static inline void planes8_to_bytes(const uint64_t* p, int np, uint8_t* out64) {
  for (int g = 0; g < 8; ++g) {
    uint64_t m = 0;
    for (int i = 0; i < np; ++i) m |= ((p[i] >> (8 * g)) & 0xFFULL) << (8 * i);
    /* 8x8 bit-matrix transpose (three delta swaps): byte j gets bit j of every input byte */
    uint64_t t;
    t = (m ^ (m >> 7)) & 0x00AA00AA00AA00AAULL;  m = m ^ t ^ (t << 7);
    t = (m ^ (m >> 14)) & 0x0000CCCC0000CCCCULL; m = m ^ t ^ (t << 14);
    t = (m ^ (m >> 28)) & 0x00000000F0F0F0F0ULL; m = m ^ t ^ (t << 28);
    /* byte j of m <-> plane bit 8g+j <-> row 63-(8g+j) */
    m = __builtin_bswap64(m);
    memcpy(out64 + 56 - 8 * g, &m, 8);
  }
}

static void unpack_block_u32_swar(const uint64_t* blk, int bw, uint32_t* out64) {
  uint8_t b[4][64];
  int groups = (bw + 7) / 8;
  for (int c = 0; c < groups; ++c) {
    int np = bw - 8 * c < 8 ? bw - 8 * c : 8;
    planes8_to_bytes(blk + 8 * c, np, b[c]);
  }
  for (int k = 0; k < 64; ++k) {
    uint32_t v = b[0][k];
    if (groups > 1) v |= (uint32_t)b[1][k] << 8;
    if (groups > 2) v |= (uint32_t)b[2][k] << 16;
    if (groups > 3) v |= (uint32_t)b[3][k] << 24;
    out64[k] = v;
  }
}

#if defined(__x86_64__)
#include <immintrin.h>
/* The same unpack with AVX2 (written for this baseline, not the reference's shuffle/cmpeq scheme,
 * fle-encoding.h:569-7329): per group of 8 planes, an 8x8 BYTE transpose (unpack ladder) puts
 * byte g of every plane into one 64-bit lane, the three delta swaps of the 8x8 BIT transpose then
 * run on four lanes at a time, and reversing the 64 bytes puts row k at byte k. */
__attribute__((target("avx2")))
static void group_to_bytes_avx2(const uint64_t* p, int np, uint8_t* out64) {
  uint64_t pl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < np; ++i) pl[i] = p[i];
  __m128i r01 = _mm_loadu_si128((const __m128i*)(pl + 0));
  __m128i r23 = _mm_loadu_si128((const __m128i*)(pl + 2));
  __m128i r45 = _mm_loadu_si128((const __m128i*)(pl + 4));
  __m128i r67 = _mm_loadu_si128((const __m128i*)(pl + 6));
  __m128i i01 = _mm_unpacklo_epi8(r01, _mm_unpackhi_epi64(r01, r01));
  __m128i i23 = _mm_unpacklo_epi8(r23, _mm_unpackhi_epi64(r23, r23));
  __m128i i45 = _mm_unpacklo_epi8(r45, _mm_unpackhi_epi64(r45, r45));
  __m128i i67 = _mm_unpacklo_epi8(r67, _mm_unpackhi_epi64(r67, r67));
  __m128i a_lo = _mm_unpacklo_epi16(i01, i23), a_hi = _mm_unpackhi_epi16(i01, i23);
  __m128i b_lo = _mm_unpacklo_epi16(i45, i67), b_hi = _mm_unpackhi_epi16(i45, i67);
  /* lane g of (m0123 | m4567): byte i = byte g of plane i */
  __m256i m0123 = _mm256_set_m128i(_mm_unpackhi_epi32(a_lo, b_lo), _mm_unpacklo_epi32(a_lo, b_lo));
  __m256i m4567 = _mm256_set_m128i(_mm_unpackhi_epi32(a_hi, b_hi), _mm_unpacklo_epi32(a_hi, b_hi));
#define ORC_DELTA(m, s, k)                                                          \
  do {                                                                              \
    __m256i t_ = _mm256_and_si256(_mm256_xor_si256(m, _mm256_srli_epi64(m, s)),     \
                                  _mm256_set1_epi64x((long long)(k)));              \
    m = _mm256_xor_si256(_mm256_xor_si256(m, t_), _mm256_slli_epi64(t_, s));        \
  } while (0)
  ORC_DELTA(m0123, 7, 0x00AA00AA00AA00AAULL);  ORC_DELTA(m4567, 7, 0x00AA00AA00AA00AAULL);
  ORC_DELTA(m0123, 14, 0x0000CCCC0000CCCCULL); ORC_DELTA(m4567, 14, 0x0000CCCC0000CCCCULL);
  ORC_DELTA(m0123, 28, 0x00000000F0F0F0F0ULL); ORC_DELTA(m4567, 28, 0x00000000F0F0F0F0ULL);
#undef ORC_DELTA
  /* byte j of lane g <-> plane bit 8g+j <-> row 63-(8g+j): reverse all 64 bytes */
  const __m256i rev = _mm256_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0,
                                       15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
  __m256i hi = _mm256_permute4x64_epi64(_mm256_shuffle_epi8(m4567, rev), 0x4E);
  __m256i lo = _mm256_permute4x64_epi64(_mm256_shuffle_epi8(m0123, rev), 0x4E);
  _mm256_storeu_si256((__m256i*)out64, hi);
  _mm256_storeu_si256((__m256i*)(out64 + 32), lo);
}

__attribute__((target("avx2")))
static void unpack_block_u32_avx2(const uint64_t* blk, int bw, uint32_t* out64) {
  uint8_t b[4][64] __attribute__((aligned(32)));
  int groups = (bw + 7) / 8;
  for (int c = 0; c < 4; ++c) {
    if (c < groups) group_to_bytes_avx2(blk + 8 * c, bw - 8 * c < 8 ? bw - 8 * c : 8, b[c]);
    else memset(b[c], 0, 64);
  }
  for (int h = 0; h < 2; ++h) { /* 32 rows per half: interleave the four byte arrays into dwords */
    __m256i b0 = _mm256_load_si256((const __m256i*)(b[0] + 32 * h));
    __m256i b1 = _mm256_load_si256((const __m256i*)(b[1] + 32 * h));
    __m256i b2 = _mm256_load_si256((const __m256i*)(b[2] + 32 * h));
    __m256i b3 = _mm256_load_si256((const __m256i*)(b[3] + 32 * h));
    __m256i w01l = _mm256_unpacklo_epi8(b0, b1), w01h = _mm256_unpackhi_epi8(b0, b1);
    __m256i w23l = _mm256_unpacklo_epi8(b2, b3), w23h = _mm256_unpackhi_epi8(b2, b3);
    __m256i d0 = _mm256_unpacklo_epi16(w01l, w23l), d1 = _mm256_unpackhi_epi16(w01l, w23l);
    __m256i d2 = _mm256_unpacklo_epi16(w01h, w23h), d3 = _mm256_unpackhi_epi16(w01h, w23h);
    /* 128-bit lane 0 of d0..d3 = rows 0..15, lane 1 = rows 16..31 */
    uint32_t* o = out64 + 32 * h;
    _mm256_storeu_si256((__m256i*)(o + 0), _mm256_permute2x128_si256(d0, d1, 0x20));
    _mm256_storeu_si256((__m256i*)(o + 8), _mm256_permute2x128_si256(d2, d3, 0x20));
    _mm256_storeu_si256((__m256i*)(o + 16), _mm256_permute2x128_si256(d0, d1, 0x31));
    _mm256_storeu_si256((__m256i*)(o + 24), _mm256_permute2x128_si256(d2, d3, 0x31));
  }
}
#endif

static int g_use_avx2 = -1;
static void unpack_block_u32(const uint64_t* blk, int bw, uint32_t* out64) {
#if defined(__x86_64__)
  if (g_use_avx2 < 0) g_use_avx2 = orc_has_avx2();
  if (g_use_avx2) { unpack_block_u32_avx2(blk, bw, out64); return; }
#endif
  unpack_block_u32_swar(blk, bw, out64);
}

/* exported for tests: the fast unpack must equal the scalar restatement */
void orc_fast_unpack_block(const uint64_t* blk, int bw, uint32_t* out64) {
  unpack_block_u32(blk, bw, out64);
}
void orc_swar_unpack_block(const uint64_t* blk, int bw, uint32_t* out64) {
  unpack_block_u32_swar(blk, bw, out64);
}

static inline uint64_t block_pred(const uint64_t* blk, int bw, int op, uint64_t value) {
  uint64_t lt = 0, eq = ~0ULL;
  for (int i = bw - 1; i >= 0; --i) {
    uint64_t c = ((value >> i) & 1) ? ~0ULL : 0ULL;
    uint64_t x = blk[i];
    lt |= eq & c & ~x;
    eq &= ~(x ^ c);
  }
  uint64_t m;
  switch (op) {
    case ORC_OP_EQ: m = eq; break;
    case ORC_OP_LT: m = lt; break;
    case ORC_OP_LE: m = lt | eq; break;
    case ORC_OP_GT: m = ~(lt | eq); break;
    default: m = ~lt; break; /* GE */
  }
  return rev64(m);
}

typedef struct {
  const uint64_t* enc;
  int64_t row0, row1; /* stripe, 64-aligned start */
  int bw, op, mode;
  uint64_t value;
  uint64_t* bitmap;
  uint32_t* sel_out; /* stripe-local region starting at sel_out + row0 */
  int64_t n_sel;
} stripe_job;

static void run_range(stripe_job* j, int64_t r0, int64_t r1, uint32_t** wr) {
  /* predicate pass over [r0, r1) then materialise, as one reference batch does */
  for (int64_t b = r0 / 64; b * 64 < r1; ++b) {
    uint64_t m = block_pred(j->enc + b * j->bw, j->bw, j->op, j->value);
    int64_t left = r1 - b * 64;
    if (left < 64) m &= (1ULL << left) - 1;
    j->bitmap[b] = m;
  }
  uint32_t tmp[64];
  for (int64_t b = r0 / 64; b * 64 < r1; ++b) {
    uint64_t m = j->bitmap[b];
    if (!m) continue; /* Skip(): whole blocks are jumped by pointer arithmetic */
    unpack_block_u32(j->enc + b * j->bw, j->bw, tmp);
    uint32_t* w = *wr;
    while (m) {
      int k = __builtin_ctzll(m);
      *w++ = tmp[k];
      m &= m - 1;
    }
    *wr = w;
  }
}

static void* stripe_main(void* arg) {
  stripe_job* j = (stripe_job*)arg;
  uint32_t* wr = j->sel_out + j->row0;
  if (j->mode == 0) {
    run_range(j, j->row0, j->row1, &wr);
  } else {
    /* reference-shaped batches of 1024 rows, hdfs-parquet-scanner.cc:1838 */
    for (int64_t r = j->row0; r < j->row1; r += 1024)
      run_range(j, r, r + 1024 < j->row1 ? r + 1024 : j->row1, &wr);
  }
  j->n_sel = wr - (j->sel_out + j->row0);
  return NULL;
}

int64_t orc_bench_fused(const uint64_t* enc, int64_t n, int bw, int op, uint64_t value,
                        int threads, int mode, uint64_t* bitmap, uint32_t* sel_out) {
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  stripe_job jobs[256];
  pthread_t tid[256];
  int64_t blocks = (n + 63) / 64;
  int64_t per = ((blocks + threads - 1) / threads + 15) & ~15LL; /* 1024-row aligned stripes */
  int used = 0;
  for (int t = 0; t < threads; ++t) {
    int64_t b0 = per * t, b1 = per * (t + 1);
    if (b0 >= blocks) break;
    if (b1 > blocks) b1 = blocks;
    stripe_job* j = &jobs[used++];
    j->enc = enc; j->row0 = b0 * 64; j->row1 = b1 * 64 < n ? b1 * 64 : n;
    j->bw = bw; j->op = op; j->mode = mode; j->value = value;
    j->bitmap = bitmap; j->sel_out = sel_out; j->n_sel = 0;
  }
  for (int t = 1; t < used; ++t) pthread_create(&tid[t], NULL, stripe_main, &jobs[t]);
  stripe_main(&jobs[0]);
  int64_t total = jobs[0].n_sel;
  for (int t = 1; t < used; ++t) { pthread_join(tid[t], NULL); total += jobs[t].n_sel; }
  return total;
}

/* Best-of-reps wall time of orc_bench_fused, measured here so that nothing but the scan is timed:
 * the caller hands in bitmap / sel_out buffers it has already touched (a fresh 4n-byte output
 * would add one page fault per 4 KiB to every run), one untimed warm-up pass runs first. */
double orc_bench_fused_best(const uint64_t* enc, int64_t n, int bw, int op, uint64_t value,
                            int threads, int mode, uint64_t* bitmap, uint32_t* sel_out, int reps,
                            int64_t* n_sel) {
  int64_t cnt = orc_bench_fused(enc, n, bw, op, value, threads, mode, bitmap, sel_out);
  double best = 1e30;
  for (int r = 0; r < reps; ++r) {
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    cnt = orc_bench_fused(enc, n, bw, op, value, threads, mode, bitmap, sel_out);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    double dt = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    if (dt < best) best = dt;
  }
  if (n_sel) *n_sel = cnt;
  return best;
}
