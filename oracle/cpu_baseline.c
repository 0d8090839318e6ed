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

__attribute__((target_clones("avx2", "default")))
static void unpack_block_u32(const uint64_t* blk, int bw, uint32_t* out64) {
  uint8_t b[4][64];
  int groups = (bw + 7) / 8;
  for (int c = 0; c < groups; ++c) {
    int np = bw - 8 * c < 8 ? bw - 8 * c : 8;
    planes8_to_bytes(blk + 8 * c, np, b[c]);
  }
  switch (groups) {
    case 1: for (int k = 0; k < 64; ++k) out64[k] = b[0][k]; break;
    case 2: for (int k = 0; k < 64; ++k) out64[k] = b[0][k] | ((uint32_t)b[1][k] << 8); break;
    case 3:
      for (int k = 0; k < 64; ++k)
        out64[k] = b[0][k] | ((uint32_t)b[1][k] << 8) | ((uint32_t)b[2][k] << 16);
      break;
    default:
      for (int k = 0; k < 64; ++k)
        out64[k] = b[0][k] | ((uint32_t)b[1][k] << 8) | ((uint32_t)b[2][k] << 16) |
                   ((uint32_t)b[3][k] << 24);
  }
}

/* exported for tests: the fast unpack must equal the scalar restatement */
void orc_fast_unpack_block(const uint64_t* blk, int bw, uint32_t* out64) {
  unpack_block_u32(blk, bw, out64);
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
