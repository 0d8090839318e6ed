// ips_comm.hip -- the one exchange step of the multi-GPU scan: an all-gather of the per-stripe
// bitmap words with RCCL over xGMI, for hosts that do not go through torch.distributed.
//
// librccl is opened lazily (dlopen) the first time a communicator is requested, so single-GPU
// users of libips_hip.so do not need it at load time; types and prototypes come from
// <rccl/rccl.h>.  One process per GPU: every rank calls ips_comm_init with the same unique id
// (created by rank 0 with ips_comm_unique_id and passed around by the host's own bootstrap --
// MPI, a file, torch.distributed ...).
//
// The sharded step ("chunk the stripe ... overlap gather of chunk i with scan of chunk
// i+1") is ONE scan launch whose pages are the exchange pieces (blockIdx.y = piece, dispatched in
// order): every wave counts itself on its piece when its results are visible device-wide, the wave
// that completes a piece raises the piece's flag, and on the communicator's stream a one-wave kernel
// waits for that flag in front of the piece's ncclAllGather.  Round 2 launched one scan per piece
// (consecutive kernels of a stream do not overlap: every piece paid its own ramp-up and tail,
// 25-30 % of the scan).
#include <dlfcn.h>
#include <string.h>

#include <vector>

#include <rccl/rccl.h>

#include "ips_chunk_host.h"

static_assert(NCCL_UNIQUE_ID_BYTES == IPS_COMM_ID_BYTES, "include/ips.h states the size of ncclUniqueId");

namespace {

struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r.handle ? &r : nullptr;
  tried = true;
  // a copy already loaded by the process (torch ships one) is found first by soname
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (r.handle) break;
  }
  if (!r.handle) {
    ips::set_error("RCCL not found: %s", dlerror());
    return nullptr;
  }
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
  r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.handle, "ncclAllGather"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather) {
    ips::set_error("RCCL: missing symbols");
    dlclose(r.handle);
    r.handle = nullptr;
    return nullptr;
  }
  return &r;
}

ips_status rccl_fail(Rccl* r, ncclResult_t rc, const char* what) {
  ips::set_error("RCCL error %d (%s) in %s", (int)rc, r->GetErrorString ? r->GetErrorString(rc) : "?", what);
  return IPS_ERR_HIP;
}

constexpr int kMaxPieces = ips::kDonePages;
// device words of a communicator: the completion words of ips_chunk_device.h (page_done) + [kDoneWords] a
// waiter gave up
constexpr int kCommWords = ips::kDoneWords + 1;

// Waits (one wave, on the communicator's stream) until piece 'piece' of the step is complete.  It
// always ends: after ~2 s without the flag it records the timeout and lets the stream go on (the
// exchange then carries incomplete words, which ips_comm_check reports) -- a scan that never ran
// must not leave a kernel spinning on the device.
__global__ void wait_piece_kernel(const uint32_t* __restrict__ done, int piece, uint32_t epoch,
                                  uint32_t* __restrict__ timed_out) {
  const uint32_t* flags = done + ips::kDoneFlag;
  piece *= ips::kDoneLine;
  if (threadIdx.x != 0) return;
  const uint64_t t0 = wall_clock64();  // 100 MHz
  // (relaxed polls: an acquire load at device scope invalidates the L2 of the XCD this wave sits on
  // at every turn of the loop, under the scan's streaming waves -- 0.44 ms instead of 0.2x)
  while (__hip_atomic_load(flags + piece, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
    __builtin_amdgcn_s_sleep(127);
    if (wall_clock64() - t0 > 200000000ull) {
      *timed_out = 1u;
      break;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// the fallback when the last launch of a step cannot signal its pieces: all of them at once
__global__ void raise_flags_kernel(uint32_t* __restrict__ done, int n, uint32_t epoch) {
  if ((int)threadIdx.x < n)
    __hip_atomic_store(done + ips::kDoneFlag + threadIdx.x * ips::kDoneLine, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

namespace ips {
// ips_program.hip: the per-operand plan over chunks; the launches of the LAST operand count their
// waves on done[page] and raise flags[page] (ips_chunk_device.h: page_done).  *signalled = false:
// the last operand has no counting kernel (PLAIN pages, a merge of two bitmaps).
ips_status eval_program_chunks_signalled(const ips_node* nodes, int n_nodes, const ips_chunk* const* chunks, int n_chunks,
                                         uint64_t* d_bitmap, void* d_workspace, uint32_t* done, uint32_t done_epoch,
                                         bool* signalled, hipStream_t s);
}  // namespace ips

struct ips_comm {
  ncclComm_t comm;
  int nranks;
  int rank;
  // the exchange runs on the communicator's own stream; events order it against the scan stream
  hipStream_t stream = nullptr;
  std::vector<hipEvent_t> events;  // grow-only pool, reused round-robin (created outside captures)
  size_t next_event = 0;
  uint32_t* d_done = nullptr;      // kCommWords words
  uint32_t epoch = 0;              // the current step
  // the column of ips_fle_scan_allgather as a chunk of n_chunks pieces (page table uploaded once)
  ips_chunk* scan_chunk = nullptr;
  const void* scan_enc = nullptr;
  int64_t scan_rows = 0;
  int scan_bw = 0, scan_pieces = 0;
  hipEvent_t event() {
    if (events.size() < 384) {  // two steps of up to 64 pieces (a flush event each) + joins can be pending
      hipEvent_t e = nullptr;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
      events.push_back(e);
      return e;
    }
    return events[next_event++ % events.size()];
  }
};

using namespace ips;

namespace {

// The exchange of one step whose last launch signals its pieces: begin_step in front of the step's
// launches, then per piece a waiter and the all-gather on the communicator's stream.
ips_status begin_step(ips_comm* comm, hipStream_t s) {
  // the gathers of an earlier step may still be reading the local bitmap: this step's launches wait
  // for whatever the communicator's stream has been given so far.  Nothing is reset: the counters clean
  // up after themselves and a complete page's flag takes this step's epoch (page_done)
  ips_status st = ips_comm_join(comm, reinterpret_cast<ips_stream>(s));
  if (st != IPS_OK) return st;
  comm->epoch += 1;
  if (comm->epoch == 0) comm->epoch = 1;  // (0 is what the flags start as)
  return IPS_OK;
}

ips_status gather_pieces(ips_comm* comm, Rccl* r, int n_pieces, int64_t words_per_piece, const uint64_t* d_local,
                         uint64_t* d_all) {
  for (int i = 0; i < n_pieces; ++i) {
    hipLaunchKernelGGL(wait_piece_kernel, dim3(1), dim3(64), 0, comm->stream, comm->d_done, i, comm->epoch,
                       comm->d_done + ips::kDoneWords);
    IPS_HIP_TRY(hipGetLastError());
    // the end of the waiter is where the piece's bitmap words -- still dirty in the L2s of the XCDs that
    // produced them -- are written back: an event behind it makes that release explicit
    hipEvent_t flushed = comm->event();
    IPS_REQUIRE(flushed != nullptr, "sharded step: cannot create an event");
    IPS_HIP_TRY(hipEventRecord(flushed, comm->stream));
    // piece (i, rank) of the block-cyclic layout: piece i of all ranks is contiguous in the column
    ncclResult_t rc = r->AllGather(d_local + (size_t)i * words_per_piece,
                                   d_all + (size_t)i * (size_t)comm->nranks * words_per_piece, (size_t)words_per_piece,
                                   ncclUint64, comm->comm, comm->stream);
    if (rc != ncclSuccess) return rccl_fail(r, rc, "ncclAllGather");
  }
  return IPS_OK;
}

}  // namespace

extern "C" {

ips_status ips_comm_unique_id(void* id_bytes, int len) {
  IPS_REQUIRE(id_bytes && len >= IPS_COMM_ID_BYTES, "ips_comm_unique_id: need %d bytes", IPS_COMM_ID_BYTES);
  Rccl* r = rccl();
  if (!r) return IPS_ERR_UNSUPPORTED;
  ncclUniqueId id;
  ncclResult_t rc = r->GetUniqueId(&id);
  if (rc != ncclSuccess) return rccl_fail(r, rc, "ncclGetUniqueId");
  memcpy(id_bytes, &id, sizeof(id));
  return IPS_OK;
}

ips_status ips_comm_init(const void* id_bytes, int nranks, int rank, ips_comm** comm) {
  IPS_REQUIRE(id_bytes && comm && nranks >= 1 && rank >= 0 && rank < nranks, "ips_comm_init: bad argument");
  Rccl* r = rccl();
  if (!r) return IPS_ERR_UNSUPPORTED;
  ncclUniqueId id;
  memcpy(&id, id_bytes, sizeof(id));
  ncclComm_t c = nullptr;
  ncclResult_t rc = r->CommInitRank(&c, nranks, id, rank);
  if (rc != ncclSuccess) return rccl_fail(r, rc, "ncclCommInitRank");
  ips_comm* out = new ips_comm();
  out->comm = c;
  out->nranks = nranks;
  out->rank = rank;
  if (hipStreamCreateWithFlags(&out->stream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&out->d_done), kCommWords * sizeof(uint32_t)) != hipSuccess ||
      hipMemset(out->d_done, 0, kCommWords * sizeof(uint32_t)) != hipSuccess) {
    r->CommDestroy(c);
    if (out->stream) (void)hipStreamDestroy(out->stream);
    if (out->d_done) (void)hipFree(out->d_done);
    delete out;
    ips::set_error("ips_comm_init: cannot create the exchange stream / its flags");
    (void)hipGetLastError();
    return IPS_ERR_HIP;
  }
  *comm = out;
  return IPS_OK;
}

ips_status ips_comm_destroy(ips_comm* comm) {
  if (!comm) return IPS_OK;
  Rccl* r = rccl();
  if (comm->stream) (void)hipStreamSynchronize(comm->stream);
  if (r) r->CommDestroy(comm->comm);
  for (hipEvent_t e : comm->events) (void)hipEventDestroy(e);
  if (comm->stream) (void)hipStreamDestroy(comm->stream);
  if (comm->scan_chunk) (void)ips_chunk_close(comm->scan_chunk);
  if (comm->d_done) (void)hipFree(comm->d_done);
  delete comm;
  return IPS_OK;
}

ips_status ips_allgather_bitmap(ips_comm* comm, const uint64_t* d_local_words, int64_t n_words,
                                uint64_t* d_all_words, ips_stream stream) {
  IPS_REQUIRE(comm && n_words >= 0 && (n_words == 0 || (d_local_words && d_all_words)),
              "ips_allgather_bitmap: bad argument");
  if (n_words == 0) return IPS_OK;
  Rccl* r = rccl();
  if (!r) return IPS_ERR_UNSUPPORTED;
  ncclResult_t rc = r->AllGather(d_local_words, d_all_words, (size_t)n_words, ncclUint64, comm->comm,
                                 reinterpret_cast<hipStream_t>(stream));
  if (rc != ncclSuccess) return rccl_fail(r, rc, "ncclAllGather");
  return IPS_OK;
}

// One step of the sharded fused scan: ONE launch scans the rank's n_chunks pieces
// (blockIdx.y = piece); the all-gather of every piece starts on the communicator's stream as soon as
// the piece is complete, while the later pieces are still being scanned.
ips_status ips_fle_scan_allgather(ips_comm* comm, const void* d_enc, int64_t n_rows, int bit_width,
                                  ips_op op, const uint64_t* consts, int n_consts, int n_chunks,
                                  uint64_t* d_local_bitmap, uint32_t* d_batch_values,
                                  uint32_t* d_batch_counts, uint64_t* d_all_bitmap, ips_stream stream) {
  IPS_REQUIRE(comm != nullptr, "ips_fle_scan_allgather: NULL communicator");
  IPS_REQUIRE(n_chunks >= 1 && n_chunks <= kMaxPieces, "ips_fle_scan_allgather: 1..%d chunks per step", kMaxPieces);
  IPS_REQUIRE(n_rows >= 0 && n_rows % ((int64_t)n_chunks * IPS_BATCH_ROWS) == 0,
              "ips_fle_scan_allgather: n_rows must be n_chunks whole pieces of a multiple of %d rows", IPS_BATCH_ROWS);
  IPS_REQUIRE(bit_width >= 1 && bit_width <= 32, "ips_fle_scan_allgather: bit width %d", bit_width);
  IPS_REQUIRE(n_rows == 0 || (d_enc && aligned16(d_enc) && d_local_bitmap && aligned16(d_local_bitmap) && d_batch_values &&
                              aligned16(d_batch_values) && d_batch_counts && d_all_bitmap),
              "ips_fle_scan_allgather: NULL or misaligned argument");
  IPS_REQUIRE(op >= IPS_OP_EQ && op <= IPS_OP_IN && consts && n_consts >= 1 && n_consts <= IPS_MAX_IN_LIST &&
              (op == IPS_OP_IN || n_consts == 1), "ips_fle_scan_allgather: bad predicate");
  if (n_rows == 0) return IPS_OK;
  Rccl* r = rccl();
  if (!r) return IPS_ERR_UNSUPPORTED;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t rows_c = n_rows / n_chunks;
  const int64_t words_c = rows_c / 64;
  // the column as a chunk of n_chunks pieces: the page table is uploaded the first time this column
  // is seen (synchronous, like ips_chunk_open) and kept by the communicator
  if (!comm->scan_chunk || comm->scan_enc != d_enc || comm->scan_rows != n_rows || comm->scan_bw != bit_width ||
      comm->scan_pieces != n_chunks) {
    if (comm->scan_chunk) {
      (void)hipStreamSynchronize(s);  // (an earlier step may still be using the old table)
      (void)ips_chunk_close(comm->scan_chunk);
      comm->scan_chunk = nullptr;
    }
    std::vector<ips_chunk_page> pages((size_t)n_chunks);
    const uint8_t* enc = reinterpret_cast<const uint8_t*>(d_enc);
    for (int i = 0; i < n_chunks; ++i) {
      memset(&pages[(size_t)i], 0, sizeof(ips_chunk_page));
      pages[(size_t)i].d_data = enc + (size_t)i * (size_t)words_c * (size_t)bit_width * 8;
      pages[(size_t)i].n_rows = rows_c;
      pages[(size_t)i].bit_width = bit_width;
    }
    ips_status st = ips_chunk_open(pages.data(), n_chunks, IPS_COL_FLE, IPS_T_INT32, 0, &comm->scan_chunk);
    if (st != IPS_OK) return st;
    comm->scan_enc = d_enc;
    comm->scan_rows = n_rows;
    comm->scan_bw = bit_width;
    comm->scan_pieces = n_chunks;
  }
  ips_status st = begin_step(comm, s);
  if (st != IPS_OK) return st;
  PredArgs args;
  ConstKind kind;
  st = build_pred_args(bit_width, op, consts, n_consts, &args, &kind, "ips_fle_scan_allgather");
  if (st != IPS_OK) return st;
  if (kind != kEvaluate) {  // a constant outside the column's domain: always false (LT 0) / always true (GE 0)
    args.op = kind == kAllTrue ? IPS_OP_GE : IPS_OP_LT;
    args.n_consts = 1;
    args.consts[0] = 0u;
  }
  args.done = comm->d_done;
  args.done_page0 = 0;
  args.done_epoch = comm->epoch;
  const ips_chunk* c = comm->scan_chunk;
  st = launch_fle_scan_chunk(bit_width, args.op == IPS_OP_IN ? kScanInList : kScanPredicate, 0, c->d_pages,
                             (int)c->pages.size(), rows_c, n_rows, args, reinterpret_cast<uint32_t*>(d_local_bitmap),
                             nullptr, d_batch_values, d_batch_counts, nullptr, 0, nullptr, s);
  if (st != IPS_OK) {  // nothing will ever raise the flags: release the communicator's stream by hand
    hipLaunchKernelGGL(raise_flags_kernel, dim3(1), dim3(64), 0, s, comm->d_done, n_chunks, comm->epoch);
    return st;
  }
  if (const char* e = dev_env("IPS_SHARD_NO_EXCHANGE")) {  // dev: 1 the signalling scan alone, 2 + waiters, 3 + gathers
    for (int i = 0; i < n_chunks && e[0] != '1'; ++i) {
      if (e[0] == '2')
        hipLaunchKernelGGL(wait_piece_kernel, dim3(1), dim3(64), 0, comm->stream, comm->d_done, i, comm->epoch,
                           comm->d_done + ips::kDoneWords);
      else
        r->AllGather(d_local_bitmap + (size_t)i * words_c, d_all_bitmap + (size_t)i * (size_t)comm->nranks * words_c,
                     (size_t)words_c, ncclUint64, comm->comm, comm->stream);
    }
    return IPS_OK;
  }
  return gather_pieces(comm, r, n_chunks, words_c, d_local_bitmap, d_all_bitmap);
}

// The same for a predicate tree over several columns (configs[4]: the three Q6 columns sharded over
// the ranks): 'chunks' are the rank's column chunks, cut alike into the exchange pieces -- every
// chunk has the same pages, each a whole number of 64-row bitmap words -- and piece i of rank r is
// piece i * nranks + r of the whole column (block-cyclic), so that gathering piece i fills words
// [i * nranks * w, (i + 1) * nranks * w) of d_all_bitmap in natural row order.  The plan's launches
// run over all pieces; the launches of the LAST operand signal piece after piece, and the gathers of
// the early pieces overlap its later ones.
ips_status ips_eval_program_chunks_allgather(ips_comm* comm, const ips_node* nodes, int n_nodes,
                                             const ips_chunk* const* chunks, int n_chunks,
                                             uint64_t* d_local_bitmap, uint64_t* d_all_bitmap,
                                             void* d_workspace, ips_stream stream) {
  IPS_REQUIRE(comm != nullptr, "ips_eval_program_chunks_allgather: NULL communicator");
  IPS_REQUIRE(chunks && n_chunks >= 1 && n_chunks <= IPS_PROGRAM_MAX_COLS && chunks[0],
              "ips_eval_program_chunks_allgather: bad chunk list");
  const ips_chunk* c0 = chunks[0];
  const int n_pieces = (int)c0->pages.size();
  IPS_REQUIRE(n_pieces <= kMaxPieces, "ips_eval_program_chunks_allgather: at most %d pieces per step", kMaxPieces);
  if (c0->n_rows == 0) return IPS_OK;
  const int64_t piece_rows = c0->pages[0].n_rows;
  IPS_REQUIRE(piece_rows % 64 == 0, "ips_eval_program_chunks_allgather: a piece must be whole bitmap words (multiple of 64 rows)");
  for (int c = 0; c < n_chunks; ++c) {
    IPS_REQUIRE(chunks[c] && (int)chunks[c]->pages.size() == n_pieces, "ips_eval_program_chunks_allgather: chunk %d is cut differently", c);
    for (const ChunkPage& pg : chunks[c]->pages)
      IPS_REQUIRE(pg.n_rows == piece_rows, "ips_eval_program_chunks_allgather: chunk %d: every piece holds the same rows on every rank", c);
  }
  IPS_REQUIRE(d_local_bitmap && aligned16(d_local_bitmap) && d_all_bitmap, "ips_eval_program_chunks_allgather: NULL or misaligned bitmap");
  Rccl* r = rccl();
  if (!r) return IPS_ERR_UNSUPPORTED;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  ips_status st = begin_step(comm, s);
  if (st != IPS_OK) return st;
  bool signalled = false;
  st = eval_program_chunks_signalled(nodes, n_nodes, chunks, n_chunks, d_local_bitmap, d_workspace, comm->d_done,
                                     comm->epoch, &signalled, s);
  if (st != IPS_OK || !signalled)
    hipLaunchKernelGGL(raise_flags_kernel, dim3(1), dim3(64), 0, s, comm->d_done, n_pieces, comm->epoch);
  if (st != IPS_OK) return st;
  return gather_pieces(comm, r, n_pieces, piece_rows / 64, d_local_bitmap, d_all_bitmap);
}

// Make 'stream' wait for everything the communicator's stream has been given so far (the gathers
// of earlier sharded steps): call it before reading their buffers.
ips_status ips_comm_join(ips_comm* comm, ips_stream stream) {
  IPS_REQUIRE(comm != nullptr, "ips_comm_join: NULL communicator");
  hipEvent_t e = comm->event();
  IPS_REQUIRE(e != nullptr, "ips_comm_join: cannot create an event");
  IPS_HIP_TRY(hipEventRecord(e, comm->stream));
  IPS_HIP_TRY(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), e, 0));
  return IPS_OK;
}

// Synchronises with the communicator's stream and reports whether a waiter of a sharded step ever
// gave up (a piece that was not complete after ~2 s): IPS_ERR_HIP then, and the flag is cleared.
ips_status ips_comm_check(ips_comm* comm) {
  IPS_REQUIRE(comm != nullptr, "ips_comm_check: NULL communicator");
  IPS_HIP_TRY(hipStreamSynchronize(comm->stream));
  uint32_t timed_out = 0;
  IPS_HIP_TRY(hipMemcpy(&timed_out, comm->d_done + ips::kDoneWords, 4, hipMemcpyDeviceToHost));
  if (timed_out) {
    IPS_HIP_TRY(hipMemset(comm->d_done + ips::kDoneWords, 0, 4));
    ips::set_error("ips_comm_check: a piece of a sharded step was not complete after 2 s; its exchange carried stale words");
    return IPS_ERR_HIP;
  }
  return IPS_OK;
}

}  // extern "C"
