// ips_comm.hip -- the one exchange step of the multi-GPU scan: an all-gather of the per-stripe
// bitmap words with RCCL over xGMI, for hosts that do not go through torch.distributed.
//
// librccl is opened lazily (dlopen) the first time a communicator is requested, so single-GPU
// users of libips_hip.so do not need it at load time.  One process per GPU: every rank calls
// ips_comm_init with the same unique id (created by rank 0 with ips_comm_unique_id and passed
// around by the host's own bootstrap -- MPI, a file, torch.distributed ...).
#include <dlfcn.h>
#include <string.h>

#include <vector>

#include "ips_host.h"

namespace {

// the few RCCL declarations needed (rccl/rccl.h; the ABI is NCCL's)
typedef struct { char internal[128]; } NcclUniqueId;
typedef void* NcclComm;
enum { kNcclSuccess = 0, kNcclUint64 = 5 };  // ncclDataType_t: ncclUint64 = 5

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(NcclUniqueId*) = nullptr;
  int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
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

ips_status rccl_fail(Rccl* r, int rc, const char* what) {
  ips::set_error("RCCL error %d (%s) in %s", rc, r->GetErrorString ? r->GetErrorString(rc) : "?", what);
  return IPS_ERR_HIP;
}

}  // namespace

struct ips_comm {
  NcclComm comm;
  int nranks;
  int rank;
  // the exchange runs on the communicator's own stream; events order it against the scan stream
  hipStream_t stream = nullptr;
  std::vector<hipEvent_t> events;  // grow-only pool, reused round-robin (created outside captures)
  size_t next_event = 0;
  hipEvent_t event() {
    if (events.size() < 192) {  // two steps of up to 64 chunk events + joins can be pending
      hipEvent_t e = nullptr;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
      events.push_back(e);
      return e;
    }
    return events[next_event++ % events.size()];
  }
};

extern "C" {

ips_status ips_comm_join(ips_comm* comm, ips_stream stream);

ips_status ips_comm_unique_id(void* id_bytes, int len) {
  IPS_REQUIRE(id_bytes && len >= IPS_COMM_ID_BYTES, "ips_comm_unique_id: need %d bytes", IPS_COMM_ID_BYTES);
  Rccl* r = rccl();
  if (!r) return IPS_ERR_UNSUPPORTED;
  NcclUniqueId id;
  int rc = r->GetUniqueId(&id);
  if (rc != kNcclSuccess) return rccl_fail(r, rc, "ncclGetUniqueId");
  memcpy(id_bytes, &id, sizeof(id));
  return IPS_OK;
}

ips_status ips_comm_init(const void* id_bytes, int nranks, int rank, ips_comm** comm) {
  IPS_REQUIRE(id_bytes && comm && nranks >= 1 && rank >= 0 && rank < nranks, "ips_comm_init: bad argument");
  Rccl* r = rccl();
  if (!r) return IPS_ERR_UNSUPPORTED;
  NcclUniqueId id;
  memcpy(&id, id_bytes, sizeof(id));
  NcclComm c = nullptr;
  int rc = r->CommInitRank(&c, nranks, id, rank);
  if (rc != kNcclSuccess) return rccl_fail(r, rc, "ncclCommInitRank");
  ips_comm* out = new ips_comm();
  out->comm = c;
  out->nranks = nranks;
  out->rank = rank;
  if (hipStreamCreateWithFlags(&out->stream, hipStreamNonBlocking) != hipSuccess) {
    r->CommDestroy(c);
    delete out;
    ips::set_error("ips_comm_init: cannot create the exchange stream");
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
  int rc = r->AllGather(d_local_words, d_all_words, (size_t)n_words, kNcclUint64, comm->comm,
                        reinterpret_cast<hipStream_t>(stream));
  if (rc != kNcclSuccess) return rccl_fail(r, rc, "ncclAllGather");
  return IPS_OK;
}

// One step of the sharded fused scan: chunk i is scanned on 'stream'; as soon as it
// has finished, its bitmap words are all-gathered on the communicator's stream while chunk i+1 is
// being scanned.  All launches of the step are issued by this one call.
ips_status ips_fle_scan_allgather(ips_comm* comm, const void* d_enc, int64_t n_rows, int bit_width,
                                  ips_op op, const uint64_t* consts, int n_consts, int n_chunks,
                                  uint64_t* d_local_bitmap, uint32_t* d_batch_values,
                                  uint32_t* d_batch_counts, uint64_t* d_all_bitmap, ips_stream stream) {
  IPS_REQUIRE(comm != nullptr, "ips_fle_scan_allgather: NULL communicator");
  IPS_REQUIRE(n_chunks >= 1 && n_rows >= 0 && n_rows % ((int64_t)n_chunks * IPS_BATCH_ROWS) == 0,
              "ips_fle_scan_allgather: n_rows must be n_chunks whole pieces of a multiple of %d rows", IPS_BATCH_ROWS);
  IPS_REQUIRE(bit_width >= 1 && bit_width <= 32, "ips_fle_scan_allgather: bit width %d", bit_width);
  IPS_REQUIRE(n_rows == 0 || (d_enc && d_local_bitmap && d_batch_values && d_batch_counts && d_all_bitmap),
              "ips_fle_scan_allgather: NULL argument");
  if (n_rows == 0) return IPS_OK;
  Rccl* r = rccl();
  if (!r) return IPS_ERR_UNSUPPORTED;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t rows_c = n_rows / n_chunks;
  const int64_t words_c = rows_c / 64;
  const uint8_t* enc = reinterpret_cast<const uint8_t*>(d_enc);
  // all chunk scans (and the event after each) go to the scan stream first, the gathers second: an
  // ncclAllGather call takes the host 10-20 us, and issued between the scans it left the GPU idle
  // between chunk kernels
  IPS_REQUIRE(n_chunks <= 64, "ips_fle_scan_allgather: at most 64 chunks per step");
  // the gathers of an earlier step may still be reading d_local_bitmap: the scans of this step wait
  // for whatever the communicator's stream has been given so far (what ips_comm_join does), so the
  // buffers can be reused step after step without a caller-side contract
  {
    ips_status jst = ips_comm_join(comm, stream);
    if (jst != IPS_OK) return jst;
  }
  hipEvent_t done[64];
  for (int i = 0; i < n_chunks; ++i) {
    ips_status st = ips_fle_scan(enc + (size_t)i * (size_t)words_c * (size_t)bit_width * 8, rows_c, bit_width,
                                 op, consts, n_consts, d_local_bitmap + (size_t)i * words_c,
                                 d_batch_values + (size_t)i * rows_c,
                                 d_batch_counts + (size_t)i * (rows_c / IPS_BATCH_ROWS), stream);
    if (st != IPS_OK) return st;
    done[i] = comm->event();
    IPS_REQUIRE(done[i] != nullptr, "ips_fle_scan_allgather: cannot create an event");
    IPS_HIP_TRY(hipEventRecord(done[i], s));
  }
  for (int i = 0; i < n_chunks; ++i) {
    IPS_HIP_TRY(hipStreamWaitEvent(comm->stream, done[i], 0));
    // piece (i, rank) of the block-cyclic layout: chunk i of all ranks is contiguous in the column
    int rc = r->AllGather(d_local_bitmap + (size_t)i * words_c,
                          d_all_bitmap + (size_t)i * (size_t)comm->nranks * words_c, (size_t)words_c,
                          kNcclUint64, comm->comm, comm->stream);
    if (rc != kNcclSuccess) return rccl_fail(r, rc, "ncclAllGather");
  }
  return IPS_OK;
}

// Make 'stream' wait for everything the communicator's stream has been given so far (the gathers
// of earlier ips_fle_scan_allgather calls): call it before reusing or reading their buffers.
ips_status ips_comm_join(ips_comm* comm, ips_stream stream) {
  IPS_REQUIRE(comm != nullptr, "ips_comm_join: NULL communicator");
  hipEvent_t e = comm->event();
  IPS_REQUIRE(e != nullptr, "ips_comm_join: cannot create an event");
  IPS_HIP_TRY(hipEventRecord(e, comm->stream));
  IPS_HIP_TRY(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), e, 0));
  return IPS_OK;
}

}  // extern "C"
