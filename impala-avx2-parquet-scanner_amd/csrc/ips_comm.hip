// ips_comm.hip -- the one exchange step of the multi-GPU scan: an all-gather of the per-stripe
// bitmap words with RCCL over xGMI, for hosts that do not go through torch.distributed.
//
// librccl is opened lazily (dlopen) the first time a communicator is requested, so single-GPU
// users of libips_hip.so do not need it at load time.  One process per GPU: every rank calls
// ips_comm_init with the same unique id (created by rank 0 with ips_comm_unique_id and passed
// around by the host's own bootstrap -- MPI, a file, torch.distributed ...).
#include <dlfcn.h>
#include <string.h>

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
};

extern "C" {

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
  *comm = new ips_comm{c, nranks, rank};
  return IPS_OK;
}

ips_status ips_comm_destroy(ips_comm* comm) {
  if (!comm) return IPS_OK;
  Rccl* r = rccl();
  if (r) r->CommDestroy(comm->comm);
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

}  // extern "C"
