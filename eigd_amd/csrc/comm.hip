// The one data-path collective of the mode-sharded job: all-reduce (sum, fp64) of the partial df/dx vectors
// that the ranks accumulate over their own modes (the sum over i of reference
// eigd/eigenvector_derivatives.py:93-134 / 135-180), on the device buffer they already live in, over RCCL / xGMI.
//
// librccl is bound at run time (dlopen) the first time a communicator is asked for: single-GPU users and the
// CPU-side ABI tests never load it.  One process per GPU; the 128-byte unique id travels between the rank
// processes by whatever channel the host has (eigd_amd/comm.py: a file rendezvous on the node).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "common.h"

struct eigd_comm {
  eigd_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  int nranks = 1, rank = 0;
};

namespace eigd {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi g_rccl;
static std::mutex g_rccl_mutex;

static int load_rccl() {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (g_rccl.handle) return EIGD_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* nm : names) {
    h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
  }
  if (!h) {
    set_error("librccl not found: %s", dlerror());
    return EIGD_E_HIP;
  }
  RcclApi api;
  api.handle = h;
  api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(h, "ncclAllReduce"));
  api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.GetErrorString) {
    dlclose(h);
    set_error("librccl lacks an expected entry point");
    return EIGD_E_HIP;
  }
  g_rccl = api;
  return EIGD_OK;
}

#define EIGD_RCCL(call)                                                                          \
  do {                                                                                           \
    ncclResult_t _r = (call);                                                                    \
    if (_r != ncclSuccess) {                                                                     \
      ::eigd::set_error("%s failed: %s", #call, ::eigd::g_rccl.GetErrorString(_r));              \
      return EIGD_E_HIP;                                                                         \
    }                                                                                            \
  } while (0)

static int allreduce(eigd_comm* c, double* dbuf, int64_t len, ncclRedOp_t op) {
  EIGD_REQUIRE(c && dbuf, "null argument");
  EIGD_REQUIRE(len >= 0, "negative length");
  if (len == 0 || c->nranks == 1) return EIGD_OK;  // one rank: the buffer already holds the sum
  EIGD_HIP(hipSetDevice(c->ctx->device));
  EIGD_RCCL(g_rccl.AllReduce(dbuf, dbuf, static_cast<size_t>(len), ncclDouble, op, c->comm, c->ctx->stream));
  return EIGD_OK;
}

}  // namespace eigd

using namespace eigd;

extern "C" {

int eigd_comm_unique_id(void* hid128) {
  EIGD_REQUIRE(hid128, "null argument");
  static_assert(sizeof(ncclUniqueId) == EIGD_COMM_ID_BYTES, "unique id size");
  if (int rc = load_rccl()) return rc;
  ncclUniqueId id;
  EIGD_RCCL(g_rccl.GetUniqueId(&id));
  std::memcpy(hid128, &id, sizeof(id));
  return EIGD_OK;
}

int eigd_comm_init(eigd_ctx* ctx, int nranks, int rank, const void* hid128, eigd_comm** out) {
  EIGD_REQUIRE(ctx && out, "null argument");
  EIGD_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank %d of %d", rank, nranks);
  *out = nullptr;
  eigd_comm* c = new eigd_comm();
  c->ctx = ctx;
  c->nranks = nranks;
  c->rank = rank;
  if (nranks > 1) {
    if (!hid128) {
      delete c;
      set_error("a unique id is needed for more than one rank");
      return EIGD_E_INVALID;
    }
    if (int rc = load_rccl()) {
      delete c;
      return rc;
    }
    ncclUniqueId id;
    std::memcpy(&id, hid128, sizeof(id));
    hipError_t he = hipSetDevice(ctx->device);
    ncclResult_t r = (he == hipSuccess) ? g_rccl.CommInitRank(&c->comm, nranks, id, rank) : ncclUnhandledCudaError;
    if (r != ncclSuccess) {
      set_error("ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, nranks, ctx->device,
                g_rccl.GetErrorString(r));
      delete c;
      return EIGD_E_HIP;
    }
  }
  *out = c;
  return EIGD_OK;
}

int eigd_comm_destroy(eigd_comm* c) {
  if (!c) return EIGD_OK;
  if (c->comm) {
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    (void)g_rccl.CommDestroy(c->comm);
  }
  delete c;
  return EIGD_OK;
}

int eigd_comm_info(eigd_comm* c, int* nranks, int* rank) {
  EIGD_REQUIRE(c && nranks && rank, "null argument");
  *nranks = c->nranks;
  *rank = c->rank;
  return EIGD_OK;
}

int eigd_allreduce_sum(eigd_comm* c, double* dbuf, int64_t len) { return allreduce(c, dbuf, len, ncclSum); }

int eigd_allreduce_max(eigd_comm* c, double* dbuf, int64_t len) { return allreduce(c, dbuf, len, ncclMax); }

}  // extern "C"
