// RCCL seam of the C-ABI: the per-iteration gradient exchange of sharded
// mapping (ONE all-reduce (SUM) of a flat fp32 bucket holding
// map, decoder and bundle-adjustment pose gradients) on the caller's HIP
// stream, so that it orders with the kernels before and after it without a
// host round trip.  The reference has no collective at all.
//
// RCCL is bound at run time (dlopen): a process that already holds an RCCL
// instance — PyTorch bundles its own librccl.so — hands its path to
// xrd_comm_load and shares it instead of initialising a second library.
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include <rccl/rccl.h>

#include "common.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t,
                            ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
} g_rccl;

thread_local char g_comm_error[256];

int fail(const char* what, const char* detail) {
  snprintf(g_comm_error, sizeof(g_comm_error), "%s: %s", what,
           detail ? detail : "?");
  xrd::g_last_error = g_comm_error;
  return XRD_ERR_LAUNCH;
}

int rccl_check(ncclResult_t r, const char* what) {
  if (r == ncclSuccess) return XRD_OK;
  return fail(what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r)
                                          : "rccl error");
}

template <class F>
bool bind(F& fn, const char* name) {
  fn = reinterpret_cast<F>(dlsym(g_rccl.handle, name));
  return fn != nullptr;
}

}  // namespace

extern "C" {

int xrd_comm_load(const char* rccl_path) {
  if (g_rccl.handle != nullptr) return XRD_OK;
  const char* path = (rccl_path && rccl_path[0]) ? rccl_path : "librccl.so.1";
  g_rccl.handle = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (g_rccl.handle == nullptr) return fail("dlopen", dlerror());
  if (!bind(g_rccl.GetUniqueId, "ncclGetUniqueId") ||
      !bind(g_rccl.CommInitRank, "ncclCommInitRank") ||
      !bind(g_rccl.AllReduce, "ncclAllReduce") ||
      !bind(g_rccl.CommDestroy, "ncclCommDestroy") ||
      !bind(g_rccl.CommCount, "ncclCommCount") ||
      !bind(g_rccl.GetErrorString, "ncclGetErrorString")) {
    dlclose(g_rccl.handle);
    g_rccl.handle = nullptr;
    return fail("dlsym", "RCCL entry point missing");
  }
  return XRD_OK;
}

int xrd_comm_unique_id_bytes(void) { return NCCL_UNIQUE_ID_BYTES; }

int xrd_comm_unique_id(void* out_id) {
  if (out_id == nullptr) return XRD_ERR_ARG;
  if (g_rccl.handle == nullptr) return fail("xrd_comm_unique_id",
                                            "xrd_comm_load first");
  return rccl_check(g_rccl.GetUniqueId(static_cast<ncclUniqueId*>(out_id)),
                    "ncclGetUniqueId");
}

void* xrd_comm_create(const void* id, int rank, int world) {
  if (id == nullptr || world < 1 || rank < 0 || rank >= world) {
    fail("xrd_comm_create", "bad argument");
    return nullptr;
  }
  if (g_rccl.handle == nullptr) {
    fail("xrd_comm_create", "xrd_comm_load first");
    return nullptr;
  }
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclComm_t comm = nullptr;
  if (rccl_check(g_rccl.CommInitRank(&comm, world, uid, rank),
                 "ncclCommInitRank") != XRD_OK)
    return nullptr;
  return comm;
}

int xrd_comm_world(void* comm) {
  int n = 0;
  if (comm == nullptr || g_rccl.handle == nullptr) return 0;
  if (g_rccl.CommCount(static_cast<ncclComm_t>(comm), &n) != ncclSuccess)
    return 0;
  return n;
}

int xrd_allreduce_grads(void* comm, float* bucket, int64_t n,
                        xrd_stream_t stream) {
  if (comm == nullptr || n < 0) return XRD_ERR_ARG;
  if (n == 0) return XRD_OK;
  if (bucket == nullptr) return XRD_ERR_ARG;
  return rccl_check(
      g_rccl.AllReduce(bucket, bucket, (size_t)n, ncclFloat, ncclSum,
                       static_cast<ncclComm_t>(comm), (hipStream_t)stream),
      "ncclAllReduce");
}

int xrd_allreduce_max_i32(void* comm, int32_t* values, int64_t n,
                          xrd_stream_t stream) {
  if (comm == nullptr || n < 0) return XRD_ERR_ARG;
  if (n == 0) return XRD_OK;
  if (values == nullptr) return XRD_ERR_ARG;
  return rccl_check(
      g_rccl.AllReduce(values, values, (size_t)n, ncclInt32, ncclMax,
                       static_cast<ncclComm_t>(comm), (hipStream_t)stream),
      "ncclAllReduce");
}

void xrd_comm_destroy(void* comm) {
  if (comm != nullptr && g_rccl.handle != nullptr)
    g_rccl.CommDestroy(static_cast<ncclComm_t>(comm));
}

}  // extern "C"
