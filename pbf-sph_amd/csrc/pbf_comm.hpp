// pbf_comm.hpp — the slab decomposition's neighbour exchange behind the C ABI (include/pbf_hip.h "communicator").
// No reference counterpart: the reference is single-device.
//
// Two transports, one interface (exchange with the left and the right slab in ONE round):
//   * RCCL over xGMI: ncclGroupStart; ncclSend / ncclRecv per neighbour; ncclGroupEnd — enqueued on the solver's
//     stream, so the exchange is ordered against the kernels that pack and unpack the wire buffers without any
//     host synchronisation.  librccl is resolved at run time (dlopen: the copy a hosting process — e.g. PyTorch —
//     has already loaded, else the ROCm one), so libpbf_hip.so carries no link-time dependency on it.
//   * host callback (bring-up and tests: several ranks sharing ONE GPU under gloo): the wire buffers are staged
//     through pinned host memory and the caller's function moves the bytes.
#pragma once

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>
#include <type_traits>

#include "pbf_hip.h"

struct pbf_comm {
  int nranks = 1, rank = 0, device = 0;
  // ---- RCCL ----
  void *lib = nullptr;
  ncclComm_t comm = nullptr;
  ncclResult_t (*fGetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*fCommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*fCommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*fSend)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*fRecv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*fGroupStart)() = nullptr;
  ncclResult_t (*fGroupEnd)() = nullptr;
  ncclResult_t (*fAllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*fErrorString)(ncclResult_t) = nullptr;
  // ---- host callback ----
  pbf_exchange_fn fn = nullptr;
  void *user = nullptr;
  void *host[4] = {nullptr, nullptr, nullptr, nullptr};  // pinned staging: send L/R, recv L/R
  size_t hostCap[4] = {0, 0, 0, 0};
  std::string err;
  uint64_t rounds = 0;  // exchange rounds so far (diagnostic: pbf_comm_rounds)
};

namespace pbf {

// RCCL only works when it and this library drive the SAME HIP runtime.  A process that loads libpbf_hip.so (linked
// against ROCm's libamdhip64) and then PyTorch (which ships a libamdhip64 of its own, and the librccl bound to it) holds
// two: ncclCommInitRank then fails with "unhandled cuda error" somewhere inside.  Refuse with the reason instead:
// the runtime this library's HIP calls resolve to must be the one librccl's resolve to.
inline bool comm_same_hip_runtime(pbf_comm *c) {
  Dl_info mine{}, theirs{};
  void *rcclHip = dlsym(c->lib, "hipGetDeviceCount");  // searched in librccl and ITS dependencies
  if (!rcclHip || !dladdr(rcclHip, &theirs) || !dladdr(reinterpret_cast<void *>(&hipGetDeviceCount), &mine)) return true;
  if (!mine.dli_fname || !theirs.dli_fname || std::strcmp(mine.dli_fname, theirs.dli_fname) == 0) return true;
  c->err = std::string("two HIP runtimes in this process: libpbf_hip.so uses ") + mine.dli_fname + ", the loaded librccl uses " +
           theirs.dli_fname + " — load the hosting framework (e.g. import torch) BEFORE libpbf_hip.so, or run without it";
  return false;
}

inline bool comm_load_rccl(pbf_comm *c) {
  const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char *n : names)  // a copy the process already holds (PyTorch ships its own) wins: one RCCL per process
    if ((c->lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
  if (!c->lib)
    for (const char *n : names)
      if ((c->lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!c->lib) {
    c->err = std::string("cannot load librccl: ") + dlerror();
    return false;
  }
  auto sym = [&](const char *name, auto &fp) {
    fp = reinterpret_cast<std::remove_reference_t<decltype(fp)>>(dlsym(c->lib, name));
    if (!fp) c->err = std::string("librccl lacks ") + name;
    return fp != nullptr;
  };
  if (!comm_same_hip_runtime(c)) return false;
  return sym("ncclGetUniqueId", c->fGetUniqueId) && sym("ncclCommInitRank", c->fCommInitRank) &&
         sym("ncclCommDestroy", c->fCommDestroy) && sym("ncclSend", c->fSend) && sym("ncclRecv", c->fRecv) &&
         sym("ncclGroupStart", c->fGroupStart) && sym("ncclGroupEnd", c->fGroupEnd) &&
         sym("ncclAllReduce", c->fAllReduce) && sym("ncclGetErrorString", c->fErrorString);
}

inline int comm_fail(pbf_comm *c, const char *what, ncclResult_t r) {
  c->err = std::string(what) + ": " + (c->fErrorString ? c->fErrorString(r) : "RCCL error");
  return PBF_ERR_COMM;
}

// One exchange round with both neighbours.  All four buffers are DEVICE memory; a size of 0 skips that message
// (both ends of a link always agree on its size: it is either a capacity fixed at attach time or a count both
// learnt in the assembly round).  Rank r's left neighbour is r - 1, its right neighbour r + 1.
inline int comm_exchange(pbf_comm *c, hipStream_t stream, const void *sendL, size_t nSL, const void *sendR, size_t nSR,
                         void *recvL, size_t nRL, void *recvR, size_t nRR) {
  const bool hasL = c->rank > 0, hasR = c->rank + 1 < c->nranks;
  if (!hasL) nSL = nRL = 0;
  if (!hasR) nSR = nRR = 0;
  if (nSL + nSR + nRL + nRR == 0) return PBF_OK;
  c->rounds++;
  if (c->comm) {
    ncclResult_t r;
    if ((r = c->fGroupStart()) != ncclSuccess) return comm_fail(c, "ncclGroupStart", r);
    // a failing send / recv must not leave the group open (the next RCCL call on this thread would be queued into it):
    // remember the first error, stop posting, ALWAYS close the group
    const char *what = nullptr;
    ncclResult_t bad = ncclSuccess;
    auto post = [&](const char *name, ncclResult_t res) {
      if (res != ncclSuccess && !what) what = name, bad = res;
    };
    if (nRL && !what) post("ncclRecv", c->fRecv(recvL, nRL, ncclUint8, c->rank - 1, c->comm, stream));
    if (nRR && !what) post("ncclRecv", c->fRecv(recvR, nRR, ncclUint8, c->rank + 1, c->comm, stream));
    if (nSL && !what) post("ncclSend", c->fSend(sendL, nSL, ncclUint8, c->rank - 1, c->comm, stream));
    if (nSR && !what) post("ncclSend", c->fSend(sendR, nSR, ncclUint8, c->rank + 1, c->comm, stream));
    r = c->fGroupEnd();
    if (what) return comm_fail(c, what, bad);
    if (r != ncclSuccess) return comm_fail(c, "ncclGroupEnd", r);
    return PBF_OK;
  }
  if (!c->fn) {
    c->err = "communicator has no transport";
    return PBF_ERR_COMM;
  }
  // host-staged: device -> pinned host, caller moves the bytes, pinned host -> device
  const size_t need[4] = {nSL, nSR, nRL, nRR};
  for (int k = 0; k < 4; ++k)
    if (need[k] > c->hostCap[k]) {
      if (c->host[k]) (void)hipHostFree(c->host[k]);
      c->hostCap[k] = need[k] + need[k] / 2 + 4096;
      if (hipHostMalloc(&c->host[k], c->hostCap[k], hipHostMallocDefault) != hipSuccess) {
        c->err = "hipHostMalloc (exchange staging) failed";
        return PBF_ERR_HIP;
      }
    }
  if (nSL && hipMemcpyAsync(c->host[0], sendL, nSL, hipMemcpyDeviceToHost, stream) != hipSuccess) return PBF_ERR_HIP;
  if (nSR && hipMemcpyAsync(c->host[1], sendR, nSR, hipMemcpyDeviceToHost, stream) != hipSuccess) return PBF_ERR_HIP;
  if (hipStreamSynchronize(stream) != hipSuccess) return PBF_ERR_HIP;
  if (int rc = c->fn(c->user, c->host[0], nSL, c->host[1], nSR, c->host[2], nRL, c->host[3], nRR)) {
    c->err = "exchange callback failed (" + std::to_string(rc) + ")";
    return PBF_ERR_COMM;
  }
  if (nRL && hipMemcpyAsync(recvL, c->host[2], nRL, hipMemcpyHostToDevice, stream) != hipSuccess) return PBF_ERR_HIP;
  if (nRR && hipMemcpyAsync(recvR, c->host[3], nRR, hipMemcpyHostToDevice, stream) != hipSuccess) return PBF_ERR_HIP;
  return PBF_OK;
}

}  // namespace pbf
