// comm.hip -- the two all-gather transports of the row-sharded path (comm.hpp).
#include "comm.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: the functions are resolved with dlsym below

#include <cstring>
#include <string>

namespace oq {

namespace {

// ---------------------------------------------------------------------------------------------
// host-staged transport
// ---------------------------------------------------------------------------------------------
struct HostComm : Comm {
  host_allgather_fn fn;
  void *ctx;
  double *stage = nullptr;
  size_t stage_count = 0;
  HostComm(int r, int w, host_allgather_fn f, void *c) : fn(f), ctx(c) { rank = r; world = w; }
  ~HostComm() override { if (stage) (void)hipHostFree(stage); }
  const char *kind() const override { return "host"; }
  void all_gather(double *buf, size_t count, hipStream_t s) override {
    const size_t total = count * (size_t)world;
    if (total > stage_count) {
      if (stage) (void)hipHostFree(stage);
      HIP_CHECK(hipHostMalloc((void **)&stage, sizeof(double) * total));
      stage_count = total;
    }
    HIP_CHECK(hipMemcpyAsync(stage + (size_t)rank * count, buf + (size_t)rank * count, sizeof(double) * count,
                             hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    if (fn(ctx, stage, (long long)count) != 0) throw Error(6, "host all-gather callback failed");
    HIP_CHECK(hipMemcpyAsync(buf, stage, sizeof(double) * total, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));  // the stage is reused by the next exchange
    exchanges += 1; bytes += (double)(sizeof(double) * (total - count));
  }
};

// ---------------------------------------------------------------------------------------------
// RCCL transport
// ---------------------------------------------------------------------------------------------
struct RcclApi {
  void *handle = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllGather) all_gather = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
  decltype(&ncclCommCount) comm_count = nullptr;  // optional
};

RcclApi &rccl_api(const char *library_path) {
  static RcclApi api;
  if (api.handle) return api;
  void *h = nullptr;
  if (library_path && *library_path) h = dlopen(library_path, RTLD_NOW | RTLD_GLOBAL);
  // a copy the process already holds (e.g. the one PyTorch ships) wins over a second load
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) throw Error(6, "librccl not found (needed for the RCCL transport of the sharded path)");
  auto sym = [&](const char *name) {
    void *p = dlsym(h, name);
    if (!p) throw Error(6, std::string("librccl lacks symbol ") + name);
    return p;
  };
  api.get_unique_id = (decltype(api.get_unique_id))sym("ncclGetUniqueId");
  api.comm_init_rank = (decltype(api.comm_init_rank))sym("ncclCommInitRank");
  api.comm_destroy = (decltype(api.comm_destroy))sym("ncclCommDestroy");
  api.all_gather = (decltype(api.all_gather))sym("ncclAllGather");
  api.error_string = (decltype(api.error_string))sym("ncclGetErrorString");
  api.comm_count = (decltype(api.comm_count))dlsym(h, "ncclCommCount");
  api.handle = h;
  return api;
}

void rccl_check(RcclApi &api, ncclResult_t r, const char *what) {
  if (r != ncclSuccess) throw Error(6, std::string("RCCL: ") + what + ": " + api.error_string(r));
}

struct RcclComm : Comm {
  RcclApi &api;
  ncclComm_t comm = nullptr;
  RcclComm(int r, int w, const void *id128, const char *library_path) : api(rccl_api(library_path)) {
    rank = r; world = w;
    ncclUniqueId id;
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(&id, id128, sizeof(id));
    rccl_check(api, api.comm_init_rank(&comm, world, id, rank), "ncclCommInitRank");
  }
  ~RcclComm() override { if (comm) (void)api.comm_destroy(comm); }
  const char *kind() const override { return "rccl"; }
  int transport_ranks() const override {  // what the collective library itself says its communicator spans
    int cnt = -1;
    if (api.comm_count && api.comm_count(comm, &cnt) == ncclSuccess) return cnt;
    return -1;
  }
  void all_gather(double *buf, size_t count, hipStream_t s) override {
    // in-place form: the send buffer is this rank's chunk of the receive buffer
    rccl_check(api, api.all_gather(buf + (size_t)rank * count, buf, count, ncclDouble, comm, s), "ncclAllGather");
    exchanges += 1; bytes += (double)(sizeof(double) * count * (size_t)(world - 1));
  }
};

}  // namespace

Comm *make_host_comm(int rank, int world, host_allgather_fn fn, void *ctx) {
  if (world < 1 || rank < 0 || rank >= world || !fn) throw Error(1, "invalid communicator arguments");
  return new HostComm(rank, world, fn, ctx);
}

Comm *make_rccl_comm(int rank, int world, const void *unique_id, const char *library_path) {
  if (world < 1 || rank < 0 || rank >= world || !unique_id) throw Error(1, "invalid communicator arguments");
  return new RcclComm(rank, world, unique_id, library_path);
}

void rccl_unique_id(void *out128, const char *library_path) {
  RcclApi &api = rccl_api(library_path);
  ncclUniqueId id;
  rccl_check(api, api.get_unique_id(&id), "ncclGetUniqueId");
  memcpy(out128, &id, sizeof(id));
}

}  // namespace oq
