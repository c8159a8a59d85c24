// comm.hpp -- the one exchange primitive of the row-sharded PCG path (row N4 of SURVEY.md section 8f).
//
// A single large QP is cut into R row blocks (rank r owns rows [r*chunk, (r+1)*chunk) of A, of A' and of
// the full symmetric P, and the matching slices of every vector).  The only data that ever crosses ranks is
//   * the input vector of a sparse product (n or m doubles), all-gathered in place, and
//   * a handful of scalar slots (norms, dot products), all-gathered and combined in rank order,
// so one in-place all-gather of doubles is the whole interface.  Two implementations:
//   RcclComm -- ncclAllGather on the engine's stream (RCCL over xGMI; symbols resolved at run time from the
//               librccl the process already holds, so that libosqp_amd.so itself has no link dependency);
//   HostComm -- stages through pinned host memory and hands the buffer to a caller-supplied function
//               (MPI, gloo, ... ; also how the path is tested with several ranks on one GPU).
#pragma once
#include "common.hpp"

namespace oq {

struct Comm {
  int rank = 0, world = 1;
  double exchanges = 0, bytes = 0;  // statistics: number of all-gathers, bytes received per rank
  virtual ~Comm() {}
  // In place: on entry chunk `rank` (count doubles at buf + rank*count) is valid; on return all `world` chunks are.
  // Stream-ordered with respect to `s` on return (the host may or may not have blocked).
  virtual void all_gather(double *buf, size_t count, hipStream_t s) = 0;
  virtual const char *kind() const = 0;
  virtual int transport_ranks() const { return world; }  // ranks the transport itself reports (RCCL: ncclCommCount; -1: unknown)
};

// fn(ctx, host_buf, count): host_buf holds world*count doubles, chunk `rank` filled in; fill in the others.  Returns 0.
typedef int (*host_allgather_fn)(void *ctx, double *host_buf, long long count);

Comm *make_host_comm(int rank, int world, host_allgather_fn fn, void *ctx);
// unique_id: the 128 bytes of an ncclUniqueId created on one rank (rccl_unique_id) and distributed by the caller
Comm *make_rccl_comm(int rank, int world, const void *unique_id, const char *library_path);
void rccl_unique_id(void *out128, const char *library_path);

}  // namespace oq
