// engine.hip -- host control of the device-resident ADMM loop.
//
// Mirrors, step for step, the algorithm behind osqp_setup / osqp_solve /
// osqp_update_* [REF src/interface.jl:147, 171, 241-382, 476-709] as laid out
// in SURVEY.md Appendix A; all arithmetic runs in the kernels of kernels.hip,
// the host only sequences launches and reads back a handful of scalars every
// `check_termination` iterations.
#include "engine.hpp"
#include <csignal>
#include <mutex>

#include <algorithm>
#include <cmath>

namespace oq {

thread_local bool g_unsorted_columns = false;
static thread_local std::string g_last_error;
void set_last_error(const std::string &m) { g_last_error = m; }
const char *last_error_cstr() { return g_last_error.c_str(); }

#define RHO_MIN 1e-6
#define RHO_MAX 1e6
#define MIN_SCALING 1e-4
#define MAX_SCALING 1e4

void update_status(OSQPInfo *info, c_int status_val) {
  const char *s = "unsolved";
  info->status_val = status_val;
  switch (status_val) {
  case OSQP_SOLVED: s = "solved"; break;
  case OSQP_SOLVED_INACCURATE: s = "solved inaccurate"; break;
  case OSQP_PRIMAL_INFEASIBLE: s = "primal infeasible"; break;
  case OSQP_PRIMAL_INFEASIBLE_INACCURATE: s = "primal infeasible inaccurate"; break;
  case OSQP_DUAL_INFEASIBLE: s = "dual infeasible"; break;
  case OSQP_DUAL_INFEASIBLE_INACCURATE: s = "dual infeasible inaccurate"; break;
  case OSQP_MAX_ITER_REACHED: s = "maximum iterations reached"; break;
  case OSQP_TIME_LIMIT_REACHED: s = "run time limit reached"; break;
  case OSQP_SIGINT: s = "interrupted"; break;
  case OSQP_NON_CVX: s = "problem non convex"; break;
  default: break;
  }
  memset(info->status, 0, sizeof(info->status));
  strncpy(info->status, s, sizeof(info->status) - 1);
}

static void reset_info(OSQPInfo *info) {
  info->solve_time = 0.0;
  info->polish_time = 0.0;
  update_status(info, OSQP_UNSOLVED);
  info->rho_updates = 0;
}

static double limit_scaling(double v) {
  v = v < MIN_SCALING ? 1.0 : v;
  return v > MAX_SCALING ? MAX_SCALING : v;
}

Engine::Engine() {}
Engine::~Engine() {
  lin.reset();
  if (chunk_exec) (void)hipGraphExecDestroy(chunk_exec);
  if (h_slots) (void)hipHostFree(h_slots);
  if (ev_fork) (void)hipEventDestroy(ev_fork);
  if (ev_join) (void)hipEventDestroy(ev_join);
  if (aux_stream) (void)hipStreamDestroy(aux_stream);
  if (stream) (void)hipStreamDestroy(stream);
}

void Engine::fetch_slots(int first, int count, unsigned sum_mask) {
  combine_slots(first, count, sum_mask);
  read_slots(first, count);
}

// The host's view of `count` reduction slots: a one-wavefront kernel copies them into pinned host memory (mapped into the
// device's address space) and then raises a sequence number there; the host spins on the number.  No copy-engine
// transfer, no stream synchronisation: the round trip is the kernel launch plus a PCIe write (~5 us) instead of ~25 us,
// and the CG loop of a mid-size problem makes two of them per iteration.
__global__ void k_publish_slots(const double *__restrict__ src, double *__restrict__ host_dst, int count, volatile unsigned long long *host_seq,
                                unsigned long long seq) {
  if ((int)threadIdx.x < count) host_dst[threadIdx.x] = src[threadIdx.x];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) { *host_seq = seq; __threadfence_system(); }
}
Publish Engine::begin_publish() {
  Publish p;
  if (!h_slots_dev || g_debug_sync) return p;
  p.host_slots = h_slots_dev;
  p.host_seq = (volatile unsigned long long *)h_seq_dev;
  p.seq = ++publish_seq;
  return p;
}
void Engine::wait_publish(const Publish &p) {
  volatile unsigned long long *seen = (volatile unsigned long long *)h_seq;
  long long spins = 0;
  while (*seen < p.seq) {  // sequence numbers only rise; a later publish (a speculative launch behind this one) may already be in
    if (++spins > 20000000) {  // ~ a second: something is wrong with the stream rather than slow
      HIP_CHECK(hipStreamSynchronize(stream));
      if (*seen < p.seq) throw Error(6, "internal: published slots did not arrive");
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
}
void Engine::read_slots(int first, int count) {
  if (!h_slots_dev || g_debug_sync) {
    HIP_CHECK(hipMemcpyAsync(h_slots + first, slots.get() + first, sizeof(double) * count, hipMemcpyDeviceToHost, stream));
    sync();
    return;
  }
  const Publish p = begin_publish();
  OQ_LAUNCH(k_publish_slots, dim3(1), dim3(64), 0, stream, (const double *)(slots.get() + first), h_slots_dev + first, count, p.host_seq, p.seq);
  wait_publish(p);
}

// --------------------------------------------------------------------------
// row-sharded mode (row N4; comm.hpp)
// --------------------------------------------------------------------------
// the maximum of a host value over the ranks (decisions taken from a clock must be the same everywhere)
double Engine::agree_max(double v) {
  if (!comm) return v;
  HIP_CHECK(hipMemcpyAsync(slots.get() + S_T5, &v, sizeof(double), hipMemcpyHostToDevice, stream));
  fetch_slots(S_T5, 1);
  return h_slots[S_T5];
}

void Engine::combine_slots(int first, int count, unsigned sum_mask) {
  if (!comm) return;
  HIP_CHECK(hipMemcpyAsync(gslots.get() + (size_t)comm->rank * count, slots.get() + first, sizeof(double) * count,
                           hipMemcpyDeviceToDevice, stream));
  comm->all_gather(gslots.get(), (size_t)count, stream);
  combine_rank_slots(gslots.get(), comm->world, count, sum_mask, slots.get() + first, stream);
}

const double *Engine::full_n(const double *v) {
  if (!comm) return v;
  vec_copy(gn.get() + (size_t)comm->rank * chunk_n, v, n, stream);
  comm->all_gather(gn.get(), (size_t)chunk_n, stream);
  return gn.get();
}

const double *Engine::full_m(const double *v) {
  if (!comm) return v;
  vec_copy(gm.get() + (size_t)comm->rank * chunk_m, v, m, stream);
  comm->all_gather(gm.get(), (size_t)chunk_m, stream);
  return gm.get();
}

// The same exchange on a second stream, ordered behind everything enqueued on `stream` so far: what is enqueued on
// `stream` between full_m_begin and full_m_end runs while the m-vector travels (the P product of a CG iteration while
// t = rho (A p) is gathered for the A' product).  full_m_end makes `stream` wait for the gathered vector.
const double *Engine::full_m_begin(const double *v) {
  if (!comm) return v;
  if (!aux_stream) {
    HIP_CHECK(hipStreamCreateWithFlags(&aux_stream, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
  }
  HIP_CHECK(hipEventRecord(ev_fork, stream));
  HIP_CHECK(hipStreamWaitEvent(aux_stream, ev_fork, 0));
  vec_copy(gm.get() + (size_t)comm->rank * chunk_m, v, m, aux_stream);
  comm->all_gather(gm.get(), (size_t)chunk_m, aux_stream);
  HIP_CHECK(hipEventRecord(ev_join, aux_stream));
  return gm.get();
}
void Engine::full_m_end() {
  if (comm) HIP_CHECK(hipStreamWaitEvent(stream, ev_join, 0));
}

// The row partition of a sharded workspace: ceil-sized contiguous blocks (include/osqp_amd.h).  Returns the local ranges.
void Engine::shard_layout(int &n1, int &m1) {
  const int R = comm->world, r = comm->rank;
  chunk_n = (ng + R - 1) / R;
  chunk_m = mg > 0 ? (mg + R - 1) / R : 0;
  n0 = std::min(r * chunk_n, ng);
  m0 = std::min(r * chunk_m, mg);
  n1 = std::min(n0 + chunk_n, ng); m1 = std::min(m0 + chunk_m, mg);
  if (n1 <= n0 || (mg > 0 && m1 <= m0))
    throw Error(1, "sharded setup: with ceil-sized blocks the last rank would own no rows (e.g. n = 9 over 4 ranks: 3 + 3 + 3 + 0); use fewer ranks");
}
// the matching slices of q, l, u and the gather buffers; n, m become the local sizes
void Engine::shard_vectors(int n1, int m1, DevBuf<double> &q_, DevBuf<double> &l_, DevBuf<double> &u_) {
  const int R = comm->world;
  n = n1 - n0; m = m1 - m0;
  auto slice = [&](DevBuf<double> &b, int first, int count) {
    DevBuf<double> out((size_t)count);
    if (count) HIP_CHECK(hipMemcpyAsync(out.get(), b.get() + first, sizeof(double) * count, hipMemcpyDeviceToDevice, stream));
    sync();
    b = std::move(out);
  };
  slice(q_, n0, n); slice(l_, m0, m); slice(u_, m0, m);
  // value updates by nnz index and the direct back-end's symbolic phase are not available on a row block
  A_k2pos.release(); P_k2lo.release(); P_k2up.release(); Pp_keep.release(); Pi_keep.release();
  gn.alloc((size_t)chunk_n * R); gn.zero(stream);
  gm.alloc((size_t)std::max(chunk_m, 1) * R); gm.zero(stream);
  gslots.alloc((size_t)S_COUNT * R); gslots.zero(stream);
}
// Keep block `rank` of the rows of matrices that were built whole (setup_device with a communicator; the entry points
// go through setup_sharded, which never holds more than the block).
void Engine::shard_rows(DevBuf<double> &q_, DevBuf<double> &l_, DevBuf<double> &u_) {
  int n1 = 0, m1 = 0;
  shard_layout(n1, m1);
  csr_slice_rows(At, n0, n1, stream);
  csr_slice_rows(Pf, n0, n1, stream);
  if (mg > 0) csr_slice_rows(A, m0, m1, stream);
  shard_vectors(n1, m1, q_, l_, u_);
}

// ---- sharded setup from column ranges ------------------------------------------------------------------------------
// Stream compaction of the entries a rank keeps: a block counts its keepers (ballot + LDS), reserves a range of the
// output list with one atomic and writes in thread order.  fill = 0 only counts (first pass sizes the lists exactly).
__device__ __forceinline__ long long block_reserve(bool keep, unsigned long long *counter, bool fill) {
  __shared__ int wave_total[kBlock / 64];
  __shared__ long long base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long mask = __ballot(keep);
  const int before = __popcll(mask & ((1ull << lane) - 1ull));
  if (lane == 0) wave_total[wave] = __popcll(mask);
  __syncthreads();
  int offset = 0, total = 0;
  for (int w = 0; w < kBlock / 64; w++) { if (w < wave) offset += wave_total[w]; total += wave_total[w]; }
  if (threadIdx.x == 0) base = total ? (long long)atomicAdd(counter, (unsigned long long)total) : 0;
  __syncthreads();
  const long long pos = base + offset + before;
  __syncthreads();  // the shared words are reused by the caller's next reservation
  return fill ? pos : -1;
}
// entries of columns [j0, ...) of A: list 0 = rows [n0, n1) of A' (entry (j, i)), list 1 = rows [m0, m1) of A (entry (i, j))
__global__ __launch_bounds__(kBlock) void k_shard_pick_A(int64_t E, int j0, const int *__restrict__ colid, const int *__restrict__ Ai,
                                                        const double *__restrict__ Ax, int n0, int n1, int m0, int m1, int fill,
                                                        unsigned long long *__restrict__ counters, int *__restrict__ tr,
                                                        int *__restrict__ tc, double *__restrict__ tv, int *__restrict__ ar,
                                                        int *__restrict__ ac, double *__restrict__ av, int64_t base,
                                                        int *__restrict__ to, int *__restrict__ ao) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool in = e < E;
  const int j = in ? j0 + colid[e] : -1, i = in ? Ai[e] : -1;
  const double v = (in && fill) ? Ax[e] : 0.0;
  const bool kt = in && j >= n0 && j < n1, ka = in && i >= m0 && i < m1;
  long long p = block_reserve(kt, &counters[0], fill);
  if (fill && kt) { tr[p] = j - n0; tc[p] = i; tv[p] = v; if (to) to[p] = (int)(base + e); }  // base + e: the caller's nnz index
  p = block_reserve(ka, &counters[1], fill);
  if (fill && ka) { ar[p] = i - m0; ac[p] = j; av[p] = v; if (ao) ao[p] = (int)(base + e); }
}
// entries (i, j), i <= j, of columns [j0, ...) of triu(P): rows [n0, n1) of the full symmetric P get (j, i) and, off the
// diagonal, the mirror (i, j)
__global__ __launch_bounds__(kBlock) void k_shard_pick_P(int64_t E, int j0, const int *__restrict__ colid, const int *__restrict__ Pi,
                                                        const double *__restrict__ Px, int n0, int n1, int fill,
                                                        unsigned long long *__restrict__ counter, int *__restrict__ pr,
                                                        int *__restrict__ pc, double *__restrict__ pv, int *__restrict__ bad, int64_t base,
                                                        int *__restrict__ po) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool in = e < E;
  const int j = in ? j0 + colid[e] : -1, i = in ? Pi[e] : -1;
  const double v = (in && fill) ? Px[e] : 0.0;
  if (in && i > j) *bad = 1;  // not upper triangular
  const bool kl = in && j >= n0 && j < n1, ku = in && i != j && i >= n0 && i < n1;
  long long p = block_reserve(kl, counter, fill);
  if (fill && kl) { pr[p] = j - n0; pc[p] = i; pv[p] = v; if (po) po[p] = (int)(base + e); }
  p = block_reserve(ku, counter, fill);
  if (fill && ku) { pr[p] = i - n0; pc[p] = j; pv[p] = v; if (po) po[p] = (int)(base + e); }
}

// flag = 1 when two neighbouring entries of a (sorted) row share their column
__global__ __launch_bounds__(kBlock) void k_csr_repeats(int64_t nnz, int rows, const int64_t *__restrict__ rowptr, const int *__restrict__ col,
                                                        int *__restrict__ flag) {
  const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k < 1 || k >= nnz || col[k] != col[k - 1]) return;
  int lo = 0, hi = rows;  // same column: a repeat unless k opens a row
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (rowptr[mid] <= k) lo = mid; else hi = mid; }
  if (rowptr[lo] != k) *flag = 1;
}
__global__ __launch_bounds__(kBlock) void k_gather_ints(int64_t n, const int *__restrict__ src, const int *__restrict__ in, int *__restrict__ out) {
  const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k < n) out[k] = in[src[k]];
}

void Engine::setup_sharded(ColumnSource &src, const OSQPSettings &s) {
  n = src.n; m = src.m; ng = src.n; mg = src.m; st = s;
  nnzPtriu = src.nnzP; nnzA = src.nnzA;
  open_device();
  int n1 = 0, m1 = 0;
  shard_layout(n1, m1);
  // column ranges of about a quarter of a rank's share, at least 64 columns
  const int step = std::max(64, (ng + 4 * comm->world - 1) / (4 * comm->world));
  DevBuf<unsigned long long> counters(3);
  counters.zero(stream);
  DevBuf<int> tr, tc, ar, ac, pr, pc, to, ao, po;
  DevBuf<double> tv, av, pv;
  unsigned long long total[3] = {0, 0, 0};
  // The caller's nnz index of every kept entry (what osqp_update_P / _A pick their new values by): 4 B per stored entry on
  // the path whose purpose is memory headroom, so OSQP_AMD_SHARD_UPDATES=0 leaves them out (value updates are then refused
  // with exit flag 6), as does a problem whose nnz indices do not fit 32 bits; a block that goes compact releases its map.
  const bool keep_org = src.nnzA < 2147483647LL && src.nnzP < 2147483647LL &&
                        !(getenv("OSQP_AMD_SHARD_UPDATES") && atoi(getenv("OSQP_AMD_SHARD_UPDATES")) == 0);
  for (int fill = 0; fill < 2; fill++) {
    if (fill) {
      counters.download(total, 3, stream);
      sync();
      if (total[0] >= 2147483647ULL || total[1] >= 2147483647ULL || total[2] >= 2147483647ULL)
        throw Error(6, "matrix block too large: more than 2^31-1 non-zeros on one rank");
      tr.alloc(total[0]); tc.alloc(total[0]); tv.alloc(total[0]);
      ar.alloc(total[1]); ac.alloc(total[1]); av.alloc(total[1]);
      pr.alloc(total[2]); pc.alloc(total[2]); pv.alloc(total[2]);
      if (keep_org) { to.alloc(total[0]); ao.alloc(total[1]); po.alloc(total[2]); }
      counters.zero(stream);
    }
    int64_t baseA = 0, baseP = 0;  // the caller's nnz index of the first entry of the column range (ranges come in column order)
    for (int j0 = 0; j0 < ng; j0 += step) {
      const int j1 = std::min(ng, j0 + step);
      DevBuf<int64_t> p;
      DevBuf<int> idx;
      DevBuf<double> val;
      {
        const int64_t E = src.A_chunk(j0, j1, p, idx, val, stream);
        if (E > 0) {
          DevBuf<int> colid((size_t)E);
          expand_colptr(j1 - j0, p.get(), E, colid.get(), stream);
          OQ_LAUNCH(k_shard_pick_A, dim3(blocks_for(E)), dim3(kBlock), 0, stream, E, j0, colid.get(), idx.get(), val.get(), n0, n1, m0, m1, fill,
                    counters.get(), tr.get(), tc.get(), tv.get(), ar.get(), ac.get(), av.get(), baseA, to.get(), ao.get());
          sync();
          baseA += E;
        }
      }
      {
        const int64_t E = src.P_chunk(j0, j1, p, idx, val, stream);
        if (E > 0) {
          DevBuf<int> colid((size_t)E);
          expand_colptr(j1 - j0, p.get(), E, colid.get(), stream);
          OQ_LAUNCH(k_shard_pick_P, dim3(blocks_for(E)), dim3(kBlock), 0, stream, E, j0, colid.get(), idx.get(), val.get(), n0, n1, fill,
                    counters.get() + 2, pr.get(), pc.get(), pv.get(), flag.get(), baseP, po.get());
          sync();
          baseP += E;
        }
      }
    }
  }
  int bad = 0;
  flag.download(&bad, 1, stream);
  sync();
  if (bad) throw Error(1, "P is not upper triangular");
  auto build = [&](DevCsr &M, int rows, int cols, unsigned long long E, DevBuf<int> &er, DevBuf<int> &ec, DevBuf<double> &ev, DevBuf<int> &eo,
                   DevBuf<int> &org) {
    DevBuf<int> order;
    csr_from_coo(rows, cols, (int64_t)E, er.get(), ec.get(), M, order, stream);
    if (M.nnz > 1) {  // the same (row, column) twice: refused, as on one device (the sort accepts any order of the caller's entries)
      flag.zero(stream);
      OQ_LAUNCH(k_csr_repeats, dim3(blocks_for(M.nnz)), dim3(kBlock), 0, stream, M.nnz, rows, M.rowptr.get(), M.col.get(), flag.get());
      int rep = 0;
      flag.download(&rep, 1, stream);
      sync();
      flag.zero(stream);
      if (rep) throw Error(1, "a matrix of the problem holds the same entry twice");
    }
    gather_values(M.nnz, order.get(), ev.get(), M.val.get(), 0, stream);
    if (keep_org) org.alloc(std::max<size_t>(1, (size_t)M.nnz));
    if (keep_org && M.nnz > 0) OQ_LAUNCH(k_gather_ints, dim3(blocks_for(M.nnz)), dim3(kBlock), 0, stream, (int64_t)M.nnz, order.get(), eo.get(), org.get());
    sync();
    er.release(); ec.release(); ev.release(); eo.release();
  };
  build(At, n1 - n0, mg, total[0], tr, tc, tv, to, At_org);
  build(A, m1 - m0, ng, total[1], ar, ac, av, ao, A_org);
  build(Pf, n1 - n0, ng, total[2], pr, pc, pv, po, Pf_org);
  DevBuf<double> q_, l_, u_;
  src.vectors(q_, l_, u_, stream);
  sync();
  shard_vectors(n1, m1, q_, l_, u_);
  finish_setup(q_, l_, u_);
}

// the host caller's CSC arrays, uploaded range by range
struct HostColumns : ColumnSource {
  const OSQPData *d;
  explicit HostColumns(const OSQPData *data) : d(data) {
    n = (int)d->n; m = (int)d->m; nnzP = d->P->p[n]; nnzA = d->A->p[n];
    if (nnzA >= 2147483647LL || 2 * nnzP >= 2147483647LL) throw Error(6, "matrix too large: more than 2^31-1 non-zeros");
    for (int64_t k = 0; k < nnzA; k++) if (d->A->i[k] < 0 || d->A->i[k] >= m) throw Error(1, "row index of A out of range");
  }
  static int64_t chunk(const csc *M, int j0, int j1, DevBuf<int64_t> &p, DevBuf<int> &i, DevBuf<double> &x, hipStream_t s) {
    const int64_t b = M->p[j0], e = M->p[j1];
    std::vector<int64_t> hp((size_t)(j1 - j0) + 1);
    for (int j = j0; j <= j1; j++) hp[j - j0] = M->p[j] - b;
    std::vector<int> hi((size_t)(e - b));
    for (int64_t k = b; k < e; k++) hi[k - b] = (int)M->i[k];
    p.alloc(hp.size()); i.alloc(std::max<size_t>(1, hi.size())); x.alloc(std::max<size_t>(1, hi.size()));
    p.upload(hp.data(), hp.size(), s); i.upload(hi.data(), hi.size(), s); x.upload(M->x + b, (size_t)(e - b), s);
    HIP_CHECK(hipStreamSynchronize(s));
    return e - b;
  }
  int64_t P_chunk(int j0, int j1, DevBuf<int64_t> &p, DevBuf<int> &i, DevBuf<double> &x, hipStream_t s) override { return chunk(d->P, j0, j1, p, i, x, s); }
  int64_t A_chunk(int j0, int j1, DevBuf<int64_t> &p, DevBuf<int> &i, DevBuf<double> &x, hipStream_t s) override { return chunk(d->A, j0, j1, p, i, x, s); }
  void vectors(DevBuf<double> &q, DevBuf<double> &l, DevBuf<double> &u, hipStream_t s) override {
    q.alloc(n); l.alloc(m); u.alloc(m);
    q.upload(d->q, n, s); l.upload(d->l, m, s); u.upload(d->u, m, s);
    HIP_CHECK(hipStreamSynchronize(s));
  }
};
std::unique_ptr<ColumnSource> host_columns(const OSQPData *data) { return std::unique_ptr<ColumnSource>(new HostColumns(data)); }

// --------------------------------------------------------------------------
// setup
// --------------------------------------------------------------------------
// coordinate list of the full symmetric P from its upper triangle (CSC): entry k = (i, j), i <= j
//   e = k          -> (row j, col i)   (the CSC arrays read as CSR of the lower triangle)
//   e = nnz + k    -> (row i, col j)   (mirror; dropped on the diagonal)
__global__ __launch_bounds__(kBlock) void k_sym_coo(int64_t nnz, const int *__restrict__ Pi, const int *__restrict__ colid,
                                                    int *__restrict__ erow, int *__restrict__ ecol, int *__restrict__ bad) {
  int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= nnz) return;
  int i = Pi[k], j = colid[k];
  if (i > j) *bad = 1;  // not upper triangular
  erow[k] = j; ecol[k] = i;
  erow[nnz + k] = (i == j) ? -1 : i;
  ecol[nnz + k] = j;
}
__global__ __launch_bounds__(kBlock) void k_fill_int(int64_t n, int *p, int v) {
  int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k < n) p[k] = v;
}

// OSQP_AMD_SETUP_TRACE=1: wall time and device bytes after every phase of the setup, on stderr (experiments only; each
// mark synchronises the stream)
void Engine::setup_mark(const char *what) {
  static const bool on = getenv("OSQP_AMD_SETUP_TRACE") && atoi(getenv("OSQP_AMD_SETUP_TRACE")) == 1;
  if (!on) return;
  sync();
  const double t = toc();
  static double alloc_prev = 0., free_prev = 0.;
  fprintf(stderr, "[setup] %-28s %8.1f ms (+%7.1f; hipMalloc %6.1f, hipFree %6.1f)  device %6.2f GB + %5.2f parked  peak %6.2f GB\n", what,
          1e3 * t, 1e3 * (t - mark_prev), 1e3 * (g_alloc_s - alloc_prev), 1e3 * (g_free_s - free_prev), g_device_bytes / 1e9,
          g_cache_bytes / 1e9, g_device_peak / 1e9);
  mark_prev = t; alloc_prev = g_alloc_s; free_prev = g_free_s;
}

void Engine::open_device() {
  HIP_CHECK(hipGetDevice(&device));
  HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  HIP_CHECK(hipHostMalloc((void **)&h_slots, sizeof(double) * (S_COUNT + 8)));
  h_seq = (unsigned long long *)(h_slots + S_COUNT);
  *h_seq = 0;
  {
    void *dp = nullptr;
    static const bool mapped = !(getenv("OSQP_AMD_MAPPED_SLOTS") && atoi(getenv("OSQP_AMD_MAPPED_SLOTS")) == 0);
    if (mapped && hipHostGetDevicePointer(&dp, h_slots, 0) == hipSuccess && dp) {
      h_slots_dev = (double *)dp;
      h_seq_dev = (unsigned long long *)(h_slots_dev + S_COUNT);
    }
  }
  slots.alloc(S_COUNT); slots.zero(stream);
  partials.alloc(16 * kReduceBlocks);
  flag.alloc(4); flag.zero(stream);
}

// the caller's CSC arrays of A serve as the CSR arrays of A' as they come: the row indices of every column have to ascend (what
// Julia's SparseMatrixCSC guarantees [REF src/types.jl:21-47 copies it verbatim]); the panel layout bisects them
__global__ __launch_bounds__(kBlock) void k_csr_cols_descend(int64_t nnz, int rows, const int64_t *__restrict__ rp, const int *__restrict__ col,
                                                             int *__restrict__ flag) {
  const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k < 1 || k >= nnz) return;
  if (col[k - 1] < col[k]) return;
  // not ascending across k-1 -> k: fine only if k starts a row
  int lo = 0, hi = rows;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (rp[mid] <= k) lo = mid; else hi = mid; }
  if (rp[lo] != k) *flag = 1;
}

void Engine::setup_device(int n_, int m_, DevBuf<int64_t> &Pp, DevBuf<int> &Pi, DevBuf<double> &Px, DevBuf<int64_t> &Ap,
                          DevBuf<int> &Ai, DevBuf<double> &Ax_in, DevBuf<double> &q_, DevBuf<double> &l_, DevBuf<double> &u_,
                          const OSQPSettings &s) {
  n = n_; m = m_; ng = n_; mg = m_; st = s;
  open_device();

  int64_t ends[2] = {0, 0};
  HIP_CHECK(hipMemcpyAsync(&ends[0], Pp.get() + n, sizeof(int64_t), hipMemcpyDeviceToHost, stream));
  HIP_CHECK(hipMemcpyAsync(&ends[1], Ap.get() + n, sizeof(int64_t), hipMemcpyDeviceToHost, stream));
  sync();
  nnzPtriu = ends[0]; nnzA = ends[1];
  if (nnzA >= 2147483647LL || 2 * nnzPtriu >= 2147483647LL) throw Error(6, "matrix too large: more than 2^31-1 non-zeros");

  // ---- A' is the caller's CSC as it comes; A (CSR) is its transpose ----
  At.rows = n; At.cols = m; At.nnz = nnzA;
  At.rowptr = std::move(Ap); At.col = std::move(Ai); At.val = std::move(Ax_in);
  At.group = pick_group(n, nnzA);
  if (nnzA > 1) {
    flag.zero(stream);
    OQ_LAUNCH(k_csr_cols_descend, dim3(blocks_for(nnzA)), dim3(kBlock), 0, stream, nnzA, n, At.rowptr.get(), At.col.get(), flag.get());
    int bad = 0;
    flag.download(&bad, 1, stream);
    sync();
    flag.zero(stream);
    if (bad) { g_unsorted_columns = true; throw UnsortedColumns(); }
  }
  // A problem that is certain to run the indirect back-end at a size where the workspace goes compact: every matrix gets
  // its sliced-ELL copy and gives up its CSR arrays NOW, one after the other (Ruiz scaling then runs over the slices:
  // scale_data), instead of all three copies living side by side until the end of the setup.
  const bool early = pcg_certain() && !comm && compact_wanted(nnzA + std::max<int64_t>(0, 2 * nnzPtriu - n));
  auto early_compact = [&](int which) {
    DevCsr &M = which == 0 ? A : (which == 1 ? At : Pf);
    if (!early || M.rows == 0 || M.nnz == 0 || !panel_wanted(M)) return;
    DevBuf<uint32_t> p2s;
    if (!comm) p2s.alloc((size_t)M.nnz);  // the slot of every entry, recorded while the slices are filled
    bool maps_done = false;
    // as soon as the column pass of the fill has recorded every slot the nnz-index maps are rewritten and the 4 B per entry of
    // the slot array go, BEFORE the slice values are allocated (the high-water mark of the setup is in here)
    auto fold = [&]() { if (!comm) { fold_slot_maps(which, p2s); maps_done = true; } };
    try { panel_build(M, stream, p2s.get(), true, fold); }
    catch (const PanelRefused &) { M.panel = DevPanel(); return; }  // declined before anything was released: stays on its CSR arrays
    compact_one(which, &p2s, maps_done);
  };
  auto transpose_A = [&]() {
    {
      DevBuf<int> colid((size_t)nnzA), src;
      expand_colptr(n, At.rowptr.get(), nnzA, colid.get(), stream);
      csr_from_coo(m, n, nnzA, At.col.get(), colid.get(), A, src, stream);
      sync();
      colid.release();
      gather_values(A.nnz, src.get(), At.val.get(), A.val.get(), 0, stream);
      A_k2pos.alloc((size_t)nnzA);
      invert_map(A.nnz, src.get(), 0, nnzA, A_k2pos.get(), stream);
      sync();
    }
  };
  auto symmetric_P = [&]() {  // full symmetric P from the upper triangle
    {
      DevBuf<int> colid((size_t)nnzPtriu), erow((size_t)(2 * nnzPtriu)), ecol((size_t)(2 * nnzPtriu)), src;
      expand_colptr(n, Pp.get(), nnzPtriu, colid.get(), stream);
      if (nnzPtriu > 0)
        OQ_LAUNCH(k_sym_coo, dim3(blocks_for(nnzPtriu)), dim3(kBlock), 0, stream, nnzPtriu, Pi.get(), colid.get(),
                           erow.get(), ecol.get(), flag.get());
      int bad = 0;
      flag.download(&bad, 1, stream);
      sync();
      colid.release();  // every temporary goes as soon as it has been read: this block is the high-water mark of a large setup
      if (bad) throw Error(1, "P is not upper triangular");
      csr_from_coo(n, n, 2 * nnzPtriu, erow.get(), ecol.get(), Pf, src, stream);
      sync();
      erow.release(); ecol.release();
      gather_values(Pf.nnz, src.get(), Px.get(), Pf.val.get(), nnzPtriu, stream);
      sync();
      Px.release();
      P_k2lo.alloc((size_t)nnzPtriu); P_k2up.alloc((size_t)nnzPtriu);
      if (nnzPtriu > 0) {
        OQ_LAUNCH(k_fill_int, dim3(blocks_for(nnzPtriu)), dim3(kBlock), 0, stream, nnzPtriu, P_k2up.get(), -1);
        invert_map(Pf.nnz, src.get(), 0, nnzPtriu, P_k2lo.get(), stream);
        invert_map(Pf.nnz, src.get(), nnzPtriu, 2 * nnzPtriu, P_k2up.get(), stream);
      }
      sync();
    }
  };
  // Early compaction: P first -- while it is built and sliced only the caller's arrays are there, and the high-water mark of
  // the setup moves from "slices of P on top of both compact copies of A" to "slices of A' next to compact P and A".
  if (early) {
    symmetric_P();
    Pp_keep = std::move(Pp); Pi_keep = std::move(Pi);
    Px.release();
    setup_mark("full symmetric P");
    early_compact(2);
    setup_mark("slices of P");
    transpose_A();
    setup_mark("A = transpose(A')");
    if (m > 0) { early_compact(0); setup_mark("slices of A"); early_compact(1); setup_mark("slices of A'"); }
  } else {
    transpose_A();
    setup_mark("A = transpose(A')");
    symmetric_P();
    Pp_keep = std::move(Pp); Pi_keep = std::move(Pi);
    Px.release();
    setup_mark("full symmetric P");
  }

  if (comm) shard_rows(q_, l_, u_);  // from here on n, m are the local sizes
  finish_setup(q_, l_, u_);
}

void Engine::finish_setup(DevBuf<double> &q_, DevBuf<double> &l_, DevBuf<double> &u_) {
  q = std::move(q_); l = std::move(l_); u = std::move(u_);
  auto alloc0 = [&](DevBuf<double> &b, size_t cnt) { b.alloc(cnt); b.zero(stream); };
  alloc0(D, n); alloc0(Dinv, n); alloc0(E, m); alloc0(Einv, m); alloc0(rho, m); alloc0(rho_inv, m);
  ctype.alloc(m); ctype.zero(stream);
  alloc0(x, n); alloc0(z, m); alloc0(y, m); alloc0(xz, (size_t)n + m);
  alloc0(dx, n); alloc0(dy, m); alloc0(Ax, m); alloc0(Px_, n); alloc0(Aty, n);
  alloc0(tn, n); alloc0(tm, m); alloc0(tn2, n); alloc0(tm2, m);
  vec_set(D.get(), 1.0, n, stream); vec_set(Dinv.get(), 1.0, n, stream);
  vec_set(E.get(), 1.0, m, stream); vec_set(Einv.get(), 1.0, m, stream);
  c = 1.0; cinv = 1.0;
  // host copies of the unscaled bounds (validation of single-sided bound updates)
  h_l.resize(m); h_u.resize(m);
  l.download(h_l.data(), m, stream); u.download(h_u.data(), m, stream);
  sync();
  for (int i = 0; i < m; i++) if (h_l[i] > h_u[i]) throw Error(1, "lower bound greater than upper bound");

  setup_mark("vectors");
  if (st.scaling) scale_data();
  setup_mark("Ruiz scaling");
  refresh_panels();
  setup_mark("slices (late)");
  set_rho_vec();
  h_x.assign(ng, 0.0); h_y.assign(mg, 0.0); h_dx.assign(ng, 0.0); h_dy.assign(mg, 0.0);
  lambda0 = 0.015;
  if (const char *e = getenv("OSQP_AMD_PCG_LAMBDA")) lambda0 = atof(e);
  lambda = lambda0;
  select_linsys();
  setup_mark("back-end");
  compact_matrices();
  setup_mark("compaction (late)");
  sync();
  // The setup banner of a verbose run (the reference's default is verbose = true, and libosqp prints its settings there with
  // "linear system solver = qdldl"): what a drop-in caller otherwise only finds in osqp_amd_get_stats -- WHICH back-end the
  // default setting resolved to, and in which form the factor is.
  if (st.verbose && rank() == 0) {
    const char *asked = st.linsys_solver == 0 ? "qdldl" : (st.linsys_solver == 1 ? "mkl pardiso" : (st.linsys_solver == AMD_PCG_SOLVER ? "amd pcg" : "amd direct"));
    printf("-----------------------------------------------------------------\n");
    printf("  OSQP ADMM engine for AMD MI355X (libosqp v0.6.2 interface)\n");
    printf("-----------------------------------------------------------------\n");
    printf("problem:  variables n = %d, constraints m = %d\n          nnz(P) + nnz(A) = %lld\n", n, m, (long long)(nnzPtriu + nnzA));
    if (lin->kind() == 0) {
      printf("settings: linear system solver = %s -> direct (LDL' on the device)\n", asked);
      printf("          nnz(L) = %.0f, %d levels", lin->nnzL(), (int)lin->levels());
      if (lin->supernode_levels() > 0) printf(" (%d supernode levels%s)", (int)lin->supernode_levels(), lin->multifrontal() > 0 ? ", multifrontal factorisation" : "");
      if (lin->dense_block() > 0) printf(", dense top block of %d pivots", (int)lin->dense_block());
      printf("\n");
    } else
      printf("settings: linear system solver = %s -> indirect (preconditioned CG%s)\n", asked,
             st.linsys_solver == AMD_PCG_SOLVER ? "" : ": the factor of this problem would not fit / its KKT matrix is beyond the direct back-end's limit");
    printf("          eps_abs = %.1e, eps_rel = %.1e, rho = %.2e%s, sigma = %.2e, alpha = %.2f, max_iter = %lld\n", st.eps_abs, st.eps_rel, st.rho,
           st.adaptive_rho ? " (adaptive)" : "", st.sigma, st.alpha, (long long)st.max_iter);
    printf("          scaling: %s, polish: %s, warm start: %s\n\n", st.scaling ? "on" : "off", st.polish ? "on" : "off", st.warm_start ? "on" : "off");
  }
}

// 64-bit indices of the caller, as they arrive in a staging buffer, to the 32-bit arrays of the engine; an index outside
// [0, limit) raises the flag (the host loop this replaces checked A only)
__global__ __launch_bounds__(kBlock) void k_narrow_indices(int64_t cnt, const long long *__restrict__ in, int *__restrict__ out, long long limit,
                                                           int *__restrict__ flag) {
  const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= cnt) return;
  const long long v = in[k];
  if (v < 0 || v >= limit) { *flag = 1; out[k] = 0; return; }
  out[k] = (int)v;
}

void Engine::setup_host(const OSQPData *d, const OSQPSettings &s) {
  const int n_ = (int)d->n, m_ = (int)d->m;
  hipStream_t s0 = nullptr;  // uploads on the null stream, synchronous
  const int64_t nzP = d->P->p[n_], nzA = d->A->p[n_];
  if (nzA >= 2147483647LL || 2 * nzP >= 2147483647LL) throw Error(6, "matrix too large: more than 2^31-1 non-zeros");
  // A problem that is certain to run the indirect back-end never looks at the host copies of the patterns (they feed the
  // direct back-end's symbolic phase): at rand-1e6 they are 6 GB of host memory and 1.5e9 single-threaded conversions.  Its
  // 64-bit index arrays go to the device as they are, in pieces, and are narrowed (and range-checked) there.
  const double nnzK = (double)nzP + (double)nzA + (double)n_ + (double)m_;
  const bool indirect_for_sure = s.linsys_solver == AMD_PCG_SOLVER || (s.linsys_solver != AMD_DIRECT_SOLVER && nnzK > 4e7);
  DevBuf<int64_t> Pp((size_t)n_ + 1), Ap((size_t)n_ + 1);
  DevBuf<int> Pi((size_t)nzP), Ai((size_t)nzA);
  DevBuf<double> Px((size_t)nzP), Ax_((size_t)nzA), q_((size_t)n_), l_((size_t)m_), u_((size_t)m_);
  DevBuf<int> range_flag(1);
  range_flag.zero(s0);
  if (indirect_for_sure) {
    have_host_pattern = false;
    const int64_t piece = (int64_t)1 << 25;  // 256 MB of indices per copy
    DevBuf<long long> stage((size_t)std::min<int64_t>(piece, std::max<int64_t>(1, std::max(nzP, nzA))));
    auto narrow = [&](const c_int *src, int64_t cnt, int *dst, long long limit) {
      for (int64_t o = 0; o < cnt; o += piece) {
        const int64_t c = std::min(piece, cnt - o);
        HIP_CHECK(hipMemcpyAsync(stage.get(), src + o, sizeof(long long) * (size_t)c, hipMemcpyHostToDevice, s0));
        OQ_LAUNCH(k_narrow_indices, dim3(blocks_for(c)), dim3(kBlock), 0, s0, c, (const long long *)stage.get(), dst + o, limit, range_flag.get());
      }
    };
    narrow(d->P->i, nzP, Pi.get(), (long long)n_);
    narrow(d->A->i, nzA, Ai.get(), (long long)m_);
    Pp.upload((const int64_t *)d->P->p, (size_t)n_ + 1, s0); Ap.upload((const int64_t *)d->A->p, (size_t)n_ + 1, s0);
  } else {
    // host patterns (32-bit row indices) are kept for the direct back-end
    hP.rows = n_; hP.cols = n_; hP.p.assign(d->P->p, d->P->p + n_ + 1); hP.i.resize(nzP);
    for (int64_t k = 0; k < nzP; k++) hP.i[k] = (int)d->P->i[k];
    hA.rows = m_; hA.cols = n_; hA.p.assign(d->A->p, d->A->p + n_ + 1); hA.i.resize(nzA);
    for (int64_t k = 0; k < nzA; k++) hA.i[k] = (int)d->A->i[k];
    for (int64_t k = 0; k < nzA; k++) if (hA.i[k] < 0 || hA.i[k] >= m_) throw Error(1, "row index of A out of range");
    for (int64_t k = 0; k < nzP; k++) if (hP.i[k] < 0 || hP.i[k] >= n_) throw Error(1, "row index of P out of range");
    have_host_pattern = true;
    Pp.upload((const int64_t *)hP.p.data(), (size_t)n_ + 1, s0); Ap.upload((const int64_t *)hA.p.data(), (size_t)n_ + 1, s0);
    Pi.upload(hP.i.data(), nzP, s0); Ai.upload(hA.i.data(), nzA, s0);
  }
  Px.upload(d->P->x, nzP, s0); Ax_.upload(d->A->x, nzA, s0);
  q_.upload(d->q, n_, s0); l_.upload(d->l, m_, s0); u_.upload(d->u, m_, s0);
  int bad = 0;
  range_flag.download(&bad, 1, s0);
  HIP_CHECK(hipDeviceSynchronize());
  if (bad) throw Error(1, "row index of P or A out of range");
  setup_device(n_, m_, Pp, Pi, Px, Ap, Ai, Ax_, q_, l_, u_, s);
}

void Engine::fetch_host_pattern() {
  if (have_host_pattern) return;
  hP.rows = n; hP.cols = n; hP.p.resize((size_t)n + 1); hP.i.resize(nnzPtriu);
  Pp_keep.download(hP.p.data(), (size_t)n + 1, stream);
  Pi_keep.download(hP.i.data(), nnzPtriu, stream);
  hA.rows = m; hA.cols = n; hA.p.resize((size_t)n + 1); hA.i.resize(nnzA);
  At.rowptr.download(hA.p.data(), (size_t)n + 1, stream);
  At.col.download(hA.i.data(), nnzA, stream);
  sync();
  have_host_pattern = true;
}

// --------------------------------------------------------------------------
// K0: Ruiz equilibration + cost scaling
// --------------------------------------------------------------------------
// Every iteration touches each matrix ONCE: the scaling pass leaves the row norms of its result behind (what the next
// iteration starts from), and the cost scaling of P is not a pass of its own -- the scalar waits in c_pend and goes in
// first in the next pass, ((v c) D_lo) D_hi, which is bit for bit what scaling by c in place and by D afterwards gives;
// the column norms of c P are fl(c * norm) (x -> fl(c x) is monotone, so it commutes with the maximum).  Same values
// as the plain statement (oracle/osqp_oracle.c scale_data) with a third of its passes; on compact matrices the passes
// run over the sliced-ELL copies with the column factors staged through LDS (panel.hip, k_sell_scale_norm).
void Engine::scale_data() {
  c = 1.0;
  vec_set(D.get(), 1.0, n, stream); vec_set(E.get(), 1.0, m, stream);
  double *Dt = tn.get(), *Et = tm.get();
  double *nP = tn2.get(), *nAc = Px_.get(), *nAr = tm2.get();  // column norms of P, of A, row norms of A as they stand
  csr_row_absmax(Pf, nP, false, stream);             // ||P[:,j]||inf (P symmetric: row = column)
  if (m > 0) {
    csr_row_absmax(At, nAc, false, stream);          // ||A[:,j]||inf
    csr_row_absmax(A, nAr, false, stream);           // ||A[i,:]||inf
  }
  double c_pend = 1.0;  // cost scaling not yet applied to the stored values of P
  for (int it = 0; it < st.scaling; it++) {
    vec_ruiz_factor(Dt, c_pend, nP, m > 0 ? nAc : nullptr, n, stream);
    vec_ruiz_factor(Et, 1.0, nAr, nullptr, m, stream);
    const double *Dg = full_n(Dt);  // column scalings are indexed by global ids
    csr_scale_rows_cols(Pf, Dt, Dg, 1, 1.0, stream, n0, c_pend, nP);
    if (m > 0) {
      csr_scale_rows_cols(A, Et, Dg, 0, 1.0, stream, 0, 1.0, nAr);
      csr_scale_rows_cols(At, Dt, full_m(Et), 2, 1.0, stream, 0, 1.0, nAc);
    }
    vec_ew_prod(q.get(), q.get(), Dt, n, stream);
    vec_ew_prod(D.get(), D.get(), Dt, n, stream);
    vec_ew_prod(E.get(), E.get(), Et, m, stream);
    // cost scaling: mean column norm of D P D (nP, just computed) and ||q||inf
    HIP_CHECK(hipMemsetAsync(slots.get() + S_T0, 0, sizeof(double) * 2, stream));
    reduce_sum(nP, n, partials.get(), slots.get() + S_T0, stream);
    reduce_absmax(q.get(), nullptr, n, slots.get() + S_T1, stream);
    fetch_slots(S_T0, 2, 1u);
    double c_temp = h_slots[S_T0] / (double)ng;
    double qn = limit_scaling(h_slots[S_T1]);
    c_temp = limit_scaling(std::max(c_temp, qn));
    c_temp = 1.0 / c_temp;
    c_pend = c_temp;
    vec_scale(q.get(), c_temp, n, stream);
    c *= c_temp;
  }
  if (c_pend != 1.0) csr_scale_rows_cols(Pf, nullptr, nullptr, 0, c_pend, stream);
  cinv = 1.0 / c;
  vec_ew_recip(Dinv.get(), D.get(), n, stream);
  vec_ew_recip(Einv.get(), E.get(), m, stream);
  vec_ew_prod(l.get(), l.get(), E.get(), m, stream);
  vec_ew_prod(u.get(), u.get(), E.get(), m, stream);
}

// LDS-staged panel copies of the matrices whose x vector does not fit the caches (panel.hip)
void Engine::refresh_panels() {
  for (DevCsr *M : {&A, &At, &Pf}) {
    if (M->rows == 0 || M->nnz == 0 || M->compact) continue;  // compact: the values were changed in place
    if (M->panel.active) panel_fill(*M, false, stream);
    else if (panel_wanted(*M)) {
      // the sliced-ELL copy is an optimisation: a layout it cannot express (more than 2^32 padded entries, ...) leaves the
      // matrix on the CSR kernel instead of failing the setup
      try { panel_build(*M, stream); }
      catch (const Error &) { M->panel = DevPanel(); }  // nothing of the matrix is released on this (non-compacting) path: any failure is survivable
    }
  }
}

// Large problems on the indirect back-end: once the sliced-ELL copies exist, the CSR column / value arrays of A, A' and P
// (12 B per stored entry, 36 GB of the 79 GB of rand-1e6) are only ever read by maintenance passes -- Ruiz scaling
// inside osqp_update_P / _A, the Jacobi diagonal after a rho update, value updates by nnz index.  Those walk the slices
// instead (panel.hip, compact mode), the nnz-index maps are rewritten as positions in the slice arrays, and the CSR
// arrays go.  Row pointers stay (8 B per row).
__global__ __launch_bounds__(kBlock) void k_compose_slot_map(int64_t k, const int *k2pos, const uint32_t *__restrict__ pos2slot,
                                                             uint32_t *out) {  // out may alias k2pos
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= k) return;
  const int p = k2pos ? k2pos[i] : (int)i;
  out[i] = p >= 0 ? pos2slot[p] : 0xFFFFFFFFu;
}
// which: 0 = A, 1 = A', 2 = P.  One matrix at a time, so that a setup which knows it will run the indirect back-end can
// release each CSR copy as soon as its sliced-ELL copy exists (setup_device: the high-water mark of the device memory stays
// near the resident size instead of CSR + slices of all three), and so that a matrix whose slices cannot be built (mostly
// padding) simply keeps its CSR arrays: every consumer looks at the matrix's own flag.
// The slot maps take the place of the position maps, entry by entry, in the same buffers (a thread reads its position and
// writes its slot): A_k2pos / P_k2lo / P_k2up hold CSR positions while their matrix has CSR arrays, slots of the sliced
// copy once it is compact (the consumers go by the matrix's flag).  p2s: slot of every CSR position; given up here
// (A, P) or kept as the map itself (A': position k of A' is the caller's nnz index k).
void Engine::fold_slot_maps(int which, DevBuf<uint32_t> &p2s) {
  auto compose_in_place = [&](int64_t k, DevBuf<int> &k2pos) {
    if (k > 0) OQ_LAUNCH(k_compose_slot_map, dim3(blocks_for(k)), dim3(kBlock), 0, stream, k, (const int *)k2pos.get(), p2s.get(), (uint32_t *)k2pos.get());
  };
  if (which == 0) { compose_in_place(nnzA, A_k2pos); sync(); p2s.release(); }
  else if (which == 1) { sync(); At_k2slot = std::move(p2s); }
  else { compose_in_place(nnzPtriu, P_k2lo); compose_in_place(nnzPtriu, P_k2up); sync(); p2s.release(); }
}

void Engine::compact_one(int which, DevBuf<uint32_t> *known_slots, bool maps_done) {
  DevCsr &M = which == 0 ? A : (which == 1 ? At : Pf);
  if (M.compact || !panel_can_compact(M)) return;
  const bool maps = !comm && !maps_done;  // a row block (sharded) has no nnz-index maps: value updates are refused there anyway
  if (maps) {
    DevBuf<uint32_t> own;
    if (!(known_slots && known_slots->n > 0)) { own.alloc((size_t)M.nnz); panel_slot_of_pos(M, own.get(), stream); }
    fold_slot_maps(which, (known_slots && known_slots->n > 0) ? *known_slots : own);
  }
  if (which == 2) Pi_keep.release();  // the direct back-end's symbolic phase is out of reach at this size
  if (comm) (which == 0 ? A_org : (which == 1 ? At_org : Pf_org)).release();  // a compact row block refuses value updates (update_PA)
  panel_compact(M);
  compact = true;
}

bool Engine::compact_wanted(int64_t stored) const {
  const double limit = getenv("OSQP_AMD_COMPACT_NNZ") ? atof(getenv("OSQP_AMD_COMPACT_NNZ")) : 5e7;  // stored entries of A + P (< 0: never)
  return limit >= 0.0 && (double)stored >= limit;
}

// the indirect back-end whatever the symbolic phase would say (select_linsys below)
bool Engine::pcg_certain() const {
  const int want = st.linsys_solver;
  if (comm || want == AMD_PCG_SOLVER) return true;
  const double nnzK = (double)nnzPtriu + (double)nnzA + (double)ng + (double)mg;
  return want != AMD_DIRECT_SOLVER && nnzK > 4e7;
}

void Engine::compact_matrices() {
  if (!lin || lin->kind() != 2) return;
  if (!compact_wanted(nnzA + Pf.nnz)) return;
  if (!compact && (!panel_can_compact(Pf) || (m > 0 && !(panel_can_compact(A) && panel_can_compact(At))))) return;
  if (m > 0) { compact_one(0); compact_one(1); }
  compact_one(2);
}

void Engine::unscale_data() {
  csr_scale_rows_cols(Pf, nullptr, nullptr, 0, cinv, stream);
  const double *Dg = full_n(Dinv.get());  // column scalings are indexed by global ids (a row block: the gathered vector)
  csr_scale_rows_cols(Pf, Dinv.get(), Dg, 1, 1.0, stream, n0);
  vec_scale_by_vec_scalar(q.get(), Dinv.get(), cinv, n, stream);
  if (m > 0) {
    csr_scale_rows_cols(A, Einv.get(), Dg, 0, 1.0, stream);
    csr_scale_rows_cols(At, Dinv.get(), full_m(Einv.get()), 2, 1.0, stream);
    vec_ew_prod(l.get(), l.get(), Einv.get(), m, stream);
    vec_ew_prod(u.get(), u.get(), Einv.get(), m, stream);
  }
}

// --------------------------------------------------------------------------
// K1: rho vector
// --------------------------------------------------------------------------
void Engine::set_rho_vec() {
  st.rho = std::min(std::max(st.rho, RHO_MIN), RHO_MAX);
  rho_vec_update(m, l.get(), u.get(), ctype.get(), rho.get(), rho_inv.get(), st.rho, 0, flag.get(), stream);
}

int Engine::update_rho_vec_from_bounds() {
  if (m == 0) return 0;
  flag.zero(stream);
  rho_vec_update(m, l.get(), u.get(), ctype.get(), rho.get(), rho_inv.get(), st.rho, 1, flag.get(), stream);
  int changed = 0;
  flag.download(&changed, 1, stream);
  sync();
  if (comm) changed = agree_max(changed ? 1.0 : 0.0) > 0.0;
  if (changed && lin) return lin->update_rho();
  return 0;
}

int Engine::update_rho(double rho_new) {
  if (rho_new <= 0) return 1;
  st.rho = std::min(std::max(rho_new, RHO_MIN), RHO_MAX);
  rho_vec_update(m, l.get(), u.get(), ctype.get(), rho.get(), rho_inv.get(), st.rho, 2, flag.get(), stream);
  settings_changed();
  return lin ? lin->update_rho() : 0;
}

void Engine::cold_start() {
  x.zero(stream); z.zero(stream); y.zero(stream);
}

// --------------------------------------------------------------------------
// ADMM iteration (K5 + KKT back-end)
// --------------------------------------------------------------------------
int Engine::kkt_solve() {
  double cand = -1.0;
  if (have_res) cand = lambda * std::sqrt(sc_pri * sc_dua);
  else if (have_seed) cand = lambda * g_seed;
  return lin->solve(xz.get(), cand);
}

// one iteration, in place on (x, z, y)
int Engine::admm_step() {
  admm_iters_total++;
  { const int rc = lin->fused_step(); if (rc >= 0) return rc; }  // back-end specific fusion of rhs / solve / update (direct.hip, pcg.hip)
  admm_rhs(n, m, st.sigma, x.get(), q.get(), z.get(), rho_inv.get(), y.get(), xz.get(), stream);
  int rc = kkt_solve();
  admm_update(n, m, st.alpha, xz.get(), rho.get(), rho_inv.get(), l.get(), u.get(), x.get(), z.get(), y.get(), dx.get(),
              dy.get(), stream);
  return rc;
}

// A chunk = `check_termination` iterations of the direct back-end (no host round trip inside), captured once
// in a hipGraph and replayed: removes the per-launch host cost that dominates problems whose iteration is a
// few short kernels.  Only when nothing has to happen between two residual evaluations.
// iterations per chunk: the distance between two residual evaluations; with the evaluations switched off
// (check_termination = 0 [REF test/basic.jl:168-171]) the adaptive-rho interval, or 50 -- nothing else happens between iterations
int Engine::chunk_k() const {
  const long long k = st.check_termination;
  if (k >= 2) return (int)k;
  if (k != 0) return 0;
  if (!st.adaptive_rho) return 50;
  return st.adaptive_rho_interval > 1 ? (int)std::min<long long>(st.adaptive_rho_interval, 100) : 0;
}
bool Engine::can_chunk(long long iter, long long max_iter) const {
  static const bool enabled = !(getenv("OSQP_AMD_GRAPH") && atoi(getenv("OSQP_AMD_GRAPH")) == 0);
  const long long k = chunk_k();
  if (!enabled || g_debug_sync || lin->kind() != 0 || k < 2 || st.verbose || st.time_limit != 0.0) return false;
  if ((iter - 1) % k != 0 || iter + k - 1 > max_iter) return false;
  if (st.adaptive_rho && (st.adaptive_rho_interval == 0 || st.adaptive_rho_interval % k != 0)) return false;
  return true;
}

void Engine::drop_chunk_graph() {
  if (chunk_exec) { (void)hipGraphExecDestroy(chunk_exec); chunk_exec = nullptr; chunk_len = 0; }
}
void Engine::settings_changed() {
  drop_chunk_graph();
  if (lin) { if (int rc = lin->flush()) deferred_error = rc; lin->invalidate(); }
}

void Engine::run_chunk() {
  const int k = chunk_k();
  if (chunk_exec && chunk_len != k) { (void)hipGraphExecDestroy(chunk_exec); chunk_exec = nullptr; }
  if (!chunk_exec) {
    hipGraph_t graph = nullptr;
    HIP_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    // a launch that throws inside the capture must not leave the stream capturing, nor the back-end believing that a
    // next step follows (its right-hand side `left` in the factor's vector): the next chunk would solve with a stale one
    struct CaptureGuard {
      Engine &e; bool armed = true;
      ~CaptureGuard() {
        if (!armed) return;
        e.lin->next_follows = false;
        e.lin->invalidate();
        hipGraph_t g = nullptr;
        (void)hipStreamEndCapture(e.stream, &g);
        if (g) (void)hipGraphDestroy(g);
      }
    } guard{*this};
    for (int i = 0; i < k; i++) { lin->next_follows = i + 1 < k; admm_step(); }
    lin->next_follows = false;
    guard.armed = false;
    HIP_CHECK(hipStreamEndCapture(stream, &graph));
    HIP_CHECK(hipGraphInstantiate(&chunk_exec, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    chunk_len = k;
    admm_iters_total -= k;  // the capture pass did not execute anything
  }
  HIP_CHECK(hipGraphLaunch(chunk_exec, stream));
  admm_iters_total += k;
}

// --------------------------------------------------------------------------
// K8: residuals, objective
// --------------------------------------------------------------------------
// Ax, Px, A'y at the current iterate and the 16 norms / sums of Slot order into h_slots
void Engine::residual_evaluation() {
  if (int rc = lin->flush()) {  // the iterate must be the one the host believes it is
    if (rc == 6) throw TreeFault();
    deferred_error = rc;
  }
  const double *xg = full_n(x.get());
  spmv(A, xg, Ax.get(), nullptr, 0.0, 0.0, nullptr, stream);
  spmv(Pf, xg, Px_.get(), nullptr, 0.0, 0.0, nullptr, stream);
  if (m > 0) spmv(At, full_m(y.get()), Aty.get(), nullptr, 0.0, 0.0, nullptr, stream);
  residual_norms(n, m, x.get(), z.get(), Ax.get(), Px_.get(), Aty.get(), q.get(), Dinv.get(), Einv.get(), slots.get(),
                 partials.get(), stream);
  fetch_slots(0, 16, (1u << S_XPX) | (1u << S_QX));
  // the read-back has drained the stream: a wait that timed out in the iterations that were still in flight at the
  // test above shows now, before anything is decided from these numbers
  if (lin->flush() == 6) throw TreeFault();
}

void Engine::update_info(long long iter, bool compute_objective) {
  OSQPInfo *info = ws->info;
  residual_evaluation();
  memcpy(res, h_slots, sizeof(double) * 16);
  const bool uns = st.scaling && !st.scaled_termination;
  info->iter = iter;
  if (compute_objective) info->obj_val = obj_from_slots();
  info->pri_res = m == 0 ? 0.0 : (uns ? res[S_PRI_UNS] : res[S_PRI]);
  info->dua_res = uns ? cinv * res[S_DUA_UNS] : res[S_DUA];
  sc_pri = m == 0 ? 0.0 : res[S_PRI];
  sc_dua = res[S_DUA];
  // progress monitor of the PCG tolerance rule (same statement as oracle/osqp_oracle.c)
  double g = std::sqrt(sc_pri * sc_dua);
  have_res = true;
  if (!have_ref) { g_ref = g; it_ref = iter; have_ref = true; }
  else if (iter - it_ref >= 25) {
    if (g > 0.5 * g_ref) lambda = std::max(0.25 * lambda, 1e-6);
    g_ref = g; it_ref = iter;
  }
  info->solve_time = toc();
}

double Engine::obj_from_slots() const {
  double obj = 0.5 * res[S_XPX] + res[S_QX];
  if (st.scaling) obj *= cinv;
  return obj;
}

bool Engine::is_primal_infeasible(double eps) {
  const bool uns = st.scaling && !st.scaled_termination;
  prim_infeas_prep(m, dy.get(), l.get(), u.get(), uns ? E.get() : nullptr, slots.get(), partials.get(), stream);
  fetch_slots(S_T0, 2, 2u);
  double norm_dy = h_slots[S_T0], ineq_lhs = h_slots[S_T1];
  if (norm_dy > eps) {
    if (ineq_lhs < -eps * norm_dy) {
      spmv(At, full_m(dy.get()), tn.get(), nullptr, 0.0, 0.0, nullptr, stream);
      HIP_CHECK(hipMemsetAsync(slots.get() + S_T3, 0, sizeof(double), stream));
      reduce_absmax(tn.get(), uns ? Dinv.get() : nullptr, n, slots.get() + S_T3, stream);
      fetch_slots(S_T3, 1);
      return h_slots[S_T3] < eps * norm_dy;
    }
  }
  return false;
}

bool Engine::is_dual_infeasible(double eps) {
  const bool uns = st.scaling && !st.scaled_termination;
  HIP_CHECK(hipMemsetAsync(slots.get() + S_T0, 0, sizeof(double) * 6, stream));
  reduce_absmax(dx.get(), uns ? D.get() : nullptr, n, slots.get() + S_T0, stream);
  reduce_dot(q.get(), dx.get(), n, partials.get(), slots.get() + S_T1, stream);
  fetch_slots(S_T0, 2, 2u);
  double norm_dx = h_slots[S_T0], qdx = h_slots[S_T1];
  double cost_scaling = uns ? c : 1.0;
  if (norm_dx > eps) {
    if (qdx < -cost_scaling * eps * norm_dx) {
      const double *dxg = full_n(dx.get());
      spmv(Pf, dxg, tn2.get(), nullptr, 0.0, 0.0, nullptr, stream);
      HIP_CHECK(hipMemsetAsync(slots.get() + S_T3, 0, sizeof(double), stream));
      reduce_absmax(tn2.get(), uns ? Dinv.get() : nullptr, n, slots.get() + S_T3, stream);
      fetch_slots(S_T3, 1);
      if (h_slots[S_T3] < cost_scaling * eps * norm_dx) {
        spmv(A, dxg, tm.get(), nullptr, 0.0, 0.0, nullptr, stream);
        dual_infeas_rows(m, tm.get(), uns ? Einv.get() : nullptr, l.get(), u.get(), eps * norm_dx, slots.get(), stream);
        fetch_slots(S_T2, 1);  // a count of violating rows: the maximum over ranks is zero iff every count is
        return h_slots[S_T2] == 0.0;
      }
    }
  }
  return false;
}

int Engine::check_termination(bool approximate) {
  OSQPInfo *info = ws->info;
  double eps_abs = st.eps_abs, eps_rel = st.eps_rel, eps_prim_inf = st.eps_prim_inf, eps_dual_inf = st.eps_dual_inf;
  const bool uns = st.scaling && !st.scaled_termination;
  if (!(info->pri_res <= OSQP_INFTY) || !(info->dua_res <= OSQP_INFTY)) {  // also catches NaN
    update_status(info, OSQP_NON_CVX);
    info->obj_val = NAN;
    return 1;
  }
  if (approximate) { eps_abs *= 10; eps_rel *= 10; eps_prim_inf *= 10; eps_dual_inf *= 10; }
  bool prim_res_check = false, dual_res_check = false, prim_inf_check = false, dual_inf_check = false;
  if (m == 0) prim_res_check = true;
  else {
    double mx = uns ? std::max(res[S_Z_UNS], res[S_AX_UNS]) : std::max(res[S_Z], res[S_AX]);
    double eps_prim = eps_abs + eps_rel * mx;
    if (info->pri_res < eps_prim) prim_res_check = true;
    else prim_inf_check = is_primal_infeasible(eps_prim_inf);
  }
  double mxd = uns ? cinv * std::max(res[S_Q_UNS], std::max(res[S_ATY_UNS], res[S_PX_UNS]))
                   : std::max(res[S_Q], std::max(res[S_ATY], res[S_PX]));
  double eps_dual = eps_abs + eps_rel * mxd;
  if (info->dua_res < eps_dual) dual_res_check = true;
  else dual_inf_check = is_dual_infeasible(eps_dual_inf);

  if (prim_res_check && dual_res_check) {
    update_status(info, approximate ? OSQP_SOLVED_INACCURATE : OSQP_SOLVED);
    return 1;
  } else if (prim_inf_check) {
    update_status(info, approximate ? OSQP_PRIMAL_INFEASIBLE_INACCURATE : OSQP_PRIMAL_INFEASIBLE);
    if (uns) vec_ew_prod(dy.get(), dy.get(), E.get(), m, stream);
    info->obj_val = OSQP_INFTY;
    return 1;
  } else if (dual_inf_check) {
    update_status(info, approximate ? OSQP_DUAL_INFEASIBLE_INACCURATE : OSQP_DUAL_INFEASIBLE);
    if (uns) vec_ew_prod(dx.get(), dx.get(), D.get(), n, stream);
    info->obj_val = -OSQP_INFTY;
    return 1;
  }
  return 0;
}

// A.4: on scaled quantities, from the norms of the last residual evaluation
double Engine::compute_rho_estimate() {
  double pri = m == 0 ? 0.0 : res[S_PRI], dua = res[S_DUA];
  double pri_norm = m == 0 ? 0.0 : std::max(res[S_Z], res[S_AX]);
  pri /= (pri_norm + 1e-10);
  double dua_norm = std::max(std::max(res[S_Q], m == 0 ? 0.0 : res[S_ATY]), res[S_PX]);
  dua /= (dua_norm + 1e-10);
  double est = st.rho * std::sqrt(pri / (dua + 1e-10));
  return std::min(std::max(est, RHO_MIN), RHO_MAX);
}

int Engine::adapt_rho() {
  double rho_new = compute_rho_estimate();
  ws->info->rho_estimate = rho_new;
  if (rho_new > st.rho * st.adaptive_rho_tolerance || rho_new < st.rho / st.adaptive_rho_tolerance) {
    int e = update_rho(rho_new);
    ws->info->rho_updates += 1;
    return e;
  }
  return 0;
}

static bool has_solution(const OSQPInfo *info) {
  return info->status_val != OSQP_PRIMAL_INFEASIBLE && info->status_val != OSQP_PRIMAL_INFEASIBLE_INACCURATE &&
         info->status_val != OSQP_DUAL_INFEASIBLE && info->status_val != OSQP_DUAL_INFEASIBLE_INACCURATE &&
         info->status_val != OSQP_NON_CVX;
}

// full-length host copies of an n-vector and / or an m-vector (sharded: gathered first, every rank gets all of it)
void Engine::download_full(const double *vn, const double *vm, double *hn, double *hm) {
  if (vn) HIP_CHECK(hipMemcpyAsync(hn, full_n(vn), sizeof(double) * ng, hipMemcpyDeviceToHost, stream));
  if (vm && mg > 0) HIP_CHECK(hipMemcpyAsync(hm, full_m(vm), sizeof(double) * mg, hipMemcpyDeviceToHost, stream));
  sync();
}

// the current iterate, unscaled, to caller buffers (osqp_amd_get_iterate; the iterate itself is not touched)
void Engine::get_iterate(double *hx, double *hy) {
  if (st.scaling) {
    if (hx) vec_ew_prod(tn.get(), x.get(), D.get(), n, stream);
    if (hy) { vec_ew_prod(tm.get(), y.get(), E.get(), m, stream); vec_scale(tm.get(), cinv, m, stream); }
    download_full(hx ? tn.get() : nullptr, hy ? tm.get() : nullptr, hx, hy);
  } else {
    download_full(hx ? x.get() : nullptr, hy ? y.get() : nullptr, hx, hy);
  }
}

// A.5: host mirrors refreshed here (the Julia side reads solution->x/y, delta_x, delta_y as host pointers)
void Engine::store_solution() {
  OSQPInfo *info = ws->info;
  if (has_solution(info)) {
    if (st.scaling) {
      vec_ew_prod(tn.get(), x.get(), D.get(), n, stream);
      vec_ew_prod(tm.get(), y.get(), E.get(), m, stream);
      vec_scale(tm.get(), cinv, m, stream);
      download_full(tn.get(), tm.get(), h_x.data(), h_y.data());
    } else {
      download_full(x.get(), y.get(), h_x.data(), h_y.data());
    }
  } else {
    std::fill(h_x.begin(), h_x.end(), NAN);
    std::fill(h_y.begin(), h_y.end(), NAN);
    if (info->status_val == OSQP_PRIMAL_INFEASIBLE || info->status_val == OSQP_PRIMAL_INFEASIBLE_INACCURATE) {
      HIP_CHECK(hipMemsetAsync(slots.get() + S_T0, 0, sizeof(double), stream));
      reduce_absmax(dy.get(), nullptr, m, slots.get() + S_T0, stream);
      fetch_slots(S_T0, 1);
      vec_scale(dy.get(), 1.0 / h_slots[S_T0], m, stream);
      download_full(nullptr, dy.get(), nullptr, h_dy.data());
    }
    if (info->status_val == OSQP_DUAL_INFEASIBLE || info->status_val == OSQP_DUAL_INFEASIBLE_INACCURATE) {
      HIP_CHECK(hipMemsetAsync(slots.get() + S_T0, 0, sizeof(double), stream));
      reduce_absmax(dx.get(), nullptr, n, slots.get() + S_T0, stream);
      fetch_slots(S_T0, 1);
      vec_scale(dx.get(), 1.0 / h_slots[S_T0], n, stream);
      download_full(dx.get(), nullptr, h_dx.data(), nullptr);
    }
    cold_start();
    sync();
  }
}

// --------------------------------------------------------------------------
// osqp_solve [REF src/interface.jl:171]
// --------------------------------------------------------------------------
// A timed-out wait inside k_sn_tree (direct back-end, supernodal solves in one launch per direction) leaves an iterate
// that cannot be trusted: the factor has gone back to one launch per level (Direct::flush), the solve starts again from
// a cold start.  A second fault cannot happen (nothing waits inside a kernel any more); if it does it is error 6.
// Ctrl-C [REF src/constants.jl:17 :Interrupted = -5]: for the duration of osqp_solve a SIGINT handler that only sets a flag
// (never calls back into the host runtime: the caller may be a Julia task or a Python thread); the loop tests it at the top
// of an iteration -- between chunks when iterations run as captured graphs -- and the handler found at entry is put back on
// every way out.  Nested / concurrent solves (distinct workspaces on different threads) share one installation.
namespace {
volatile std::sig_atomic_t g_sigint = 0;
std::mutex g_sigint_mutex;
int g_sigint_depth = 0;
struct sigaction g_sigint_prev;
void on_sigint(int) { g_sigint = 1; }
struct InterruptListener {
  InterruptListener() {
    std::lock_guard<std::mutex> lock(g_sigint_mutex);
    if (g_sigint_depth++ == 0) {
      g_sigint = 0;
      struct sigaction sa;
      memset(&sa, 0, sizeof sa);
      sa.sa_handler = on_sigint;
      sigemptyset(&sa.sa_mask);
      sigaction(SIGINT, &sa, &g_sigint_prev);
    }
  }
  ~InterruptListener() {
    std::lock_guard<std::mutex> lock(g_sigint_mutex);
    if (--g_sigint_depth == 0) sigaction(SIGINT, &g_sigint_prev, nullptr);
  }
};
}  // namespace

int Engine::solve() {
  InterruptListener listener;
  // what an abandoned attempt may have changed besides the iterate: rho (adaptive updates) and the interval it settled on
  const double rho_at_entry = st.rho;
  const c_int interval_at_entry = st.adaptive_rho_interval;
  for (int attempt = 0;; attempt++) {
    try {
      int rc = solve_attempt(attempt > 0);
      if (lin->flush() == 6) throw TreeFault();  // store_solution has synchronised: nothing is in flight
      return rc;
    } catch (const TreeFault &) {
      if (attempt >= 1) throw;
      tree_restarts++;
      deferred_error = 0;
      // the second attempt is the solve the caller asked for, from the state the caller left: rho and the adaptive-rho
      // interval go back to their values at entry (a refactorisation when rho had moved)
      st.adaptive_rho_interval = interval_at_entry;
      if (st.rho != rho_at_entry) { const int rc = update_rho(rho_at_entry); if (rc) return rc; }
    }
  }
}

int Engine::solve_attempt(bool restarted) {
  OSQPInfo *info = ws->info;
  long long iter, max_iter = st.max_iter;
  bool can_check_termination = false, can_print = st.verbose != 0;
  const bool compute_cost_function = st.verbose != 0;
  double temp_run_time;
  if (clear_update_time) info->update_time = 0.0;
  rho_update_from_solve = true;
  if (!restarted) tic();  // the time of the abandoned attempt counts
  if (st.verbose && rank() == 0) printf("iter   objective    pri res    dua res    rho\n");
  if (!st.warm_start || restarted) cold_start();
  if (restarted) update_status(info, OSQP_UNSOLVED);
  lin->set_guess(x.get());
  have_res = false; have_ref = false; lambda = lambda0; have_seed = false;
  if (lin->kind() == 2) {  // seed the PCG tolerance rule with the residuals of the start point (as oracle/osqp_oracle.c)
    residual_evaluation();
    g_seed = std::max(m == 0 ? 0.0 : h_slots[S_PRI], h_slots[S_DUA]);
    have_seed = true;
  }

  for (iter = 1; iter <= max_iter; iter++) {
    {
      bool interrupted = g_sigint != 0;
      if (comm) {  // one decision for all ranks, taken where they meet anyway: right after a termination test
        const long long cadence = st.check_termination ? st.check_termination : 25;
        interrupted = (iter % cadence == 1 || cadence == 1) ? agree_max(interrupted ? 1.0 : 0.0) > 0.0 : false;
      }
      if (interrupted) {  // as the published library: no solution stored, the iterate stays (a later solve warm-starts from it)
        update_status(info, OSQP_SIGINT);
        sync();
        info->solve_time = toc();
        rho_update_from_solve = false;
        std::fill(h_x.begin(), h_x.end(), NAN);
        std::fill(h_y.begin(), h_y.end(), NAN);
        *ws->settings = st;
        return 1;
      }
    }
    if (ws->first_run) temp_run_time = info->setup_time + toc();
    else temp_run_time = info->update_time + toc();
    bool out_of_time = st.time_limit && temp_run_time >= st.time_limit;
    if (comm && st.time_limit) out_of_time = agree_max(out_of_time ? 1.0 : 0.0) > 0.0;  // clocks differ between ranks
    if (out_of_time) {
      update_status(info, OSQP_TIME_LIMIT_REACHED);
      can_check_termination = false;
      break;
    }
    if (can_chunk(iter, max_iter)) {
      run_chunk();
      iter += chunk_k() - 1;
    } else if (admm_step()) {  // negative curvature in the indirect solve
      update_status(info, OSQP_NON_CVX); info->obj_val = NAN; info->iter = iter;
      break;
    }
    can_check_termination = st.check_termination && (iter % st.check_termination == 0);
    can_print = st.verbose && ((iter % 200 == 0) || iter == 1);
    if (can_check_termination || can_print) {
      update_info(iter, compute_cost_function);
      if (deferred_error) { deferred_error = 0; update_status(info, OSQP_NON_CVX); info->obj_val = NAN; info->iter = iter; break; }
      if (can_print && rank() == 0) printf("%4lld  %11.4e  %9.2e  %9.2e  %9.2e\n", iter, info->obj_val, info->pri_res, info->dua_res, st.rho);
      if (can_check_termination && check_termination(false)) break;
    }
    if (st.adaptive_rho && !st.adaptive_rho_interval) {
      sync();  // the automatic interval is defined on elapsed solve time (nondeterministic, as in the reference)
      bool reached = toc() > st.adaptive_rho_fraction * info->setup_time;
      if (comm) reached = agree_max(reached ? 1.0 : 0.0) > 0.0;  // one decision for all ranks
      if (reached) {
        long long base = st.check_termination ? st.check_termination : 25;
        long long rounded = base * (long long)std::floor((double)iter / (double)base + 0.5);
        if (rounded < base) rounded = base;
        st.adaptive_rho_interval = rounded;
        if (st.adaptive_rho_interval < st.check_termination) st.adaptive_rho_interval = st.check_termination;
      }
    }
    if (st.adaptive_rho && st.adaptive_rho_interval && (iter % st.adaptive_rho_interval == 0)) {
      if (!can_check_termination && !can_print) update_info(iter, compute_cost_function);
      if (deferred_error) { deferred_error = 0; update_status(info, OSQP_NON_CVX); info->obj_val = NAN; info->iter = iter; break; }
      if (adapt_rho()) { update_status(info, OSQP_NON_CVX); break; }
    }
  }

  if (!can_check_termination && info->status_val != OSQP_NON_CVX) {
    if (!can_print) update_info(iter - 1, compute_cost_function);
    if (deferred_error) { deferred_error = 0; update_status(info, OSQP_NON_CVX); info->obj_val = NAN; }
    else check_termination(false);
  }
  if (!compute_cost_function && has_solution(info)) info->obj_val = obj_from_slots_fresh();
  if (info->status_val == OSQP_UNSOLVED) {
    if (!check_termination(true)) update_status(info, OSQP_MAX_ITER_REACHED);
  }
  if (info->status_val == OSQP_TIME_LIMIT_REACHED) {
    if (!check_termination(true)) update_status(info, OSQP_TIME_LIMIT_REACHED);
  }
  info->rho_estimate = compute_rho_estimate();
  sync();
  info->solve_time = toc();
  if (st.polish && info->status_val == OSQP_SOLVED) polish();  // a row block polishes iteratively on the operator of its indirect back-end (pcg.hip)
  if (ws->first_run) info->run_time = info->setup_time + info->solve_time + info->polish_time;
  else info->run_time = info->update_time + info->solve_time + info->polish_time;
  ws->first_run = 0;
  clear_update_time = true;
  rho_update_from_solve = false;
  if (st.verbose && rank() == 0)
    printf("status: %s, iterations: %lld, objective: %.6e, run time: %.3es\n", info->status, (long long)info->iter,
           info->obj_val, info->run_time);
  store_solution();
  *ws->settings = st;
  return 0;
}

// objective at the current x (Px recomputed: the last residual evaluation may be older than x)
double Engine::obj_from_slots_fresh() {
  spmv(Pf, full_n(x.get()), Px_.get(), nullptr, 0.0, 0.0, nullptr, stream);
  reduce_dot(x.get(), Px_.get(), n, partials.get(), slots.get() + S_T4, stream);
  reduce_dot(q.get(), x.get(), n, partials.get(), slots.get() + S_T5, stream);
  fetch_slots(S_T4, 2, 3u);
  double obj = 0.5 * h_slots[S_T4] + h_slots[S_T5];
  if (st.scaling) obj *= cinv;
  return obj;
}

int Engine::iterate(long long iters) {
  tic();
  lin->set_guess(x.get());
  for (long long it = 1; it <= iters; it++) {
    if (can_chunk(it, iters)) { run_chunk(); it += chunk_k() - 1; }
    else admm_step();
    if (st.check_termination && (it % st.check_termination == 0)) update_info(it, false);
  }
  update_info(iters, true);
  sync();
  if (deferred_error) { int rc = deferred_error; deferred_error = 0; return rc; }
  return 0;
}

// polish (row N1) needs a reduced-KKT factorisation; provided by the direct back-end (direct.hip)
void Engine::polish() {
  tic();
  ws->info->status_polish = polish_run(*this);
  sync();
  ws->info->polish_time = toc();
}

// --------------------------------------------------------------------------
// updates
// --------------------------------------------------------------------------
void Engine::begin_update() {
  if (clear_update_time) { clear_update_time = false; ws->info->update_time = 0.0; }
  tic();
}
void Engine::end_update() { sync(); ws->info->update_time += toc(); }

int Engine::update_lin_cost(const double *q_new) {
  begin_update();
  q.upload(q_new + n0, n, stream);
  sync();
  if (st.scaling) vec_scale_by_vec_scalar(q.get(), D.get(), c, n, stream);
  reset_info(ws->info);
  end_update();
  return 0;
}

int Engine::update_bounds(const double *l_new, const double *u_new) {
  begin_update();
  std::vector<double> nl(h_l), nu(h_u);
  if (l_new) nl.assign(l_new + m0, l_new + m0 + m);
  if (u_new) nu.assign(u_new + m0, u_new + m0 + m);
  bool bad = false;
  for (int i = 0; i < m; i++) if (nl[i] > nu[i]) bad = true;
  if (comm) bad = agree_max(bad ? 1.0 : 0.0) > 0.0;
  if (bad) return 1;
  h_l.swap(nl); h_u.swap(nu);
  if (l_new) { l.upload(h_l.data(), m, stream); sync(); if (st.scaling) vec_ew_prod(l.get(), l.get(), E.get(), m, stream); }
  if (u_new) { u.upload(h_u.data(), m, stream); sync(); if (st.scaling) vec_ew_prod(u.get(), u.get(), E.get(), m, stream); }
  reset_info(ws->info);
  int e = update_rho_vec_from_bounds();
  end_update();
  return e;
}

__global__ __launch_bounds__(kBlock) void k_scatter_vals(int64_t k, const long long *__restrict__ idx, const double *__restrict__ v,
                                                         double *__restrict__ t1, const int *__restrict__ map1,
                                                         double *__restrict__ t2, const int *__restrict__ map2) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= k) return;
  int64_t e = idx ? idx[i] : i;
  double val = v[i];
  if (map1) { int p = map1[e]; if (p >= 0) t1[p] = val; } else t1[e] = val;
  if (t2) { int p = map2[e]; if (p >= 0) t2[p] = val; }
}

__global__ __launch_bounds__(kBlock) void k_scatter_vals_slot(int64_t k, const long long *__restrict__ idx, const double *__restrict__ v,
                                                              double *__restrict__ t1, const uint32_t *__restrict__ map1,
                                                              double *__restrict__ t2, const uint32_t *__restrict__ map2) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= k) return;
  const int64_t e = idx ? idx[i] : i;
  const double val = v[i];
  const uint32_t p1 = map1[e], p2 = map2[e];
  if (p1 != 0xFFFFFFFFu) t1[p1] = val;
  if (p2 != 0xFFFFFFFFu) t2[p2] = val;
}

// a row block's entry takes the value at its caller-order index unless that one carries the "not updated" mark (all ones)
__global__ __launch_bounds__(kBlock) void k_pick_new_vals(int64_t n, const int *__restrict__ org, const double *__restrict__ all, double *__restrict__ val) {
  const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (q >= n) return;
  const double v = all[org[q]];
  if (__double_as_longlong(v) != -1LL) val[q] = v;
}

int Engine::update_PA(const double *Px_new, const c_int *Pidx, c_int Pn, const double *Ax_new, const c_int *Aidx, c_int An,
                      bool doP, bool doA) {
  begin_update();
  if (doP) { if (Pidx) { if (Pn > nnzPtriu) return 1; } else if (Pn != nnzPtriu && Pn != 0) return 1; }
  if (doA) { if (Aidx) { if (An > nnzA) return 2; } else if (An != nnzA && An != 0) return 2; }
  if (doP && Pidx) for (c_int i = 0; i < Pn; i++) if (Pidx[i] < 0 || Pidx[i] >= nnzPtriu) return 1;
  if (doA && Aidx) for (c_int i = 0; i < An; i++) if (Aidx[i] < 0 || Aidx[i] >= nnzA) return 2;
  // a workspace built from a sorted copy of the caller's A (unsorted columns at setup): the caller's nnz indices -> ours
  std::vector<c_int> Aidx_sorted;
  std::vector<double> Ax_sorted;
  if (doA && !A_to_sorted.empty() && Ax_new) {
    if (Aidx) {
      Aidx_sorted.resize((size_t)An);
      for (c_int i = 0; i < An; i++) Aidx_sorted[(size_t)i] = (c_int)A_to_sorted[(size_t)Aidx[i]];
      Aidx = Aidx_sorted.data();
    } else {  // a full replace in the caller's order, whatever A_new_n says (0 and nnz are both accepted above, as libosqp ignores it)
      Ax_sorted.resize((size_t)nnzA);
      for (int64_t k = 0; k < nnzA; k++) Ax_sorted[(size_t)A_to_sorted[(size_t)k]] = Ax_new[k];
      Ax_new = Ax_sorted.data();
    }
  }
  if (comm) {
    // a compact row block has released its CSR values (and its origin map with them): refused before anything is touched
    if ((doP && Pf.compact) || (doA && (A.compact || At.compact)))
      throw Error(6, "osqp_update_P / osqp_update_A: not available on a compact row-sharded workspace (set OSQP_AMD_COMPACT_NNZ=-1 to keep the CSR arrays)");
    if ((doP && Pf.nnz > 0 && Pf_org.n == 0) || (doA && ((A.nnz > 0 && A_org.n == 0) || (At.nnz > 0 && At_org.n == 0))))
      throw Error(6, "osqp_update_P / osqp_update_A: this row-sharded workspace keeps no nnz-index maps (not set up by column ranges, OSQP_AMD_SHARD_UPDATES=0, or nnz >= 2^31)");
  }
  if (st.scaling) unscale_data();
  auto scatter = [&](const double *vals, const c_int *idx, c_int k, double *t1, const int *map1, double *t2, const int *map2) {
    if (k <= 0) return;
    DevBuf<double> dv((size_t)k);
    DevBuf<long long> di;
    dv.upload(vals, (size_t)k, stream);
    if (idx) { di.alloc((size_t)k); di.upload((const long long *)idx, (size_t)k, stream); }
    OQ_LAUNCH(k_scatter_vals, dim3(blocks_for(k)), dim3(kBlock), 0, stream, (int64_t)k, idx ? di.get() : (const long long *)nullptr,
                       dv.get(), t1, map1, t2, map2);
    sync();
  };
  auto scatter_slots = [&](const double *vals, const c_int *idx, c_int k, double *t1, const uint32_t *map1, double *t2, const uint32_t *map2) {
    if (k <= 0) return;
    DevBuf<double> dv((size_t)k);
    DevBuf<long long> di;
    dv.upload(vals, (size_t)k, stream);
    if (idx) { di.alloc((size_t)k); di.upload((const long long *)idx, (size_t)k, stream); }
    OQ_LAUNCH(k_scatter_vals_slot, dim3(blocks_for(k)), dim3(kBlock), 0, stream, (int64_t)k, idx ? di.get() : (const long long *)nullptr,
              dv.get(), t1, map1, t2, map2);
    sync();
  };
  // every copy through its own map: slots of the sliced-ELL copy where the CSR arrays are gone, CSR positions otherwise
  const c_int kP = Pidx ? Pn : (c_int)nnzPtriu, kA = Aidx ? An : (c_int)nnzA;
  if (comm) {
    // a row block (round 4): every rank is handed the same new values; they go into a device array in the caller's nnz order
    // (by index: behind a fill that marks the untouched entries) and every entry of this rank's blocks picks its own by
    // the index recorded at setup (At_org / A_org / Pf_org)
    auto replace = [&](const double *vals, const c_int *idx, c_int k, int64_t nnz_all, std::initializer_list<std::pair<DevCsr *, DevBuf<int> *>> blocks) {
      if (k <= 0 || nnz_all <= 0) return;
      DevBuf<double> all((size_t)nnz_all);
      if (idx) {
        HIP_CHECK(hipMemsetAsync(all.get(), 0xFF, sizeof(double) * (size_t)nnz_all, stream));  // all ones: "not updated"
        DevBuf<double> dv((size_t)k);
        DevBuf<long long> di((size_t)k);
        dv.upload(vals, (size_t)k, stream);
        di.upload((const long long *)idx, (size_t)k, stream);
        OQ_LAUNCH(k_scatter_vals, dim3(blocks_for(k)), dim3(kBlock), 0, stream, (int64_t)k, (const long long *)di.get(), (const double *)dv.get(),
                  all.get(), (const int *)nullptr, (double *)nullptr, (const int *)nullptr);
        sync();
      } else all.upload(vals, (size_t)nnz_all, stream);
      for (auto &b : blocks)
        if (b.first->nnz > 0)
          OQ_LAUNCH(k_pick_new_vals, dim3(blocks_for(b.first->nnz)), dim3(kBlock), 0, stream, (int64_t)b.first->nnz, (const int *)b.second->get(),
                    (const double *)all.get(), b.first->val.get());
      sync();
    };

    if (doP) replace(Px_new, Pidx, kP, nnzPtriu, {{&Pf, &Pf_org}});
    if (doA) replace(Ax_new, Aidx, kA, nnzA, {{&At, &At_org}, {&A, &A_org}});
  } else {
  if (doP) {
    if (Pf.compact) scatter_slots(Px_new, Pidx, kP, Pf.panel.sval.get(), (const uint32_t *)P_k2lo.get(), Pf.panel.sval.get(), (const uint32_t *)P_k2up.get());
    else scatter(Px_new, Pidx, kP, Pf.val.get(), P_k2lo.get(), Pf.val.get(), P_k2up.get());
  }
  if (doA) {
    if (At.compact && A.compact) scatter_slots(Ax_new, Aidx, kA, At.panel.sval.get(), At_k2slot.get(), A.panel.sval.get(), (const uint32_t *)A_k2pos.get());
    else if (!At.compact && !A.compact) scatter(Ax_new, Aidx, kA, At.val.get(), nullptr, A.val.get(), A_k2pos.get());
    else {
      if (At.compact) scatter_slots(Ax_new, Aidx, kA, At.panel.sval.get(), At_k2slot.get(), At.panel.sval.get(), At_k2slot.get());
      else scatter(Ax_new, Aidx, kA, At.val.get(), nullptr, nullptr, nullptr);
      if (A.compact) scatter_slots(Ax_new, Aidx, kA, A.panel.sval.get(), (const uint32_t *)A_k2pos.get(), A.panel.sval.get(), (const uint32_t *)A_k2pos.get());
      else scatter(Ax_new, Aidx, kA, A.val.get(), A_k2pos.get(), nullptr, nullptr);
    }
  }
  }
  if (st.scaling) scale_data();
  refresh_panels();
  settings_changed();
  int e = lin->update_matrices();
  reset_info(ws->info);
  end_update();
  return e;
}

int Engine::warm_start(const double *xw, const double *yw) {
  st.warm_start = 1;
  ws->settings->warm_start = 1;
  if (xw) {
    x.upload(xw + n0, n, stream); sync();
    if (st.scaling) vec_ew_prod(x.get(), x.get(), Dinv.get(), n, stream);
    if (m > 0) spmv(A, full_n(x.get()), z.get(), nullptr, 0.0, 0.0, nullptr, stream);
  } else {
    // The single-vector forms reset the other block: "setting warm start for x only zeroes the stored warm start
    // for y and vice versa" [REF src/modcaches.jl:196] -- the reference's own statement of what the pinned
    // libosqp does, and the reason its MOI layer sends both vectors together when both are dirty.
    x.zero(stream); z.zero(stream);
  }
  if (yw) {
    y.upload(yw + m0, m, stream); sync();
    if (st.scaling) vec_scale_by_vec_scalar(y.get(), Einv.get(), c, m, stream);
  } else {
    y.zero(stream);
  }
  sync();
  return 0;
}

// --------------------------------------------------------------------------
// back-end selection: the reference's `linsys_solver` setting
// [REF src/constants.jl:1-2, src/interface.jl:749-773] plus two extension values
// --------------------------------------------------------------------------
void Engine::select_linsys() {
  int want = st.linsys_solver;
  if (comm) {  // a row block: only the indirect back-end shards (triangular solves are a dependency chain)
    if (want == AMD_DIRECT_SOLVER) throw Error(2, "the direct back-end is not available on a row-sharded workspace");
    lin = make_pcg(*this);
    return;
  }
  if (want == AMD_PCG_SOLVER) { lin = make_pcg(*this); return; }
  // QDLDL / MKL Pardiso / AMD_DIRECT: direct LDL'.  For QDLDL ("auto") fall back to PCG when the
  // KKT matrix is too large for a host symbolic analysis or its factor cannot fit.
  const double nnzK = (double)nnzPtriu + (double)nnzA + (double)n + (double)m;
  if (want != AMD_DIRECT_SOLVER && nnzK > 4e7) { lin = make_pcg(*this); return; }
  int err = 0;
  lin = make_direct(*this, &err);
  if (!lin) {
    if (err == -1 && want != AMD_DIRECT_SOLVER) { lin = make_pcg(*this); return; }  // predicted fill too large
    throw Error(err > 0 ? err : 4, err == 5 ? "KKT matrix has the wrong inertia: problem non-convex" : "direct back-end failed to initialise");
  }
}

}  // namespace oq
