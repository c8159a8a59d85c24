// abi.hip -- the C-ABI of libosqp_amd.so (include/osqp_amd.h): the 30 symbols
// osqp/OSQP.jl binds [REF src/interface.jl:147-709, src/types.jl:139] plus the
// extension entry points.  Exceptions stop here and become the integer exit
// flags the Julia side tests with `!= 0` [REF src/interface.jl:157-159].
#include "engine.hpp"
#include "symbolic.hpp"

#include <cstring>
#include <thread>
#include <atomic>
#include <cmath>

using namespace oq;

namespace {

struct Impl {
  Engine eng;
  OSQPData data;
  OSQPSettings settings;
  OSQPSolution solution;
  OSQPInfo info;
};

Engine *E(OSQPWorkspace *w) { return &((Impl *)w->impl)->eng; }

// Every entry point that takes a workspace runs on the device the workspace was set up on, whatever device the
// calling thread has current, and leaves the caller's device as it found it.
struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int want) {
    if (hipGetDevice(&prev) != hipSuccess) { prev = -1; return; }
    if (prev == want) { prev = -1; return; }
    if (hipSetDevice(want) != hipSuccess) prev = -1;
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
  DeviceGuard(const DeviceGuard &) = delete;
  DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define OQ_ON_DEVICE(w) DeviceGuard _dev_guard(E(w)->device)
const Engine *E(const OSQPWorkspace *w) { return &((const Impl *)w->impl)->eng; }

}  // namespace

namespace oq {

int validate_data(const OSQPData *d) {
  if (!d || !d->P || !d->A || !d->q) return 1;
  if (d->n <= 0 || d->m < 0) return 1;
  if (d->n >= 2147483647LL || d->m >= 2147483647LL) return 1;
  if (d->P->m != d->n || d->P->n != d->n) return 1;
  if (d->P->p[d->n] <= 10000000)  // larger ones: the device checks the same while it builds the full symmetric P (engine.hip: k_narrow_indices, k_sym_coo)
    for (c_int j = 0; j < d->n; j++)
      for (c_int k = d->P->p[j]; k < d->P->p[j + 1]; k++)
        if (d->P->i[k] > j || d->P->i[k] < 0) return 1;  // P must be upper triangular
  if (d->A->m != d->m || d->A->n != d->n) return 1;
  for (c_int j = 0; j < d->m; j++)
    if (d->l[j] > d->u[j]) return 1;
  return 0;
}

int validate_settings(const OSQPSettings *s) {
  if (!s) return 1;
  if (s->scaling < 0) return 1;
  if (s->adaptive_rho != 0 && s->adaptive_rho != 1) return 1;
  if (s->adaptive_rho_interval < 0) return 1;
  if (s->adaptive_rho_fraction <= 0) return 1;
  if (s->adaptive_rho_tolerance < 1.0) return 1;
  if (s->polish_refine_iter < 0) return 1;
  if (s->rho <= 0.0 || s->sigma <= 0.0 || s->delta <= 0.0) return 1;
  if (s->max_iter <= 0) return 1;
  if (s->eps_abs < 0.0 || s->eps_rel < 0.0) return 1;
  if (s->eps_abs == 0.0 && s->eps_rel == 0.0) return 1;
  if (s->eps_prim_inf <= 0.0 || s->eps_dual_inf <= 0.0) return 1;
  if (s->alpha <= 0.0 || s->alpha >= 2.0) return 1;
  if (s->linsys_solver < 0 || s->linsys_solver > 3) return 1;
  if (s->verbose != 0 && s->verbose != 1) return 1;
  if (s->scaled_termination != 0 && s->scaled_termination != 1) return 1;
  if (s->check_termination < 0) return 1;
  if (s->warm_start != 0 && s->warm_start != 1) return 1;
  if (s->time_limit < 0.0) return 1;
  return 0;
}

}  // namespace oq

namespace {

OSQPWorkspace *new_workspace() {
  OSQPWorkspace *w = (OSQPWorkspace *)calloc(1, sizeof(OSQPWorkspace));
  Impl *im = new Impl();
  memset(&im->data, 0, sizeof(OSQPData));
  memset(&im->info, 0, sizeof(OSQPInfo));
  w->impl = im;
  w->data = &im->data;
  w->settings = &im->settings;
  w->solution = &im->solution;
  w->info = &im->info;
  im->eng.ws = w;
  return w;
}

void finish_setup(OSQPWorkspace *w) {
  Impl *im = (Impl *)w->impl;
  Engine &e = im->eng;
  im->data.n = e.ng; im->data.m = e.mg;
  im->settings = e.st;
  im->solution.x = e.h_x.data(); im->solution.y = e.h_y.data();
  w->delta_x = e.h_dx.data(); w->delta_y = e.h_dy.data();
  update_status(w->info, OSQP_UNSOLVED);
  w->info->status_polish = 0;
  w->info->rho_estimate = e.st.rho;
  w->first_run = 1;
  w->summary_printed = 0;
  if (e.st.verbose && e.comm)  // (a single device: the setup banner of Engine::setup says it all; a row block adds where it sits)
    printf("[osqp-amd] row block of rank %d of %d on device %d: n = %d, m = %d (global), linsys = %s\n", e.rank(), e.comm->world, e.device, e.ng, e.mg,
           e.lin->kind() == 0 ? "direct LDL' (HIP)" : "PCG (HIP)");
}

void destroy(OSQPWorkspace *w) {
  if (!w) return;
  delete (Impl *)w->impl;
  free(w);
}

template <typename F>
c_int guarded(F &&f) {
  try {
    return (c_int)f();
  } catch (const Error &er) {
    set_last_error(er.what());
    return er.code ? er.code : 6;
  } catch (const std::exception &ex) {
    set_last_error(ex.what());
    return 6;
  }
}

void require_device() {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0)
    throw Error(6, "libosqp_amd: no HIP device available (this library has no CPU fallback)");
}

}  // namespace

extern "C" {

void osqp_set_default_settings(OSQPSettings *s) {
  s->rho = 0.1; s->sigma = 1e-6; s->scaling = 10;
  s->adaptive_rho = 1; s->adaptive_rho_interval = 0;
  s->adaptive_rho_tolerance = 5.0; s->adaptive_rho_fraction = 0.4;
  s->max_iter = 4000; s->eps_abs = 1e-3; s->eps_rel = 1e-3;
  s->eps_prim_inf = 1e-4; s->eps_dual_inf = 1e-4; s->alpha = 1.6;
  s->linsys_solver = QDLDL_SOLVER; s->delta = 1e-6; s->polish = 0;
  s->polish_refine_iter = 3; s->verbose = 1; s->scaled_termination = 0;
  s->check_termination = 25; s->warm_start = 1; s->time_limit = 0.0;
}

const char *osqp_version(void) { return "0.6.2"; }

// The columns of A sorted by row on the host (host threads over column ranges), with the map caller index -> sorted index.
// false: a column holds a row twice (refused: the two entries would have to be one).
static bool sort_columns_of_A(const csc *A, std::vector<c_int> &rows, std::vector<c_float> &vals, std::vector<int64_t> &to_sorted) {
  const c_int n = A->n;
  const int64_t nnz = A->p[n];
  rows.resize((size_t)nnz); vals.resize((size_t)nnz); to_sorted.resize((size_t)nnz);
  const int nt = nnz < (1 << 20) ? 1 : (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
  std::atomic<bool> repeated{false};
  auto work = [&](c_int j0, c_int j1) {
    std::vector<std::pair<c_int, int64_t>> col;
    for (c_int j = j0; j < j1; j++) {
      col.clear();
      for (int64_t k = A->p[j]; k < A->p[j + 1]; k++) col.push_back({A->i[k], k});
      std::sort(col.begin(), col.end());
      int64_t s = A->p[j];
      for (size_t t = 0; t < col.size(); t++, s++) {
        if (t > 0 && col[t].first == col[t - 1].first) repeated = true;
        rows[(size_t)s] = col[t].first; vals[(size_t)s] = A->x[col[t].second]; to_sorted[(size_t)col[t].second] = s;
      }
    }
  };
  if (nt == 1) work(0, n);
  else {
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; t++) pool.emplace_back(work, (c_int)((int64_t)n * t / nt), (c_int)((int64_t)n * (t + 1) / nt));
    for (auto &th : pool) th.join();
  }
  return !repeated;
}

static c_int setup_from_host_once(OSQPWorkspace **workp, const OSQPData *data, const OSQPSettings *settings, Comm *comm);

// [REF src/interface.jl:132-155].  libosqp takes the columns of A with their rows in any order; this library's layouts want them
// ascending (what SparseMatrixCSC always hands over).  A caller whose columns are not sorted gets the same behaviour as from
// libosqp at the price of one sorted host copy: the setup is repeated on it and osqp_update_A translates nnz indices.
static c_int setup_from_host(OSQPWorkspace **workp, const OSQPData *data, const OSQPSettings *settings, Comm *comm) {
  g_unsorted_columns = false;
  c_int rc = setup_from_host_once(workp, data, settings, comm);
  const bool unsorted = g_unsorted_columns;  // raised with the UnsortedColumns error (engine.hip), not read off the message text
  g_unsorted_columns = false;
  if (rc != 1 || !unsorted || !data || !data->A) return rc;
  std::vector<c_int> rows;
  std::vector<c_float> vals;
  std::vector<int64_t> to_sorted;
  if (!sort_columns_of_A(data->A, rows, vals, to_sorted)) { set_last_error("a column of A holds the same row twice"); return 1; }
  csc A2 = *data->A;
  A2.i = rows.data(); A2.x = vals.data();
  OSQPData d2 = *data;
  d2.A = &A2;
  rc = setup_from_host_once(workp, &d2, settings, comm);
  if (rc == 0) E(*workp)->A_to_sorted = std::move(to_sorted);
  return rc;
}

static c_int setup_from_host_once(OSQPWorkspace **workp, const OSQPData *data, const OSQPSettings *settings, Comm *comm) {
  if (!workp) return 1;
  *workp = nullptr;
  if (validate_data(data)) { set_last_error("invalid problem data"); return 1; }
  if (validate_settings(settings)) { set_last_error("invalid settings"); return 2; }
  OSQPWorkspace *w = nullptr;
  c_int rc = guarded([&]() {
    require_device();
    w = new_workspace();
    Engine &e = *E(w);
    e.comm = comm;
    e.tic();
    {
      DevCacheScope pooled;  // one stream throughout; what is pooled goes back to the driver inside setup_time
      if (comm) { auto src = host_columns(data); e.setup_sharded(*src, *settings); }
      else e.setup_host(data, *settings);
      dev_cache_trim();
    }
    finish_setup(w);
    w->info->setup_time = e.toc();
    return 0;
  });
  if (rc != 0) { destroy(w); return rc; }
  *workp = w;
  return 0;
}

static c_int setup_from_generator(OSQPWorkspace **workp, c_int kind, c_int n, c_int per_row, unsigned long long seed,
                                  const OSQPSettings *settings, Comm *comm) {
  if (!workp) return 1;
  *workp = nullptr;
  if (validate_settings(settings)) { set_last_error("invalid settings"); return 2; }
  if (n <= 0 || n >= 2147483647LL) return 1;
  OSQPWorkspace *w = nullptr;
  c_int rc = guarded([&]() {
    require_device();
    w = new_workspace();
    Engine &e = *E(w);
    e.comm = comm;
    DevCacheScope pooled;  // the generator's stream and the engine's are separated by a synchronisation
    if (comm) {  // only this rank's row blocks are ever resident
      e.tic();
      auto src = generated_columns((int)kind, (int)n, (int)per_row, seed, nullptr);
      e.setup_sharded(*src, *settings);
      dev_cache_trim();
      finish_setup(w);
      w->info->setup_time = e.toc();
      return 0;
    }
    DevBuf<int64_t> Pp, Ap;
    DevBuf<int> Pi, Ai;
    DevBuf<double> Px, Ax, q, l, u;
    int nn = 0, mm = 0;
    generate_problem((int)kind, (int)n, (int)per_row, seed, nullptr, nn, mm, Pp, Pi, Px, Ap, Ai, Ax, q, l, u);
    HIP_CHECK(hipDeviceSynchronize());
    e.tic();  // setup_time covers setup only; generation stands in for the caller's own data
    e.setup_device(nn, mm, Pp, Pi, Px, Ap, Ai, Ax, q, l, u, *settings);
    dev_cache_trim();  // inside setup_time: pooled chunks are this process's memory until they are returned
    finish_setup(w);
    w->info->setup_time = e.toc();
    return 0;
  });
  if (rc != 0) { destroy(w); return rc; }
  *workp = w;
  return 0;
}

c_int osqp_setup(OSQPWorkspace **workp, const OSQPData *data, const OSQPSettings *settings) {
  return setup_from_host(workp, data, settings, nullptr);
}

c_int osqp_amd_setup_generated(OSQPWorkspace **workp, c_int kind, c_int n, c_int per_row, unsigned long long seed,
                               const OSQPSettings *settings) {
  return setup_from_generator(workp, kind, n, per_row, seed, settings, nullptr);
}

// ---- row-sharded workspaces (row N4): every rank passes the same problem and keeps its own row block ----
c_int osqp_amd_comm_create_host(osqp_amd_comm **out, c_int rank, c_int world, osqp_amd_allgather_fn fn, void *ctx) {
  if (!out) return 1;
  *out = nullptr;
  return guarded([&]() { *out = (osqp_amd_comm *)make_host_comm((int)rank, (int)world, (host_allgather_fn)fn, ctx); return 0; });
}
// OSQP_AMD_RCCL_STUB=1 (tests/test_rccl_stub_transport.py): `librccl_path` names a host-side stand-in for librccl, the
// "device" buffers of osqp_amd_comm_all_gather are host memory and no HIP device is needed -- the RCCL transport's own logic
// (id hand-over, rank bookkeeping, the in-place gather call) runs with several ranks on a box without GPUs
static bool rccl_stub_mode() { const char *e = getenv("OSQP_AMD_RCCL_STUB"); return e && atoi(e) == 1; }
c_int osqp_amd_comm_unique_id(void *out128, const char *librccl_path) {
  if (!out128) return 1;
  return guarded([&]() { if (!rccl_stub_mode()) require_device(); rccl_unique_id(out128, librccl_path); return 0; });
}
c_int osqp_amd_comm_create_rccl(osqp_amd_comm **out, c_int rank, c_int world, const void *unique_id, const char *librccl_path) {
  if (!out) return 1;
  *out = nullptr;
  return guarded([&]() {
    if (!rccl_stub_mode()) require_device();
    *out = (osqp_amd_comm *)make_rccl_comm((int)rank, (int)world, unique_id, librccl_path);
    return 0;
  });
}
c_int osqp_amd_comm_all_gather(osqp_amd_comm *c, c_float *dev_buf, c_int count) {
  if (!c || !dev_buf || count < 0) return 1;
  return guarded([&]() {
    ((Comm *)c)->all_gather(dev_buf, (size_t)count, nullptr);
    if (!(rccl_stub_mode() && std::string(((Comm *)c)->kind()) == "rccl")) HIP_CHECK(hipStreamSynchronize(nullptr));
    return 0;
  });
}
c_int osqp_amd_comm_info(const osqp_amd_comm *c, c_int *rank, c_int *world, c_int *transport_ranks) {
  if (!c) return 1;
  const Comm *k = (const Comm *)c;
  if (rank) *rank = k->rank;
  if (world) *world = k->world;
  if (transport_ranks) *transport_ranks = k->transport_ranks();
  return 0;
}
// Device memory for callers that have no allocator of their own (the packed result array of the batched path: the Python
// mirror and bench.py hand these to osqp_amd_batch_mpc_solve -- torch is not needed for a buffer)
void *osqp_amd_device_alloc(c_int bytes, c_int device) {
  if (bytes <= 0) return nullptr;
  void *p = nullptr;
  c_int rc = guarded([&]() {
    require_device();
    DeviceScope on_device((int)device);
    HIP_CHECK(hipMalloc(&p, (size_t)bytes));
    return 0;
  });
  return rc == 0 ? p : nullptr;
}
c_int osqp_amd_device_free(void *p, c_int device) {
  if (!p) return 0;
  return guarded([&]() { DeviceScope on_device((int)device); HIP_CHECK(hipFree(p)); return 0; });
}
c_int osqp_amd_device_copy(void *dst, const void *src, c_int bytes, c_int kind, c_int device) {  // kind 0: device -> host, 1: host -> device, 2: device -> device
  if (!dst || !src || bytes < 0 || kind < 0 || kind > 2) return 1;
  return guarded([&]() {
    DeviceScope on_device((int)device);
    HIP_CHECK(hipMemcpy(dst, src, (size_t)bytes, kind == 0 ? hipMemcpyDeviceToHost : (kind == 1 ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice)));
    return 0;
  });
}
c_int osqp_amd_symbolic_probe(c_int n, c_int m, const c_int *Pp, const c_int *Pi, const c_int *Ap, const c_int *Ai,
                              c_int ordering, c_int smax, c_float *out, c_int count) {
  if (n <= 0 || m < 0 || !Pp || !Ap || !out || count < 13 || smax < 1) return 1;
  return guarded([&]() {
    HostCsc P, A;
    P.rows = (int)n; P.cols = (int)n; P.p.assign(Pp, Pp + n + 1); P.i.assign(Pi, Pi + Pp[n]);
    A.rows = (int)m; A.cols = (int)n; A.p.assign(Ap, Ap + n + 1); A.i.assign(Ai, Ai + Ap[n]);
    std::vector<int> ident((size_t)m);
    for (int i = 0; i < (int)m; i++) ident[i] = i;
    Symbolic S;
    symbolic_analyse(P, A, ident, (int)m, (int64_t)4000000000LL, 0.0, (int)ordering, S);
    if (S.too_large) return 2;
    Supernodes T;
    build_supernodes(S, (int)smax, T);
    // invariants: slots are a permutation; supernodes are numbered level by level; every entry of L is either inside
    // a diagonal block (below its diagonal) or points from a supernode to one of a strictly lower level
    bool ok = (int)T.piv.size() == S.N && T.ptr.back() == S.N && T.lvl_ptr.back() == T.count;
    std::vector<int> owner(S.N, -1), lvl(T.count, -1);
    for (int L = 0; L < T.nlev; L++)
      for (int J = T.lvl_ptr[L]; J < T.lvl_ptr[L + 1]; J++) lvl[J] = L;
    int largest = 0;
    for (int J = 0; J < T.count && ok; J++) {
      largest = std::max(largest, T.ptr[J + 1] - T.ptr[J]);
      for (int q = T.ptr[J]; q < T.ptr[J + 1]; q++) {
        ok = ok && T.slot[T.piv[q]] == q && (q == T.ptr[J] || T.piv[q] > T.piv[q - 1]);
        owner[q] = J;
      }
    }
    ok = ok && largest <= (int)smax;
    int64_t inside = 0;
    for (int J = 0; J < T.count && ok; J++) {
      const int s = T.ptr[J + 1] - T.ptr[J];
      for (int a = 0; a < s; a++)
        for (int b = 0; b <= a; b++) {
          const int64_t t = T.wmap[T.woff[J] + (int64_t)a * (a + 1) / 2 + b];
          if (t < 0) continue;
          inside++;
          const int col = T.piv[T.ptr[J] + b], row = T.piv[T.ptr[J] + a];
          ok = ok && a > b && t >= S.Lp[col] && t < S.Lp[col + 1] && S.Li[t] == row;
        }
    }
    for (int q = 0; q < S.N && ok; q++) {
      for (int64_t i = T.Fp[q]; i < T.Fp[q + 1]; i++)
        ok = ok && lvl[owner[T.Fj[i]]] < lvl[owner[q]] && S.Li[T.Fpos[i]] == T.piv[q] && (i == T.Fp[q] || T.Fj[i] > T.Fj[i - 1]) &&
             ((i < T.Fsplit[q]) == (lvl[owner[T.Fj[i]]] == 0));
      for (int64_t i = T.Gp[q]; i < T.Gp[q + 1]; i++)
        ok = ok && lvl[owner[T.Gi[i]]] > lvl[owner[q]] && S.Li[T.Gpos[i]] == T.piv[T.Gi[i]] && (i == T.Gp[q] || T.Gi[i] > T.Gi[i - 1]);
    }
    ok = ok && inside + T.Fp[S.N] == S.nnzL && T.Gp[S.N] == T.Fp[S.N];
    out[0] = S.N; out[1] = (double)S.nnzL; out[2] = (double)S.level_ptr.size() - 1; out[3] = T.count; out[4] = T.nlev;
    if (count >= 14) out[13] = (double)kkt_graph_depth(P, A, ident, (int)m);  // what decides the first ordering on large problems
    if (count >= 15) {  // a lean analysis completed on the host must be the full analysis, array for array
      Symbolic S2;
      symbolic_analyse(P, A, ident, (int)m, (int64_t)4000000000LL, 0.0, (int)ordering, S2, true);
      bool same = S2.lean && S2.Lp == S.Lp && S2.Rp == S.Rp && S2.perm == S.perm && S2.level_ptr == S.level_ptr && S2.parent == S.parent &&
                  S2.nnzL == S.nnzL && S2.flops == S.flops && S2.Li.empty() && S2.lean_rows;
      if (same) {
        int64_t held = 0;
        for (const auto &c : S2.lean_rows->cols) held += (int64_t)c.size();
        same = held == S.nnzL && (int)S2.lean_rows->rowid.size() == S.N;
        {  // every row of the walk holds exactly the columns of the row of L it says it is
          std::vector<char> seen((size_t)S.N, 0);
          const LeanRows &R = *S2.lean_rows;
          for (size_t b = 0; b + 1 < R.first.size() && same; b++) {
            size_t c = 0;
            for (int r = R.first[b]; r < R.first[b + 1] && same; r++) {
              const int k = R.rowid[r];
              same = k >= 0 && k < S.N && !seen[k];
              if (!same) break;
              seen[k] = 1;
              const int64_t len = S.Rp[k + 1] - S.Rp[k];
              same = c + (size_t)len <= R.cols[b].size();
              if (!same) break;
              std::vector<int> got(R.cols[b].begin() + c, R.cols[b].begin() + c + len);
              std::sort(got.begin(), got.end());
              same = std::equal(got.begin(), got.end(), S.Rj.begin() + S.Rp[k]);
              c += (size_t)len;
            }
          }
        }
        Supernodes T2;  // the partition does not depend on the lists
        build_supernodes(S2, (int)smax, T2, false, true);
        same = same && T2.count == T.count && T2.ptr == T.ptr && T2.piv == T.piv && T2.up == T.up && T2.lvl_ptr == T.lvl_ptr && T2.woff == T.woff;
        for (int q = 0; q < S.N && same; q++) same = T2.Fp[q + 1] - T2.Fp[q] >= T.Fp[q + 1] - T.Fp[q] && T2.Gp[q + 1] - T2.Gp[q] >= T.Gp[q + 1] - T.Gp[q];
      }
      symbolic_complete(S2);
      same = same && !S2.lean && S2.Lp == S.Lp && S2.Li == S.Li && S2.Rp == S.Rp && S2.Rj == S.Rj && S2.Rmap == S.Rmap && S2.PtoL == S.PtoL &&
             S2.AtoL == S.AtoL;
      out[14] = same ? 1.0 : 0.0;
    }
    out[5] = (double)T.Fp[S.N]; out[6] = (double)T.wdoubles; out[7] = largest; out[8] = ok ? 1.0 : 0.0; out[9] = (double)inside;
    int lD, cD, kD;
    choose_dense_top(S, 512, getenv("OSQP_AMD_DENSE_MAX") ? atoi(getenv("OSQP_AMD_DENSE_MAX")) : 12288, 1024, 32, lD, cD, kD);
    out[10] = level_solve_cost_us(S, 512, lD, kD); out[11] = supernode_solve_cost_us(T, 256);
    out[12] = supernodes_pay(S, T, 512, lD, kD, 256) ? 1.0 : 0.0;
    if (getenv("OSQP_AMD_PROBE_VERBOSE"))
      for (int L = 0; L < T.nlev; L++) {
        int64_t rows = 0, pre = 0, post = 0, back = 0;
        int smax_l = 0;
        for (int J = T.lvl_ptr[L]; J < T.lvl_ptr[L + 1]; J++) {
          smax_l = std::max(smax_l, T.ptr[J + 1] - T.ptr[J]);
          for (int q = T.ptr[J]; q < T.ptr[J + 1]; q++) { rows++; pre += T.Fsplit[q] - T.Fp[q]; post += T.Fp[q + 1] - T.Fsplit[q]; back += T.Gp[q + 1] - T.Gp[q]; }
        }
        {
          int64_t cnt[3] = {0, 0, 0}, rw[3] = {0, 0, 0}, ef[3] = {0, 0, 0}, eb[3] = {0, 0, 0}, mxb[3] = {0, 0, 0};
          for (int J = T.lvl_ptr[L]; J < T.lvl_ptr[L + 1]; J++) {
            const int64_t sz = T.ptr[J + 1] - T.ptr[J];
            const int c = sz == 1 ? 0 : (sz <= Supernodes::kSmall ? 1 : 2);
            cnt[c]++; rw[c] += sz;
            for (int q = T.ptr[J]; q < T.ptr[J + 1]; q++) { ef[c] += T.Fp[q + 1] - T.Fp[q]; eb[c] += T.Gp[q + 1] - T.Gp[q]; mxb[c] = std::max<int64_t>(mxb[c], T.Gp[q + 1] - T.Gp[q]); }
          }
          for (int c = 0; c < 3; c++)
            fprintf(stderr, "level %d class %d (1 / <= small / larger): %lld supernodes, %lld rows, forward entries %lld, backward entries %lld (longest row %lld)\n", L, c,
                    (long long)cnt[c], (long long)rw[c], (long long)ef[c], (long long)eb[c], (long long)mxb[c]);
          int64_t wsm = 0, wbg = 0, rsm = 0, rbg = 0;
          for (int J = T.lvl_ptr[L]; J < T.lvl_ptr[L + 1]; J++) {
            const int64_t sz = T.ptr[J + 1] - T.ptr[J];
            if (J < T.lvl_ptr[L] + T.lvl_small[L]) { wsm += sz * (sz + 1) / 2; rsm += sz; } else { wbg += sz * (sz + 1) / 2; rbg += sz; }
          }
          fprintf(stderr, "level %d: %d small supernodes (%lld rows, %lld block doubles), %d larger (%lld rows, %lld block doubles)\n", L, T.lvl_small[L],
                  (long long)rsm, (long long)wsm, T.lvl_ptr[L + 1] - T.lvl_ptr[L] - T.lvl_small[L], (long long)rbg, (long long)wbg);
        }
        fprintf(stderr, "level %d: %d supernodes, %lld rows (largest %d), per row: %.1f entries at level 0, %.1f above, %.1f backward\n", L,
                T.lvl_ptr[L + 1] - T.lvl_ptr[L], (long long)rows, smax_l, (double)pre / rows, (double)post / rows, (double)back / rows);
        // fronts: border = pattern of the top node's column
        int64_t sb = 0, sb2 = 0, spanel = 0, bmax = 0, fmax = 0; double fl = 0;
        std::vector<int> nch(T.count, 0);
        for (int J = 0; J < T.count; J++) if (T.up[J] >= 0) nch[T.up[J]]++;
        int chmax = 0; int64_t chsum = 0;
        for (int J = T.lvl_ptr[L]; J < T.lvl_ptr[L + 1]; J++) {
          const int top = T.piv[T.ptr[J + 1] - 1];
          const int64_t b = S.Lp[top + 1] - S.Lp[top], s = T.ptr[J + 1] - T.ptr[J];
          sb += b; sb2 += b * b; spanel += (s + b) * s; bmax = std::max(bmax, b); fmax = std::max(fmax, s + b);
          fl += (double)s * (s + b) * (s + b);
          chmax = std::max(chmax, nch[J]); chsum += nch[J];
        }
        fprintf(stderr, "   fronts: border mean %.1f max %lld, front max %lld, sum b^2 %.3g, panel doubles %.3g, dense flops %.3g, children mean %.1f max %d\n",
                (double)sb / (T.lvl_ptr[L + 1] - T.lvl_ptr[L]), (long long)bmax, (long long)fmax, (double)sb2, (double)spanel, fl,
                (double)chsum / (T.lvl_ptr[L + 1] - T.lvl_ptr[L]), chmax);
      }
    if (getenv("OSQP_AMD_PROBE_VERBOSE")) fprintf(stderr, "sum colcount^2 = %.4g\n", S.flops);
    if (getenv("OSQP_AMD_PROBE_VERBOSE")) {  // what the inverted blocks would weigh without the leaf pivots (no row entries, one column entry)
      int64_t leaves = 0, w_now = 0, w_without = 0;
      for (int J = 0; J < T.count; J++) {
        int lv = 0;
        const int64_t sz = T.ptr[J + 1] - T.ptr[J];
        for (int q = T.ptr[J]; q < T.ptr[J + 1]; q++) {
          const int v = T.piv[q];
          if (S.Rp[v + 1] == S.Rp[v] && S.Lp[v + 1] - S.Lp[v] == 1) lv++;
        }
        leaves += lv;
        w_now += sz * (sz + 1) / 2;
        w_without += (sz - lv) * (sz - lv + 1) / 2;
      }
      fprintf(stderr, "leaf pivots %lld of %d; packed block doubles %lld, without the leaves %lld\n", (long long)leaves, S.N, (long long)w_now, (long long)w_without);
    }
    return 0;
  });
}
c_int osqp_amd_comm_destroy(osqp_amd_comm *c) {
  if (!c) return 0;
  try { delete (Comm *)c; } catch (...) { return 1; }
  return 0;
}
c_int osqp_amd_setup_sharded(OSQPWorkspace **workp, const OSQPData *data, const OSQPSettings *settings, osqp_amd_comm *comm) {
  if (!comm) { set_last_error("no communicator"); return 1; }
  return setup_from_host(workp, data, settings, (Comm *)comm);
}
c_int osqp_amd_setup_generated_sharded(OSQPWorkspace **workp, c_int kind, c_int n, c_int per_row, unsigned long long seed,
                                       const OSQPSettings *settings, osqp_amd_comm *comm) {
  if (!comm) { set_last_error("no communicator"); return 1; }
  return setup_from_generator(workp, kind, n, per_row, seed, settings, (Comm *)comm);
}

c_int osqp_solve(OSQPWorkspace *w) {
  if (!w) return 7;
  OQ_ON_DEVICE(w);
  return guarded([&]() { return E(w)->solve(); });
}

c_int osqp_cleanup(OSQPWorkspace *w) {
  if (!w) return 0;  // finalizer of a never-set-up Model [REF src/interface.jl:24-25, 223-229]
  OQ_ON_DEVICE(w);
  try { destroy(w); } catch (...) { return 1; }
  return 0;
}

c_int osqp_update_lin_cost(OSQPWorkspace *w, const c_float *q_new) {
  if (!w) return 7;
  OQ_ON_DEVICE(w);
  return guarded([&]() { return E(w)->update_lin_cost(q_new); });
}
c_int osqp_update_bounds(OSQPWorkspace *w, const c_float *l_new, const c_float *u_new) {
  if (!w) return 7;
  OQ_ON_DEVICE(w);
  return guarded([&]() { return E(w)->update_bounds(l_new, u_new); });
}
c_int osqp_update_lower_bound(OSQPWorkspace *w, const c_float *l_new) {
  if (!w) return 7;
  OQ_ON_DEVICE(w);
  return guarded([&]() { return E(w)->update_bounds(l_new, nullptr); });
}
c_int osqp_update_upper_bound(OSQPWorkspace *w, const c_float *u_new) {
  if (!w) return 7;
  OQ_ON_DEVICE(w);
  return guarded([&]() { return E(w)->update_bounds(nullptr, u_new); });
}
c_int osqp_update_P(OSQPWorkspace *w, const c_float *Px_new, const c_int *Px_new_idx, c_int P_new_n) {
  if (!w) return 7;
  OQ_ON_DEVICE(w);
  return guarded([&]() { return E(w)->update_PA(Px_new, Px_new_idx, P_new_n, nullptr, nullptr, 0, true, false); });
}
c_int osqp_update_A(OSQPWorkspace *w, const c_float *Ax_new, const c_int *Ax_new_idx, c_int A_new_n) {
  if (!w) return 7;
  OQ_ON_DEVICE(w);
  return guarded([&]() { return E(w)->update_PA(nullptr, nullptr, 0, Ax_new, Ax_new_idx, A_new_n, false, true); });
}
c_int osqp_update_P_A(OSQPWorkspace *w, const c_float *Px_new, const c_int *Px_new_idx, c_int P_new_n, const c_float *Ax_new,
                      const c_int *Ax_new_idx, c_int A_new_n) {
  if (!w) return 7;
  OQ_ON_DEVICE(w);
  return guarded([&]() { return E(w)->update_PA(Px_new, Px_new_idx, P_new_n, Ax_new, Ax_new_idx, A_new_n, true, true); });
}

c_int osqp_update_rho(OSQPWorkspace *w, c_float rho_new) {
  if (!w) return 7;
  OQ_ON_DEVICE(w);
  if (rho_new <= 0) return 1;
  return guarded([&]() {
    Engine &e = *E(w);
    e.begin_update();
    int rc = e.update_rho(rho_new);
    w->settings->rho = e.st.rho;
    e.end_update();
    return rc;
  });
}

#define OQ_SETTING(fn, type, field, cond)                 \
  c_int fn(OSQPWorkspace *w, type v) {                    \
    if (!w) return 7;                                     \
    if (!(cond)) return 1;                                \
    OQ_ON_DEVICE(w);                                      \
    return guarded([&]() {                                \
      E(w)->st.field = v;                                 \
      E(w)->settings_changed();                           \
      w->settings->field = v;                             \
      return 0;                                           \
    });                                                   \
  }
OQ_SETTING(osqp_update_max_iter, c_int, max_iter, v > 0)
OQ_SETTING(osqp_update_eps_abs, c_float, eps_abs, v >= 0.)
OQ_SETTING(osqp_update_eps_rel, c_float, eps_rel, v >= 0.)
// libosqp's update functions reject only negative values here (the > 0 rule is the setup validation's): a caller's
// update_settings!(eps_prim_inf = 0) [REF src/interface.jl:506-530] stays legal
OQ_SETTING(osqp_update_eps_prim_inf, c_float, eps_prim_inf, v >= 0.)
OQ_SETTING(osqp_update_eps_dual_inf, c_float, eps_dual_inf, v >= 0.)
OQ_SETTING(osqp_update_alpha, c_float, alpha, v > 0. && v < 2.)
OQ_SETTING(osqp_update_delta, c_float, delta, v > 0.)
OQ_SETTING(osqp_update_polish_refine_iter, c_int, polish_refine_iter, v >= 0)
OQ_SETTING(osqp_update_verbose, c_int, verbose, v == 0 || v == 1)
OQ_SETTING(osqp_update_scaled_termination, c_int, scaled_termination, v == 0 || v == 1)
OQ_SETTING(osqp_update_check_termination, c_int, check_termination, v >= 0)
OQ_SETTING(osqp_update_warm_start, c_int, warm_start, v == 0 || v == 1)
OQ_SETTING(osqp_update_time_limit, c_float, time_limit, v >= 0.)

c_int osqp_update_polish(OSQPWorkspace *w, c_int v) {
  if (!w) return 7;
  OQ_ON_DEVICE(w);
  if (v != 0 && v != 1) return 1;
  return guarded([&]() {
    E(w)->st.polish = v;
    w->settings->polish = v;
    w->info->polish_time = 0.0;
    return 0;
  });
}

c_int osqp_warm_start(OSQPWorkspace *w, const c_float *x, const c_float *y) {
  if (!w) return 7;
  OQ_ON_DEVICE(w);
  return guarded([&]() { return E(w)->warm_start(x, y); });
}
c_int osqp_warm_start_x(OSQPWorkspace *w, const c_float *x) {
  if (!w) return 7;
  OQ_ON_DEVICE(w);
  return guarded([&]() { return E(w)->warm_start(x, nullptr); });
}
c_int osqp_warm_start_y(OSQPWorkspace *w, const c_float *y) {
  if (!w) return 7;
  OQ_ON_DEVICE(w);
  return guarded([&]() { return E(w)->warm_start(nullptr, y); });
}

// ---------------------------------------------------------------- extensions
c_int osqp_amd_get_stats(const OSQPWorkspace *w, c_float *out, c_int count) {
  if (!w || !out) return 0;
  const Engine &e = *E(w);
  c_float v[OSQP_AMD_STATS_COUNT] = {0};
  v[0] = (c_float)e.lin->kind();
  v[1] = (c_float)e.nnzA;
  v[2] = (c_float)e.Pf.nnz;
  v[3] = (c_float)e.nnzPtriu;
  v[4] = e.lin->nnzL();
  v[5] = e.lin->levels();
  v[6] = e.lin->cg_iters();
  v[7] = (c_float)e.admm_iters_total;
  v[8] = e.lin->factorizations();
  v[9] = (c_float)g_device_bytes;
  v[10] = e.A.spmv_bytes();
  v[11] = e.lin->trisolve_bytes();
  v[12] = e.A.panel.active ? (e.A.panel.wide ? 3.0 : 2.0) : 0.0;
  v[13] = e.comm ? (c_float)e.comm->world : 1.0;
  v[14] = e.comm ? e.comm->exchanges : 0.0;
  v[15] = e.comm ? e.comm->bytes : 0.0;
  v[16] = (c_float)e.n;  // local block sizes
  v[17] = (c_float)e.m;
  v[18] = e.compact ? 1.0 : 0.0;
  v[19] = e.lin->supernode_levels();
  v[20] = (c_float)g_device_peak;
  v[21] = (c_float)e.tree_restarts;
  v[22] = e.lin->multifrontal();
  v[23] = e.lin->lean_setup();
  v[24] = (c_float)dev_va_reserved();
  v[25] = e.lin->dense_block();
  c_int k = 0;
  for (; k < count && k < OSQP_AMD_STATS_COUNT; k++) out[k] = v[k];
  return k;
}

c_float osqp_amd_time_kernel(OSQPWorkspace *w, c_int which, c_int reps) {
  if (!w || reps <= 0) return -1.0;
  OQ_ON_DEVICE(w);
  c_float result = -1.0;
  guarded([&]() {
    Engine &e = *E(w);
    hipStream_t s = e.stream;
    if (which == 3) { result = e.lin->time_solve((int)reps); return 0; }
    // sharded: the local block's product on whatever the gather buffers hold (exchange timed separately, id 7)
    const double *xin = e.comm ? e.gn.get() : e.x.get(), *yin = e.comm ? e.gm.get() : e.y.get();
    auto run = [&]() {
      switch (which) {
      case 0: spmv(e.A, xin, e.tm2.get(), nullptr, 0.0, 0.0, nullptr, s); break;
      case 1: spmv(e.At, yin, e.tn2.get(), nullptr, 0.0, 0.0, nullptr, s); break;
      case 2: spmv(e.Pf, xin, e.tn2.get(), nullptr, 0.0, 0.0, nullptr, s); break;
      case 7: if (!e.comm) throw Error(1, "not a sharded workspace"); e.full_n(e.tn.get()); break;
      case 5: e.admm_step(); break;  // one whole ADMM iteration of the back-end in use (advances the iterate)
      case 4:
        admm_update(e.n, e.m, e.st.alpha, e.xz.get(), e.rho.get(), e.rho_inv.get(), e.l.get(),
                    e.u.get(), e.tn.get(), e.tm.get(), e.tm2.get(), e.tn2.get(), e.Ax.get(), s);
        break;
      default: throw Error(1, "unknown kernel id");
      }
    };
    hipEvent_t a, b;
    HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b));
    run();  // warm
    HIP_CHECK(hipEventRecord(a, s));
    for (c_int i = 0; i < reps; i++) run();
    HIP_CHECK(hipEventRecord(b, s));
    HIP_CHECK(hipEventSynchronize(b));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    result = ms / (float)reps;
    return 0;
  });
  return result;
}

c_int osqp_amd_iterate(OSQPWorkspace *w, c_int iters) {
  if (!w) return 7;
  OQ_ON_DEVICE(w);
  return guarded([&]() { return E(w)->iterate(iters); });
}

c_int osqp_amd_get_iterate(OSQPWorkspace *w, c_float *x_out, c_float *y_out) {
  if (!w) return 7;
  OQ_ON_DEVICE(w);
  return guarded([&]() { E(w)->get_iterate(x_out, y_out); return 0; });
}

c_int osqp_amd_apply(OSQPWorkspace *w, c_int op, const c_float *in, c_float *out) {
  if (!w) return 7;
  OQ_ON_DEVICE(w);
  return guarded([&]() {
    Engine &e = *E(w);
    if (e.comm) throw Error(6, "osqp_amd_apply is not available on a row-sharded workspace");
    hipStream_t s = e.stream;
    int nin = op == 0 || op == 2 ? e.n : (op == 1 ? e.m : e.n + e.m);
    int nout = op == 0 ? e.m : (op == 3 ? e.n + e.m : e.n);
    DevBuf<double> di((size_t)nin), dout((size_t)nout);
    di.upload(in, nin, s);
    if (op == 0) spmv(e.A, di.get(), dout.get(), nullptr, 0.0, 0.0, nullptr, s);
    else if (op == 1) spmv(e.At, di.get(), dout.get(), nullptr, 0.0, 0.0, nullptr, s);
    else if (op == 2) spmv(e.Pf, di.get(), dout.get(), nullptr, 0.0, 0.0, nullptr, s);
    else if (op == 3) {
      vec_copy(dout.get(), di.get(), nin, s);
      e.tn.zero(s);
      e.lin->set_guess(e.tn.get());
      int rc = e.lin->solve(dout.get(), 0.0);
      if (rc) return rc;
    } else return 1;
    dout.download(out, nout, s);
    e.sync();
    return 0;
  });
}

c_int osqp_amd_set_device(c_int device) {
  return guarded([&]() { HIP_CHECK(hipSetDevice((int)device)); return 0; });
}

const char *osqp_amd_last_error(void) { return last_error_cstr(); }

}  // extern "C"
