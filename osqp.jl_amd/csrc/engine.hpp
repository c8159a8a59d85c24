// engine.hpp -- device-resident OSQP ADMM engine (one per OSQPWorkspace).
//
// Data layout in HBM (all fp64 values, 32-bit column indices, 64-bit row pointers):
//   A   CSR m x n   (rows of A, for z~ = A x, A dx, residual A x)
//   At  CSR n x m   (rows of A' = the caller's CSC arrays as they come, for A'y)
//   Pf  CSR n x n   (full symmetric P expanded from the caller's upper triangle)
//   k2pos maps from the caller's nnz order (CSC of A, CSC of triu P) into A / Pf,
//   so that osqp_update_P / osqp_update_A are O(k) scatters.
// Iterates x, z, y (updated in place), xz_tilde, delta_x, delta_y, Ax, Px, Aty stay
// in HBM for the whole solve; every check_termination iterations 16 scalars
// come back to the host.
#pragma once
#include <chrono>
#include <memory>

#include "comm.hpp"
#include "kernels.hpp"

namespace oq {

struct Engine;

// KKT back-end interface (the reference's `linsys_solver` plug-in point,
// [REF src/constants.jl:1-2, src/interface.jl:749-773]).
struct Linsys {
  virtual ~Linsys() {}
  virtual int kind() const = 0;  // 0 direct LDL', 2 PCG
  // xz = [sigma x_prev - q ; z_prev - rho^-1 y] on entry, [x~ ; z~] on exit.
  // tol_candidate: lambda*sqrt(pri*dua) of the last residual evaluation (<0: none yet).
  // Returns 0, or 5 when negative curvature is met (problem non-convex).
  virtual int solve(double *xz, double tol_candidate) = 0;
  virtual int update_rho() = 0;       // engine.rho / rho_inv changed
  virtual int update_matrices() = 0;  // engine.A / At / Pf values changed
  virtual void set_guess(const double *x) {}
  // one whole ADMM iteration with back-end specific fusion; -1 = not provided (generic path runs), 0 done, 5 negative
  // curvature met (problem non-convex)
  virtual int fused_step() { return -1; }
  // set by the engine around a fused_step that is captured into a chunk graph: another iteration follows this one inside the
  // same graph with nothing in between, so the step may leave that iteration's right-hand side behind (direct.hip)
  bool next_follows = false;
  // back-ends that enqueue work ahead of the host (pcg.hip): wait for it; 0, or 5 when negative curvature was met
  virtual int flush() { return 0; }
  // a scalar the enqueued / captured work holds by value changed (alpha, sigma)
  virtual void invalidate() {}
  virtual double nnzL() const { return 0.0; }
  virtual double levels() const { return 0.0; }
  virtual double supernode_levels() const { return 0.0; }  // 0: the solves walk the level schedule
  virtual double multifrontal() const { return 0.0; }      // 1: the numeric factorisation runs by supernodes (mfront.hpp)
  virtual double dense_block() const { return 0.0; }       // pivots of the dense top block inverted explicitly (0: none)
  virtual double lean_setup() const { return 0.0; }        // 1: the factor's index arrays were built on the device from a lean analysis
  virtual double trisolve_bytes() const { return 0.0; }
  virtual double factorizations() const { return 0.0; }
  virtual double cg_iters() const { return 0.0; }
  virtual float time_solve(int reps) { return -1.f; }
};

struct HostCsc {  // host copy of a sparsity pattern (for the symbolic phase of the direct back-end)
  int rows = 0, cols = 0;
  std::vector<int64_t> p;
  std::vector<int> i;
};

// The caller's problem in column ranges of its CSC form (sharded setup: a rank looks at every column once or twice
// but keeps only its row block, so its peak memory is the block plus one range).
struct ColumnSource {
  int n = 0, m = 0;
  int64_t nnzP = 0, nnzA = 0;  // of triu(P) and A, whole problem
  virtual ~ColumnSource() {}
  // columns [j0, j1): pointers relative to the range (j1 - j0 + 1 of them), row indices, values; returns the entry count
  virtual int64_t P_chunk(int j0, int j1, DevBuf<int64_t> &p, DevBuf<int> &i, DevBuf<double> &x, hipStream_t s) = 0;
  virtual int64_t A_chunk(int j0, int j1, DevBuf<int64_t> &p, DevBuf<int> &i, DevBuf<double> &x, hipStream_t s) = 0;
  virtual void vectors(DevBuf<double> &q, DevBuf<double> &l, DevBuf<double> &u, hipStream_t s) = 0;  // full length
};
std::unique_ptr<ColumnSource> generated_columns(int kind, int n, int per_row, unsigned long long seed, hipStream_t s);
std::unique_ptr<ColumnSource> host_columns(const OSQPData *data);

struct Engine {
  int n = 0, m = 0;
  // Row-sharded mode (row N4, comm.hpp): `comm` is set before setup and this engine holds block `rank` of the
  // row partition.  n, m are then the LOCAL sizes (all per-element kernels run on the local slices), ng, mg the
  // global ones; A is m x ng, At n x mg, Pf n x ng with global column ids; the input of a sparse product is
  // all-gathered into gn / gm first (full_n / full_m) and every scalar the host or a later kernel reads is
  // combined over the ranks in rank order (combine_slots).  Without a communicator ng = n, mg = m and all of
  // that is a no-op.
  Comm *comm = nullptr;  // not owned
  int ng = 0, mg = 0, n0 = 0, m0 = 0, chunk_n = 0, chunk_m = 0;
  DevBuf<double> gn, gm, gslots;
  hipStream_t aux_stream = nullptr;  // the m-vector exchange of a CG iteration runs here, beside the P product (full_m_begin / _end)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  OSQPSettings st;
  hipStream_t stream = nullptr;
  int device = 0;

  DevCsr A, At, Pf;
  DevBuf<int> A_k2pos, P_k2lo, P_k2up;
  // compact mode (compact_matrices): the same maps as positions in the sliced-ELL value arrays (0xFFFFFFFF: none)
  // the caller handed over columns of A whose row indices do not ascend (libosqp accepts that; the panel layout and the
  // CSR view of A' need them sorted): the workspace was built from a sorted copy, and osqp_update_A translates the caller's
  // nnz indices through this map (caller index -> index in the sorted copy); empty otherwise
  std::vector<int64_t> A_to_sorted;
  // row-sharded workspaces: the caller's nnz index of every entry of this rank's blocks (osqp_update_P / _A pick their new values by it)
  DevBuf<int> At_org, A_org, Pf_org;
  DevBuf<uint32_t> At_k2slot;  // slot of A' entry k in its sliced copy; A and P keep theirs in A_k2pos / P_k2lo / P_k2up (compact_one)
  bool compact = false;
  DevBuf<int64_t> Pp_keep;  // caller's triu(P) CSC pattern, kept for the direct back-end's symbolic phase
  DevBuf<int> Pi_keep;
  int64_t nnzA = 0, nnzPtriu = 0;
  HostCsc hP, hA;           // patterns on the host (filled lazily)
  bool have_host_pattern = false;

  DevBuf<double> q, l, u, D, Dinv, E, Einv, rho, rho_inv;
  DevBuf<int> ctype, flag;
  DevBuf<double> x, z, y, xz, dx, dy, Ax, Px_, Aty, tn, tm, tn2, tm2;
  DevBuf<double> slots, partials;
  double *h_slots = nullptr;  // pinned; read_slots publishes into it through its device mapping
  double *h_slots_dev = nullptr;
  unsigned long long *h_seq = nullptr, *h_seq_dev = nullptr, publish_seq = 0;
  double res[16] = {0};       // norms of the last residual evaluation (Slot order)
  double c = 1.0, cinv = 1.0;
  std::vector<double> h_l, h_u;  // unscaled bounds on the host (validation of bound updates)

  std::unique_ptr<Linsys> lin;

  // PCG tolerance rule state
  double sc_pri = 0, sc_dua = 0, lambda0 = 0.015, lambda = 0.015, g_ref = 0;
  long long it_ref = 0;
  bool have_res = false, have_ref = false, have_seed = false;
  double g_seed = 0;

  int deferred_error = 0;  // an asynchronous back-end met negative curvature some iterations ago (reported at the next check)
  // statistics
  long long admm_iters_total = 0;
  // graph of `check_termination` iterations (direct back-end)
  hipGraphExec_t chunk_exec = nullptr;
  int chunk_len = 0;
  // timers
  std::chrono::steady_clock::time_point t0;
  bool clear_update_time = false, rho_update_from_solve = false;

  // host mirrors (ABI-visible)
  OSQPWorkspace *ws = nullptr;
  std::vector<double> h_x, h_y, h_dx, h_dy;

  Engine();
  ~Engine();

  // ---- setup ----
  // device-resident CSC inputs (ownership of the buffers moves into the engine)
  void setup_device(int n, int m, DevBuf<int64_t> &Pp, DevBuf<int> &Pi, DevBuf<double> &Px, DevBuf<int64_t> &Ap,
                    DevBuf<int> &Ai, DevBuf<double> &Ax, DevBuf<double> &q, DevBuf<double> &l, DevBuf<double> &u,
                    const OSQPSettings &s);
  void setup_host(const OSQPData *data, const OSQPSettings &s);
  // row-sharded: builds this rank's row blocks of A, A' and the full symmetric P from column ranges of the source
  void setup_sharded(ColumnSource &src, const OSQPSettings &s);
  void shard_layout(int &n1, int &m1);
  void shard_vectors(int n1, int m1, DevBuf<double> &q, DevBuf<double> &l, DevBuf<double> &u);
  void open_device();                                                       // stream, readback slots
  void finish_setup(DevBuf<double> &q, DevBuf<double> &l, DevBuf<double> &u);  // everything after the matrices exist
  void fetch_host_pattern();

  // ---- algorithm ----
  void scale_data();
  void unscale_data();
  void refresh_panels();
  void compact_matrices();
  void setup_mark(const char *what);
  double mark_prev = 0.0;
  void compact_one(int which, DevBuf<uint32_t> *known_slots = nullptr, bool maps_done = false);
  void fold_slot_maps(int which, DevBuf<uint32_t> &p2s);
  bool compact_wanted(int64_t stored) const;
  bool pcg_certain() const;
  void set_rho_vec();
  int update_rho_vec_from_bounds();
  void cold_start();
  int solve();
  int iterate(long long iters);
  int admm_step();
  int chunk_k() const;
  bool can_chunk(long long iter, long long max_iter) const;
  void run_chunk();
  // The captured chunk bakes the scalar launch arguments (alpha, sigma) and the kernel choice into its nodes:
  // whatever changes a setting, rho or the matrices drops it, and the next chunk is captured afresh.
  void settings_changed();
  void drop_chunk_graph();
  int kkt_solve();
  int solve_attempt(bool restarted);
  long long tree_restarts = 0;  // solves run again after a TreeFault
  void residual_evaluation();
  void update_info(long long iter, bool compute_objective);
  double obj_from_slots_fresh();
  void polish();
  void begin_update();
  void end_update();
  int check_termination(bool approximate);
  bool is_primal_infeasible(double eps);
  bool is_dual_infeasible(double eps);
  double compute_rho_estimate();
  int adapt_rho();
  int update_rho(double rho_new);
  void store_solution();
  void get_iterate(double *hx, double *hy);
  void download_full(const double *vn, const double *vm, double *hn, double *hm);
  double obj_from_slots() const;

  // ---- updates ----
  int update_lin_cost(const double *q_new);
  int update_bounds(const double *l_new, const double *u_new);
  int update_PA(const double *Px, const c_int *Pidx, c_int Pn, const double *Ax, const c_int *Aidx, c_int An, bool doP, bool doA);
  int warm_start(const double *x, const double *y);

  // ---- helpers ----
  void tic() { t0 = std::chrono::steady_clock::now(); }
  double toc() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
  void sync() { HIP_CHECK(hipStreamSynchronize(stream)); }
  // slots [first, first + count) to h_slots; sharded: combined over the ranks first (bit k of sum_mask: slot
  // first + k is a sum, otherwise a max).  A slot must not be combined twice.
  void fetch_slots(int first, int count, unsigned sum_mask = 0);
  void combine_slots(int first, int count, unsigned sum_mask);
  void read_slots(int first, int count);  // slots [first, first + count) to h_slots (publish kernel + spin, or copy + stream sync)
  // the same in two halves for kernels that publish their own results: hand begin_publish() to the kernel (it writes
  // h_slots through the mapping, then the sequence number), wait_publish() spins until it has
  Publish begin_publish();
  void wait_publish(const Publish &p);
  double agree_max(double v);
  const double *full_n(const double *v);  // v (n local entries) as a full-length vector
  const double *full_m(const double *v);
  const double *full_m_begin(const double *v);  // the same exchange on a second stream ...
  void full_m_end();                            // ... joined here
  void shard_rows(DevBuf<double> &q_, DevBuf<double> &l_, DevBuf<double> &u_);
  int rank() const { return comm ? comm->rank : 0; }
  void select_linsys();
};

std::unique_ptr<Linsys> make_pcg(Engine &e);
// returns nullptr with *err = -1 when the predicted factor is too large (caller may fall back to PCG),
// 4 on a numeric failure, 5 on wrong inertia (non-convex)
std::unique_ptr<Linsys> make_direct(Engine &e, int *err);
// polish: returns status_polish (1 success, -1 failed, 0 not attempted)
int polish_run(Engine &e);
// the same without a factorisation, on the operator of the indirect back-end (pcg.hip): compact workspaces, factors that do not fit
int polish_run_pcg(Engine &e);
const char *last_error_cstr();

// device generators (gen.hip)
void generate_problem(int kind, int n, int per_row, unsigned long long seed, hipStream_t s, int &n_out, int &m_out,
                      DevBuf<int64_t> &Pp, DevBuf<int> &Pi, DevBuf<double> &Px, DevBuf<int64_t> &Ap, DevBuf<int> &Ai,
                      DevBuf<double> &Ax, DevBuf<double> &q, DevBuf<double> &l, DevBuf<double> &u);

void update_status(OSQPInfo *info, c_int status_val);
// the checks of osqp_setup (abi.hip); 0 = valid
int validate_data(const OSQPData *d);
int validate_settings(const OSQPSettings *s);
void set_last_error(const std::string &m);

// runs the enclosing scope on `device` and puts the caller's current device back on exit
struct DeviceScope {
  int prev = -1;
  explicit DeviceScope(int want) {
    HIP_CHECK(hipGetDevice(&prev));
    if (prev != want) HIP_CHECK(hipSetDevice(want)); else prev = -1;
  }
  ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
  DeviceScope(const DeviceScope &) = delete;
  DeviceScope &operator=(const DeviceScope &) = delete;
};

}  // namespace oq
