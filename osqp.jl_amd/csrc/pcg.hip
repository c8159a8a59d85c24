// pcg.hip -- indirect KKT back-end (row K9 of SURVEY.md section 8a).
//
// Eliminating nu from [P + sigma I, A'; A, -diag(rho)^-1] [x~; nu] = [r_x; r_z] gives
//     M x~ = r_x + A'(rho .* r_z),   M = P + sigma I + A' diag(rho) A,   z~ = A x~,
// solved by Jacobi-preconditioned CG warm-started from the previous x~.  Needed for
// the random-sparsity configs, whose LDL' factor cannot fit in any memory
// (SURVEY.md section 0.3).  CPU statement: oracle/pcg.c.
//
// SpMV budget.  A textbook statement spends 5 SpMV per ADMM iteration outside the CG loop
// (right-hand side, M x0 for the initial residual, z~ = A x~).  Here A x~ and M x~ are carried
// along the CG recurrences (A x~ += alpha A p, M x~ += alpha M p, both products exist anyway),
// so a solve costs 1 SpMV (right-hand side) + 3 per CG iteration; the two carried vectors are
// recomputed from scratch every kRefresh solves and whenever rho or the matrices change, which
// bounds the drift of the recurrences.
//
// Per CG iteration: A p, P p, A' t + 3 fused vector kernels; alpha and beta never leave the
// device (read from reduction slots by the next kernel); the host reads back ||r||inf and
// p'Mp once per iteration for the stopping test.
//
// Asynchronous form (opt-in: OSQP_AMD_PCG_ASYNC=1).  The loop above needs the host twice per CG iteration -- to read the
// residual norm and p'Mp -- which at mid size (n ~ 1e5: a sparse product takes 30 us) costs more than the kernels.  Here
// the convergence test runs on the device: every kernel of a CG iteration takes a flag and returns at once when it is
// set (kernels.hpp: SkipScope), a one-thread kernel after each iteration (k_pcg_decide) sets the flag of the next one,
// and the host enqueues a whole ADMM iteration -- right-hand side, start vector, `spec` CG iterations, the x / z / y
// update -- as one hipGraph without ever waiting.  Iterations beyond convergence fall through as empty kernels, so the
// arithmetic is that of the loop above, bit for bit.  If `spec` iterations were not enough the step stalls: a flag makes
// every later kernel of every later step fall through, nothing is lost; the next flush (the engine flushes before it
// looks at an iterate: every residual evaluation) finishes that solve with the host loop, re-issues the steps
// behind it and raises `spec`.
#include "engine.hpp"

#include <cmath>
#include <map>

namespace oq {

namespace {

constexpr int kRefresh = 25;
constexpr int kMaxSpec = 8;
enum { C_DONE0 = 0, C_DONE1 = 1, C_STALL = 2, C_ERR = 3, C_STEPS = 4, C_ITERS = 5, C_STALL_IT = 6, C_COUNT = 8 };

// tol from ||b1||inf and the candidate of the tolerance rule; done[0] for the start vector
__global__ void k_pcg_begin(int *ctl, const double *slots, const double *cand_p, double *tol_p) {
  if (ctl[C_STALL]) { ctl[C_DONE0] = 1; return; }
  const double bnorm = slots[S_T5], cand = *cand_p;
  const double hi = 1e-2 * bnorm, lo = 1e-13 * bnorm + 1e-300;
  double tol = hi;
  if (cand >= 0.0) tol = cand;
  if (!(tol < hi)) tol = hi;
  if (tol < lo) tol = lo;
  *tol_p = tol;
  const double rn = slots[S_T1];
  if (rn != rn) { ctl[C_ERR] = 1; ctl[C_STALL] = 1; ctl[C_DONE0] = 1; return; }
  ctl[C_DONE0] = rn <= tol ? 1 : 0;
}
// after CG iteration `it` (flag index cur = it & 1): the flag of the next iteration
__global__ void k_pcg_decide(int *ctl, const double *slots, const double *tol_p, int cur) {
  if (ctl[cur]) { ctl[1 - cur] = 1; return; }
  ctl[C_ITERS] += 1;
  const double pw = slots[S_T4], rn = slots[S_T1 + 2 * (1 - cur)];
  if (!(pw > 0.0) || rn != rn) { ctl[C_ERR] = 1; ctl[C_STALL] = 1; ctl[1 - cur] = 1; return; }  // M is not positive definite
  ctl[1 - cur] = rn <= *tol_p ? 1 : 0;
}
// after the last enqueued iteration: converged -> the step completes, otherwise everything behind it waits for the host
__global__ void k_pcg_end(int *ctl, int final_flag, int spec) {
  if (ctl[C_STALL]) return;
  if (!ctl[final_flag]) { ctl[C_STALL] = 1; ctl[C_STALL_IT] = spec; return; }
  ctl[C_STEPS] += 1;
}

struct Pcg : Linsys {
  Engine &e;
  DevBuf<double> xs, r, zz, p, w, t, u, b1, dinv, Axs, Mxs, xs0, Axs0, Mxs0;
  bool extrapolate = true, have_prev = false;
  long long total_iters = 0;
  int max_iter = 20000;
  bool carried_valid = false;
  int since_refresh = 0;
  // asynchronous form
  bool async_on = false;
  DevBuf<int> ctl;
  DevBuf<double> dctl;  // [0] candidate tolerance of the rule, [1] tolerance of the solve in flight
  int *h_ctl = nullptr;
  double cand_dev = -2.0;           // what dctl[0] holds
  long long issued = 0;             // steps enqueued since the last flush
  int spec = 2;                     // CG iterations enqueued per step
  std::map<int, hipGraphExec_t> graphs;
  // slots: (S_T0 r'z, S_T1 ||r||inf) and (S_T2, S_T3) alternate between iterations, S_T4 p'Mp, S_T5 ||b1||inf
  explicit Pcg(Engine &en) : e(en) {
    size_t n = e.n, m = e.m;
    xs.alloc(n); r.alloc(n); zz.alloc(n); p.alloc(n); w.alloc(n); b1.alloc(n); dinv.alloc(n); Mxs.alloc(n);
    t.alloc(m); u.alloc(m); Axs.alloc(m);
    xs.zero(e.stream);
    if (const char *ev = getenv("OSQP_AMD_PCG_EXTRAP")) extrapolate = atoi(ev) != 0;
    if (extrapolate) { xs0.alloc(n); Axs0.alloc(m); Mxs0.alloc(n); }
    precond();
    const char *ev = getenv("OSQP_AMD_PCG_ASYNC");
    // opt-in (OSQP_AMD_PCG_ASYNC=1): measured on rand-1e5 the empty kernels of the iterations enqueued beyond convergence
    // (~5 us each inside a graph, ~14 per CG iteration) cost more than the two host round trips they replace; it needs the
    // fused CG kernels (open items) to pay off
    async_on = !e.comm && !g_debug_sync && (ev && atoi(ev) == 1);
    if (async_on) {
      ctl.alloc(C_COUNT); ctl.zero(e.stream);
      dctl.alloc(2); dctl.zero(e.stream);
      HIP_CHECK(hipHostMalloc((void **)&h_ctl, sizeof(int) * C_COUNT));
    }
  }
  ~Pcg() override {
    drop_graphs();
    if (h_ctl) (void)hipHostFree(h_ctl);
  }
  void drop_graphs() {
    for (auto &kv : graphs) (void)hipGraphExecDestroy(kv.second);
    graphs.clear();
  }
  void invalidate() override { drop_graphs(); }
  int kind() const override { return 2; }
  double cg_iters() const override { return (double)total_iters; }

  void precond() { pcg_precond(e.At, e.Pf, e.full_m(e.rho.get()), e.st.sigma, dinv.get(), e.stream, e.n0); }

  // Av = A v ; out = (P + sigma I) v + A'(rho .* Av)
  void apply_M(const double *v, double *Av, double *out) {
    hipStream_t s = e.stream;
    const double *vg = e.full_n(v);  // sharded: the one n-vector exchange of the product ...
    if (e.m > 0) {
      spmv(e.A, vg, Av, nullptr, 0.0, 0.0, nullptr, s);
      vec_ew_prod(t.get(), e.rho.get(), Av, e.m, s);
    }
    spmv(e.Pf, vg, out, nullptr, 0.0, e.st.sigma, v, s);
    if (e.m > 0) spmv(e.At, e.full_m(t.get()), out, nullptr, 1.0, 0.0, nullptr, s);  // ... and the one m-vector exchange
  }

  int solve(double *xz, double cand) override {
    if (int rc = flush()) return rc;
    hipStream_t s = e.stream;
    const int n = e.n, m = e.m;
    double *slots = e.slots.get();
    // b1 = r_x + A'(rho .* r_z)
    if (m > 0) {
      vec_ew_prod(t.get(), e.rho.get(), xz + n, m, s);
      spmv(e.At, e.full_m(t.get()), b1.get(), nullptr, 0.0, 1.0, xz, s);
    } else {
      vec_copy(b1.get(), xz, n, s);
    }
    HIP_CHECK(hipMemsetAsync(slots + S_T0, 0, sizeof(double) * 6, s));
    reduce_absmax(b1.get(), nullptr, n, slots + S_T5, s);
    // carried products of the start vector
    if (!carried_valid || ++since_refresh >= kRefresh) {
      apply_M(xs.get(), Axs.get(), Mxs.get());
      carried_valid = true;
      since_refresh = 0;
    }
    // Start vector: the energy-optimal point on the line through the last two solutions (k_extrap_dots); the
    // carried products move along with it.  Off for the first solve after M changed or x~ was reset.
    if (extrapolate) {
      if (have_prev) {
        pcg_extrap_dots(n, xs.get(), xs0.get(), Mxs.get(), Mxs0.get(), b1.get(), e.partials.get(), slots + S_T2, slots + S_T3, s);
        e.combine_slots(S_T2, 2, 3u);
        pcg_extrapolate3(xs.get(), xs0.get(), Mxs.get(), Mxs0.get(), n, Axs.get(), Axs0.get(), m, slots + S_T2, slots + S_T3, s);
      } else {
        vec_copy(xs0.get(), xs.get(), n, s); vec_copy(Mxs0.get(), Mxs.get(), n, s);
        if (m > 0) vec_copy(Axs0.get(), Axs.get(), m, s);
        have_prev = true;
      }
    }
    // r = b1 - M x0 ; zz = dinv r ; p = zz
    pcg_init_residual(n, b1.get(), Mxs.get(), dinv.get(), r.get(), zz.get(), p.get(), e.partials.get(), slots + S_T0,
                      slots + S_T1, s);
    e.fetch_slots(S_T0, 6, 1u);  // r'z is a sum, the norms are maxima, the rest is zero
    const double bnorm = e.h_slots[S_T5];
    // tolerance rule (same statement as oracle/osqp_oracle.c pcg_tolerance)
    const double hi = 1e-2 * bnorm, lo = 1e-13 * bnorm + 1e-300;
    double tol = hi;
    if (cand >= 0.0) tol = cand;
    if (!(tol < hi)) tol = hi;
    if (tol < lo) tol = lo;
    double rn = e.h_slots[S_T1];
    int it = 0, cur = 0;  // cur: which pair holds the current r'z
    int status = 0;
    while (it < max_iter) {
      if (rn <= tol) break;
      if (rn != rn) { status = 5; break; }
      apply_M(p.get(), u.get(), w.get());
      double *rz = slots + S_T0 + 2 * cur, *rz_new = slots + S_T0 + 2 * (1 - cur), *pw = slots + S_T4;
      reduce_dot(p.get(), w.get(), n, e.partials.get(), pw, s);
      e.combine_slots(S_T4, 1, 1u);
      // alpha = rz / pw on the device: A x~ += alpha A p ; x~ += alpha p ; M x~ += alpha w ; r -= alpha w ; zz = dinv r
      vec_axpy2_dev(Mxs.get(), w.get(), n, Axs.get(), u.get(), m, rz, pw, s);
      pcg_update_xr(n, rz, pw, xs.get(), p.get(), r.get(), w.get(), dinv.get(), zz.get(), e.partials.get(), rz_new,
                    rz_new + 1, s);
      e.combine_slots(S_T0 + 2 * (1 - cur), 2, 1u);  // the new r'z (sum) and ||r||inf (max) in one exchange
      pcg_update_p(n, rz_new, rz, zz.get(), p.get(), s);
      e.read_slots(S_T0, 5);  // already combined
      if (!(e.h_slots[S_T4] > 0.0)) { status = 5; carried_valid = false; break; }  // p'Mp <= 0: M is not positive definite
      rn = e.h_slots[S_T1 + 2 * (1 - cur)];
      cur = 1 - cur;
      it++;
    }
    total_iters += it;
    vec_copy2(xz, xs.get(), n, xz + n, Axs.get(), m, s);  // x~ and z~ = A x~
    return status;
  }
  int update_rho() override { int rc = flush(); precond(); carried_valid = false; have_prev = false; return rc; }
  int update_matrices() override { int rc = flush(); precond(); carried_valid = false; have_prev = false; return rc; }
  void set_guess(const double *x) override { (void)flush(); vec_copy(xs.get(), x, e.n, e.stream); carried_valid = false; have_prev = false; }

  // ---------------------------------------------------------------- asynchronous form
  // One ADMM iteration, enqueued: right-hand side, start vector, `c` predicated CG iterations, update of (x, z, y).
  void enqueue_step(int c, bool hp, bool refresh) {
    hipStream_t s = e.stream;
    const int n = e.n, m = e.m;
    double *slots = e.slots.get(), *xz = e.xz.get();
    int *flags = ctl.get();
    {
      SkipScope on_stall(flags + C_STALL);
      admm_rhs(n, m, e.st.sigma, e.x.get(), e.q.get(), e.z.get(), e.rho_inv.get(), e.y.get(), xz, s);
      if (m > 0) {
        vec_ew_prod(t.get(), e.rho.get(), xz + n, m, s);
        spmv(e.At, t.get(), b1.get(), nullptr, 0.0, 1.0, xz, s);
      } else {
        vec_copy(b1.get(), xz, n, s);
      }
      fill_slots(slots + S_T0, 6, 0.0, s);
      reduce_absmax(b1.get(), nullptr, n, slots + S_T5, s);
      if (refresh) apply_M(xs.get(), Axs.get(), Mxs.get());
      if (extrapolate) {
        if (hp) {
          pcg_extrap_dots(n, xs.get(), xs0.get(), Mxs.get(), Mxs0.get(), b1.get(), e.partials.get(), slots + S_T2, slots + S_T3, s);
          pcg_extrapolate3(xs.get(), xs0.get(), Mxs.get(), Mxs0.get(), n, Axs.get(), Axs0.get(), m, slots + S_T2, slots + S_T3, s);
        } else {
          vec_copy(xs0.get(), xs.get(), n, s); vec_copy(Mxs0.get(), Mxs.get(), n, s);
          if (m > 0) vec_copy(Axs0.get(), Axs.get(), m, s);
        }
      }
      pcg_init_residual(n, b1.get(), Mxs.get(), dinv.get(), r.get(), zz.get(), p.get(), e.partials.get(), slots + S_T0, slots + S_T1, s);
    }
    OQ_LAUNCH(k_pcg_begin, dim3(1), dim3(1), 0, s, flags, (const double *)slots, (const double *)dctl.get(), dctl.get() + 1);
    for (int it = 0; it < c; it++) {
      const int cur = it & 1;
      {
        SkipScope on_done(flags + cur);
        apply_M(p.get(), u.get(), w.get());
        double *rz = slots + S_T0 + 2 * cur, *rz_new = slots + S_T0 + 2 * (1 - cur), *pw = slots + S_T4;
        reduce_dot(p.get(), w.get(), n, e.partials.get(), pw, s);
        vec_axpy2_dev(Mxs.get(), w.get(), n, Axs.get(), u.get(), m, rz, pw, s);
        pcg_update_xr(n, rz, pw, xs.get(), p.get(), r.get(), w.get(), dinv.get(), zz.get(), e.partials.get(), rz_new, rz_new + 1, s);
        pcg_update_p(n, rz_new, rz, zz.get(), p.get(), s);
      }
      OQ_LAUNCH(k_pcg_decide, dim3(1), dim3(1), 0, s, flags, (const double *)slots, (const double *)(dctl.get() + 1), cur);
    }
    OQ_LAUNCH(k_pcg_end, dim3(1), dim3(1), 0, s, flags, c & 1, c);
    {
      SkipScope on_stall(flags + C_STALL);
      vec_copy2(xz, xs.get(), n, xz + n, Axs.get(), m, s);  // x~ and z~ = A x~
      admm_update(n, m, e.st.alpha, xz, e.rho.get(), e.rho_inv.get(), e.l.get(), e.u.get(), e.x.get(), e.z.get(), e.y.get(), e.dx.get(),
                  e.dy.get(), s);
    }
  }

  bool fused_step() override {
    if (!async_on) return false;
    hipStream_t s = e.stream;
    // the candidate tolerance only changes at a residual evaluation, i.e. right after a flush
    double cand = -1.0;
    if (e.have_res) cand = e.lambda * std::sqrt(e.sc_pri * e.sc_dua);
    else if (e.have_seed) cand = e.lambda * e.g_seed;
    if (cand != cand_dev) {
      if (issued) { if (flush()) return true; }
      HIP_CHECK(hipMemcpyAsync(dctl.get(), &cand, sizeof(double), hipMemcpyHostToDevice, s));
      HIP_CHECK(hipStreamSynchronize(s));  // `cand` lives on this frame
      cand_dev = cand;
    }
    bool refresh = !carried_valid || since_refresh + 1 >= kRefresh;
    issue(spec, have_prev, refresh);
    return true;
  }
  void issue(int c, bool hp, bool refresh) {
    hipStream_t s = e.stream;
    const int key = c * 4 + (hp ? 2 : 0) + (refresh ? 1 : 0);
    static const bool use_graph = !(getenv("OSQP_AMD_GRAPH") && atoi(getenv("OSQP_AMD_GRAPH")) == 0);
    if (!use_graph) enqueue_step(c, hp, refresh);
    else {
      auto it = graphs.find(key);
      if (it == graphs.end()) {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        enqueue_step(c, hp, refresh);
        HIP_CHECK(hipStreamEndCapture(s, &graph));
        HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        (void)hipGraphDestroy(graph);
        it = graphs.emplace(key, exec).first;
      }
      HIP_CHECK(hipGraphLaunch(it->second, s));
    }
    issued++;
    if (refresh) { carried_valid = true; since_refresh = 0; } else since_refresh++;
    if (extrapolate) have_prev = true;
    if (issued >= 64) (void)flush_keep_error();
  }
  int deferred = 0;
  int flush_keep_error() { int rc = flush(); if (rc) deferred = rc; return rc; }

  // Wait for what was enqueued; finish a stalled solve on the host loop and re-issue the steps behind it.
  int flush() override {
    if (!async_on) return 0;
    if (deferred) { int rc = deferred; deferred = 0; issued = 0; return rc; }
    hipStream_t s = e.stream;
    int guard = 0;
    while (issued > 0) {
      HIP_CHECK(hipMemcpyAsync(h_ctl, ctl.get(), sizeof(int) * C_COUNT, hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
      const long long steps = h_ctl[C_STEPS], iters = h_ctl[C_ITERS];
      const bool stalled = h_ctl[C_STALL] != 0, err = h_ctl[C_ERR] != 0;
      const int stall_it = h_ctl[C_STALL_IT];
      total_iters += iters;
      HIP_CHECK(hipMemsetAsync(ctl.get(), 0, sizeof(int) * C_COUNT, s));
      if (err) { issued = 0; carried_valid = false; have_prev = false; return 5; }
      if (!stalled) {
        // speculation depth for the next window: one more than the mean needed
        const double mean = steps > 0 ? (double)iters / (double)steps : 0.0;
        int want = (int)std::ceil(mean) + 1;
        spec = std::min(kMaxSpec, std::max(1, std::max(want, spec - 1)));
        issued = 0;
        break;
      }
      // the step after `steps` completed ones ran out of enqueued iterations: its CG state is intact
      const long long behind = issued - steps - 1;
      int rc = finish_on_host(stall_it);
      if (rc) { issued = 0; return rc; }
      vec_copy2(e.xz.get(), xs.get(), e.n, e.xz.get() + e.n, Axs.get(), e.m, s);
      admm_update(e.n, e.m, e.st.alpha, e.xz.get(), e.rho.get(), e.rho_inv.get(), e.l.get(), e.u.get(), e.x.get(), e.z.get(), e.y.get(),
                  e.dx.get(), e.dy.get(), s);
      spec = std::min(kMaxSpec, spec + 1);
      issued = 0;
      since_refresh = (int)std::max<long long>(0, since_refresh - behind);  // the steps behind the stall fell through: their bookkeeping is redone
      for (long long i = 0; i < behind; i++) issue(spec, have_prev, since_refresh + 1 >= kRefresh);
      if (++guard > 1000) throw Error(6, "internal: the asynchronous CG path does not make progress");
    }
    return 0;
  }
  // the synchronous loop of solve(), entered after `it0` iterations of a solve whose state is on the device
  int finish_on_host(int it0) {
    hipStream_t s = e.stream;
    const int n = e.n, m = e.m;
    double *slots = e.slots.get();
    int cur = it0 & 1, it = it0;
    double tol = 0.0;
    HIP_CHECK(hipMemcpyAsync(&tol, dctl.get() + 1, sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    e.read_slots(S_T0, 5);
    double rn = e.h_slots[S_T1 + 2 * cur];
    while (it < max_iter) {
      if (rn <= tol) break;
      if (rn != rn) { carried_valid = false; return 5; }
      apply_M(p.get(), u.get(), w.get());
      double *rz = slots + S_T0 + 2 * cur, *rz_new = slots + S_T0 + 2 * (1 - cur), *pw = slots + S_T4;
      reduce_dot(p.get(), w.get(), n, e.partials.get(), pw, s);
      vec_axpy2_dev(Mxs.get(), w.get(), n, Axs.get(), u.get(), m, rz, pw, s);
      pcg_update_xr(n, rz, pw, xs.get(), p.get(), r.get(), w.get(), dinv.get(), zz.get(), e.partials.get(), rz_new, rz_new + 1, s);
      pcg_update_p(n, rz_new, rz, zz.get(), p.get(), s);
      e.read_slots(S_T0, 5);
      if (!(e.h_slots[S_T4] > 0.0)) { carried_valid = false; return 5; }
      rn = e.h_slots[S_T1 + 2 * (1 - cur)];
      cur = 1 - cur;
      it++;
    }
    total_iters += it - it0;
    return 0;
  }
  float time_solve(int reps) override {
    // one operator application (3 SpMV) as the unit of the indirect back-end
    hipEvent_t a, b;
    HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b));
    apply_M(p.get(), u.get(), w.get());
    HIP_CHECK(hipEventRecord(a, e.stream));
    for (int i = 0; i < reps; i++) apply_M(p.get(), u.get(), w.get());
    HIP_CHECK(hipEventRecord(b, e.stream));
    HIP_CHECK(hipEventSynchronize(b));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return ms / (float)reps;
  }
};

}  // namespace

std::unique_ptr<Linsys> make_pcg(Engine &e) { return std::unique_ptr<Linsys>(new Pcg(e)); }

}  // namespace oq
