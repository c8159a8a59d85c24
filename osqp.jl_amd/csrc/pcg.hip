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
#include "engine.hpp"

#include <cmath>

namespace oq {

namespace {

constexpr int kRefresh = 25;

struct Pcg : Linsys {
  Engine &e;
  DevBuf<double> xs, r, zz, p, w, t, u, b1, dinv, Axs, Mxs, xs0, Axs0, Mxs0;
  bool extrapolate = true, have_prev = false;
  long long total_iters = 0;
  int max_iter = 20000;
  bool carried_valid = false;
  int since_refresh = 0;
  // slots: (S_T0 r'z, S_T1 ||r||inf) and (S_T2, S_T3) alternate between iterations, S_T4 p'Mp, S_T5 ||b1||inf
  explicit Pcg(Engine &en) : e(en) {
    size_t n = e.n, m = e.m;
    xs.alloc(n); r.alloc(n); zz.alloc(n); p.alloc(n); w.alloc(n); b1.alloc(n); dinv.alloc(n); Mxs.alloc(n);
    t.alloc(m); u.alloc(m); Axs.alloc(m);
    xs.zero(e.stream);
    if (const char *ev = getenv("OSQP_AMD_PCG_EXTRAP")) extrapolate = atoi(ev) != 0;
    if (extrapolate) { xs0.alloc(n); Axs0.alloc(m); Mxs0.alloc(n); }
    precond();
  }
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
  int update_rho() override { precond(); carried_valid = false; have_prev = false; return 0; }
  int update_matrices() override { precond(); carried_valid = false; have_prev = false; return 0; }
  void set_guess(const double *x) override { vec_copy(xs.get(), x, e.n, e.stream); carried_valid = false; have_prev = false; }
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
