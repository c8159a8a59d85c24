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
#include "devutil.hpp"

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

// ---------------------------------------------------------------------------------------------------------
// Fused kernels of the CG path (single device, all three matrices on the panel kernels).  A step of the host loop
// above is ~29 launches at one CG iteration; a short kernel costs ~4.5 us whichever way it is issued, so at mid size
// the launches, not the bytes, are the time.  Fused: 6 launches of start-up, 8 per CG iteration (6 of them the three
// products), 1 update.  Every sum is recombined from the same block partials by the same functions (devutil.hpp), so the
// iterates are those of the unfused sequence, bit for bit.
// Regions of the partials buffer (kReduceBlocks doubles each):
constexpr int R_DOT = 0;                  // p'w from the epilogue of the A' product  /  numerator of the extrapolation
constexpr int R_AUX = 1 * kReduceBlocks;  // denominator of the extrapolation
constexpr int R_RZ = 2 * kReduceBlocks;   // r'z
constexpr int R_RN = 3 * kReduceBlocks;   // max |r| per block

// xz_x = sigma x - q ;  t = rho (z - rho^-1 y)  (= rho .* the z-part of the ADMM right-hand side) ; scratch slots cleared
__global__ __launch_bounds__(kBlock) void k_pcg_rhs(int n, int m, double sigma, const double *__restrict__ x, const double *__restrict__ q,
                                                    const double *__restrict__ z, const double *__restrict__ rho,
                                                    const double *__restrict__ rho_inv, const double *__restrict__ y,
                                                    double *__restrict__ xz_x, double *__restrict__ t, double *__restrict__ slots,
                                                    const int *__restrict__ skip) {
  if (skip && *skip) return;
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < 6) slots[S_T0 + i] = 0.0;
  if (i < n) xz_x[i] = sigma * x[i] - q[i];
  else if (i < n + m) { const int j = i - n; const double rz = z[j] - rho_inv[j] * y[j]; t[j] = rho[j] * rz; }
}
// start vector (energy-optimal extrapolation from the last two solutions, or a plain copy) and initial residual
__global__ __launch_bounds__(kBlock) void k_pcg_start(int n, int m, int have_prev, double *__restrict__ xs, double *__restrict__ xs0,
                                                      double *__restrict__ Mxs, double *__restrict__ Mxs0, double *__restrict__ Axs,
                                                      double *__restrict__ Axs0, const double *__restrict__ b1,
                                                      const double *__restrict__ dinv, double *__restrict__ r, double *__restrict__ zz,
                                                      double *__restrict__ p, double *__restrict__ partials, const int *__restrict__ skip) {
  if (skip && *skip) return;
  double theta = 0.0;
  if (have_prev) {
    const double num = sum_partials(partials + R_DOT), den = sum_partials(partials + R_AUX);
    theta = den > 0.0 ? num / den : 0.0;
    theta = theta != theta ? 0.0 : fmin(fmax(theta, -1.0), 4.0);
  }
  double rz = 0.0, mx = 0.0;
  const int stride = gridDim.x * kBlock;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    double cur = xs[i], old = xs0[i];
    xs0[i] = cur;
    if (have_prev) xs[i] = cur + theta * (cur - old);
    cur = Mxs[i]; old = Mxs0[i];
    Mxs0[i] = cur;
    const double mnew = have_prev ? cur + theta * (cur - old) : cur;
    if (have_prev) Mxs[i] = mnew;
    const double ri = b1[i] - mnew, zi = dinv[i] * ri;
    r[i] = ri; zz[i] = zi; p[i] = zi;
    rz += ri * zi;
    mx = nanmax(mx, fabs(ri));
  }
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < m; j += stride) {
    const double cur = Axs[j], old = Axs0[j];
    Axs0[j] = cur;
    if (have_prev) Axs[j] = cur + theta * (cur - old);
  }
  rz = block_sum(rz);
  mx = block_max(mx);
  if (threadIdx.x == 0) { partials[R_RZ + blockIdx.x] = rz; partials[R_RN + blockIdx.x] = mx; }
}
// r'z and ||r||inf of the start vector into their slots; asynchronous form: also the tolerance and the first flag
// spec_done (host loop with a speculative first iteration): the verdict on the start vector, taken here with the host's
// arithmetic (tolerance from cand_v) and handed to both sides -- the flag the kernels of the iteration enqueued behind this
// kernel look at, and word S_COUNT + 1 of the published slots, which the host follows instead of deciding again.
__global__ __launch_bounds__(kBlock) void k_pcg_finish_start(const double *__restrict__ partials, double *__restrict__ slots, int *ctl,
                                                             const double *cand_p, double *tol_p, Publish pub, int *spec_done, double cand_v,
                                                             const int *__restrict__ skip) {
  if (skip && *skip) { if (ctl && threadIdx.x == 0) ctl[C_DONE0] = 1; return; }
  const double rz = sum_partials(partials + R_RZ), rn = max_partials(partials + R_RN);
  if (threadIdx.x != 0) return;
  slots[S_T0] = rz; slots[S_T1] = rn;
  int done = 0;
  if (spec_done) {
    const double bnorm = slots[S_T5];
    const double hi = 1e-2 * bnorm, lo = 1e-13 * bnorm + 1e-300;
    double tol = hi;
    if (cand_v >= 0.0) tol = cand_v;
    if (!(tol < hi)) tol = hi;
    if (tol < lo) tol = lo;
    done = (rn != rn || rn <= tol) ? 1 : 0;
    *spec_done = done;
  }
  if (pub.host_slots) {  // the host loop waits for exactly these three (and the verdict, when it launched ahead)
    pub.host_slots[S_T0] = rz; pub.host_slots[S_T1] = rn; pub.host_slots[S_T5] = slots[S_T5];
    pub.host_slots[S_COUNT + 1] = (double)done;
    __threadfence_system();
    *pub.host_seq = pub.seq;
    __threadfence_system();
  }
  if (!ctl) return;
  const double bnorm = slots[S_T5], cand = *cand_p;
  const double hi = 1e-2 * bnorm, lo = 1e-13 * bnorm + 1e-300;
  double tol = hi;
  if (cand >= 0.0) tol = cand;
  if (!(tol < hi)) tol = hi;
  if (tol < lo) tol = lo;
  *tol_p = tol;
  if (rn != rn) { ctl[C_ERR] = 1; ctl[C_STALL] = 1; ctl[C_DONE0] = 1; return; }
  ctl[C_DONE0] = rn <= tol ? 1 : 0;
}
// alpha = r'z / p'w ;  M x~ += alpha w ; A x~ += alpha u ; x~ += alpha p ; r -= alpha w ; zz = dinv r ; partials of r'z, max |r|
__global__ __launch_bounds__(kBlock) void k_pcg_step(int n, int m, const double *__restrict__ slot_rz, double *__restrict__ partials,
                                                     double *__restrict__ xs, const double *__restrict__ p, double *__restrict__ r,
                                                     const double *__restrict__ w, const double *__restrict__ dinv, double *__restrict__ zz,
                                                     double *__restrict__ Mxs, double *__restrict__ Axs, const double *__restrict__ u,
                                                     double *__restrict__ slot_pw, const int *__restrict__ skip) {
  if (skip && *skip) return;
  const double pw = sum_partials(partials + R_DOT);
  const double alpha = *slot_rz / pw;
  double rz = 0.0, mx = 0.0;
  const int stride = gridDim.x * kBlock;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const double wi = w[i];
    Mxs[i] += alpha * wi;
    xs[i] += alpha * p[i];
    const double ri = r[i] - alpha * wi, zi = dinv[i] * ri;
    r[i] = ri; zz[i] = zi;
    rz += ri * zi;
    mx = nanmax(mx, fabs(ri));
  }
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < m; j += stride) Axs[j] += alpha * u[j];
  rz = block_sum(rz);
  mx = block_max(mx);
  if (threadIdx.x == 0) {
    partials[R_RZ + blockIdx.x] = rz; partials[R_RN + blockIdx.x] = mx;
    if (blockIdx.x == 0) *slot_pw = pw;
  }
}
// beta = r'z_new / r'z ;  p = zz + beta p ;  the new r'z and ||r||inf into their slots; asynchronous form: the next flag
__global__ __launch_bounds__(kBlock) void k_pcg_next(int n, const double *__restrict__ partials, const double *__restrict__ slot_rz,
                                                     double *__restrict__ slot_rz_new, const double *__restrict__ zz, double *__restrict__ p,
                                                     const double *__restrict__ slot_pw, int *ctl, const double *tol_p, int cur,
                                                     Publish pub, const int *__restrict__ skip) {
  if (skip && *skip) { if (ctl && blockIdx.x == 0 && threadIdx.x == 0) ctl[1 - cur] = 1; return; }
  const double rz_new = sum_partials(partials + R_RZ), rn = max_partials(partials + R_RN);
  const double beta = rz_new / *slot_rz;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    slot_rz_new[0] = rz_new; slot_rz_new[1] = rn;
    if (pub.host_slots) {  // the host's stopping test needs ||r||inf and p'Mp; it does not wait for the update of p
      pub.host_slots[S_T0 + 2 * (1 - cur)] = rz_new; pub.host_slots[S_T1 + 2 * (1 - cur)] = rn; pub.host_slots[S_T4] = *slot_pw;
      __threadfence_system();
      *pub.host_seq = pub.seq;
      __threadfence_system();
    }
  }
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) p[i] = zz[i] + beta * p[i];
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  if (!ctl) return;
  ctl[C_ITERS] += 1;
  const double pw = *slot_pw;
  if (!(pw > 0.0) || rn != rn) { ctl[C_ERR] = 1; ctl[C_STALL] = 1; ctl[1 - cur] = 1; return; }  // M is not positive definite
  ctl[1 - cur] = rn <= *tol_p ? 1 : 0;
}
// ---------------------------------------------------------------------------------------------------------
// Single-reduction recurrence (Chronopoulos & Gear), OPT-IN: OSQP_AMD_PCG_SR=1, fused single-device host loop only.  The product
// of an iteration is taken of z = M^-1-preconditioned r (not of p): u = A z, w = M z, delta = z'w from the epilogue of the A'
// product; with gamma = r'z carried from the previous update, beta = gamma / gamma_old, alpha = gamma / (delta - beta gamma / alpha_old)
// and p = z + beta p, q = w + beta q (= M p), s = u + beta s (= A p) every vector of the iteration is updated in ONE kernel,
// whose last workgroup to finish adds up the block partials of the new gamma and ||r||inf and hands them to the host -- instead of
// k_pcg_step + k_pcg_next.  The partials cross workgroups inside a launch: stored and loaded at device scope (past the per-XCD
// L2s, as the counters of k_sn_tree in direct.hip), the ticket counter is back at zero when the launch ends (graph-safe).
// CPU statement: oracle/pcg.c with OSQP_ORACLE_PCG_SINGLE_REDUCTION=1 (tests/test_gpu_parity.py holds the two together).
constexpr int S_ALPHA = S_T5 + 1;  // alpha of the previous iteration (a free scratch slot)
__device__ __forceinline__ double sum_partials_dev(const double *partials) {
  double v = 0.0;
  for (int i = threadIdx.x; i < kReduceBlocks; i += kBlock) v += __hip_atomic_load(&partials[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return block_sum(v);
}
__device__ __forceinline__ double max_partials_dev(const double *partials) {
  double v = 0.0;
  for (int i = threadIdx.x; i < kReduceBlocks; i += kBlock) v = nanmax(v, __hip_atomic_load(&partials[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  return block_max(v);
}
__global__ __launch_bounds__(kBlock) void k_pcg_sr(int n, int m, int first, int cur, double *__restrict__ slots, double *partials,
                                                   double *__restrict__ xs, double *__restrict__ p, double *__restrict__ qv,
                                                   double *__restrict__ r, const double *__restrict__ w, const double *__restrict__ dinv,
                                                   double *__restrict__ zz, double *__restrict__ Mxs, double *__restrict__ Axs,
                                                   const double *__restrict__ u, double *__restrict__ sA, int *counter, Publish pub,
                                                   const int *__restrict__ skip) {
  if (skip && *skip) return;
  __shared__ int is_last;
  const double delta = sum_partials(partials + R_DOT);
  const double gamma = slots[S_T0 + 2 * cur], gamma_old = slots[S_T0 + 2 * (1 - cur)], alpha_old = slots[S_ALPHA];
  const double beta = first ? 0.0 : gamma / gamma_old;
  const double denom = first ? delta : delta - beta * gamma / alpha_old;
  const double alpha = gamma / denom;
  double rz = 0.0, mx = 0.0;
  const int stride = gridDim.x * kBlock;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const double zi = zz[i], wi = w[i];
    const double pi = first ? zi : zi + beta * p[i];
    const double qi = first ? wi : wi + beta * qv[i];
    p[i] = pi; qv[i] = qi;
    Mxs[i] += alpha * qi;
    xs[i] += alpha * pi;
    const double ri = r[i] - alpha * qi, zn = dinv[i] * ri;
    r[i] = ri; zz[i] = zn;
    rz += ri * zn;
    mx = nanmax(mx, fabs(ri));
  }
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < m; j += stride) {
    const double sj = first ? u[j] : u[j] + beta * sA[j];
    sA[j] = sj;
    Axs[j] += alpha * sj;
  }
  rz = block_sum(rz);
  mx = block_max(mx);
  if (threadIdx.x == 0) {
    __hip_atomic_store(&partials[R_RZ + blockIdx.x], rz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&partials[R_RN + blockIdx.x], mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the two stores have reached the device-coherent level
    const int ticket = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = ticket == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (!is_last) return;
  // every other workgroup has stored its partials (and read the slots it needs) before its ticket
  const double rz_new = sum_partials_dev(partials + R_RZ), rn = max_partials_dev(partials + R_RN);
  if (threadIdx.x != 0) return;
  __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  slots[S_T0 + 2 * (1 - cur)] = rz_new; slots[S_T1 + 2 * (1 - cur)] = rn;
  slots[S_T4] = denom; slots[S_ALPHA] = alpha;
  if (pub.host_slots) {
    pub.host_slots[S_T0 + 2 * (1 - cur)] = rz_new; pub.host_slots[S_T1 + 2 * (1 - cur)] = rn; pub.host_slots[S_T4] = denom;
    __threadfence_system();
    *pub.host_seq = pub.seq;
    __threadfence_system();
  }
}

// first stage of the extrapolation dot products alone (the consumer recombines the partials)
__global__ __launch_bounds__(kBlock) void k_extrap_partials(int n, const double *__restrict__ x1, const double *__restrict__ x0,
                                                            const double *__restrict__ Mx1, const double *__restrict__ Mx0,
                                                            const double *__restrict__ b, double *__restrict__ partials,
                                                            const int *__restrict__ skip) {
  if (skip && *skip) return;
  double num = 0.0, den = 0.0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const double e = x1[i] - x0[i], m1 = Mx1[i];
    num += e * (b[i] - m1);
    den += e * (m1 - Mx0[i]);
  }
  num = block_sum(num);
  den = block_sum(den);
  if (threadIdx.x == 0) { partials[R_DOT + blockIdx.x] = num; partials[R_AUX + blockIdx.x] = den; }
}

struct Pcg : Linsys {
  Engine &e;
  DevBuf<double> xs, r, zz, p, w, t, u, b1, dinv, Axs, Mxs, xs0, Axs0, Mxs0;
  DevBuf<double> qv, sA;            // single-reduction recurrence: M p and A p carried
  DevBuf<int> sr_counter;
  bool sr_on = false, sr_first = true;
  bool extrapolate = true, have_prev = false;
  long long total_iters = 0;
  int max_iter = 20000;
  double rel_tol = -1.0;            // >= 0: solve() stops at rel_tol * ||rhs||inf instead of the ADMM rule (polish_run_pcg)
  bool carried_valid = false;
  int since_refresh = 0;
  bool fused_on = false;            // the fused kernels apply (single device, panel kernels on all three matrices)
  bool pair_on = false;             // A p and P p of a CG iteration in one launch (mid-size matrices: neither fills the device alone)
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
    {
      const char *fv = getenv("OSQP_AMD_PCG_FUSED");
      fused_on = !e.comm && e.m > 0 && e.A.panel.active && e.At.panel.active && e.Pf.panel.active && !(fv && atoi(fv) == 0);
      pair_on = fused_on && spmv_pair_ok(e.A, e.Pf);
      const char *sv = getenv("OSQP_AMD_PCG_SR");
      sr_on = fused_on && extrapolate && sv && atoi(sv) == 1;
      if (sr_on) { qv.alloc(n); sA.alloc(std::max<size_t>(1, m)); sr_counter.alloc(1); sr_counter.zero(e.stream); }
    }
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
  void invalidate() override { drop_graphs(); rhs_ready = false; }
  int kind() const override { return 2; }
  double cg_iters() const override { return (double)total_iters; }

  void precond() { pcg_precond(e.At, e.Pf, e.full_m(e.rho.get()), e.st.sigma, dinv.get(), e.stream, e.n0); }

  // Av = A v ; out = (P + sigma I) v + A'(rho .* Av)
  void apply_M(const double *v, double *Av, double *out) {
    hipStream_t s = e.stream;
    const double *vg = e.full_n(v);  // sharded: the one n-vector exchange of the product ...
    const double *tg = nullptr;
    if (e.m > 0) {
      spmv(e.A, vg, Av, nullptr, 0.0, 0.0, nullptr, s);
      vec_ew_prod(t.get(), e.rho.get(), Av, e.m, s);
      tg = e.full_m_begin(t.get());  // ... and the one m-vector exchange, on its own stream while the P product runs
    }
    spmv(e.Pf, vg, out, nullptr, 0.0, e.st.sigma, v, s);
    if (e.m > 0) {
      e.full_m_end();
      spmv(e.At, tg, out, nullptr, 1.0, 0.0, nullptr, s);
    }
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
    if (rel_tol >= 0.0) tol = rel_tol * bnorm + 1e-300;  // polish: a plain relative tolerance on the reduced residual
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
  int update_rho() override { int rc = flush(); precond(); carried_valid = false; have_prev = false; rhs_ready = false; return rc; }
  int update_matrices() override { int rc = flush(); precond(); carried_valid = false; have_prev = false; rhs_ready = false; return rc; }
  void set_guess(const double *x) override { (void)flush(); vec_copy(xs.get(), x, e.n, e.stream); carried_valid = false; have_prev = false; rhs_ready = false; }

  // ---------------------------------------------------------------- asynchronous form
  // ---- the pieces of a step, fused kernels where they apply (fused_on), the launches of solve() otherwise -------------
  // right-hand side b1, its norm, the carried products, the start vector, the initial residual; r'z -> S_T0, ||r||inf -> S_T1
  Publish pub_next;  // set by the host loop before a launch whose kernel publishes its scalars itself
  // host loop, first CG iteration launched ahead of the verdict on the start vector (step_sync)
  DevBuf<int> spec_flag;
  bool spec_now = false;     // this start_solve hands its verdict to spec_flag
  double spec_cand = -1.0;
  double need_one = 1.0;     // running share of the steps that needed at least one CG iteration
  void start_solve(bool hp, bool refresh, bool async) {
    hipStream_t s = e.stream;
    const int n = e.n, m = e.m;
    double *slots = e.slots.get(), *xz = e.xz.get(), *part = e.partials.get();
    int *flags = async ? ctl.get() : nullptr;
    sr_first = true;
    if (fused_on) {
      if (!(rhs_ready && !async))
        OQ_LAUNCH(k_pcg_rhs, dim3(blocks_for((int64_t)n + m)), dim3(kBlock), 0, s, n, m, e.st.sigma, e.x.get(), e.q.get(), e.z.get(),
                  e.rho.get(), e.rho_inv.get(), e.y.get(), xz, t.get(), slots, g_skip);
      rhs_ready = false;
      SpmvExtra ex;
      ex.absmax_slot = slots + S_T5;
      const bool ext = extrapolate && hp;
      // the two sums of the extrapolation ride on the reduce of the right-hand side (same blocks, same order as
      // k_extrap_partials) unless the carried products are about to be refreshed
      static const bool fuse_e2 = !(getenv("OSQP_AMD_PCG_FUSE_EXTRAP") && atoi(getenv("OSQP_AMD_PCG_FUSE_EXTRAP")) == 0);
      const bool e2 = ext && !refresh && fuse_e2;
      if (e2) { ex.e2_x1 = xs.get(); ex.e2_x0 = xs0.get(); ex.e2_m1 = Mxs.get(); ex.e2_m0 = Mxs0.get(); ex.e2_partials = part + R_DOT; }
      spmv(e.At, t.get(), b1.get(), nullptr, 0.0, 1.0, xz, s, &ex);
      if (refresh) apply_M(xs.get(), Axs.get(), Mxs.get());
      if (ext && !e2) OQ_LAUNCH(k_extrap_partials, dim3(kReduceBlocks), dim3(kBlock), 0, s, n, xs.get(), xs0.get(), Mxs.get(), Mxs0.get(), b1.get(), part, g_skip);
      if (extrapolate)
        OQ_LAUNCH(k_pcg_start, dim3(kReduceBlocks), dim3(kBlock), 0, s, n, m, ext ? 1 : 0, xs.get(), xs0.get(), Mxs.get(), Mxs0.get(), Axs.get(),
                  Axs0.get(), b1.get(), dinv.get(), r.get(), zz.get(), p.get(), part, g_skip);
      else
        pcg_init_residual(n, b1.get(), Mxs.get(), dinv.get(), r.get(), zz.get(), p.get(), part + R_RZ, slots + S_T0, slots + S_T1, s);
      if (extrapolate)
        OQ_LAUNCH(k_pcg_finish_start, dim3(1), dim3(kBlock), 0, s, (const double *)part, slots, flags, (const double *)dctl.get(), dctl.get() + 1,
                  async ? Publish() : pub_next, (!async && spec_now) ? spec_flag.get() : (int *)nullptr, spec_cand, g_skip);
      else if (async)
        OQ_LAUNCH(k_pcg_begin, dim3(1), dim3(1), 0, s, flags, (const double *)slots, (const double *)dctl.get(), dctl.get() + 1);
      return;
    }
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
        pcg_extrap_dots(n, xs.get(), xs0.get(), Mxs.get(), Mxs0.get(), b1.get(), part, slots + S_T2, slots + S_T3, s);
        pcg_extrapolate3(xs.get(), xs0.get(), Mxs.get(), Mxs0.get(), n, Axs.get(), Axs0.get(), m, slots + S_T2, slots + S_T3, s);
      } else {
        vec_copy(xs0.get(), xs.get(), n, s); vec_copy(Mxs0.get(), Mxs.get(), n, s);
        if (m > 0) vec_copy(Axs0.get(), Axs.get(), m, s);
      }
    }
    pcg_init_residual(n, b1.get(), Mxs.get(), dinv.get(), r.get(), zz.get(), p.get(), part, slots + S_T0, slots + S_T1, s);
    if (async) {
      SkipScope none(nullptr);
      OQ_LAUNCH(k_pcg_begin, dim3(1), dim3(1), 0, s, flags, (const double *)slots, (const double *)dctl.get(), dctl.get() + 1);
    }
  }
  // one CG iteration; cur: which slot pair holds the current r'z
  void cg_iteration(int cur, bool async) {
    hipStream_t s = e.stream;
    const int n = e.n, m = e.m;
    double *slots = e.slots.get(), *part = e.partials.get();
    double *rz = slots + S_T0 + 2 * cur, *rz_new = slots + S_T0 + 2 * (1 - cur), *pw = slots + S_T4;
    int *flags = async ? ctl.get() : nullptr;
    if (fused_on && sr_on && !async) {  // single-reduction recurrence (k_pcg_sr): the product is taken of zz
      SpmvExtra exA;
      exA.y2 = t.get(); exA.s2 = e.rho.get();
      if (pair_on) spmv_pair(e.A, e.Pf, zz.get(), u.get(), &exA, w.get(), e.st.sigma, zz.get(), s);
      else {
        spmv(e.A, zz.get(), u.get(), nullptr, 0.0, 0.0, nullptr, s, &exA);
        spmv(e.Pf, zz.get(), w.get(), nullptr, 0.0, e.st.sigma, zz.get(), s);
      }
      SpmvExtra exT;
      exT.dotv = zz.get(); exT.dot_partials = part + R_DOT;
      spmv(e.At, t.get(), w.get(), nullptr, 1.0, 0.0, nullptr, s, &exT);
      OQ_LAUNCH(k_pcg_sr, dim3(kReduceBlocks), dim3(kBlock), 0, s, n, m, sr_first ? 1 : 0, cur, slots, part, xs.get(), p.get(), qv.get(), r.get(),
                (const double *)w.get(), (const double *)dinv.get(), zz.get(), Mxs.get(), Axs.get(), (const double *)u.get(), sA.get(),
                sr_counter.get(), pub_next, g_skip);
      sr_first = false;
      return;
    }
    if (fused_on) {
      SpmvExtra exA;
      exA.y2 = t.get(); exA.s2 = e.rho.get();                        // t = rho .* (A p) next to u = A p
      if (pair_on) spmv_pair(e.A, e.Pf, p.get(), u.get(), &exA, w.get(), e.st.sigma, p.get(), s);  // u = A p, w = P p + sigma p
      else {
        spmv(e.A, p.get(), u.get(), nullptr, 0.0, 0.0, nullptr, s, &exA);
        spmv(e.Pf, p.get(), w.get(), nullptr, 0.0, e.st.sigma, p.get(), s);
      }
      SpmvExtra exT;
      exT.dotv = p.get(); exT.dot_partials = part + R_DOT;            // p'w with the result of w += A' t
      spmv(e.At, t.get(), w.get(), nullptr, 1.0, 0.0, nullptr, s, &exT);
      OQ_LAUNCH(k_pcg_step, dim3(kReduceBlocks), dim3(kBlock), 0, s, n, m, (const double *)rz, part, xs.get(), (const double *)p.get(), r.get(),
                (const double *)w.get(), (const double *)dinv.get(), zz.get(), Mxs.get(), Axs.get(), (const double *)u.get(), pw, g_skip);
      OQ_LAUNCH(k_pcg_next, dim3(std::min(blocks_for(n), kReduceBlocks)), dim3(kBlock), 0, s, n, (const double *)part, (const double *)rz, rz_new,
                (const double *)zz.get(), p.get(), (const double *)pw, flags, (const double *)(dctl.get() + 1), cur, async ? Publish() : pub_next,
                g_skip);
      return;
    }
    apply_M(p.get(), u.get(), w.get());
    reduce_dot(p.get(), w.get(), n, part, pw, s);
    vec_axpy2_dev(Mxs.get(), w.get(), n, Axs.get(), u.get(), m, rz, pw, s);
    pcg_update_xr(n, rz, pw, xs.get(), p.get(), r.get(), w.get(), dinv.get(), zz.get(), part, rz_new, rz_new + 1, s);
    pcg_update_p(n, rz_new, rz, zz.get(), p.get(), s);
    if (async) {
      SkipScope none(nullptr);
      OQ_LAUNCH(k_pcg_decide, dim3(1), dim3(1), 0, s, flags, (const double *)slots, (const double *)(dctl.get() + 1), cur);
    }
  }
  // host loop: the update of step k leaves the vector part of step k + 1's right-hand side behind (k_admm_update_rhs), and
  // k_pcg_rhs is skipped -- as long as nothing looks at or changes the iterate in between: flush() (before every residual
  // evaluation), update_rho / update_matrices / set_guess (start of every solve) clear the mark
  bool rhs_ready = false;
  void finish_step(bool leave_rhs = false) {  // x~ = xs, z~ = A xs carried along: the ADMM update of (x, z, y)
    hipStream_t s = e.stream;
    if (leave_rhs) {
      admm_update2_rhs(e.n, e.m, e.st.alpha, xs.get(), Axs.get(), e.rho.get(), e.rho_inv.get(), e.l.get(), e.u.get(), e.x.get(), e.z.get(),
                       e.y.get(), e.dx.get(), e.dy.get(), e.st.sigma, e.q.get(), e.xz.get(), t.get(), e.slots.get(), s);
      rhs_ready = true;
      return;
    }
    rhs_ready = false;
    admm_update2(e.n, e.m, e.st.alpha, xs.get(), Axs.get(), e.rho.get(), e.rho_inv.get(), e.l.get(), e.u.get(), e.x.get(), e.z.get(), e.y.get(),
                 e.dx.get(), e.dy.get(), s);
  }

  // One ADMM iteration, enqueued: right-hand side, start vector, `c` predicated CG iterations, update of (x, z, y).
  void enqueue_step(int c, bool hp, bool refresh) {
    hipStream_t s = e.stream;
    int *flags = ctl.get();
    {
      SkipScope on_stall(flags + C_STALL);
      start_solve(hp, refresh, true);
    }
    for (int it = 0; it < c; it++) {
      SkipScope on_done(flags + (it & 1));
      cg_iteration(it & 1, true);
    }
    OQ_LAUNCH(k_pcg_end, dim3(1), dim3(1), 0, s, flags, c & 1, c);
    {
      SkipScope on_stall(flags + C_STALL);
      finish_step();
    }
  }

  // One ADMM iteration with the host in the CG loop (two scalar read-backs per CG iteration), fused kernels
  int step_sync() {
    double cand = -1.0;
    if (e.have_res) cand = e.lambda * std::sqrt(e.sc_pri * e.sc_dua);
    else if (e.have_seed) cand = e.lambda * e.g_seed;
    const bool refresh = !carried_valid || since_refresh + 1 >= kRefresh;
    const bool self_publish = fused_on && extrapolate;  // k_pcg_finish_start / k_pcg_next hand their scalars to the host themselves
    pub_next = self_publish ? e.begin_publish() : Publish();
    // Most steps need at least one CG iteration, and the verdict on the start vector costs the device ~12 us of idling while
    // the host reads it and launches: the first iteration goes in behind the start kernels at once, its kernels looking at
    // the verdict k_pcg_finish_start leaves on the device (they fall through when the start vector already passes).  The
    // host follows the device's verdict, so both sides always agree; the arithmetic is that of the plain loop.
    static const bool spec_allowed = !(getenv("OSQP_AMD_PCG_SPEC") && atoi(getenv("OSQP_AMD_PCG_SPEC")) == 0);
    const bool spec = spec_allowed && self_publish && pub_next.host_slots && max_iter > 0 && need_one >= 0.5;
    if (spec && spec_flag.n == 0) spec_flag.alloc(1);
    spec_now = spec; spec_cand = cand;
    start_solve(have_prev, refresh, false);
    spec_now = false;
    const Publish pub_start = pub_next;
    Publish pub_first;
    if (spec) {
      pub_first = pub_next = e.begin_publish();
      SkipScope on_verdict(spec_flag.get());
      cg_iteration(0, false);
    }
    if (refresh) { carried_valid = true; since_refresh = 0; } else since_refresh++;
    if (extrapolate) have_prev = true;
    if (pub_start.host_slots) e.wait_publish(pub_start); else e.read_slots(S_T0, 6);
    const double bnorm = e.h_slots[S_T5];
    const double hi = 1e-2 * bnorm, lo = 1e-13 * bnorm + 1e-300;
    double tol = hi;
    if (cand >= 0.0) tol = cand;
    if (!(tol < hi)) tol = hi;
    if (tol < lo) tol = lo;
    double rn = e.h_slots[S_T1];
    int it = 0, cur = 0, status = 0;
    bool ahead = spec;  // the iteration about to be looked at is already enqueued
    while (it < max_iter) {
      if (ahead) {
        if (e.h_slots[S_COUNT + 1] != 0.0) { if (rn != rn) status = 5; break; }  // the device's verdict: its kernels fell through
      } else {
        if (rn <= tol) break;
        if (rn != rn) { status = 5; break; }
        pub_next = fused_on ? e.begin_publish() : Publish();
        cg_iteration(cur, false);
      }
      const Publish &pub_it = ahead ? pub_first : pub_next;
      ahead = false;
      if (pub_it.host_slots) e.wait_publish(pub_it); else e.read_slots(S_T0, 5);
      if (!(e.h_slots[S_T4] > 0.0)) { status = 5; carried_valid = false; break; }
      rn = e.h_slots[S_T1 + 2 * (1 - cur)];
      cur = 1 - cur;
      it++;
    }
    need_one = 0.9 * need_one + (it > 0 ? 0.1 : 0.0);
    total_iters += it;
    static const bool fuse_rhs = !(getenv("OSQP_AMD_PCG_FUSE_RHS") && atoi(getenv("OSQP_AMD_PCG_FUSE_RHS")) == 0);
    finish_step(fuse_rhs && fused_on && status == 0);
    return status;
  }

  int fused_step() override {
    if (!async_on) return fused_on ? step_sync() : -1;
    hipStream_t s = e.stream;
    // the candidate tolerance only changes at a residual evaluation, i.e. right after a flush
    double cand = -1.0;
    if (e.have_res) cand = e.lambda * std::sqrt(e.sc_pri * e.sc_dua);
    else if (e.have_seed) cand = e.lambda * e.g_seed;
    if (cand != cand_dev) {
      if (issued) { if (int rc = flush()) return rc; }
      HIP_CHECK(hipMemcpyAsync(dctl.get(), &cand, sizeof(double), hipMemcpyHostToDevice, s));
      HIP_CHECK(hipStreamSynchronize(s));  // `cand` lives on this frame
      cand_dev = cand;
    }
    bool refresh = !carried_valid || since_refresh + 1 >= kRefresh;
    issue(spec, have_prev, refresh);
    return 0;
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
    rhs_ready = false;
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
      finish_step();
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
    int cur = it0 & 1, it = it0;
    double tol = 0.0;
    HIP_CHECK(hipMemcpyAsync(&tol, dctl.get() + 1, sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    e.read_slots(S_T0, 5);
    double rn = e.h_slots[S_T1 + 2 * cur];
    while (it < max_iter) {
      if (rn <= tol) break;
      if (rn != rn) { carried_valid = false; return 5; }
      pub_next = Publish();
      cg_iteration(cur, false);
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

// ---------------------------------------------------------------------------------------------------------
// Polish without a factorisation (SURVEY.md A.6 on a workspace that runs the indirect back-end -- compact ones have no CSR
// arrays to assemble a reduced KKT matrix from, the large ones no factor that fits).  With the active rows as a mask the
// regularised system [P + delta I, A_act'; A_act, -delta I] [x; y] = [r1; r2] is the operator this back-end already applies:
//   (P + delta I + A' diag(mask / delta) A) x = r1 + A' (mask / delta .* r2),   y = mask .* (A x - r2) / delta
// i.e. Pcg::solve with sigma = delta and rho = mask / delta (inactive rows: rho = 0, they drop out of every product).  The
// refinement steps of the reference run against the unregularised matrix exactly as in polish_run.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_polish_sets(int m, double delta, const double *__restrict__ z, const double *__restrict__ y,
                                                        const double *__restrict__ l, const double *__restrict__ u, double *__restrict__ rho,
                                                        double *__restrict__ bound) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= m) return;
  const bool low = z[i] - l[i] < -y[i], upp = u[i] - z[i] < y[i];
  rho[i] = (low || upp) ? 1.0 / delta : 0.0;
  bound[i] = low ? l[i] : (upp ? u[i] : 0.0);  // a row active on both sides keeps its lower bound, as the reduced matrix of polish_run does
}
// y = mask .* (Ax - r2) / delta  (mask = rho > 0)
__global__ __launch_bounds__(kBlock) void k_polish_dual(int m, double delta, const double *__restrict__ rho, const double *__restrict__ Ax,
                                                        const double *__restrict__ r2, double *__restrict__ y, int accumulate) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= m) return;
  const double v = rho[i] > 0.0 ? (Ax[i] - r2[i]) / delta : 0.0;
  y[i] = accumulate ? y[i] + v : v;
}
// r2 = mask .* (bound - A x)
__global__ __launch_bounds__(kBlock) void k_polish_r2(int m, const double *__restrict__ rho, const double *__restrict__ bound,
                                                      const double *__restrict__ Ax, double *__restrict__ r2) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < m) r2[i] = rho[i] > 0.0 ? bound[i] - Ax[i] : 0.0;
}
__global__ __launch_bounds__(kBlock) void k_polish_cone(int m, double *__restrict__ z, double *__restrict__ y, const double *__restrict__ l,
                                                        const double *__restrict__ u) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= m) return;
  const double sv = z[i] + y[i];
  const double zn = fmin(fmax(sv, l[i]), u[i]);
  z[i] = zn; y[i] = sv - zn;
}

int polish_run_pcg(Engine &e) {
  Pcg *P = dynamic_cast<Pcg *>(e.lin.get());
  if (!P) return -1;  // row-sharded workspaces included (round 4): every product takes its input through the all-gather of the rank blocks
  if (P->flush()) return -1;
  hipStream_t s = e.stream;
  const int n = e.n, m = e.m;
  OSQPInfo *info = e.ws->info;
  const double delta = e.st.delta;
  DevBuf<double> rho_keep(m ? m : 1), rho_pol(m ? m : 1), bound(m ? m : 1), xz(n + m), px(n), py(m ? m : 1), pz(m ? m : 1), tmp(n), r2(m ? m : 1);
  if (m > 0) {
    vec_copy(rho_keep.get(), e.rho.get(), m, s);
    OQ_LAUNCH(k_polish_sets, dim3(blocks_for(m)), dim3(kBlock), 0, s, m, delta, e.z.get(), e.y.get(), e.l.get(), e.u.get(), rho_pol.get(), bound.get());
    vec_copy(e.rho.get(), rho_pol.get(), m, s);
  }
  // Conjugate gradients cannot work on the operator at the reference's delta = 1e-6 (P + delta I + A_act' A_act / delta:
  // condition ~ 1e8 and worse -- 4000 iterations left the residuals where they started).  The inner regularisation is
  // therefore delta_in = max(delta, 1e-3) (condition ~ 1e5: a few hundred iterations) and the refinement against the
  // UNREGULARISED matrix -- the reference's own device for removing the effect of delta, here a method of multipliers on the
  // equality-constrained QP -- runs until the residual of that system has dropped by 1e-10 (at most kPolishRefine steps, at
  // least the caller's polish_refine_iter) instead of a fixed three times: same fixed point, reached by cheaper solves.
  const double delta_in = std::max(delta, 1e-3);
  constexpr int kPolishRefine = 40;
  if (m > 0) {  // rho of the active rows at the inner regularisation
    vec_scale(rho_pol.get(), delta / delta_in, m, s);
    vec_copy(e.rho.get(), rho_pol.get(), m, s);
  }
  const double sigma_keep = e.st.sigma, rel_keep = P->rel_tol;
  const bool extrap_keep = P->extrapolate;
  const int iter_keep = P->max_iter;
  e.st.sigma = delta_in;
  P->precond();
  P->extrapolate = false; P->carried_valid = false; P->have_prev = false; P->rhs_ready = false;
  P->rel_tol = 1e-6;
  P->max_iter = 2000;
  auto restore = [&]() {
    if (m > 0) vec_copy(e.rho.get(), rho_keep.get(), m, s);
    e.st.sigma = sigma_keep;
    P->precond();
    P->extrapolate = extrap_keep; P->rel_tol = rel_keep; P->max_iter = iter_keep;
    P->carried_valid = false; P->have_prev = false; P->rhs_ready = false;
    vec_copy(P->xs.get(), e.x.get(), n, s);
  };
  px.zero(s);
  if (m > 0) py.zero(s);
  double first_norm = -1.0;
  // whatever ends the refinement -- convergence, a refused solve, an exception out of a launch or a collective -- the workspace
  // gets its rho, sigma, preconditioner and CG settings back (advisor, round 4: a throw left the polish values behind and
  // later solves silently ran with rho = mask / delta, sigma = delta, rel_tol = 1e-6)
  try {
  for (int it = 0; it < kPolishRefine; it++) {
    // residual of the unregularised system at (x, y): r1 = -q - (P x + A' y), r2 = bound - A x on the active rows
    vec_copy(xz.get(), e.q.get(), n, s);
    vec_scale(xz.get(), -1.0, n, s);
    if (it > 0) {
      spmv(e.Pf, e.full_n(px.get()), tmp.get(), nullptr, 0.0, 0.0, nullptr, s);
      vec_axpy(xz.get(), -1.0, tmp.get(), n, s);
    }
    if (m > 0) {
      if (it > 0) {
        spmv(e.At, e.full_m(py.get()), tmp.get(), nullptr, 0.0, 0.0, nullptr, s);
        vec_axpy(xz.get(), -1.0, tmp.get(), n, s);
        spmv(e.A, e.full_n(px.get()), pz.get(), nullptr, 0.0, 0.0, nullptr, s);
      } else pz.zero(s);
      OQ_LAUNCH(k_polish_r2, dim3(blocks_for(m)), dim3(kBlock), 0, s, m, e.rho.get(), bound.get(), pz.get(), r2.get());
      vec_copy(xz.get() + n, r2.get(), m, s);
    }
    P->xs.zero(s);
    P->carried_valid = false; P->have_prev = false;
    if (P->solve(xz.get(), -1.0)) { restore(); return -1; }
    const double rn = e.h_slots[S_T5];  // ||r1 + A' rho r2||inf of the system just solved: how far (x, y) was from the fixed point
    vec_axpy(px.get(), 1.0, xz.get(), n, s);
    if (m > 0) OQ_LAUNCH(k_polish_dual, dim3(blocks_for(m)), dim3(kBlock), 0, s, m, delta_in, e.rho.get(), xz.get() + n, r2.get(), py.get(), 1);
    if (first_norm < 0.0) first_norm = rn;
    if (it >= e.st.polish_refine_iter && rn <= 1e-10 * first_norm) break;
  }
  } catch (...) {
    try { restore(); } catch (...) {}
    throw;
  }
  restore();
  // polished (x, z, y) and its residuals, as in polish_run
  if (m > 0) {
    spmv(e.A, e.full_n(px.get()), pz.get(), nullptr, 0.0, 0.0, nullptr, s);
    OQ_LAUNCH(k_polish_cone, dim3(blocks_for(m)), dim3(kBlock), 0, s, m, pz.get(), py.get(), e.l.get(), e.u.get());
  }
  {
    const double *xg = e.full_n(px.get());
    spmv(e.A, xg, e.Ax.get(), nullptr, 0.0, 0.0, nullptr, s);
    spmv(e.Pf, xg, e.Px_.get(), nullptr, 0.0, 0.0, nullptr, s);
  }
  if (m > 0) spmv(e.At, e.full_m(py.get()), e.Aty.get(), nullptr, 0.0, 0.0, nullptr, s);
  residual_norms(n, m, px.get(), pz.get(), e.Ax.get(), e.Px_.get(), e.Aty.get(), e.q.get(), e.Dinv.get(), e.Einv.get(), e.slots.get(),
                 e.partials.get(), s);
  e.fetch_slots(0, 16, (1u << S_XPX) | (1u << S_QX));
  const double *r = e.h_slots;
  const bool uns = e.st.scaling && !e.st.scaled_termination;
  const double pol_pri = m == 0 ? 0.0 : (uns ? r[S_PRI_UNS] : r[S_PRI]);
  const double pol_dua = uns ? e.cinv * r[S_DUA_UNS] : r[S_DUA];
  double pol_obj = 0.5 * r[S_XPX] + r[S_QX];
  if (e.st.scaling) pol_obj *= e.cinv;
  const bool ok = (pol_pri < info->pri_res && pol_dua < info->dua_res) || (pol_pri < info->pri_res && info->dua_res < 1e-10) ||
                  (pol_dua < info->dua_res && info->pri_res < 1e-10);
  if (!ok) {
    set_last_error("polish (iterative, indirect back-end): the polished point did not improve both residuals (pri " + std::to_string(pol_pri) +
                   " vs " + std::to_string(info->pri_res) + ", dua " + std::to_string(pol_dua) + " vs " + std::to_string(info->dua_res) + ")");
    return -1;
  }
  info->obj_val = pol_obj; info->pri_res = pol_pri; info->dua_res = pol_dua;
  vec_copy(e.x.get(), px.get(), n, s);
  if (m > 0) { vec_copy(e.z.get(), pz.get(), m, s); vec_copy(e.y.get(), py.get(), m, s); }
  P->set_guess(e.x.get());
  return 1;
}

}  // namespace oq
