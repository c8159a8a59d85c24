// kernels.hpp -- launchers of the hand-written gfx950 kernels (kernels.hip).
// Rows K0-K10 of SURVEY.md section 8a; each launcher names its row.
#pragma once
#include <functional>
#include "common.hpp"

namespace oq {

// Predication of the kernels of the conjugate-gradient path.  While a SkipScope is alive, every launcher below passes its
// flag to the kernel, which returns at once when the flag is set: the host can enqueue CG iterations (or whole ADMM
// iterations) ahead of the convergence test that runs on the device, without a round trip per iteration (pcg.hip).
extern thread_local const int *g_skip;
struct SkipScope {
  const int *prev;
  explicit SkipScope(const int *flag) : prev(g_skip) { g_skip = flag; }
  ~SkipScope() { g_skip = prev; }
  SkipScope(const SkipScope &) = delete;
  SkipScope &operator=(const SkipScope &) = delete;
};
void fill_slots(double *slots, int count, double value, hipStream_t s);  // slots[0..count) = value, count <= 64 (predicated)

// ---------------- sparse structure (setup only) ----------------
// out[k] = column id of CSC entry k (binary search in colptr)
void expand_colptr(int cols, const int64_t *colptr, int64_t nnz, int *out, hipStream_t s);
// exclusive scan of counts[0..n) into out[0..n] (out[n] = total), 64-bit
void exclusive_scan(const int64_t *counts, int64_t *out, int64_t n, hipStream_t s);
// Build CSR (rowptr/col/src) from E coordinate entries; entries with erow < 0 are dropped.
// src[pos] = index e of the entry stored at pos.  Rows come out sorted by (col, e).
void csr_from_coo(int rows, int cols, int64_t E, const int *erow, const int *ecol, DevCsr &out,
                  DevBuf<int> &src, hipStream_t s);
void gather_values(int64_t nnz, const int *src, const double *in, double *out, int64_t modulo, hipStream_t s);
void invert_map(int64_t nnz, const int *src, int64_t lo, int64_t hi, int *k2pos, hipStream_t s);
void convert_i64_i32(int64_t n, const int64_t *in, int *out, hipStream_t s);
// choose lanes-per-row for the SpMV kernel from the mean row length
int pick_group(int rows, int64_t nnz);

// ---------------- K6 / K7: CSR SpMV ----------------
// Optional epilogue work of a product (panel kernels only: fused CG path, pcg.hip); all pointers may be null.
struct SpmvExtra {
  double *y2 = nullptr;               // y2[i] = s2[i] * (M x)[i]  -- the raw product, before rscale / beta / gamma
  const double *s2 = nullptr;
  const double *dotv = nullptr;       // dot_partials[block] = sum over the block's rows of dotv[i] * y[i] (the layout and
  double *dot_partials = nullptr;     //   order of reduce_dot's first stage: kReduceBlocks entries, unused ones zeroed)
  double *absmax_slot = nullptr;      // *absmax_slot = max(*absmax_slot, |y[i]|)  (zero it first)
  // first stage of the two sums behind the CG start vector (pcg.hip, k_extrap_partials), y being the right-hand side:
  // e2_partials[block] = sum (x1 - x0) (y - m1),  e2_partials[kReduceBlocks + block] = sum (x1 - x0) (m1 - m0); same blocks, same order
  const double *e2_x1 = nullptr, *e2_x0 = nullptr, *e2_m1 = nullptr, *e2_m0 = nullptr;
  double *e2_partials = nullptr;
};
// y[i] = (rscale ? rscale[i] : 1) * sum_k val[k] x[col[k]] + beta * y[i] + gamma * v[i]
void spmv(const DevCsr &M, const double *x, double *y, const double *rscale, double beta, double gamma,
          const double *v, hipStream_t s, const SpmvExtra *extra = nullptr);

// ya = Ma x (+ extras) and yb = Mb x + gamma_b vb in one product launch + one reduce launch (panel.hip; bit-identical to
// the two single products); spmv_pair_ok: both matrices run the LDS-staged panel kernel with the same panel width
bool spmv_pair_ok(const DevCsr &Ma, const DevCsr &Mb);
void spmv_pair(const DevCsr &Ma, const DevCsr &Mb, const double *x, double *ya, const SpmvExtra *extra_a, double *yb, double gamma_b,
               const double *vb, hipStream_t s);

// LDS-staged column-panel variant (panel.hip); spmv() dispatches to it when M.panel.active
bool panel_wanted(const DevCsr &M);
// slot (may be null): slot[k] = position of CSR entry k inside the sliced copy, recorded while the entries are placed
void panel_build(DevCsr &M, hipStream_t s, uint32_t *slot = nullptr, bool will_compact = false,
                 const std::function<void()> &after_cols = std::function<void()>());                 // structure + values from the CSR arrays
void panel_fill(DevCsr &M, bool with_cols, hipStream_t s, uint32_t *slot = nullptr);  // refresh the values after the CSR values changed
void spmv_panel(const DevCsr &M, const double *x, double *y, const double *rscale, double beta, double gamma,
                const double *v, hipStream_t s, const SpmvExtra *extra = nullptr);
// compact mode: the CSR column / value arrays released, every maintenance pass on the sliced-ELL copy (panel.hip)
bool panel_can_compact(const DevCsr &M);
void panel_compact(DevCsr &M);
void panel_slot_of_pos(const DevCsr &M, uint32_t *slot, hipStream_t s);  // before panel_compact
void panel_row_absmax(const DevCsr &M, double *out, bool accumulate, hipStream_t s);
void panel_scale(DevCsr &M, const double *r, const double *c, int order, double scalar, hipStream_t s, int row0);
void panel_diag(const DevCsr &M, double *diag, int row0, hipStream_t s);
void panel_scale_norm(DevCsr &M, const double *r, const double *c, int order, double pre, int row0, double *norm, hipStream_t s);
void spmv_panel_squared(const DevCsr &M, const double *x, double *y, double gamma, const double *v, hipStream_t s);

// ---------------- K0: Ruiz equilibration pieces ----------------
void csr_row_absmax(const DevCsr &M, double *out, bool accumulate, hipStream_t s);  // out[i] = max(|row i|) (or max with old)
// order 0: (v*r[row])*c[col]; 1: symmetric (v*r[min])*r[max]; 2: (v*c[col])*r[row]; then *scalar
// row0: global id of the first row when M is a row block (order 1 indexes c by global row and column)
// pre: a scalar applied to the value FIRST; norm (may be null): norm[i] = max |row i| of the result (one pass: the fused
// Ruiz iteration of Engine::scale_data)
void csr_scale_rows_cols(DevCsr &M, const double *r, const double *c, int order, double scalar, hipStream_t s, int row0 = 0,
                         double pre = 1.0, double *norm = nullptr);
void vec_limit_rsqrt(double *d, int n, hipStream_t s);  // d <- 1/sqrt(limit(d))
void vec_limit(double *d, int n, hipStream_t s);
// out = 1/sqrt(limit(max(sc * a, b))) (b may be null): the factor of a Ruiz iteration from the norms the last one left
void vec_ruiz_factor(double *out, double sc, const double *a, const double *b, int n, hipStream_t s);
void vec_ew_prod(double *out, const double *a, const double *b, int n, hipStream_t s);       // out = a.*b
void vec_ew_recip(double *out, const double *a, int n, hipStream_t s);                       // out = 1./a
void vec_scale(double *x, double a, int n, hipStream_t s);                                   // x *= a
void vec_set(double *x, double a, int n, hipStream_t s);
void vec_copy(double *dst, const double *src, int n, hipStream_t s);
void vec_copy2(double *d1, const double *s1, int n1, double *d2, const double *s2, int n2, hipStream_t s);  // two copies, one launch
void vec_scale_by_vec_scalar(double *x, const double *d, double a, int n, hipStream_t s);    // x = (x.*d)*a
void vec_axpy(double *y, double a, const double *x, int n, hipStream_t s);                   // y += a x
void vec_clamp(double *x, double lo, double hi, int n, hipStream_t s);

// ---------------- reductions ----------------
void zero_slots(double *slots, hipStream_t s);
void reduce_absmax(const double *x, const double *scale, int n, double *slot, hipStream_t s);  // slot = max(slot, |scale.*x|)
void reduce_sum(const double *x, int n, double *partials, double *slot, hipStream_t s);        // deterministic two-stage
void reduce_dot(const double *a, const double *b, int n, double *partials, double *slot, hipStream_t s);

// ---------------- K1: constraint classification + rho vector ----------------
// mode 0: set types from bounds and fill rho (setup); mode 1: re-classify, flag[0] set if any type changed;
// mode 2: keep types, refresh rho / rho_inv from the new scalar rho.
void rho_vec_update(int m, const double *l, const double *u, int *ctype, double *rho, double *rho_inv,
                    double rho_scalar, int mode, int *flag, hipStream_t s);

// ---------------- K5: ADMM vector updates ----------------
void admm_rhs(int n, int m, double sigma, const double *x_prev, const double *q, const double *z_prev,
              const double *rho_inv, const double *y, double *xz, hipStream_t s);
// in place on x, z (previous iterate in, new iterate out)
void admm_update(int n, int m, double alpha, const double *xz, const double *rho, const double *rho_inv, const double *l,
                 const double *u, double *x, double *z, double *y, double *delta_x, double *delta_y, hipStream_t s);
// the same with x~ and z~ in two separate arrays
void admm_update2(int n, int m, double alpha, const double *xt, const double *zt, const double *rho, const double *rho_inv,
                  const double *l, const double *u, double *x, double *z, double *y, double *delta_x, double *delta_y, hipStream_t s);
void admm_update2_rhs(int n, int m, double alpha, const double *xt, const double *zt, const double *rho, const double *rho_inv, const double *l,
                      const double *u, double *x, double *z, double *y, double *delta_x, double *delta_y, double sigma, const double *q,
                      double *xz_x, double *t, double *slots, hipStream_t s);

// ---------------- K8: residual norms + objective pieces ----------------
void residual_norms(int n, int m, const double *x, const double *z, const double *Ax, const double *Px,
                    const double *Aty, const double *q, const double *Dinv, const double *Einv, double *slots,
                    double *partials, hipStream_t s);

// ---------------- K10: infeasibility tests ----------------
// projects delta_y onto the polar of the recession cone of [l,u] (in place), then
// slots[S_T0] = ||E.*dy||inf (E may be null), slots[S_T1] = u'max(dy,0) + l'min(dy,0)
void prim_infeas_prep(int m, double *dy, const double *l, const double *u, const double *E, double *slots,
                      double *partials, hipStream_t s);
// slots[S_T2] = number of rows violating the dual-infeasibility sign conditions
void dual_infeas_rows(int m, const double *Adx, const double *Einv, const double *l, const double *u, double thr,
                      double *slots, hipStream_t s);

// ---------------- K9: PCG pieces ----------------
// dinv[j] = 1 / (sigma + Pdiag[j] + sum_i rho[i] A[i,j]^2) with At = CSR of A'
// (row blocks: rho is indexed by global constraint id, row0 = global id of the first row of the blocks)
void pcg_precond(const DevCsr &At, const DevCsr &Pf, const double *rho, double sigma, double *dinv, hipStream_t s,
                 int row0 = 0);
// r = b - w; zz = dinv.*r; p = zz; partials -> slot rz = r'zz ; slot rn = max(slot rn, ||r||inf) (zero it before)
void pcg_init_residual(int n, const double *b, const double *w, const double *dinv, double *r, double *zz, double *p,
                       double *partials, double *slot_rz, double *slot_rn, hipStream_t s);
// alpha = rz / pw (read from slots); x += alpha p; Ax += alpha t...; r -= alpha w; zz = dinv r; -> rz_new, ||r||inf
void pcg_update_xr(int n, const double *slot_rz, const double *slot_pw, double *x, const double *p, double *r,
                   const double *w, const double *dinv, double *zz, double *partials, double *slot_rz_new,
                   double *slot_rn, hipStream_t s);
// beta = rz_new / rz ; p = zz + beta p
void pcg_update_p(int n, const double *slot_rz_new, const double *slot_rz, const double *zz, double *p, hipStream_t s);
// row block [r0, r1) of M (pattern and values; panels are not carried over)
void csr_slice_rows(DevCsr &M, int r0, int r1, hipStream_t s);
// out[first + k] = sum (bit k of sum_mask set) or max over the `world` rank copies gathered[r * count + k]
void combine_rank_slots(const double *gathered, int world, int count, unsigned sum_mask, double *out, hipStream_t s);
// energy-optimal extrapolation of the CG start vector from the last two solutions (kernels.hip: k_extrap_dots)
void pcg_extrap_dots(int n, const double *x1, const double *x0, const double *Mx1, const double *Mx0, const double *b,
                     double *partials, double *slot_num, double *slot_den, hipStream_t s);
// v1 <- v1 + theta (v1 - v0), v0 <- old v1 for the pairs (x, M x, A x), theta = clamp(slot_num / slot_den) on the device
void pcg_extrapolate3(double *x1, double *x0, double *Mx1, double *Mx0, int n, double *Ax1, double *Ax0, int m,
                      const double *slot_num, const double *slot_den, hipStream_t s);  // the three pairs of a CG start at once
void vec_axpy2_dev(double *y1, const double *x1, int n1, double *y2, const double *x2, int n2, const double *slot_num,
                   const double *slot_den, hipStream_t s);  // two device-scalar axpys in one launch

}  // namespace oq
