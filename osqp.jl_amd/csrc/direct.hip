// direct.hip -- direct KKT back-end: numeric LDL' factorisation and sparse
// triangular solves on the device (rows K2, K3, K4 of SURVEY.md section 8a), and
// polish (row N1) on top of the same machinery.
//
//   K = [P + sigma I, A'; A, -diag(rho)^-1]  =  Pm' L D L' Pm      (quasi-definite: exactly n positive pivots)
//
// Host (symbolic.hip): ordering, elimination tree, pattern of L, level schedule.
// Pivots are numbered level by level (level = height in the elimination tree), so
// that level l is the contiguous index range [level_ptr[l], level_ptr[l+1]):
//   * numeric factorisation: by supernodes where the solves run by supernodes (multifrontal, fronts in LDS, one launch
//     per supernode level and size class: mfront.hpp, round 5); otherwise in dot-product form level by level, levels in
//     ascending order (a column only needs columns of lower levels);
//   * forward solve  L v = b: row-oriented over the CSR copy of L, ascending levels;
//   * backward solve L' w = D^-1 v: row-oriented over the CSC arrays, descending levels.
// Runs of narrow levels (the top of the tree) are chained inside one workgroup
// with barriers instead of one launch per level.
// Per solve the kernels stream L twice: 2 * (12 nnz(L) + 4 (N+1)) + ~40 N bytes.
#include <climits>
#include <cstring>
#include <cmath>

#include "engine.hpp"
#include "symbolic.hpp"
#include "mfront.hpp"
#include "mfront_big.hpp"

namespace oq {

namespace {

constexpr int kChainRows = 512;   // levels at most this wide are chained in one workgroup
constexpr int kChainThreads = 1024;

}  // namespace
}  // namespace oq

#include "direct_factor_kernels.hpp"
#include "direct_dense_kernels.hpp"
#include "direct_level_kernels.hpp"
#include "direct_sn_kernels.hpp"
#include "direct_sndense_kernels.hpp"

namespace oq {
namespace {

struct Step { int kind; int a, b, G; int U = 2, L = 64; };  // L: lanes per row inside an LDS chain  // kind 0: single level, rows [a,b), G lanes per row; 1: chain of levels [a,b),
                                         // G threads per row for the part of its rows that lies before the chain

// ------------------------------------------------------------------ factor object
int level_limit();
struct LdlFactor {
  Engine &e;
  Symbolic S;
  int N = 0, n = 0, mr = 0, nlev = 0;
  double sigma = 0, cconst = 0;
  DevBuf<int64_t> Lp, Rp, Rmap, PtoL, AtoL, Rsplit, Lsplit;
  DevBuf<int> Li, Lcol, Rj, perm, pinv, level_ptr, status;
  DevBuf<double> Lx, Rx, D, Dinv, bp, W, S0a, S0b, x2, gjT, gjW, gjC;  // gj*: pivot block, row panel and panel copy of the block sweeps
  int lD = 0, cD = 0, kD = 0;   // dense top block: levels [lD, nlev), pivots [cD, N), kD = N - cD (0: none)
  bool kD_dense = false;        // ... taken because it IS dense (choose_dense_top)
  DevBuf<int> schur_cols;       // the columns below the block with entries in its rows (k_dense_chunk); empty: the Schur complement entry by entry
  int schur_ncols = 0;
  DevBuf<double> dsP1, dsP2;    // shares of the symmetric dense product, per tile (k_dense_apply_sym)
  double *Sinv = nullptr;       // which of S0a / S0b holds -S0^-1 after the last factorisation
  int ldD = 0;                  // leading dimension of the dense block's array (kD, or kD padded to 64 for the block sweeps)
  std::vector<char> long_rows;  // per level: phase 2 of the factorisation through dense work rows (k_ldl_entries_w)
  size_t w_half = 0;            // doubles in one half of W
  std::vector<Step> fwd, bwd;
  long long factorizations = 0;
  // supernodal solves (k_sn_*): chosen when the level schedule is deep and the supernode graph is shallow
  bool sn = false;
  Supernodes T;
  DevBuf<int> sn_ptr, sn_piv, perm_s, pinv_s, sn_Fj, sn_Gi, sn_up, sn_waits, sn_pending, sn_ready;
  int *sn_fault = nullptr, *sn_fault_host = nullptr;  // mapped host memory: a wait inside k_sn_tree timed out
  ~LdlFactor() { if (sn_fault_host) (void)hipHostFree(sn_fault_host); }
  bool faulted() const {
    if (!sn_fault_host) return false;
    if (inject_fault) { inject_fault = false; *sn_fault_host = 1; }  // OSQP_AMD_SNODE_FAULT_TEST=1: the host side of a timed-out wait
    return *(volatile int *)sn_fault_host != 0;
  }
  mutable bool inject_fault = getenv("OSQP_AMD_SNODE_FAULT_TEST") && atoi(getenv("OSQP_AMD_SNODE_FAULT_TEST")) == 1;
  bool sn_tree = false;     // the levels from sn_tree_L0 on in one launch per direction (k_sn_tree) instead of one per level
  int sn_tree_threads = 1024;  // threads per supernode of that launch (512: twice the resident workgroups, one more level fits)
  int sn_tree_grid = 0;        // persistent form: workgroups of the launch (0: one per supernode, all resident)
  DevBuf<int> sn_ticket;       // [forward, backward] counters of the persistent form
  int sn_tree_L0 = 1;       // first level of that launch: the lowest one from which all supernodes above fit the device at once
  DevBuf<int64_t> sn_woff, sn_wmap, sn_Fp, sn_Fpos, sn_Gp, sn_Gpos, sn_Fsplit;
  DevBuf<double> sn_Wc, sn_Wr, sn_Fx, sn_Gx, sn_Dinv;
  std::vector<int> sn_lanes_f, sn_lanes_b;  // per level: lanes per row for the entries outside the blocks
  // multifrontal numeric factorisation on the supernode partition (mfront.hpp): one launch per supernode level and size
  // class instead of two per pivot level
  bool mf = false;
  DevBuf<int> mf_snof, mf_slot, mf_bsz, mf_chp, mf_chl, mf_list, mf_err;
  DevBuf<int64_t> mf_uoff, mf_reloff;
  DevBuf<uint16_t> mf_rel, mf_loc;
  DevBuf<double> mf_U;
  struct MfLaunch { int cls, off, count, fcap; int tile_off = 0, tile_count = 0, chunk_off = 0, chunk_count = 0; };  // cls kMfBig: the fronts beyond LDS (mfront_big.hpp)
  std::vector<MfLaunch> mf_launches;  // in level order
  DevBuf<int64_t> mf_poff;            // panel scratch of the big fronts
  DevBuf<double> mf_panel;
  DevBuf<int64_t> mf_ct0;
  DevBuf<int> mf_tiles, mf_chunks, mf_wide, mf_nwide;  // (chunks: 64-row pieces of the big fronts' panels, k_mfb_rows)
  std::vector<int> mfh_chunks;
  std::vector<int64_t> mfh_poff;
  std::vector<char> mf_is_big;
  std::vector<int> mfh_tiles;
  int mf_big_count = 0, mf_big_fmax = 0;
  int mf_fmax = 0;
  bool mf_ok = false;                 // the host plan exists (mf_plan): every front fits LDS
  // a dense top block OVER the supernode partition (direct_sndense_kernels.hpp): the supernodes of the levels [snd_L0, T.nlev)
  // = [snd_J0, T.count) = the slots [snd_q0, N), snd_K pivots (0: none), are not factorised by fronts and not solved level by
  // level: their Schur complement is inverted explicitly (S0a, the block sweeps) and a solve multiplies by it once
  int snd_L0 = -1, snd_J0 = -1, snd_q0 = 0, snd_K = 0;
  int snd_nb = 0, snd_nv = 0, snd_ntiles = 0;   // boundary children, those of them that hand a front vector up, tiles of the lower triangle
  bool snd_vec_mode = false;                     // the rows of D gather up to sn_Ftop and take the front vectors (the one-launch tree runs the top part)
  double snd_skipped = 0.0;                      // entries of the rows of D that point into D (never read by a solve)
  DevBuf<int> snd_bch, snd_bslot, snd_crange, snd_vptr, snd_tiles, snd_trow_ptr, snd_trow_list, snd_pend_ptr;
  DevBuf<int64_t> snd_boff, snd_pend_src, snd_Fd, snd_vsrc;
  DevBuf<int> snd_dpos, snd_dinv;                // slot - snd_q0 -> row / column of the dense array, and back
  bool lean_built = false;            // the index arrays of the factor were built on the device (lean_device_*)
  std::vector<int> mfh_bsz, mfh_snof, mfh_chp, mfh_chl, mfh_list;
  std::vector<int64_t> mfh_uoff, mfh_reloff;
  const int *vec_perm() const { return sn ? perm_s.get() : perm.get(); }   // order of the solve vector bp
  const int *vec_pinv() const { return sn ? pinv_s.get() : pinv.get(); }
  int solve_levels() const { return sn ? T.nlev : lD; }
  // multiply-adds of a numeric factorisation WITHOUT the dense top block (sum of squared column counts of the columns below
  // it): the block itself is inverted on the matrix cores at ~10 TFLOP/s and is bounded by dense_max()
  double flops_below_dense_block() const {
    if (!kD) return S.flops;
    double f = 0.0;
    for (int c = 0; c < cD; c++) { const double cc = (double)(S.Lp[c + 1] - S.Lp[c]); f += cc * cc; }
    return f;
  }

  LdlFactor(Engine &en, const std::vector<int> &row_map, int mr_, double sigma_, double cconst_, int64_t limit,
            double flops_limit = 0.0)
      : e(en), sigma(sigma_), cconst(cconst_) {
    e.fetch_host_pattern();
    e.setup_mark("  host pattern");
    // OSQP_AMD_FIRST_ORDERING=1: nested dissection at once (a caller who knows the problem is a long banded one saves the
    // min-degree analysis that would only establish that: ~half of the setup of the control-1e6 bench workload)
    int first_ordering = getenv("OSQP_AMD_FIRST_ORDERING") ? atoi(getenv("OSQP_AMD_FIRST_ORDERING")) : -1;  // read per setup: a process may set up problems of both kinds
    // unset (round 4): on a LARGE problem whose KKT graph is long -- a breadth-first level structure hundreds of levels
    // deep: banded, multi-stage, grid-like; random sparsity is ~log N deep -- nested dissection goes first without being
    // asked (the min-degree analysis of such a graph only finds the chain: 2 - 3 s at 2.7e6 nodes), and min-degree is the
    // second opinion when the dissection comes out deep
    bool nd_by_depth = false;
    // (round 6: "long" is a level structure of 100 levels or more -- 400 until then: a 50 x 50 x 50 grid is 150 deep, and its
    // minimum-degree analysis was 2.2 of its 3.8 s of setup only to be replaced by the dissection; random sparsity at this size is
    // 10 - 20 deep.  A dissection that comes out badly is still replaced below.  OSQP_AMD_ND_DEPTH overrides.)
    const int nd_depth = getenv("OSQP_AMD_ND_DEPTH") ? atoi(getenv("OSQP_AMD_ND_DEPTH")) : 100;
    if (first_ordering < 0 && e.hP.cols + mr_ >= 200000) nd_by_depth = kkt_graph_depth(e.hP, e.hA, row_map, mr_) >= nd_depth;
    if (first_ordering < 0) first_ordering = nd_by_depth ? 1 : 0;
    // Round 5: a large long problem (the class nested dissection goes first on) gets a LEAN analysis -- numbering, tree,
    // levels, counts and the unsorted rows of the pattern -- and, when the factor is a supernodal one with fronts that fit
    // LDS, everything else (CSC arrays of L, scatter maps, row / column lists of the supernodes) is built on the device from
    // the rows (lean_device): control-1e6 spent 1.5 of its 3 s of setup writing, sorting and uploading those arrays on the
    // host.  Any other outcome completes the analysis on the host (symbolic_complete: the same arrays as a full analysis).
    // OSQP_AMD_LEAN = 0 never, 1 whenever the row map is the identity (tests: the path on small problems).
    const int lean_mode = getenv("OSQP_AMD_LEAN") ? atoi(getenv("OSQP_AMD_LEAN")) : -1;
    bool identity = mr_ == e.m;
    for (int i = 0; identity && i < mr_; i++) identity = row_map[i] == i;
    const bool lean_try = lean_mode != 0 && (lean_mode == 1 || nd_by_depth) && identity;
    // the scatter maps of the whole KKT system (every row of A in it) are built on the device from the CSC arrays of L
    const bool device_maps = identity && !(getenv("OSQP_AMD_DEVICE_MAPS") && atoi(getenv("OSQP_AMD_DEVICE_MAPS")) == 0);
    S.no_host_maps = device_maps;
    symbolic_analyse(e.hP, e.hA, row_map, mr_, limit, flops_limit, first_ordering == 1 ? 1 : 0, S, lean_try);
    // (the depth that counts for a supernodal factor is that of the SUPERNODE graph -- up to 64 pivots a step -- which is not
    // known yet: a dissection is only given up here when even that bound is hopeless; grid 1000 x 1000: 3 500 pivot levels, 60
    // supernode levels)
    if (nd_by_depth && (S.too_large || (int)S.level_ptr.size() - 1 > 16 * level_limit())) {  // the dissection did not deliver: as before
      first_ordering = 0;
      S = Symbolic();
      S.no_host_maps = device_maps;
      symbolic_analyse(e.hP, e.hA, row_map, mr_, limit, flops_limit, 0, S);
    }
    if (S.lean && !S.too_large) {  // the decision a full analysis takes in choose_supernodes, from the counts
      N = S.N; n = S.n; mr = S.mr; nlev = (int)S.level_ptr.size() - 1;
      lD = nlev; cD = N; kD = 0;
      e.setup_mark("  lean analysis");
      decide_supernodes(true);
      // (the device build of the index arrays sorts 32-bit positions: a factor beyond 2^31 entries completes on the host, int64 throughout)
      if (!(sn && mf_ok) || S.nnzL >= (int64_t)2147483646) { sn = false; mf_ok = false; T = Supernodes(); symbolic_complete(S); }
    }
    e.setup_mark("  symbolic analysis");
    if (S.too_large) return;
    // A deep level schedule under min-degree (banded / multi-stage structure: the elimination tree is a chain) gets a
    // second analysis with nested dissection; the cheaper triangular solve by the model below wins.
    const bool try_nd = !(getenv("OSQP_AMD_ND") && atoi(getenv("OSQP_AMD_ND")) == 0);
    // (not when the depth is a dense trailing block -- a dense P: no ordering shortens that, and the block is inverted
    // explicitly anyway)
    int lD0 = 0, cD0 = 0, kD0 = 0;
    if (!S.lean) choose_dense_top(S, kChainRows, dense_max(), kDenseSparseMax, kDenseMin, lD0, cD0, kD0);
    if (!S.lean && try_nd && first_ordering != 1 && (kD0 ? lD0 : (int)S.level_ptr.size() - 1) > 400) {
      Symbolic S2;
      S2.no_host_maps = device_maps;
      symbolic_analyse(e.hP, e.hA, row_map, mr_, limit, flops_limit, 1, S2);
      if (!S2.too_large && solve_cost_us(S2) < 0.7 * solve_cost_us(S)) S = std::move(S2);
    }
    // A short level schedule is launch-bound: the tie-breaking variant of the same ordering (symbolic.hpp, ordering 2)
    // often folds it further (bound constraints: row - variable - row chains of height 2 become height 1).
    const bool try_fifo = !(getenv("OSQP_AMD_MD_FIFO") && atoi(getenv("OSQP_AMD_MD_FIFO")) == 0);
    if (!S.lean && try_fifo && (int)S.level_ptr.size() - 1 >= 3 && (int)S.level_ptr.size() - 1 <= 400) {
      Symbolic S3;
      S3.no_host_maps = device_maps;
      symbolic_analyse(e.hP, e.hA, row_map, mr_, limit, flops_limit, 2, S3);
      if (!S3.too_large && S3.nnzL <= S.nnzL + S.nnzL / 10 && solve_cost_us(S3) < 0.9 * solve_cost_us(S)) S = std::move(S3);
    }
    e.setup_mark("  other orderings");
    hipStream_t s = e.stream;
    N = S.N; n = S.n; mr = S.mr; nlev = (int)S.level_ptr.size() - 1;
    auto up64 = [&](DevBuf<int64_t> &d, const std::vector<int64_t> &h) { d.alloc(h.size()); d.upload(h.data(), h.size(), s); };
    auto up32 = [&](DevBuf<int> &d, const std::vector<int> &h) { d.alloc(h.size()); d.upload(h.data(), h.size(), s); };
    up32(perm, S.perm); up32(pinv, S.pinv); up32(level_ptr, S.level_ptr);
    const bool lean = S.lean;
    lean_built = lean;
    if (lean) { lean_device_pattern(); e.setup_mark("    scatter maps"); }
    else {
      up64(Lp, S.Lp); up64(Rp, S.Rp); up64(Rmap, S.Rmap);
      up32(Li, S.Li); up32(Rj, S.Rj);
      if (S.no_host_maps) device_scatter_maps();
      else { up64(PtoL, S.PtoL); up64(AtoL, S.AtoL); }
    }
    Lcol.alloc(std::max<size_t>(1, (size_t)S.nnzL));  // the column of every entry of L: what the entry kernels of the factorisation start from
    expand_colptr(N, Lp.get(), S.nnzL, Lcol.get(), s);
    Lx.alloc(S.nnzL); D.alloc(N); Dinv.alloc(N); bp.alloc(N); status.alloc(2);
    if (!lean) { choose_dense_block(); decide_supernodes(false); }
    if (sn) { lD = nlev; cD = N; kD = 0; supernodes_on_device(); }
    else Rx.alloc(S.nnzL);  // (the CSR copy of the values serves the level-scheduled solves only)
    if (sn) e.setup_mark("    supernodes on the device");
    build_mf();
    if (mf) e.setup_mark("    fronts");
    setup_top();
    setup_dense_top();
    {  // levels of few columns with long rows: work rows of N doubles each, at most 256 MB
      long_rows.assign(nlev, 0);
      size_t wmax = 0;
      for (int l = 0; l < (mf ? 0 : lD); l++) {
        const int c0 = S.level_ptr[l], c1 = S.level_ptr[l + 1], width = c1 - c0;
        if (width == 0 || S.Lp[c1] == S.Lp[c0]) continue;
        const double mean = (double)(S.Rp[c1] - S.Rp[c0]) / (double)width;
        if (mean >= long_row_mean() && (size_t)width * (size_t)N * sizeof(double) <= ((size_t)256 << 20)) {
          long_rows[l] = 1;
          wmax = std::max(wmax, (size_t)width * (size_t)N);
        }
      }
      if (kD) wmax = std::max(wmax, (size_t)dense_batch() * (size_t)N);
      w_half = wmax;
      if (wmax) { W.alloc(2 * wmax); W.zero(s); }  // two halves: the work rows of a level are cleared while the next level fills its own
      if (kD) {
        // from kDenseBlocked pivots on the inverse is formed in place by block sweeps on the matrix cores (k_gj_*), below
        // by single / double sweeps between two copies of the array
        const bool blocked = kD >= kDenseBlocked;
        ldD = blocked ? (kD + 63) / 64 * 64 : kD;
        S0a.alloc((size_t)ldD * ldD);
        if (blocked) { gjT.alloc(2 * kGjK * kGjK); gjW.alloc((size_t)kGjK * ldD); gjC.alloc((size_t)kGjK * ldD); }
        else S0b.alloc((size_t)kD * kD);
        schur_ncols = 0;
        if (blocked && !S.Li.empty() && !(getenv("OSQP_AMD_SCHUR_DENSE") && atoi(getenv("OSQP_AMD_SCHUR_DENSE")) == 0)) {
          // the rank-64 form of the Schur complement when L21 is not very sparse: a rank-64 update costs (ld / 64)^2 / 2 tiles
          // whatever its columns hold; entry by entry costs a 64-lane gather round per 64 entries of a row of L21
          std::vector<int> cols;
          int64_t entries = 0;
          for (int j = 0; j < cD; j++) {
            const int *beg = S.Li.data() + S.Lp[j], *end = S.Li.data() + S.Lp[j + 1];
            const int64_t in_block = end - std::lower_bound(beg, end, cD);
            if (in_block > 0) { cols.push_back(j); entries += in_block; }
          }
          if (!cols.empty() && (double)entries >= 0.005 * (double)cols.size() * (double)kD) {
            schur_ncols = (int)cols.size();
            schur_cols.alloc(cols.size());
            schur_cols.upload(cols.data(), cols.size(), s);
          }
        }
        x2.alloc(kD);
        // the symmetric form of the product (k_dense_apply_sym) for the blocked inverse; OSQP_AMD_DENSE_SYM=0: the full one
        if (kD >= kDenseBlocked && ldD % kDsT == 0 && !(getenv("OSQP_AMD_DENSE_SYM") && atoi(getenv("OSQP_AMD_DENSE_SYM")) == 0)) {
          const size_t nb = (size_t)ldD / kDsT;
          dsP1.alloc(nb * nb * kDsT); dsP2.alloc(nb * nb * kDsT);
        }
      }
    }
    e.sync();
    e.setup_mark("  factor pattern on the device");
    if (!sn) build_schedule();  // (supernodal solves have their own schedule: levels of the supernode graph)
    e.setup_mark("  solve schedule");
    if (lean) { S.lean_rows.reset(); S.lean_state.reset(); }
    // the big index arrays are only needed on the device from here on
    std::vector<int>().swap(S.Li); std::vector<int>().swap(S.Rj); std::vector<int64_t>().swap(S.Rmap);
    std::vector<int64_t>().swap(S.PtoL); std::vector<int64_t>().swap(S.AtoL);
  }

  // Supernodal solves when their modelled time (a launch per level; the slowest workgroup of each level walks its
  // entries outside the blocks with 256 threads, then its s x s block) is well below the level schedule's.
  // OSQP_AMD_SNODE = 0 never, 2 always (tests), otherwise by the model.
  void decide_supernodes(bool lean) {
    const int mode = getenv("OSQP_AMD_SNODE") ? atoi(getenv("OSQP_AMD_SNODE")) : 1;
    if (mode == 0 || N < 2) return;
    if (mode != 2 && nlev < 48) return;
    // the depth IS a dense block (a dense P, dense constraint rows): the few levels below it stay a level schedule (no
    // partition to build).  A block-sparse top chain taken as a block -- the separators of a dissection -- is no such case:
    // the supernodal form replaces it when its model is the cheaper one (round 5: with the one-level-structure dissection
    // the control problem with T = 400 has 44 levels below such a block and 5 supernode levels)
    if (mode != 2 && kD > 0 && lD < 48 && kD_dense) return;
    int smax = kSnMax;  // OSQP_AMD_SNODE_MAX: smaller supernodes (tests: many levels on small problems)
    if (const char *v = getenv("OSQP_AMD_SNODE_MAX")) smax = std::max(1, std::min(kSnMax, atoi(v)));
    build_supernodes(S, smax, T, false, lean);  // lean: the partition, list lengths as upper bounds from the counts
    sn = mode == 2 || supernodes_pay(S, T, kChainRows, lD, kD, kSnThreads);
    bool planned = false;
    if (!sn) {
      // (round 6) second look with the plan of the fronts: where it puts a dense top over the partition -- a 2-D / 3-D structure: the
      // long chains of large separators at the top, which the model above charges level by level, are ONE product -- the solve is
      // modelled with it; and a level schedule that would carry a dense block of thousands of pivots above a deep tree pays for
      // that block's Schur complement at every refactorisation (a 50 x 50 x 50 grid: 12 277 pivots, 0.9 s; by fronts: 22 ms)
      mf_ok = mf_plan();
      planned = true;
      if (mf_ok && snd_K > 0) {
        const double with_top = supernode_solve_cost_us(T, kSnThreads, snd_L0) + (double)snd_K * (double)snd_K * 4.0 / 4.0e6 + 30.0;
        sn = with_top < level_solve_cost_us(S, kChainRows, lD, kD) || kD >= 2048;
      }
    }
    if (!sn) { T = Supernodes(); mf_ok = false; snd_L0 = snd_J0 = -1; snd_q0 = 0; snd_K = 0; return; }
    if (!planned) mf_ok = mf_plan();
    if (!mf_ok && !lean) supernode_wmap(S, T);  // k_sn_invert gathers the blocks through it; the fronts invert theirs in place
  }

  // Workgroups of `threads` threads of the tree kernel that one compute unit holds at once: the occupancy query (the kernel takes
  // 92 - 97 registers: one workgroup of 1024 threads, two of 512; compiled for 64 registers -- __launch_bounds__(NT, 8) -- it
  // spills 50 of them and loses: grid 700 x 700 2 405 -> 2 170 it/s, control T = 800 8.6 -> 5.5 k, measured in round 6)
  static int tree_resident(const void *f, int threads) {
    int api = 0;
    HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, f, threads, 0));
    if (getenv("OSQP_AMD_SETUP_TRACE")) {
      hipFuncAttributes fa;
      if (hipFuncGetAttributes(&fa, f) == hipSuccess)
        fprintf(stderr, "[supernodes] tree kernel, %d threads: %d workgroups per compute unit (%zu B of LDS, %d registers)\n", threads, api, (size_t)fa.sharedSizeBytes, fa.numRegs);
      else (void)hipGetLastError();
    }
    return std::max(1, api);
  }
  // The device side of a supernodal factor: the lists of the entries outside the blocks (uploaded from a full analysis,
  // built by lean_device_lists otherwise), the counters of the one-launch tree, the arrays of the inverted blocks.
  void supernodes_on_device() {
    hipStream_t s = e.stream;
    auto up64 = [&](DevBuf<int64_t> &d, const std::vector<int64_t> &h) { d.alloc(std::max<size_t>(1, h.size())); d.upload(h.data(), h.size(), s); };
    auto up32 = [&](DevBuf<int> &d, const std::vector<int> &h) { d.alloc(std::max<size_t>(1, h.size())); d.upload(h.data(), h.size(), s); };
    const bool lean = S.lean;
    up32(sn_ptr, T.ptr); up32(sn_piv, T.piv);
    if (!lean) { up32(sn_Fj, T.Fj); up32(sn_Gi, T.Gi); }
    sn_tree = T.nlev > 2 && !(getenv("OSQP_AMD_SNODE_TREE") && atoi(getenv("OSQP_AMD_SNODE_TREE")) == 0);
    sn_tree_L0 = 1;
    if (sn_tree) {
      // every workgroup of the launch must be resident at once: then a waiting workgroup can never keep the one it waits
      // for off the device, whatever order the dispatcher picks (the blockIdx-order argument in the kernel's comment is
      // the second line of defence, the 200 ms flag the third).  Round 4: when the levels from 1 on are too many
      // supernodes for that (control-1e6: 7 000), the launch takes the TOP of the tree -- the levels from the lowest one
      // on from which everything above fits -- and the wide levels below it stay one plain launch each: those have the
      // parallelism to fill the device anyway, the narrow top is where a launch per level is all latency.
      int per_cu_f = 0, per_cu_b = 0, cus = 0, dev = 0;
      HIP_CHECK(hipGetDevice(&dev));
      HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
      per_cu_f = tree_resident((const void *)k_sn_tree<true, 1024>, 1024);
      per_cu_b = tree_resident((const void *)k_sn_tree<false, 1024>, 1024);
      if (const char *v = getenv("OSQP_AMD_SNODE_TREE_PER_CU")) per_cu_f = per_cu_b = std::max(1, atoi(v));  // (experiments)
      // Round 6: where fronts go through global memory (a 2-D / 3-D structure: many narrow levels of large supernodes) the launch may
      // hold up to TWICE what is resident at once.  That is safe for the reason the kernel's comment gives as its second line of
      // defence -- workgroups are dispatched in block order per XCD and a workgroup only waits on lower blocks, so the lowest
      // unfinished one is resident or next in line whatever the grid -- with the 200 ms flag and the per-level fallback behind it;
      // full residency (the first line) was a margin, and on these structures it costs two levels of the launch: grid 700 x 700
      // 2 405 -> 2 754 it/s, 1000 x 1000 1 036 -> 1 222, no restart in any run.  The control family (fronts in LDS, 512-thread
      // variant below) is unchanged by it and keeps the margin.  OSQP_AMD_SNODE_TREE_OVERSUB = 1 restores it everywhere.
      const int oversub = (mf_ok && mf_big_count > 0) ? (getenv("OSQP_AMD_SNODE_TREE_OVERSUB") ? std::max(1, atoi(getenv("OSQP_AMD_SNODE_TREE_OVERSUB"))) : 2) : 1;
      const long long cap = (long long)oversub * std::min(per_cu_f, per_cu_b) * cus;
      if (getenv("OSQP_AMD_SETUP_TRACE"))
        fprintf(stderr, "[supernodes] one-launch tree: %d / %d resident workgroups of 1024 threads per compute unit (forward / backward), %d compute units, launch of at most %lld\n", per_cu_f, per_cu_b, cus, cap);
      // (a dense top over the supernodes, direct_sndense_kernels.hpp, takes the last levels out of the launch)
      const int top_end = snd_K ? snd_L0 : T.nlev, j_end = snd_K ? snd_J0 : T.count;
      while (sn_tree_L0 < top_end && (long long)(j_end - T.lvl_ptr[sn_tree_L0]) > cap) sn_tree_L0++;
      // round 5: with 512 threads per supernode twice as many workgroups are resident; taken when that brings one more level
      // (or more) into the launch -- a plain launch per level and direction is ~50 us on control-1e6, a level inside the
      // launch a hand-over of a few microseconds
      sn_tree_threads = 1024;
      // (round 6: not where fronts go through global memory -- a 2-D structure: the supernodes at the top of such a tree carry
      // long rows and borders of hundreds of rows, and sixteen lanes a row instead of eight count for more than a level gained:
      // grid 700 x 700 597 -> 712 it/s, 1000 x 1000 321 -> 334; control-1e6, all fronts in LDS, keeps 512: 1 792 against 1 731)
      const bool long_rows = mf_ok && mf_big_count > 0;
      const bool try512 = getenv("OSQP_AMD_SNODE_TREE_512") ? atoi(getenv("OSQP_AMD_SNODE_TREE_512")) != 0 : !long_rows;
      if (sn_tree_L0 > 1 && try512) {
        int pf = 0, pb = 0;
        pf = tree_resident((const void *)k_sn_tree<true, 512>, 512);
        pb = tree_resident((const void *)k_sn_tree<false, 512>, 512);
        if (const char *v = getenv("OSQP_AMD_SNODE_TREE_PER_CU")) pf = pb = 2 * std::max(1, atoi(v));  // (experiments)
        long long cap2 = (long long)std::min(pf, pb) * cus;
        if (getenv("OSQP_AMD_SETUP_TRACE")) fprintf(stderr, "[supernodes] one-launch tree: %d / %d resident workgroups of 512 threads per compute unit\n", pf, pb);
        if (getenv("OSQP_AMD_SNODE_TREE_CAP")) cap2 = std::min<long long>(cap2, atoll(getenv("OSQP_AMD_SNODE_TREE_CAP")));  // (experiments: a later start)
        int L2 = 1;
        while (L2 < top_end && (long long)(j_end - T.lvl_ptr[L2]) > cap2) L2++;
        if (L2 < sn_tree_L0) { sn_tree_L0 = L2; sn_tree_threads = 512; }
        // persistent workgroups: every level above level 0 in the launch, whatever the count (k_sn_tree, ticket)
        // (measured on control-1e6: from level 1 on 1 165 -> 858 it/s -- the wide levels are throughput work that the plain
        // level kernels do better than 512-thread workgroups with device-scope loads; OSQP_AMD_SNODE_TREE_PERSIST=k takes the
        // levels from k on, default: off)
        sn_tree_grid = 0;
        int persist_from = 0;
        if (getenv("OSQP_AMD_SNODE_TREE_PERSIST")) persist_from = atoi(getenv("OSQP_AMD_SNODE_TREE_PERSIST"));
        // (It was the default for a while -- from the lowest level with at most three device-fuls of supernodes above, control-1e6
        // 1 169 -> 1 192 it/s -- until the level kernels took their entries flat (k_sn_level_f): a level of a thousand
        // supernodes is now cheaper as a plain launch than inside the launch.  Without it: control-1e6 1 704 -> 1 737 it/s,
        // control T = 8000 5.09 -> 5.56 k, T = 30 000 2.50 -> 2.67 k, T = 800 unchanged.)
        if (persist_from >= 1 && persist_from < sn_tree_L0) {
          sn_tree_L0 = persist_from; sn_tree_threads = 512;
          sn_tree_grid = (int)std::min<long long>(cap2, (long long)(j_end - T.lvl_ptr[sn_tree_L0]));
          sn_ticket.alloc(2);
        }
      }
      if (top_end - sn_tree_L0 < 2) sn_tree = false;  // a single level (or none) left: nothing to fuse
    }
    // the split of the forward rows: where the entries that point at the first level of the one-launch tree (level 1
    // without it) begin -- children below that level have finished in earlier launches (not waited for), entries that
    // point below it are read through the caches
    const int split_level = (sn_tree && sn_tree_L0 > 1) ? sn_tree_L0 : std::min(1, T.nlev);
    const int q_upper = split_level < T.nlev ? T.ptr[T.lvl_ptr[split_level]] : N;
    if (sn_tree && sn_tree_L0 > 1) {
      // the counters are relative to the first level of the launch
      const int J0 = T.lvl_ptr[sn_tree_L0];
      std::fill(T.waits.begin(), T.waits.end(), 0);
      for (int J = J0; J < T.count; J++)
        if (T.up[J] >= 0) T.waits[T.up[J]]++;
      if (!lean)
        for (int q = 0; q < N; q++)
          T.Fsplit[q] = T.Fp[q] + (std::lower_bound(T.Fj.begin() + T.Fp[q], T.Fj.begin() + T.Fp[q + 1], q_upper) - (T.Fj.begin() + T.Fp[q]));
    }
    up32(sn_up, T.up); up32(sn_waits, T.waits); up32(sn_pending, T.waits);
    sn_ready.alloc(T.count); sn_ready.zero(s);
    if (sn_tree) {
      HIP_CHECK(hipHostMalloc((void **)&sn_fault_host, sizeof(int), hipHostMallocMapped));
      *sn_fault_host = 0;
      HIP_CHECK(hipHostGetDevicePointer((void **)&sn_fault, sn_fault_host, 0));
    }
    up64(sn_woff, T.woff);
    if (lean) lean_device_lists(q_upper);
    else {
      if (!mf_ok) up64(sn_wmap, T.wmap);
      up64(sn_Fp, T.Fp); up64(sn_Fpos, T.Fpos); up64(sn_Gp, T.Gp); up64(sn_Gpos, T.Gpos); up64(sn_Fsplit, T.Fsplit);
    }
    sn_lanes_f.assign(T.nlev, 1); sn_lanes_b.assign(T.nlev, 1);
    for (int L = 0; L < T.nlev; L++) {
      int64_t ef = 0, eb = 0, rows = 0;
      for (int J = T.lvl_ptr[L]; J < T.lvl_ptr[L + 1]; J++) {
        const int q0 = T.ptr[J], q1 = T.ptr[J + 1];
        ef += T.Fp[q1] - T.Fp[q0]; eb += T.Gp[q1] - T.Gp[q0]; rows += q1 - q0;
      }
      sn_lanes_f[L] = pick((double)ef / (double)std::max<int64_t>(1, rows));
      sn_lanes_b[L] = pick((double)eb / (double)std::max<int64_t>(1, rows));
    }
    std::vector<int> ps(N), pis(N);
    for (int q = 0; q < N; q++) { ps[q] = S.perm[T.piv[q]]; pis[ps[q]] = q; }
    up32(perm_s, ps); up32(pinv_s, pis);
    sn_Wc.alloc(T.woff[T.count]); sn_Wr.alloc(T.woff[T.count]);
    sn_Fx.alloc(std::max<size_t>(1, (size_t)T.Fp[N])); sn_Gx.alloc(std::max<size_t>(1, (size_t)T.Gp[N])); sn_Dinv.alloc(N);
    e.sync();
    // only the shapes are needed on the host from here on
    std::vector<int64_t>().swap(T.wmap); std::vector<int64_t>().swap(T.Fpos); std::vector<int64_t>().swap(T.Gpos);
    std::vector<int>().swap(T.Fj); std::vector<int>().swap(T.Gi);
  }

  // ---- the top of the tree by front vectors (direct_sn_kernels.hpp, SnTop): the levels from which every supernode has a panel ----
  int sn_top_Jt = -1, sn_top_level = -1;
  DevBuf<int64_t> sn_Ftop;
  DevBuf<int> sn_tchp, sn_tchl;
  DevBuf<double> sn_uvec;
  SnTop sn_top_args() const {
    if (sn_top_Jt < 0 || !mf) return SnTop{-1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    return SnTop{sn_top_Jt, sn_Ftop.get(), sn_tchp.get(), sn_tchl.get(), mf_bsz.get(), mf_reloff.get(), mf_rel.get(), mf_poff.get(), mf_panel.get(), sn_uvec.get()};
  }
  void setup_top() {
    sn_top_Jt = -1; sn_top_level = -1;
    if (!mf || !sn_tree || mf_big_count == 0) return;
    if (getenv("OSQP_AMD_SNODE_TOP") && atoi(getenv("OSQP_AMD_SNODE_TOP")) == 0) return;  // A/B runs: rows gathered entry by entry everywhere
    // Under a dense top (direct_sndense_kernels.hpp) the long chains of large separators -- what the front vectors were built for --
    // are gone from the launch, and for the levels that remain the plain gathers are the faster form (grid 700 x 700: 2 083 -> 2 409
    // it/s, 1000 x 1000: 920 -> 1 036 without the vectors: a supernode's hand-over is two loops over its children with a barrier
    // each and a panel product by one workgroup): off unless asked for (OSQP_AMD_SNODE_TOP=1; the tests of the vector path do)
    if (snd_K && !(getenv("OSQP_AMD_SNODE_TOP") && atoi(getenv("OSQP_AMD_SNODE_TOP")) == 1)) return;
    // the lowest level, not below the first level of the one-launch tree, from which every supernode has a panel and a border
    // that fits the LDS the kernel has free (4096 doubles)
    int Lt = T.nlev;
    for (int L = T.nlev - 1; L >= std::max(1, sn_tree_L0); L--) {
      bool all = true;
      for (int J = T.lvl_ptr[L]; J < T.lvl_ptr[L + 1] && all; J++) all = mfh_poff[J + 1] > mfh_poff[J] && mfh_bsz[J] <= 4096;
      if (!all) break;
      Lt = L;
    }
    if (T.nlev - Lt < 2) return;  // a single level: nothing is handed over
    hipStream_t s = e.stream;
    const int Jt = T.lvl_ptr[Lt], q_top = T.ptr[Jt];
    std::vector<int> tchp(T.count - Jt + 1, 0), tchl;
    for (int J = Jt; J < T.count; J++) if (T.up[J] >= 0) tchp[T.up[J] - Jt + 1]++;
    for (int k = 0; k < T.count - Jt; k++) tchp[k + 1] += tchp[k];
    tchl.resize(tchp[T.count - Jt]);
    {
      std::vector<int> fill(tchp.begin(), tchp.end() - 1);
      for (int J = Jt; J < T.count; J++) if (T.up[J] >= 0) tchl[fill[T.up[J] - Jt]++] = J;  // ascending: the order of the sums
    }
    sn_tchp.alloc(tchp.size()); sn_tchp.upload(tchp.data(), tchp.size(), s);
    sn_tchl.alloc(std::max<size_t>(1, tchl.size())); sn_tchl.upload(tchl.data(), tchl.size(), s);
    sn_Ftop.alloc(N);
    OQ_LAUNCH(k_lean_split, dim3(blocks_for(N)), dim3(kBlock), 0, s, N, (const int64_t *)sn_Fp.get(), (const int *)sn_Fj.get(), q_top, sn_Ftop.get());
    sn_uvec.alloc(std::max<int64_t>(1, mfh_reloff[T.count]));
    sn_uvec.zero(s);
    e.sync();
    sn_top_Jt = Jt; sn_top_level = Lt;
    if (getenv("OSQP_AMD_SETUP_TRACE"))
      fprintf(stderr, "[supernodes] top part by front vectors: levels %d .. %d (%d supernodes from %d on; the one-launch tree starts at level %d)\n", Lt,
              T.nlev - 1, T.count - Jt, Jt, sn_tree_L0);
  }

  // ---- the dense top over the supernodes (direct_sndense_kernels.hpp): device arrays, numeric part, its step of a solve -----
  void setup_dense_top() {
    if (!snd_K) return;
    if (!mf) throw Error(6, "internal: a dense top over the supernodes without the multifrontal plan");
    hipStream_t s = e.stream;
    const int K = snd_K;
    std::vector<int> bch;
    std::vector<int64_t> boff{0};
    for (int J = 0; J < snd_J0; J++)
      if (T.up[J] >= snd_J0 && mfh_bsz[J] > 0) { bch.push_back(J); boff.push_back(boff.back() + mfh_bsz[J]); }
    snd_nb = (int)bch.size();
    auto up32 = [&](DevBuf<int> &d, const std::vector<int> &h) { d.alloc(std::max<size_t>(1, h.size())); d.upload(h.data(), h.size(), s); };
    auto up64 = [&](DevBuf<int64_t> &d, const std::vector<int64_t> &h) { d.alloc(std::max<size_t>(1, h.size())); d.upload(h.data(), h.size(), s); };
    // the order of the pivots inside the dense array: the supernodes of D in postorder of their tree (children ascending), so that
    // the segments of a separator -- a chain -- and then whole subtrees are contiguous; OSQP_AMD_SN_DENSE_ORDER=0: slot order
    std::vector<int> dpos((size_t)K), dinv((size_t)K);
    {
      const int nd = T.count - snd_J0;
      std::vector<int> order;
      order.reserve(nd);
      if (getenv("OSQP_AMD_SN_DENSE_ORDER") && atoi(getenv("OSQP_AMD_SN_DENSE_ORDER")) == 0) { for (int J = snd_J0; J < T.count; J++) order.push_back(J); }
      else {
        std::vector<int> chp((size_t)nd + 1, 0), chl, stack, next((size_t)nd, 0);
        for (int J = snd_J0; J < T.count; J++) if (T.up[J] >= snd_J0) chp[(size_t)(T.up[J] - snd_J0) + 1]++;
        for (int k = 0; k < nd; k++) chp[k + 1] += chp[k];
        chl.resize((size_t)chp[nd]);
        { std::vector<int> f(chp.begin(), chp.end() - 1); for (int J = snd_J0; J < T.count; J++) if (T.up[J] >= snd_J0) chl[(size_t)f[T.up[J] - snd_J0]++] = J; }
        for (int R = snd_J0; R < T.count; R++) {
          if (T.up[R] >= snd_J0) continue;  // (roots of the forest of D: up = -1)
          stack.push_back(R);
          while (!stack.empty()) {
            const int J = stack.back(), k = J - snd_J0;
            if (next[k] < chp[k + 1] - chp[k]) stack.push_back(chl[(size_t)chp[k] + next[k]++]);
            else { order.push_back(J); stack.pop_back(); }
          }
        }
      }
      if ((int)order.size() != nd) throw Error(6, "internal: the supernodes of the dense top are not a forest");
      int pos = 0;
      for (int J : order) for (int q = T.ptr[J]; q < T.ptr[J + 1]; q++) { dpos[(size_t)(q - snd_q0)] = pos; dinv[(size_t)pos] = q - snd_q0; pos++; }
    }
    up32(snd_dpos, dpos); up32(snd_dinv, dinv);
    up32(snd_bch, bch); up64(snd_boff, boff);
    std::vector<int> hs((size_t)boff.back());
    snd_bslot.alloc(std::max<size_t>(1, hs.size()));
    if (snd_nb) {
      mf_err.zero(s);
      OQ_LAUNCH(k_snd_slots, dim3(snd_nb), dim3(256), 0, s, (const int *)snd_bch.get(), (const int *)sn_ptr.get(), (const int *)sn_piv.get(),
                (const int64_t *)Lp.get(), (const int *)Li.get(), (const int *)mf_slot.get(), snd_q0, (const int *)snd_dpos.get(), (const int64_t *)snd_boff.get(), snd_bslot.get(), mf_err.get());
      int err = 0;
      mf_err.download(&err, 1, s);
      snd_bslot.download(hs.data(), hs.size(), s);
      e.sync();
      if (err) throw Error(6, "internal: the dense top over the supernodes is not closed towards the root");
    }
    // per child: its range of slots; the one-row children (a pendant constraint row adds one number to one diagonal entry) as a
    // list per slot; per tile row of 64 slots the other children that have a border row there -- all ascending by child
    ldD = (K + 63) / 64 * 64;
    const int nt = ldD / 64;
    std::vector<int> crange(2 * (size_t)std::max(1, snd_nb)), pend_ptr((size_t)K + 1, 0), trow_ptr((size_t)nt + 1, 0), trow_list, tiles;
    std::vector<int64_t> pend_src;
    for (int k = 0; k < snd_nb; k++)
      if (mfh_bsz[bch[k]] == 1) pend_ptr[(size_t)hs[boff[k]] + 1]++;
    for (int c = 0; c < K; c++) pend_ptr[c + 1] += pend_ptr[c];
    pend_src.resize((size_t)pend_ptr[K]);
    {
      std::vector<int> fill(pend_ptr.begin(), pend_ptr.end() - 1), seen((size_t)nt, -1);
      std::vector<std::vector<int>> rows((size_t)nt);
      for (int k = 0; k < snd_nb; k++) {
        const int J = bch[k], b = mfh_bsz[J];
        if (b == 1) { pend_src[fill[hs[boff[k]]]++] = mfh_uoff[J]; crange[2 * k] = INT_MAX; crange[2 * k + 1] = -1; continue; }
        int mn = INT_MAX, mx = -1;
        for (int i = 0; i < b; i++) {
          const int r = hs[boff[k] + i];
          mn = std::min(mn, r); mx = std::max(mx, r);
          if (seen[r / 64] != k) { seen[r / 64] = k; rows[r / 64].push_back(k); }
        }
        crange[2 * k] = mn; crange[2 * k + 1] = mx;
      }
      for (int t = 0; t < nt; t++) { trow_ptr[t + 1] = trow_ptr[t] + (int)rows[t].size(); trow_list.insert(trow_list.end(), rows[t].begin(), rows[t].end()); }
    }
    for (int ti = 0; ti < nt; ti++)
      for (int tj = 0; tj <= ti; tj++) { tiles.push_back(ti); tiles.push_back(tj); }
    snd_ntiles = (int)(tiles.size() / 2);
    {  // block pattern of S and of everything its elimination fills: the cliques of the fronts of the supernodes of D (their own
       // slots and their border rows; a boundary child's border lies inside its parent's front) -- what gj_symbolic starts from
      std::vector<int> dl;
      std::vector<int64_t> doff{0};
      for (int J = snd_J0; J < T.count; J++) if (mfh_bsz[J] > 0) { dl.push_back(J); doff.push_back(doff.back() + mfh_bsz[J]); }
      std::vector<int> ds((size_t)doff.back());
      if (!dl.empty()) {
        DevBuf<int> d_dl, d_ds(std::max<size_t>(1, ds.size()));
        DevBuf<int64_t> d_doff;
        up32(d_dl, dl); up64(d_doff, doff);
        mf_err.zero(s);
        OQ_LAUNCH(k_snd_slots, dim3((int)dl.size()), dim3(256), 0, s, (const int *)d_dl.get(), (const int *)sn_ptr.get(), (const int *)sn_piv.get(),
                  (const int64_t *)Lp.get(), (const int *)Li.get(), (const int *)mf_slot.get(), snd_q0, (const int *)snd_dpos.get(), (const int64_t *)d_doff.get(), d_ds.get(), mf_err.get());
        int err = 0;
        mf_err.download(&err, 1, s);
        d_ds.download(ds.data(), ds.size(), s);
        e.sync();
        if (err) throw Error(6, "internal: a front of the dense top reaches below it");
      }
      std::vector<char> pat((size_t)nt * nt, 0);
      std::vector<int> blk;
      size_t di = 0;
      for (int J = snd_J0; J < T.count; J++) {
        blk.clear();
        for (int q = T.ptr[J]; q < T.ptr[J + 1]; q++) blk.push_back(dpos[(size_t)(q - snd_q0)] / 64);
        if (mfh_bsz[J] > 0) { for (int64_t i = doff[di]; i < doff[di + 1]; i++) blk.push_back(ds[(size_t)i] / 64); di++; }
        std::sort(blk.begin(), blk.end());
        blk.erase(std::unique(blk.begin(), blk.end()), blk.end());
        for (int a : blk) for (int b : blk) pat[(size_t)a * nt + b] = 1;
      }
      gj_symbolic(pat, nt, s);
    }
    up32(snd_crange, crange); up32(snd_pend_ptr, pend_ptr); up64(snd_pend_src, pend_src);
    up32(snd_trow_ptr, trow_ptr); up32(snd_trow_list, trow_list); up32(snd_tiles, tiles);
    // the boundary children inside the part of the tree that hands front vectors up (setup_top)
    std::vector<int> vch;
    if (sn_top_Jt >= 0 && sn_top_Jt < snd_J0)
      for (int k = 0; k < snd_nb; k++) if (bch[k] >= sn_top_Jt) vch.push_back(k);
    snd_nv = (int)vch.size();
    {  // per row of the dense array: where its entries of those vectors sit in uvec, children ascending
      std::vector<int> vptr((size_t)K + 1, 0);
      for (int k : vch) for (int i = 0; i < mfh_bsz[bch[k]]; i++) vptr[(size_t)hs[boff[k] + i] + 1]++;
      for (int a = 0; a < K; a++) vptr[a + 1] += vptr[a];
      std::vector<int64_t> vsrc((size_t)vptr[K]);
      std::vector<int> fill(vptr.begin(), vptr.end() - 1);
      for (int k : vch) for (int i = 0; i < mfh_bsz[bch[k]]; i++) vsrc[(size_t)fill[hs[boff[k] + i]]++] = mfh_reloff[bch[k]] + i;
      up32(snd_vptr, vptr); up64(snd_vsrc, vsrc);
    }
    // where the entries of a forward row that point into D begin (what a row of D gathers when no front vectors arrive)
    snd_Fd.alloc((size_t)N);
    OQ_LAUNCH(k_lean_split, dim3(blocks_for(N)), dim3(kBlock), 0, s, N, (const int64_t *)sn_Fp.get(), (const int *)sn_Fj.get(), snd_q0, snd_Fd.get());
    {
      std::vector<int64_t> fd((size_t)K);
      HIP_CHECK(hipMemcpyAsync(fd.data(), snd_Fd.get() + snd_q0, (size_t)K * sizeof(int64_t), hipMemcpyDeviceToHost, s));
      e.sync();
      snd_skipped = 0.0;
      for (int r = 0; r < K; r++) snd_skipped += (double)(T.Fp[(size_t)snd_q0 + r + 1] - fd[r]);
    }
    S0a.alloc((size_t)ldD * ldD);
    gjT.alloc(2 * kGjK * kGjK); gjW.alloc((size_t)kGjK * ldD); gjC.alloc((size_t)kGjK * ldD);
    x2.alloc((size_t)K);
    const size_t nbk = (size_t)ldD / kDsT;
    dsP1.alloc(nbk * nbk * kDsT); dsP2.alloc(nbk * nbk * kDsT);
    e.sync();
    if (getenv("OSQP_AMD_SETUP_TRACE"))
      fprintf(stderr, "[supernodes] dense top: levels %d .. %d (%d supernodes, %d pivots, %.1f MB of inverse), %d boundary children (%d one-row, %d with a front vector)\n",
              snd_L0, T.nlev - 1, T.count - snd_J0, K, 8e-6 * (double)ldD * (double)ldD, snd_nb, pend_ptr[K], snd_nv);
  }
  // numeric: S = K_DD + the boundary children's update matrices, then -S^-1 by the block sweeps
  void factor_dense_top(hipStream_t s) {
    S0a.zero(s);
    OQ_LAUNCH(k_snd_init, dim3(snd_K), dim3(64), 0, s, snd_q0, ldD, (const int *)sn_piv.get(), (const int *)mf_slot.get(), (const int *)snd_dpos.get(), (const int64_t *)Lp.get(),
              (const int *)Li.get(), (const double *)Lx.get(), (const double *)D.get(), (const int *)snd_pend_ptr.get(), (const int64_t *)snd_pend_src.get(),
              (const double *)mf_U.get(), S0a.get());
    if (snd_nb) {
      SndExtArgs a{snd_tiles.get(), snd_trow_ptr.get(), snd_trow_list.get(), snd_bch.get(), mf_bsz.get(), mf_uoff.get(), snd_boff.get(),
                   snd_bslot.get(), snd_crange.get(), mf_U.get(), S0a.get(), ldD};
      OQ_LAUNCH(k_snd_extend, dim3(snd_ntiles), dim3(256), 0, s, a);
    }
    gj_invert(snd_K, s);
  }
  // x_D = S^-1 (b_D - what the levels below contribute), between the forward and the backward sweep of the levels below
  void solve_dense_top(hipStream_t s, bool tree) {
    const bool vec = tree && sn_top_Jt >= 0 && sn_top_Jt < snd_J0;  // the top part below D ran in the one-launch tree: its front vectors exist
    OQ_LAUNCH(k_snd_rhs, dim3(blocks_for((int64_t)snd_K * 64)), dim3(kBlock), 0, s, snd_q0, snd_K, (const int *)snd_dpos.get(), (const int64_t *)sn_Fp.get(),
              (const int64_t *)(vec ? sn_Ftop.get() : snd_Fd.get()), (const int *)sn_Fj.get(), (const double *)sn_Fx.get(), (const double *)bp.get(),
              (const int *)(vec && snd_nv ? snd_vptr.get() : nullptr), (const int64_t *)snd_vsrc.get(), (const double *)sn_uvec.get(), x2.get());
    const int nb = ldD / kDsT;
    OQ_LAUNCH(k_dense_apply_sym, dim3(nb * (nb + 1) / 2), dim3(256), 0, s, snd_K, ldD, nb, (const double *)Sinv, (const double *)x2.get(), dsP1.get(), dsP2.get());
    OQ_LAUNCH(k_dense_sym_reduce, dim3(nb), dim3(256), 0, s, snd_K, nb, (const double *)dsP1.get(), (const double *)dsP2.get(), bp.get() + snd_q0,
              (const int *)snd_dinv.get());
  }
  // -S^-1 in place of the K x K array in S0a (leading dimension ldD, a multiple of 64; identity on the padding): block sweeps of
  // kGjK pivots on the matrix cores
  // Block pattern of the sweeps (round 6).  The Schur complement of a dense top over the supernodes is block-sparse -- two
  // separators on either side of a third never meet until that one is eliminated -- and a sweep step on pivot block k changes
  // only the tiles (i, j) whose blocks i and j are both coupled with k (everything else is multiplied by the zeros of the
  // panels): R_k = { i : S_ik != 0 } + k, then S_ij becomes nonzero for all i, j in R_k -- the in-place inverse of the blocks
  // already swept included.  Simulated once at setup on the 64 x 64 block pattern (`pat`, symmetric, nt x nt); the step lists
  // go to the device.  Skipping a tile skips a subtraction of exact zeros: the same bits as the full sweeps.
  std::vector<int> gj_rptr, gj_tptr;
  std::vector<char> gj_next;
  DevBuf<int> gj_rows, gj_tiles;
  bool gj_sparse = false;
  void gj_symbolic(std::vector<char> &pat, int nt, hipStream_t s) {
    std::vector<int> rows, tiles;
    gj_rptr.assign(1, 0); gj_tptr.assign(1, 0); gj_next.clear();
    for (int k = 0; k < nt; k++) {
      std::vector<int> R;
      for (int i = 0; i < nt; i++) if (i == k || pat[(size_t)i * nt + k]) R.push_back(i);
      for (int i : R) for (int j : R) pat[(size_t)i * nt + j] = 1;
      rows.insert(rows.end(), R.begin(), R.end());
      // (the next pivot tile is left to the workgroup that sweeps it inside this step's launch, k_gj_update: gj_next says whether
      // this step changes it at all; a launch without that workgroup -- OSQP_AMD_GJ_FUSE=0 -- must not use these lists: see gj_invert)
      char nextc = 0;
      for (int i : R) for (int j : R) if (j <= i) { if (i == k + 1 && j == k + 1) { nextc = 1; continue; } tiles.push_back(i); tiles.push_back(j); }
      gj_next.push_back(nextc);
      gj_rptr.push_back((int)rows.size()); gj_tptr.push_back((int)(tiles.size() / 2));
    }
    gj_rows.alloc(std::max<size_t>(1, rows.size())); gj_rows.upload(rows.data(), rows.size(), s);
    gj_tiles.alloc(std::max<size_t>(1, tiles.size())); gj_tiles.upload(tiles.data(), tiles.size(), s);
    e.sync();
    gj_sparse = true;
    if (getenv("OSQP_AMD_SETUP_TRACE"))
      fprintf(stderr, "[supernodes] dense top: block sweeps touch %.1f %% of the tiles of a full sweep (%d steps)\n",
              100.0 * (double)(tiles.size() / 2) / ((double)nt * (double)nt * (double)(nt + 1) / 2.0), nt);
  }
  void gj_invert(int K, hipStream_t s) {
    const bool in_registers = !(getenv("OSQP_AMD_GJ_PIVOT_LDS") && atoi(getenv("OSQP_AMD_GJ_PIVOT_LDS")) == 1);   // 1: the round-5 pivot kernel (A/B, test)
    const bool fuse = in_registers && !(getenv("OSQP_AMD_GJ_FUSE") && atoi(getenv("OSQP_AMD_GJ_FUSE")) == 0);      // 0: the pivot block a launch of its own
    // (the tile lists leave the next pivot tile to the fused workgroup: without it, full sweeps)
    const bool sparse = gj_sparse && fuse && !(getenv("OSQP_AMD_GJ_SPARSE") && atoi(getenv("OSQP_AMD_GJ_SPARSE")) == 0);  // 0: full sweeps (A/B, test)
    if (ldD > K) OQ_LAUNCH(k_gj_pad, dim3(blocks_for(ldD - K)), dim3(kBlock), 0, s, K, ldD, S0a.get());
    const int nt = ldD / 64;
    const dim3 gu(nt, nt);
    auto pivot = [&](int p0, double *Tk) {
      if (in_registers) OQ_LAUNCH(k_gj_pivot_r, dim3(1), dim3(kGjPivotRThreads), 0, s, K, ldD, p0, (const double *)S0a.get(), Tk, status.get());
      else {
        // per device, not per process (the library serves several devices): set on every factorisation, as build_mf does
        HIP_CHECK(hipFuncSetAttribute((const void *)k_gj_pivot, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * 2 * kGjK * (kGjK + 1))));
        OQ_LAUNCH(k_gj_pivot, dim3(1), dim3(kGjPivotThreads), sizeof(double) * 2 * kGjK * (kGjK + 1), s, K, ldD, p0, (const double *)S0a.get(), Tk, status.get());
      }
    };
    for (int k = 0; k < nt; k++) {
      const int p0 = k * kGjK;
      double *Tk = gjT.get() + (size_t)(k & 1) * kGjK * kGjK, *Tn = gjT.get() + (size_t)((k + 1) & 1) * kGjK * kGjK;
      if (k == 0 || !fuse) pivot(p0, Tk);
      const int next = (fuse && k + 1 < nt) ? (sparse ? (gj_next[k] ? 2 : 1) : 2) : 0;
      if (sparse) {  // only the blocks coupled with the pivot block (gj_symbolic): the others see zeros in the panels
        const int nr = gj_rptr[k + 1] - gj_rptr[k], ntl = gj_tptr[k + 1] - gj_tptr[k];
        OQ_LAUNCH(k_gj_panel, dim3(nr), dim3(256), 0, s, ldD, p0, (const double *)S0a.get(), (const double *)Tk, gjW.get(), gjC.get(), (const int *)gj_rows.get() + gj_rptr[k]);
        OQ_LAUNCH(k_gj_update, dim3(ntl + (next ? 1 : 0)), dim3(256), 0, s, ldD, p0, S0a.get(), (const double *)Tk, (const double *)gjW.get(), (const double *)gjC.get(),
                  (const int *)gj_tiles.get() + 2 * (size_t)gj_tptr[k], ntl, next, Tn, K, status.get());
      } else {
        const int ntl = nt * (nt + 1) / 2;
        OQ_LAUNCH(k_gj_panel, dim3(nt), dim3(256), 0, s, ldD, p0, (const double *)S0a.get(), (const double *)Tk, gjW.get(), gjC.get(), (const int *)nullptr);
        OQ_LAUNCH(k_gj_update, dim3(ntl + (next ? 1 : 0)), dim3(256), 0, s, ldD, p0, S0a.get(), (const double *)Tk, (const double *)gjW.get(), (const double *)gjC.get(),
                  (const int *)nullptr, ntl, next, Tn, K, status.get());
      }
    }
    OQ_LAUNCH(k_gj_mirror, gu, dim3(256), 0, s, ldD, S0a.get());
    Sinv = S0a.get();
  }

  // ---- the device side of a LEAN analysis (symbolic.hpp): from the unsorted rows of the pattern of L ----------------------
  // CSC arrays of L: the rows go up as they were found, every entry with its row id beside it, and one sort by (column, row)
  // -- the transposition the setup already runs on A (kernels.hip csr_from_coo: counting / LSD radix sort, rows of the
  // result ascending) -- gives Lp / Li.  The scatter maps are one bisection per entry of triu(P) / A in its column of L.
  void lean_device_pattern() {
    hipStream_t s = e.stream;
    const int64_t nnzL = S.nnzL;
    DevBuf<int> ecol(std::max<int64_t>(1, nnzL)), erow(std::max<int64_t>(1, nnzL));
    {
      const LeanRows &R = *S.lean_rows;
      std::vector<int64_t> wp((size_t)N + 1, 0);  // row pointers in the order of the walk
      for (int r = 0; r < N; r++) wp[(size_t)r + 1] = wp[r] + (S.Rp[R.rowid[r] + 1] - S.Rp[R.rowid[r]]);
      DevBuf<int64_t> rp((size_t)N + 1);
      DevBuf<int> rowid((size_t)N), walk(std::max<int64_t>(1, nnzL));
      rp.upload(wp.data(), (size_t)N + 1, s);
      rowid.upload(R.rowid.data(), (size_t)N, s);
      for (size_t b = 0; b + 1 < R.first.size(); b++) {
        const std::vector<int> &c = R.cols[b];
        if ((int64_t)c.size() != wp[R.first[b + 1]] - wp[R.first[b]]) throw Error(6, "internal: a block of the lean analysis does not hold its rows");
        if (!c.empty()) HIP_CHECK(hipMemcpyAsync(ecol.get() + wp[R.first[b]], c.data(), c.size() * sizeof(int), hipMemcpyHostToDevice, s));
      }
      expand_colptr(N, rp.get(), nnzL, walk.get(), s);  // which row of the walk every entry belongs to ...
      if (nnzL > 0) OQ_LAUNCH(k_lean_rowid, dim3(blocks_for(nnzL)), dim3(kBlock), 0, s, nnzL, (const int *)walk.get(), (const int *)rowid.get(), erow.get());  // ... and which row of L that is
      e.sync();
    }
    DevCsr Lc;
    DevBuf<int> src;
    e.setup_mark("    rows of L uploaded");
    csr_from_coo(N, N, nnzL, ecol.get(), erow.get(), Lc, src, s);  // "rows" of the result = columns of L
    e.setup_mark("    CSC arrays of L");
    if (Lc.nnz != nnzL) throw Error(6, "internal: the transposed pattern of L lost entries");
    Lc.val.release(); src.release(); ecol.release(); erow.release();
    Lp = std::move(Lc.rowptr); Li = std::move(Lc.col);
    device_scatter_maps();
  }
  // where every entry of triu(P) and of A sits in L: one bisection per entry in its column of the CSC arrays (identity row map)
  void device_scatter_maps() {
    hipStream_t s = e.stream;
    PtoL.alloc(std::max<int64_t>(1, e.nnzPtriu)); AtoL.alloc(std::max<int64_t>(1, e.nnzA));
    if (e.nnzPtriu > 0) {
      DevBuf<int> pcol((size_t)e.nnzPtriu);
      expand_colptr(n, e.Pp_keep.get(), e.nnzPtriu, pcol.get(), s);
      OQ_LAUNCH(k_lean_map, dim3(blocks_for(e.nnzPtriu)), dim3(kBlock), 0, s, e.nnzPtriu, (const int *)pcol.get(), (const int *)e.Pi_keep.get(), 0, pinv.get(),
                Lp.get(), Li.get(), PtoL.get());
      e.sync();
    }
    if (e.nnzA > 0) {
      DevBuf<int> acol((size_t)e.nnzA);
      expand_colptr(n, e.At.rowptr.get(), e.nnzA, acol.get(), s);
      OQ_LAUNCH(k_lean_map, dim3(blocks_for(e.nnzA)), dim3(kBlock), 0, s, e.nnzA, (const int *)acol.get(), (const int *)e.At.col.get(), n, pinv.get(),
                Lp.get(), Li.get(), AtoL.get());
      e.sync();
    }
  }
  // Row and column lists of the entries outside the diagonal blocks, in slot order: two more sorts of the entries of L,
  // keyed (slot of the row, slot of the column) and the other way round; an entry inside a block carries row -1 (dropped).
  void lean_device_lists(int q_upper) {
    hipStream_t s = e.stream;
    const int64_t nnzL = S.nnzL;
    auto up32 = [&](DevBuf<int> &d, const std::vector<int> &h) { d.alloc(std::max<size_t>(1, h.size())); d.upload(h.data(), h.size(), s); };
    up32(mf_snof, mfh_snof); up32(mf_slot, T.slot);
    DevBuf<int> lcol(std::max<int64_t>(1, nnzL)), er(std::max<int64_t>(1, nnzL)), ec(std::max<int64_t>(1, nnzL));
    expand_colptr(N, Lp.get(), nnzL, lcol.get(), s);
    for (int pass = 0; pass < 2; pass++) {  // 0: rows (forward lists F), 1: columns (backward lists G)
      if (nnzL > 0)
        OQ_LAUNCH(k_lean_keys, dim3(blocks_for(nnzL)), dim3(kBlock), 0, s, nnzL, (const int *)lcol.get(), (const int *)Li.get(), (const int *)mf_snof.get(),
                  (const int *)mf_slot.get(), pass, er.get(), ec.get());
      DevCsr M;
      DevBuf<int> src;
      csr_from_coo(N, N, nnzL, er.get(), ec.get(), M, src, s);
      M.val.release();
      DevBuf<int64_t> pos(std::max<int64_t>(1, M.nnz));
      if (M.nnz > 0) OQ_LAUNCH(k_widen, dim3(blocks_for(M.nnz)), dim3(kBlock), 0, s, M.nnz, (const int *)src.get(), pos.get());
      std::vector<int64_t> &hp = pass == 0 ? T.Fp : T.Gp;
      hp.resize((size_t)N + 1);
      M.rowptr.download(hp.data(), (size_t)N + 1, s);
      e.sync();
      if (pass == 0) { sn_Fp = std::move(M.rowptr); sn_Fj = std::move(M.col); sn_Fpos = std::move(pos); }
      else { sn_Gp = std::move(M.rowptr); sn_Gi = std::move(M.col); sn_Gpos = std::move(pos); }
      e.setup_mark(pass == 0 ? "    forward lists" : "    backward lists");
    }
    sn_Fsplit.alloc((size_t)N);
    OQ_LAUNCH(k_lean_split, dim3(blocks_for(N)), dim3(kBlock), 0, s, N, (const int64_t *)sn_Fp.get(), (const int *)sn_Fj.get(), q_upper, sn_Fsplit.get());
    T.flops = 0.0;
    for (int J = 0; J < T.count; J++) { const double sz = (double)(T.ptr[J + 1] - T.ptr[J]); T.flops += sz * sz; }
    T.flops = 2.0 * (T.flops + (double)T.Fp[N]);
    e.sync();
  }

  // Plan of the multifrontal factorisation (mfront.hpp): fronts, update-matrix offsets, children lists, launches by level
  // and size class on the host (arrays of one entry per supernode); the two index maps that have one entry per entry of L /
  // per border row are built by device kernels.  Not taken (the level-by-level factorisation stays) when a front does not
  // fit LDS or the update matrices would need more than 8 GB.  OSQP_AMD_MF=0 switches it off.
  static constexpr int kMfMaxFront = 192;
  bool mf_plan() {
    const int mode = getenv("OSQP_AMD_MF") ? atoi(getenv("OSQP_AMD_MF")) : 1;
    if (mode == 0) return false;
    const int count = T.count;
    snd_L0 = snd_J0 = -1; snd_q0 = 0; snd_K = 0;
    mfh_bsz.assign(count, 0); mfh_snof.assign(N, 0);
    mfh_uoff.assign(count + 1, 0); mfh_reloff.assign(count + 1, 0); mfh_poff.assign(count + 1, 0);
    mf_fmax = 0; mf_big_count = 0; mf_big_fmax = 0;
    mfh_tiles.clear(); mfh_chunks.clear();
    // tests: OSQP_AMD_MF_MAX_FRONT sends smaller fronts through the global-memory kernels too (the zoo at test sizes has none beyond 192 rows)
    const int max_front = getenv("OSQP_AMD_MF_MAX_FRONT") ? std::min(kMfMaxFront, std::max(1, atoi(getenv("OSQP_AMD_MF_MAX_FRONT")))) : kMfMaxFront;
    const bool big_ok = !(getenv("OSQP_AMD_MF_BIG") && atoi(getenv("OSQP_AMD_MF_BIG")) == 0);  // 0: the round-5 behaviour (A/B runs)
    // which fronts go through global memory: those beyond LDS, and -- so that the set is closed towards the root: the
    // solves treat the top of the tree by front vectors (direct_sn_kernels.hpp, SnTop), which needs a resident panel for
    // every supernode above a large front -- their ancestors (the last segments of a top separator: a few small fronts at the
    // end of a chain of large ones)
    mf_is_big.assign(count, 0);
    for (int J = 0; J < count; J++) {
      const int top = T.piv[T.ptr[J + 1] - 1], s_ = T.ptr[J + 1] - T.ptr[J];
      if (s_ + (S.Lp[top + 1] - S.Lp[top]) > max_front) mf_is_big[J] = 1;
    }
    if (!(getenv("OSQP_AMD_MF_BIG_ANCESTORS") && atoi(getenv("OSQP_AMD_MF_BIG_ANCESTORS")) == 0))
      for (int J = 0; J < count; J++)  // parents have larger numbers: one ascending pass
        if (mf_is_big[J] && T.up[J] >= 0 && T.ptr[T.up[J] + 1] - T.ptr[T.up[J]] <= kMfbSmax) mf_is_big[T.up[J]] = 1;
    for (int J = 0; J < count; J++) {
      const int top = T.piv[T.ptr[J + 1] - 1], s_ = T.ptr[J + 1] - T.ptr[J];
      const int64_t b = S.Lp[top + 1] - S.Lp[top];
      // (round 6) a front beyond one workgroup's LDS is factorised out of global memory (mfront_big.hpp) instead of sending
      // the whole matrix back to the level-by-level factorisation; what still does: a front beyond the 16-bit row indices of
      // `rel` / `loc`, or a supernode of more pivots than the pivot block of that kernel holds
      if (mf_is_big[J]) {
        if (s_ + b > 65000 || s_ > kMfbSmax || !big_ok) return false;
        mf_big_count++;
        mf_big_fmax = std::max(mf_big_fmax, s_ + (int)b);
        mfh_poff[J + 1] = (int64_t)(s_ + b) * s_;
      } else mf_fmax = std::max(mf_fmax, s_ + (int)b);
      mfh_bsz[J] = (int)b;  // (where its update matrix lives: place_update_matrices, once the dense top is known)
      mfh_reloff[J + 1] = mfh_reloff[J] + b;
      for (int q = T.ptr[J]; q < T.ptr[J + 1]; q++) mfh_snof[T.piv[q]] = J;
    }
    for (int J = 0; J < count; J++) mfh_poff[J + 1] += mfh_poff[J];
    choose_dense_top_over_supernodes();
    place_update_matrices();
    {  // the update matrices (as placed: the large ones share memory over their lifetimes), beside the factor: they have to fit what
       // the device has free
      size_t free_b = 0, total_b = 0;
      const int64_t need = 8 * (mfh_uoff[count] + mfh_poff[count]);
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
      if (mfh_uoff[count] > ((int64_t)1 << 32) || (free_b && (double)need > 0.5 * (double)free_b)) return false;
    }
    mfh_chp.assign(count + 1, 0);
    for (int J = 0; J < count; J++) if (T.up[J] >= 0) mfh_chp[T.up[J] + 1]++;
    for (int J = 0; J < count; J++) mfh_chp[J + 1] += mfh_chp[J];
    mfh_chl.resize(mfh_chp[count]);
    {
      std::vector<int> f(mfh_chp.begin(), mfh_chp.end() - 1);
      for (int J = 0; J < count; J++) if (T.up[J] >= 0) mfh_chl[f[T.up[J]]++] = J;  // ascending: the order of the sums
    }
    mfh_list.clear();
    mfh_list.reserve(count);
    mf_launches.clear();
    for (int L = 0; L < (snd_K ? snd_L0 : T.nlev); L++) {  // (the supernodes of the dense top have no front)
      std::vector<int> cls[kMfClasses];
      int cap[kMfClasses] = {0};
      std::vector<int> bigs;
      int bigcap = 0;
      for (int J = T.lvl_ptr[L]; J < T.lvl_ptr[L + 1]; J++) {
        const int f = T.ptr[J + 1] - T.ptr[J] + mfh_bsz[J];
        if (mf_is_big[J]) { bigs.push_back(J); bigcap = std::max(bigcap, f); continue; }
        int c = 0;
        while (f > kMfClassCap[c]) c++;
        cls[c].push_back(J);
        cap[c] = std::max(cap[c], f);
      }
      // a workgroup class of few fronts joins the next larger one of its level: a launch costs the latency of one front
      // whatever its size (~90 us), the finer slabs only pay where thousands of fronts share the device
      for (int c = 0; c + 1 < kMfClasses; c++) {
        if (cls[c].empty() || cls[c].size() >= 1024) continue;
        int up = c + 1;
        while (up + 1 < kMfClasses && cls[up].empty()) up++;
        if (cls[up].empty()) continue;
        cls[up].insert(cls[up].end(), cls[c].begin(), cls[c].end());
        std::sort(cls[up].begin(), cls[up].end());
        cap[up] = std::max(cap[up], cap[c]);
        cls[c].clear();
      }
      for (int c = 0; c < kMfClasses; c++) {
        if (cls[c].empty()) continue;
        mf_launches.push_back({c, (int)mfh_list.size(), (int)cls[c].size(), c < 2 ? kMfClassCap[c] : cap[c]});
        mfh_list.insert(mfh_list.end(), cls[c].begin(), cls[c].end());
      }
      if (!bigs.empty()) {  // the level's fronts beyond LDS: one panel launch, one launch over the tiles of their update matrices
        MfLaunch l{kMfBig, (int)mfh_list.size(), (int)bigs.size(), bigcap};
        l.tile_off = (int)(mfh_tiles.size() / 3);
        for (int i = 0; i < (int)bigs.size(); i++) {
          const int nt = (mfh_bsz[bigs[i]] + kMfbTile - 1) / kMfbTile;
          for (int ti = 0; ti < nt; ti++)
            for (int tj = 0; tj <= ti; tj++) { mfh_tiles.push_back(i); mfh_tiles.push_back(ti); mfh_tiles.push_back(tj); }
        }
        l.tile_count = (int)(mfh_tiles.size() / 3) - l.tile_off;
        l.chunk_off = (int)(mfh_chunks.size() / 2);
        for (int i = 0; i < (int)bigs.size(); i++)
          for (int ch = 0; ch < (mfh_bsz[bigs[i]] + 63) / 64; ch++) { mfh_chunks.push_back(i); mfh_chunks.push_back(ch); }
        l.chunk_count = (int)(mfh_chunks.size() / 2) - l.chunk_off;
        mf_launches.push_back(l);
        mfh_list.insert(mfh_list.end(), bigs.begin(), bigs.end());
      }
    }
    return true;
  }
  // Where the update matrices live (round 6).  Up to now every supernode had its own place for the whole factorisation: the sum
  // of b (b + 1) / 2 over all fronts -- fine for chains and 2-D structures (0.3 GB on control-1e6, 1.1 GB on a 1000 x 1000 grid),
  // beyond the 2^32-double limit of the plan on a 50 x 50 x 50 grid (borders of 4 000 rows: 60 - 190 MB per front, ~5e9 doubles
  // in all), which sent the whole matrix back to the level-by-level factorisation: 1.07 s per refactorisation.  An update matrix
  // is written in the launches of its supernode's level and read in the launches of its parent's level, nowhere else: the LARGE
  // ones (>= kMfShareMin doubles; a few thousand at most) are placed by their lifetimes -- level by level: place the level's own,
  // then release those whose parent sits on this level (first fit over a free list, neighbours merged) --, the small ones keep
  // a place of their own (their sum is small, and a million of them through a free list would be a quadratic plan).
  // mfh_uoff[J] = offset of supernode J, mfh_uoff[count] = doubles in all.  The supernodes of a dense top have no update matrix;
  // their boundary children's live to the end (the dense array is assembled after the last level).
  static constexpr int64_t kMfShareMin = 16384;
  void place_update_matrices() {
    const int count = T.count;
    const bool share = !(getenv("OSQP_AMD_MF_SHARE_U") && atoi(getenv("OSQP_AMD_MF_SHARE_U")) == 0);  // 0: every update matrix a place of its own (A/B, tests)
    const int j_end = snd_K ? snd_J0 : count;
    const int64_t share_min = getenv("OSQP_AMD_MF_SHARE_MIN") ? std::max(1, atoi(getenv("OSQP_AMD_MF_SHARE_MIN"))) : kMfShareMin;  // (tests: small problems)
    auto size_of = [&](int J) { return J < j_end ? (int64_t)mfh_bsz[J] * (mfh_bsz[J] + 1) / 2 : (int64_t)0; };
    int64_t base = 0;
    for (int J = 0; J < count; J++) {
      const int64_t sz = size_of(J);
      mfh_uoff[J] = base;
      if (!share || sz < share_min) base += sz;
    }
    if (!share) { mfh_uoff[count] = base; return; }
    // the large ones: [offset, size) blocks above `base`
    std::vector<std::pair<int64_t, int64_t>> free_list;  // sorted by offset
    int64_t top = base;
    std::vector<std::vector<int>> dies((size_t)T.nlev);   // per level: the large update matrices whose parent sits there
    std::vector<int> level_of((size_t)count, 0);
    for (int L = 0; L < T.nlev; L++) for (int J = T.lvl_ptr[L]; J < T.lvl_ptr[L + 1]; J++) level_of[J] = L;
    for (int L = 0; L < T.nlev; L++) {
      for (int J = T.lvl_ptr[L]; J < T.lvl_ptr[L + 1] && J < j_end; J++) {
        const int64_t sz = size_of(J);
        if (sz < share_min) continue;
        size_t k = 0;
        while (k < free_list.size() && free_list[k].second < sz) k++;
        if (k < free_list.size()) {
          mfh_uoff[J] = free_list[k].first;
          if (free_list[k].second == sz) free_list.erase(free_list.begin() + (long)k);
          else { free_list[k].first += sz; free_list[k].second -= sz; }
        } else if (!free_list.empty() && free_list.back().first + free_list.back().second == top) {  // grow the last free block at the top
          mfh_uoff[J] = free_list.back().first;
          top = free_list.back().first + sz;
          free_list.pop_back();
        } else { mfh_uoff[J] = top; top += sz; }
        const int P = T.up[J];
        if (P >= 0 && P < j_end) dies[level_of[P]].push_back(J);  // (a parent in the dense top: read after the last level -- never released)
      }
      for (int J : dies[L]) {
        std::pair<int64_t, int64_t> blk(mfh_uoff[J], size_of(J));
        auto it = std::lower_bound(free_list.begin(), free_list.end(), blk);
        it = free_list.insert(it, blk);
        if (it + 1 != free_list.end() && it->first + it->second == (it + 1)->first) { it->second += (it + 1)->second; free_list.erase(it + 1); }
        if (it != free_list.begin() && (it - 1)->first + (it - 1)->second == it->first) { (it - 1)->second += it->second; free_list.erase(it); }
      }
    }
    mfh_uoff[count] = top;
  }
  // The dense top over the supernodes (direct_sndense_kernels.hpp): the largest set of whole top levels with at most
  // OSQP_AMD_SN_DENSE_MAX pivots.  Taken by default where fronts go through global memory (a 2-D structure: the top levels are
  // chains of large separators, ~300 us of factorisation and ~28 us of solve per level) and at least eight levels go; a tree whose
  // fronts all fit LDS (the control family) keeps its one-launch top: six levels of small supernodes cost less than the product
  // with a block of their thousands of pivots.  OSQP_AMD_SN_DENSE = 0 never, 2 whenever a level fits (tests: small problems).
  void choose_dense_top_over_supernodes() {
    snd_L0 = snd_J0 = -1; snd_q0 = 0; snd_K = 0;
    const int mode = getenv("OSQP_AMD_SN_DENSE") ? atoi(getenv("OSQP_AMD_SN_DENSE")) : 1;
    if (mode == 0 || T.nlev < 2) return;
    const int kmax = getenv("OSQP_AMD_SN_DENSE_MAX") ? atoi(getenv("OSQP_AMD_SN_DENSE_MAX")) : 4300;
    int L = T.nlev;
    while (L - 1 >= 1 && N - T.ptr[T.lvl_ptr[L - 1]] <= kmax) L--;
    // ... and on while the next level down is still a chain (at most four supernodes: a level of pure latency in the solves and of
    // a handful of workgroups in the factorisation), up to 8 192 pivots: the separators of a 3-D structure are planes -- a
    // 50 x 50 x 50 grid has 63 levels of one supernode each above 4 247 pivots and is still at 1.5 supernodes a level at 7 817
    // (293 -> 694 it/s for 22 -> 32 ms of refactorisation); a 700 x 700 grid is at eleven a level below its 4 137 and stops there.
    // (Not when the size was asked for explicitly.)
    if (!getenv("OSQP_AMD_SN_DENSE_MAX"))
      while (L - 1 >= 1 && T.lvl_ptr[L] - T.lvl_ptr[L - 1] <= 4 && N - T.ptr[T.lvl_ptr[L - 1]] <= 8192) L--;
    if (L == T.nlev) return;
    const int K = N - T.ptr[T.lvl_ptr[L]];
    if (mode != 2 && (mf_big_count == 0 || T.nlev - L < 8 || K < 256)) return;
    snd_L0 = L; snd_J0 = T.lvl_ptr[L]; snd_q0 = T.ptr[snd_J0]; snd_K = K;
  }
  // size classes of the fronts: rows at most 16 / 48 (16 lanes / a wavefront each, fixed slabs), then workgroups with the
  // slab of the launch's largest front -- cut at 64 / 80 / 96 / 128 rows so that a few large fronts do not take the occupancy
  // of many mid-size ones (the fronts of a compute unit are as many as their slabs fit its 160 KB of LDS)
  static constexpr int kMfWideCount = 192;  // launches of at most this many workgroup-class fronts give each 1024 threads (the device is not full either way)
  static constexpr int kMfClasses = 7, kMfBig = 100;
  static constexpr int kMfClassCap[kMfClasses] = {16, 48, 64, 80, 96, 128, kMfMaxFront};
  void build_mf() {
    mf = false;
    if (!sn || !mf_ok) return;
    const int count = T.count;
    hipStream_t s = e.stream;
    auto up64 = [&](DevBuf<int64_t> &d, const std::vector<int64_t> &h) { d.alloc(std::max<size_t>(1, h.size())); d.upload(h.data(), h.size(), s); };
    auto up32 = [&](DevBuf<int> &d, const std::vector<int> &h) { d.alloc(std::max<size_t>(1, h.size())); d.upload(h.data(), h.size(), s); };
    if (!mf_snof.n) { up32(mf_snof, mfh_snof); up32(mf_slot, T.slot); }
    up32(mf_bsz, mfh_bsz); up32(mf_chp, mfh_chp); up32(mf_chl, mfh_chl); up32(mf_list, mfh_list);
    up64(mf_uoff, mfh_uoff); up64(mf_reloff, mfh_reloff);
    mf_rel.alloc(std::max<int64_t>(1, mfh_reloff[count])); mf_loc.alloc(std::max<int64_t>(1, S.nnzL)); mf_U.alloc(std::max<int64_t>(1, mfh_uoff[count]));
    up64(mf_poff, mfh_poff); up32(mf_tiles, mfh_tiles); up32(mf_chunks, mfh_chunks);
    mf_wide.alloc(std::max<size_t>(1, mfh_chl.size())); mf_nwide.alloc((size_t)count);
    mf_ct0.alloc(std::max<size_t>(1, (mfh_chunks.size() / 2) * kMfbSmax));
    mf_panel.alloc(std::max<int64_t>(1, mfh_poff[count]));
    mf_err.alloc(1); mf_err.zero(s);
    OQ_LAUNCH(k_mf_rel, dim3(blocks_for(count)), dim3(kBlock), 0, s, count, sn_ptr.get(), sn_piv.get(), sn_up.get(), mf_snof.get(), mf_slot.get(),
              Lp.get(), Li.get(), mf_reloff.get(), mf_rel.get(), mf_err.get());
    if (S.nnzL > 0)
      OQ_LAUNCH(k_mf_loc, dim3(blocks_for(S.nnzL)), dim3(kBlock), 0, s, S.nnzL, (const int *)Lcol.get(), Li.get(), Lp.get(), sn_ptr.get(), sn_piv.get(),
                mf_snof.get(), mf_slot.get(), mf_loc.get(), mf_err.get());
    int err = 0;
    mf_err.download(&err, 1, s);
    e.sync();
    if (err) throw Error(6, "internal: the fronts of the supernodes do not cover the pattern of L (code " + std::to_string(err) + ")");
    for (const MfLaunch &l : mf_launches)  // where the 64-row pieces of the big fronts begin in the columns of L (k_mfb_rows)
      if (l.cls == kMfBig && l.chunk_count) {
        MfArgs a{mf_list.get() + l.off, l.count, l.fcap, sn_ptr.get(), sn_piv.get(), mf_bsz.get(), mf_uoff.get(), mf_reloff.get(), mf_rel.get(),
                 mf_chp.get(), mf_chl.get(), Lp.get(), mf_loc.get(), Lx.get(), D.get(), Dinv.get(), mf_U.get(), status.get(),
                 sn_Wc.get(), sn_Wr.get(), sn_woff.get()};
        MfbSplitArgs sa{MfbArgs{a, mf_poff.get(), mf_panel.get(), nullptr}, mf_wide.get(), mf_nwide.get(), mf_chunks.get() + 2 * (size_t)l.chunk_off,
                        mf_ct0.get() + (size_t)l.chunk_off * kMfbSmax};
        OQ_LAUNCH(k_mfb_chunk_t0, dim3(blocks_for((int64_t)l.chunk_count * kMfbSmax, 256)), dim3(256), 0, s, l.chunk_count, sa);
      }
    e.sync();
    const int lds = (int)(mf_slab_doubles(mf_fmax) * sizeof(double));
    HIP_CHECK(hipFuncSetAttribute((const void *)k_mf_front<256>, hipFuncAttributeMaxDynamicSharedMemorySize, std::max(lds, 65536)));
    HIP_CHECK(hipFuncSetAttribute((const void *)k_mf_front<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, std::max(lds, 65536)));
    HIP_CHECK(hipFuncSetAttribute((const void *)k_mf_front<512>, hipFuncAttributeMaxDynamicSharedMemorySize, std::max(lds, 65536)));
    if (mf_big_count) {
      HIP_CHECK(hipFuncSetAttribute((const void *)k_mfb_panel, hipFuncAttributeMaxDynamicSharedMemorySize, std::max((int)mfb_panel_lds(mf_big_fmax), 65536)));
      HIP_CHECK(hipFuncSetAttribute((const void *)k_mfb_pivot, hipFuncAttributeMaxDynamicSharedMemorySize, std::max((int)mfb_panel_lds(0), 65536)));
      HIP_CHECK(hipFuncSetAttribute((const void *)k_mfb_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * kMfbSmax * 64 * sizeof(double))));
    }
    if (getenv("OSQP_AMD_SETUP_TRACE")) {
      for (const MfLaunch &l : mf_launches) fprintf(stderr, "[fronts] class %d: %d fronts, slab for %d rows\n", l.cls, l.count, l.fcap);
      fprintf(stderr, "[fronts] beyond LDS: %d fronts (largest %d rows), %.1f MB of panels, update matrices of all fronts %.1f MB\n", mf_big_count,
              mf_big_fmax, 8e-6 * (double)mfh_poff[count], 8e-6 * (double)mfh_uoff[count]);
    }
    std::vector<int>().swap(mfh_snof); std::vector<int>().swap(mfh_list); std::vector<int>().swap(mfh_chl);
    mf = true;
  }
  const bool mfb_split = !(getenv("OSQP_AMD_MFB_SPLIT") && atoi(getenv("OSQP_AMD_MFB_SPLIT")) == 0);  // 0: one workgroup per big front (A/B, tests)
  void run_mf(hipStream_t s) {
    for (const MfLaunch &l : mf_launches) {
      MfArgs a{mf_list.get() + l.off, l.count, l.fcap, sn_ptr.get(), sn_piv.get(), mf_bsz.get(), mf_uoff.get(), mf_reloff.get(), mf_rel.get(),
               mf_chp.get(), mf_chl.get(), Lp.get(), mf_loc.get(), Lx.get(), D.get(), Dinv.get(), mf_U.get(), status.get(),
               sn_Wc.get(), sn_Wr.get(), sn_woff.get()};
      if (l.cls == kMfBig) {
        MfbArgs g{a, mf_poff.get(), mf_panel.get(), mf_tiles.get() + 3 * (size_t)l.tile_off};
        if (mfb_split) {  // the pivot blocks, then the panels in 64-row pieces (mfront_big.hpp)
          MfbSplitArgs sa{g, mf_wide.get(), mf_nwide.get(), mf_chunks.get() + 2 * (size_t)l.chunk_off, mf_ct0.get() + (size_t)l.chunk_off * kMfbSmax};
          OQ_LAUNCH(k_mfb_pivot, dim3(l.count), dim3(kMfbPanelThreads), mfb_panel_lds(0), s, sa);
          if (l.chunk_count) OQ_LAUNCH(k_mfb_rows, dim3(l.chunk_count), dim3(256), 2 * kMfbSmax * 64 * sizeof(double), s, sa);
        } else
        OQ_LAUNCH(k_mfb_panel, dim3(l.count), dim3(kMfbPanelThreads), mfb_panel_lds(l.fcap), s, g);
        if (l.tile_count) OQ_LAUNCH(k_mfb_update, dim3(l.tile_count), dim3(256), 0, s, g);
      } else if (l.cls == 0) OQ_LAUNCH(k_mf_front<16>, dim3((l.count + 15) / 16), dim3(kMfBlock), 16 * mf_slab_doubles(l.fcap) * sizeof(double), s, a);
      else if (l.cls == 1) OQ_LAUNCH(k_mf_front<64>, dim3((l.count + 3) / 4), dim3(kMfBlock), 4 * mf_slab_doubles(l.fcap) * sizeof(double), s, a);
      else {
        // threads per front by what the slab leaves of a compute unit (160 KB of LDS): a slab beyond 80 KB is alone there whatever
        // its workgroup size -- 1024 threads instead of four wavefronts on an otherwise empty unit (round 6: grid 700 x 700, 653
        // fronts of up to 187 rows in one launch, 821 us) --, one beyond 40 KB shares it with one other: 512
        const size_t slab = mf_slab_doubles(l.fcap) * sizeof(double);
        const int wide = getenv("OSQP_AMD_MF_WIDE") ? atoi(getenv("OSQP_AMD_MF_WIDE")) : 1;  // 0: the round-5 rule (A/B)
        if (l.count <= kMfWideCount || (wide && slab > 80 * 1024)) OQ_LAUNCH(k_mf_front<1024>, dim3(l.count), dim3(1024), slab, s, a);
        else if (wide && slab > 40 * 1024) OQ_LAUNCH(k_mf_front<512>, dim3(l.count), dim3(512), slab, s, a);
        else OQ_LAUNCH(k_mf_front<256>, dim3(l.count), dim3(kMfBlock), slab, s, a);
      }
    }
  }

  // rows at least this long (mean over a level) go through the work-row form of phase 2: the thread-per-entry merge
  // of two rows of 50-300 entries is ~100 us of dependent loads, the wave-per-entry product ~7 us (control, T = 800)
  static double long_row_mean() { const char *v = getenv("OSQP_AMD_LONG_ROW_MEAN"); return v ? atof(v) : 24.0; }
  static int pick(double mean) { return mean <= 2.0 ? 1 : (mean <= 8.0 ? 4 : (mean <= 32.0 ? 16 : 64)); }

  static double solve_cost_us(const Symbolic &Y, double chain_level_us = 1.4) {
    return level_solve_cost_us(Y, kChainRows, dense_max(), kDenseMin, chain_level_us);
  }

  // The dense top block (symbolic.hpp, choose_dense_top)
  // Largest block: 12288 pivots (1.2 GB of inverse, a 1.2 GB product per solve -- ~0.2 ms -- against 24 000 dependent steps
  // of the chain it replaces); up to round 3 the limit was 2048 because the inversion was one pass over the array per two
  // pivots.  OSQP_AMD_DENSE_MAX overrides.
  static constexpr int kDenseSparseMax = 1024, kDenseMin = 32, kDenseBlocked = 512;
  static int dense_max() { static const int v = getenv("OSQP_AMD_DENSE_MAX") ? atoi(getenv("OSQP_AMD_DENSE_MAX")) : 12288; return v; }
  void choose_dense_block() {
    lD = nlev; cD = N; kD = 0;
    static const bool enabled = !(getenv("OSQP_AMD_DENSE_TOP") && atoi(getenv("OSQP_AMD_DENSE_TOP")) == 0);
    kD_dense = false;
    if (enabled) choose_dense_top(S, kChainRows, dense_max(), kDenseSparseMax, kDenseMin, lD, cD, kD, &kD_dense);
  }
  // columns of the block whose work rows (N doubles each) are held at once: at most 256 MB
  int dense_batch() const { return (int)std::max<size_t>(1, std::min<size_t>((size_t)kD, ((size_t)256 << 20) / ((size_t)N * sizeof(double)))); }

  void build_schedule() {
    const auto &lp = S.level_ptr;
    // forward: level 0 rows have no predecessors (nothing to do); backward: every level (D^-1 applies everywhere)
    std::vector<int64_t> split(S.Rp.begin(), S.Rp.end() - 1), lsplit(S.Lp.begin() + 1, S.Lp.end());  // default: no far part
    auto make = [&](bool forward) {
      std::vector<Step> steps;
      int l = forward ? 1 : 0;
      while (l < lD) {
        int width = lp[l + 1] - lp[l];
        if (width <= kChainRows) {
          int l2 = l;
          while (l2 < lD && lp[l2 + 1] - lp[l2] <= kChainRows && lp[l2 + 1] - lp[l] <= kChainLdsRows) l2++;  // pieces that fit LDS
          if (l2 == l) l2 = l + 1;
          int T = 64;
          if (forward) {  // split the rows of the chain at its first pivot
            const int c0 = lp[l], c1 = lp[l2];
            int64_t far = 0;
            for (int r = c0; r < c1; r++) {
              const int *beg = S.Rj.data() + S.Rp[r], *end = S.Rj.data() + S.Rp[r + 1];
              split[r] = S.Rp[r] + (std::lower_bound(beg, end, c0) - beg);
              far += split[r] - S.Rp[r];
            }
            if (far / std::max(1, c1 - c0) > 1024) T = kBlock;
          }
          if (!forward) {  // split the columns of the chain at its end
            const int c0 = lp[l], c1 = lp[l2];
            int64_t far = 0;
            for (int r = c0; r < c1; r++) {
              const int *beg = S.Li.data() + S.Lp[r], *end = S.Li.data() + S.Lp[r + 1];
              lsplit[r] = S.Lp[r] + (std::lower_bound(beg, end, c1) - beg);
              far += S.Lp[r + 1] - lsplit[r];
            }
            if (far / std::max(1, c1 - c0) > 1024) T = kBlock;
          }
          Step st{1, l, l2, T};
          {  // longest row inside the chain decides how many entries a lane keeps prefetched
            const int c0 = lp[l], c1 = lp[l2];
            int64_t longest = 0;
            for (int r = c0; r < c1; r++) longest = std::max(longest, forward ? S.Rp[r + 1] - split[r] : lsplit[r] - S.Lp[r]);
            st.U = longest > 320 ? 8 : (longest > 224 ? 4 : 2);  // measured: 200-row blocks are fastest with 2, 500-row with 8
            // short rows (the separators of a nested-dissection tree, banded factors): several rows per wavefront
            int64_t inside = 0;
            for (int r = c0; r < c1; r++) inside += forward ? S.Rp[r + 1] - split[r] : lsplit[r] - S.Lp[r];
            const double mean = (double)inside / (double)std::max(1, c1 - c0);
            const bool wide = (c1 - c0) > 2 * (l2 - l);  // more than two rows per level on average
            if (wide && longest <= 64 && mean <= 12.0) { st.L = 4; st.U = longest > 8 ? 4 : 2; }
            else if (wide && longest <= 256 && mean <= 48.0) { st.L = 16; st.U = longest > 64 ? 8 : (longest > 32 ? 4 : 2); }
          }
          steps.push_back(st);
          l = l2;
        } else {
          const std::vector<int64_t> &ptr = forward ? S.Rp : S.Lp;
          double mean = (double)(ptr[lp[l + 1]] - ptr[lp[l]]) / (double)width;
          steps.push_back({0, lp[l], lp[l + 1], pick(mean)});
          l++;
        }
      }
      return steps;
    };
    fwd = make(true);
    bwd = make(false);
    std::reverse(bwd.begin(), bwd.end());
    for (int r = cD; r < N; r++) {  // rows of the dense block split at its first pivot
      const int *beg = S.Rj.data() + S.Rp[r], *end = S.Rj.data() + S.Rp[r + 1];
      split[r] = S.Rp[r] + (std::lower_bound(beg, end, cD) - beg);
    }
    Rsplit.alloc(split.size());
    Rsplit.upload(split.data(), split.size(), e.stream);
    Lsplit.alloc(lsplit.size());
    Lsplit.upload(lsplit.data(), lsplit.size(), e.stream);
    e.sync();
  }

  // returns 0 ok, 4 zero pivot, 5 wrong inertia
  int refactor(const double *cdiag) {
    hipStream_t s = e.stream;
    Lx.zero(s);
    status.zero(s);
    OQ_LAUNCH(k_diag_init, dim3(blocks_for(N)), dim3(kBlock), 0, s, N, n, sigma, pinv.get(), cdiag, cconst, D.get());
    if (e.nnzPtriu > 0)
      OQ_LAUNCH(k_scatter_P, dim3(blocks_for(e.nnzPtriu)), dim3(kBlock), 0, s, e.nnzPtriu, PtoL.get(), e.P_k2lo.get(),
                e.Pf.val.get(), Lx.get(), D.get());
    if (e.nnzA > 0)
      OQ_LAUNCH(k_scatter_A, dim3(blocks_for(e.nnzA)), dim3(kBlock), 0, s, e.nnzA, AtoL.get(), e.At.val.get(), Lx.get());
    int p0 = 0, p1 = 0, half = 0;  // the previous level that went through work rows, and the half of W it used
    if (mf) run_mf(s);
    if (mf && snd_K) factor_dense_top(s);
    for (int l = 0; l < (mf ? 0 : lD); l++) {
      const int c0 = S.level_ptr[l], c1 = S.level_ptr[l + 1];
      const int64_t entries = S.Lp[c1] - S.Lp[c0];
      const bool through_w = entries > 0 && long_rows[l];
      double *wf = through_w ? W.get() + (size_t)(half ^ 1) * w_half : nullptr, *wc = p1 > p0 ? W.get() + (size_t)half * w_half : nullptr;
      const int G = (through_w || p1 > p0) ? 64 : pick((double)(S.Rp[c1] - S.Rp[c0]) / (double)std::max(1, c1 - c0));
      if (G == 1) OQ_LAUNCH(k_ldl_diag_g<1>, dim3(blocks_for(c1 - c0)), dim3(kBlock), 0, s, c0, c1, Lx.get(), Rp.get(), Rj.get(), Rmap.get(), D.get(), Dinv.get(), status.get());
      else if (G == 4) OQ_LAUNCH(k_ldl_diag_g<4>, dim3(blocks_for((int64_t)(c1 - c0) * 4)), dim3(kBlock), 0, s, c0, c1, Lx.get(), Rp.get(), Rj.get(), Rmap.get(), D.get(), Dinv.get(), status.get());
      else if (G == 16) OQ_LAUNCH(k_ldl_diag_g<16>, dim3(blocks_for((int64_t)(c1 - c0) * 16)), dim3(kBlock), 0, s, c0, c1, Lx.get(), Rp.get(), Rj.get(), Rmap.get(), D.get(), Dinv.get(), status.get());
      else
      OQ_LAUNCH(k_ldl_diag_w, dim3(blocks_for((int64_t)(c1 - c0 + (p1 - p0)) * 64)), dim3(kBlock), 0, s, c0, c1, N, Lx.get(), Rp.get(), Rj.get(),
                Rmap.get(), D.get(), Dinv.get(), status.get(), wf, p0, p1, wc);
      p0 = p1 = 0;
      if (through_w) {
        OQ_LAUNCH(k_ldl_entries_w, dim3(blocks_for(entries * 64)), dim3(kBlock), 0, s, c0, c1, N, Lp.get(), Li.get(), Lx.get(),
                  Rp.get(), Rj.get(), Rmap.get(), wf, Dinv.get(), (const int *)Lcol.get());
        p0 = c0; p1 = c1; half ^= 1;
      } else if (entries > 0) {
        // long rows, too many columns for work rows: a wavefront per entry, bisection instead of the merge
        const double mean = (double)(S.Rp[c1] - S.Rp[c0]) / (double)std::max(1, c1 - c0);
        if (mean >= long_row_mean() && entries <= ((int64_t)1 << 18))  // few entries: the level is latency, not throughput
          OQ_LAUNCH(k_ldl_entries_bs<64>, dim3(blocks_for(entries * 64)), dim3(kBlock), 0, s, c0, c1, Lp.get(), Li.get(), Lx.get(), Rp.get(),
                    Rj.get(), Rmap.get(), D.get(), Dinv.get(), (const int *)Lcol.get());
        else if (mean >= long_row_mean() && entries * 16 < ((int64_t)1 << 31))
          OQ_LAUNCH(k_ldl_entries_bs<16>, dim3(blocks_for(entries * 16)), dim3(kBlock), 0, s, c0, c1, Lp.get(), Li.get(), Lx.get(), Rp.get(),
                    Rj.get(), Rmap.get(), D.get(), Dinv.get(), (const int *)Lcol.get());
        else
          OQ_LAUNCH(k_ldl_entries, dim3(blocks_for(entries)), dim3(kBlock), 0, s, c0, c1, Lp.get(), Li.get(), Lx.get(), Rp.get(),
                    Rj.get(), Rmap.get(), D.get(), Dinv.get(), (const int *)Lcol.get());
      }
    }
    if (p1 > p0)  // the last work rows
      OQ_LAUNCH(k_ldl_wrow, dim3(blocks_for((int64_t)(p1 - p0) * 64)), dim3(kBlock), 0, s, p0, p1, N, Lx.get(), Rp.get(), Rj.get(), Rmap.get(),
                D.get(), W.get() + (size_t)half * w_half, 0);
    if (factorizations == 0) e.setup_mark("  numeric: levels");
    if (kD) factor_dense_block();
    if (factorizations == 0) e.setup_mark("  numeric: dense block");
    if (sn) {
      const int64_t nf = T.Fp[N], big = std::max<int64_t>(N, nf);
      OQ_LAUNCH(k_sn_gather, dim3(blocks_for(big)), dim3(kBlock), 0, s, nf, sn_Fpos.get(), sn_Fx.get(), nf, sn_Gpos.get(), sn_Gx.get(), N,
                sn_piv.get(), Dinv.get(), sn_Dinv.get(), Lx.get());
      if (!mf)  // (the fronts leave the inverted blocks behind themselves)
        OQ_LAUNCH(k_sn_invert, dim3(T.count), dim3(kSnThreads), 0, s, sn_ptr.get(), sn_woff.get(), sn_wmap.get(), Lx.get(), sn_Wc.get(), sn_Wr.get());
      fold_blocks(s);
    }
    // (the CSR copy of the values serves the level-scheduled solves only)
    if (S.nnzL > 0 && !sn) OQ_LAUNCH(k_gather_csr, dim3(blocks_for(S.nnzL)), dim3(kBlock), 0, s, S.nnzL, Rmap.get(), Lx.get(), Rx.get());
    int st[2] = {0, 0};
    status.download(st, 2, s);
    e.sync();
    factorizations++;
    if (st[0]) return 4;
    if (st[1] != n) return 5;
    return 0;
  }

  void factor_dense_block() {
    hipStream_t s = e.stream;
    S0a.zero(s);
    const int bw = dense_batch();
    if (schur_ncols > 0) {
      const int64_t inside = S.Lp[N] - S.Lp[cD];
      OQ_LAUNCH(k_dense_init, dim3(blocks_for(std::max<int64_t>(inside, kD))), dim3(kBlock), 0, s, cD, N, ldD, Lp.get(), Li.get(), (const int *)Lcol.get(),
                Lx.get(), D.get(), S0a.get());
      const dim3 gu(ldD / 64, ldD / 64);
      for (int c0 = 0; c0 < schur_ncols; c0 += kGjK) {
        gjW.zero(s); gjC.zero(s);
        OQ_LAUNCH(k_dense_chunk, dim3(blocks_for((int64_t)kGjK * 64)), dim3(kBlock), 0, s, c0, schur_ncols, cD, ldD, (const int *)schur_cols.get(), Lp.get(),
                  Li.get(), Lx.get(), D.get(), gjW.get(), gjC.get());
        OQ_LAUNCH(k_gj_update, dim3((ldD / 64) * (ldD / 64 + 1) / 2), dim3(256), 0, s, ldD, -kGjK, S0a.get(), (const double *)gjT.get(), (const double *)gjW.get(),
                  (const double *)gjC.get(), (const int *)nullptr, (ldD / 64) * (ldD / 64 + 1) / 2, 0, (double *)nullptr, kD, status.get());
      }
    }
    for (int b0 = cD; b0 < N && schur_ncols == 0; b0 += bw) {
      const int b1 = std::min(N, b0 + bw);
      const dim3 gw(blocks_for((int64_t)(b1 - b0) * 64));
      OQ_LAUNCH(k_ldl_wrow, gw, dim3(kBlock), 0, s, b0, b1, N, Lx.get(), Rp.get(), Rj.get(), Rmap.get(), D.get(), W.get(), 1);
      const int64_t entries = S.Lp[b1] - S.Lp[b0];
      if (entries > 0)
        OQ_LAUNCH(k_dense_entries, dim3(blocks_for(entries * 64)), dim3(kBlock), 0, s, b0, b1, cD, kD, ldD, N, Lp.get(), Li.get(), Lx.get(),
                  Rp.get(), Rj.get(), Rmap.get(), W.get(), S0a.get());
      OQ_LAUNCH(k_dense_diag, gw, dim3(kBlock), 0, s, b0, b1, cD, kD, ldD, N, Lx.get(), Rp.get(), Rj.get(), Rmap.get(), D.get(), W.get(), S0a.get());
      OQ_LAUNCH(k_ldl_wrow, gw, dim3(kBlock), 0, s, b0, b1, N, Lx.get(), Rp.get(), Rj.get(), Rmap.get(), D.get(), W.get(), 0);
    }
    if (factorizations == 0) e.setup_mark("  numeric: Schur complement of the block");
    if (kD >= kDenseBlocked) {  // block sweeps of kGjK pivots on the matrix cores, in place
      gj_invert(kD, s);
      return;
    }
    double *cur = S0a.get(), *nxt = S0b.get();
    const dim3 gs(blocks_for((int64_t)kD * kD));
    int p = 0;
    for (; p + 1 < kD; p += 2) {
      OQ_LAUNCH(k_dense_sweep2, gs, dim3(kBlock), 0, s, kD, p, cur, nxt, status.get());
      std::swap(cur, nxt);
    }
    for (; p < kD; p++) {
      OQ_LAUNCH(k_dense_sweep, gs, dim3(kBlock), 0, s, kD, p, cur, nxt, status.get());
      std::swap(cur, nxt);
    }
    Sinv = cur;
  }

#define OQ_CHAIN_CASE(UU, LL)                                                                                                      \
  if (t.U == UU && t.L == LL) {                                                                                                     \
    if (fwd_) OQ_LAUNCH((k_fwd_chain_lds<UU, LL>), dim3(1), dim3(kChainThreads), 0, s, t.a, t.b, level_ptr.get(), Rsplit.get(),    \
                        Rp.get(), Rj.get(), Rx.get(), bp.get());                                                                    \
    else OQ_LAUNCH((k_bwd_chain_lds<UU, LL>), dim3(1), dim3(kChainThreads), 0, s, t.a, t.b, level_ptr.get(), Lp.get(),            \
                   Lsplit.get(), Li.get(), Lx.get(), bp.get());                                                                     \
    return;                                                                                                                         \
  }
  void launch_chain(const Step &t, bool fwd_, hipStream_t s) {
    OQ_CHAIN_CASE(2, 64) OQ_CHAIN_CASE(4, 64) OQ_CHAIN_CASE(8, 64)
    OQ_CHAIN_CASE(2, 16) OQ_CHAIN_CASE(4, 16) OQ_CHAIN_CASE(8, 16)
    OQ_CHAIN_CASE(2, 4) OQ_CHAIN_CASE(4, 4)
    throw Error(6, "internal: no chain kernel for this step");
  }
#undef OQ_CHAIN_CASE
  void launch_fwd_chain(const Step &t, hipStream_t s) { launch_chain(t, true, s); }
  void launch_bwd_chain(const Step &t, hipStream_t s) { launch_chain(t, false, s); }

  // skip_first_fwd / skip_last_bwd: those two level steps are folded into the kernels around the solve (fused_ends)
  // a level of many supernodes: a wavefront each, a quarter wavefront for the small ones that come first in the level
  // (k_sn_level_w); a level of few: a workgroup each
#define OQ_SN_LEVEL_W(LA, FWD, GS, JA, JB)                                                                                            \
  OQ_LAUNCH((k_sn_level_w<LA, FWD, GS>), dim3(((JB) - (JA) + 4 * (64 / GS) - 1) / (4 * (64 / GS))), dim3(kSnThreads), 0, s, JA, JB, sn_ptr.get(), \
            sn_woff.get(), FWD ? sn_Fp.get() : sn_Gp.get(), FWD ? sn_Fj.get() : sn_Gi.get(), FWD ? sn_Fx.get() : sn_Gx.get(),          \
            FWD ? sn_Wc.get() : sn_Wr.get(), sn_Dinv.get(), bp.get())
#define OQ_SN_LEVEL(LA, FWD, L)                                                                                                   \
  do {                                                                                                                            \
    const int cnt_ = T.lvl_ptr[L + 1] - T.lvl_ptr[L];                                                                             \
    if (cnt_ >= sn_wave_min_fixed) {                                                                                                \
      const int mid_ = T.lvl_ptr[L] + T.lvl_small[L];                                                                             \
      /* the single pivots the level starts with: forward nothing to do at level 0, backward a lane each */                       \
      const int ones_ = sn_singles ? T.lvl_single[L] : 0, first_ = T.lvl_ptr[L] + ((FWD && L > 0) ? 0 : ones_);                   \
      if (!FWD && ones_ > 0)                                                                                                      \
        OQ_LAUNCH(k_sn_single_bwd, dim3(blocks_for(ones_)), dim3(kBlock), 0, s, T.ptr[T.lvl_ptr[L]], ones_, sn_Gp.get(),          \
                  sn_Gi.get(), sn_Gx.get(), sn_Dinv.get(), bp.get());                                                             \
      /* a wide level is bound by rows in flight x latency, not by the latency of one row: a notch fewer lanes per row */         \
      if (mid_ > first_) OQ_SN_LEVEL_W((LA == 64 ? 16 : (LA == 16 ? 4 : LA)), FWD, 16, first_, mid_);                             \
      if (T.lvl_ptr[L + 1] > mid_ && !lvl_fold.empty() && lvl_fold[L])                                                            \
        OQ_LAUNCH((k_sn_level_wf<FWD, true>), dim3((T.lvl_ptr[L + 1] - mid_ + 3) / 4), dim3(kSnThreads), 0, s, mid_,               \
                  T.lvl_ptr[L + 1], sn_ptr.get(), sn_woff.get(), FWD ? sn_Fp.get() : sn_Gp.get(), FWD ? sn_Fj.get() : sn_Gi.get(),  \
                  FWD ? sn_Fx.get() : sn_Gx.get(), FWD ? sn_Wc.get() : sn_Wr.get(), sn_Dinv.get(), bp.get());                      \
      else if (T.lvl_ptr[L + 1] > mid_ && sn_flat && !(FWD && L == 0))                                                            \
        OQ_LAUNCH((k_sn_level_wf<FWD, false>), dim3((T.lvl_ptr[L + 1] - mid_ + 3) / 4), dim3(kSnThreads), 0, s, mid_,              \
                  T.lvl_ptr[L + 1], sn_ptr.get(), sn_woff.get(), FWD ? sn_Fp.get() : sn_Gp.get(), FWD ? sn_Fj.get() : sn_Gi.get(),  \
                  FWD ? sn_Fx.get() : sn_Gx.get(), FWD ? sn_Wc.get() : sn_Wr.get(), sn_Dinv.get(), bp.get());                      \
      else if (T.lvl_ptr[L + 1] > mid_) OQ_SN_LEVEL_W((LA == 64 ? 16 : (LA == 16 ? 4 : LA)), FWD, 64, mid_, T.lvl_ptr[L + 1]);     \
    } else if (sn_flat && !lvl_fold.empty() && lvl_fold[L])                                                                       \
      OQ_LAUNCH((k_sn_level_f<FWD, true>), dim3(cnt_), dim3(kSnThreads), 0, s, T.lvl_ptr[L], sn_ptr.get(), sn_woff.get(),          \
                FWD ? sn_Fp.get() : sn_Gp.get(), FWD ? sn_Fj.get() : sn_Gi.get(), FWD ? sn_Fx.get() : sn_Gx.get(),                  \
                FWD ? sn_Wc.get() : sn_Wr.get(), sn_Dinv.get(), bp.get());                                                        \
    else if (sn_flat)                                                                                                             \
      OQ_LAUNCH((k_sn_level_f<FWD, false>), dim3(cnt_), dim3(kSnThreads), 0, s, T.lvl_ptr[L], sn_ptr.get(), sn_woff.get(),         \
                FWD ? sn_Fp.get() : sn_Gp.get(), FWD ? sn_Fj.get() : sn_Gi.get(), FWD ? sn_Fx.get() : sn_Gx.get(),                  \
                FWD ? sn_Wc.get() : sn_Wr.get(), sn_Dinv.get(), bp.get());                                                        \
    else if (cnt_ >= kSnBusyLevel)  /* more workgroups than the device holds at once: rows in flight count, as above */         \
      OQ_LAUNCH((k_sn_level<(LA == 64 ? 16 : (LA == 16 ? 4 : LA)), FWD>), dim3(cnt_), dim3(kSnThreads), 0, s, T.lvl_ptr[L], sn_ptr.get(), \
                sn_woff.get(), FWD ? sn_Fp.get() : sn_Gp.get(), FWD ? sn_Fj.get() : sn_Gi.get(), FWD ? sn_Fx.get() : sn_Gx.get(),  \
                FWD ? sn_Wc.get() : sn_Wr.get(), sn_Dinv.get(), bp.get());                                                        \
    else                                                                                                                          \
      OQ_LAUNCH((k_sn_level<LA, FWD>), dim3(cnt_), dim3(kSnThreads), 0, s, T.lvl_ptr[L], sn_ptr.get(),                           \
                sn_woff.get(), FWD ? sn_Fp.get() : sn_Gp.get(), FWD ? sn_Fj.get() : sn_Gi.get(), FWD ? sn_Fx.get() : sn_Gx.get(),  \
                FWD ? sn_Wc.get() : sn_Wr.get(), sn_Dinv.get(), bp.get());                                                        \
  } while (0)
#define OQ_SN_TREE(FWD)                                                                                                           \
  if (sn_tree_threads == 512) OQ_SN_TREE_N(FWD, 512); else OQ_SN_TREE_N(FWD, 1024)
#define OQ_SN_TREE_N(FWD, NT_)                                                                                                    \
  OQ_LAUNCH((k_sn_tree<FWD, NT_>), dim3(sn_tree_grid ? sn_tree_grid : sn_j_end() - T.lvl_ptr[sn_tree_L0]), dim3(NT_), 0, s, sn_top_args(), T.lvl_ptr[sn_tree_L0], sn_j_end(), sn_ptr.get(),    \
            sn_woff.get(), FWD ? sn_Fp.get() : sn_Gp.get(), FWD ? sn_Fsplit.get() : sn_Gp.get(), FWD ? sn_Fj.get() : sn_Gi.get(),      \
            FWD ? sn_Fx.get() : sn_Gx.get(), FWD ? sn_Wc.get() : sn_Wr.get(), sn_Dinv.get(), sn_up.get(), sn_waits.get(),          \
            FWD ? sn_pending.get() : sn_ready.get(), sn_fault, bp.get(), sn_tree_grid ? sn_ticket.get() + (FWD ? 0 : 1) : (int *)nullptr)
  // the blocks of the supernodes that the wavefront form solves are kept folded (k_sn_fold / sn_block_fold); fixed at setup:
  // the levels are chosen by the same count the launches look at
  const bool sn_fold = !(getenv("OSQP_AMD_SNODE_FOLD") && atoi(getenv("OSQP_AMD_SNODE_FOLD")) == 0) &&
                       !(getenv("OSQP_AMD_SNODE_FLAT") && atoi(getenv("OSQP_AMD_SNODE_FLAT")) == 0);
  const int sn_wave_min_fixed = sn_wave_min();
  const bool sn_fold_wg = !(getenv("OSQP_AMD_SNODE_FOLD") && atoi(getenv("OSQP_AMD_SNODE_FOLD")) == 2);  // 2: wavefront-form levels only
  std::vector<char> lvl_fold;  // per level: the blocks of its larger supernodes are folded (decided once: a factor that falls
                               // back from k_sn_tree to one launch per level keeps reading them the way they are stored)
  void fold_blocks(hipStream_t s) {
    if (!sn || !sn_fold) return;
    if (lvl_fold.empty()) {
      lvl_fold.assign(T.nlev, 0);
      for (int L = 0; L < T.nlev; L++) {
        const int cnt = T.lvl_ptr[L + 1] - T.lvl_ptr[L], mid = T.lvl_ptr[L] + T.lvl_small[L];
        // the larger supernodes of a level in the wavefront form, every supernode of a level in the workgroup form
        // (k_sn_level_f); the levels from sn_tree_L0 on are solved by k_sn_tree whatever their count: packed triangles
        lvl_fold[L] = (cnt >= sn_wave_min_fixed ? T.lvl_ptr[L + 1] > mid : sn_fold_wg) && !(sn_tree && L >= sn_tree_L0);
      }
    }
    for (int L = 0; L < sn_l_end(); L++) {  // (the blocks of a dense top are never formed)
      const int cnt = T.lvl_ptr[L + 1] - T.lvl_ptr[L];
      const int mid = cnt >= sn_wave_min_fixed ? T.lvl_ptr[L] + T.lvl_small[L] : T.lvl_ptr[L];
      if (!lvl_fold[L]) continue;
      OQ_LAUNCH(k_sn_fold<true>, dim3(T.lvl_ptr[L + 1] - mid), dim3(64), 0, s, mid, sn_ptr.get(), sn_woff.get(), sn_Wc.get());
      OQ_LAUNCH(k_sn_fold<false>, dim3(T.lvl_ptr[L + 1] - mid), dim3(64), 0, s, mid, sn_ptr.get(), sn_woff.get(), sn_Wr.get());
    }
  }
  const bool sn_flat = !(getenv("OSQP_AMD_SNODE_FLAT") && atoi(getenv("OSQP_AMD_SNODE_FLAT")) == 0);  // k_sn_level_f / _wf
  const bool sn_singles = !(getenv("OSQP_AMD_SNODE_SINGLE") && atoi(getenv("OSQP_AMD_SNODE_SINGLE")) == 0);
  // OSQP_AMD_SNODE_WAVE_MIN (tests): supernodes in a level from which the wavefront / quarter-wavefront form is used
  static int sn_wave_min() { const char *v = getenv("OSQP_AMD_SNODE_WAVE_MIN"); return v ? atoi(v) : kSnWaveLevel; }
  int sn_l_end() const { return snd_K ? snd_L0 : T.nlev; }   // levels / supernodes below the dense top (all of them without one)
  int sn_j_end() const { return snd_K ? snd_J0 : T.count; }
  void run_supernodes() {
    hipStream_t s = e.stream;
    const bool tree = sn_tree && sn_tree_L0 < sn_l_end();
    const int plain = tree ? sn_tree_L0 : sn_l_end();  // levels [0, plain): one launch each; the rest: one launch per direction
    for (int L = 0; L < plain; L++) switch (sn_lanes_f[L]) {
      case 1: OQ_SN_LEVEL(1, true, L); break;
      case 4: OQ_SN_LEVEL(4, true, L); break;
      case 16: OQ_SN_LEVEL(16, true, L); break;
      default: OQ_SN_LEVEL(64, true, L); break;
    }
    if (tree && sn_tree_grid) HIP_CHECK(hipMemsetAsync(sn_ticket.get(), 0, 2 * sizeof(int), s));
    if (tree) { OQ_SN_TREE(true); }
    if (snd_K) solve_dense_top(s, tree);
    if (tree) { OQ_SN_TREE(false); }
    for (int L = plain - 1; L >= 0; L--) switch (sn_lanes_b[L]) {
      case 1: OQ_SN_LEVEL(1, false, L); break;
      case 4: OQ_SN_LEVEL(4, false, L); break;
      case 16: OQ_SN_LEVEL(16, false, L); break;
      default: OQ_SN_LEVEL(64, false, L); break;
    }
  }
#undef OQ_SN_LEVEL
#undef OQ_SN_LEVEL_W
#undef OQ_SN_TREE
#undef OQ_SN_TREE_N

  void run_steps(bool skip_first_fwd = false, bool skip_last_bwd = false) {
    hipStream_t s = e.stream;
    if (sn) { run_supernodes(); return; }
    for (size_t si = 0; si < fwd.size(); si++) {
      const Step &t = fwd[si];
      if (si == 0 && skip_first_fwd) continue;
      if (t.kind == 1) {
        const int c0 = S.level_ptr[t.a], c1 = S.level_ptr[t.b];
        if (t.G == kBlock) OQ_LAUNCH(k_fwd_far<kBlock>, dim3(c1 - c0), dim3(kBlock), 0, s, c0, c1, Rp.get(), Rsplit.get(), Rj.get(), Rx.get(), bp.get(), (double *)nullptr);
        else OQ_LAUNCH(k_fwd_far<64>, dim3(blocks_for((int64_t)(c1 - c0) * 64)), dim3(kBlock), 0, s, c0, c1, Rp.get(), Rsplit.get(), Rj.get(), Rx.get(), bp.get(), (double *)nullptr);
        launch_fwd_chain(t, s);
        continue;
      }
      dim3 grid(blocks_for((int64_t)(t.b - t.a) * t.G)), block(kBlock);
      switch (t.G) {
      case 1: OQ_LAUNCH(k_fwd_level<1>, grid, block, 0, s, t.a, t.b, Rp.get(), Rj.get(), Rx.get(), bp.get()); break;
      case 4: OQ_LAUNCH(k_fwd_level<4>, grid, block, 0, s, t.a, t.b, Rp.get(), Rj.get(), Rx.get(), bp.get()); break;
      case 16: OQ_LAUNCH(k_fwd_level<16>, grid, block, 0, s, t.a, t.b, Rp.get(), Rj.get(), Rx.get(), bp.get()); break;
      default: OQ_LAUNCH(k_fwd_level<64>, grid, block, 0, s, t.a, t.b, Rp.get(), Rj.get(), Rx.get(), bp.get()); break;
      }
    }
    if (kD) {  // x2 = S0^-1 (b2 - L21 y1)
      // the reduced right-hand side goes to x2, the product straight back into the block's slots of the solution
      OQ_LAUNCH(k_fwd_far<kBlock>, dim3(kD), dim3(kBlock), 0, s, cD, N, Rp.get(), Rsplit.get(), Rj.get(), Rx.get(), bp.get(), x2.get());
      if (dsP1.n) {  // the blocked inverse: symmetric, padded to whole tiles -- the product reads its lower triangle only
        const int nb = ldD / kDsT;
        OQ_LAUNCH(k_dense_apply_sym, dim3(nb * (nb + 1) / 2), dim3(256), 0, s, kD, ldD, nb, (const double *)Sinv, (const double *)x2.get(), dsP1.get(), dsP2.get());
        OQ_LAUNCH(k_dense_sym_reduce, dim3(nb), dim3(256), 0, s, kD, nb, (const double *)dsP1.get(), (const double *)dsP2.get(), bp.get() + cD, (const int *)nullptr);
      } else
        OQ_LAUNCH(k_dense_apply, dim3(blocks_for((int64_t)kD * 64)), dim3(kBlock), 0, s, kD, ldD, Sinv, x2.get(), bp.get() + cD);
    }
    for (size_t si = 0; si < bwd.size(); si++) {
      const Step &t = bwd[si];
      if (si + 1 == bwd.size() && skip_last_bwd) continue;
      if (t.kind == 1) {
        const int c0 = S.level_ptr[t.a], c1 = S.level_ptr[t.b];
        if (t.G == kBlock) OQ_LAUNCH(k_bwd_far<kBlock>, dim3(c1 - c0), dim3(kBlock), 0, s, c0, c1, Lsplit.get(), Lp.get(), Li.get(), Lx.get(), Dinv.get(), bp.get());
        else OQ_LAUNCH(k_bwd_far<64>, dim3(blocks_for((int64_t)(c1 - c0) * 64)), dim3(kBlock), 0, s, c0, c1, Lsplit.get(), Lp.get(), Li.get(), Lx.get(), Dinv.get(), bp.get());
        launch_bwd_chain(t, s);
        continue;
      }
      dim3 grid(blocks_for((int64_t)(t.b - t.a) * t.G)), block(kBlock);
      switch (t.G) {
      case 1: OQ_LAUNCH(k_bwd_level<1>, grid, block, 0, s, t.a, t.b, Lp.get(), Li.get(), Lx.get(), Dinv.get(), bp.get()); break;
      case 4: OQ_LAUNCH(k_bwd_level<4>, grid, block, 0, s, t.a, t.b, Lp.get(), Li.get(), Lx.get(), Dinv.get(), bp.get()); break;
      case 16: OQ_LAUNCH(k_bwd_level<16>, grid, block, 0, s, t.a, t.b, Lp.get(), Li.get(), Lx.get(), Dinv.get(), bp.get()); break;
      default: OQ_LAUNCH(k_bwd_level<64>, grid, block, 0, s, t.a, t.b, Lp.get(), Li.get(), Lx.get(), Dinv.get(), bp.get()); break;
      }
    }
  }

  // in place on b (length N, KKT order).  rho_inv != nullptr: the ADMM form with the z~ fix-up.
  void solve(double *b, const double *rho_inv) {
    hipStream_t s = e.stream;
    OQ_LAUNCH(k_perm_in, dim3(blocks_for(N)), dim3(kBlock), 0, s, N, vec_perm(), b, bp.get());
    run_steps();
    OQ_LAUNCH(k_perm_out, dim3(blocks_for(N)), dim3(kBlock), 0, s, N, n, vec_pinv(), bp.get(), rho_inv, b);
  }

  // level 1 forward / level 0 backward are plain level steps over short rows: they can ride with the iteration's ends
  bool can_fuse_fwd1() const {
    return !sn && nlev >= 2 && !fwd.empty() && fwd[0].kind == 0 && fwd[0].a == S.level_ptr[1] && fwd[0].b == S.level_ptr[2] && fwd[0].G <= 4;
  }
  bool can_fuse_bwd0() const {
    return !sn && !bwd.empty() && bwd.back().kind == 0 && bwd.back().a == 0 && bwd.back().b == S.level_ptr[1] && bwd.back().G <= 4;
  }

  // the whole iteration in two launches (k_direct2_fwd / k_direct2_bwd_update): a two-level factor with short rows
  bool can_fuse2() const {
    if (sn || nlev != 2 || kD != 0 || fwd.size() != 1 || bwd.size() != 2) return false;
    if (fwd[0].kind != 0 || fwd[0].G > 4 || bwd[0].kind != 0 || bwd[1].kind != 0 || bwd[1].G > 4) return false;
    return S.Lp[N] == S.Lp[S.level_ptr[1]];  // nothing above level 1: its backward step is the D^-1 scaling alone
  }

  double trisolve_bytes() const {
    // (a dense top over the supernodes: the entries of its rows that point into it and its blocks are not read, the lower triangle
    // of its inverse is -- once, not per direction)
    if (sn && snd_K) {
      const double blocks = 8.0 * (double)(T.woff[T.count] - T.woff[snd_J0]);
      return 2.0 * (12.0 * ((double)T.Fp[N] - snd_skipped) + 8.0 * (double)T.wdoubles - blocks + 8.0 * ((double)N + 1.0)) + 40.0 * (double)N +
             4.0 * (double)snd_K * (double)snd_K;
    }
    if (sn) return 2.0 * (12.0 * (double)T.Fp[N] + 8.0 * (double)T.wdoubles + 8.0 * ((double)N + 1.0)) + 40.0 * (double)N;
    return 2.0 * (12.0 * (double)S.nnzL + 4.0 * ((double)N + 1.0)) + 40.0 * (double)N;
  }
};

// k_direct_update that also leaves the right-hand side of the NEXT iteration behind (k_direct_rhs's arithmetic on the values it
// has just formed, into the slot it has just read): inside a chunk graph iteration k + 1 follows iteration k with nothing in
// between, and its own pass over x, q, z, y for the right-hand side -- one launch, 64 MB on control-1e6 -- is not needed
__global__ __launch_bounds__(kBlock) void k_direct_update_rhs(int n, int m, double alpha, double sigma, const int *__restrict__ pinv,
                                                              double *__restrict__ bp, const double *__restrict__ q,
                                                              const double *__restrict__ rho, const double *__restrict__ rho_inv,
                                                              const double *__restrict__ l, const double *__restrict__ u,
                                                              double *__restrict__ x, double *__restrict__ z, double *__restrict__ y,
                                                              double *__restrict__ delta_x, double *__restrict__ delta_y) {
  int o = blockIdx.x * kBlock + threadIdx.x;
  if (o < n) {
    const int slot = pinv[o];
    double xp = x[o];
    double xn = alpha * bp[slot] + (1.0 - alpha) * xp;
    x[o] = xn;
    delta_x[o] = xn - xp;
    bp[slot] = sigma * xn - q[o];
  } else if (o < n + m) {
    int j = o - n;
    const int slot = pinv[o];
    double zp = z[j], yj = y[j], ri = rho_inv[j];
    double zt = (zp - ri * yj) + ri * bp[slot];
    double zh = alpha * zt + (1.0 - alpha) * zp;
    double zn = fmin(fmax(zh + ri * yj, l[j]), u[j]);
    z[j] = zn;
    double dy = rho[j] * (zh - zn);
    delta_y[j] = dy;
    const double yn = yj + dy;
    y[j] = yn;
    bp[slot] = zn - ri * yn;
  }
}

struct Direct : Linsys {
  Engine &e;
  std::unique_ptr<LdlFactor> F;
  explicit Direct(Engine &en) : e(en) {}
  int kind() const override { return 0; }
  int solve(double *xz, double) override { F->solve(xz, e.rho_inv.get()); return 0; }
  int fused_step() override {
    hipStream_t s = e.stream;
    const int N = e.n + e.m;
    static const bool fuse_ends = !(getenv("OSQP_AMD_FUSE_ENDS") && atoi(getenv("OSQP_AMD_FUSE_ENDS")) == 0);
    const bool f1 = fuse_ends && F->can_fuse_fwd1(), b0 = fuse_ends && F->can_fuse_bwd0();
    static const bool fuse2 = !(getenv("OSQP_AMD_FUSE2") && atoi(getenv("OSQP_AMD_FUSE2")) == 0);
    if (fuse_ends && fuse2 && F->can_fuse2()) {
      const int l1 = F->S.level_ptr[1];
      OQ_LAUNCH(k_direct2_fwd, dim3(blocks_for(N - l1)), dim3(kBlock), 0, s, e.n, N, e.st.sigma, F->perm.get(), l1, F->Rp.get(), F->Rj.get(),
                F->Rx.get(), F->Dinv.get(), e.x.get(), e.q.get(), e.z.get(), e.rho_inv.get(), e.y.get(), F->bp.get());
      OQ_LAUNCH(k_direct2_bwd_update, dim3(blocks_for(N)), dim3(kBlock), 0, s, e.n, e.m, e.st.sigma, e.st.alpha, F->pinv.get(), l1, F->Lp.get(),
                F->Li.get(), F->Lx.get(), F->Dinv.get(), F->bp.get(), e.q.get(), e.rho.get(), e.rho_inv.get(), e.l.get(), e.u.get(),
                e.x.get(), e.z.get(), e.y.get(), e.dx.get(), e.dy.get());
      return 0;
    }
    if (f1)
      OQ_LAUNCH(k_direct_rhs_fwd1, dim3(blocks_for(N)), dim3(kBlock), 0, s, e.n, e.m, e.st.sigma, F->pinv.get(), F->perm.get(),
                F->S.level_ptr[1], F->S.level_ptr[2], F->Rp.get(), F->Rj.get(), F->Rx.get(), e.x.get(), e.q.get(), e.z.get(),
                e.rho_inv.get(), e.y.get(), F->bp.get());
    else if (!rhs_left)
      OQ_LAUNCH(k_direct_rhs, dim3(blocks_for(N)), dim3(kBlock), 0, s, e.n, e.m, e.st.sigma, F->vec_pinv(), e.x.get(), e.q.get(),
                e.z.get(), e.rho_inv.get(), e.y.get(), F->bp.get());
    rhs_left = false;
    F->run_steps(f1, b0);
    if (b0)
      OQ_LAUNCH(k_direct_bwd0_update, dim3(blocks_for(N)), dim3(kBlock), 0, s, e.n, e.m, e.st.alpha, F->pinv.get(), F->S.level_ptr[1],
                F->Lp.get(), F->Li.get(), F->Lx.get(), F->Dinv.get(), F->bp.get(), e.rho.get(), e.rho_inv.get(), e.l.get(), e.u.get(),
                e.x.get(), e.z.get(), e.y.get(), e.dx.get(), e.dy.get());
    else if (next_follows && !f1 && leave_rhs) {
      OQ_LAUNCH(k_direct_update_rhs, dim3(blocks_for(N)), dim3(kBlock), 0, s, e.n, e.m, e.st.alpha, e.st.sigma, F->vec_pinv(), F->bp.get(),
                e.q.get(), e.rho.get(), e.rho_inv.get(), e.l.get(), e.u.get(), e.x.get(), e.z.get(), e.y.get(), e.dx.get(), e.dy.get());
      rhs_left = true;  // (only ever true between two steps of one capture: the last step of a chunk has next_follows = false)
    } else
      OQ_LAUNCH(k_direct_update, dim3(blocks_for(N)), dim3(kBlock), 0, s, e.n, e.m, e.st.alpha, F->vec_pinv(), F->bp.get(),
                e.rho.get(), e.rho_inv.get(), e.l.get(), e.u.get(), e.x.get(), e.z.get(), e.y.get(), e.dx.get(), e.dy.get());
    return 0;
  }
  bool rhs_left = false;  // the previous step of this capture left this step's right-hand side in the factor's vector
  const bool leave_rhs = !(getenv("OSQP_AMD_DIRECT_LEAVE_RHS") && atoi(getenv("OSQP_AMD_DIRECT_LEAVE_RHS")) == 0);
  // a supernode waited 200 ms for its children: the factor goes back to one launch per level (no waiting inside a
  // kernel); 6 tells the engine that the iterations since the last test cannot be trusted (TreeFault: the solve restarts)
  void invalidate() override { rhs_left = false; }
  int flush() override {
    rhs_left = false;
    if (!F->faulted()) return 0;
    *F->sn_fault_host = 0;
    F->sn_tree = false;
    e.drop_chunk_graph();
    return 6;
  }
  int update_rho() override { return F->refactor(e.rho_inv.get()); }
  int update_matrices() override { return F->refactor(e.rho_inv.get()); }
  double nnzL() const override { return (double)F->S.nnzL; }
  double levels() const override { return (double)F->nlev; }
  double supernode_levels() const override { return F->sn ? (double)F->T.nlev : 0.0; }
  double multifrontal() const override { return F->mf ? 1.0 : 0.0; }
  double lean_setup() const override { return F->lean_built ? 1.0 : 0.0; }
  double dense_block() const override { return (double)(F->kD ? F->kD : F->snd_K); }
  double trisolve_bytes() const override { return F->trisolve_bytes(); }
  double factorizations() const override { return (double)F->factorizations; }
  float time_solve(int reps) override {
    hipEvent_t a, b;
    HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b));
    F->run_steps();
    HIP_CHECK(hipEventRecord(a, e.stream));
    for (int i = 0; i < reps; i++) F->run_steps();
    HIP_CHECK(hipEventRecord(b, e.stream));
    HIP_CHECK(hipEventSynchronize(b));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return ms / (float)reps;
  }
};

double factor_flops_limit() {
  if (const char *v = getenv("OSQP_AMD_FLOPS_LIMIT")) return atof(v);
  return 5e8;
}
int level_limit() {
  if (const char *v = getenv("OSQP_AMD_LEVEL_LIMIT")) return atoi(v);
  return 3000;
}

int64_t factor_limit(const Engine &e, bool forced) {
  if (const char *v = getenv("OSQP_AMD_NNZL_LIMIT")) return atoll(v);
  (void)e;
  return forced ? (int64_t)2000000000LL : (int64_t)400000000LL;  // 4e8 entries = 9.6 GB of L (both copies)
}

}  // namespace

std::unique_ptr<Linsys> make_direct(Engine &e, int *err) {
  *err = 0;
  std::vector<int> ident(e.m);
  for (int i = 0; i < e.m; i++) ident[i] = i;
  std::unique_ptr<Direct> d(new Direct(e));
  const bool forced = e.st.linsys_solver == AMD_DIRECT_SOLVER;
  d->F.reset(new LdlFactor(e, ident, e.m, e.st.sigma, 0.0, factor_limit(e, forced), forced ? 0.0 : factor_flops_limit()));
  if (d->F->S.too_large) { *err = -1; return nullptr; }
  // auto mode also gives the problem to PCG when the factorisation itself would take too long
  // (sum of squared column counts ~ multiply-adds of one numeric factorisation; it is redone at every rho update)
  // or when the level schedule below the dense top block is so deep that the triangular solves are a serial chain
  // (measured, n = m = 5000 with 10 per row: 4623 levels, 60 ms per iteration -- PCG needs a fraction of a millisecond there)
  // (round 4: a dense top block is priced by what inverting it costs now -- 31 ms for 6000 pivots -- not by its n^3 / 3 in the
  // sum: a dense P no longer sends a problem to PCG once its analysis has been paid for; equality_qp: 9 k it/s instead of 260)
  if (e.st.linsys_solver != AMD_DIRECT_SOLVER && (d->F->flops_below_dense_block() > factor_flops_limit() || d->F->solve_levels() > level_limit())) { *err = -1; return nullptr; }
  int rc = d->F->refactor(e.rho_inv.get());
  if (rc) { *err = rc; return nullptr; }
  return std::unique_ptr<Linsys>(d.release());
}

// ---------------------------------------------------------------------------
// Polish: guess the active constraints from (z, y), solve the
// equality-constrained QP on them through a delta-regularised KKT system with
// iterative refinement, accept if the residuals improve.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_gather_idx(int k, const int *__restrict__ idx, const double *__restrict__ in, double *__restrict__ out) {
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < k) out[i] = in[idx[i]];
}
__global__ __launch_bounds__(kBlock) void k_scatter_idx(int k, const int *__restrict__ idx, const double *__restrict__ in, double *__restrict__ out) {
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < k) out[idx[i]] = in[i];
}
__global__ __launch_bounds__(kBlock) void k_normal_cone(int m, double *__restrict__ z, double *__restrict__ y, const double *__restrict__ l,
                                                        const double *__restrict__ u) {
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= m) return;
  double s = z[i] + y[i];
  double zn = fmin(fmax(s, l[i]), u[i]);
  z[i] = zn; y[i] = s - zn;
}

int polish_run(Engine &e) {
  // the reduced KKT system is assembled from the CSR arrays, which a compact workspace has released: iterative form (pcg.hip)
  // OSQP_AMD_POLISH_ITERATIVE=1 (tests): the iterative form wherever the indirect back-end runs, so that it can be compared
  // with a factorisation-based polish on problems small enough to have one
  const bool force_iterative = getenv("OSQP_AMD_POLISH_ITERATIVE") && atoi(getenv("OSQP_AMD_POLISH_ITERATIVE")) != 0;
  if (e.compact || e.comm || (force_iterative && e.lin && e.lin->kind() == 2)) return polish_run_pcg(e);  // a row block has no reduced KKT matrix either
  hipStream_t s = e.stream;
  const int n = e.n, m = e.m;
  OSQPInfo *info = e.ws->info;
  // active sets from the ADMM solution (host decides; m doubles x 4 come back once)
  std::vector<double> hz(m), hy(m), hl(m), hu(m);
  e.z.download(hz.data(), m, s); e.y.download(hy.data(), m, s); e.l.download(hl.data(), m, s); e.u.download(hu.data(), m, s);
  e.sync();
  std::vector<int> ind_low, ind_upp, row_map(m, -1);
  for (int i = 0; i < m; i++) if (hz[i] - hl[i] < -hy[i]) ind_low.push_back(i);
  for (int i = 0; i < m; i++) if (hu[i] - hz[i] < hy[i]) ind_upp.push_back(i);
  const int n_low = (int)ind_low.size(), n_upp = (int)ind_upp.size(), mr = n_low + n_upp;
  // rows that are both lower- and upper-active keep their lower slot in the reduced matrix (as the CPU statement does)
  for (int k = 0; k < n_upp; k++) row_map[ind_upp[k]] = n_low + k;
  for (int k = 0; k < n_low; k++) row_map[ind_low[k]] = k;
  std::vector<int> act(mr);
  for (int k = 0; k < n_low; k++) act[k] = ind_low[k];
  for (int k = 0; k < n_upp; k++) act[n_low + k] = ind_upp[k];

  LdlFactor F(e, row_map, mr, e.st.delta, -e.st.delta, 400000000LL);
  if (F.S.too_large) return e.lin && e.lin->kind() == 2 ? polish_run_pcg(e) : -1;  // no factor of the reduced system that fits: iterate instead
  if (F.refactor(nullptr) != 0) return -1;
  const int nr = n + mr;
  DevBuf<double> rhs_red(nr), sol(nr), rhs(nr), yfull(m), Axv(m), px(n), pz(m), py(m);
  DevBuf<int> dact(mr ? mr : 1);
  dact.upload(act.data(), mr, s);
  // rhs_red = [-q ; l_low ; u_upp]
  vec_copy(rhs_red.get(), e.q.get(), n, s);
  vec_scale(rhs_red.get(), -1.0, n, s);
  if (n_low) OQ_LAUNCH(k_gather_idx, dim3(blocks_for(n_low)), dim3(kBlock), 0, s, n_low, dact.get(), e.l.get(), rhs_red.get() + n);
  if (n_upp) OQ_LAUNCH(k_gather_idx, dim3(blocks_for(n_upp)), dim3(kBlock), 0, s, n_upp, dact.get() + n_low, e.u.get(), rhs_red.get() + n + n_low);
  // solve, then iterative refinement against the unregularised matrix: rhs = rhs_red - [P x + Ared' y ; Ared x]
  DevBuf<double> ared_x(mr ? mr : 1);
  vec_copy(sol.get(), rhs_red.get(), nr, s);
  F.solve(sol.get(), nullptr);
  for (int it = 0; it < e.st.polish_refine_iter; it++) {
    vec_copy(rhs.get(), rhs_red.get(), nr, s);
    spmv(e.Pf, sol.get(), px.get(), nullptr, 0.0, 0.0, nullptr, s);
    vec_axpy(rhs.get(), -1.0, px.get(), n, s);
    if (mr > 0) {
      yfull.zero(s);
      OQ_LAUNCH(k_scatter_idx, dim3(blocks_for(mr)), dim3(kBlock), 0, s, mr, dact.get(), sol.get() + n, yfull.get());
      spmv(e.At, yfull.get(), px.get(), nullptr, 0.0, 0.0, nullptr, s);
      vec_axpy(rhs.get(), -1.0, px.get(), n, s);
      spmv(e.A, sol.get(), Axv.get(), nullptr, 0.0, 0.0, nullptr, s);
      OQ_LAUNCH(k_gather_idx, dim3(blocks_for(mr)), dim3(kBlock), 0, s, mr, dact.get(), Axv.get(), ared_x.get());
      vec_axpy(rhs.get() + n, -1.0, ared_x.get(), mr, s);
    }
    F.solve(rhs.get(), nullptr);
    vec_axpy(sol.get(), 1.0, rhs.get(), nr, s);
  }
  // polished (x, z, y)
  vec_copy(px.get(), sol.get(), n, s);
  if (m > 0) spmv(e.A, px.get(), pz.get(), nullptr, 0.0, 0.0, nullptr, s);
  py.zero(s);
  if (mr > 0) OQ_LAUNCH(k_scatter_idx, dim3(blocks_for(mr)), dim3(kBlock), 0, s, mr, dact.get(), sol.get() + n, py.get());
  if (m > 0) OQ_LAUNCH(k_normal_cone, dim3(blocks_for(m)), dim3(kBlock), 0, s, m, pz.get(), py.get(), e.l.get(), e.u.get());
  // residuals at the polished point (same kernels as update_info)
  spmv(e.A, px.get(), e.Ax.get(), nullptr, 0.0, 0.0, nullptr, s);
  spmv(e.Pf, px.get(), e.Px_.get(), nullptr, 0.0, 0.0, nullptr, s);
  if (m > 0) spmv(e.At, py.get(), e.Aty.get(), nullptr, 0.0, 0.0, nullptr, s);
  residual_norms(n, m, px.get(), pz.get(), e.Ax.get(), e.Px_.get(), e.Aty.get(), e.q.get(), e.Dinv.get(), e.Einv.get(),
                 e.slots.get(), e.partials.get(), s);
  e.fetch_slots(0, 16, (1u << S_XPX) | (1u << S_QX));
  const double *r = e.h_slots;
  const bool uns = e.st.scaling && !e.st.scaled_termination;
  double pol_pri = m == 0 ? 0.0 : (uns ? r[S_PRI_UNS] : r[S_PRI]);
  double pol_dua = uns ? e.cinv * r[S_DUA_UNS] : r[S_DUA];
  double pol_obj = 0.5 * r[S_XPX] + r[S_QX];
  if (e.st.scaling) pol_obj *= e.cinv;
  bool ok = (pol_pri < info->pri_res && pol_dua < info->dua_res) || (pol_pri < info->pri_res && info->dua_res < 1e-10) ||
            (pol_dua < info->dua_res && info->pri_res < 1e-10);
  if (!ok) return -1;
  info->obj_val = pol_obj; info->pri_res = pol_pri; info->dua_res = pol_dua;
  vec_copy(e.x.get(), px.get(), n, s);
  vec_copy(e.z.get(), pz.get(), m, s);
  vec_copy(e.y.get(), py.get(), m, s);
  return 1;
}

}  // namespace oq
