// symbolic.hpp -- host-side symbolic analysis of the quasi-definite KKT matrix
//   K = [P + sigma I, A_s'; A_s, -diag(d)]      (A_s = selected rows of A)
// for the direct back-end (row K2 of SURVEY.md section 8a): fill-reducing
// ordering, elimination tree, level schedule, pattern of L in both CSC and CSR
// form, and the scatter maps from the caller's nnz order into L / D.
//
// The numeric work (LDL' and the triangular solves) runs on the device
// (direct.hip); this file only produces index arrays.
#pragma once
#include <cstdint>
#include <memory>
#include <vector>

namespace oq {

struct HostCsc;

// The rows of the pattern of L as a lean analysis leaves them, in the order its tree walk took them (the numbering of the
// ordering, not the final one): the r-th row of the walk is row rowid[r] of L; block b holds the rows [first[b], first[b + 1])
// of the walk back to back, the columns of every row (final ids) in the order they were met (unsorted).
struct LeanRows {
  std::vector<int> first, rowid;
  std::vector<std::vector<int>> cols;
};

struct Symbolic {
  int n = 0, mr = 0, N = 0;          // variables, selected constraint rows, n + mr
  std::vector<int> perm, pinv;       // perm[k] = KKT index (0..N) eliminated k-th; pinv = inverse
  // strictly-lower L, column-compressed, rows ascending within a column
  std::vector<int64_t> Lp;
  std::vector<int> Li;
  // the same pattern row-compressed: row k lists columns j < k ascending, Rmap = position in Li/Lx
  std::vector<int64_t> Rp, Rmap;
  std::vector<int> Rj;
  // level schedule: pivots are numbered so that level l is the index range [level_ptr[l], level_ptr[l+1])
  std::vector<int> level_ptr;
  std::vector<int> parent;           // elimination tree of the final numbering (-1: root)
  // scatter maps: target >= 0 is a position in Lx, target < 0 encodes the diagonal entry D[-target-1]
  std::vector<int64_t> PtoL;         // one per nnz of triu(P) (caller's CSC order)
  std::vector<int64_t> AtoL;         // one per nnz of A (caller's CSC order); INT64_MIN for rows not selected
  int64_t nnzL = 0;
  double flops = 0.0;                // sum of squared column counts of L (cost model of one numeric factorisation)
  bool too_large = false;            // predicted factor exceeds the limit: nothing else is filled in
  // A LEAN analysis (symbolic_analyse(..., lean = true)) stops once the numbering, the tree, the levels, Lp / Rp (counts) and
  // the unsorted rows of the pattern are known: Li, Rj, Rmap, PtoL, AtoL stay empty -- a supernodal factor builds them on
  // the device from the rows (direct.hip LdlFactor::lean_device) -- until symbolic_complete fills them in on the host.
  bool lean = false;
  bool no_host_maps = false;         // set by the caller BEFORE the analysis: leave PtoL / AtoL empty (the device builds them by one
                                     // bisection per entry of K, direct.hip device_scatter_maps -- 16e6 cache-missing searches on the
                                     // host were the largest piece of the setup of a dense-P problem)
  std::shared_ptr<LeanRows> lean_rows;
  std::shared_ptr<void> lean_state;
};

// row_map[i] = index of constraint row i inside the reduced block (0..mr-1) or -1 when the row is left out
// (identity for the ADMM KKT system; the active-set selection for polish).
// nnzL_limit / flops_limit (0 = none): stop and set too_large as soon as the factor is known to have more entries
// than the first or a sum of squared column counts above the second (checked while the ordering runs, so that a
// hopeless problem is handed to the indirect back-end without paying for the whole analysis).
// ordering: 0 approximate minimum degree, 1 nested dissection by level structures (banded / chain-like graphs),
// 2 approximate minimum degree with updated nodes queued behind their equals (multiple-elimination tie-breaking:
// flatter elimination trees on graphs with many degree-1 nodes, e.g. bound constraints).
// Depth of a breadth-first level structure of the KKT graph (two passes towards a pseudo-peripheral root, component of node 0):
// large on banded / multi-stage / grid-like problems (the graph is long), ~log N on random sparsity.  What the direct back-end
// looks at to decide, on large problems, whether nested dissection should be the FIRST ordering it tries.
int kkt_graph_depth(const HostCsc &P, const HostCsc &A, const std::vector<int> &row_map, int mr);

void symbolic_analyse(const HostCsc &P, const HostCsc &A, const std::vector<int> &row_map, int mr, int64_t nnzL_limit,
                      double flops_limit, int ordering, Symbolic &out, bool lean = false);
void symbolic_complete(Symbolic &S);  // the rest of a lean analysis on the host (identical arrays to a full analysis)

// Supernodes for the triangular solves (direct.hip, k_sn_*): sets of at most `smax` pivots whose diagonal block of L is
// inverted once per factorisation, so that a solve needs one step per supernode instead of one per pivot.  Two kinds:
// a whole subtree of the elimination tree with at most smax nodes (the leaves of a nested-dissection tree), and a
// segment of a path of the tree above those (a separator, cut into pieces of smax).  Everything the solves index is
// in SLOT order: slot q = position of the pivot in `piv`, supernodes numbered level by level (a supernode's rows only
// reference slots of lower levels outside its own block).
struct Supernodes {
  int count = 0, nlev = 0, smax = 0;
  std::vector<int> lvl_ptr;   // level L = supernodes [lvl_ptr[L], lvl_ptr[L+1])
  std::vector<int> ptr;       // supernode J = slots [ptr[J], ptr[J+1])
  std::vector<int> piv;       // slot -> pivot (ascending inside a supernode)
  std::vector<int> slot;      // pivot -> slot
  std::vector<int> up;        // supernode holding the elimination-tree parent of its top node (-1: a root)
  std::vector<int> waits;     // number of supernodes of level >= 1 that have it as `up`
  static constexpr int kSmall = 16;  // supernodes of at most this many pivots are numbered first inside their level
  std::vector<int> lvl_small;        // per level: how many of its supernodes are that small
  std::vector<int> lvl_single;       // per level: how many of those have ONE pivot (they come first of all)
  int64_t wdoubles = 0;        // entries of all blocks together (woff[count] includes the room the folded form of an odd triangle needs)
  std::vector<int64_t> woff;  // offset of the block of supernode J in the W arrays: its lower triangle, packed, s (s + 1) / 2 entries (count + 1 offsets)
  std::vector<int64_t> wmap;  // per block entry a (a + 1) / 2 + b, b <= a: position in Lx of L(slot a, slot b) for a > b, -1 if not in the pattern (and on the diagonal)
  // the entries of L outside the diagonal blocks, by row (forward solve) and by column (backward solve), rows and
  // columns in slot order; the index stored is the slot of the other end, pos the position of the value in Lx
  std::vector<int64_t> Fp, Fpos, Gp, Gpos;   // every list ascending by slot
  std::vector<int64_t> Fsplit;               // per row: first entry that points at a slot of level >= 1
  std::vector<int> Fj, Gi;
  double flops = 0.0;         // multiply-adds of one forward + backward solve
};
// with_wmap = false leaves `wmap` empty (the multifrontal factorisation inverts the blocks inside its fronts and never
// looks at it); supernode_wmap fills it in afterwards
// partition_only: stop after the partition (levels, slots, block offsets), with Fp / Gp as upper bounds from the row and
// column counts -- all a lean analysis can give; the lists are then built on the device
void build_supernodes(const Symbolic &S, int smax, Supernodes &out, bool with_wmap = true, bool partition_only = false);
void supernode_wmap(const Symbolic &S, Supernodes &out);

// The dense top block of the level schedule: the longest suffix of the top chain (levels of at most chain_rows pivots)
// with at most dense_max pivots, taken when at least an eighth of its lower triangle is in the pattern of L or when
// one dense product costs less than its levels.  lD = first level of the block, cD its first pivot, kD = N - cD (0: none).
// is_dense (may be null): the block was taken because an eighth of it is in the pattern (a dense P, dense rows), not as the cheaper
// form of a block-sparse top chain
void choose_dense_top(const Symbolic &S, int chain_rows, int dense_max, int dense_sparse_max, int dense_min, int &lD, int &cD, int &kD,
                      bool *is_dense = nullptr);

// Rough time of one forward + backward solve by the level schedule: a launch per wide level, a chain step per narrow
// level below the dense top block, the bytes of L at 2 TB/s, the dense block's product.  Used to compare orderings
// (chain_level_us = 1.4, the best case; the block guessed from dense_max / dense_min).
double level_solve_cost_us(const Symbolic &Y, int chain_rows, int dense_max, int dense_min, double chain_level_us);
// ... with the block that was actually chosen and 2.4 us per chain level (measured on deep nested-dissection trees)
double level_solve_cost_us(const Symbolic &Y, int chain_rows, int lD, int kD);
// ... and by supernodes: two launches per direction, inside them one hand-over between workgroups per level; the
// slowest workgroup of each level walks its entries outside the blocks and its block with `threads` threads.
// (levels >= 0: only the first `levels` levels -- what is left below a dense top over the partition, direct_sndense_kernels.hpp)
double supernode_solve_cost_us(const Supernodes &T, int threads, int levels = -1);
bool supernodes_pay(const Symbolic &S, const Supernodes &T, int chain_rows, int lD, int kD, int threads);

}  // namespace oq
