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
#include <vector>

namespace oq {

struct HostCsc;

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
  // scatter maps: target >= 0 is a position in Lx, target < 0 encodes the diagonal entry D[-target-1]
  std::vector<int64_t> PtoL;         // one per nnz of triu(P) (caller's CSC order)
  std::vector<int64_t> AtoL;         // one per nnz of A (caller's CSC order); INT64_MIN for rows not selected
  int64_t nnzL = 0;
  double flops = 0.0;                // sum of squared column counts of L (cost model of one numeric factorisation)
  bool too_large = false;            // predicted factor exceeds the limit: nothing else is filled in
};

// row_map[i] = index of constraint row i inside the reduced block (0..mr-1) or -1 when the row is left out
// (identity for the ADMM KKT system; the active-set selection for polish).
// nnzL_limit / flops_limit (0 = none): stop and set too_large as soon as the factor is known to have more entries
// than the first or a sum of squared column counts above the second (checked while the ordering runs, so that a
// hopeless problem is handed to the indirect back-end without paying for the whole analysis).
// ordering: 0 approximate minimum degree, 1 nested dissection by level structures (banded / chain-like graphs),
// 2 approximate minimum degree with updated nodes queued behind their equals (multiple-elimination tie-breaking:
// flatter elimination trees on graphs with many degree-1 nodes, e.g. bound constraints).
void symbolic_analyse(const HostCsc &P, const HostCsc &A, const std::vector<int> &row_map, int mr, int64_t nnzL_limit,
                      double flops_limit, int ordering, Symbolic &out);

}  // namespace oq
