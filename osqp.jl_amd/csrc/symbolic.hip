// symbolic.hip -- host-side symbolic analysis for the direct KKT back-end (see symbolic.hpp).
// Plain C++ (no device code): ordering + elimination tree + level schedule + pattern of L + scatter maps.
#include "symbolic.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <exception>
#include <functional>
#include <memory>
#include <cmath>
#include <numeric>
#include <thread>

#include "engine.hpp"

namespace oq {

namespace {

// Host threads for the passes of the analysis that are independent per row / column / subtree (round 4: at 2.7e6 pivots and
// 6.7e7 entries of L -- control-1e6 -- the single-threaded analysis was 4.7 s of a 7.2 s setup).  Small problems stay on one.
inline int host_threads(int64_t work) {
  if (const char *v = getenv("OSQP_AMD_HOST_THREADS")) return std::max(1, std::min(64, atoi(v)));  // tests: the threaded passes on small problems
  if (work < ((int64_t)1 << 20)) return 1;
  return (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
}
// blocks [b0, b1) of [0, n) dealt to threads through a counter; fn(block index, begin, end, thread index)
template <typename F>
void parallel_blocks(int n, int nblocks, int nthreads, F fn) {
  nblocks = std::max(1, std::min(nblocks, std::max(n, 1)));
  auto range = [&](int b) { return std::pair<int, int>((int)((int64_t)n * b / nblocks), (int)((int64_t)n * (b + 1) / nblocks)); };
  if (nthreads <= 1) { for (int b = 0; b < nblocks; b++) { auto r = range(b); fn(b, r.first, r.second, 0); } return; }
  std::atomic<int> next{0};
  std::vector<std::thread> pool;
  std::vector<std::exception_ptr> failed(nthreads);  // an exception of a worker (out of memory) is the caller's, not std::terminate's
  for (int t = 0; t < nthreads; t++)
    pool.emplace_back([&, t]() {
      try {
        for (int b = next++; b < nblocks; b = next++) { auto r = range(b); fn(b, r.first, r.second, t); }
      } catch (...) { failed[t] = std::current_exception(); next = nblocks; }
    });
  for (auto &th : pool) th.join();
  for (auto &f : failed) if (f) std::rethrow_exception(f);
}

// ---------------------------------------------------------------------------
// Fill-reducing ordering: approximate minimum degree on a quotient graph.
// Nodes are variables until eliminated; an eliminated node becomes an "element"
// whose member list is the clique it created.  Degrees are the usual upper bound
// |A_i| + |L_p \ i| + sum_e |L_e \ L_p|; elements swallowed by a new one are dropped.
// ---------------------------------------------------------------------------
struct MinDegree {
  int N;
  std::vector<std::vector<int>> var_adj, elem_adj, members;
  std::vector<char> state;  // 0 variable, 1 element, 2 dead
  std::vector<int> degree, bucket_head, next, prev, stamp, wstamp, wcount;
  std::vector<int> extra;  // neighbours kept out of the graph (dense nodes, eliminated last): a constant part of every degree
  // budget: the clique sizes are the column counts of L for this ordering, so the run can stop as soon as the
  // factor is known to be too large (the caller then goes to the indirect back-end without waiting for the rest)
  double nnz_limit = 0.0, flops_limit = 0.0, nnz = 0.0, flops = 0.0;
  bool aborted = false;
  int min_bucket = 0;
  // Tie-breaking.  false: a node whose degree was just updated goes to the HEAD of its bucket (it is the next pivot among
  // equals: long dependency chains, e.g. row - variable - row on bound-constrained problems).  true: it goes to the TAIL,
  // so all nodes that already had the minimum degree go first -- the independent set of Liu's multiple elimination --
  // which gives flatter elimination trees (fewer levels for the triangular solves) at the same fill on such graphs.
  bool fifo = false;
  std::vector<int> bucket_tail;

  explicit MinDegree(int n) : N(n), var_adj(n), elem_adj(n), members(n), state(n, 0), degree(n, 0), bucket_head(n + 1, -1),
                              next(n, -1), prev(n, -1), stamp(n, 0), wstamp(n, 0), wcount(n, 0), extra(n, 0), bucket_tail(n + 1, -1) {}

  void unlink(int i) {
    if (prev[i] >= 0) next[prev[i]] = next[i]; else bucket_head[degree[i]] = next[i];
    if (next[i] >= 0) prev[next[i]] = prev[i]; else bucket_tail[degree[i]] = prev[i];
    next[i] = prev[i] = -1;
  }
  void link(int i) {
    int d = degree[i];
    prev[i] = -1; next[i] = bucket_head[d];
    if (bucket_head[d] >= 0) prev[bucket_head[d]] = i; else bucket_tail[d] = i;
    bucket_head[d] = i;
  }
  void link_tail(int i) {
    int d = degree[i];
    next[i] = -1; prev[i] = bucket_tail[d];
    if (bucket_tail[d] >= 0) next[bucket_tail[d]] = i; else bucket_head[d] = i;
    bucket_tail[d] = i;
  }

  void run(std::vector<int> &order) {
    order.resize(N);
    for (int i = 0; i < N; i++) degree[i] = std::min(N - 1, (int)var_adj[i].size() + extra[i]);
    for (int i = N - 1; i >= 0; i--) link(i);
    std::vector<int> clique;
    int tag = 0;
    for (int k = 0; k < N; k++) {
      while (min_bucket < N && bucket_head[min_bucket] < 0) min_bucket++;
      const int p = bucket_head[min_bucket];
      unlink(p);
      order[k] = p;
      ++tag;
      stamp[p] = tag;
      clique.clear();
      for (int v : var_adj[p])
        if (state[v] == 0 && stamp[v] != tag) { stamp[v] = tag; clique.push_back(v); }
      for (int e : elem_adj[p]) {
        if (state[e] != 1) continue;
        for (int v : members[e])
          if (state[v] == 0 && stamp[v] != tag) { stamp[v] = tag; clique.push_back(v); }
        state[e] = 2;
        std::vector<int>().swap(members[e]);
      }
      state[p] = 1;
      std::vector<int>().swap(var_adj[p]);
      std::vector<int>().swap(elem_adj[p]);
      members[p] = clique;
      const int csize = (int)clique.size();
      const double colcount = (double)csize + (double)extra[p];  // dense neighbours sit below every column they touch
      nnz += colcount; flops += colcount * colcount;
      if ((nnz_limit > 0.0 && nnz > nnz_limit) || (flops_limit > 0.0 && flops > flops_limit)) { aborted = true; return; }
      // |L_e \ L_p| for every live element that touches the new clique
      for (int i : clique)
        for (int e : elem_adj[i]) {
          if (state[e] != 1) continue;
          if (wstamp[e] != tag) { wstamp[e] = tag; wcount[e] = (int)members[e].size(); }
          wcount[e]--;
        }
      for (int i : clique) {
        unlink(i);
        auto &va = var_adj[i];
        size_t w = 0;
        for (int v : va)
          if (state[v] == 0 && stamp[v] != tag) va[w++] = v;
        va.resize(w);
        auto &ea = elem_adj[i];
        w = 0;
        long long d = 0;
        for (int e : ea) {
          if (state[e] != 1) continue;
          if (wcount[e] == 0) { state[e] = 2; std::vector<int>().swap(members[e]); continue; }  // subset of the new clique
          ea[w++] = e;
          d += wcount[e];
        }
        ea.resize(w);
        ea.push_back(p);
        d += (long long)va.size() + (csize - 1) + extra[i];
        long long bound = (long long)degree[i] + (csize - 1);
        if (d > bound) d = bound;
        if (d > N - 1) d = N - 1;
        if (d < 0) d = 0;
        degree[i] = (int)d;
        if (fifo) link_tail(i); else link(i);
        if (degree[i] < min_bucket) min_bucket = degree[i];
      }
    }
  }
};

// ---------------------------------------------------------------------------
// Nested dissection by level structures (George & Liu's automatic nested dissection): a connected piece is cut at a
// small middle level of the breadth-first level structure rooted at a pseudo-peripheral node; the two sides are
// ordered first (recursively), the separator last.  On banded / chain-like graphs (multi-stage control problems) the
// min-degree elimination tree is one long chain -- thousands of sequential levels -- while this ordering gives a
// balanced tree of height O(separator size x log stages) at 2-3x the fill; the separators at the top of the tree
// form the dense block that the direct back-end inverts explicitly.
// ---------------------------------------------------------------------------
struct NestedDissection {
  int N;
  const std::vector<int64_t> &xadj;
  const std::vector<int> &adj;
  // tag: the piece a node currently belongs to; dist: the stamp of the last search that reached it (or its local index
  // inside a leaf).  Pieces handled by different threads are disjoint node sets: a thread writes only its own nodes and
  // reads a foreign node's tag at most (never equal to its own: tags are unique), through relaxed atomics.
  std::vector<int> tag, dist, order, mark;
  std::atomic<int> next_tag{0}, stamp{0}, spare_threads{0}, mark_stamp{0};
  int leaf_size;
  size_t parallel_min = (size_t)1 << 62;  // pieces from this size on put their two sides on two threads

  NestedDissection(int n, const std::vector<int64_t> &xa, const std::vector<int> &ad, int leaf)
      : N(n), xadj(xa), adj(ad), tag(n, -1), dist(n, -1), mark(n, 0), leaf_size(leaf) {}

  // A level of a level structure separates what lies before it from what lies after it -- but only its nodes that HAVE a
  // neighbour in the next level are needed for that (George & Liu): the others (on a KKT graph: the constraint rows that
  // hang off a variable of the level before, pendant nodes among them) join the side before it.  Without this a separator
  // of a grid QP carried one pendant row per variable, ordered last with it: twice the fill of minimum degree.
  void trim(const std::vector<int> &level, const std::vector<int> &next_level, std::vector<int> &sep, std::vector<int> &before) {
    const int st = ++mark_stamp;
    for (int w : next_level) mark[w] = st;
    for (int v : level) {
      bool needed = false;
      for (int64_t q = xadj[v]; q < xadj[v + 1] && !needed; q++) needed = mark[adj[q]] == st;
      (needed ? sep : before).push_back(v);
    }
  }

  int tag_of(int w) const { return __atomic_load_n(&tag[w], __ATOMIC_RELAXED); }
  void set_tag(int v, int t) { __atomic_store_n(&tag[v], t, __ATOMIC_RELAXED); }

  // breadth-first levels of the piece marked `t` from root r; returns the level sets
  void bfs(int r, int t, std::vector<std::vector<int>> &levels) {
    levels.clear();
    std::vector<int> cur{r};
    const int st = ++stamp;
    dist[r] = st;
    while (!cur.empty()) {
      levels.push_back(cur);
      std::vector<int> nxt;
      for (int v : cur)
        for (int64_t q = xadj[v]; q < xadj[v + 1]; q++) {
          int w = adj[q];
          if (tag_of(w) == t && dist[w] != st) { dist[w] = st; nxt.push_back(w); }
        }
      cur.swap(nxt);
    }
  }

  void leaf(const std::vector<int> &V, std::vector<int> &out) {  // min-degree on the induced subgraph
    const int k = (int)V.size();
    if (k <= 2) { for (int v : V) out.push_back(v); return; }
    std::vector<int> local(k);
    const int t = ++next_tag;
    for (int i = 0; i < k; i++) { set_tag(V[i], t); dist[V[i]] = i; }
    MinDegree md(k);
    for (int i = 0; i < k; i++)
      for (int64_t q = xadj[V[i]]; q < xadj[V[i] + 1]; q++) {
        int w = adj[q];
        if (tag_of(w) == t) md.var_adj[i].push_back(dist[w]);
        else md.extra[i]++;  // neighbours outside the leaf (separators above it) are eliminated later
      }
    md.run(local);
    for (int i = 0; i < k; i++) out.push_back(V[local[i]]);
    for (int v : V) dist[v] = -1;
  }

  void run(std::vector<int> V) {
    order.clear();
    order.reserve(V.size());
    dissect(std::move(V), order);
  }

  void dissect(std::vector<int> V, std::vector<int> &out) {
    if ((int)V.size() <= leaf_size) { leaf(V, out); return; }
    // connected components, one after the other (a loop: a graph can fall into thousands of pieces)
    const int t = ++next_tag;
    for (int v : V) set_tag(v, t);
    std::vector<std::vector<int>> levels;
    std::vector<std::vector<int>> comps;
    const int base = stamp;  // stamps handed out from here on that land on THIS piece's nodes come from this call's searches
    for (size_t i = 0; i < V.size(); i++) {
      if (dist[V[i]] > base) continue;  // reached by one of this call's searches
      bfs(V[i], t, levels);
      std::vector<int> comp;
      for (auto &L : levels) comp.insert(comp.end(), L.begin(), L.end());
      comps.push_back(std::move(comp));
    }
    if (comps.size() > 1) {
      for (auto &c : comps) dissect(std::move(c), out);
      return;
    }
    // a single connected piece: levels from V[0] are in `levels`
    connected(std::move(comps[0]), t, levels, out);
  }

  // levels [lo, hi) of a level structure as one piece: separator = the smallest level with 30 % of the piece on both sides
  // (the median level otherwise), sides first, separator last; at most leaf_size nodes: min-degree; fewer than three levels
  // but more nodes than a leaf: the general path (its own searches)
  void range(const std::vector<std::vector<int>> &levels, int lo, int hi, std::vector<int> &out) {
    size_t total = 0;
    for (int l = lo; l < hi; l++) total += levels[l].size();
    if (total == 0) return;
    if (total <= (size_t)leaf_size || hi - lo < 3) {
      std::vector<int> V;
      V.reserve(total);
      for (int l = lo; l < hi; l++) V.insert(V.end(), levels[l].begin(), levels[l].end());
      if (total <= (size_t)leaf_size) leaf(V, out); else dissect(std::move(V), out);
      return;
    }
    int best = -1, median = lo + 1;
    size_t before = levels[lo].size();
    double gap = 1e300;
    for (int l = lo + 1; l + 1 < hi; l++) {
      const double a = (double)before, b = (double)total - a - (double)levels[l].size();
      if (a >= 0.3 * (double)total && b >= 0.3 * (double)total && (best < 0 || levels[l].size() < levels[best].size())) best = l;
      if (std::fabs(a - b) < gap) { gap = std::fabs(a - b); median = l; }
      before += levels[l].size();
    }
    if (best < 0) best = median;
    if (total >= parallel_min && spare_threads.fetch_sub(1) > 0) {
      std::vector<int> outA, outB;
      std::exception_ptr failed, mine;
      std::thread other([&]() { try { range(levels, lo, best, outA); } catch (...) { failed = std::current_exception(); } });
      try { range(levels, best + 1, hi, outB); } catch (...) { mine = std::current_exception(); }
      other.join();
      spare_threads.fetch_add(1);
      if (failed) std::rethrow_exception(failed);
      if (mine) std::rethrow_exception(mine);
      out.insert(out.end(), outA.begin(), outA.end());
      out.insert(out.end(), outB.begin(), outB.end());
    } else {
      if (total >= parallel_min) spare_threads.fetch_add(1);  // none was to be had: give the claim back
      range(levels, lo, best, out);
      range(levels, best + 1, hi, out);
    }
    out.insert(out.end(), levels[best].begin(), levels[best].end());
  }

  void connected(std::vector<int> V, int t, std::vector<std::vector<int>> &levels, std::vector<int> &out) {
    // pseudo-peripheral root: restart from a smallest-degree node of the last level while the depth grows
    for (int pass = 0; pass < 4; pass++) {
      const auto &last = levels.back();
      int r = last[0];
      for (int v : last) if (xadj[v + 1] - xadj[v] < xadj[r + 1] - xadj[r]) r = v;
      std::vector<std::vector<int>> l2;
      bfs(r, t, l2);
      const bool deeper = l2.size() > levels.size();
      levels.swap(l2);
      if (!deeper) break;
    }
    const int nl = (int)levels.size();
    if (nl < 3) { leaf(V, out); return; }
    // Round 5: a LONG THIN piece (a banded / multi-stage problem: mean level width below a sixteenth of the depth) is
    // dissected on this one level structure all the way down -- a middle level separates the levels before it from those
    // after it whatever the sub-piece, so the sides need no searches of their own (every recursion step used to cost ~5
    // breadth-first passes over its piece: 0.65 s of the 2 s setup of control-1e6).  Anything else -- grid-like pieces, whose
    // sub-pieces have better roots than the parent's -- keeps the per-piece level structures.
    static const bool reuse = !(getenv("OSQP_AMD_ND_REUSE") && atoi(getenv("OSQP_AMD_ND_REUSE")) == 0);
    if (reuse && V.size() > (size_t)leaf_size && (double)nl * (double)nl >= 16.0 * (double)V.size()) {
      std::vector<int>().swap(V);
      range(levels, 0, nl, out);
      return;
    }
    // separator: the smallest level whose sides both hold at least 30 % of the piece; otherwise the level at the median
    const double total = (double)V.size();
    int best = -1;
    size_t before = levels[0].size();
    int median = 1;
    double gap = 1e300;
    for (int l = 1; l + 1 < nl; l++) {
      const double a = (double)before, b = total - a - (double)levels[l].size();
      if (a >= 0.3 * total && b >= 0.3 * total && (best < 0 || levels[l].size() < levels[best].size())) best = l;
      if (std::fabs(a - b) < gap) { gap = std::fabs(a - b); median = l; }
      before += levels[l].size();
    }
    if (best < 0) best = median;
    std::vector<int> A, B, Sep;
    for (int l = 0; l < best; l++) A.insert(A.end(), levels[l].begin(), levels[l].end());
    trim(levels[best], levels[best + 1], Sep, A);
    for (int l = best + 1; l < nl; l++) B.insert(B.end(), levels[l].begin(), levels[l].end());
    std::vector<std::vector<int>>().swap(levels);
    std::vector<int>().swap(V);
    // the two sides are independent: a large piece hands one of them to another thread while threads are to be had
    if (A.size() + B.size() >= parallel_min && spare_threads.fetch_sub(1) > 0) {
      std::vector<int> outA;
      outA.reserve(A.size());
      std::exception_ptr failed, mine;
      std::thread other([&]() { try { dissect(std::move(A), outA); } catch (...) { failed = std::current_exception(); } });
      std::vector<int> outB;
      try {
        outB.reserve(B.size());
        dissect(std::move(B), outB);
      } catch (...) { mine = std::current_exception(); }
      other.join();
      spare_threads.fetch_add(1);
      if (failed) std::rethrow_exception(failed);
      if (mine) std::rethrow_exception(mine);
      out.insert(out.end(), outA.begin(), outA.end());
      out.insert(out.end(), outB.begin(), outB.end());
    } else {
      if (A.size() + B.size() >= parallel_min) spare_threads.fetch_add(1);  // none was to be had: give the claim back
      dissect(std::move(A), out);
      dissect(std::move(B), out);
    }
    for (int v : Sep) out.push_back(v);
  }
};

struct Upper {  // upper-triangular pattern, column compressed, with the origin of every entry
  int N = 0;
  std::vector<int64_t> p;
  std::vector<int> i;
  std::vector<int64_t> origin;  // >= 0: nnz index in triu(P); <= -2: -(nnz index in A) - 2; -1: structural diagonal only
};

}  // namespace

int kkt_graph_depth(const HostCsc &P, const HostCsc &A, const std::vector<int> &row_map, int mr) {
  const int n = P.cols, N = n + mr;
  if (N < 2) return 0;
  // adjacency of the KKT graph: variable - variable through P, variable - row through A
  std::vector<int64_t> xadj(N + 1, 0);
  for (int j = 0; j < n; j++)
    for (int64_t k = P.p[j]; k < P.p[j + 1]; k++) if (P.i[k] != j) { xadj[P.i[k] + 1]++; xadj[j + 1]++; }
  for (int j = 0; j < n; j++)
    for (int64_t k = A.p[j]; k < A.p[j + 1]; k++) { const int r = row_map[A.i[k]]; if (r >= 0) { xadj[j + 1]++; xadj[n + r + 1]++; } }
  for (int i = 0; i < N; i++) xadj[i + 1] += xadj[i];
  std::vector<int> adj((size_t)xadj[N]);
  {
    std::vector<int64_t> f(xadj.begin(), xadj.end() - 1);
    for (int j = 0; j < n; j++)
      for (int64_t k = P.p[j]; k < P.p[j + 1]; k++) if (P.i[k] != j) { adj[f[P.i[k]]++] = j; adj[f[j]++] = P.i[k]; }
    for (int j = 0; j < n; j++)
      for (int64_t k = A.p[j]; k < A.p[j + 1]; k++) { const int r = row_map[A.i[k]]; if (r >= 0) { adj[f[j]++] = n + r; adj[f[n + r]++] = j; } }
  }
  std::vector<int> dist(N, -1), cur, nxt;
  int root = 0, depth = 0;
  for (int pass = 0; pass < 2; pass++) {
    std::fill(dist.begin(), dist.end(), -1);
    cur.assign(1, root);
    dist[root] = 0;
    int d = 0, last = root;
    while (!cur.empty()) {
      nxt.clear();
      for (int v : cur)
        for (int64_t q = xadj[v]; q < xadj[v + 1]; q++) { const int w = adj[q]; if (dist[w] < 0) { dist[w] = d + 1; nxt.push_back(w); } }
      if (!nxt.empty()) { d++; last = nxt[0]; }
      cur.swap(nxt);
    }
    depth = std::max(depth, d);
    root = last;
  }
  return depth;
}

namespace {
// What a lean analysis (symbolic.hpp) keeps for symbolic_complete: the pattern of K with its origins and the row patterns
struct LeanState {
  Upper K;
  int64_t nnzP = 0, nnzA = 0;
};
void build_cols(const Upper &K, const std::vector<int> &pv, std::vector<int64_t> &cp, std::vector<int> &ci);
void etree_of(int N, const std::vector<int64_t> &cp, const std::vector<int> &ci, std::vector<int> &parent);
void full_pattern(Symbolic &S, const Upper &K, int64_t nnzL_limit, int64_t nnzP, int64_t nnzA, const std::function<void(const char *)> &stage);
void finish_pattern(Symbolic &S, const Upper &K, std::vector<std::vector<int>> &blk_cols, std::vector<std::vector<int>> &blk_cnt,
                    const std::vector<int> &rowlen, int nblk, int nt4, int64_t nnzP, int64_t nnzA,
                    const std::function<void(const char *)> &stage);
}  // namespace

void symbolic_analyse(const HostCsc &P, const HostCsc &A, const std::vector<int> &row_map, int mr, int64_t nnzL_limit,
                      double flops_limit, int ordering, Symbolic &S, bool lean) {
  const int n = P.cols, N = n + mr;
  S.n = n; S.mr = mr; S.N = N; S.too_large = false; S.lean = false;
  // OSQP_AMD_SYMBOLIC_TRACE=1: wall time of every stage on stderr
  static const bool trace = getenv("OSQP_AMD_SYMBOLIC_TRACE") && atoi(getenv("OSQP_AMD_SYMBOLIC_TRACE")) == 1;
  auto t_prev = std::chrono::steady_clock::now();
  std::function<void(const char *)> stage = [&](const char *what) {
    if (!trace) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[symbolic %d] %-34s %8.1f ms\n", ordering, what, 1e3 * std::chrono::duration<double>(now - t_prev).count());
    t_prev = now;
  };
  const int64_t nnzP = P.p[n], nnzA = A.p[n];

  // ---- 1. upper-triangular pattern of K with origins --------------------------------
  std::shared_ptr<LeanState> keep = std::make_shared<LeanState>();
  Upper &K = keep->K;
  K.N = N;
  std::vector<int64_t> cnt(N + 1, 0);
  std::vector<char> has_diag(n, 0);
  for (int j = 0; j < n; j++) {
    for (int64_t k = P.p[j]; k < P.p[j + 1]; k++) if (P.i[k] == j) has_diag[j] = 1;
    cnt[j] = (P.p[j + 1] - P.p[j]) + (has_diag[j] ? 0 : 1);
  }
  for (int r = 0; r < mr; r++) cnt[n + r] = 1;
  for (int64_t k = 0; k < nnzA; k++) { int r = row_map[A.i[k]]; if (r >= 0) cnt[n + r]++; }
  K.p.assign(N + 1, 0);
  for (int j = 0; j < N; j++) K.p[j + 1] = K.p[j] + cnt[j];
  K.i.resize(K.p[N]); K.origin.resize(K.p[N]);
  std::vector<int64_t> fill(K.p.begin(), K.p.end() - 1);
  for (int j = 0; j < n; j++) {
    for (int64_t k = P.p[j]; k < P.p[j + 1]; k++) { int64_t q = fill[j]++; K.i[q] = P.i[k]; K.origin[q] = k; }
    if (!has_diag[j]) { int64_t q = fill[j]++; K.i[q] = j; K.origin[q] = -1; }
  }
  for (int j = 0; j < n; j++)
    for (int64_t k = A.p[j]; k < A.p[j + 1]; k++) {
      int r = row_map[A.i[k]];
      if (r < 0) continue;
      int64_t q = fill[n + r]++;
      K.i[q] = j; K.origin[q] = -(k + 2);
    }
  for (int r = 0; r < mr; r++) { int64_t q = fill[n + r]++; K.i[q] = n + r; K.origin[q] = -1; }

  stage("upper pattern of K");
  // ---- 2. fill-reducing ordering -------------------------------------------------------
  // Nodes of very high degree (a dense constraint row such as a budget 1'x = 1, a dense column of P) are kept out
  // of the quotient graph and eliminated last: pruning their adjacency at every step would make the ordering
  // quadratic, and wherever they are eliminated they fill their whole row anyway.
  std::vector<int> order;
  {
    std::vector<int> deg(N, 0);
    for (int j = 0; j < N; j++)
      for (int64_t q = K.p[j]; q < K.p[j + 1]; q++) if (K.i[q] < j) { deg[K.i[q]]++; deg[j]++; }
    const double dense_limit = std::max(16.0, 10.0 * std::sqrt((double)N));
    std::vector<char> dense(N, 0);
    int ndense = 0;
    for (int i = 0; i < N; i++) if ((double)deg[i] > dense_limit) { dense[i] = 1; ndense++; }
    if (ndense) {
      std::fill(deg.begin(), deg.end(), 0);
      for (int j = 0; j < N; j++)
        for (int64_t q = K.p[j]; q < K.p[j + 1]; q++) { int i = K.i[q]; if (i < j && !dense[i] && !dense[j]) { deg[i]++; deg[j]++; } }
    }
    if (ordering == 1) {  // nested dissection of the graph without its dense nodes (adjacency as plain CSR arrays)
      std::vector<int64_t> xadj(N + 1, 0);
      for (int i = 0; i < N; i++) xadj[i + 1] = xadj[i] + deg[i];
      std::vector<int> adj((size_t)xadj[N]);
      {
        std::vector<int64_t> f(xadj.begin(), xadj.end() - 1);
        for (int j = 0; j < N; j++)
          for (int64_t q = K.p[j]; q < K.p[j + 1]; q++) {
            int i = K.i[q];
            if (i < j && !dense[i] && !dense[j]) { adj[f[i]++] = j; adj[f[j]++] = i; }
          }
      }
      NestedDissection nd(N, xadj, adj, 64);
      const int nt = host_threads(xadj[N]);
      if (nt > 1) { nd.spare_threads = nt - 1; nd.parallel_min = std::max<size_t>((size_t)N / (4 * (size_t)nt), getenv("OSQP_AMD_HOST_THREADS") ? 200 : 20000); }
      std::vector<int> all;
      all.reserve(N);
      for (int i = 0; i < N; i++) if (!dense[i]) all.push_back(i);
      if (!all.empty()) nd.run(std::move(all));
      order = nd.order;
      for (int i = 0; i < N; i++) if (dense[i]) order.push_back(i);
    } else {
      MinDegree md(N);
      for (int i = 0; i < N; i++) md.var_adj[i].reserve(deg[i]);
      for (int j = 0; j < N; j++)
        for (int64_t q = K.p[j]; q < K.p[j + 1]; q++) {
          int i = K.i[q];
          if (i >= j) continue;
          if (!dense[i] && !dense[j]) { md.var_adj[i].push_back(j); md.var_adj[j].push_back(i); }
          else { if (!dense[i]) md.extra[i]++; if (!dense[j]) md.extra[j]++; }
        }
      md.nnz_limit = (double)nnzL_limit; md.flops_limit = flops_limit;
      md.fifo = ordering == 2;
      md.run(order);
      if (md.aborted) { S.too_large = true; S.nnzL = (int64_t)md.nnz; S.flops = md.flops; return; }
    }
    if (ndense)  // isolated in the pruned graph, so their position is free: move them to the end
      std::stable_partition(order.begin(), order.end(), [&](int v) { return !dense[v]; });
  }
  std::vector<int> pinv(N);
  for (int k = 0; k < N; k++) pinv[order[k]] = k;
  stage("ordering");

  // ---- 3. elimination tree of the permuted matrix, node heights, level renumbering ----
  std::vector<int64_t> cp;
  std::vector<int> ci, parent;
  build_cols(K, pinv, cp, ci);
  etree_of(N, cp, ci, parent);
  std::vector<int> height(N, 0);
  for (int k = 0; k < N; k++)
    if (parent[k] >= 0) height[parent[k]] = std::max(height[parent[k]], height[k] + 1);
  int nlevels = 0;
  for (int k = 0; k < N; k++) nlevels = std::max(nlevels, height[k] + 1);
  // stable counting sort of the pivots by height: children still precede parents, so the fill is unchanged
  std::vector<int> lp(nlevels + 1, 0);
  for (int k = 0; k < N; k++) lp[height[k] + 1]++;
  for (int l = 0; l < nlevels; l++) lp[l + 1] += lp[l];
  S.level_ptr = lp;
  std::vector<int> newpos(N);
  {
    std::vector<int> f(lp.begin(), lp.end() - 1);
    for (int k = 0; k < N; k++) newpos[k] = f[height[k]]++;
  }
  S.perm.resize(N); S.pinv.resize(N);
  for (int k = 0; k < N; k++) S.perm[newpos[k]] = order[k];
  for (int k = 0; k < N; k++) S.pinv[S.perm[k]] = k;
  // the tree of the final numbering is the same tree relabelled (the renumbering keeps children before parents)
  S.parent.assign(N, -1);
  for (int k = 0; k < N; k++) if (parent[k] >= 0) S.parent[newpos[k]] = newpos[parent[k]];
  stage("elimination tree, levels");
  if (!lean) { full_pattern(S, K, nnzL_limit, nnzP, nnzA, stage); return; }

  // ---- 4 (lean). the rows of the pattern, walked in the numbering the ORDERING produced -----------------------------------
  // Row k of L = the nodes met climbing the tree from every entry of row k of K towards k.  The final numbering is level by
  // level -- a row's descendants lie all over it, and every step of a climb (mark, parent, count) is a cache miss: 0.3 - 0.7 s
  // for the 6e7 entries of control-1e6 on 16 threads.  In the numbering of the ordering itself a subtree is a contiguous
  // range (a dissection orders the two sides, then the separator; min-degree eliminates neighbours together), so the walk
  // runs there and translates what it emits: the columns of a row through newpos, the row's own id likewise.  The rows
  // leave in the order of the walk (LeanRows::rowid says which row each one is); the device sorts them anyway.
  const int nt4 = host_threads(cp[N] * 8);
  const int nblk = std::max(1, std::min(std::max(N, 1), nt4 == 1 ? 1 : 4 * nt4));
  std::shared_ptr<LeanRows> rows = std::make_shared<LeanRows>();
  rows->cols.assign(nblk, std::vector<int>());
  rows->first.resize(nblk + 1);
  for (int b = 0; b <= nblk; b++) rows->first[b] = (int)((int64_t)N * b / nblk);
  std::vector<int> rowlen(N, 0);                      // by final id
  std::vector<std::vector<int>> marks(nt4), cnts(nt4);  // by the ordering's id, one pair per thread
  std::atomic<int64_t> total{0};
  std::atomic<bool> over{false};
  parallel_blocks(N, nblk, nt4, [&](int b, int k0, int k1, int t) {
    if (marks[t].empty()) { marks[t].assign(N, -1); cnts[t].assign(N, 0); }
    std::vector<int> &mark = marks[t], &cnt = cnts[t], &cols = rows->cols[b];
    cols.reserve((size_t)(k1 - k0) * 24);
    for (int k = k0; k < k1 && !over.load(std::memory_order_relaxed); k++) {
      mark[k] = k;
      const size_t c0 = cols.size();
      for (int64_t q = cp[k]; q < cp[k + 1]; q++)
        for (int i = ci[q]; mark[i] != k; i = parent[i]) { mark[i] = k; cols.push_back(newpos[i]); cnt[i]++; }
      const int len = (int)(cols.size() - c0);
      rowlen[newpos[k]] = len;
      if (total.fetch_add(len) + len > nnzL_limit) over = true;
    }
  });
  if (over) { S.too_large = true; S.nnzL = total; return; }
  S.nnzL = total;
  std::vector<int64_t> colcount(N, 0);
  parallel_blocks(N, 4 * nt4, nt4, [&](int, int i0, int i1, int) {
    for (int i = i0; i < i1; i++) {
      int64_t c = 0;
      for (int t = 0; t < nt4; t++) if (!cnts[t].empty()) c += cnts[t][i];
      colcount[newpos[i]] = c;
    }
  });
  S.Lp.assign(N + 1, 0); S.Rp.assign(N + 1, 0);
  S.flops = 0.0;
  for (int j = 0; j < N; j++) {
    S.Lp[j + 1] = S.Lp[j] + colcount[j];
    S.flops += (double)colcount[j] * (double)colcount[j];
    S.Rp[j + 1] = S.Rp[j] + rowlen[j];
  }
  rows->rowid = std::move(newpos);
  keep->nnzP = nnzP; keep->nnzA = nnzA;
  S.lean_rows = rows;
  S.lean_state = keep;
  S.lean = true;
  stage("row patterns, column counts (lean)");
}

void symbolic_complete(Symbolic &S) {
  if (!S.lean) return;
  std::shared_ptr<LeanState> keep = std::static_pointer_cast<LeanState>(S.lean_state);
  static const bool trace = getenv("OSQP_AMD_SYMBOLIC_TRACE") && atoi(getenv("OSQP_AMD_SYMBOLIC_TRACE")) == 1;
  auto t_prev = std::chrono::steady_clock::now();
  std::function<void(const char *)> stage = [&](const char *what) {
    if (!trace) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[symbolic +] %-34s %8.1f ms\n", what, 1e3 * std::chrono::duration<double>(now - t_prev).count());
    t_prev = now;
  };
  S.lean = false;
  S.lean_rows.reset();
  full_pattern(S, keep->K, INT64_MAX, keep->nnzP, keep->nnzA, stage);  // the rows again, in the final numbering (the rare case)
  S.lean_state.reset();
}

namespace {
// rows r < c of every column c of the permuted upper pattern
void build_cols(const Upper &K, const std::vector<int> &pv, std::vector<int64_t> &cp, std::vector<int> &ci) {
  const int N = K.N;
  std::vector<int64_t> c2(N + 1, 0);
  for (int j = 0; j < N; j++)
    for (int64_t q = K.p[j]; q < K.p[j + 1]; q++) {
      int a = pv[K.i[q]], b = pv[j];
      if (a != b) c2[std::max(a, b) + 1]++;
    }
  for (int j = 0; j < N; j++) c2[j + 1] += c2[j];
  cp = c2;
  ci.resize(c2[N]);
  std::vector<int64_t> f(c2.begin(), c2.end() - 1);
  for (int j = 0; j < N; j++)
    for (int64_t q = K.p[j]; q < K.p[j + 1]; q++) {
      int a = pv[K.i[q]], b = pv[j];
      if (a != b) ci[f[std::max(a, b)]++] = std::min(a, b);
    }
}
void etree_of(int N, const std::vector<int64_t> &cp, const std::vector<int> &ci, std::vector<int> &parent) {
  parent.assign(N, -1);
  std::vector<int> anc(N, -1);
  for (int k = 0; k < N; k++)
    for (int64_t q = cp[k]; q < cp[k + 1]; q++) {
      int i = ci[q];
      while (i != -1 && i < k) {  // path compression towards k
        int nx = anc[i];
        anc[i] = k;
        if (nx == -1) { parent[i] = k; break; }
        i = nx;
      }
    }
}

// ---- 4. pattern of L in the final numbering: row patterns by climbing the tree from each entry of the row ----
// The rows are independent: blocks of consecutive rows go to host threads (own mark array), each block keeps its rows'
// patterns and counts its entries per column; a block's first position inside column i is Lp[i] + the counts of the blocks
// before it, so the second pass (finish_pattern) writes the CSC row lists (ascending: blocks and rows in order), and the CSR
// view with the CSC position of every entry (row k: columns ascending after a sort of its small buffer), without any two
// threads touching the same slot.
void full_pattern(Symbolic &S, const Upper &K, int64_t nnzL_limit, int64_t nnzP, int64_t nnzA, const std::function<void(const char *)> &stage) {
  const int N = S.N;
  std::vector<int64_t> cp;
  std::vector<int> ci;
  build_cols(K, S.pinv, cp, ci);
  const std::vector<int> &parent = S.parent;
  const int nt4 = host_threads(cp[N] * 8);
  // more blocks than threads, dealt through a counter: the rows of a factor with a dense block differ in length by orders of
  // magnitude (a block holds one count per column: N ints, which bounds how many a long problem can afford)
  const int nblk = std::max(1, std::min(std::max(N, 1), nt4 == 1 ? 1 : std::max(2 * nt4, std::min(8 * nt4, (int)(((int64_t)1 << 27) / std::max(1, N))))));
  std::vector<std::vector<int>> blk_cols(nblk), blk_cnt(nblk);
  std::vector<int> rowlen(N, 0);
  std::vector<std::vector<int>> marks(nt4);
  std::atomic<int64_t> total{0};
  std::atomic<bool> over{false};
  parallel_blocks(N, nblk, nt4, [&](int b, int k0, int k1, int t) {
    if (marks[t].empty()) marks[t].assign(N, -1);
    std::vector<int> &mark = marks[t], &cols = blk_cols[b], &cnt = blk_cnt[b];
    cnt.assign(N, 0);
    for (int k = k0; k < k1 && !over.load(std::memory_order_relaxed); k++) {
      mark[k] = k;
      const size_t c0 = cols.size();
      for (int64_t q = cp[k]; q < cp[k + 1]; q++)
        for (int i = ci[q]; mark[i] != k; i = parent[i]) { mark[i] = k; cols.push_back(i); cnt[i]++; }
      rowlen[k] = (int)(cols.size() - c0);
      if (total.fetch_add(rowlen[k]) + rowlen[k] > nnzL_limit) over = true;
    }
  });
  if (over) { S.too_large = true; S.nnzL = total; return; }
  S.nnzL = total;
  finish_pattern(S, K, blk_cols, blk_cnt, rowlen, nblk, nt4, nnzP, nnzA, stage);
}
}  // namespace

namespace {
void finish_pattern(Symbolic &S, const Upper &K, std::vector<std::vector<int>> &blk_cols, std::vector<std::vector<int>> &blk_cnt,
                    const std::vector<int> &rowlen, int nblk, int nt4, int64_t nnzP, int64_t nnzA,
                    const std::function<void(const char *)> &stage) {
  const int N = S.N;
  {
    std::vector<int64_t> colcount(N, 0);
    // counts -> first positions, block after block, in place (the counts become offsets relative to the column's start)
    parallel_blocks(N, 4 * nt4, nt4, [&](int, int i0, int i1, int) {
      for (int i = i0; i < i1; i++) {
        int64_t run = 0;
        for (int b = 0; b < nblk; b++) { const int c = blk_cnt[b][i]; blk_cnt[b][i] = (int)run; run += c; }
        colcount[i] = run;
      }
    });
    S.flops = 0.0;
    for (int k = 0; k < N; k++) S.flops += (double)colcount[k] * (double)colcount[k];
    stage("column counts");
    S.Lp.assign(N + 1, 0);
    for (int j = 0; j < N; j++) S.Lp[j + 1] = S.Lp[j] + colcount[j];
    S.Rp.assign(N + 1, 0);
    for (int k = 0; k < N; k++) S.Rp[k + 1] = S.Rp[k] + rowlen[k];
    S.Li.resize(S.nnzL); S.Rj.resize(S.nnzL); S.Rmap.resize(S.nnzL);
    stage("(arrays of L allocated)");
    parallel_blocks(N, nblk, nt4, [&](int b, int k0, int k1, int) {
      const std::vector<int> &cols = blk_cols[b];
      std::vector<int> &off = blk_cnt[b];
      std::vector<std::pair<int, int64_t>> rowbuf;
      std::vector<uint64_t> bits;  // long rows (a dense block: thousands of columns) come out ascending from a bitmap, not a sort
      size_t c = 0;
      for (int k = k0; k < k1; k++) {
        const int len = rowlen[k];
        int64_t w = S.Rp[k];
        if (len > 64 && (int64_t)len * 32 > k) {
          if (bits.empty()) bits.assign((size_t)N / 64 + 1, 0);
          int lo = N, hi = -1;
          for (int e = 0; e < len; e++) {
            const int i = cols[c++];
            bits[(size_t)i >> 6] |= (uint64_t)1 << (i & 63);
            lo = std::min(lo, i); hi = std::max(hi, i);
          }
          for (int wd = lo >> 6; wd <= hi >> 6; wd++) {
            uint64_t b = bits[(size_t)wd];
            bits[(size_t)wd] = 0;
            while (b) {
              const int i = wd * 64 + __builtin_ctzll(b);
              b &= b - 1;
              const int64_t t = S.Lp[i] + off[i]++;
              S.Li[t] = k; S.Rj[w] = i; S.Rmap[w] = t; w++;
            }
          }
          continue;
        }
        rowbuf.clear();
        for (int e = 0; e < len; e++) {
          const int i = cols[c++];
          const int64_t t = S.Lp[i] + off[i]++;
          S.Li[t] = k;
          rowbuf.emplace_back(i, t);
        }
        std::sort(rowbuf.begin(), rowbuf.end());
        for (const auto &e : rowbuf) { S.Rj[w] = e.first; S.Rmap[w] = e.second; w++; }
      }
    });
    stage("(row lists and CSR view written)");
  }
  stage("CSR view of L");
  // ---- 5. scatter maps from the caller's nnz order into Lx / D ------------------------
  if (S.no_host_maps) { S.PtoL.clear(); S.AtoL.clear(); stage("(scatter maps left to the device)"); return; }
  S.PtoL.assign(nnzP, 0);
  S.AtoL.assign(nnzA, INT64_MIN);
  // every entry of K on its own (a binary search in its column of L; distinct targets): columns dealt to host threads --
  // 3.4 s on one core for a 6000 x 6000 dense P (1.6e7 searches that miss the cache), the largest piece of that setup
  auto map_columns = [&](int j0, int j1) {
    for (int j = j0; j < j1; j++)
      for (int64_t q = K.p[j]; q < K.p[j + 1]; q++) {
        int64_t org = K.origin[q];
        if (org == -1) continue;
        int a = S.pinv[K.i[q]], b = S.pinv[j];
        int64_t target;
        if (a == b) target = -(int64_t)a - 1;
        else {
          int c = std::min(a, b), r = std::max(a, b);
          const int *beg = S.Li.data() + S.Lp[c], *end = S.Li.data() + S.Lp[c + 1];
          const int *it = std::lower_bound(beg, end, r);
          target = S.Lp[c] + (it - beg);
        }
        if (org >= 0) S.PtoL[org] = target; else S.AtoL[-(org + 2)] = target;
      }
  };
  {
    const int64_t entries = K.p[N];
    int nt = entries < (int64_t)1 << 20 ? 1 : (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (nt == 1) map_columns(0, N);
    else {
      std::vector<std::thread> pool;
      int j0 = 0;
      for (int t = 0; t < nt; t++) {  // equal shares of the entries
        const int64_t goal = entries * (t + 1) / nt;
        int j1 = t + 1 == nt ? N : (int)(std::lower_bound(K.p.begin(), K.p.end(), goal) - K.p.begin());
        j1 = std::max(j0, std::min(N, j1));
        pool.emplace_back(map_columns, j0, j1);
        j0 = j1;
      }
      for (auto &th : pool) th.join();
    }
  }
  stage("scatter maps");
}
}  // namespace

void supernode_wmap(const Symbolic &S, Supernodes &out) {
  const int N = S.N;
  std::vector<int> owner(N);
  for (int J = 0; J < out.count; J++)
    for (int q = out.ptr[J]; q < out.ptr[J + 1]; q++) owner[q] = J;
  out.wmap.assign(out.woff[out.count], -1);
  const int nt = host_threads(S.nnzL);
  parallel_blocks(N, 8 * nt, nt, [&](int, int v0, int v1, int) {
    for (int v = v0; v < v1; v++) {
      const int J = owner[out.slot[v]];
      for (int64_t t = S.Lp[v]; t < S.Lp[v + 1]; t++) {
        const int r = S.Li[t];
        if (owner[out.slot[r]] != J) continue;
        const int64_t a = out.slot[r] - out.ptr[J], b = out.slot[v] - out.ptr[J];
        out.wmap[out.woff[J] + a * (a + 1) / 2 + b] = t;
      }
    }
  });
}

void build_supernodes(const Symbolic &S, int smax, Supernodes &out, bool with_wmap, bool partition_only) {
  const int N = S.N;
  const std::vector<int> &parent = S.parent;
  static const bool trace = getenv("OSQP_AMD_SYMBOLIC_TRACE") && atoi(getenv("OSQP_AMD_SYMBOLIC_TRACE")) == 1;
  auto t_prev = std::chrono::steady_clock::now();
  auto stage = [&](const char *what) {
    if (!trace) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[supernodes] %-36s %8.1f ms\n", what, 1e3 * std::chrono::duration<double>(now - t_prev).count());
    t_prev = now;
  };
  out = Supernodes();
  out.smax = smax;
  // Lone leaves (round 5): a tree leaf whose column holds ONE entry (a box-constraint row hanging off its variable: 7.8e5 of
  // the 2.67e6 pivots of control-1e6) stays out of its parent's subtree supernode.  Inside, it adds a row and a column to an
  // inverted block (a leaf subtree of 40 pivots holds 11 of them: half the triangle); outside, it is a supernode of one
  // pivot -- nothing to do forward, one lane's work backward (direct.hip k_sn_single_bwd) -- and one entry in its parent's
  // forward row.  The price is one more level of the supernode graph (~10 us per iteration: control T = 800 10.5 -> 8.8 k
  // it/s, T = 8000 5.0 -> 4.5 k, T = 30000 2.41 -> 2.47 k, control-1e6 1 647 -> 1 714 it/s and 0.058 -> 0.055 s to eps), so
  // by default only from 400 000 such leaves on.  OSQP_AMD_SNODE_LEAF=0: never; 1: always; 2: always, and they do not
  // count towards the size of a subtree either (larger blocks: 1 721 it/s, but no gain in the factorisation).
  const int leaf_env = getenv("OSQP_AMD_SNODE_LEAF") ? atoi(getenv("OSQP_AMD_SNODE_LEAF")) : -1;
  auto is_lone = [&](int v) { return parent[v] >= 0 && S.Rp[v + 1] == S.Rp[v] && S.Lp[v + 1] - S.Lp[v] == 1; };
  int leaf_mode = leaf_env;
  if (leaf_env < 0) {
    int64_t lones = 0;
    for (int v = 0; v < N; v++) lones += is_lone(v);
    leaf_mode = lones >= 400000 ? 1 : 0;
  }
  auto lone = [&](int v) { return leaf_mode > 0 && smax > 1 && is_lone(v); };
  // subtree sizes (parents have larger indices than their children)
  std::vector<int> size(N, 1), big_children(N, 0);
  if (leaf_mode == 2) for (int v = 0; v < N; v++) if (lone(v)) size[v] = 0;
  for (int v = 0; v < N; v++)
    if (parent[v] >= 0) size[parent[v]] += size[v];
  auto big = [&](int v) { return size[v] > smax; };
  for (int v = 0; v < N; v++)
    if (parent[v] >= 0 && big(v)) big_children[parent[v]]++;
  // Roots first.  A small node joins its parent's subtree supernode.  The big nodes form the top of the tree; there a
  // supernode is a connected piece of it: a node continues the path it is on (it is its parent's only big child), and the
  // top of a path (a separator of a nested-dissection tree) joins the piece of its parent when the whole path still fits,
  // so that a separator and the separators right below it share one inverted block and one step of the solves.
  std::vector<int> only_big(N, -1), plen(N, 1);
  for (int v = 0; v < N; v++)
    if (parent[v] >= 0 && big(v)) only_big[parent[v]] = v;           // meaningful when big_children == 1
  for (int v = 0; v < N; v++)                                         // children before parents
    if (big(v) && big_children[v] == 1) plen[v] = 1 + plen[only_big[v]];
  std::vector<int> sn(N, -1), members, reserved;
  for (int v = N - 1; v >= 0; v--) {
    const int p = parent[v];
    if (!big(v)) {
      if (p >= 0 && !big(p) && !lone(v)) sn[v] = sn[p];
    } else if (p >= 0) {
      if (big_children[p] == 1 && (members[sn[p]] < reserved[sn[p]])) sn[v] = sn[p];              // the path goes on
      else if (reserved[sn[p]] + plen[v] <= smax) { sn[v] = sn[p]; reserved[sn[p]] += plen[v]; }  // a new path joins
    }
    if (sn[v] < 0) {
      sn[v] = (int)members.size();
      members.push_back(0);
      reserved.push_back(big(v) ? std::min(plen[v], smax) : 0);
    }
    members[sn[v]]++;
  }
  const int count = (int)members.size();
  // levels of the supernode graph: one more than the deepest supernode holding a child of one of its nodes
  std::vector<int> level(count, 0);
  for (int v = 0; v < N; v++) {  // ascending: every child is final before its parent is looked at ...
    const int p = parent[v];
    if (p >= 0 && sn[p] != sn[v]) level[sn[p]] = std::max(level[sn[p]], level[sn[v]] + 1);
  }
  // ... except that a path segment's level can still grow after a lower node of it was used: iterate to a fixed point
  for (bool changed = true; changed;) {
    changed = false;
    for (int v = 0; v < N; v++) {
      const int p = parent[v];
      if (p >= 0 && sn[p] != sn[v] && level[sn[p]] < level[sn[v]] + 1) { level[sn[p]] = level[sn[v]] + 1; changed = true; }
    }
  }
  int nlev = 0;
  for (int J = 0; J < count; J++) nlev = std::max(nlev, level[J] + 1);
  out.count = count; out.nlev = nlev;
  out.lvl_ptr.assign(nlev + 1, 0);
  for (int J = 0; J < count; J++) out.lvl_ptr[level[J] + 1]++;
  for (int l = 0; l < nlev; l++) out.lvl_ptr[l + 1] += out.lvl_ptr[l];
  // inside a level the small supernodes (<= kSmall pivots) come first: a level of many small ones is solved with a quarter
  // wavefront each (direct.hip k_sn_level_w), the rest with a whole one
  std::vector<int> newid(count);
  out.lvl_small.assign(nlev, 0);
  out.lvl_single.assign(nlev, 0);
  // ... inside each of the two classes in the order of discovery.  (Measured and dropped in round 5, OSQP_AMD_SNODE_ORDER=1:
  // the supernodes of a level in the order of the supernode that holds their tree parent, so that the rows of a parent gather
  // the solution from neighbouring slots -- control-1e6 1118 -> 1090 it/s: the order of discovery follows the pivot numbering,
  // which is what the streams of the entries and of the blocks are laid out by.)
  {
    static const bool by_parent = getenv("OSQP_AMD_SNODE_ORDER") && atoi(getenv("OSQP_AMD_SNODE_ORDER")) == 1;
    std::vector<int> up_old(count, -1);
    for (int v = 0; v < N; v++) {
      const int p = parent[v];
      if (p >= 0 && sn[p] != sn[v]) up_old[sn[v]] = sn[p];
    }
    std::vector<std::vector<int>> by_level(nlev);
    for (int L = 0; L < nlev; L++) by_level[L].reserve(out.lvl_ptr[L + 1] - out.lvl_ptr[L]);
    for (int J = 0; J < count; J++) by_level[level[J]].push_back(J);
    for (int L = nlev - 1; L >= 0; L--) {  // parents (higher levels) are numbered before their children look at them
      std::vector<int> &v = by_level[L];
      auto small = [&](int J) { return members[J] <= Supernodes::kSmall; };
      if (by_parent)
        std::stable_sort(v.begin(), v.end(), [&](int a, int b) {
          if (small(a) != small(b)) return small(a);
          const int pa = up_old[a] >= 0 ? newid[up_old[a]] : -1, pb = up_old[b] >= 0 ? newid[up_old[b]] : -1;
          return pa < pb;
        });
      else {
        // ... and the single pivots first among the small ones: a leaf of one pivot has nothing to do in the forward sweep
        // (no entries outside, a 1 x 1 unit block) and one lane's work in the backward one (direct.hip k_sn_single_bwd)
        auto mid = std::stable_partition(v.begin(), v.end(), small);
        std::stable_partition(v.begin(), mid, [&](int J) { return members[J] == 1; });
      }
      for (size_t k = 0; k < v.size(); k++) { newid[v[k]] = out.lvl_ptr[L] + (int)k; out.lvl_small[L] += small(v[k]); }
      if (!by_parent) for (size_t k = 0; k < v.size() && members[v[k]] == 1; k++) out.lvl_single[L]++;
    }
  }
  out.ptr.assign(count + 1, 0);
  for (int J = 0; J < count; J++) out.ptr[newid[J] + 1] = members[J];
  for (int J = 0; J < count; J++) out.ptr[J + 1] += out.ptr[J];
  out.piv.resize(N); out.slot.resize(N);
  {
    std::vector<int> f(out.ptr.begin(), out.ptr.end() - 1);
    for (int v = 0; v < N; v++) { const int q = f[newid[sn[v]]]++; out.piv[q] = v; out.slot[v] = q; }
  }
  out.up.assign(count, -1); out.waits.assign(count, 0);
  for (int v = 0; v < N; v++) {
    const int p = parent[v];
    if (p >= 0 && sn[p] != sn[v]) out.up[newid[sn[v]]] = newid[sn[p]];  // only the top node of a supernode leaves it
  }
  for (int J = out.lvl_ptr[std::min(1, nlev)]; J < count; J++)
    if (out.up[J] >= 0) out.waits[out.up[J]]++;
  out.woff.assign(count + 1, 0);
  out.wdoubles = 0;
  for (int J = 0; J < count; J++) {
    const int64_t s = out.ptr[J + 1] - out.ptr[J];
    // the lower triangle, packed (round 4: the dense s x s blocks were 45 % zeros); room for its folded form (round 5,
    // direct.hip k_sn_fold: s + 1 steps of ceil(s / 2) lanes -- the triangle itself for even s, half a row more for odd s)
    out.woff[J + 1] = out.woff[J] + (s + 1) * ((s + 1) / 2);
    out.wdoubles += s * (s + 1) / 2;
    out.flops += (double)(s * s);
  }
  stage("partition, levels, slots");
  if (partition_only) {
    // list lengths from the counts alone (rows and columns including the entries inside the blocks: upper bounds, what the
    // cost model of the solves is asked with before the lists exist -- they are built on the device, direct.hip)
    out.Fp.assign(N + 1, 0); out.Gp.assign(N + 1, 0);
    for (int q = 0; q < N; q++) {
      const int v = out.piv[q];
      out.Fp[q + 1] = out.Fp[q] + (S.Rp[v + 1] - S.Rp[v]);
      out.Gp[q + 1] = out.Gp[q] + (S.Lp[v + 1] - S.Lp[v]);
    }
    out.flops = 2.0 * (out.flops + (double)out.Fp[N]);
    return;
  }
  if (with_wmap) out.wmap.assign(out.woff[count], -1);
  // The entries of L (column v, rows r > v) split into block entries (both ends in one supernode: their place in the dense
  // block) and the rest, which every row lists by the slot of the column (F, forward) and every column by the slot of the row
  // (G, backward).  Columns and rows are independent: host threads take ranges of them -- a column's entries from the CSC
  // arrays, a row's from the CSR view of the same pattern -- and sort each short list by slot (= by level, then supernode).
  const int nt = host_threads(S.nnzL);
  out.Fp.assign(N + 1, 0); out.Gp.assign(N + 1, 0);
  parallel_blocks(N, 8 * nt, nt, [&](int, int v0, int v1, int) {
    for (int v = v0; v < v1; v++) {
      const int J = newid[sn[v]];
      int64_t g = 0;
      for (int64_t t = S.Lp[v]; t < S.Lp[v + 1]; t++) {
        const int r = S.Li[t];
        if (sn[r] == sn[v]) { if (with_wmap) { const int64_t a = out.slot[r] - out.ptr[J], b = out.slot[v] - out.ptr[J]; out.wmap[out.woff[J] + a * (a + 1) / 2 + b] = t; } }
        else g++;
      }
      out.Gp[out.slot[v] + 1] = g;
      int64_t f = 0;
      for (int64_t w = S.Rp[v]; w < S.Rp[v + 1]; w++) f += sn[S.Rj[w]] != sn[v];
      out.Fp[out.slot[v] + 1] = f;
    }
  });
  for (int q = 0; q < N; q++) { out.Fp[q + 1] += out.Fp[q]; out.Gp[q + 1] += out.Gp[q]; }
  stage("block maps, list lengths");
  out.Fj.resize(out.Fp[N]); out.Fpos.resize(out.Fp[N]); out.Gi.resize(out.Gp[N]); out.Gpos.resize(out.Gp[N]);
  stage("(lists allocated)");
  parallel_blocks(N, 8 * nt, nt, [&](int, int q0, int q1, int) {
    std::vector<std::pair<int, int64_t>> buf;
    for (int q = q0; q < q1; q++) {
      const int v = out.piv[q];
      buf.clear();
      for (int64_t w = S.Rp[v]; w < S.Rp[v + 1]; w++)       // row v: the columns outside its supernode, by slot
        if (sn[S.Rj[w]] != sn[v]) buf.push_back({out.slot[S.Rj[w]], S.Rmap[w]});
      std::sort(buf.begin(), buf.end());
      int64_t a = out.Fp[q];
      for (auto &c : buf) { out.Fj[a] = c.first; out.Fpos[a] = c.second; a++; }
      buf.clear();
      for (int64_t t = S.Lp[v]; t < S.Lp[v + 1]; t++)       // column v: the rows outside its supernode, by slot
        if (sn[S.Li[t]] != sn[v]) buf.push_back({out.slot[S.Li[t]], t});
      std::sort(buf.begin(), buf.end());
      int64_t b = out.Gp[q];
      for (auto &c : buf) { out.Gi[b] = c.first; out.Gpos[b] = c.second; b++; }
    }
  });
  stage("row and column lists");
  // forward rows: where the entries that point above level 0 begin
  const int q_upper = nlev > 1 ? out.ptr[out.lvl_ptr[1]] : N;
  out.Fsplit.resize(N);
  for (int q = 0; q < N; q++)
    out.Fsplit[q] = out.Fp[q] + (std::lower_bound(out.Fj.begin() + out.Fp[q], out.Fj.begin() + out.Fp[q + 1], q_upper) - (out.Fj.begin() + out.Fp[q]));
  out.flops = 2.0 * (out.flops + (double)out.Fp[N]);
}

double level_solve_cost_us(const Symbolic &Y, int chain_rows, int dense_max, int dense_min, double chain_level_us) {
  const auto &lp = Y.level_ptr;
  const int nl = (int)lp.size() - 1;
  // the block the factor would really get (choose_dense_top): a block-sparse top chain -- the separators of a banded
  // problem under min-degree -- is NOT taken as one large dense block just because the limit for dense blocks is generous
  int lD, cD, kD;
  choose_dense_top(Y, chain_rows, dense_max, 1024, dense_min, lD, cD, kD);
  const int top = kD ? lD : nl;
  double us = 0.0;
  for (int q = 0; q < top; q++) us += (lp[q + 1] - lp[q] > chain_rows) ? 6.0 : chain_level_us;
  us += (double)Y.nnzL * 24.0 / 2.0e6;
  if (kD) us += (double)kD * (double)kD * 8.0 / 4.0e6 + 10.0;
  return us;
}

double supernode_solve_cost_us(const Supernodes &T, int threads, int levels) {
  double cost = 0.0;
  for (int L = 0; L < (levels >= 0 ? std::min(levels, T.nlev) : T.nlev); L++) {
    double worst = 0.0;
    for (int J = T.lvl_ptr[L]; J < T.lvl_ptr[L + 1]; J++) {
      const int q0 = T.ptr[J], q1 = T.ptr[J + 1];
      const double f = (double)(T.Fp[q1] - T.Fp[q0]), g = (double)(T.Gp[q1] - T.Gp[q0]), blk = (double)(q1 - q0) * (q1 - q0);
      worst = std::max(worst, (f + g + 2.0 * blk) / (double)threads * 0.05);  // ~50 ns per dependent gather of a thread
    }
    cost += 2.0 * 3.0 + worst;  // one hand-over between workgroups per level and direction (k_sn_tree)
  }
  return cost + 4.0 * 6.0;      // level 0 and the rest: two launches per direction
}

double level_solve_cost_us(const Symbolic &Y, int chain_rows, int lD, int kD) {
  const auto &lp = Y.level_ptr;
  double us = 0.0;
  // a chain level: one barrier-separated step of a single workgroup, 2.4 us plus the passes its rows need
  for (int q = 0; q < lD; q++) us += (lp[q + 1] - lp[q] > chain_rows) ? 6.0 : 2.4 + 0.01 * (double)(lp[q + 1] - lp[q]);
  us += (double)Y.nnzL * 24.0 / 2.0e6;
  if (kD) us += (double)kD * (double)kD * 8.0 / 4.0e6 + 10.0;
  return us;
}

bool supernodes_pay(const Symbolic &S, const Supernodes &T, int chain_rows, int lD, int kD, int threads) {
  const int nlev = (int)S.level_ptr.size() - 1;
  return 4 * T.nlev <= nlev && supernode_solve_cost_us(T, threads) < 0.7 * level_solve_cost_us(S, chain_rows, lD, kD);
}

void choose_dense_top(const Symbolic &S, int chain_rows, int dense_max, int dense_sparse_max, int dense_min, int &lD, int &cD, int &kD, bool *is_dense) {
  if (is_dense) *is_dense = false;
  const auto &lp = S.level_ptr;
  const int N = S.N, nlev = (int)lp.size() - 1;
  lD = nlev; cD = N; kD = 0;
  if (nlev < 2) return;
  auto suffix = [&](int limit) {
    int l = nlev;
    while (l > 1 && lp[l] - lp[l - 1] <= chain_rows && N - lp[l - 1] <= limit) l--;
    return l;
  };
  auto entries_inside = [&](int c) {
    int64_t inside = 0;
    for (int r = c; r < N; r++) {
      const int *beg = S.Rj.data() + S.Rp[r], *end = S.Rj.data() + S.Rp[r + 1];
      inside += end - std::lower_bound(beg, end, c);
    }
    return inside;
  };
  int l = suffix(dense_max);
  int c = lp[l], k = N - c;
  if (k < dense_min) return;
  bool dense = entries_inside(c) * 8 >= (int64_t)k * k;
  if (!dense && k > dense_sparse_max) {
    // a block-sparse top (the separators of a nested-dissection tree): the inversion costs k^3, the levels it
    // replaces only k, so a smaller block is the better trade
    l = suffix(dense_sparse_max); c = lp[l]; k = N - c;
    if (k < dense_min) return;
    dense = entries_inside(c) * 8 >= (int64_t)k * k;
  }
  // worth it when the block is dense (then the chain rows are long) or when the dense product is cheaper than
  // walking the block's levels one by one
  const double dense_us = (double)k * (double)k * 8.0 / 4.0e6 + 10.0, chain_us = 1.4 * (double)(nlev - l);
  if (!dense && dense_us > chain_us) return;
  lD = l; cD = c; kD = k;
  if (is_dense) *is_dense = dense;
}

}  // namespace oq
