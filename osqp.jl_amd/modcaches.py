"""Mirror of the reference's modification caches [REF src/modcaches.jl:1-205]: host-side
dirty tracking that the MathOptInterface wrapper uses to turn in-place problem edits into
the minimal sequence of `update_*` / `warm_start_*` calls before `solve`
[REF src/MOI_wrapper.jl:567-576].  Pure bookkeeping, no arithmetic; SURVEY.md row N2.

Indices follow Python conventions (0-based); the sparsity pattern is frozen at
construction, as in the reference.
"""
import numpy as np
import scipy.sparse as sp

from . import interface as oq


class VectorModificationCache:
    """[REF src/modcaches.jl:14-38]"""

    def __init__(self, data):
        self.data = np.array(data, dtype=np.float64, copy=True)
        self.dirty = False

    def __setitem__(self, i, x):
        self.dirty = True
        self.data[i] = x  # i may be an index or slice(None) (the reference's `cache[:] = x`)

    def __getitem__(self, i):
        return self.data[i]

    def processupdates(self, model, updatefun):
        if self.dirty:
            updatefun(model, self.data)
            self.dirty = False


class MatrixModificationCache:
    """[REF src/modcaches.jl:40-142]  Maps (row, col) to the nz index of the CSC matrix given at
    construction; collects modifications; flushes them as (values, indices) in nz order."""

    def __init__(self, S):
        S = sp.csc_matrix(S)
        S.sort_indices()
        self.cartesian_indices = []
        self.cartesian_indices_per_row = {}
        for col in range(S.shape[1]):
            for k in range(S.indptr[col], S.indptr[col + 1]):
                I = (int(S.indices[k]), col)
                self.cartesian_indices.append(I)
                self.cartesian_indices_per_row.setdefault(I[0], []).append(I)
        self.cartesian_indices_set = set(self.cartesian_indices)
        self.modifications = {}

    def __setitem__(self, key, x):
        if isinstance(key, slice) and key == slice(None):  # cache[:] = 0: zero the whole matrix
            if x != 0:
                raise ValueError("Changing the sparsity pattern is not allowed.")
            for I in self.cartesian_indices:
                self.modifications[I] = 0.0
            return
        row, col = key
        if isinstance(col, slice) and col == slice(None):  # cache[row, :] = 0: zero a row
            if x != 0:
                raise ValueError("Changing the sparsity pattern is not allowed.")
            for I in self.cartesian_indices_per_row.get(int(row), []):
                self.modifications[I] = 0.0
            return
        I = (int(row), int(col))
        if I not in self.cartesian_indices_set:
            raise ValueError("Changing the sparsity pattern is not allowed.")
        self.modifications[I] = float(x)

    def __getitem__(self, key):
        return self.modifications[(int(key[0]), int(key[1]))]

    def processupdates(self, model, updatefun):
        if self.modifications:
            vals, inds = [], []
            for i, I in enumerate(self.cartesian_indices):
                if I in self.modifications:
                    vals.append(self.modifications[I])
                    inds.append(i)
            updatefun(model, np.array(vals), np.array(inds, dtype=np.int64))
            self.modifications.clear()


class ProblemModificationCache:
    """[REF src/modcaches.jl:145-179]; P is cached as its upper triangle, like the C side sees it."""

    def __init__(self, P, q, A, l, u):
        self.P = MatrixModificationCache(sp.triu(sp.csc_matrix(P), format="csc"))
        self.q = VectorModificationCache(q)
        self.A = MatrixModificationCache(A)
        self.l = VectorModificationCache(l)
        self.u = VectorModificationCache(u)

    def processupdates(self, model):
        """Flush order of the reference: bounds together (setting just one may violate l <= u), then
        P, q, A, l, u [REF src/modcaches.jl:166-179]."""
        if self.l.dirty and self.u.dirty:
            oq.update_bounds(model, self.l.data, self.u.data)
            self.l.dirty = False
            self.u.dirty = False
        self.P.processupdates(model, oq.update_P)
        self.q.processupdates(model, oq.update_q)
        self.A.processupdates(model, oq.update_A)
        self.l.processupdates(model, oq.update_l)
        self.u.processupdates(model, oq.update_u)


class WarmStartCache:
    """[REF src/modcaches.jl:181-203]"""

    def __init__(self, n, m):
        self.x = VectorModificationCache(np.zeros(n))
        self.y = VectorModificationCache(np.zeros(m))

    def processupdates(self, model):
        if self.x.dirty and self.y.dirty:
            # setting the warm start for x only zeroes the stored warm start for y and vice versa
            oq.warm_start_x_y(model, self.x.data, self.y.data)
            self.x.dirty = False
            self.y.dirty = False
        self.x.processupdates(model, oq.warm_start_x)
        self.y.processupdates(model, oq.warm_start_y)


def optimize(model, modcache, warmstartcache, results=None):
    """The body of `MOI.optimize!` [REF src/MOI_wrapper.jl:567-576]: flush the caches, solve, keep the
    solution as the next warm start without dirtying the cache."""
    modcache.processupdates(model)
    warmstartcache.processupdates(model)
    results = oq.solve(model, results)
    if results.info.status in ("Solved", "Solved_inaccurate", "Max_iter_reached"):
        warmstartcache.x.data[:] = results.x
        warmstartcache.y.data[:] = results.y
    return results
