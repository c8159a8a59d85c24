"""Host analysis of the direct back-end on a g x g grid QP (tests/qp_zoo.py grid2d), stage by stage
(OSQP_AMD_SYMBOLIC_TRACE=1): ordering, pattern of L, supernode partition.  No device work.
usage: python tools/symbolic_probe_grid.py [g] [ordering ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OSQP_AMD_SYMBOLIC_TRACE", "1")
import osqp_jl_amd as oq  # noqa: E402
import qp_zoo  # noqa: E402
from test_symbolic_host import probe  # noqa: E402

g = int(sys.argv[1]) if len(sys.argv) > 1 else 700
orderings = [int(a) for a in sys.argv[2:]] or [0, 1]
lib = oq.load_library()
prob = qp_zoo.grid2d(g)
for o in orderings:
    t = time.time()
    r = probe(lib, prob, o)
    print("ordering %d: probe %.2f s (includes the probe's own invariant checks)" % (o, time.time() - t), r, file=sys.stderr)
