"""Host analysis of the direct back-end at the control-1e6 size, stage by stage (OSQP_AMD_SYMBOLIC_TRACE=1): ordering,
pattern of L, supernode partition.  No device work.  usage: python tools/symbolic_probe_control.py [T]"""
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

T = int(sys.argv[1]) if len(sys.argv) > 1 else 55555
lib = oq.load_library()
t = time.time()
prob = qp_zoo.control(nx=12, nu=6, T=T)
print("generate %.2f s" % (time.time() - t), file=sys.stderr)
t = time.time()
r = probe(lib, prob, 1)
print("probe %.2f s (includes the probe's own invariant checks)" % (time.time() - t), r, file=sys.stderr)
