"""osqp.jl_amd -- MI355X-native OSQP ADMM engine behind the OSQP.jl boundary.

Layout (only what the hot path needs):
  csrc/        HIP kernels + the C-ABI library libosqp_amd.so (include/osqp_amd.h)
  types.py     ctypes mirrors of the ABI structs      [REF src/types.jl]
  constants.py status codes, updatable lists          [REF src/constants.jl]
  interface.py Model / setup / solve / update / ...   [REF src/interface.jl]
  modcaches.py modification / warm-start caches       [REF src/modcaches.jl]
  moi.py       the MathOptInterface face              [REF src/MOI_wrapper.jl]
  batch.py     batched small-QP path + multi-GPU shard/gather
  sharded.py   communicators of the row-sharded / batched multi-GPU paths
  julia/       the same host layer in Julia (cannot be executed in this image)
"""
from .constants import *  # noqa: F401,F403
from .interface import (  # noqa: F401
    Model, Results, Info, OSQPError, setup, setup_generated, solve, update, update_settings, warm_start,
    warm_start_x, warm_start_y, warm_start_x_y, update_q, update_l, update_u, update_bounds, update_P, update_A,
    update_P_A, clean, version, dimensions, default_settings, make_settings, stats, ManagedCcsc, ccsc_to_scipy,
)
from .types import load_library, PRODUCT_LIB_PATH, ORACLE_LIB_PATH  # noqa: F401
