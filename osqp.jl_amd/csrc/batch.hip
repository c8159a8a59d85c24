// batch.hip -- batched small-QP path (row K11): placeholder.
#include "engine.hpp"
extern "C" {
c_int osqp_amd_batch_solve(c_int, c_int, c_int, const c_int *, const c_int *, const c_float *, const c_int *, const c_int *,
                           const c_float *, const c_float *, const c_float *, const c_float *, const OSQPSettings *, c_float *,
                           c_float *, OSQPInfo *, c_int) {
  oq::set_last_error("batched path not built yet");
  return 6;
}
c_int osqp_amd_batch_solve_generated(c_int, c_int, unsigned long long, const OSQPSettings *, c_float *, c_float *, c_float *, c_int) {
  oq::set_last_error("batched path not built yet");
  return 6;
}
}
