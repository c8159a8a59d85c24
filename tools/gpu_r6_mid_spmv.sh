# round 6, bounded experiment (kill criterion written first): mid-size SpMV geometry on rand-1e5.
# A 64 KB x panel (8192 columns: OSQP_AMD_PANEL_SHIFT=13) lets two workgroups share a compute unit and halves the tile a workgroup
# needs to amortise its panel load, so >= 306 tiles become possible where the 128 KB panel gives 153 - 217 tiles on 256 CUs.
# KEEP ONLY IF rand-1e5 at the driver's window (K = 20, W = 5) reaches >= 5 600 it/s (5 100 today) with bit-identical products
# (tests/test_gpu_parity.py::test_panel_spmv_matches_scipy); otherwise this sweep is the record and the item is closed.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_mid_spmv; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0 OSQP_AMD_BENCH_OTHERS=0
for shift in 14 13; do
  for tile in auto 46000 32768 24576 16384; do
    if [ $tile = auto ]; then unset OSQP_AMD_PANEL_TILE_NNZ; else export OSQP_AMD_PANEL_TILE_NNZ=$tile; fi
    OSQP_AMD_PANEL_SHIFT=$shift timeout 300 python bench.py --workload rand-1e5 --no-cpu --traffic off --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('shift=$shift tile_nnz=$tile: %.1f it/s  %.4f ms/step  spmv %.4f ms (frac %.3f)  to eps %.4f s' % (d['value'], d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline']['frac'], d['time_to_eps_s']))"
  done
done | tee $O/rand1e5_panel_shift_sweep.txt
