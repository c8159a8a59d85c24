# round 6: the next pivot sweep inside the update launch, the panel through LDS: tests (dense top, dense block of the level schedule), A/B, profile
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_snd; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0 OSQP_AMD_BENCH_OTHERS=0
timeout 1200 python -m pytest tests/test_multifrontal_gpu.py -m gpu -q -x -k "dense_top or pivot_block" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_problem_zoo.py tests/test_gpu_parity.py -m gpu -q -x -k "dense or equality or portfolio or zoo" 2>&1 | tail -3
for f in 1 0; do
  REFACTOR_GRID=1 OSQP_AMD_GJ_FUSE=$f timeout 600 python tools/refactor_time.py --child 700 2>&1 | grep "T=" | sed "s/^/fuse=$f /"
done | tee $O/gj_fuse_ab.txt
ZOO_LABELS=gpu_direct OSQP_AMD_SETUP_TRACE=1 timeout 600 python tools/zoo_rates.py equality_qp 2>&1 | grep -E "numeric|it_per_s" | cut -c1-200
bash tools/gpu_r6_snd_prof.sh 2>&1 | grep -A12 "^factorisation" | tail -13
