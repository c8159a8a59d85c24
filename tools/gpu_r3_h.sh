# round 3, call H: whole gpu suite on the current state + bench lines of the four workloads
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3h
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3h/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3h/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
for w in rand-1e5 lasso-5e5 mpc-batch; do
  timeout 600 python bench.py --workload $w --steps 20 --warmup 5 > gpurun_out/r3h/bench_$w.json 2>gpurun_out/r3h/bench_$w.err; echo "$w rc=$?"; cut -c1-250 gpurun_out/r3h/bench_$w.json
done
