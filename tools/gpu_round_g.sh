set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02g
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r02g/pytest.log 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/r02g/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --traffic off 2>/dev/null | cut -c1-1200
