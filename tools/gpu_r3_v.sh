#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3v; mkdir -p $O
export OSQP_AMD_SKIP_RAND1E6=1
echo "== poison, no VMM"
OSQP_AMD_POISON=1 OSQP_AMD_VMM=0 timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_full_size_gpu.py > $O/pytest_poison.log 2>&1; echo "rc=$?"; grep -E "passed|failed|FAILED|Error" $O/pytest_poison.log | head -20
echo "== poison, VMM from 1 MiB"
OSQP_AMD_POISON=1 OSQP_AMD_VMM_MIN_MB=1 timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_full_size_gpu.py > $O/pytest_poison_vmm.log 2>&1; echo "rc=$?"; grep -E "passed|failed|FAILED|Error" $O/pytest_poison_vmm.log | head -20
