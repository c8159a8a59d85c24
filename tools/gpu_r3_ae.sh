#!/bin/bash
# (dispatch timelines of rand-1e5: profiles/r03_rand1e5_timeline.md)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3ae; mkdir -p $O
for cfg in "A=1" "OSQP_AMD_PCG_SPEC=0"; do
  cd /tmp; rm -rf /tmp/prof_tl
  env $cfg timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_tl -o tl -- python $GRAFT_REPO_ROOT/bench.py --workload rand-1e5 --steps 100 --warmup 25 --no-cpu --traffic off > /dev/null 2>&1
  DB=$(find /tmp/prof_tl -name "*_results.db" | head -1)
  cd $GRAFT_REPO_ROOT
  python tools/rocpd_timeline.py $DB 1400 > $O/timeline_$(echo $cfg | tr '=' '_').md
  tail -22 $O/timeline_$(echo $cfg | tr '=' '_').md
done
