cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02o
mkdir -p $O
timeout 600 python bench.py --workload rand-1e5 --steps 20 --warmup 5 --no-cpu --traffic off > $O/bench_rand1e5_k20w5.json 2>/dev/null; cat $O/bench_rand1e5_k20w5.json | cut -c1-400
timeout 600 python bench.py --workload rand-1e5 --steps 200 --warmup 25 --no-cpu --traffic off > $O/bench_rand1e5_k200.json 2>/dev/null; cat $O/bench_rand1e5_k200.json | cut -c1-400
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --workload rand-1e5 --steps 20 --warmup 5 --no-cpu --traffic off > $O/prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $O/prof -name '*_results.db' | head -1) > $O/kernel_stats_rand1e5.md
O=$O python - <<'PY'
import sqlite3, glob, os
db = glob.glob(os.environ.get('O', '/root/repo/gpurun_out/r02o') + '/prof/**/*_results.db', recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
# the last 400 dispatches: timeline with gaps
tail = rows[-260:]
t0 = tail[0][1]
for name, a, b in tail:
    short = name.split('(')[0].replace('void oq::','').replace('oq::','')[:60]
    print("%9.1f %7.1f  %s" % ((a - t0)/1e3, (b - a)/1e3, short))
PY
rm -rf $O/prof
