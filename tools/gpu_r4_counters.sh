# PMC counters behind the "issue-bound" statements of round 4: the batched kernel (vector instruction mix, VALU activity, LDS
# conflicts, occupancy) and the block sweeps of the dense inverse (matrix-core utilisation).  Separate passes per counter group.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04_counters; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
: > $O/pmc_mpc-batch_counters.txt
for c in "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "LDSBankConflict" "MeanOccupancyPerCU" "SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM"; do
  d=$O/p_$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c -d $d -o p -- python $GRAFT_REPO_ROOT/bench.py --workload mpc-batch --steps 10 --warmup 2 --no-cpu --traffic off > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $(find $d -name '*_results.db' | head -1) k_batch >> $O/pmc_mpc-batch_counters.txt
  rm -rf $d
done
: > $O/pmc_equality_qp_counters.txt
for c in "MfmaUtil" "SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU SQ_BUSY_CYCLES"; do
  d=$O/e_$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c -d $d -o p -- python $GRAFT_REPO_ROOT/tools/equality_qp_profile.py > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $(find $d -name '*_results.db' | head -1) k_gj >> $O/pmc_equality_qp_counters.txt
  python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $(find $d -name '*_results.db' | head -1) k_dense_apply >> $O/pmc_equality_qp_counters.txt
  rm -rf $d
done
cat $O/pmc_mpc-batch_counters.txt; cat $O/pmc_equality_qp_counters.txt
