#!/bin/bash
# usage: tools/isa_stats.sh <kernel-name-substring>   (after `hipcc --save-temps=obj` of batch.hip in csrc/build)
S=/root/repo/osqp.jl_amd/csrc/build/batch-hip-amdgcn-amd-amdhsa-gfx950.s
grep -A40 "\.name:.*$1" $S | grep -E "vgpr|sgpr_spill|private|wavefront" | head -6
L=$(grep -n "^_Z.*$1.*:" $S | head -1 | cut -d: -f1)
sed -n "${L},\$p" $S | awk '/s_endpgm/{print; exit} {print}' > /tmp/kern.s
echo "lines $(wc -l < /tmp/kern.s)"
for k in scratch_load scratch_store v_fmac_f64_dpp s_barrier s_cbranch v_accvgpr; do echo "$k $(grep -c $k /tmp/kern.s)"; done
