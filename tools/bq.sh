#!/bin/bash
# compile csrc/batch.hip with --save-temps and print register / spill statistics of a kernel (default: k_batch_quad)
cd /root/repo/osqp.jl_amd/csrc || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -Wno-c++20-extensions -c batch.hip -o build/batch.o --save-temps=obj 2>&1 | grep -v "^$" | head -30
/root/repo/tools/isa_stats.sh ${1:-k_batch_quad}
awk '/scratch_load/{l++} /scratch_store/{s++} /s_barrier/{printf "B%d: L%d S%d F%d | ", NR, l, s, f; l=0; s=0; f=0} /v_fmac_f64_dpp/{f++}' /tmp/kern.s | fold -w 180 | head -8
