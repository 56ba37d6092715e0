#!/bin/bash
# Where does the dominant kernel wait?  A handful of rocprofv3 counter groups, one run each (GPU box; run from the repo root).
# usage: bash tools/pmc_groups.sh <tag> [bench.py workload arguments]     -> gpurun_out/pmcg_<tag>/summary.txt
tag=$1; shift
out=gpurun_out/pmcg_$tag; mkdir -p $out; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "TCC_REQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/g$i -- python3 bench.py "$@" --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $out/g$i.log 2>&1 || { echo "group $i failed: $grp" >> $out/summary.txt; continue; }
  for c in $grp; do python tools/pmc_summary.py $out/g$i $c mpcqp_oc_ 3 >> $out/summary.txt; done
done
cat $out/summary.txt
