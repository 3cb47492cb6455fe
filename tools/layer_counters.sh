#!/bin/bash
# SQ counters of ONE layer under ONE launch configuration (two --pmc passes, no trace flags):
#   tools/layer_counters.sh D40 variant=20   -> gpurun_out/ctr_D40_variant=20.txt  (per-launch means, tools/pmc_generic.py)
# Run through gpurun.  The harness is tools/layer_time.py (one convolution + fused SiLU at batch 256).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
L=$1; CFG=${2:-default}; EXTRA=${3:-}
O=$R/gpurun_out/ctr_${L}_${CFG}
rm -rf $O; mkdir -p $O
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
P2="SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P3="SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT SQ_LDS_UNALIGNED_STALL"
for i in 1 2 3; do
  eval P=\$P$i
  rocprofv3 --pmc $P --output-format csv -d $O/p$i -o p -- python3 $R/tools/layer_time.py $L --cfg $CFG --no-oracle $EXTRA > $O/p$i.log 2>&1 || { tail -5 $O/p$i.log; }
done
python3 $R/tools/pmc_generic.py $O | grep -i "conv" > $O.txt
cat $O.txt
