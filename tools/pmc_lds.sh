#!/bin/bash
# LDS / MFMA counters of the bf16 conv microbench:  tools/pmc_lds.sh <layer> [dtype]     (GPU box; separate --pmc passes)
L=${1:-conv4_2}; D=${2:-bf16}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_lds; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
    T=$(echo $C | tr ' ' '_')
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pl_$T -o c -- python $R/tools/bench_conv.py $L $D > /dev/null 2>&1
    echo "== $C"; python $R/tools/pmc_summary.py /tmp/pl_$T/c_counter_collection.csv conv_ 2>&1 | grep -v "^$"
done > $O/${L}_${D}.txt 2>&1
tail -80 $O/${L}_${D}.txt
