#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_dma; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
for pass in "a:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "c:TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "d:SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    timeout -k 10 200 rocprofv3 --pmc $ctrs --kernel-trace -d $O/$name -o $name --output-format csv -- python3 $R/tests/gpu_tune.py --config H --rows 262144 --dtype bf16x3 --reps 1 > $O/$name.log 2>&1 || { tail -n 5 $O/$name.log; }
done
echo ALLDONE
