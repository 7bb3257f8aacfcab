set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2k
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $O/counters.txt 2>&1
export SCFGP_LIB_VARIANT=_w2
for pass in "a:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "b:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE" "c:TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    timeout -k 10 200 rocprofv3 --pmc $ctrs --kernel-trace -d $O/$name -o $name --output-format csv -- python3 $GRAFT_REPO_ROOT/tests/gpu_tune.py --config H --rows 262144 --dtype bf16x3 --reps 1 > $O/$name.log 2>&1 || { tail -n 5 $O/$name.log; }
done
echo ALLDONE
