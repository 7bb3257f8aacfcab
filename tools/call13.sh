set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2l
mkdir -p $O
SCFGP_LIB_VARIANT=_w2 timeout -k 10 300 python3 tests/gpu_tune.py --config H --rows 262144 --dtype bf16x3 --reps 3 > $O/tune_bf3_w2.txt 2>&1 || { tail $O/tune_bf3_w2.txt; exit 1; }
echo ALLDONE
