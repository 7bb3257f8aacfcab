set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2n
mkdir -p $O
for v in _w2 _d1 _d2 _d3; do
  SCFGP_LIB_VARIANT=$v timeout -k 10 200 python3 tests/gpu_tune.py --config H --rows 262144 --dtype bf16x3 --reps 2 > $O/tune$v.txt 2>&1 || { tail -n 3 $O/tune$v.txt; }
done
echo ALLDONE
