#!/bin/bash
# one scfgp_set_option swept over a list of values at any bench shape, one process per value on ONE box
# usage: bash tools/opt_sweep.sh <outdir> <option> "<values>" "<stage names to print>" [bench args]
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-sweep}; OPT=$2; L=$3; ST=$4; shift; shift; shift; shift; mkdir -p $O
for v in $L; do
  timeout -k 10 400 python3 bench.py --steps 8 --warmup 3 --no-secondary --no-cpu --opt $OPT=$v "$@" > $O/${OPT}_$v.json 2> $O/${OPT}_$v.err || { tail $O/${OPT}_$v.err; exit 1; }
  python3 - <<PY
import json
o = json.load(open('$O/${OPT}_$v.json')); st = o['stages_ms']
print('$* $OPT = %6d: %.2f ms  ' % ($v, o['ms_per_step_median']) + '  '.join('%s %.3f' % (k, st.get(k, float('nan'))) for k in '$ST'.split()), flush=True)
PY
done
echo ALLDONE
