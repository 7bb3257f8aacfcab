#!/bin/bash
# Gram row splits at any bench shape: bench.py <bench args> with gram_nsplit = each value of the list (0 = the library's own choice)
# usage: bash tools/gram_splits.sh <outdir> "<nsplit values>" [bench args, e.g. --config C5]
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-splits}; L=${2:-0}; shift; shift; mkdir -p $O
for ns in $L; do
  timeout -k 10 400 python3 bench.py --steps 5 --warmup 2 --no-secondary --no-cpu --opt gram_nsplit=$ns "$@" > $O/ns_$ns.json 2> $O/ns_$ns.err || { tail $O/ns_$ns.err; exit 1; }
  python3 - <<PY
import json
o = json.load(open('$O/ns_$ns.json')); st = o['stages_ms']
print('$* gram_nsplit %3d: %.2f ms  gram %.2f  gram_w %.2f  reduce_tiles %.2f' % ($ns, o['ms_per_step_median'], st['gram'], st['gram_w'], st.get('reduce_tiles', 0)), flush=True)
PY
done
echo ALLDONE
