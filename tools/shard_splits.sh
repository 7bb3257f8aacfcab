#!/bin/bash
# Gram row splits at a row shard: bench.py --rows R with gram_nsplit = each value of the list (0 = the library's own choice)
# usage: bash tools/shard_splits.sh <outdir> <rows> "<nsplit values>"
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-splits}; R=${2:-125000}; mkdir -p $O
for ns in ${3:-0 24 32 40 48 64}; do
  timeout -k 10 300 python3 bench.py --rows $R --steps 10 --warmup 3 --no-secondary --no-cpu --opt gram_nsplit=$ns > $O/ns_$ns.json 2> $O/ns_$ns.err || { tail $O/ns_$ns.err; exit 1; }
  python3 - <<PY
import json
o = json.load(open('$O/ns_$ns.json')); st = o['stages_ms']
print('rows $R gram_nsplit %3d: %.2f ms  gram %.2f  gram_w %.2f  reduce_tiles %.2f' % ($ns, o['ms_per_step_median'], st['gram'], st['gram_w'], st.get('reduce_tiles', 0)))
PY
done
echo ALLDONE
