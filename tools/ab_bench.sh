#!/bin/bash
# A/B of library variants on ONE box: bench.py (no secondary legs) alternating default and variant builds
# usage: bash tools/ab_bench.sh <outdir> <variant e.g. _late> [bench args]
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-ab}; V=$2; shift; shift
mkdir -p $O
for rep in 1 2; do
  for v in "" "$V"; do
    SCFGP_LIB_VARIANT=$v timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-secondary --no-cpu "$@" > $O/b${v}_$rep.json 2> $O/b${v}_$rep.err || { tail $O/b${v}_$rep.err; exit 1; }
    python3 - <<PY
import json
o = json.load(open('$O/b${v}_$rep.json'))
print('variant [%s] rep $rep: %.2f ms' % ('$v', o['ms_per_step']), {k: round(x, 2) for k, x in o['stages_ms'].items() if k in ('gram', 'gram_w', 'apply_v', 'apply_phibar', 'kstage_factor', 'xtz', 'featuremap')})
PY
  done
done
echo ALLDONE
