#!/bin/bash
# A/B/C... of library variants on ONE box: bench.py (no secondary legs), the variants interleaved, REPS rounds
# usage: bash tools/ab_multi.sh <outdir> "<variants, '-' for the default build>" [bench args]      e.g. "- _prev _k1"
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-ab}; VS=$2; shift; shift
mkdir -p $O
for rep in 1 2 ${REPS3:+3}; do
  for vv in $VS; do
    v=$vv; [ "$v" = "-" ] && v=""
    SCFGP_LIB_VARIANT=$v timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-secondary --no-cpu "$@" > $O/b${v}_$rep.json 2> $O/b${v}_$rep.err || { tail $O/b${v}_$rep.err; exit 1; }
    python3 - <<PY
import json
o = json.load(open('$O/b${v}_$rep.json'))
print('variant [%s] rep $rep: %.2f ms' % ('$v', o['ms_per_step']), {k: round(x, 2) for k, x in o['stages_ms'].items() if k in ('gram', 'gram_w', 'apply_v', 'apply_phibar', 'kstage_factor', 'xtz', 'featuremap')})
PY
  done
done
echo ALLDONE
