#!/bin/bash
# what one row shard of the headline problem costs on one GPU: rows = 1e6 / ranks (compute-side ceiling of strong scaling)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-shards}
mkdir -p $O
for rows in 1000000 500000 250000 125000; do
  timeout -k 10 300 python3 bench.py --rows $rows --steps 10 --warmup 3 --no-secondary --no-cpu > $O/rows_$rows.json 2> $O/rows_$rows.err || { tail $O/rows_$rows.err; exit 1; }
done
python3 - <<PY
import json
base = None
print('rows      ms/eval  vs linear   gram   gram_w  apply_v  apply_phibar  kstage  fmap   xtz   rest')
for rows in (1000000, 500000, 250000, 125000):
    o = json.load(open('$O/rows_%d.json' % rows)); st = o['stages_ms']; ms = o['ms_per_step_median']
    if base is None: base = ms
    lin = base * rows / 1e6
    named = sum(st.get(k, 0) for k in ('gram', 'gram_w', 'apply_v', 'apply_phibar', 'kstage_factor', 'featuremap', 'xtz'))
    print('%8d  %7.2f  %6.3f   %6.2f  %6.2f  %6.2f   %6.2f       %5.2f  %5.2f  %5.2f  %5.2f'
          % (rows, ms, lin / ms, st['gram'], st['gram_w'], st['apply_v'], st['apply_phibar'], st['kstage_factor'], st['featuremap'], st['xtz'], ms - named))
PY
echo ALLDONE
