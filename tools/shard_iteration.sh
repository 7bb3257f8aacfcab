#!/bin/bash
# what a TRAINING ITERATION costs on a row shard of the headline problem: the bare evaluation against the
# reference-shaped triple (host rule + parameter upload + residency check) and the on-device loop (scfgp_train), as the captured
# graph and as what a rank of a sharded job runs: eager launches with the three sums inside the library (one-rank communicator)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-shard_iter}
mkdir -p $O
for rows in 1000000 125000; do
  timeout -k 10 600 python3 bench.py --rows $rows --steps 10 --warmup 3 --no-cpu --triple > $O/rows_$rows.json 2> $O/rows_$rows.err || { tail $O/rows_$rows.err; exit 1; }
done
python3 - <<PY
import json
print('rows      bare eval   triple (frozen arrays)   triple (writeable)   device loop (graph)   device loop, sums inside (eager)   [ms per iteration]')
for rows in (1000000, 125000):
    o = json.load(open('$O/rows_%d.json' % rows)); t = o['secondary']['through_triple']
    g = lambda k, f: t[k][f] if f in t.get(k, {}) else float('nan')
    print('%8d  %8.2f   %8.2f                 %8.2f             %8.2f              %8.2f'
          % (rows, o['ms_per_step_median'], g('read_only_arrays', 'ms_per_call'), g('writeable_arrays', 'ms_per_call'),
             g('device_loop', 'ms_per_iteration'), g('device_loop_sums_inside', 'ms_per_iteration')))
    for k in ('device_loop', 'device_loop_sums_inside'):
        if 'error' in t.get(k, {}): print('   ', k, t[k]['error'])
PY
echo ALLDONE
