#!/bin/bash
# where the LDS-DMA apply kernels start to pay: bench.py at small shapes with apply_dma forced off / on
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-dmathr}; mkdir -p $O
for dt in f64 f32; do for M in 112 256 480; do for rows in 16384 32768 65536 100000; do
  line="$dt M=$M rows=$rows:"
  for o in 0 1 2; do
    [ $dt = f64 ] && [ $o = 2 ] && continue
    timeout -k 10 200 python3 bench.py --config C2 --dtype $dt --M $M --rows $rows --steps 20 --warmup 5 --no-secondary --no-cpu --opt apply_dma=$o > $O/x.json 2> $O/x.err || { tail $O/x.err; exit 1; }
    line="$line  dma=$o $(python3 -c "
import json; o=json.load(open('$O/x.json')); st=o['stages_ms']; print('%.3f (%.3f+%.3f)' % (o['ms_per_step'], st['apply_v'], st['apply_phibar']))")"
  done
  echo "$line"
done; done; done
echo ALLDONE
