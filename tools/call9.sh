set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2h
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -n 30 $O/gpu_tests.log; exit 1; }
for cfg in H C3 C5; do
  timeout -k 10 600 bash profiles/collect_pmc.sh r2h_$cfg --config $cfg > $O/pmc_$cfg.log 2>&1 || exit 1
  python3 profiles/pmc_summarize.py gpurun_out/pmc_r2h_$cfg ${cfg}_f32 profiles/r02_pmc_traffic.json > $O/pmc_${cfg}_summary.json || exit 1
done
cp profiles/r02_pmc_traffic.json $O/
for cfg in C1 C2 C3; do
  timeout -k 10 400 python3 bench.py --config $cfg > $O/bench_$cfg.json 2> $O/bench_$cfg.err || exit 1
done
timeout -k 10 500 python3 bench.py --config C4 --steps 5 --warmup 2 > $O/bench_C4.json 2> $O/bench_C4.err || exit 1
timeout -k 10 500 python3 bench.py --config C5 --steps 5 --warmup 2 --cpu-rows 30000 > $O/bench_C5.json 2> $O/bench_C5.err || exit 1
timeout -k 10 500 python3 bench.py > $O/bench_H.json 2> $O/bench_H.err || exit 1
timeout -k 10 300 python3 bench.py --dtype f64 --no-cpu > $O/bench_H_f64.json 2> $O/bench_H_f64.err || exit 1
(cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_H -o H --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-secondary > $GRAFT_REPO_ROOT/$O/prof_H.log 2>&1) || exit 1
echo ALLDONE
