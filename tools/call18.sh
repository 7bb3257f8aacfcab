set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2q
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --durations=8 > $O/gpu_tests.log 2>&1 || { tail -n 40 $O/gpu_tests.log; exit 1; }
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail $O/smoke.log; exit 1; }
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $O/bench_H.json 2> $O/bench_H.err || { tail $O/bench_H.err; exit 1; }
timeout -k 10 400 python3 bench.py --gpus 2 --backend gloo --config H --rows 250000 --dtype f64 --steps 3 --warmup 1 --no-cpu > $O/bench_gloo2.json 2> $O/bench_gloo2.err || { tail $O/bench_gloo2.err; exit 1; }
(cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_H -o H --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-secondary > $GRAFT_REPO_ROOT/$O/prof_H.log 2>&1) || exit 1
(cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --marker-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_C2 -o C2 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --config C2 --no-cpu > $GRAFT_REPO_ROOT/$O/prof_C2.log 2>&1) || echo "marker trace run failed"
echo ALLDONE
