set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
timeout -k 10 400 python3 bench.py > gpurun_out/r2a/bench_H.json 2> gpurun_out/r2a/bench_H.err && \
timeout -k 10 200 python3 tests/gpu_gram_trace.py --config H > gpurun_out/r2a/gram_trace_H.txt 2>&1 && \
timeout -k 10 500 bash profiles/collect_pmc.sh r2a_H --config H > gpurun_out/r2a/pmc_H.log 2>&1 && \
python3 profiles/pmc_summarize.py gpurun_out/pmc_r2a_H H_f32 gpurun_out/r2a/pmc_traffic.json > gpurun_out/r2a/pmc_H_summary.json && \
(cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2a/c1 -o c1 --output-format csv -- python3 $GRAFT_REPO_ROOT/tests/gpu_tune.py --config C1 --reps 3 > $GRAFT_REPO_ROOT/gpurun_out/r2a/c1.log 2>&1) && \
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r2a/gpu_tests.log 2>&1
echo "rc=$?"
tail -3 gpurun_out/r2a/gpu_tests.log
