set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
timeout -k 10 300 python3 tests/gpu_gram_trace.py --config H --opts "" "gram_wgs=1024" "gram_wgs=2048" "gram_wgs=512,gram_w_sq=20,gram_w_strip=15" "gram_wgs=512,gram_w_sq=16,gram_w_strip=10" > gpurun_out/r2b/gram_trace_H.txt 2>&1 && \
timeout -k 10 300 python3 tests/gpu_tune.py --config H --reps 3 --opts "" "gram_wgs=1024" "gram_wgs=512,gram_w_sq=20,gram_w_strip=15" "gram_wgs=512,gram_w_sq=18,gram_w_strip=12" > gpurun_out/r2b/tune_H.txt 2>&1 && \
timeout -k 10 200 python3 tests/gpu_tune.py --config H --dtype f64 --rows 500000 --reps 2 --opts "" "gram_wgs=1024" "gram_w_strip=10" > gpurun_out/r2b/tune_H_f64.txt 2>&1 && \
timeout -k 10 500 bash profiles/collect_pmc.sh r2b_H --config H > gpurun_out/r2b/pmc_H.log 2>&1 && \
python3 profiles/pmc_summarize.py gpurun_out/pmc_r2b_H H_f32 gpurun_out/r2b/pmc_traffic.json > gpurun_out/r2b/pmc_H_summary.json && \
(cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2b/c1 -o c1 --output-format csv -- python3 $GRAFT_REPO_ROOT/tests/gpu_tune.py --config C1 --reps 3 > $GRAFT_REPO_ROOT/gpurun_out/r2b/c1.log 2>&1) && \
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r2b/gpu_tests.log 2>&1
echo "rc=$?"
tail -5 gpurun_out/r2b/gpu_tests.log
