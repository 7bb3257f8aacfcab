set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
timeout -k 10 900 python3 -m pytest tests/test_gpu_round2.py tests/test_gpu_stages.py -m gpu -x -q > gpurun_out/r2g/gpu_tests.log 2>&1 && \
timeout -k 10 300 python3 tests/gpu_tune.py --config H --reps 3 --opts "fuse_fmap=0" "fuse_fmap=1" "fuse_fmap=0" "fuse_fmap=1" > gpurun_out/r2g/tune_H_fuse.txt 2>&1 && \
timeout -k 10 300 python3 tests/gpu_tune.py --config C2 --reps 5 --opts "fuse_fmap=0" "fuse_fmap=1" "fuse_fmap=0" "fuse_fmap=1" > gpurun_out/r2g/tune_C2_fuse.txt 2>&1 && \
timeout -k 10 300 python3 tests/gpu_tune.py --config C1 --reps 5 > gpurun_out/r2g/tune_C1.txt 2>&1 && \
timeout -k 10 200 python3 tests/gpu_tune.py --config H --rows 125000 --reps 3 --opts "gram_taper=0" "gram_taper=1" > gpurun_out/r2g/tune_H_125k.txt 2>&1
echo "rc=$?"
tail -n 5 gpurun_out/r2g/gpu_tests.log
