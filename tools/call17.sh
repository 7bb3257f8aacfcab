set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2p
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_size" --durations=5 > $O/gpu_tests.log 2>&1 || { tail -n 40 $O/gpu_tests.log; exit 1; }
echo ALLDONE
