#!/bin/bash
# rocprofv3 PMC passes of one evaluation (tests/gpu_tune.py), one counter group per pass as
# MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with
# trace domains).  Usage on the GPU box, from the repo root:
#     bash profiles/collect_pmc.sh <tag> [gpu_tune.py arguments...]
# writes gpurun_out/pmc_<tag>/<pass>/..._counter_collection.csv; summarise with profiles/pmc_summarize.py
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "busy:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    echo "== pass $name: $ctrs"
    rocprofv3 --pmc $ctrs --kernel-trace -d $out/$name -o $name --output-format csv -- \
        python3 $root/tests/gpu_tune.py --reps 1 "$@" > $out/$name.log 2>&1
    grep -E "^\[|total" $out/$name.log | tail -2 || true
done
