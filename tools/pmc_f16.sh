#!/bin/bash
# PMC passes over one f16x3 evaluation at the headline shape: matrix-pipe occupancy and the LDS side of the two f16x3 kernels
# (counters alone, no trace domains beyond --kernel-trace).  On the GPU box, from the repo root: bash tools/pmc_f16.sh
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_f16
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "busy:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA" "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    echo "== pass $name: $ctrs"
    rocprofv3 --pmc $ctrs --kernel-trace -d $out/$name -o $name --output-format csv -- \
        python3 $root/tests/gpu_tune.py --reps 1 --config H --dtype f16x3 > $out/$name.log 2>&1
    python3 - $out/$name <<'PY'
import sys, glob, csv, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0][:60]
    if 'f16' not in k and 'split_' not in k: continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k, d in acc.items():
    print(k, {c: '%.4g' % (v / cnt[(k, c)]) for c, v in d.items()}, 'launches', max(cnt[(k, c)] for c in d))
PY
done
