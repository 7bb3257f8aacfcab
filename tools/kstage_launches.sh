#!/bin/bash
# per-launch durations and gaps of the K x K stage's kernels from a rocprofv3 kernel trace (gpurun_out/$1/)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-ks}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf $O/trace
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/trace -o ks --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/kstage_run.py ${2:-H} > $O/run.log 2>&1 || { tail $O/run.log; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob('$O/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last evaluation: find the last run of chol_step_kernel launches
idx = [i for i, r in enumerate(rows) if 'chol_step_kernel' in r['Kernel_Name']]
last = idx[-1]; first = last
while first - 1 in idx: first -= 1
seq = rows[first:last + 1]
t0 = int(seq[0]['Start_Timestamp'])
print('chol_step launches of the last evaluation: %d, first start to last end %.1f us' % (len(seq), (int(seq[-1]['End_Timestamp']) - t0) / 1e3))
for k, r in enumerate(seq):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - int(seq[k - 1]['End_Timestamp'])) / 1e3 if k else 0.0
    print('  p=%2d grid %5s  start %8.1f  dur %6.1f us  gap %5.1f us' % (k, r.get('Grid_Size', r.get('Grid_Size_X', '?')), (s - t0) / 1e3, (e - s) / 1e3, gap))
# everything between the unpack and the scalars
names = ['kstage_unpack_kernel', 'gemv_rows_kernel', 'factor_scalars_kernel']
for r in rows[first - 3:last + 4]:
    if any(n in r['Kernel_Name'] for n in names):
        print('  %-28s start %8.1f dur %6.1f us' % (r['Kernel_Name'][:28], (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
PY
