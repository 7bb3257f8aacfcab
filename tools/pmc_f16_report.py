"""profiles/r05_f16x3_pmc.txt from the passes of tools/pmc_f16.sh: python tools/pmc_f16_report.py gpurun_out/prof/pmc_f16.txt gpurun_out/pmc_f16 > profiles/..."""
import collections
import csv
import sys

import numpy as np

txt, d = sys.argv[1], sys.argv[2] + '/busy/'
print('# f16x3 mode at the headline shape (tests/gpu_tune.py --config H --dtype f16x3): rocprofv3 PMC passes, one counter group per pass (tools/pmc_f16.sh);')
print('# per-launch means.  FETCH_SIZE / WRITE_SIZE are in KB; fabric-side bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB (gfx950 tallies 128-B read')
print('# requests at 64 B: profiles/pmc_summarize.py).')
print(''.join(l for l in open(txt) if not l.startswith('+')))
print("# clock and matrix-pipe occupancy from the 'busy' pass: GRBM_GUI_ACTIVE / 8 XCDs / kernel duration; SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / that")
dur = collections.defaultdict(list)
for r in csv.DictReader(open(d + 'busy_kernel_trace.csv')):
    k = r['Kernel_Name'].split('(')[0][:40]
    if 'f16' in k or 'split_' in k:
        dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(d + 'busy_counter_collection.csv')):
    k = r['Kernel_Name'].split('(')[0][:40]
    if 'f16' in k or 'split_' in k:
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in dur:
    ms = np.mean(dur[k]); gui = np.mean(acc[k]['GRBM_GUI_ACTIVE']) / 8; busy = np.mean(acc[k]['SQ_VALU_MFMA_BUSY_CYCLES']) / 1024
    print('%-40s %.2f ms  clock %.2f GHz  mfma busy %.2f  busy x clock / 2.4 GHz = %.2f' % (k, ms, gui / ms / 1e6, busy / gui, busy / ms / 1e6 / 2.4))
