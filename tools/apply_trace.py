"""Per-XCD timeline of the LDS-DMA apply launch (diagnostic _trace build): do the eight XCDs finish their static shares of the
tile list together?   SCFGP_LIB_VARIANT=_trace python tools/apply_trace.py [config] [apply_v|apply_phibar|apply_c|apply_vc] [opt=value ...]
(apply_c / apply_vc: the triangular products of the factor form; give apply_dma=1 or 2 and gram64=3 to put them on the LDS-DMA
kernel -- per column tile: stages, k-loop time per stage, fixed cost around the loop)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('SCFGP_LIB_VARIANT', '_trace')
import bench
from scfgp_amd import synth
from scfgp_amd.engine import HipEngine
cfg = sys.argv[1] if len(sys.argv) > 1 else 'H'
N, D, S, M, dtype = bench.CONFIGS[cfg][:5]
X = synth.make_X(bench.SEED, N, D); y = synth.normal(bench.SEED + 9, 0, N).reshape(-1, 1)
params = synth.make_params(bench.SEED + 0x0202, D, S, M, abc=(-1.0, 0.0, -1.0))
eng = HipEngine(D, S, M, dtype=dtype)
for kv in sys.argv[3:]:
    k, v = kv.split('='); eng.set_option(k, int(v))
eng.set_params(params); eng.set_data(X, y)
eng.eval(); eng.set_profiling(True)
which = sys.argv[2] if len(sys.argv) > 2 else 'apply_v'         # or apply_phibar
eng.pass1(); eng.factor()                                        # the last LDS-DMA launch traced: the 128- or 256-wide tiles
if which in ('apply_v', 'apply_c', 'apply_vc'): eng.pass2(False)
else: eng.pass2(True); eng.adjoint(); eng.pass3()
tm = dict(eng.timings())
epi = {'apply_v': 0, 'apply_phibar': 1, 'apply_c': 3, 'apply_vc': 4}[which]
tr = eng.debug_read('apply_trace', (5 << 16, 5), dtype=np.uint64)[epi << 16:(epi + 1) << 16]
tr = tr[tr[:, 1] > 0].astype(np.int64)
tr = tr[tr[:, 0] >= tr[:, 1].max() - int(100e6 * 1.2e-3 * tm[which])]
# the short launches that follow in the same stage (ragged 64 columns, 128-wide tiles of a thin last round) overwrite the first
# entries of the stamp array: keep the 256-wide launch's own workgroups
dur = tr[:, 1] - tr[:, 0]
if which in ('apply_v', 'apply_phibar'):
    tr = tr[dur > 0.6 * np.median(dur)]
t0 = tr[:, 0].min(); st, en, xcc = (tr[:, 0] - t0) / 100.0, (tr[:, 1] - t0) / 100.0, tr[:, 2] & 0xF
print('%s hipEvent %.2f ms; %d workgroups traced; span %.2f ms' % (which, tm[which], len(tr), en.max() / 1e3))
pro, loop, epi = (tr[:, 3] - tr[:, 0]) / 100.0, (tr[:, 4] - tr[:, 3]) / 100.0, (tr[:, 1] - tr[:, 4]) / 100.0
for nm, a in (('start -> first barrier passed', pro), ('k loop', loop), ('epilogue', epi)):
    print('  %-30s us: mean %.2f  p10 %.2f  median %.2f  p90 %.2f  max %.2f' % (nm, a.mean(), np.percentile(a, 10), np.median(a), np.percentile(a, 90), a.max()))
jt, nst = (tr[:, 2] >> 8) & 0xFFF, tr[:, 2] >> 20
if which in ('apply_c', 'apply_vc'):
    print('  per column tile: stages, workgroups, k loop us per stage, start->loop us, epilogue us, share of the summed workgroup time')
    tot = (en - st).sum()
    for j in sorted(set(jt.tolist())):
        m = jt == j
        print('    jt %2d  nst %3d  n %5d  %.3f us/stage  pro %.2f  epi %.2f  share %.3f' % (j, int(np.median(nst[m])), m.sum(), (loop[m] / np.maximum(nst[m], 1)).mean(), pro[m].mean(), epi[m].mean(), (en - st)[m].sum() / tot))
    print('  summed: k loops %.1f ms, around them %.1f ms (%.1f %%)' % (loop.sum() / 1e3, (pro + epi).sum() / 1e3, 100 * (pro + epi).sum() / (en - st).sum()))
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    print('  xcc %d: %5d workgroups, mean length %.1f us, last start %.2f ms, last end %.2f ms' % (x, m.sum(), (en - st)[m].mean(), st[m].max() / 1e3, en[m].max() / 1e3))
ev = np.concatenate([np.stack([st, np.ones_like(st)], 1), np.stack([en, -np.ones_like(en)], 1)]); ev = ev[np.argsort(ev[:, 0], kind='stable')]
busy = np.cumsum(ev[:, 1]); area = np.sum(busy[:-1] * np.diff(ev[:, 0]))
print('  peak resident workgroups %d, mean occupancy of the slots %.3f' % (busy.max(), area / (busy.max() * en.max())))
eng.close()
