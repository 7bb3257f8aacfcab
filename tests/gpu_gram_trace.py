"""
Diagnostic (not a test): per-workgroup timeline of the Gram launch from the "_trace" build of the library
(make -C scfgp_amd/csrc VARIANT=_trace EXTRA=-DSCFGP_TRACE).  Prints, per XCD and overall, when the last
workgroup started and ended, the spread of job lengths by kind and how much of the launch is tail (time during
which fewer than all workgroup slots are busy).
    SCFGP_LIB_VARIANT=_trace python tests/gpu_gram_trace.py [--config H]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('SCFGP_LIB_VARIANT', '_trace')
import bench                                                          # noqa: E402
from scfgp_amd import synth                                           # noqa: E402
from scfgp_amd.engine import HipEngine                                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='H')
    ap.add_argument('--opts', nargs='*', default=[''])
    a = ap.parse_args()
    N, D, S, M, dtype = bench.CONFIGS[a.config][:5]
    seed = 0x5CF600FF
    X = synth.make_X(seed, N, D)
    y = synth.normal(seed + 9, 0, N).reshape(-1, 1)
    params = synth.make_params(seed + 0x0202, D, S, M, abc=(-1.0, 0.0, -1.0))
    eng = HipEngine(D, S, M, dtype=dtype)
    eng.set_params(params); eng.set_data(X, y)
    for spec in a.opts:
        print('== options [%s]' % spec)
        for kv in [s for s in spec.split(',') if s]:
            k, v = kv.split('=')
            eng.set_option(k, int(v))
        eng.eval(want_grad=True)
        eng.set_profiling(True)
        eng.pass1()                                    # the plain Gram launch is the last one traced
        tm = dict(eng.timings())
        tr = eng.debug_read('trace', (1 << 16, 4), dtype=np.uint64)
        tr = tr[tr[:, 1] > 0].astype(np.int64)
        tr = tr[tr[:, 0] >= tr[:, 1].max() - int(100e6 * 1.5e-3 * max(tm.get('gram', 40.0), 1.0))]   # drop stale entries of wider launches
        t0 = tr[:, 0].min()
        st, en = (tr[:, 0] - t0) / 100.0, (tr[:, 1] - t0) / 100.0       # microseconds (100 MHz clock)
        xcc, kind = tr[:, 2] & 0xF, tr[:, 3]
        hwid = tr[:, 2] >> 8
        cu = xcc * 256 + ((hwid >> 8) & 0xFF)                      # (xcc, se/sh/cu bits of HW_ID)
        print('gram hipEvent %.2f ms; %d workgroups; span %.2f ms' % (tm.get('gram', -1), len(tr), en.max() / 1e3))
        names = {0: '128x128', 1: 'strip', 2: '256x128', 3: 'wide 64x512', 4: 'two 128x128', 5: '256x128 transposed'}
        area = {0: 128 * 128, 1: 64 * 128, 2: 256 * 128, 3: 64 * 512, 4: 2 * 128 * 128, 5: 256 * 128}
        Kp = eng.dims()['Kp']; nfull = Kp // 128 - (1 if (2 * (S + M) + 63) // 64 % 2 else 0); nstrip = Kp // 128 - nfull
        R = nfull // 2; nsb = nstrip * (nfull + 1)
        per_split = {2: R * R, 3: nsb // 4, 4: R // 2, 0: R % 2 + (nfull & 1), 5: (nfull & 1) * R, 1: nsb % 4} if dtype == 'f32' else \
                    {0: nfull * (nfull + 1) // 2, 1: nsb}
        for k in sorted(set(kind.tolist())):
            d = (en - st)[kind == k]
            # all rows pass through every tile of the kind once: flops of the kind / its summed workgroup time = rate per resident
            # workgroup; x 512 resident workgroups = what the chip would deliver on this kind alone
            fl = 2.0 * eng.dims()['Np'] * area[int(k)] * per_split.get(int(k), 0)
            print('  kind %-14s n=%5d  length us: min %.0f  median %.0f  max %.0f   summed %.1f ms   %.1f TFLOP/s at 512 resident workgroups (%.3f of 157.3)'
                  % (names.get(int(k), k), len(d), d.min(), np.median(d), d.max(), d.sum() / 1e3, fl / (d.sum() * 1e-6 / 512) / 1e12,
                     fl / (d.sum() * 1e-6 / 512) / 1e12 / (157.3 if dtype == 'f32' else 78.6)))
        # busy slots over time
        ev = np.concatenate([np.stack([st, np.ones_like(st)], 1), np.stack([en, -np.ones_like(en)], 1)])
        ev = ev[np.argsort(ev[:, 0], kind='stable')]
        busy = np.cumsum(ev[:, 1])
        peak = busy.max()
        t_full_end = ev[np.where(busy >= 0.95 * peak)[0][-1], 0]
        area = np.sum(busy[:-1] * np.diff(ev[:, 0]))
        print('  peak concurrent workgroups %d; last time >=95%% of them busy: %.2f ms (tail %.2f ms); mean occupancy of the slots %.3f'
              % (peak, t_full_end / 1e3, (en.max() - t_full_end) / 1e3, area / (peak * en.max())))
        # per-CU: time with no workgroup at all on the CU
        idle = []
        for cid in sorted(set(cu.tolist())):
            m = cu == cid
            iv = sorted(zip(st[m].tolist(), en[m].tolist()))
            covered, cur_s, cur_e = 0.0, iv[0][0], iv[0][1]
            for s_, e_ in iv[1:]:
                if s_ > cur_e:
                    covered += cur_e - cur_s; cur_s, cur_e = s_, e_
                else:
                    cur_e = max(cur_e, e_)
            covered += cur_e - cur_s
            idle.append(en.max() - covered)
        idle = np.array(idle)
        print('  %d CUs seen; CU time with no workgroup resident: mean %.2f ms, max %.2f ms (= %.1f %% of the launch on average)'
              % (len(idle), idle.mean() / 1e3, idle.max() / 1e3, 100 * idle.mean() / en.max()))
        for x in sorted(set(xcc.tolist())):
            m = xcc == x
            print('  xcc %d: %4d workgroups, first start %.0f us, last start %.2f ms, last end %.2f ms'
                  % (x, m.sum(), st[m].min(), st[m].max() / 1e3, en[m].max() / 1e3))
    eng.close()


if __name__ == '__main__':
    main()
