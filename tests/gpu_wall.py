"""
Diagnostic (not a test): wall time of one evaluation against the sum of its hipEvent stage times,
with and without the alpha/Li transfer.  Usage on the GPU box:  python tests/gpu_wall.py [--rows N]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scfgp_amd import synth                                           # noqa: E402
from scfgp_amd._lib import dptr                                       # noqa: E402
from scfgp_amd.engine import HipEngine                                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=1000000)
    ap.add_argument('--dtype', default='f32')
    a = ap.parse_args()
    N, D, S, M = a.rows, 64, 32, 1024
    seed = 0x5CF600FF
    X = synth.make_X(seed, N, D)
    y = synth.normal(seed + 9, 0, N).reshape(-1, 1)
    params = synth.make_params(seed + 0x0202, D, S, M, abc=(-1.0, 0.0, -1.0))
    eng = HipEngine(D, S, M, dtype=a.dtype)
    eng.set_params(params); eng.set_data(X, y)
    for _ in range(2):
        eng.eval(want_grad=True)

    def timed(fn, reps=5):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
        return min(ts), float(np.median(ts))

    print('eval, profiling off        : min %.2f  median %.2f ms' % timed(lambda: eng.eval(want_grad=True)))
    eng.set_profiling(True)
    print('eval, profiling on         : min %.2f  median %.2f ms' % timed(lambda: eng.eval(want_grad=True)))
    print('   stage sum               : %.2f ms' % sum(ms for _, ms in eng.timings()))
    eng.set_profiling(False)
    cost = np.zeros(1); grad = np.empty(eng.P); alpha = np.empty(eng.K); Li = np.empty((eng.K, eng.K))

    def raw(al, li):
        rc = eng.lib.scfgp_eval(eng.ctx, None, None, 0, 1, dptr(cost), dptr(grad), dptr(al), dptr(li))
        assert rc == 0
    print('C call, reused buffers     : min %.2f  median %.2f ms' % timed(lambda: raw(alpha, Li)))
    print('C call, no alpha/Li        : min %.2f  median %.2f ms' % timed(lambda: raw(None, None)))
    print('set_params                 : min %.2f  median %.2f ms' % timed(lambda: eng.set_params(params)))
    eng.close()


if __name__ == '__main__':
    main()
