"""
Tuning helper (not a test): per-stage hipEvent timings of one NLML+grad evaluation for a
list of option sets.  Usage on the GPU box:
    python tests/gpu_tune.py --rows 1000000 --dtype f32 --opts "gram_nsplit=7" "gram_nsplit=54" ...
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scfgp_amd import synth                                           # noqa: E402
from scfgp_amd.engine import HipEngine                                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default=None, help='C1..C5 / H of bench.py (sets rows, D, S, M, dtype)')
    ap.add_argument('--rows', type=int, default=None)
    ap.add_argument('--D', type=int, default=None)
    ap.add_argument('--S', type=int, default=None)
    ap.add_argument('--M', type=int, default=None)
    ap.add_argument('--dtype', default=None)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--opts', nargs='*', default=[''])
    a = ap.parse_args()
    import bench
    base = bench.CONFIGS[a.config or 'H'][:5]                  # explicit flags override the config's shape
    a.rows, a.D, a.S, a.M, a.dtype = [v if v is not None else b for v, b in zip((a.rows, a.D, a.S, a.M, a.dtype), base)]
    N, D, S, M = a.rows, a.D, a.S, a.M
    K = 2 * (S + M)
    seed = 0x5CF600FF
    X = synth.make_X(seed, N, D)
    y = synth.normal(seed + 9, 0, N).reshape(-1, 1)
    params = synth.make_params(seed + 0x0202, D, S, M, abc=(-1.0, 0.0, -1.0))
    eng = HipEngine(D, S, M, dtype=a.dtype)
    eng.set_params(params); eng.set_data(X, y)
    try:
        eng.eval(want_grad=True)
    except Exception as e:
        print('warm-up eval failed:', type(e).__name__)
    eng.set_profiling(True)
    ref = None
    for spec in a.opts:
        for kv in [s for s in spec.split(',') if s]:
            k, v = kv.split('=')
            eng.set_option(k, int(v))
        acc = {}
        tot = []
        for _ in range(a.reps):
            try:
                cost, g, al, Li = eng.eval(want_grad=True)
            except Exception:
                cost, g = np.nan, np.zeros(eng.P)
            tm = eng.timings()
            tot.append(sum(ms for _, ms in tm))
            for i, (name, ms) in enumerate(tm):
                acc.setdefault('%02d_%s' % (i, name), []).append(ms)
        if ref is None:
            ref = (float(cost), g.copy())
        dg = np.linalg.norm(g - ref[1]) / max(np.linalg.norm(ref[1]), 1e-300)
        print('[%s] total %.2f ms  cost %.10f  dgrad %.1e' % (spec, min(tot), float(cost), dg))
        print('    ' + '  '.join('%s=%.2f' % (k[3:], min(v)) for k, v in sorted(acc.items())), flush=True)
    eng.close()


if __name__ == '__main__':
    main()
