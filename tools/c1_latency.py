"""Small-problem latency (Boston shape, fp64): eager evaluation calls, on-device training iterations with and without
hipGraph replay.  Usage: python tools/c1_latency.py"""
import sys, time, json
import numpy as np
sys.path.insert(0, '.')
from scfgp_amd.engine import HipEngine
from scfgp_amd import synth

N, D, S, M = 506, 13, 8, 64
seed = 0x5CF60001
X = synth.make_X(seed, N, D)
y = synth.normal(seed + 9, 0, N).reshape(-1, 1)
params = synth.make_params(seed + 0x0202, D, S, M, abc=(-1.0, 0.0, -1.0))
out = {}
eng = HipEngine(D, S, M, dtype='f64'); eng.set_params(params); eng.set_data(X, y)
for _ in range(20): eng.eval(want_grad=True)
ts = []
for _ in range(200):
    t0 = time.perf_counter(); eng.eval(want_grad=True); ts.append(time.perf_counter() - t0)
out['eval_ms_median'] = 1e3 * float(np.median(ts))
for graph in (0, 1):
    eng.set_params(params)
    eng.set_option('use_graph', graph)
    eng.opt_init('adam', learning_rate=1e-3)
    eng.train(20, want_factors=False)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); eng.train(400, want_factors=False); ts.append((time.perf_counter() - t0) / 400)
    out['train_iter_ms_graph%d' % graph] = 1e3 * float(np.median(ts))
eng.close()
print(json.dumps(out))
