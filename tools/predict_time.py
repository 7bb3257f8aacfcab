"""Time scfgp_predict at the headline shape (D=64, S=32, M=1024) on T test rows, per dtype; prints rows/s.
Usage: python tools/predict_time.py [T]"""
import sys, time, json
import numpy as np
sys.path.insert(0, '.')
from scfgp_amd.engine import HipEngine, num_params

T = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
D, S, M = 64, 32, 1024
K = 2 * (S + M)
rng = np.random.default_rng(7)
Xs = rng.standard_normal((T, D))
params = 0.1 * rng.standard_normal(num_params(D, S, M))
alpha = rng.standard_normal(K) / np.sqrt(K)
Li = np.tril(rng.standard_normal((K, K))) / np.sqrt(K)
out = {}
ref = None
for dt in ('f64', 'f32'):
    eng = HipEngine(D, S, M, dtype=dt)
    eng.set_params(params)
    eng.predict(Xs[:4096], alpha, Li)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); mu, sd = eng.predict(Xs, alpha, Li); ts.append(time.perf_counter() - t0)
    if ref is None: ref = (mu, sd)
    out[dt] = {'sec': min(ts), 'rows_per_s': T / min(ts),
               'mu_rel': float(np.abs(mu - ref[0]).max() / np.abs(ref[0]).max()),
               'sd_rel': float(np.abs(sd - ref[1]).max() / np.abs(ref[1]).max())}
    eng.close()
print(json.dumps({'T': T, 'K': K, 'predict': out}))
