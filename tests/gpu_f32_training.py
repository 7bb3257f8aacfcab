"""Diagnostic (not a test): does training in fp32 mode follow the fp64 trajectory?  (teacher data, adam + Nesterov)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scfgp_amd import synth
from scfgp_amd.engine import HipEngine
from scfgp_amd.funcs import CompiledFuncs

N, D, S, M = 100000, 32, 16, 256
seed = 0x5CF60222
K = 2 * (S + M)
X = synth.make_X(seed, N, D)
eng = HipEngine(D, S, M)
eng.set_params(synth.make_params(seed + 0x0101, D, S, M, abc=(-1.0, 0.0, -1.0)))
f, _ = eng.predict(X, synth.teacher_weights(seed + 0x0303, K), np.eye(K))
y = synth.finish_targets(seed + 0x0404, f).reshape(-1, 1); eng.close()
p0 = synth.make_params(seed + 0x0202, D, S, M, abc=(-1.0, 0.0, -1.0))
kw = {'learning_rate': 0.01, 'beta1': 0.9, 'beta2': 0.999, 'epsilon': 1e-8}
hist = {}
for dt in ('f64', 'f32'):
    cf = CompiledFuncs(D, S, M, p0.copy(), 'adam', kw, dtype=dt, device_optimizer=True)
    hist[dt], _, _ = cf.train_iters(X, y, 60)
for i in (0, 1, 5, 10, 20, 40, 59):
    print('iter %2d  f64 %.9f  f32 %.9f  rel diff %.1e' % (i, hist['f64'][i], hist['f32'][i], abs(hist['f32'][i] - hist['f64'][i]) / abs(hist['f64'][i])))
