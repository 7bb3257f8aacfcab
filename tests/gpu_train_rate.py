"""Diagnostic (not a test): training iterations per second at the Boston shape (C1), host update rule vs the
captured device iteration.  Usage on the GPU box:  python tests/gpu_train_rate.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scfgp_amd import synth
from scfgp_amd.funcs import CompiledFuncs

N, D, S, M = 506, 13, 8, 64
seed = 0x5CF60001
X = synth.make_X(seed, N, D); y = synth.normal(seed + 9, 0, N).reshape(-1, 1)
params = synth.make_params(seed + 0x0202, D, S, M, abc=(-1.0, 0.0, -1.0))
kw = {'learning_rate': 0.01, 'beta1': 0.9, 'beta2': 0.999, 'epsilon': 1e-8}
host = CompiledFuncs(D, S, M, params.copy(), 'adam', kw)
for _ in range(20): host.train_iter_func(X, y)
t0 = time.perf_counter(); n = 300
for _ in range(n): host.train_iter_func(X, y)
print('host update rule     : %.0f iterations/s' % (n / (time.perf_counter() - t0)))
dev = CompiledFuncs(D, S, M, params.copy(), 'adam', kw, device_optimizer=True)
dev.train_iters(X, y, 20)
t0 = time.perf_counter(); n = 2000
dev.train_iters(X, y, n)
print('device rule, 1 graph : %.0f iterations/s' % (n / (time.perf_counter() - t0)))
