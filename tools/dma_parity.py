"""fp32 mode against fp64 mode with and without the LDS-DMA apply tiles (option apply_dma): python tools/dma_parity.py [C3|H|C5]"""
import sys, json
import numpy as np
sys.path.insert(0, '.')
import bench
from scfgp_amd.engine import HipEngine
from scfgp_amd import synth

cfg = sys.argv[1] if len(sys.argv) > 1 else 'C3'
N, D, S, M = bench.CONFIGS[cfg][:4]
X = synth.make_X(bench.SEED, N, D)
y = synth.normal(bench.SEED + 9, 0, N).reshape(-1, 1)
params = synth.make_params(bench.SEED + 0x0202, D, S, M, abc=(-1.0, 0.0, -1.0))
e64 = HipEngine(D, S, M, dtype='f64'); e64.set_params(params); e64.set_data(X, y)
ref = e64.eval(want_grad=True)
for dma in (0, 1, 2, -1):
    e = HipEngine(D, S, M, dtype='f32'); e.set_params(params); e.set_option('apply_dma', dma); e.set_data(X, y)
    out = e.eval(want_grad=True)
    p = bench._parity(out, ref, e, e64, D, S, M, 'apply_dma %d' % dma)
    p['grad_all'] = bench.rel(out[1], ref[1])
    print(json.dumps({k: (v if isinstance(v, str) else float('%.3g' % v)) for k, v in p.items()}))
    e.close()
