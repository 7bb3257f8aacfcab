"""fp32 mode against fp64 mode at full size for several flush intervals of the Gram products (option gram_chunk):
python tools/chunk_parity.py [rows]"""
import sys, json
import numpy as np
sys.path.insert(0, '.')
import bench
from scfgp_amd.engine import HipEngine
from scfgp_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
D, S, M = 64, 32, 1024
X = synth.make_X(bench.SEED, N, D)
y = synth.normal(bench.SEED + 9, 0, N).reshape(-1, 1)
params = synth.make_params(bench.SEED + 0x0202, D, S, M, abc=(-1.0, 0.0, -1.0))
e64 = HipEngine(D, S, M, dtype='f64'); e64.set_params(params); e64.set_data(X, y)
ref = e64.eval(want_grad=True)
for chunk in (2048, 4096, 8192, 16384, 65536):
    e = HipEngine(D, S, M, dtype='f32'); e.set_params(params); e.set_option('gram_chunk', chunk); e.set_data(X, y)
    out = e.eval(want_grad=True)
    p = bench._parity(out, ref, e, e64, D, S, M, 'chunk %d' % chunk)
    print(json.dumps({k: (v if isinstance(v, str) else float('%.3g' % v)) for k, v in p.items()}))
    e.close()
