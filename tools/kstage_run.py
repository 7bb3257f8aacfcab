"""Three evaluations at 65536 rows of a BASELINE shape (the K x K stage does not depend on N): the workload tools/kstage_launches.sh traces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from scfgp_amd import synth
from scfgp_amd.engine import HipEngine
cfg = sys.argv[1] if len(sys.argv) > 1 else 'H'
_, D, S, M, dtype = bench.CONFIGS[cfg][:5]
N = 65536
X = synth.make_X(bench.SEED, N, D); y = synth.normal(bench.SEED + 9, 0, N).reshape(-1, 1)
params = synth.make_params(bench.SEED + 0x0202, D, S, M, abc=(-1.0, 0.0, -1.0))
eng = HipEngine(D, S, M, dtype=dtype); eng.set_params(params); eng.set_data(X, y)
for _ in range(3):
    eng.eval(want_grad=True)
eng.set_profiling(True); eng.eval(want_grad=True)
print({k: round(v, 3) for k, v in eng.timings() if k.startswith('kstage')})
eng.close()
