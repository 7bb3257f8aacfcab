"""Phase breakdown of workgroup 0 of the Cholesky steps (diagnostic _trace build): SCFGP_LIB_VARIANT=_trace python tools/chol_trace.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scfgp_amd.engine import HipEngine
from scfgp_amd import synth
N, D, S, M = 65536, 64, 32, 1024
seed = 0x5CF600FF
X = synth.make_X(seed, N, D); y = synth.normal(seed + 9, 0, N).reshape(-1, 1)
params = synth.make_params(seed + 0x0202, D, S, M, abc=(-1.0, 0.0, -1.0))
eng = HipEngine(D, S, M, dtype='f32'); eng.set_params(params); eng.set_data(X, y)
eng.eval(want_grad=True); eng.eval(want_grad=True)
t = eng.debug_read('chol_trace', (64, 16), dtype=np.uint64).astype(np.int64)
nb = (eng.K + 63) // 64                         # live steps (the padding blocks are skipped)
t = t[1:nb]                                      # steps p >= 1 (the first has no diagonal update)
names = ['diag update (2 MFMA products)', 'panel 0', 'rank-16 upd 0', 'panel 1', 'rank-16 upd 1', 'panel 2', 'rank-16 upd 2',
         'panel 3', 'rank-16 upd 3', 'inverse level 0', 'inverse doubling + store']
d = np.diff(t[:, :12], axis=1) * 10.0 / 1e3       # 100 MHz ticks -> us
sub = [('  loads of A[p][p-1], Inv(p-1) -> LDS', 0, 12), ('  product 1 + L_p to LDS', 12, 13), ('  load of A[p][p] (issue)', 13, 14), ('  product 2', 14, 15), ('  store D, clear', 15, 1)]
for n, a, b in sub: print('%-40s %6.2f us' % (n, (t[:, b] - t[:, a]).mean() * 10.0 / 1e3))
for k, n in enumerate(names):
    print('%-32s %6.2f us' % (n, d[:, k].mean()))
print('%-32s %6.2f us' % ('workgroup 0 total', (t[:, 11] - t[:, 0]).mean() * 10.0 / 1e3))
print('step-to-step (start to start)    %6.2f us' % (np.diff(t[:, 0]).mean() * 10.0 / 1e3))
eng.close()
