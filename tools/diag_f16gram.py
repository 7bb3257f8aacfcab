import sys, numpy as np
sys.path.insert(0, '/root/repo')
from scfgp_amd import synth
from scfgp_amd.engine import HipEngine
N, D, S, M = 20000, 16, 16, 256
seed = 0x5CF60A00 + M
X = synth.make_X(seed, N, D); y = synth.normal(seed + 1, 0, N).reshape(-1, 1)
params = synth.make_params(seed + 2, D, S, M, abc=(-1.0, 0.0, -1.0))
K = 2 * (S + M)
def run(dtype, **opt):
    e = HipEngine(D, S, M, dtype)
    if dtype != 'f64':
        e.set_option('gram64', 0); e.set_option('apply_dma', 2)
    for k, v in opt.items(): e.set_option(k, v)
    e.set_params(params); e.set_data(X, y)
    c, g, a, L = e.eval()
    Kp = e.Kp if hasattr(e, 'Kp') else None
    G = e.debug_read('G', (640, 640))
    e.close()
    return G[:K, :K], a
G64, a64 = run('f64')
d = np.sqrt(np.diag(G64))
def rep(name, G, a):
    E = (G - G64) / np.outer(d, d)
    print('%-28s G err rms %.2e max %.2e  mean signed diag %.2e  mean signed all %.2e | alpha %.2e' % (
        name, np.sqrt((E**2).mean()), np.abs(E).max(), np.diag(E).mean(), (E*np.sign(G64)).mean(), np.linalg.norm(a - a64) / np.linalg.norm(a64)))
rep('f32', *run('f32'))
for ch in (2048, 4096, 8192, 20480):
    rep('f16x3 chunk %d' % ch, *run('f16x3', gram_chunk=ch))
rep('f16x3 f16_gram=0', *run('f16x3', f16_gram=0))
