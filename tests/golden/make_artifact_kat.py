"""
Regenerates tests/golden/artifact_kat.npz -- the only reference-produced golden
vector for the hot path.

Runs ONLY in the build container (needs /root/reference and the Boston table
that ships with an old scikit-learn under /opt/conda); the GPU box and the test
suite consume the committed .npz.  The reference's pickle
experiments/boston_housing/boston_scfgp.pkl (written by SCFGP.save,
SCFGP/SCFGP.py:296-302) is read AS DATA through a restricted unpickler: no
reference code, Theano object or callable is ever instantiated, and nothing of
the pickle is copied into the repo except the numeric arrays listed below.

Fixture contents (float64):
  X, y      scaled 400-row training split (what train_func received)
  params    trained 1333-vector (a,b,c,l_f,r_f,l_p,p) found inside train_func
  Li, alpha Theano-computed outputs stored by the reference
  cost      evals['COST'][1][-1] recorded by optimize() (SCFGP/SCFGP.py:265-266)
  S, M, D
  Xraw, yraw           the raw rows of that split (scikit-learn's Boston table)
  xs_*, ys_*, *_algo   the fitted X / y scaler dictionaries stored in the pickle (SCFGP/Scaler.py:23-24,39-97)
Predict side (the only reference-held numbers for pred_func):
  val_rows             the 106 rows of the table that are not in the training split (what optimize() received as Xv, yv)
  Xv_raw, yv_raw       their raw features / targets (scikit-learn's table)
  val_metrics          evals[MAE, NMAE, MSE, NMSE, MNLP, SCORE][1][-1]: what SCFGP.predict (SCFGP/SCFGP.py:285-293) appended in
                       the last call of optimize() (:268-269), i.e. with the stored alpha / Li and the trained vector
  val_changed          position inside val_rows of table row 445, whose FEATURES differ between the 2016 copy of the table the
                       reference ran on and today's (its error alone closes both the MAE and the MSE gap, checked below)
"""
import collections
import os
import pickle
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
from oracle import scfgp_oracle as O   # noqa: E402

PKL = '/root/reference/experiments/boston_housing/boston_scfgp.pkl'
CSV = '/opt/conda/lib/python3.9/site-packages/sklearn/datasets/data/boston_house_prices.csv'


class _Stub(object):
    _args = (); _state = None

    def __init__(self, *a, **k):
        self._args = a

    def __setstate__(self, s):
        self._state = s


_ALLOWED = {('numpy.core.multiarray', '_reconstruct'), ('numpy', 'ndarray'), ('numpy', 'dtype'),
            ('numpy.core.multiarray', 'scalar'), ('collections', 'OrderedDict'),
            ('builtins', 'slice'), ('__builtin__', 'slice')}


class _DataOnlyUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _ALLOWED:
            return super().find_class(module.replace('numpy.core', 'numpy._core'), name)
        return type(name, (_Stub,), {})


def _walk_arrays(obj, out, seen):
    if id(obj) in seen:
        return
    seen.add(id(obj))
    if isinstance(obj, np.ndarray):
        out.append(obj)
    elif isinstance(obj, dict):
        for v in obj.values():
            _walk_arrays(v, out, seen)
    elif isinstance(obj, (list, tuple, set)):
        for v in obj:
            _walk_arrays(v, out, seen)
    elif isinstance(obj, _Stub):
        _walk_arrays(getattr(obj, '_args', None), out, seen)
        _walk_arrays(getattr(obj, '_state', None), out, seen)
        _walk_arrays(obj.__dict__, out, seen)


def _artifact_scaler(state):
    """The product's own host Scaler (scfgp_amd/scaler.py) carrying the FITTED dictionary of the reference's pickled Scaler
    (SCFGP/Scaler.py:23-24: `algo`, `data` with cols/min/max/boxcox/mu/std): what reproduces Theano's Li / alpha / COST below is
    therefore scaler.py's forward_transform itself, not a private restatement."""
    from scfgp_amd.scaler import Scaler
    sc = Scaler(str(state['algo']))
    sc.data = {k: ([int(c) for c in v] if k == 'cols' else np.asarray(v, np.float64)) for k, v in state['data'].items()}
    return sc


def main():
    with open(PKL, 'rb') as f:
        d = _DataOnlyUnpickler(f).load()
    M = int(d['M']); Li = np.asarray(d['Li'], np.float64); alpha = np.asarray(d['alpha'], np.float64)
    K = Li.shape[0]; J = K // 2; S = J - M
    xs = d['X_scaler'].__dict__ if d['X_scaler']._state is None else d['X_scaler']._state
    ys = d['y_scaler'].__dict__ if d['y_scaler']._state is None else d['y_scaler']._state
    D = len(xs['data']['cols'])
    P = O.num_params(D, S, M)
    arrs = []
    _walk_arrays(d['train_func'], arrs, set())
    cands = [a for a in arrs if a.dtype == np.float64 and a.size == P]
    assert cands, 'trained parameter vector not found in train_func'
    cost_rec = float(d['evals']['COST'][1][-1])

    raw = np.loadtxt(CSV, delimiter=',', skiprows=2)
    Xraw, yraw = raw[:, :13], raw[:, 13:14]
    x_scaler, y_scaler = _artifact_scaler(xs), _artifact_scaler(ys)
    assert x_scaler.algo == 'auto-inv-normal' and y_scaler.algo == 'auto-normal'
    Xs = x_scaler.forward_transform(Xraw)
    ysc = y_scaler.forward_transform(yraw)

    best = None
    for params in cands:
        params = np.asarray(params, np.float64).ravel()
        a = params[0]
        Phi = O.feature_map(Xs, params, D, S, M)                       # (506,K)
        # recover the 0/1 row-membership vector z from  sum_i z_i phi_i phi_i^T = L L^T - lam I
        L = np.linalg.inv(Li)
        A = L @ L.T - (np.exp(2 * a) + O.EPSILON) * np.eye(K)
        iu = np.tril_indices(K)
        Mat = np.stack([np.outer(Phi[i], Phi[i])[iu] for i in range(Phi.shape[0])], 1)
        z, *_ = np.linalg.lstsq(Mat, A[iu], rcond=None)
        dev = np.abs(z - np.round(z)).max()
        if best is None or dev < best[0]:
            best = (dev, params, np.round(z).astype(int))
    dev, params, z = best
    assert dev < 1e-6 and set(np.unique(z)) <= {0, 1}, dev
    tr = np.where(z == 1)[0]
    X, y = Xs[tr], ysc[tr]
    cost, al, Li_o = O.forward(X, y, params, S, M, True)
    rel = lambda u, v: np.linalg.norm(u - v) / np.linalg.norm(v)
    print('rows', len(tr), 'split deviation', dev)
    print('oracle vs artifact: Li %.2e alpha %.2e cost %.2e' % (
        rel(Li_o, Li), rel(al, alpha), abs(cost - cost_rec) / abs(cost_rec)))
    # the fitted scaler dictionaries (numbers from the pickle) and the raw rows of the split (scikit-learn's Boston table, not
    # part of the reference): tests/test_oracle_golden.py drives scfgp_amd/scaler.py with them and must land on X, y above
    # ---- predict side: the 106 complement rows through SCFGP.predict's arithmetic (SCFGP/SCFGP.py:278-293) ----
    va = np.setdiff1d(np.arange(Xraw.shape[0]), tr)
    names = ('MAE', 'NMAE', 'MSE', 'NMSE', 'MNLP', 'SCORE')
    rec = np.array([float(d['evals'][k][1][-1]) for k in names])
    Xv, yv = Xraw[va], yraw[va]
    mu_f, std_f = O.predict(x_scaler.forward_transform(Xv), alpha, Li, params, S, M)
    mu_y = y_scaler.backward_transform(mu_f)
    e = (mu_y - yv).ravel()
    assert abs(np.var(yv) - rec[2] / rec[3]) < 1e-13 * np.var(yv), 'validation rows are not the complement'
    # which single row's error closes sum|e| and sum e^2 at once?  (two equations, one unknown)
    closure = []
    for i in range(len(va)):
        keep = np.arange(len(va)) != i
        ep = len(va) * rec[0] - np.abs(e[keep]).sum()
        closure.append(abs(len(va) * rec[2] - (e[keep] ** 2).sum() - ep * ep))
    changed = int(np.argmin(closure))
    print('validation: %d rows, var(yv) == MSE/NMSE, changed row = table row %d (closure %.2e; next best %.2e)' % (
        len(va), va[changed], closure[changed], np.partition(closure, 1)[1]))
    assert closure[changed] < 1e-10 and va[changed] == 445
    sc = {}
    for tag, sd in (('xs', x_scaler), ('ys', y_scaler)):
        for k, v in sd.data.items():
            sc['%s_%s' % (tag, k)] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, 'artifact_kat.npz'), X=X, y=y, params=params,
                        Li=Li, alpha=alpha, cost=cost_rec, S=S, M=M, D=D, train_rows=tr,
                        Xraw=Xraw[tr], yraw=yraw[tr], xs_algo=x_scaler.algo, ys_algo=y_scaler.algo,
                        val_rows=va, Xv_raw=Xv, yv_raw=yv, val_metrics=rec, val_changed=changed, **sc)


if __name__ == '__main__':
    main()
