"""
Round-5 GPU tests.

Row h: the predictive mean / band of the HIP path on the artifact's 106 validation rows against the MAE / NMAE /
MSE / NMSE / MNLP / SCORE the REFERENCE recorded for them (tests/artifact_predict.py states the relations; the CPU tier runs
the same checker over the oracle) -- through pred_func, through scfgp_predict_y and through the facade.
"""
import numpy as np
import pytest

from tests import artifact_predict as AP

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


@pytest.mark.parametrize('dtype,tol', [('f64', 1e-10), ('f32', 2e-4)])
def test_pred_func_closes_the_artifacts_recorded_validation_metrics(dtype, tol):
    """pred_func (SCFGP/SCFGP.py:138-148) + the host tail of SCFGP.predict (:279-284)."""
    from scfgp_amd.funcs import CompiledFuncs
    z = AP.load()
    cf = CompiledFuncs(int(z['D']), int(z['S']), int(z['M']), z['params'].copy(), dtype=dtype)
    mu_y, std_y = AP.host_tail(cf.triple()[2], z)
    if dtype == 'f64':
        got = AP.check(mu_y, std_y, z, tol)
        print('pred_func vs the reference record:', got)
        assert abs(abs(got['e_changed_2016']) - 3.4613474221682736) < 1e-9
    else:
        # fp32 features: the means carry ~1e-6; the closure is then ~1e-5 and the MNLP bound (slack 1e-3) cannot be asserted
        rec = z['val_metrics']; ch = int(z['val_changed']); keep = np.arange(106) != ch
        e = (mu_y - z['yv_raw']).ravel()
        e_new = 106 * rec[0] - np.abs(e[keep]).sum()
        assert abs(106 * rec[2] - (e[keep] ** 2).sum() - e_new ** 2) < tol
    cf.engine.close()


def test_predict_y_closes_the_artifacts_recorded_validation_metrics():
    """scfgp_predict_y: X scaler, pred_func, y-scaler backward transform, band and the six metrics on the device."""
    from scfgp_amd.funcs import CompiledFuncs
    z = AP.load()
    xs, ys = AP.scalers(z)
    cf = CompiledFuncs(int(z['D']), int(z['S']), int(z['M']), z['params'].copy())
    mu_y, std_y, met = cf.pred_y(z['Xv_raw'], xs, ys, z['alpha'], z['Li'], z['yv_raw'])
    got = AP.check(mu_y, std_y, z)
    # the device's six numbers are the reference's formulas over the device's own rows ...
    m = AP.metrics(mu_y, std_y, z['yv_raw'])
    assert np.allclose([met[k] for k in AP.NAMES], m, rtol=1e-12, atol=0)
    # ... and differ from the reference's record by the changed row's share alone
    rec = dict(zip(AP.NAMES, z['val_metrics'].tolist()))
    assert abs(106 * (rec['MAE'] - met['MAE']) - (abs(got['e_changed_2016']) - abs(got['e_changed_today']))) < 1e-10
    assert abs(106 * (rec['MSE'] - met['MSE']) - (got['e_changed_2016'] ** 2 - got['e_changed_today'] ** 2)) < 1e-10
    assert abs(met['MSE'] / met['NMSE'] - rec['MSE'] / rec['NMSE']) < 1e-12 * np.var(z['yv_raw'])
    cf.engine.close()


@pytest.mark.parametrize('device_scaler', [False, True])
def test_facade_predict_on_the_artifacts_validation_rows(device_scaler):
    """SCFGP.predict(Xv, yv) of a model carrying the artifact's state: evals[...] as the reference would append them."""
    from scfgp_amd import SCFGP
    from scfgp_amd.funcs import Shared
    z = AP.load()
    model = SCFGP(sparsity=int(z['S']), nfeats=int(z['M']), device_scaler=device_scaler)
    model.X_scaler, model.y_scaler = AP.scalers(z)
    model.D = int(z['D']); model.params = Shared(z['params'].copy())
    model.build_hip_models('adam', {'learning_rate': 0.01})
    model.alpha, model.Li = z['alpha'], z['Li']
    mu_y, std_y = model.predict(z['Xv_raw'], z['yv_raw'])
    got = AP.check(mu_y, std_y, z)
    m = AP.metrics(mu_y, std_y, z['yv_raw'])
    assert np.allclose([model.evals[k][1][-1] for k in AP.NAMES], m, rtol=1e-12, atol=0)
    assert got['closure'] < 1e-10
