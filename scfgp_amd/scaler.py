"""
Column-wise pre/post-processing used by SCFGP.set_data / predict -- host-side
counterpart of SCFGP/Scaler.py.  It is O(N*D) work done once per set_data and is not
part of the accelerated path; it exists so the facade is usable end to end.

Five modes (SCFGP/Scaler.py:15-21): 'min-max', 'normal', 'inv-normal', and the two
'auto-*' modes which min-max scale, Box-Cox transform each column with an exponent
chosen to minimise squared sample skewness (SLSQP on lambda = softplus(t), t in
[-5,5]; SCFGP/Scaler.py:58-70) and then standardise ('auto-normal') or map through
the normal CDF ('auto-inv-normal').  Constant columns are dropped (:40-41).
"""
import numpy as np
from scipy.optimize import minimize
from scipy.stats import norm, skew

_AUTO = ("auto-normal", "auto-inv-normal")


def _boxcox(x, lm):
    return (np.sign(x) * np.abs(x) ** lm - 1) / lm


def _inv_boxcox(x, lm):
    return np.sign(x * lm + 1) * np.abs(x * lm + 1) ** (1. / lm)


class Scaler(object):

    algos = ["min-max", "normal", "inv-normal", "auto-normal", "auto-inv-normal"]

    def __init__(self, algo):
        assert algo.lower() in self.algos, "Invalid Scaling Algorithm!"
        self.algo = algo.lower()
        self.data = {"cols": None}

    # -- fit ------------------------------------------------------------------------------
    def _fit_boxcox(self, tX):
        lms = np.zeros(tX.shape[1])
        for d in range(tX.shape[1]):
            col = tX[:, d]
            if np.unique(col).shape[0] < 10:          # near-categorical column: identity exponent
                lms[d] = 1
                continue
            soft = lambda t: np.log(np.exp(t[0]) + 1)
            obj = lambda t: skew(_boxcox(col, soft(t)), bias=False) ** 2
            res = minimize(obj, [0.], method='SLSQP', bounds=[(-5, 5)],
                           options={'ftol': 1e-8, 'maxiter': 100, 'disp': False})
            lms[d] = soft(res['x'])
        return lms

    def fit(self, X):
        d = self.data
        constant = np.where(np.all(X == X[0, :], axis=0))[0]
        d["cols"] = list(set(range(X.shape[1])).difference(constant))
        tX = X[:, d["cols"]]
        if self.algo == "min-max" or self.algo in _AUTO:
            d["min"] = np.min(tX, axis=0)
            d["max"] = np.max(tX, axis=0)
        if self.algo in _AUTO:
            tX = (tX - d["min"]) / (d["max"] - d["min"])
            d["boxcox"] = self._fit_boxcox(tX)
            tX = _boxcox(tX, d["boxcox"][None, :])
        if self.algo != "min-max":
            d["mu"] = np.mean(tX, axis=0)
            d["std"] = np.std(tX, axis=0)

    # -- transforms -------------------------------------------------------------------------
    def forward_transform(self, X):
        d = self.data
        tX = X[:, d["cols"]]
        if self.algo == "min-max":
            return (tX - d["min"]) / (d["max"] - d["min"])
        if self.algo == "normal":
            return (tX - d["mu"]) / d["std"]
        if self.algo == "inv-normal":
            return norm.cdf((tX - d["mu"]) / d["std"])
        tX = _boxcox((tX - d["min"]) / (d["max"] - d["min"]), d["boxcox"][None, :])
        if self.algo == "auto-normal":
            return (tX - d["mu"]) / d["std"]
        return norm.cdf(tX, d["mu"], d["std"])

    def backward_transform(self, X):
        d = self.data
        assert len(d["cols"]) == X.shape[1], "Backward Transform Error"
        if self.algo == "min-max":
            return X * (d["max"] - d["min"]) + d["min"]
        if self.algo == "normal":
            return X * d["std"] + d["mu"]
        if self.algo == "inv-normal":
            # as written in SCFGP/Scaler.py:124-125 (not the algebraic inverse of forward)
            return (norm.ppf(X) - d["mu"]) / d["std"]
        tX = X * d["std"] + d["mu"] if self.algo == "auto-normal" else norm.ppf(X, d["mu"], d["std"])
        return _inv_boxcox(tX, d["boxcox"][None, :]) * (d["max"] - d["min"]) + d["min"]
