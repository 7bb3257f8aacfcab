"""
CPU oracle for the SCFGP Fourier-feature marginal-likelihood hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under scfgp_amd/ may import this module; it
is the checker for tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never the thing measured or shipped.

Parity status: PINNED for the forward path, self-consistent for the rest.  `forward()` reproduces the
Theano-computed Li / alpha / COST stored in the reference's own artifact
experiments/boston_housing/boston_scfgp.pkl (see tests/golden/make_artifact_kat.py and
tests/test_oracle_golden.py): that pins layout, feature map, Gram, Cholesky, Li, alpha, the
Gauss-Hermite term, penalty and cost.  The gradient, the predictive sigma and the update rules are
stored nowhere in the reference, so NO reference vector pins them: they rest on three restatements
of this repository agreeing with each other -- `value_and_grad()` (the hand-derived 3-sweep algorithm
the HIP kernels implement), torch float64 autograd of the literal reference graph
(oracle/autograd_ref.py) and central finite differences.

Every function cites the reference lines it restates (paths relative to the
reference checkout, file SCFGP/SCFGP.py unless stated otherwise).
"""
import numpy as np

EPSILON = 1e-6          # jitter, SCFGP.py:93,105
GH_POINTS = 30          # SCFGP.py:118


# --------------------------------------------------------------------------
# parameter vector  (SCFGP.py:64-72 layout, :74-90 unpack)
# --------------------------------------------------------------------------
def num_params(D, S, M):
    """Length of the flat hyper-parameter vector, SCFGP.py:72."""
    return 3 + D * S + M * S + S + M


def init_params(D, S, M, rng):
    """Same distributions as SCFGP.init_params (SCFGP.py:64-72)."""
    a = rng.standard_normal(1)
    b = rng.standard_normal(1)
    c = rng.standard_normal(1)
    l_f = rng.standard_normal(D * S)
    r_f = rng.random(M * S)
    l_p = 2 * np.pi * rng.random(S)
    p = 2 * np.pi * rng.random(M)
    return np.concatenate([a, b, c, l_f, r_f, l_p, p])


def unpack_params(params, D, S, M):
    """SCFGP.unpack_params, SCFGP.py:74-90 (row-major reshapes, axis-0 means)."""
    params = np.asarray(params, dtype=np.float64)
    assert params.shape == (num_params(D, S, M),)
    a, b, c = params[0], params[1], params[2]
    t = 3
    l_F = params[t:t + D * S].reshape(D, S); t += D * S
    r_F = params[t:t + M * S].reshape(M, S); t += M * S
    F = l_F @ r_F.T
    l_P = params[t:t + S].reshape(1, S); t += S
    P = params[t:t + M].reshape(1, M)
    l_FC = l_P - l_F.mean(0)[None, :]
    FC = P - F.mean(0)[None, :]
    return a, b, c, l_F, r_F, F, l_FC, FC


def feature_map(X, params, D, S, M):
    """Phi = sig_f*sqrt(2/M)*[cos FF, sin FF], SCFGP.py:98-102 (predict twin :139-142)."""
    a, b, c, l_F, r_F, F, l_FC, FC = unpack_params(params, D, S, M)
    FF = np.concatenate((X @ l_F + l_FC, X @ F + FC), 1)
    return np.exp(b) * np.sqrt(2.0 / M) * np.concatenate((np.cos(FF), np.sin(FF)), 1)


def _penalty(l_F, F, S, M):
    """SCFGP.py:94,114-117,127 (mean/std along axis=1, ddof=0)."""
    kl = lambda mu, sig: sig + mu ** 2 - np.log(sig)
    mu_l = l_F.mean(1).sum(); sig_l = l_F.std(1).sum()
    mu_w = F.mean(1).sum(); sig_w = F.std(1).sum()
    return (kl(mu_w, sig_w) * M + kl(mu_l, sig_l) * S) / (S + M)


# --------------------------------------------------------------------------
# forward: literal restatement of the train graph  (SCFGP.py:98-128)
# --------------------------------------------------------------------------
def forward(X, y, params, S, M, gauss_hermite=True):
    """Returns (cost, alpha (K,1), Li (K,K)) exactly as train_func does
    (SCFGP.py:132-135).  With gauss_hermite=True the expected-NLL term is the
    literal 30-point quadrature of SCFGP.py:118-124; with False its closed form."""
    X = np.asarray(X, np.float64); y = np.asarray(y, np.float64).reshape(-1, 1)
    N, D = X.shape
    a, b, c, l_F, r_F, F, l_FC, FC = unpack_params(params, D, S, M)
    sig2_n = np.exp(2 * a)
    Phi = feature_map(X, params, D, S, M)
    noise = np.log(1 + np.exp(c))                                   # :103
    A = Phi.T @ Phi + (sig2_n + EPSILON) * np.eye(Phi.shape[1])      # :104-105
    L = np.linalg.cholesky(A)                                        # :106 (lower)
    Li = np.linalg.inv(L)                                            # :107 general inverse
    PhiTy = Phi.T @ y                                                # :108
    beta = Li @ PhiTy                                                # :109
    alpha = Li.T @ beta                                              # :110
    mu_f = Phi @ alpha                                               # :111
    var_f = ((Phi @ Li.T) ** 2).sum(1)[:, None]                      # :112
    dsp = noise * (var_f + 1)                                        # :113
    if gauss_hermite:
        hx, hw = np.polynomial.hermite.hermgauss(GH_POINTS)          # :118
        hw = hw / np.sqrt(np.pi)
        enll_sum = 0.0
        for x_i, w_i in zip(hx, hw):                                 # :121-124 node by node
            f = np.sqrt(2 * var_f) * x_i + mu_f
            nlk = (0.5 * f ** 2 - y * f) / dsp + 0.5 * (np.log(2 * np.pi * dsp) + y ** 2 / dsp)
            enll_sum += w_i * nlk.sum()
    else:
        enll_sum = 0.5 * (((mu_f - y) ** 2 + var_f) / dsp + np.log(2 * np.pi * dsp)).sum()
    nlml = (2 * np.log(np.diagonal(L)).sum() + 2 * enll_sum
            + 1. / sig2_n * ((y ** 2).sum() - (beta ** 2).sum()) + 2 * (N - M) * a)   # :125-126
    cost = (nlml + _penalty(l_F, F, S, M)) / N                        # :127-128
    return float(cost), alpha, Li


def predict(Xs, alpha, Li, params, S, M):
    """pred_func, SCFGP.py:138-148: returns (mu (T,1), std (T,))."""
    Xs = np.asarray(Xs, np.float64)
    D = Xs.shape[1]
    c = params[2]
    Phis = feature_map(Xs, params, D, S, M)
    noise = np.log(1 + np.exp(c))
    mu = Phis @ np.asarray(alpha).reshape(-1, 1)
    std = (noise * (1 + ((Phis @ np.asarray(Li).T) ** 2).sum(1))) ** 0.5
    return mu, std


# --------------------------------------------------------------------------
# value + gradient: the 3-sweep algorithm (replaces TT.grad, SCFGP.py:129)
# --------------------------------------------------------------------------
def value_and_grad(X, y, params, S, M, chunk=8192, n_global=None):
    """Hand-derived exact gradient of `forward(...)[0]` w.r.t. the flat vector.

    Three row sweeps (chunked so N x K temporaries stay small), each ending in a
    reduction that is additive over rows -- the same structure the HIP path uses
    and the reason the data-parallel version needs three all-reduces.
    Returns (cost, grad (P,), alpha (K,1), Li (K,K)).
    """
    X = np.asarray(X, np.float64); y = np.asarray(y, np.float64).reshape(-1)
    N, D = X.shape
    Ng = N if n_global is None else n_global
    a, b, c, l_F, r_F, F, l_FC, FC = unpack_params(params, D, S, M)
    J = S + M; K = 2 * J
    s = np.exp(b) * np.sqrt(2.0 / M)
    lam = np.exp(2 * a) + EPSILON
    kappa = np.log1p(np.exp(c))
    W_all = np.concatenate((l_F, F), 1)              # (D,J)
    c_all = np.concatenate((l_FC, FC), 1)            # (1,J)

    def phi_rows(lo, hi):
        Z = X[lo:hi] @ W_all + c_all
        return s * np.concatenate((np.cos(Z), np.sin(Z)), 1)

    # ---- sweep 1: G = Phi^T Phi, g = Phi^T y, yy
    G = np.zeros((K, K)); g = np.zeros(K); yy = 0.0
    for lo in range(0, N, chunk):
        hi = min(N, lo + chunk); Ph = phi_rows(lo, hi)
        G += Ph.T @ Ph; g += Ph.T @ y[lo:hi]; yy += (y[lo:hi] ** 2).sum()
    st1 = dict(G=G, g=g, yy=yy)
    return _finish_from_sweep1(X, y, params, S, M, st1, chunk, Ng, phi_rows)


def _finish_from_sweep1(X, y, params, S, M, st1, chunk, Ng, phi_rows):
    N, D = X.shape
    a, b, c, l_F, r_F, F, l_FC, FC = unpack_params(params, D, S, M)
    J = S + M; K = 2 * J
    lam = np.exp(2 * a) + EPSILON
    kappa = np.log1p(np.exp(c))
    G, g, yy = st1['G'], st1['g'], st1['yy']
    # ---- K-stage 1
    A = G + lam * np.eye(K)
    L = np.linalg.cholesky(A)
    Li = np.linalg.solve(L, np.eye(K))
    B = Li.T @ Li
    beta = Li @ g
    alpha = Li.T @ beta
    T1 = 2 * np.log(np.diagonal(L)).sum()
    # ---- sweep 2
    T2 = 0.0; kbar = 0.0; h = np.zeros(K); W = np.zeros((K, K))
    p_all = np.empty(N); q_all = np.empty(N)
    for lo in range(0, N, chunk):
        hi = min(N, lo + chunk); Ph = phi_rows(lo, hi)
        mu = Ph @ alpha
        v = ((Ph @ B) * Ph).sum(1)
        d = kappa * (v + 1)
        r = mu - y[lo:hi]
        T2 += ((r * r + v) / d + np.log(2 * np.pi * d)).sum()
        e = 1 / d - (r * r + v) / d ** 2
        q = 1 / d + kappa * e
        p = 2 * r / d
        kbar += (e * (v + 1)).sum()
        h += Ph.T @ p
        W += Ph.T @ (q[:, None] * Ph)
        p_all[lo:hi] = p; q_all[lo:hi] = q
    # ---- K-stage 2
    em2a = np.exp(-2 * a)
    u = B @ h
    Abar = B - B @ W @ B - 0.5 * (np.outer(u, alpha) + np.outer(alpha, u)) + em2a * np.outer(alpha, alpha)
    T3 = em2a * (yy - g @ alpha)
    T4 = 2 * (Ng - M) * a
    abar = 2 * np.exp(2 * a) * np.trace(Abar) - 2 * T3 + 2 * (Ng - M)
    cbar = kbar / (1 + np.exp(-c))
    ut = u - 2 * em2a * alpha
    # ---- sweep 3
    bbar = 0.0
    XZ = np.zeros((D, J)); colsum = np.zeros(J)
    for lo in range(0, N, chunk):
        hi = min(N, lo + chunk); Ph = phi_rows(lo, hi)
        p = p_all[lo:hi]; q = q_all[lo:hi]
        Pb = (np.outer(p, alpha) + np.outer(y[lo:hi], ut)
              + 2 * q[:, None] * (Ph @ B) + 2 * (Ph @ Abar))
        bbar += (Pb * Ph).sum()
        Zb = Ph[:, :J] * Pb[:, J:] - Ph[:, J:] * Pb[:, :J]
        XZ += X[lo:hi].T @ Zb; colsum += Zb.sum(0)
    st3 = dict(XZ=XZ, colsum=colsum, bbar=bbar)
    pen = _penalty(l_F, F, S, M)
    cost = (T1 + T2 + T3 + T4 + pen) / Ng
    grad = _epilogue(params, D, S, M, st3, abar, cbar, Ng)
    return float(cost), grad, alpha.reshape(-1, 1), Li


def _epilogue(params, D, S, M, st3, abar, cbar, Ng):
    """Chain rule from (X^T Zbar, colsum Zbar) to the flat vector + penalty gradient."""
    a, b, c, l_F, r_F, F, l_FC, FC = unpack_params(params, D, S, M)
    XZ, colsum, bbar = st3['XZ'], st3['colsum'], st3['bbar']
    lF_bar = XZ[:, :S].copy(); F_bar = XZ[:, S:].copy()
    lFC_bar = colsum[:S]; FC_bar = colsum[S:]
    P_bar = FC_bar.copy(); lP_bar = lFC_bar.copy()
    F_bar -= FC_bar[None, :] / D
    lF_bar -= lFC_bar[None, :] / D
    for T, w, n, Tb in ((F, M / (S + M), M, F_bar), (l_F, S / (S + M), S, lF_bar)):
        m_d = T.mean(1); s_d = T.std(1)
        mu = m_d.sum(); sg = s_d.sum()
        Tb += w * ((1 - 1 / sg) * (T - m_d[:, None]) / (n * s_d[:, None]) + 2 * mu / n)
    lF_bar += F_bar @ r_F
    rF_bar = F_bar.T @ l_F
    return np.concatenate(([abar, bbar, cbar], lF_bar.ravel(), rF_bar.ravel(), lP_bar, P_bar)) / Ng


def fd_grad(X, y, params, S, M, idx, h=1e-6):
    """Central finite differences of forward() on the listed coordinates."""
    out = np.empty(len(idx))
    for t, i in enumerate(idx):
        pp = params.copy(); pm = params.copy()
        pp[i] += h; pm[i] -= h
        out[t] = (forward(X, y, pp, S, M, False)[0] - forward(X, y, pm, S, M, False)[0]) / (2 * h)
    return out


# --------------------------------------------------------------------------
# staged engine: the same 3-sweep algorithm split at the three exchange points,
# used by tests of the row-sharded driver (gloo, world_size 2) -- tests only.
# --------------------------------------------------------------------------
class PeerFailed(RuntimeError):
    """another rank marked an exchange of this evaluation as failed (the library's SCFGP_EPEER)"""


XS_RAN1, XS_CAP1, XS_CAP2, XS_FAIL = 4, 5, 6, 7          # status slots of the 8 scalars closing every exchange buffer (common.h)


class OracleEngine(object):
    """Implements the staged interface scfgp_amd.sharded.ShardedEvaluator drives
    (pass1/factor/pass2/adjoint/pass3/finish + exchange buffers) on the CPU.  Every exchange buffer ends, like the
    library's (include/scfgp_hip.h, "ranks decide together"), in 8 scalars: [0..3] row sums, [4..7] the status word whose sum
    over ranks counts the ranks that ran pass 1 at a raised level / cannot reach level 1 / level 2 / failed.  The oracle
    computes in float64 throughout, so its `level` is bookkeeping only: `want_level` plays the condition estimate's demand,
    `deny_level` an allocation that fails on this rank, `fail_at` a sweep that fails -- enough to rehearse, over gloo, the
    protocol by which the ranks commit to one level and fail together."""

    def __init__(self, D, S, M):
        self.D, self.S, self.M = D, S, M
        self.J = S + M; self.K = 2 * self.J
        self.params = None
        self.level, self.want_level, self.deny_level, self.fail_at = 0, 0, 0, 0
        self.denied, self.unsettled, self.ran_level, self.stage = 0, False, 0, 0

    def cap(self):
        return self.denied - 1 if self.denied > 0 else 2

    def _tail(self, vals, fail=0.0):
        t = np.zeros(8); t[:len(vals)] = vals
        t[XS_CAP1] = float(self.cap() < 1); t[XS_CAP2] = float(self.cap() < 2); t[XS_FAIL] = fail
        return t

    def _maybe_fail(self, stage):
        if self.fail_at == stage:
            self.fail_at = 0
            raise RuntimeError('pass%d: injected failure' % stage)

    def fail_stage(self, stage, want_grad=True):
        """scfgp_fail_stage: this rank owes the sum of exchange `stage` but cannot compute it"""
        K, D, J = self.K, self.D, self.J
        n = {1: K * K + K, 2: (K * K + K) if want_grad else 0, 3: D * J + J}[stage]
        buf = np.concatenate((np.zeros(n), self._tail([], fail=1.0)))
        setattr(self, 'x%d' % stage, buf)
        self.stage = 0

    def set_params(self, p):
        self.params = np.asarray(p, np.float64).copy()

    def set_data(self, X, y, n_global=None):
        self.X = np.asarray(X, np.float64); self.y = np.asarray(y, np.float64).reshape(-1)
        self.Ng = len(self.y) if n_global is None else int(n_global)

    def _phi(self):
        return feature_map(self.X, self.params, self.D, self.S, self.M)

    def pass1(self):
        self._maybe_fail(1)
        Ph = self._phi(); self.Ph = Ph
        K = self.K
        self.ran_level = self.level
        t = self._tail([(self.y ** 2).sum()]); t[XS_RAN1] = float(self.level >= 1)
        self.x1 = np.concatenate(((Ph.T @ Ph).ravel(), Ph.T @ self.y, t))
        self.stage = 1

    def exchange(self, stage):
        return {1: self.x1, 2: getattr(self, 'x2', None), 3: getattr(self, 'x3', None)}[stage]

    def factor(self):
        """False: start again at pass1 (the library's SCFGP_REDO from scfgp_factor, settle_level)"""
        K = self.K; a = self.params[0]
        if self.unsettled:                                       # commit to the lowest level any rank can reach
            self.unsettled = False
            xs = self.x1[K * K + K:]
            agreed = 0 if xs[XS_CAP1] > 0.5 else (1 if xs[XS_CAP2] > 0.5 else 2)
            if agreed < self.cap():
                self.denied = agreed + 1
            self.level = min(self.level, agreed)
            if agreed < 1 and xs[XS_RAN1] > 0.5:
                self.stage = 0
                return False
            self.ran_level = self.level
        G = self.x1[:K * K].reshape(K, K); self.g = self.x1[K * K:K * K + K]; self.yy = self.x1[K * K + K]
        A = G + (np.exp(2 * a) + EPSILON) * np.eye(K)
        L = np.linalg.cholesky(A)
        self.Li = np.linalg.solve(L, np.eye(K)); self.B = self.Li.T @ self.Li
        self.alpha = self.Li.T @ (self.Li @ self.g)
        self.T1 = 2 * np.log(np.diagonal(L)).sum()
        return True

    def pass2(self, want_grad=True):
        self._maybe_fail(2)
        Ph = self.Ph; c = self.params[2]; kappa = np.log1p(np.exp(c)); y = self.y
        mu = Ph @ self.alpha; v = ((Ph @ self.B) * Ph).sum(1); d = kappa * (v + 1); r = mu - y
        T2 = ((r * r + v) / d + np.log(2 * np.pi * d)).sum()
        e = 1 / d - (r * r + v) / d ** 2
        self.q = 1 / d + kappa * e; self.p = 2 * r / d
        kbar = (e * (v + 1)).sum()
        K = self.K
        # second exchange: B W B = V^T diag(q) V and u = B Phi^T p = V^T p with V = Phi B (row sums, like W and Phi^T p)
        V = Ph @ self.B
        if want_grad:
            self.x2 = np.concatenate(((V.T @ (self.q[:, None] * V)).ravel(), V.T @ self.p, self._tail([T2, kbar])))
        else:
            self.x2 = self._tail([T2, kbar])                    # forward only: the 8 scalars (scfgp_exchange, stage 2)

    def adjoint(self):
        K = self.K; a = self.params[0]
        BWB = self.x2[:K * K].reshape(K, K); u = self.x2[K * K:K * K + K]
        em2a = np.exp(-2 * a); B = self.B; al = self.alpha
        self.Abar = B - BWB - 0.5 * (np.outer(u, al) + np.outer(al, u)) + em2a * np.outer(al, al)
        self.ut = u - 2 * em2a * al

    def pass3(self):
        self._maybe_fail(3)
        Ph = self.Ph; J = self.J
        Pb = (np.outer(self.p, self.alpha) + np.outer(self.y, self.ut)
              + 2 * self.q[:, None] * (Ph @ self.B) + 2 * (Ph @ self.Abar))
        Zb = Ph[:, :J] * Pb[:, J:] - Ph[:, J:] * Pb[:, :J]
        self.x3 = np.concatenate(((self.X.T @ Zb).ravel(), Zb.sum(0), self._tail([(Pb * Ph).sum()])))

    def finish(self, want_grad=True):
        D, S, M, J, K = self.D, self.S, self.M, self.J, self.K
        self.stage = 0
        fails = self.x1[-8 + XS_FAIL] + self.x2[-8 + XS_FAIL] + (self.x3[-8 + XS_FAIL] if want_grad else 0.0)
        if fails > 0.5:
            raise PeerFailed('a rank failed in this evaluation')
        self.T2, self.kbar = self.x2[-8], self.x2[-7]
        # update_level of the library: the (summed) condition estimate asks for `want_level`; every rank tries to raise its
        # level in this same evaluation; what each could allocate travels with the next exchange 1 (settle_level in factor)
        top = min(self.want_level, self.cap())
        if top > self.level:
            self.unsettled = True
            lvl = top
            while lvl > self.level and self.deny_level and lvl >= self.deny_level:
                self.denied = lvl; lvl -= 1
            self.level = lvl
        if top > self.ran_level:                                 # `top` is common to the ranks; what this one could allocate is not
            return None                                          # SCFGP_REDO
        a, b, c, l_F, r_F, F, l_FC, FC = unpack_params(self.params, D, S, M)
        em2a = np.exp(-2 * a); Ng = self.Ng
        T3 = em2a * (self.yy - self.g @ self.alpha); T4 = 2 * (Ng - M) * a
        cost = (self.T1 + self.T2 + T3 + T4 + _penalty(l_F, F, S, M)) / Ng
        grad = None
        if want_grad:
            abar = 2 * np.exp(2 * a) * np.trace(self.Abar) - 2 * T3 + 2 * (Ng - M)
            cbar = self.kbar / (1 + np.exp(-c))
            st3 = dict(XZ=self.x3[:D * J].reshape(D, J), colsum=self.x3[D * J:D * J + J], bbar=self.x3[D * J + J])
            grad = _epilogue(self.params, D, S, M, st3, abar, cbar, Ng)
        return float(cost), grad, self.alpha.reshape(-1, 1).copy(), self.Li.copy()
