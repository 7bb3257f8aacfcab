"""
Torch float64 restatement of the LITERAL reference graph (SCFGP/SCFGP.py:74-129,
including the 30-point Gauss-Hermite tensor of :118-124) whose autograd stands in
for Theano's TT.grad (:129).  TEST INFRASTRUCTURE ONLY -- see scfgp_oracle.py.
"""
import numpy as np
import torch


def cost_literal(params, X, y, S, M):
    """cost of SCFGP.py:128 as a differentiable torch scalar (float64, CPU)."""
    N, D = X.shape
    a, b, c = params[0], params[1], params[2]
    t = 3
    l_F = params[t:t + D * S].reshape(D, S); t += D * S          # :79-80
    r_F = params[t:t + M * S].reshape(M, S); t += M * S          # :81-82
    F = l_F @ r_F.T                                              # :83
    l_P = params[t:t + S].reshape(1, S); t += S                  # :84-85
    P = params[t:t + M].reshape(1, M)                            # :86-87
    l_FC = l_P - l_F.mean(0)[None, :]                            # :88
    FC = P - F.mean(0)[None, :]                                  # :89
    sig2_n, sig_f = torch.exp(2 * a), torch.exp(b)               # :98
    FF = torch.cat((X @ l_F + l_FC, X @ F + FC), 1)              # :99-100
    Phi = sig_f * np.sqrt(2. / M) * torch.cat((torch.cos(FF), torch.sin(FF)), 1)   # :101-102
    noise = torch.log(1 + torch.exp(c))                          # :103
    K = Phi.shape[1]
    A = Phi.T @ Phi + (sig2_n + 1e-6) * torch.eye(K, dtype=Phi.dtype)             # :104-105
    L = torch.linalg.cholesky(A)                                 # :106
    Li = torch.linalg.inv(L)                                     # :107
    beta = Li @ (Phi.T @ y)                                      # :108-109
    alpha = Li.T @ beta                                          # :110
    mu_f = Phi @ alpha                                           # :111
    var_f = ((Phi @ Li.T) ** 2).sum(1)[:, None]                  # :112
    dsp = noise * (var_f + 1)                                    # :113
    mu_l = l_F.mean(1).sum(); sig_l = l_F.std(1, unbiased=False).sum()           # :114-115
    mu_w = F.mean(1).sum(); sig_w = F.std(1, unbiased=False).sum()               # :116-117
    hx, hw = np.polynomial.hermite.hermgauss(30)                 # :118
    herm_x = torch.tensor(hx)[None, None, :]
    herm_w = torch.tensor(hw / np.sqrt(np.pi))[None, None, :]
    herm_f = torch.sqrt(2 * var_f[:, :, None]) * herm_x + mu_f[:, :, None]        # :121
    nlk = (0.5 * herm_f ** 2. - y[:, :, None] * herm_f) / dsp[:, :, None] + 0.5 * (
        torch.log(2 * np.pi * dsp[:, :, None]) + y[:, :, None] ** 2 / dsp[:, :, None])   # :122-123
    enll = herm_w * nlk                                          # :124
    nlml = 2 * torch.log(torch.diagonal(L)).sum() + 2 * enll.sum() + 1. / sig2_n * (
        (y ** 2).sum() - (beta ** 2).sum()) + 2 * (N - M) * a    # :125-126
    kl = lambda mu, sig: sig + mu ** 2 - torch.log(sig)          # :94
    pen = (kl(mu_w, sig_w) * M + kl(mu_l, sig_l) * S) / (S + M)  # :127
    return (nlml + pen) / N, alpha, Li                           # :128


def value_and_grad(X, y, params, S, M):
    Xt = torch.tensor(np.asarray(X, np.float64))
    yt = torch.tensor(np.asarray(y, np.float64).reshape(-1, 1))
    pt = torch.tensor(np.asarray(params, np.float64), requires_grad=True)
    cost, alpha, Li = cost_literal(pt, Xt, yt, S, M)
    cost.backward()
    return float(cost.detach()), pt.grad.numpy().copy(), alpha.detach().numpy(), Li.detach().numpy()


def value_and_grad_chunked(X, y, params, S, M, chunk=65536):
    """The same literal graph with the N x K tensors formed `chunk` rows at a time (gradient checkpointing: every derivative
    is still autograd's, only WHEN the row tensors exist changes), so that N = 1e6 at K = 2112 fits in host memory:
      sweep 1  G = Phi^T Phi, g = Phi^T y, y^T y            (no graph kept)
      K stage  A, L, Li, beta, alpha, the N-free terms of nlml  as a graph on the leaves (params, G, g)
      sweep 2  per chunk: mu_f, var_f, the Gauss-Hermite term on (params, Li, alpha) -> backward into params and into the
               cotangents of Li and alpha
      K stage  backward: cotangents of G and g
      sweep 3  per chunk: <Phi^T Phi, Gbar> + <Phi^T y, gbar> -> backward into params
    Returns (cost, grad, alpha, Li) like value_and_grad."""
    Xt = torch.tensor(np.asarray(X, np.float64))
    yt = torch.tensor(np.asarray(y, np.float64).reshape(-1, 1))
    pt = torch.tensor(np.asarray(params, np.float64), requires_grad=True)
    N, D = Xt.shape
    hx, hw = np.polynomial.hermite.hermgauss(30)
    herm_x = torch.tensor(hx)[None, None, :]
    herm_w = torch.tensor(hw / np.sqrt(np.pi))[None, None, :]

    def unpack(p):
        t = 3
        l_F = p[t:t + D * S].reshape(D, S); t += D * S
        r_F = p[t:t + M * S].reshape(M, S); t += M * S
        F = l_F @ r_F.T
        l_P = p[t:t + S].reshape(1, S); t += S
        P = p[t:t + M].reshape(1, M)
        return l_F, F, l_P - l_F.mean(0)[None, :], P - F.mean(0)[None, :]

    def phi(p, lo, hi):
        l_F, F, l_FC, FC = unpack(p)
        FF = torch.cat((Xt[lo:hi] @ l_F + l_FC, Xt[lo:hi] @ F + FC), 1)
        return torch.exp(p[1]) * np.sqrt(2. / M) * torch.cat((torch.cos(FF), torch.sin(FF)), 1)

    K = 2 * (S + M)
    with torch.no_grad():
        G = torch.zeros(K, K, dtype=torch.float64); g = torch.zeros(K, 1, dtype=torch.float64)
        for lo in range(0, N, chunk):
            Ph = phi(pt, lo, min(N, lo + chunk))
            G += Ph.T @ Ph; g += Ph.T @ yt[lo:lo + chunk]
        yy = (yt ** 2).sum()
    G.requires_grad_(); g.requires_grad_()
    a = pt[0]
    sig2_n = torch.exp(2 * a)
    A = G + (sig2_n + 1e-6) * torch.eye(K, dtype=torch.float64)
    L = torch.linalg.cholesky(A)
    Li = torch.linalg.inv(L)
    beta = Li @ g
    alpha = Li.T @ beta
    l_F, F, _, _ = unpack(pt)
    mu_l = l_F.mean(1).sum(); sig_l = l_F.std(1, unbiased=False).sum()
    mu_w = F.mean(1).sum(); sig_w = F.std(1, unbiased=False).sum()
    kl = lambda mu, sig: sig + mu ** 2 - torch.log(sig)
    pen = (kl(mu_w, sig_w) * M + kl(mu_l, sig_l) * S) / (S + M)
    k_terms = (2 * torch.log(torch.diagonal(L)).sum() + 1. / sig2_n * (yy - (beta ** 2).sum()) + 2 * (N - M) * a + pen) / N
    Li_d = Li.detach().requires_grad_(); al_d = alpha.detach().requires_grad_()
    row_cost = 0.0
    for lo in range(0, N, chunk):
        hi = min(N, lo + chunk)
        Ph = phi(pt, lo, hi); yc = yt[lo:hi]
        mu_f = Ph @ al_d
        var_f = ((Ph @ Li_d.T) ** 2).sum(1)[:, None]
        dsp = torch.log(1 + torch.exp(pt[2])) * (var_f + 1)
        herm_f = torch.sqrt(2 * var_f[:, :, None]) * herm_x + mu_f[:, :, None]
        nlk = (0.5 * herm_f ** 2. - yc[:, :, None] * herm_f) / dsp[:, :, None] + 0.5 * (
            torch.log(2 * np.pi * dsp[:, :, None]) + yc[:, :, None] ** 2 / dsp[:, :, None])
        part = 2 * (herm_w * nlk).sum() / N
        part.backward()
        row_cost += float(part.detach())
    (k_terms + (Li * Li_d.grad).sum() + (alpha * al_d.grad).sum()).backward()
    Gbar, gbar = G.grad, g.grad
    for lo in range(0, N, chunk):
        hi = min(N, lo + chunk)
        Ph = phi(pt, lo, hi)
        (((Ph.T @ Ph) * Gbar).sum() + ((Ph.T @ yt[lo:hi]) * gbar).sum()).backward()
    return float(k_terms.detach()) + row_cost, pt.grad.numpy().copy(), alpha.detach().numpy(), Li.detach().numpy()
