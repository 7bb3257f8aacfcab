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
