// O(P) kernels either side of the sweeps: unpacking the flat hyper-parameter vector
// (SCFGP/SCFGP.py:74-90,98,103), the frequency penalty (:114-117,127) and the chain
// rule from (X~^T Zbar) back to the flat gradient -- the tail of TT.grad (:129).
#include "kernels.h"

// params layout (SCFGP.py:72): [a b c | l_f (D*S) | r_f (M*S) | l_p (S) | p (M)]
__global__ void scal_kernel(const double* __restrict__ params, int M, Scal* sc) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double a = params[0], b = params[1], c = params[2];
    sc->a = a; sc->b = b; sc->c = c;
    sc->s = exp(b) * sqrt(2.0 / M);
    sc->e2a = exp(2.0 * a);
    sc->lam = sc->e2a + 1e-6;
    sc->kappa = log(1.0 + exp(c));
    sc->em2a = exp(-2.0 * a);
    sc->sigc = 1.0 / (1.0 + exp(-c));
}

// F[d][m] = sum_s l_F[d][s] r_F[m][s]                                (SCFGP.py:83)
__global__ void f_kernel(const double* __restrict__ params, int D, int S, int M, double* __restrict__ F) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)D * M) return;
    const int d = (int)(i / M), m = (int)(i % M);
    const double* lF = params + 3 + (int64_t)d * S;
    const double* rF = params + 3 + (int64_t)D * S + (int64_t)m * S;
    double s = 0;
    for (int k = 0; k < S; ++k) s += lF[k] * rF[k];
    F[i] = s;
}

// Fall (Dp x Jp): rows d<D = [l_F | F]; row D = phase offsets [l_P - mean_d l_F | P - mean_d F]  (SCFGP.py:88-89)
__global__ void fall_kernel(const double* __restrict__ params, const double* __restrict__ F, int D, int S, int M, int Dp, int Jp,
                            double* __restrict__ Fall) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)Dp * Jp) return;
    const int d = (int)(i / Jp), j = (int)(i % Jp);
    const int J = S + M;
    const double* lF = params + 3;
    double v = 0;
    if (j < J) {
        if (d < D) {
            v = j < S ? lF[(int64_t)d * S + j] : F[(int64_t)d * M + (j - S)];
        } else if (d == D) {
            double s = 0;
            if (j < S) {
                for (int k = 0; k < D; ++k) s += lF[(int64_t)k * S + j];
                v = params[3 + (int64_t)D * S + (int64_t)M * S + j] - s / D;
            } else {
                for (int k = 0; k < D; ++k) s += F[(int64_t)k * M + (j - S)];
                v = params[3 + (int64_t)D * S + (int64_t)M * S + S + (j - S)] - s / D;
            }
        }
    }
    Fall[i] = v;
}

// rank-S form of the same projection: X~ Fall = (X~ Lall) Rall with
//   Lall (Dp x Spp) = [l_F | e_D | 0]     -> T~ = X~ Lall = [X l_F | 1 | 0]
//   Rall (Sp x Jp): rows s < S = [e_s | r_F[:, s]^T], row S = the phase offsets (row D of Fall)
__global__ void lowrank_kernel(const double* __restrict__ params, const double* __restrict__ Fall, int D, int S, int M, int Dp,
                               int Sp, int Spp, int Jp, double* __restrict__ Lall, double* __restrict__ Rall) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nl = (int64_t)Dp * Spp, nr = (int64_t)Sp * Jp;
    if (i < nl) {
        const int d = (int)(i / Spp), c = (int)(i % Spp);
        Lall[i] = (c < S && d < D) ? params[3 + (int64_t)d * S + c] : ((c == S && d == D) ? 1.0 : 0.0);
    } else if (i < nl + nr) {
        const int64_t e = i - nl;
        const int s = (int)(e / Jp), j = (int)(e % Jp);
        double v = 0;
        if (s < S) v = j < S ? (s == j ? 1.0 : 0.0) : (j < S + M ? params[3 + (int64_t)D * S + (int64_t)(j - S) * S + s] : 0.0);
        else if (s == S) v = Fall[(int64_t)D * Jp + j];
        Rall[e] = v;
    }
}

void unpack_params(const Geom& g, const double* params, double* F, double* Fall, double* Lall, double* Rall, Scal* sc, hipStream_t st) {
    hipLaunchKernelGGL(scal_kernel, dim3(1), dim3(64), 0, st, params, g.M, sc);
    const int64_t nf = (int64_t)g.D * g.M, nfa = (int64_t)g.Dp * g.Jp;
    hipLaunchKernelGGL(f_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, st, params, g.D, g.S, g.M, F);
    hipLaunchKernelGGL(fall_kernel, dim3((unsigned)((nfa + 255) / 256)), dim3(256), 0, st, params, F, g.D, g.S, g.M, g.Dp, g.Jp, Fall);
    if (g.lowrank) {
        const int Spp = (int)round_up(g.Sp, 64);
        const int64_t n = (int64_t)g.Dp * Spp + (int64_t)g.Sp * g.Jp;
        hipLaunchKernelGGL(lowrank_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, params, Fall, g.D, g.S, g.M, g.Dp, g.Sp,
                           Spp, g.Jp, Lall, Rall);
    }
}

// ---------------------------------------------------------------------------
// penalty statistics: per-row mean / population std of F (over M) and l_F (over S)
//   work = [mF(D) sF(D) ml(D) sl(D) | mu_w sig_w mu_l sig_l | Fbar (D*M)]
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pen_rows_kernel(const double* __restrict__ params, const double* __restrict__ F, int D, int S,
                                                       int M, double* __restrict__ work) {
    __shared__ double red[256];
    const int d = blockIdx.x;
    for (int which = 0; which < 2; ++which) {
        const double* row = which == 0 ? F + (int64_t)d * M : params + 3 + (int64_t)d * S;
        const int n = which == 0 ? M : S;
        double s = 0;
        for (int k = threadIdx.x; k < n; k += 256) s += row[k];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int m = 128; m >= 1; m >>= 1) { if (threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m]; __syncthreads(); }
        const double mean = red[0] / n;
        __syncthreads();
        s = 0;
        for (int k = threadIdx.x; k < n; k += 256) { const double t = row[k] - mean; s += t * t; }
        red[threadIdx.x] = s;
        __syncthreads();
        for (int m = 128; m >= 1; m >>= 1) { if (threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m]; __syncthreads(); }
        if (threadIdx.x == 0) { work[(2 * which) * D + d] = mean; work[(2 * which + 1) * D + d] = sqrt(red[0] / n); }
        __syncthreads();
    }
}
__global__ void pen_sums_kernel(int D, int S, int M, double* __restrict__ work, double* __restrict__ scalars) {
    if (threadIdx.x != 0) return;
    double s[4] = {0, 0, 0, 0};
    for (int q = 0; q < 4; ++q)
        for (int d = 0; d < D; ++d) s[q] += work[q * D + d];
    for (int q = 0; q < 4; ++q) work[4 * D + q] = s[q];
    const double mu_w = s[0], sig_w = s[1], mu_l = s[2], sig_l = s[3];
    // kl(mu,sig) = sig + mu^2 - log sig   (SCFGP.py:94), weights M and S (SCFGP.py:127)
    scalars[R_PEN] = ((sig_w + mu_w * mu_w - log(sig_w)) * M + (sig_l + mu_l * mu_l - log(sig_l)) * S) / (S + M);
}
// Fbar[d][m] = dcost*N / dF[d][m]  (data term, centring correction, penalty)
//   rank-S form (TZ != NULL): the data term stays factored (it reaches the gradient through TZ and XU), so Fbar holds the centring
//   correction and the penalty only; the column sums of Zbar are row S of TZ
__global__ void fbar_kernel(const double* __restrict__ F, const double* __restrict__ XZ, int64_t ldxz, int D, int S, int M,
                            double* __restrict__ work, const double* __restrict__ TZ, int64_t ldtz) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)D * M) return;
    const int d = (int)(i / M), m = (int)(i % M);
    const double mu_w = work[4 * D + 0], sig_w = work[4 * D + 1];
    const double wgt = (double)M / (S + M);
    const double pen = wgt * ((1.0 - 1.0 / sig_w) * (F[i] - work[d]) / (M * work[D + d]) + 2.0 * mu_w / M);
    if (TZ) work[4 * D + 4 + i] = -TZ[(int64_t)S * ldtz + S + m] / D + pen;
    else work[4 * D + 4 + i] = XZ[(int64_t)d * ldxz + S + m] - XZ[(int64_t)D * ldxz + S + m] / D + pen;
}
// grad[3..] = [lbar_F, rbar_F, lbar_P, Pbar] / N
__global__ void grad_tail_kernel(const double* __restrict__ params, const double* __restrict__ XZ, int64_t ldxz, int D, int S, int M,
                                 const double* __restrict__ work, double invN, double* __restrict__ grad,
                                 const double* __restrict__ TZ, int64_t ldtz) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nDS = (int64_t)D * S, nMS = (int64_t)M * S;
    const double* lF = params + 3;
    const double* Fbar = work + 4 * D + 4;
    if (i < nDS) {
        return;                                                       // lbar_F: lbar_kernel (one wave per entry)
    } else if (i < nDS + nMS) {
        const int64_t t = i - nDS;
        const int m = (int)(t / S), s = (int)(t % S);
        double v = 0;
        for (int d = 0; d < D; ++d) v += Fbar[(int64_t)d * M + m] * lF[(int64_t)d * S + s];
        if (TZ) v += TZ[(int64_t)s * ldtz + S + m];                   // (l_F^T X^T Zbar_M)[s][m]: the data term of Fbar^T l_F
        grad[3 + i] = v * invN;
    } else if (i < nDS + nMS + S + M) {
        const int j = (int)(i - nDS - nMS);                           // column of X~^T Zbar's ones row
        grad[3 + i] = (TZ ? TZ[(int64_t)S * ldtz + j] : XZ[(int64_t)D * ldxz + j]) * invN;
    }
}

// lbar_F[d][s] = direct term + penalty + sum_m Fbar[d][m] r_F[m][s]: one wave per entry (the sum over M is the long one)
__global__ __launch_bounds__(256) void lbar_kernel(const double* __restrict__ params, const double* __restrict__ XZ, int64_t ldxz, int D,
                                                   int S, int M, const double* __restrict__ work, double invN,
                                                   double* __restrict__ grad, const double* __restrict__ TZ, int64_t ldtz,
                                                   const double* __restrict__ XU, int64_t ldxu) {
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= (int64_t)D * S) return;
    const int lane = threadIdx.x & 63, d = (int)(i / S), s = (int)(i % S);
    const double* rF = params + 3 + (int64_t)D * S;
    const double* Fbar = work + 4 * D + 4;
    double v = 0;
    for (int m = lane; m < M; m += 64) v += Fbar[(int64_t)d * M + m] * rF[(int64_t)m * S + s];
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) v += __shfl_xor(v, k);
    if (lane == 0) {
        const double mu_l = work[4 * D + 2], sig_l = work[4 * D + 3];
        const double wgt = (double)S / (S + M);
        // rank-S form: XU = X^T (Zbar_L + Zbar_M r_F) already holds the direct term and the data part of Fbar r_F
        v += (TZ ? XU[(int64_t)d * ldxu + s] - TZ[(int64_t)S * ldtz + s] / D : XZ[(int64_t)d * ldxz + s] - XZ[(int64_t)D * ldxz + s] / D)
           + wgt * ((1.0 - 1.0 / sig_l) * (params[3 + i] - work[2 * D + d]) / (S * work[3 * D + d]) + 2.0 * mu_l / S);
        grad[3 + i] = v * invN;
    }
}

void grad_epilogue(const Geom& g, const double* params, const double* F, const double* XZ, int64_t ldxz, double* work,
                   double* scalars, int64_t Nglobal, double* grad, hipStream_t st, const double* TZ, int64_t ldtz, const double* XU,
                   int64_t ldxu) {
    const int D = g.D, S = g.S, M = g.M;
    hipLaunchKernelGGL(pen_rows_kernel, dim3(D), dim3(256), 0, st, params, F, D, S, M, work);
    hipLaunchKernelGGL(pen_sums_kernel, dim3(1), dim3(64), 0, st, D, S, M, work, scalars);
    if (!grad) return;
    const int64_t nf = (int64_t)D * M, nt = (int64_t)g.P - 3;
    hipLaunchKernelGGL(fbar_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, st, F, XZ, ldxz, D, S, M, work, TZ, ldtz);
    hipLaunchKernelGGL(grad_tail_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, params, XZ, ldxz, D, S, M, work,
                       1.0 / (double)Nglobal, grad, TZ, ldtz);
    hipLaunchKernelGGL(lbar_kernel, dim3((unsigned)(((int64_t)D * S + 3) / 4)), dim3(256), 0, st, params, XZ, ldxz, D, S, M, work,
                       1.0 / (double)Nglobal, grad, TZ, ldtz, XU, ldxu);
}

__global__ void status_kernel(double* __restrict__ xs, double ran1, double cap1, double cap2, double fail) {
    if (threadIdx.x == 0) { xs[XS_RAN1] = ran1; xs[XS_CAP1] = cap1; xs[XS_CAP2] = cap2; xs[XS_FAIL] = fail; }
}
void write_status(double* xs, double ran1, double cap1, double cap2, double fail, hipStream_t st) {
    hipLaunchKernelGGL(status_kernel, dim3(1), dim3(64), 0, st, xs, ran1, cap1, cap2, fail);
}

// cost and the three scalar gradient entries  (SCFGP.py:125-128)
__global__ void finalize_kernel(const Scal* __restrict__ sc, double* __restrict__ scalars, const double* __restrict__ yy,
                                const double* __restrict__ t2kb, int M, double N,
                                double* __restrict__ grad, int want_grad) {
    if (threadIdx.x != 0) return;
    const double T1 = scalars[R_LOGDET], T2 = t2kb[0];
    const double T3 = sc->em2a * (yy[0] - scalars[R_GTALPHA]);
    const double T4 = 2.0 * (N - M) * sc->a;
    scalars[R_COST] = (T1 + T2 + T3 + T4 + scalars[R_PEN]) / N;
    if (want_grad) {
        grad[0] = (2.0 * sc->e2a * scalars[R_TRABAR] - 2.0 * T3 + 2.0 * (N - M)) / N;
        // bbar = sum_n Phibar_n . phi_n in closed form (kernels_kstage.hip: kstage_bbar; t2kb[2], [3] = sum q v, sum p mu)
        scalars[R_BBAR] = 2.0 * scalars[R_TRAG] + scalars[R_UTG] + 2.0 * t2kb[2] + t2kb[3];
        grad[1] = scalars[R_BBAR] / N;
        grad[2] = t2kb[1] * sc->sigc / N;
    }
}
void finalize_cost(const Geom& g, const Scal* sc, double* scalars, const double* yy, const double* t2kb,
                   int64_t Nglobal, double* grad, int want_grad, hipStream_t st) {
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(64), 0, st, sc, scalars, yy, t2kb, g.M, (double)Nglobal, grad, want_grad);
}

// ---------------------------------------------------------------------------
// Update rules on the device (SCFGP/Optimizer.py:99-382) followed by the Nesterov wrapper exactly
// as the reference applies it (:62-97): to the FIRST key of the rule's update dictionary --
// params for sgd, `accu` for adagrad/rmsprop/adadelta, `m` for adam/adamax.  All states are read
// before any is written (Theano's simultaneous-update semantics).
//   st = [s1 (P) | s2 (P) | vel (P)], tctr = step counter t (device scalar), hist[t] = cost
// ---------------------------------------------------------------------------
__global__ void opt_update_kernel(int algo, OptHyper h, int P, double* __restrict__ theta, const double* __restrict__ grad,
                                  double* __restrict__ st, const double* __restrict__ tctr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    double* s1 = st; double* s2 = st + P; double* vel = st + 2 * (int64_t)P;
    const double g = grad[i], th = theta[i], a1 = s1[i], a2 = s2[i], v = vel[i];
    const double t = tctr[0] + 1.0;
    const bool nest = h.momentum >= 0.0;
    double th_new = th, a1_new = a1, a2_new = a2, first_step;      // first_step: plain new value of the rule's first key
    switch (algo) {
    case 0:   // sgd                                            :99-119
        th_new = th - h.lr * g; first_step = th_new; break;
    case 1:   // adagrad                                        :121-164
        a1_new = a1 + g * g; th_new = th - h.lr * g / sqrt(a1_new + h.eps); first_step = a1_new; break;
    case 2:   // rmsprop (the formula its docstring states)    :166-213
        a1_new = h.b1 * a1 + (1.0 - h.b1) * g * g; th_new = th - h.lr * g / sqrt(a1_new + h.eps); first_step = a1_new; break;
    case 3: { // adadelta                                       :215-276
        a1_new = h.b1 * a1 + (1.0 - h.b1) * g * g;
        const double up = g * sqrt(a2 + h.eps) / sqrt(a1_new + h.eps);
        th_new = th - h.lr * up; a2_new = h.b1 * a2 + (1.0 - h.b1) * up * up; first_step = a1_new; break; }
    case 4: { // adam                                           :278-331
        const double at = h.lr * sqrt(1.0 - pow(h.b2, t)) / (1.0 - pow(h.b1, t));
        a1_new = h.b1 * a1 + (1.0 - h.b1) * g; a2_new = h.b2 * a2 + (1.0 - h.b2) * g * g;
        th_new = th - at * a1_new / (sqrt(a2_new) + h.eps); first_step = a1_new; break; }
    default: { // adamax                                        :333-382
        const double at = h.lr / (1.0 - pow(h.b1, t));
        a1_new = h.b1 * a1 + (1.0 - h.b1) * g; a2_new = fmax(h.b2 * a2, fabs(g));
        th_new = th - at * a1_new / (a2_new + h.eps); first_step = a1_new; break; }
    }
    if (nest) {                                                   // :92-96 on the first key
        const double old_first = algo == 0 ? th : a1;
        const double x = h.momentum * v + first_step - old_first;
        vel[i] = x;
        if (algo == 0) th_new = h.momentum * x + first_step; else a1_new = h.momentum * x + first_step;
    }
    theta[i] = th_new; s1[i] = a1_new; s2[i] = a2_new;
}
__global__ void opt_tick_kernel(double* tctr, const double* __restrict__ scalars, double* __restrict__ hist, int hist_cap) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int t = (int)tctr[1];                                   // iterations done in this scfgp_train call
    if (t < hist_cap) hist[t] = scalars[R_COST];
    tctr[0] += 1.0; tctr[1] += 1.0;
}
void opt_update(int algo, const OptHyper& h, int P, double* theta, const double* grad, double* st, double* tctr,
                const double* scalars, double* hist, int hist_cap, hipStream_t stream) {
    hipLaunchKernelGGL(opt_update_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, algo, h, P, theta, grad, st, tctr);
    hipLaunchKernelGGL(opt_tick_kernel, dim3(1), dim3(64), 0, stream, tctr, scalars, hist, hist_cap);
}

// ----------------------------------------------------------------------------------------------
// Prediction post-processing (SCFGP/SCFGP.py:281-293, SCFGP/Scaler.py:118-135): y-scaler backward
// transform of mu and of the mu +- std bounds, and the six validation metrics, so that predict on a
// large test set returns from the GPU with nothing left to do on the host.
// ----------------------------------------------------------------------------------------------
// inverse normal CDF, Wichura's algorithm AS 241 (PPND16, relative accuracy ~1e-16); scipy's
// norm.ppf conventions at the ends: ppf(0) = -inf, ppf(1) = +inf, NaN outside [0, 1]
__device__ double norm_ppf(double p) {
    if (!(p >= 0.0 && p <= 1.0)) return __builtin_nan("");
    if (p == 0.0) return -__builtin_inf();
    if (p == 1.0) return __builtin_inf();
    const double q = p - 0.5;
    if (fabs(q) <= 0.425) {
        const double r = 0.180625 - q * q;
        const double num = (((((((2.5090809287301226727e3 * r + 3.3430575583588128105e4) * r + 6.7265770927008700853e4) * r
                                + 4.5921953931549871457e4) * r + 1.3731693765509461125e4) * r + 1.9715909503065514427e3) * r
                             + 1.3314166789178437745e2) * r + 3.3871328727963666080e0);
        const double den = (((((((5.2264952788528545610e3 * r + 2.8729085735721942674e4) * r + 3.9307895800092710610e4) * r
                                + 2.1213794301586595867e4) * r + 5.3941960214247511077e3) * r + 6.8718700749205790830e2) * r
                             + 4.2313330701600911252e1) * r + 1.0);
        return q * num / den;
    }
    double r = sqrt(-log(q < 0 ? p : 1.0 - p));
    double v;
    if (r <= 5.0) {
        r -= 1.6;
        const double num = (((((((7.74545014278341407640e-4 * r + 2.27238449892691845833e-2) * r + 2.41780725177450611770e-1) * r
                                + 1.27045825245236838258e0) * r + 3.64784832476320460504e0) * r + 5.76949722146069140550e0) * r
                             + 4.63033784615654529590e0) * r + 1.42343711074968357734e0);
        const double den = (((((((1.05075007164441684324e-9 * r + 5.47593808499534494600e-4) * r + 1.51986665636164571966e-2) * r
                                + 1.48103976427480074590e-1) * r + 6.89767334985100004550e-1) * r + 1.67638483018380384940e0) * r
                             + 2.05319162663775882187e0) * r + 1.0);
        v = num / den;
    } else {
        r -= 5.0;
        const double num = (((((((2.01033439929228813265e-7 * r + 2.71155556874348757815e-5) * r + 1.24266094738807843860e-3) * r
                                + 2.65321895265761230930e-2) * r + 2.96560571828504891230e-1) * r + 1.78482653991729133580e0) * r
                             + 5.46378491116411436990e0) * r + 6.65790464350110377720e0);
        const double den = (((((((2.04426310338993978564e-15 * r + 1.42151175831644588870e-7) * r + 1.84631831751005468180e-5) * r
                                + 7.86869131145613259100e-4) * r + 1.48753612908506148525e-2) * r + 1.36929880922735805310e-1) * r
                             + 5.99832206555887937690e-1) * r + 1.0);
        v = num / den;
    }
    return q < 0 ? -v : v;
}
// Scaler.backward_transform for one column; sp = [min, max, boxcox, mu, std]; modes as scale_x.
// Mode 3 is the reference's expression (ppf(x) - mu) / std, which is not the inverse of its forward map.
__device__ __forceinline__ double y_backward(double x, int mode, const double* __restrict__ sp) {
    const double mn = sp[0], mx = sp[1], lm = sp[2], mu = sp[3], sd = sp[4];
    if (mode == 0) return x;
    if (mode == 1) return x * (mx - mn) + mn;
    if (mode == 2) return x * sd + mu;
    if (mode == 3) return (norm_ppf(x) - mu) / sd;
    const double t = mode == 4 ? x * sd + mu : norm_ppf(x) * sd + mu;
    const double u = t * lm + 1.0;
    const double ib = (u < 0 ? -1.0 : (u > 0 ? 1.0 : 0.0)) * pow(fabs(u), 1.0 / lm);
    return ib * (mx - mn) + mn;
}
__global__ __launch_bounds__(1024) void ymean_kernel(const double* __restrict__ ys, int64_t n, double* __restrict__ out) {
    __shared__ double red[1024];
    double s = 0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) s += ys[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0] / (double)n;
}
// in place: mu <- mu_y, sd <- std_y = (back(mu+sd) - back(mu-sd)) / 2; with targets, block partials of
// [sum |e|, sum e^2, sum (e/std_y)^2 + log(2 pi std_y^2), sum (ys - mean)^2]
__global__ __launch_bounds__(256) void ypost_kernel(double* __restrict__ mu, double* __restrict__ sd, const double* __restrict__ ys,
                                                    int64_t n, int mode, const double* __restrict__ sp,
                                                    const double* __restrict__ ymean, double* __restrict__ part) {
    __shared__ double red[4][256];
    double acc[4] = {0, 0, 0, 0};
    const double ym = ys ? ymean[0] : 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double m = mu[i], s = sd[i];
        const double my = y_backward(m, mode, sp);
        const double sy = 0.5 * (y_backward(m + s, mode, sp) - y_backward(m - s, mode, sp));
        mu[i] = my; sd[i] = sy;
        if (ys) {
            const double e = my - ys[i], z = e / sy, c = ys[i] - ym;
            acc[0] += fabs(e); acc[1] += e * e; acc[2] += z * z + log(2.0 * M_PI * sy * sy); acc[3] += c * c;
        }
    }
    if (!ys) return;
    for (int k = 0; k < 4; ++k) red[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w)
            for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x < 4) part[blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
}
// out = [MAE, NMAE, MSE, NMSE, MNLP, SCORE]                      (SCFGP/SCFGP.py:286-293)
__global__ void ymetrics_kernel(const double* __restrict__ part, int nparts, int64_t n, double* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s[4] = {0, 0, 0, 0};
    for (int b = 0; b < nparts; ++b)
        for (int k = 0; k < 4; ++k) s[k] += part[b * 4 + k];
    const double mae = s[0] / n, mse = s[1] / n, mnlp = 0.5 * s[2] / n, var = s[3] / n;
    const double nmse = mse / var;
    out[0] = mae; out[1] = mae / sqrt(var); out[2] = mse; out[3] = nmse; out[4] = mnlp; out[5] = nmse / (1.0 + exp(-mnlp));
}
void ypost_mean(const double* ys, int64_t n, double* out, hipStream_t st) {
    hipLaunchKernelGGL(ymean_kernel, dim3(1), dim3(1024), 0, st, ys, n, out);
}
void ypost_chunk(double* mu, double* sd, const double* ys, int64_t n, int mode, const double* sp, const double* ymean,
                 double* part, hipStream_t st) {
    hipLaunchKernelGGL(ypost_kernel, dim3(YPOST_BLOCKS), dim3(256), 0, st, mu, sd, ys, n, mode, sp, ymean, part);
}
void ypost_metrics(const double* part, int nparts, int64_t n, double* out, hipStream_t st) {
    hipLaunchKernelGGL(ymetrics_kernel, dim3(1), dim3(64), 0, st, part, nparts, n, out);
}
