// Feature map of the SCFGP objective and the data / operand staging kernels.
//   Phi = e^b sqrt(2/M) [cos Z | sin Z],  Z = X~ Fall  or  (X~ Lall) Rall        (SCFGP/SCFGP.py:98-102, :139-142)
// Built on tile_engine.h; templated on the storage type T (double | float) of Phi.
#include "kernels.h"
#include "tile_engine.h"

#include <algorithm>

#define SMEM_DECL extern __shared__ __attribute__((aligned(16))) char smem_raw[]
#ifndef SCFGP_FMAP_WGM
#define SCFGP_FMAP_WGM 4     // 8 waves: one wave's fp64 sincos overlaps another's projection MFMAs
#endif
#ifndef SCFGP_FMAP_MIN_WGS
#define SCFGP_FMAP_MIN_WGS 6144   // workgroups below which a row block's column tiles are spread over several workgroups
#endif
#ifndef SCFGP_FMAP_REG
#define SCFGP_FMAP_REG 1          // register-resident feature-map kernel for contractions 16 or 32 deep
#endif
#ifndef SCFGP_FMAP_REG_MIN_WGS
#define SCFGP_FMAP_REG_MIN_WGS 16384
#endif
typedef TileCfg<double, 128, 64, 16, SCFGP_FMAP_WGM, 2, 16, true> FmapCfg;    // swizzled X~ image (TrLoader)

// --------------------------------------------------------------------------
// feature map:  Z = X~ . Fall  (fp64 MFMA, K-dim = Dp),  Phi = s [cos Z | sin Z]
// --------------------------------------------------------------------------
// One workgroup: 128 rows x the column tiles [jt0, jt0 + njt_wg) of Z, as one stream of k-tiles (the contraction is
// only Dp or Sp deep: 2..4 k-tiles for the headline shape, so tile-per-workgroup launches were all prologue).
template <typename T>
__global__ __launch_bounds__(FmapCfg::THREADS) void featuremap_kernel(
    const double* __restrict__ Xt, const double* __restrict__ Fall, const Scal* __restrict__ sc,
    T* __restrict__ Phi, int Dp, int Jp, int Kp, int J, int64_t N, int njt, int ncs) {
    typedef FmapCfg Cfg;
    SMEM_DECL;
    double* smem = reinterpret_cast<double*>(smem_raw);
    const int cs = blockIdx.x % ncs;                            // column range of this workgroup
    const int64_t rb = blockIdx.x / ncs;
    const int per = (njt + ncs - 1) / ncs, jt0 = cs * per, nseg = (jt0 + per <= njt ? per : njt - jt0);
    if (nseg <= 0) return;
    const int nkt = Dp / Cfg::BK;
    TrLoader<double, double, Cfg::BM, Cfg::BK, Cfg::LDA, Cfg::THREADS, false, Cfg::SWZA> la(Xt + rb * Cfg::BM * Dp, Dp, threadIdx.x);
    NatLoader<double, double, Cfg::BN, Cfg::BK, Cfg::LDB, Cfg::THREADS, false, false> lb(Fall + jt0 * Cfg::BN, Jp, threadIdx.x);
    typename Cfg::MTr::acc_t acc[Cfg::TM][Cfg::TN];
    acc_zero<Cfg>(acc);
    const double s = sc->s;
    const T s_hi = (T)s, s_lo = (T)(s - (double)s_hi);
    AccCoord<Cfg> co;
    // the next segment starts at k = 0 again, one column tile to the right
    tile_mainloop_segments<Cfg>(la, lb, nkt, nseg, -(int64_t)Dp, (int64_t)Cfg::BN - (int64_t)Dp * Jp, acc, smem,
        [&](int seg, typename Cfg::MTr::acc_t (&a)[Cfg::TM][Cfg::TN]) {
            // all sin / cos values of the tile first (independent chains the scheduler can interleave), then the stores
            T vs[Cfg::TM][Cfg::TN][Cfg::MTr::NACC], vc[Cfg::TM][Cfg::TN][Cfg::MTr::NACC];
#pragma unroll
            for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < Cfg::TN; ++tn)
#pragma unroll
                    for (int r = 0; r < Cfg::MTr::NACC; ++r) {
                        T sn, cs_;                                         // fp64 or fp32 kernels by output type
                        fast_sincos(a[tm][tn][r], sn, cs_);
                        // scale s = s_hi + s_lo in T: a rounded scale alone would bias every entry of Phi the same way
                        vc[tm][tn][r] = fma(cs_, s_hi, cs_ * s_lo);
                        vs[tm][tn][r] = fma(sn, s_hi, sn * s_lo);
                    }
#pragma unroll
            for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
                for (int r = 0; r < Cfg::MTr::NACC; ++r) {
                    const int64_t n = rb * Cfg::BM + co.row(tm, r);
                    T* __restrict__ prow = Phi + n * Kp;
#pragma unroll
                    for (int tn = 0; tn < Cfg::TN; ++tn) {
                        const int j = (jt0 + seg) * Cfg::BN + co.col(tn);
                        if (j < J) {
                            prow[j] = n < N ? vc[tm][tn][r] : (T)0;
                            prow[J + j] = n < N ? vs[tm][tn][r] : (T)0;
                        }
                    }
                }
        });
}

// Shallow contractions (4 NK <= 36 live rows: the rank-S projection or a small D; rows beyond are zero padding and are
// not multiplied): no LDS and no barriers.  A wave keeps the
// fp64 MFMA fragments of its 32 rows in registers for its whole life and streams the 32-column tiles of Z through
// them; the fragments of Fall come straight from L2 (the matrix is a few hundred KB), fetched for the next tile while
// the current one goes through sin / cos.  Waves are independent, so one wave's MFMAs run under another's VALU work.
// MFMA column (tn, i) is Z column 2 i + tn of the tile: a lane ends up with two adjacent columns and stores them as
// one vector, 16 lanes cover a full 128-byte line of a Phi row.
template <typename T, int NK>
__global__ __launch_bounds__(256) void featuremap_reg_kernel(
    const double* __restrict__ A, int lda, const double* __restrict__ Fall, const Scal* __restrict__ sc,
    T* __restrict__ Phi, int Jp, int Kp, int J, int64_t N, int nct, int ncs) {
    typedef MT<double, 16> M;
    typedef T tv2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, i = lane & 15;
    const int cs = blockIdx.x % ncs;
    const int64_t rb = blockIdx.x / ncs;
    const int per = (nct + ncs - 1) / ncs, ct0 = cs * per, nseg = ct0 + per <= nct ? per : nct - ct0;
    if (nseg <= 0) return;
    const int64_t row0 = rb * 128 + wave * 32;
    double a[2][NK];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) a[tm][ks] = A[(row0 + tm * 16 + i) * lda + ks * 4 + q];
    const double* bp = Fall + (int64_t)q * Jp + ct0 * 32 + 2 * i;
    v2d b[NK];
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) b[ks] = *reinterpret_cast<const v2d*>(bp + (int64_t)ks * 4 * Jp);
    const double s = sc->s;
    const T s_hi = (T)s, s_lo = (T)(s - (double)s_hi);
    const bool vec = (J & 1) == 0;                              // the sine half starts at column J
    for (int seg = 0; seg < nseg; ++seg) {
        v4d acc[2][2];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = v4d{0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < NK; ++ks)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) M::mfma(acc[tm][tn], a[tm][ks], b[ks][tn]);
        bp += 32;
        if (seg + 1 < nseg) {
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) b[ks] = *reinterpret_cast<const v2d*>(bp + (int64_t)ks * 4 * Jp);
        }
        const int c = (ct0 + seg) * 32 + 2 * i;
        T vs[2][4][2], vc[2][4][2];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    T sn, cs_;
                    fast_sincos(acc[tm][tn][r], sn, cs_);
                    // scale s = s_hi + s_lo in T: a rounded scale alone would bias every entry of Phi the same way
                    vc[tm][r][tn] = fma(cs_, s_hi, cs_ * s_lo);
                    vs[tm][r][tn] = fma(sn, s_hi, sn * s_lo);
                }
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t n = row0 + tm * 16 + M::crow(lane, r);
                T* __restrict__ prow = Phi + n * Kp;
                const bool live = n < N;                       // padding rows of Phi are zero
                const T c0 = live ? vc[tm][r][0] : (T)0, c1 = live ? vc[tm][r][1] : (T)0;
                const T s0 = live ? vs[tm][r][0] : (T)0, s1 = live ? vs[tm][r][1] : (T)0;
                if (vec) {
                    if (c < J) {
                        *reinterpret_cast<tv2*>(prow + c) = tv2{c0, c1};
                        *reinterpret_cast<tv2*>(prow + J + c) = tv2{s0, s1};
                    }
                } else {
                    if (c < J) { prow[c] = c0; prow[J + c] = s0; }
                    if (c + 1 < J) { prow[c + 1] = c1; prow[J + c + 1] = s1; }
                }
            }
    }
}

// T~ = X~ . Lall (fp64 MFMA), the first factor of the rank-S projection
__global__ __launch_bounds__(FmapCfg::THREADS) void project_kernel(
    const double* __restrict__ Xt, const double* __restrict__ Lall, double* __restrict__ Tt, int Dp, int Sp, int Spp, int njt) {
    typedef FmapCfg Cfg;
    SMEM_DECL;
    double* smem = reinterpret_cast<double*>(smem_raw);
    const int jt = blockIdx.x % njt;
    const int64_t rb = blockIdx.x / njt;
    TrLoader<double, double, Cfg::BM, Cfg::BK, Cfg::LDA, Cfg::THREADS, false, Cfg::SWZA> la(Xt + rb * Cfg::BM * Dp, Dp, threadIdx.x);
    NatLoader<double, double, Cfg::BN, Cfg::BK, Cfg::LDB, Cfg::THREADS, false, false> lb(Lall + jt * Cfg::BN, Spp, threadIdx.x);
    typename Cfg::MTr::acc_t acc[Cfg::TM][Cfg::TN];
    acc_zero<Cfg>(acc);
    tile_mainloop<Cfg>(la, lb, Dp / Cfg::BK, acc, smem);
    AccCoord<Cfg> co;
#pragma unroll
    for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < Cfg::TN; ++tn) {
            const int j = jt * Cfg::BN + co.col(tn);
            if (j >= Sp) continue;
#pragma unroll
            for (int r = 0; r < Cfg::MTr::NACC; ++r) Tt[(rb * Cfg::BM + co.row(tm, r)) * Sp + j] = acc[tm][tn][r];
        }
}

template <typename T>
void FmapKernels<T>::featuremap(const Geom& g, const double* Xt, const Projection& pr, const Scal* sc, T* Phi, hipStream_t st) {
    const int njt = g.Jp / FmapCfg::BN;
    const int64_t nrb = g.Np / FmapCfg::BM;
    // column ranges per row block: one for large N (>= 3 workgroups per CU and several rounds of them), more for small N
    int ncs = (int)std::min<int64_t>(njt, std::max<int64_t>(1, (SCFGP_FMAP_MIN_WGS + nrb - 1) / nrb));
    const auto launch = [&](auto kernel, const double* A, const double* Bm, int Kd) {
        allow_big_lds(kernel, FmapCfg::LDS_BYTES);
        hipLaunchKernelGGL(kernel, dim3((unsigned)(ncs * nrb)), dim3(FmapCfg::THREADS), FmapCfg::LDS_BYTES, st,
                           A, Bm, sc, Phi, Kd, g.Jp, g.Kp, g.J, g.N, njt, ncs);
    };
    const double *A = Xt, *Bm = pr.Fall; int Kd = g.Dp;
    if (g.lowrank) {
        const int Spp = (int)round_up(g.Sp, FmapCfg::BN), njs = Spp / FmapCfg::BN;
        allow_big_lds(project_kernel, FmapCfg::LDS_BYTES);
        hipLaunchKernelGGL(project_kernel, dim3((unsigned)(njs * nrb)), dim3(FmapCfg::THREADS), FmapCfg::LDS_BYTES, st,
                           Xt, pr.Lall, pr.Tt, g.Dp, g.Sp, Spp, njs);
        A = pr.Tt; Bm = pr.Rall; Kd = g.Sp;
    }
    const int live = (g.lowrank ? g.S : g.D) + 1, nk = (live + 3) / 4;     // rows of the contraction that are not padding
    if (SCFGP_FMAP_REG && nk <= 9) {
        const int nct = g.Jp / 32;
        const int rcs = (int)std::min<int64_t>(nct, std::max<int64_t>(1, (SCFGP_FMAP_REG_MIN_WGS + nrb - 1) / nrb));
        const auto reg = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3((unsigned)(rcs * nrb)), dim3(256), 0, st, A, Kd, Bm, sc, Phi, g.Jp, g.Kp, g.J, g.N, nct, rcs);
        };
        // the next instantiated depth that still lies inside the padded leading dimension
        if (nk <= 3 && Kd >= 12) { reg(featuremap_reg_kernel<T, 3>); return; }
        if (nk <= 4 && Kd >= 16) { reg(featuremap_reg_kernel<T, 4>); return; }
        if (nk <= 5 && Kd >= 20) { reg(featuremap_reg_kernel<T, 5>); return; }
        if (nk <= 8 && Kd >= 32) { reg(featuremap_reg_kernel<T, 8>); return; }
        if (Kd >= 36) { reg(featuremap_reg_kernel<T, 9>); return; }
    }
    launch(featuremap_kernel<T>, A, Bm, Kd);
}

// --------------------------------------------------------------------------
// fp64 Kp x Kp matrix -> sweep operand of type T with the padding rows/columns >= K zeroed
template <typename T>
__global__ void convert_kernel(const double* __restrict__ src, T* __restrict__ dst, int K, int Kp) {
    const int64_t n = (int64_t)Kp * Kp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / Kp), c = (int)(i % Kp);
        dst[i] = (r < K && c < K) ? (T)src[i] : (T)0;
    }
}
// dst[k][j] = src[j][k] on the K x K block, zero elsewhere (Li -> the sweep operand Li^T of the predict product)
template <typename T>
__global__ void convert_t_kernel(const double* __restrict__ src, T* __restrict__ dst, int K, int Kp) {
    const int64_t n = (int64_t)Kp * Kp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i / Kp), j = (int)(i % Kp);
        dst[i] = (k < K && j < K) ? (T)src[(int64_t)j * Kp + k] : (T)0;
    }
}
template <typename T>
void FmapKernels<T>::convert_transposed(const double* src, T* dst, int K, int Kp, hipStream_t st) {
    hipLaunchKernelGGL(convert_t_kernel<T>, dim3(2048), dim3(256), 0, st, src, dst, K, Kp);
}
template <typename T>
void FmapKernels<T>::convert(const double* src, T* dst, int K, int Kp, hipStream_t st) {
    hipLaunchKernelGGL(convert_kernel<T>, dim3(2048), dim3(256), 0, st, src, dst, K, Kp);
}
template struct FmapKernels<double>;
template struct FmapKernels<float>;
// --------------------------------------------------------------------------
// data staging
// --------------------------------------------------------------------------
// per-column input scaling of SCFGP/Scaler.py:99-116 applied while packing (predict on raw inputs):
//   mode 0 none | 1 min-max | 2 normal | 3 inv-normal | 4 auto-normal | 5 auto-inv-normal
//   sp = [min | max | boxcox | mu | std], D doubles each
__device__ __forceinline__ double scale_x(double x, int mode, const double* __restrict__ sp, int D, int d) {
    if (mode == 0) return x;
    const double mn = sp[d], mx = sp[D + d], lm = sp[2 * D + d], mu = sp[3 * D + d], sd = sp[4 * D + d];
    if (mode == 1) return (x - mn) / (mx - mn);
    if (mode == 2) return (x - mu) / sd;
    if (mode == 3) return 0.5 * erfc(-((x - mu) / sd) * 0.70710678118654752440);
    const double t = (x - mn) / (mx - mn);
    const double bc = ((t < 0 ? -1.0 : (t > 0 ? 1.0 : 0.0)) * pow(fabs(t), lm) - 1.0) / lm;       // sign(t)|t|^lm
    const double z = (bc - mu) / sd;
    return mode == 4 ? z : 0.5 * erfc(-z * 0.70710678118654752440);
}
__global__ void pack_data_kernel(const double* __restrict__ Xraw, const double* __restrict__ yraw, const int64_t* __restrict__ idx,
                                 double* __restrict__ Xt, double* __restrict__ y, int D, int Dp, int64_t N, int64_t Np,
                                 int mode, const double* __restrict__ sp) {
    const int64_t total = Np * Dp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t n = i / Dp;
        const int d = (int)(i - n * Dp);
        const int64_t src = (n < N && idx) ? idx[n] : n;           // row gather for index-list minibatches
        double v = 0;
        if (n < N) v = d < D ? scale_x(Xraw[src * D + d], mode, sp, D, d) : (d == D ? 1.0 : 0.0);
        Xt[i] = v;
        if (d == 0 && y) y[n] = (n < N && yraw) ? yraw[src] : 0.0;
    }
}
void pack_data(const Geom& g, const double* Xraw, const double* yraw, const int64_t* idx, double* Xt, double* y, hipStream_t st,
               int mode, const double* sp) {
    hipLaunchKernelGGL(pack_data_kernel, dim3(4096), dim3(256), 0, st, Xraw, yraw, idx, Xt, y, g.D, g.Dp, g.N, g.Np, mode, sp);
}
__global__ void pad_square_kernel(const double* __restrict__ src, int K, int Kp, double* __restrict__ dst) {
    const int64_t total = (int64_t)Kp * Kp;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int i = (int)(e / Kp), j = (int)(e % Kp);
        dst[e] = (i < K && j < K) ? src[(int64_t)i * K + j] : (i == j ? 1.0 : 0.0);
    }
}
void pad_square(const double* src, int K, int Kp, double* dst, hipStream_t st) {
    hipLaunchKernelGGL(pad_square_kernel, dim3(1024), dim3(256), 0, st, src, K, Kp, dst);
}
