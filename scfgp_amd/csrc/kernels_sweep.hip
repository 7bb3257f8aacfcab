// Row-sweep kernels of the SCFGP objective: everything whose cost scales with N.
// Built on tile_engine.h; templated on the compute type T (double | float).
#include "kernels.h"
#ifndef SCFGP_GRAM_PERM
#define SCFGP_GRAM_PERM 0                 // 1: Gram tiles with permuted tile rows (tile_engine.h: TileCfg PERM): 8 ds_read_b128 per 64 MFMAs
                                          // instead of 32 ds_read_b32 -- measured equal (36.7 / 37.6 against 37.0 / 37.4 ms): the reads are not the bound
#endif
#ifndef SCFGP_DIAG_EPI0
#define SCFGP_DIAG_EPI0 0                 // timing diagnostics of the apply epilogue (wrong numbers): see apply_epilogue
#endif
#ifndef SCFGP_DIAG_EPI1
#define SCFGP_DIAG_EPI1 0
#endif
#include "tile_engine.h"
#include "tile_bf16x3.h"
#include "tile_bf16x3_dma.h"

#include <algorithm>

#define SMEM_DECL extern __shared__ __attribute__((aligned(16))) char smem_raw[]

// Diagnostic build (-DSCFGP_TRACE, library variant "_trace"): every workgroup of the Gram kernel records
// [start, end] on the 100 MHz constant clock, its XCC id and its job kind, so the tail and the spread of job lengths
// can be read off (tests/gpu_gram_trace.py).  No stamp reaches any output; the product build contains none of this.
#ifdef SCFGP_TRACE
constexpr int TRACE_CAP = 1 << 16;
__device__ unsigned long long g_trace[TRACE_CAP][4];
#define TRACE_BEGIN() const unsigned long long tr_t0 = __builtin_amdgcn_s_memrealtime()
#define TRACE_END(kind)                                                                                     \
    do {                                                                                                    \
        __syncthreads();                                                                                    \
        if (threadIdx.x == 0 && blockIdx.x < TRACE_CAP) {                                                   \
            g_trace[blockIdx.x][0] = tr_t0; g_trace[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();      \
            g_trace[blockIdx.x][2] = (unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) |    \
                                     ((unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) << 8); \
            g_trace[blockIdx.x][3] = (unsigned long long)(kind);                                            \
        }                                                                                                   \
    } while (0)
int64_t trace_read(void* host, int64_t max_bytes) {
    const int64_t n = max_bytes < (int64_t)sizeof(g_trace) ? max_bytes : (int64_t)sizeof(g_trace);
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace), n) == hipSuccess ? n : -2;
}
#else
#define TRACE_BEGIN()
#define TRACE_END(kind)
int64_t trace_read(void*, int64_t) { return -1; }
#endif

// --------------------------------------------------------------------------
// tile configurations (tuning knobs)
// --------------------------------------------------------------------------
// Gram: 256 x 128 (fp32) and 128 x 128 output tiles plus a 64 x 128 strip; apply: 256 rows x {128, 64} columns.
// Measured on MI355X (N = 5e5..1e6, K = 2112; tests/gpu_tune.py, profiles/r01_tuning.md):
//   * every MFMA kernel wants two workgroups per CU: 8-wave workgroups within a 128-VGPR budget (the compiler
//     takes up to 256 unless told: amdgpu_waves_per_eu on the Gram kernel); fp64 apply: one 16-wave workgroup;
//   * 64 x 64 wave tiles and one barrier per 64 MFMAs per wave where the registers allow it (apply: BK = 16 on a
//     256 x 128 tile; fp32 Gram: the same tall tile, BK = 32 on its 128 x 128 tiles);
//   * the 32x32x2 fp32 MFMA shape, 192- / 256-wide square tiles and 16-wave 256 x 256 tiles are slower.
#ifndef SCFGP_BK
#define SCFGP_BK 16
#endif
#ifndef SCFGP_GRAM_BK_F32
#define SCFGP_GRAM_BK_F32 32       // one barrier per 64 MFMAs per wave; fits 128 VGPRs only in fp32
#endif
#ifndef SCFGP_F32_MS
#define SCFGP_F32_MS 16          // fp32 MFMA shape: 16 -> 16x16x4, 32 -> 32x32x2
#endif
#ifndef SCFGP_APPLY_BN_F32
#define SCFGP_APPLY_BN_F32 128
#endif
#ifndef SCFGP_APPLY_BM_F32          // experiment knobs of the fp32 apply tile (profiles/r02_tuning.md)
#define SCFGP_APPLY_BM_F32 256
#define SCFGP_APPLY_WGM_F32 4
#define SCFGP_APPLY_WGN_F32 2
#define SCFGP_APPLY_WAVES_F32 8       // upper bound of waves per SIMD the kernel is compiled for
#endif
template <typename T> struct Tune;
template <> struct Tune<float>  {
    static constexpr int MS = SCFGP_F32_MS, GRAM_WGM = 4, GRAM_WGN = 2, GRAM_BK = SCFGP_GRAM_BK_F32, APPLY_BM = SCFGP_APPLY_BM_F32, APPLY_BN = SCFGP_APPLY_BN_F32, APPLY_WGM = SCFGP_APPLY_WGM_F32;
    static constexpr int apply_wgn(int bn) { return bn >= 256 ? 4 : (bn >= 128 ? SCFGP_APPLY_WGN_F32 : 2); }
};
template <> struct Tune<double> {
    static constexpr int MS = 16, GRAM_WGM = 4, GRAM_WGN = 2, GRAM_BK = SCFGP_BK, APPLY_BM = 256, APPLY_BN = 128, APPLY_WGM = 4;
    static constexpr int apply_wgn(int) { return 4; }
};
template <typename T, int TILE> struct GramCfg {
    typedef TileCfg<T, TILE, TILE, Tune<T>::GRAM_BK, Tune<T>::GRAM_WGM, Tune<T>::GRAM_WGN, Tune<T>::MS, false, SCFGP_GRAM_PERM != 0> type;
};
// the 64-high strip below the square tiles: same workgroup size (one launch), 32 x 32 wave tiles
template <typename T> struct GramStripCfg {
    typedef TileCfg<T, 64, 128, Tune<T>::GRAM_BK, Tune<T>::GRAM_WGM / 2, Tune<T>::GRAM_WGN * 2, Tune<T>::MS, false, SCFGP_GRAM_PERM != 0> type;
};
// fp32 only: the tall Gram tile (64 x 64 wave tiles); fp64 would need 16 waves for the same tile
template <typename T> struct GramBigCfg {
    typedef TileCfg<T, 256, 128, SCFGP_BK, Tune<T>::GRAM_WGM, Tune<T>::GRAM_WGN, Tune<T>::MS, false, SCFGP_GRAM_PERM != 0> type;
};
// fp32 only: four strip tiles side by side as one 64 x 512 tile (eight 64 x 64 wave tiles: the MFMA-per-barrier ratio of
// the tall tile; the 64 x 128 strip tiles ran at half its rate and were 6 % of the launch)
template <typename T> struct GramWideCfg {
    typedef TileCfg<T, 64, 512, SCFGP_BK, 1, 8, Tune<T>::MS, false, SCFGP_GRAM_PERM != 0> type;
};
template <typename T, int TILE> struct ApplyCfg {
    typedef TileCfg<T, Tune<T>::APPLY_BM, TILE, SCFGP_BK, Tune<T>::APPLY_WGM, Tune<T>::apply_wgn(TILE), Tune<T>::MS,
                    SCFGP_BK == 16 && Tune<T>::MS == 16> type;                 // swizzled Phi image (TrLoader)
};
// split-precision variant of the apply tile (tile_bf16x3.h): 8 waves of 64 x 64 (BN = 128) or 64 x 32 (BN = 64)
template <int TILE> struct ApplyBf3Cfg { typedef Bf3Cfg<256, TILE, 4, 2> type; };
template <class Cfg> struct IsBf3 { static constexpr bool value = false; };
template <int BM, int BN, int WGM, int WGN> struct IsBf3<Bf3Cfg<BM, BN, WGM, WGN>> { static constexpr bool value = true; };
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
template <typename T> struct XtzCfg { typedef TileCfg<T, 128, 128, 16, 4, 2, Tune<T>::MS> type; };
// row tiles of X~^T Zbar that hold at most 96 / 64 live rows of X~^T (D + 1 = 65 at the headline shape): same 128-wide
// slabs, fewer MFMA rows
template <typename T> struct Xtz96Cfg { typedef TileCfg<T, 96, 128, 16, 2, 4, 16> type; };     // 48-row wave tiles: 16 x 16 MFMA shape only
template <typename T> struct Xtz64Cfg { typedef TileCfg<T, 64, 128, 16, 2, 4, Tune<T>::MS> type; };

// --------------------------------------------------------------------------
// feature map:  Z = X~ . Fall  (fp64 MFMA, K-dim = Dp),  Phi = s [cos Z | sin Z]
// --------------------------------------------------------------------------
// One workgroup: 128 rows x the column tiles [jt0, jt0 + njt_wg) of Z, as one stream of k-tiles (the contraction is
// only Dp or Sp deep: 2..4 k-tiles for the headline shape, so tile-per-workgroup launches were all prologue).
template <typename T, bool ZOUT>
__global__ __launch_bounds__(FmapCfg::THREADS) void featuremap_kernel(
    const double* __restrict__ Xt, const double* __restrict__ Fall, const Scal* __restrict__ sc,
    T* __restrict__ Phi, int Dp, int Jp, int Kp, int J, int64_t N, int njt, int ncs, T* __restrict__ Zout) {
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
                        if constexpr (ZOUT) {                                         // experiment: the phase itself, for ZSRC loaders
                            const double z = a[tm][tn][r];
                            if (j < J) Zout[n * Jp + j] = sizeof(T) == 4 ? (T)fma(-rint(z * 1.5915494309189535e-01), 6.283185307179586e+00, z) : (T)z;
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
void SweepKernels<T>::featuremap(const Geom& g, const double* Xt, const Projection& pr, const Scal* sc, T* Phi, hipStream_t st, T* Zout) {
    const int njt = g.Jp / FmapCfg::BN;
    const int64_t nrb = g.Np / FmapCfg::BM;
    // column ranges per row block: one for large N (>= 3 workgroups per CU and several rounds of them), more for small N
    int ncs = (int)std::min<int64_t>(njt, std::max<int64_t>(1, (SCFGP_FMAP_MIN_WGS + nrb - 1) / nrb));
    const auto launch = [&](auto kernel, const double* A, const double* Bm, int Kd) {
        allow_big_lds(kernel, FmapCfg::LDS_BYTES);
        hipLaunchKernelGGL(kernel, dim3((unsigned)(ncs * nrb)), dim3(FmapCfg::THREADS), FmapCfg::LDS_BYTES, st,
                           A, Bm, sc, Phi, Kd, g.Jp, g.Kp, g.J, g.N, njt, ncs, Zout);
    };
    const double *A = Xt, *Bm = pr.Fall; int Kd = g.Dp;
    if (g.lowrank) {
        const int Spp = (int)round_up(g.Sp, FmapCfg::BN), njs = Spp / FmapCfg::BN;
        allow_big_lds(project_kernel, FmapCfg::LDS_BYTES);
        hipLaunchKernelGGL(project_kernel, dim3((unsigned)(njs * nrb)), dim3(FmapCfg::THREADS), FmapCfg::LDS_BYTES, st,
                           Xt, pr.Lall, pr.Tt, g.Dp, g.Sp, Spp, njs);
        A = pr.Tt; Bm = pr.Rall; Kd = g.Sp;
    }
    if (Zout) { launch(featuremap_kernel<T, true>, A, Bm, Kd); return; }
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
    launch(featuremap_kernel<T, false>, A, Bm, Kd);
}

// --------------------------------------------------------------------------
// TN products (contraction over rows), one workgroup per (output tile, row split):
//   gram_kernel  lower tiles of  Phi^T diag(w) Phi; its diagonal tiles also form Phi^T y / Phi^T p in
//                fp64 from the rows they stage anyway
//   xtz_kernel   X~^T Zbar with Zbar formed on the fly (ZbarLoader)
// fp32 accumulators are flushed into the workgroup's private fp64 slab every `chunk` rows
// (one fp32 chain stays ~sqrt(chunk)*2^-24); fp64 runs one chunk.
// --------------------------------------------------------------------------
// slab_hi: for 256-row tiles, the slab of rows 128..255 (the two 128 x 128 slabs of a tall tile are not adjacent)
template <class Cfg>
__device__ __forceinline__ void slab_flush(const typename Cfg::MTr::acc_t (&acc)[Cfg::TM][Cfg::TN], double* slab, bool first,
                                           double* slab_hi = nullptr) {
    AccCoord<Cfg> co;
    if (Cfg::BM > 128 && co.wm0 >= 128) slab = slab_hi - 128 * Cfg::BN;          // a wave's rows lie in one half
    // wide tiles (BN > 128): consecutive 128 x 128 slabs, one per 128 output columns; a wave's columns lie in one of them
    constexpr int LDS_ = Cfg::BN > 128 ? 128 : Cfg::BN;
    if (Cfg::BN > 128) slab += (co.wn0 / 128) * (128 * 128) - (co.wn0 / 128) * 128;
#pragma unroll
    for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
        for (int r = 0; r < Cfg::MTr::NACC; ++r) {
            double* d = slab + co.row(tm, r) * LDS_;
#pragma unroll
            for (int tn = 0; tn < Cfg::TN; ++tn) {
                const double v = (double)acc[tm][tn][r];
                d[co.col(tn)] = first ? v : d[co.col(tn)] + v;
            }
        }
}

// Tile grid of the Gram products: `nfull` rows of square tiles (lower triangle) and, when the 64-column blocks of K
// do not pair up, one 64-high STRIP of nfull+1 tiles (64 x 128) below them whose results occupy the upper halves of
// the slabs of tile row nfull.  Diagonal tiles also produce the side vector sum_n s_n Phi[n][col] (s = y: Phi^T y,
// s = p: Phi^T p) for their columns from the rows they stream anyway: sidepart[split][col].
// One launch covers everything; the job order is described at the decode in gram_kernel.
// ZSRC (experiment): Phi points to the phase matrix Z (leading dimension ld = Jp) and the loaders form s cos / s sin
//   S: element type of the operand in memory (T, or float under T = double: the resident fp32 V multiplied in fp64)
// DIAG is a compile-time property of the instantiation (the side sums cost a dozen registers that only the diagonal jobs need:
// with them in every job the 64 x 64 wave tiles spilled inside the k-loop); gram_body dispatches on the job's flag
template <class Cfg, bool WEIGHT, bool STRIP, bool ZSRC, typename S, bool DIAG>
__device__ __forceinline__ void gram_body_impl(
    const S* __restrict__ Phi, int64_t ld, const double* __restrict__ w, const double* __restrict__ side,
    int64_t r0, int64_t r1, int64_t chunk, int acol, int bcol, double* __restrict__ sideout,
    double* __restrict__ slab, double* __restrict__ slab_hi, char* smem_raw, int zJ, typename Cfg::T zs) {
    constexpr bool diag = DIAG;
    typedef typename Cfg::T T;
    T* smem = reinterpret_cast<T*>(smem_raw);
    typename Cfg::MTr::acc_t acc[Cfg::TM][Cfg::TN];
    // consecutive chunks are consecutive k-tiles, so one pair of loaders walks the whole row range
    NatLoader<S, T, Cfg::BM, Cfg::BK, Cfg::LDA, Cfg::THREADS, WEIGHT, false, DIAG, ZSRC> la(
        Phi + r0 * ld + (ZSRC ? 0 : acol), ld, threadIdx.x, WEIGHT ? w + r0 : nullptr, 0, diag ? side + r0 : nullptr);
    NatLoader<S, T, Cfg::BN, Cfg::BK, Cfg::LDB, Cfg::THREADS, false, false, false, ZSRC> lb(Phi + r0 * ld + (ZSRC ? 0 : bcol), ld, threadIdx.x);
    if (ZSRC) { la.z_source(Phi + r0 * ld, ld, zJ, acol, zs); lb.z_source(Phi + r0 * ld, ld, zJ, bcol, zs); }
    bool first = true;
    for (int64_t c0 = r0; c0 < r1 || first; c0 += chunk) {
        const int64_t c1 = c0 + chunk < r1 ? c0 + chunk : r1;
        acc_zero<Cfg>(acc);
        if (c0 < r1) tile_mainloop<Cfg>(la, lb, (int)((c1 - c0) / Cfg::BK), acc, smem);
        slab_flush<Cfg>(acc, slab, first, slab_hi);
        if constexpr (DIAG) la.side_flush();
        first = false;
    }
    if constexpr (STRIP) {                                     // lower half of the 128 x 128 slab(s): rows the strip does not have
        constexpr int NB = Cfg::BN / 128, REST = (128 - Cfg::BM) * 128;
        for (int e = threadIdx.x; e < NB * REST; e += Cfg::THREADS) slab[(e / REST) * (128 * 128) + Cfg::BM * 128 + e % REST] = 0.0;
    }
    if constexpr (DIAG) la.side_reduce(reinterpret_cast<double*>(smem_raw), sideout);
}
template <class Cfg, bool WEIGHT, bool STRIP, bool ZSRC = false, typename S = typename Cfg::T>
__device__ __forceinline__ void gram_body(
    const S* __restrict__ Phi, int64_t ld, const double* __restrict__ w, const double* __restrict__ side,
    int64_t r0, int64_t r1, int64_t chunk, int acol, int bcol, bool diag, double* __restrict__ sideout,
    double* __restrict__ slab, double* __restrict__ slab_hi, char* smem_raw, int zJ = 0, typename Cfg::T zs = 0) {
    if (diag) gram_body_impl<Cfg, WEIGHT, STRIP, ZSRC, S, true>(Phi, ld, w, side, r0, r1, chunk, acol, bcol, sideout, slab, slab_hi, smem_raw, zJ, zs);
    else gram_body_impl<Cfg, WEIGHT, STRIP, ZSRC, S, false>(Phi, ld, w, side, r0, r1, chunk, acol, bcol, sideout, slab, slab_hi, smem_raw, zJ, zs);
}

// Job list of one row split (all kernels of the launch have 8 waves):
//   !BIG  diagonal 128 x 128 tiles, strictly lower tiles row by row, strip tiles
//    BIG  (fp32) pairs of 128-row blocks are covered by 256 x 128 tiles (64 x 64 wave tiles, the shape of the apply
//         product): tile (a, b), b <= 2a, is blocks (2a, b) and (2a+1, b); the diagonal blocks (2a+1, 2a+1), an
//         unpaired last block row and the strip stay 128- / 64-row tiles.  Tall tiles with b == 2a hold the diagonal
//         block of both of their column blocks' rows and carry the side vector for all 256 columns.
// Jobs are split-major and the XCD map hands each XCD a contiguous range of them (whole splits), so the workgroups
// running together on one L2 work on the same rows and share operand panels; inside a split the longest jobs come
// first and the short strip jobs last.
template <bool BIG> __host__ __device__ inline int gram_jobs_per_split(int nfull, int nstrip) {
    if (!BIG) return nfull * (nfull + 1) / 2 + nstrip * (nfull + 1);
    const int R = nfull / 2, odd = nfull & 1, nsb = nstrip * (nfull + 1);
    return R * R + nsb / 4 + R + odd * nfull + nsb % 4;          // tall, wide (4 strip tiles each), small, single strips
}
template <class Cfg, class SCfg, class BCfg, class WCfg, bool WEIGHT, bool BIG, bool ZSRC = false, typename S = typename Cfg::T>
__global__ __launch_bounds__(Cfg::THREADS)
__attribute__((amdgpu_waves_per_eu(4, 4)))       // two 8-wave workgroups per CU: the compiler would take up to 256 VGPRs
void gram_kernel(
    const S* __restrict__ Phi, int64_t ld, const double* __restrict__ w, const double* __restrict__ side,
    RowSplits rs, int64_t chunk, int nfull, int nstrip, double* __restrict__ sidepart, double* __restrict__ slabs,
    int64_t zld, int zJ, double zscale) {                     // ZSRC: Phi is the phase matrix Z (leading dimension zld)
    static_assert(Cfg::THREADS == SCfg::THREADS && Cfg::THREADS == BCfg::THREADS && Cfg::THREADS == WCfg::THREADS &&
                  Cfg::BN == SCfg::BN && Cfg::BN == BCfg::BN && Cfg::BM == Cfg::BN && WCfg::BM == SCfg::BM && WCfg::BN == 4 * Cfg::BN,
                  "one launch, four tile shapes");
    SMEM_DECL;
    TRACE_BEGIN();
    constexpr int B = Cfg::BN;
    const int nall = nfull + nstrip, ntile_all = nall * (nall + 1) / 2;
    const int per_split = gram_jobs_per_split<BIG>(nfull, nstrip);
    const int j = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int split = j / per_split;
    int u = j % per_split, acol, bcol, slab_t, slab_t2 = 0, kind = 0;     // kind 0: 128-row tile, 1: strip, 2: 256-row tile, 3: wide strip
    bool diag = false;
    const auto tri = [](int ti, int tj) { return ti * (ti + 1) / 2 + tj; };
    if (BIG) {
        const int R = nfull / 2, nbig = R * R, nsmall = R + (nfull & 1) * nfull, nwide = nstrip * (nfull + 1) / 4;
        if (u < nbig) {                                        // tall tile (a, b), u = a^2 + b
            int a = (int)sqrtf((float)u);
            while ((a + 1) * (a + 1) <= u) ++a;
            while (a * a > u) --a;
            const int b = u - a * a;
            acol = 2 * a * B; bcol = b * B; slab_t = tri(2 * a, b); slab_t2 = tri(2 * a + 1, b); kind = 2;
            diag = side != nullptr && b == 2 * a;
        } else if (u < nbig + nwide) {                         // wide strip tile: strip tiles 4q .. 4q+3
            const int q = u - nbig;
            acol = nfull * B; bcol = 4 * q * B; slab_t = tri(nfull, 4 * q); kind = 3;
            diag = side != nullptr && nfull >= 4 * q && nfull < 4 * q + 4;
        } else if (u < nbig + nwide + R) {                     // diagonal block of the second row of a pair
            const int i = 2 * (u - nbig - nwide) + 1;
            acol = bcol = i * B; slab_t = tri(i, i);
        } else if (u < nbig + nwide + nsmall) {                // unpaired last block row
            const int i = nfull - 1, b = u - nbig - nwide - R;
            acol = i * B; bcol = b * B; slab_t = tri(i, b); diag = side != nullptr && b == i;
        } else {                                               // the strip tiles that do not fill a wide one
            const int tj = 4 * nwide + u - nbig - nwide - nsmall;
            acol = nfull * B; bcol = tj * B; slab_t = tri(nfull, tj); kind = 1; diag = side != nullptr && tj == nfull;
        }
    } else {
        const int noff = nfull * (nfull - 1) / 2;
        if (u < nfull) {                                       // diagonal tiles (they also carry the side vector)
            acol = bcol = u * B; slab_t = tri(u, u); diag = side != nullptr;
        } else if (u < nfull + noff) {                         // strictly lower tiles: u = (ti-1) ti / 2 + tj
            u -= nfull;
            int tq = (int)((sqrtf(8.0f * u + 1.0f) - 1.0f) * 0.5f);
            while ((tq + 1) * (tq + 2) / 2 <= u) ++tq;
            while (tq * (tq + 1) / 2 > u) --tq;
            const int ti = tq + 1, tj = u - tq * (tq + 1) / 2;
            acol = ti * B; bcol = tj * B; slab_t = tri(ti, tj);
        } else {                                               // strip tiles
            const int tj = u - nfull - noff;
            acol = nfull * B; bcol = tj * B; slab_t = tri(nfull, tj); kind = 1; diag = side != nullptr && tj == nfull;
        }
    }
    int64_t r0, r1;
    rs.range(split, r0, r1);
    double* slab = slabs + ((int64_t)split * ntile_all + slab_t) * (B * B);
    double* slab2 = slabs + ((int64_t)split * ntile_all + slab_t2) * (B * B);
    double* sideout = sidepart + (int64_t)split * ld + acol;
    typedef typename Cfg::T T;
    const int64_t sld = ZSRC ? zld : ld;                       // leading dimension of the operand source
    const T zs = (T)zscale;
    if (kind == 1) { gram_body<SCfg, WEIGHT, true, ZSRC, S>(Phi, sld, w, side, r0, r1, chunk, acol, bcol, diag, sideout, slab, nullptr, smem_raw, zJ, zs); TRACE_END(kind); return; }
    if constexpr (BIG) {
        if (kind == 2) { gram_body<BCfg, WEIGHT, false, ZSRC, S>(Phi, sld, w, side, r0, r1, chunk, acol, bcol, diag, sideout, slab, slab2, smem_raw, zJ, zs); TRACE_END(kind); return; }
        if (kind == 3) { gram_body<WCfg, WEIGHT, true, ZSRC, S>(Phi, sld, w, side, r0, r1, chunk, acol, bcol, diag, sideout, slab, nullptr, smem_raw, zJ, zs); TRACE_END(kind); return; }
    }
    gram_body<Cfg, WEIGHT, false, ZSRC, S>(Phi, sld, w, side, r0, r1, chunk, acol, bcol, diag, sideout, slab, nullptr, smem_raw, zJ, zs);
    TRACE_END(kind);
}

// The same bodies driven by a job table (kernels.h: GramPlan); fp32 job list.
template <class Cfg, class SCfg, class BCfg, class WCfg, bool WEIGHT>
__global__ __launch_bounds__(Cfg::THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
void gram_table_kernel(const typename Cfg::T* __restrict__ Phi, int64_t ld, const double* __restrict__ w, const double* __restrict__ side,
                       const GramJob* __restrict__ jobs, int64_t chunk, int ntile_all, double* __restrict__ sidepart,
                       double* __restrict__ slabs) {
    SMEM_DECL;
    TRACE_BEGIN();
    constexpr int B = Cfg::BN;
    const GramJob jb = jobs[xcd_remap(blockIdx.x, gridDim.x)];
    const int kind = jb.kind;
    const bool diag = side != nullptr && jb.diag;
    double* slab = slabs + ((int64_t)jb.part * ntile_all + jb.tile) * (B * B);
    double* slab2 = slabs + ((int64_t)jb.part * ntile_all + jb.tile2) * (B * B);
    double* sideout = sidepart + (int64_t)jb.part * ld + jb.acol;
    if (kind == 1) gram_body<SCfg, WEIGHT, true>(Phi, ld, w, side, jb.r0, jb.r1, chunk, jb.acol, jb.bcol, diag, sideout, slab, nullptr, smem_raw);
    else if (kind == 2) gram_body<BCfg, WEIGHT, false>(Phi, ld, w, side, jb.r0, jb.r1, chunk, jb.acol, jb.bcol, diag, sideout, slab, slab2, smem_raw);
    else if (kind == 3) gram_body<WCfg, WEIGHT, true>(Phi, ld, w, side, jb.r0, jb.r1, chunk, jb.acol, jb.bcol, diag, sideout, slab, nullptr, smem_raw);
    else gram_body<Cfg, WEIGHT, false>(Phi, ld, w, side, jb.r0, jb.r1, chunk, jb.acol, jb.bcol, diag, sideout, slab, nullptr, smem_raw);
    TRACE_END(kind);
}

// Row tile ti0 + (t / ntn) of the output (tiles are 128 rows apart whatever Cfg::BM is: a narrower Cfg multiplies only the
// first BM rows of its tile and zeroes the rest of the 128 x 128 slab)
//   PLAIN: the B operand is a stored matrix (Phi[n][j], j < J; Pb unused) instead of Zbar formed from Phi and Phibar
template <class Cfg, typename S, bool PLAIN = false>
__global__ __launch_bounds__(Cfg::THREADS) void xtz_kernel(
    const double* __restrict__ Xt, int Dp, const S* __restrict__ Phi, const S* __restrict__ Pb, int64_t ld, int J, int64_t Np,
    int64_t rows_per_split, int64_t chunk, int ntn, int ntile, int ntile_all, int ti0, double* __restrict__ slabs) {
    typedef typename Cfg::T T;
    static_assert(Cfg::BN == 128 && Cfg::BM <= 128, "slabs are 128 x 128");
    SMEM_DECL;
    T* smem = reinterpret_cast<T*>(smem_raw);
    const unsigned wid = xcd_remap(blockIdx.x, gridDim.x);
    const int t = (int)(wid % ntile), split = (int)(wid / ntile);
    const int ti = ti0 + t / ntn, tj = t % ntn;
    const int64_t r0 = (int64_t)split * rows_per_split;
    const int64_t r1 = r0 + rows_per_split < Np ? r0 + rows_per_split : Np;
    double* slab = slabs + ((int64_t)split * ntile_all + ti * ntn + tj) * (128 * 128);
    typename Cfg::MTr::acc_t acc[Cfg::TM][Cfg::TN];
    bool first = true;
    for (int64_t c0 = r0; c0 < r1 || first; c0 += chunk) {
        const int64_t c1 = c0 + chunk < r1 ? c0 + chunk : r1;
        acc_zero<Cfg>(acc);
        if (c0 < r1) {
            NatLoader<double, T, Cfg::BM, Cfg::BK, Cfg::LDA, Cfg::THREADS, false, true> la(
                Xt + c0 * Dp + (int64_t)ti * 128, Dp, threadIdx.x, nullptr, Dp - ti * 128);
            if constexpr (PLAIN) {
                NatLoader<S, T, Cfg::BN, Cfg::BK, Cfg::LDB, Cfg::THREADS, false, true> lb(Phi + c0 * ld + tj * Cfg::BN, ld, threadIdx.x, nullptr,
                                                                                         J - tj * Cfg::BN);
                tile_mainloop<Cfg>(la, lb, (int)((c1 - c0) / Cfg::BK), acc, smem);
            } else {
                ZbarLoader<S, T, Cfg::BN, Cfg::BK, Cfg::LDB, Cfg::THREADS> lb(Phi + c0 * ld, Pb + c0 * ld, ld, J, tj * Cfg::BN, threadIdx.x);
                tile_mainloop<Cfg>(la, lb, (int)((c1 - c0) / Cfg::BK), acc, smem);
            }
        }
        slab_flush<Cfg>(acc, slab, first);
        first = false;
    }
    if (Cfg::BM < 128)
        for (int e = threadIdx.x; e < (128 - Cfg::BM) * 128; e += Cfg::THREADS) slab[Cfg::BM * 128 + e] = 0.0;
}

template <typename T>
int SweepKernels<T>::gram_jobs(const Geom& g) { return gram_jobs_per_split<sizeof(T) == 4>(g.gfull, g.gstrip); }

// Row splits: enough workgroups (>> 512 resident) that faster CUs can take more of them, but long jobs -- at least 5120
// rows per unit (fp32) -- so that the slab traffic (nsplit x K^2/2 x 8 B written and re-read by the reduction) and the
// per-job prologue stay small.  Measured (profiles/r01_tuning.md): the fp32 job list (tall tiles, 89 jobs per split at
// K = 2112) is fastest at 48 units for N = 2.5e5..1e6 and at Np/5120 below that.  From 16 units on they are dealt to the
// 8 XCD groups and the last unit of each group is tapered (kernels.h: RowSplits).
RowSplits gram_row_splits(int jobs, int64_t Np, bool f32, int nsplit_override, int taper) {
    int64_t s = ((f32 ? 4224 : 6144) + jobs - 1) / jobs;
    const int64_t smax = std::max<int64_t>(Np / (f32 ? 5120 : 2048), 1);
    if (s > smax) s = smax;
    // small problems (the job list would leave most of the 512 workgroup slots empty): 64-row granules, as many
    // splits as fill the slots -- each workgroup's k-loop is a chain of dependent fetches, so fewer rows per job
    // is what shortens the launch (Boston shape, 512 rows: 80 -> 25 us)
    int gran = 256;
    if (jobs * s < 256 && Np / 64 > s) { gran = 64; s = std::min<int64_t>(Np / 64, (512 + jobs - 1) / jobs); }
    if (nsplit_override > 0) s = std::min<int64_t>(nsplit_override, Np / gran);
    if (s < 1) s = 1;
    RowSplits rs;
    rs.gran = gran; rs.nrb = Np / gran;
    // the taper pays once a group has at least 5 units; below that its extra slabs cost more in the reduction than the tail
    // option value 1 (default): 1/2, 1/4, 1/8, 1/8; values t >= 2: t + 2 halvings
    if (s >= 16) { rs.groups = 8; rs.units = (int)((s + 4) / 8); rs.taper = taper && rs.units >= 5 ? (taper == 1 ? 3 : taper + 2) : 0; }
    else { rs.groups = 1; rs.units = (int)s; rs.taper = 0; }
    rs.nsplit = rs.groups * rs.per_group();
    return rs;
}

template <typename T>
void SweepKernels<T>::gram(const Geom& g, const T* Phi, const double* w, const double* side, const RowSplits& rs, int64_t chunk,
                           double* slabs, double* sidepart, hipStream_t st, const T* Zsrc, double zscale) {
    typedef typename GramCfg<T, 128>::type Cfg;
    typedef typename GramStripCfg<T>::type SCfg;
    typedef typename GramBigCfg<T>::type BCfg;
    typedef typename GramWideCfg<T>::type WCfg;
    constexpr bool BIG = sizeof(T) == 4;
    const int njobs = gram_jobs(g) * rs.nsplit;
    if (chunk <= 0 || chunk > g.Np) chunk = g.Np;
    chunk = round_up(chunk, 256);                              // splits start and end on 64- or 256-row granules
#ifdef SCFGP_DIAG_PLAIN_W
    w = nullptr;                                               // timing diagnostic only: wrong numbers
#endif
#ifdef SCFGP_DIAG_NOSIDE
    side = nullptr;                                            // timing diagnostic only: wrong numbers
#endif
    constexpr int L1 = Cfg::LDS_BYTES > SCfg::LDS_BYTES ? Cfg::LDS_BYTES : SCfg::LDS_BYTES;
    constexpr int L2 = BIG && BCfg::LDS_BYTES > L1 ? BCfg::LDS_BYTES : L1;
    constexpr int LDS = BIG && WCfg::LDS_BYTES > L2 ? WCfg::LDS_BYTES : L2;
    static_assert(2 * LDS <= 160 * 1024, "two workgroups per CU");
    allow_big_lds(gram_kernel<Cfg, SCfg, BCfg, WCfg, true, BIG>, LDS);
    allow_big_lds(gram_kernel<Cfg, SCfg, BCfg, WCfg, false, BIG>, LDS);
    if (Zsrc && !w) {                                          // experiment: feature map fused into the operand loaders
        allow_big_lds(gram_kernel<Cfg, SCfg, BCfg, WCfg, false, BIG, true>, LDS);
        hipLaunchKernelGGL((gram_kernel<Cfg, SCfg, BCfg, WCfg, false, BIG, true>), dim3(njobs), dim3(Cfg::THREADS), LDS, st,
                           Zsrc, (int64_t)g.Kp, w, side, rs, chunk, g.gfull, g.gstrip, sidepart, slabs, (int64_t)g.Jp, g.J, zscale);
    } else if (w)
        hipLaunchKernelGGL((gram_kernel<Cfg, SCfg, BCfg, WCfg, true, BIG>), dim3(njobs), dim3(Cfg::THREADS), LDS, st,
                           Phi, (int64_t)g.Kp, w, side, rs, chunk, g.gfull, g.gstrip, sidepart, slabs, (int64_t)0, 0, 0.0);
    else
        hipLaunchKernelGGL((gram_kernel<Cfg, SCfg, BCfg, WCfg, false, BIG>), dim3(njobs), dim3(Cfg::THREADS), LDS, st,
                           Phi, (int64_t)g.Kp, w, side, rs, chunk, g.gfull, g.gstrip, sidepart, slabs, (int64_t)0, 0, 0.0);
}

bool gram_lockstep_plan(const Geom& g, int64_t Np, std::vector<GramJob>& jobs, std::vector<int>& cnt, int& nparts) {
    constexpr int B = 128, XCDS = 8, SLOTS = 64;               // 32 CUs x 2 resident workgroups per XCD
    const int nfull = g.gfull, nstrip = g.gstrip, nall = nfull + nstrip;
    const int64_t nrb = Np / 256;
    if (nrb / XCDS < 32) return false;                          // under 8192 rows per XCD the split plan's small-problem rules win
    const auto tri = [](int ti, int tj) { return ti * (ti + 1) / 2 + tj; };
    struct Tile { int acol, bcol, kind, diag, tile, tile2; };
    std::vector<Tile> tall, rest;
    const int R = nfull / 2;
    for (int a = 0; a < R; ++a)
        for (int b = 0; b <= 2 * a; ++b) tall.push_back({2 * a * B, b * B, 2, b == 2 * a, tri(2 * a, b), tri(2 * a + 1, b)});
    const int nwide = nstrip * (nfull + 1) / 4;
    for (int q = 0; q < nwide; ++q) rest.push_back({nfull * B, 4 * q * B, 3, nfull >= 4 * q && nfull < 4 * q + 4, tri(nfull, 4 * q), 0});
    for (int k = 0; k < R; ++k) rest.push_back({(2 * k + 1) * B, (2 * k + 1) * B, 0, 0, tri(2 * k + 1, 2 * k + 1), 0});
    if (nfull & 1)
        for (int b = 0; b < nfull; ++b) rest.push_back({(nfull - 1) * B, b * B, 0, b == nfull - 1, tri(nfull - 1, b), 0});
    for (int tj = 4 * nwide; nstrip && tj <= nfull; ++tj) rest.push_back({nfull * B, tj * B, 1, tj == nfull, tri(nfull, tj), 0});
    // full waves of SLOTS tall tiles sweep the XCD's rows in lock step; what is left over is cut into row slices
    const int nwaves = (int)tall.size() / SLOTS;
    std::vector<Tile> sliced(tall.begin() + nwaves * SLOTS, tall.end());
    sliced.insert(sliced.end(), rest.begin(), rest.end());
    int S = sliced.empty() ? 0 : std::max(1, (3 * SLOTS + (int)sliced.size() - 1) / (int)sliced.size());     // >= 3 rounds of jobs
    const int64_t rows_x = nrb / XCDS * 256;
    while (S > 1 && rows_x / S < 4096) --S;
    nparts = XCDS * std::max(S, 1);
    cnt.assign(nall * (nall + 1) / 2, 0);
    jobs.clear();
    for (int x = 0; x < XCDS; ++x) {
        const int64_t g0 = nrb * x / XCDS * 256, g1 = nrb * (x + 1) / XCDS * 256;
        for (int i = 0; i < nwaves * SLOTS; ++i) {
            const Tile& t = tall[i];
            jobs.push_back({t.acol, t.bcol, t.kind, t.diag, g0, g1, x, t.tile, t.tile2, 0});
        }
        for (int sl = 0; sl < S; ++sl) {
            const int64_t len = (g1 - g0) / 256, r0 = g0 + len * sl / S * 256, r1 = g0 + len * (sl + 1) / S * 256;
            for (const Tile& t : sliced) jobs.push_back({t.acol, t.bcol, t.kind, t.diag, r0, r1, x * S + sl, t.tile, t.tile2, 0});
        }
    }
    for (int i = 0; i < nwaves * SLOTS; ++i) { cnt[tall[i].tile] = XCDS; cnt[tall[i].tile2] = XCDS; }
    for (const Tile& t : sliced) {
        const int n = t.kind == 3 ? 4 : 1;
        for (int k = 0; k < n; ++k) cnt[t.tile + k] = XCDS * S;
        if (t.kind == 2) cnt[t.tile2] = XCDS * S;
    }
    return true;
}

template <typename T>
void SweepKernels<T>::gram_planned(const Geom& g, const T* Phi, const double* w, const double* side, const GramPlan& plan, int64_t chunk,
                                   double* slabs, double* sidepart, hipStream_t st) {
    if constexpr (sizeof(T) == 4) {
        typedef typename GramCfg<T, 128>::type Cfg;
        typedef typename GramStripCfg<T>::type SCfg;
        typedef typename GramBigCfg<T>::type BCfg;
        typedef typename GramWideCfg<T>::type WCfg;
        if (chunk <= 0 || chunk > g.Np) chunk = g.Np;
        chunk = round_up(chunk, 256);
        constexpr int L1 = Cfg::LDS_BYTES > SCfg::LDS_BYTES ? Cfg::LDS_BYTES : SCfg::LDS_BYTES;
        constexpr int L2 = BCfg::LDS_BYTES > L1 ? BCfg::LDS_BYTES : L1;
        constexpr int LDS = WCfg::LDS_BYTES > L2 ? WCfg::LDS_BYTES : L2;
        const int nall = g.gfull + g.gstrip, ntile_all = nall * (nall + 1) / 2;
        if (side) (void)hipMemsetAsync(sidepart, 0, sizeof(double) * (size_t)plan.nparts * g.Kp, st);   // parts a column's diagonal job does not use
        const auto launch = [&](auto kernel) {
            allow_big_lds(kernel, LDS);
            hipLaunchKernelGGL(kernel, dim3(plan.njobs), dim3(Cfg::THREADS), LDS, st, Phi, (int64_t)g.Kp, w, side, plan.jobs, chunk, ntile_all,
                               sidepart, slabs);
        };
        if (w) launch(gram_table_kernel<Cfg, SCfg, BCfg, WCfg, true>);
        else launch(gram_table_kernel<Cfg, SCfg, BCfg, WCfg, false>);
    }
}

// A^T B over the rows, A (Np x Dp, fp64) and B either Zbar formed in the loader (PLAIN false: Bsrc = Phi, Bsrc2 = Phibar) or the
// stored matrix Bsrc (Np x J live columns, leading dimension ldb); slabs: nsplit x (ceil(Dp/128) x ceil(J/128)) tiles
template <typename T, bool PLAIN>
static void tn_product(const double* A, int Dp, const T* Bsrc, const T* Bsrc2, int64_t ldb, int J, int64_t Np, int nsplit, int64_t chunk,
                       double* slabs, hipStream_t st) {
    const int ntm = (Dp + 127) / 128, ntn = (J + 127) / 128;
    const int64_t rps = round_up((Np + nsplit - 1) / nsplit, 64);
    if (chunk <= 0 || chunk > rps) chunk = rps;
    chunk = round_up(chunk, 16);
    const auto launch = [&](auto cfg, int ti0, int nti) {
        typedef typename decltype(cfg)::type Cfg;
        if (nti <= 0) return;
        allow_big_lds(xtz_kernel<Cfg, T, PLAIN>, Cfg::LDS_BYTES);
        hipLaunchKernelGGL((xtz_kernel<Cfg, T, PLAIN>), dim3(nti * ntn * nsplit), dim3(Cfg::THREADS), Cfg::LDS_BYTES, st,
                           A, Dp, Bsrc, Bsrc2, ldb, J, Np, rps, chunk, ntn, nti * ntn, ntm * ntn, ti0, slabs);
    };
    const int last = Dp - 128 * (ntm - 1);
    const int nfull = last > 96 ? ntm : ntm - 1;
    launch(XtzCfg<T>{}, 0, nfull);
    if (nfull < ntm) {
        if (last <= 64) launch(Xtz64Cfg<T>{}, nfull, 1);
        else launch(Xtz96Cfg<T>{}, nfull, 1);
    }
}
template <typename T>
void SweepKernels<T>::tn_plain(const double* A, int Dp, const T* Bm, int64_t ldb, int J, int64_t Np, int nsplit, int64_t chunk, double* slabs,
                               hipStream_t st) {
    tn_product<T, true>(A, Dp, Bm, nullptr, ldb, J, Np, nsplit, chunk, slabs, st);
}

// Zbar[n][j] = Phi[n][j] Phibar[n][J+j] - Phi[n][J+j] Phibar[n][j], j < J, written over Phibar[n][j] (the cosine half of Phibar
// is dead afterwards; its sine half stays until the caller overwrites it)
template <typename T>
__global__ __launch_bounds__(256) void zbar_kernel(const T* __restrict__ Phi, T* Pb, int64_t ld, int J, int64_t Np) {
    const int jv = (J + 3) / 4;                                        // four columns per thread where J allows vector access
    const int64_t total = Np * jv;
    const bool vec = (J % 4 == 0) && sizeof(T) == 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t n = i / jv; const int j0 = (int)(i - n * jv) * 4;
        const T* f = Phi + n * ld; T* b = Pb + n * ld;
        if (vec) {
            typedef T t4 __attribute__((ext_vector_type(4)));
            const t4 fc = *reinterpret_cast<const t4*>(f + j0), fs = *reinterpret_cast<const t4*>(f + J + j0);
            const t4 bc = *reinterpret_cast<const t4*>(b + j0), bs = *reinterpret_cast<const t4*>(b + J + j0);
            *reinterpret_cast<t4*>(b + j0) = fc * bs - fs * bc;
        } else {
            for (int e = 0; e < 4 && j0 + e < J; ++e) { const int j = j0 + e; b[j] = f[j] * b[J + j] - f[J + j] * b[j]; }
        }
    }
}
template <typename T>
void SweepKernels<T>::zbar_inplace(const Geom& g, const T* Phi, T* Phibar, hipStream_t st) {
    hipLaunchKernelGGL(zbar_kernel<T>, dim3(8192), dim3(256), 0, st, Phi, Phibar, (int64_t)g.Kp, g.J, g.Np);
}
// Rsel (typed, leading dimension Kp, Kp rows): rows j < S the identity, rows S + m the row m of r_F, zero elsewhere -- the operand of
// U = Zbar . Rsel = Zbar_L + Zbar_M r_F
template <typename T>
__global__ void rsel_kernel(const double* __restrict__ params, int D, int S, int M, T* __restrict__ out, int Kp, int ncol) {
    const int64_t total = (int64_t)Kp * ncol;
    const double* rF = params + 3 + (int64_t)D * S;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int k = (int)(i / ncol), c = (int)(i % ncol);
        T v = 0;
        if (c < S) { if (k < S) v = k == c ? (T)1 : (T)0; else if (k < S + M) v = (T)rF[(int64_t)(k - S) * S + c]; }
        out[(int64_t)k * Kp + c] = v;
    }
}
template <typename T>
void SweepKernels<T>::rsel(const Geom& g, const double* params, T* out, hipStream_t st) {
    hipLaunchKernelGGL(rsel_kernel<T>, dim3(1024), dim3(256), 0, st, params, g.D, g.S, g.M, out, g.Kp, (int)round_up(g.S, 64));
}
template <typename T>
void SweepKernels<T>::xtz(const Geom& g, const double* Xt, const T* Phi, const T* Phibar, int nsplit, int64_t chunk, double* slabs,
                          hipStream_t st) {
    tn_product<T, false>(Xt, g.Dp, Phi, Phibar, (int64_t)g.Kp, g.J, g.Np, nsplit, chunk, slabs, st);
}

// --------------------------------------------------------------------------
// NT products (contraction over feature columns):  C = Phi . Bm  with Bm symmetric
//   EPI 0: V = C,  vpart[jt][n] = sum_j Phi[n][j] C[n][j]
//   EPI 1: Phibar = 2 C + 2 q_n V[n][j] + p_n alpha_j + y_n ut_j   (in place over V)
//   EPI 2 (predict): Bm = Li^T, so C = Phi Li^T is the reference's own product (SCFGP/SCFGP.py:144) and
//          vpart[jt][n] = sum_j C[n][j]^2; nothing is stored, and since Li^T[k][j] = 0 for k > j the contraction of column
//          tile jt stops at its last column: half the flops of the symmetric product
//   EPI 3 (factor form of pass 2, SCFGP/SCFGP.py:112): as EPI 2 but C is stored (in V's place) and mu rides along
//   EPI 4 (factor form): V = C . Li, Bm = Li lower triangular (Bm[k][j] = 0 for k < j): the contraction of column tile jt
//          STARTS at its first column; plain store, no row sums
// --------------------------------------------------------------------------
// epilogues of the apply product: V and the row dots (EPI 0) or Phibar and bbar (EPI 1) from the accumulators
//   MU (EPI 0 only): also mupart[jtg][n] = sum_{j in tile} Phi[n][j] alpha[j] from the Phi values the row dot reads anyway
//   (the DMA-fed kernel has no operand values in registers for the loader-side dot)
//   VEC4 (the LDS-DMA kernels, fp32, 64-wide wave tiles of four 16-column MFMA tiles): the B operand's rows were staged in a
//   permuted order, so that MFMA tile tn, lane column i IS output column 4 i + tn of the wave tile -- a lane then holds four
//   ADJACENT columns of each of its rows and the epilogue moves V, Phi and Phibar 16 bytes per lane (256 contiguous bytes per
//   row and 16-lane group) instead of 4 (four 64-byte pieces per instruction)
// Sum over the 16 lanes of a DPP row (the lanes that hold one output row of a 16 x 16 MFMA tile), result in every lane: two
// quad permutes, row_half_mirror, row_mirror on the two halves of the double -- VALU moves instead of the eight LDS-crossbar
// ds_bpermute_b32 a __shfl_xor butterfly costs (256 of them per lane in the 64 x 64 wave tile's epilogue: 0.9 ms of the product)
template <int CTRL> __device__ __forceinline__ double dpp_mov_f64(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double x) {
    x += dpp_mov_f64<0xB1>(x);                                  // quad_perm [1,0,3,2]
    x += dpp_mov_f64<0x4E>(x);                                  // quad_perm [2,3,0,1]
    x += dpp_mov_f64<0x141>(x);                                 // row_half_mirror
    x += dpp_mov_f64<0x140>(x);                                 // row_mirror
    return x;
}
template <class Cfg, int EPI, bool MU = false, bool VEC4 = false>
__device__ __forceinline__ void apply_epilogue(
    const typename Cfg::MTr::acc_t (&acc)[Cfg::TM][Cfg::TN], const typename Cfg::T* __restrict__ Phi, typename Cfg::T* V,
    double* __restrict__ vpart, const double* __restrict__ p, const double* __restrict__ q, const double* __restrict__ y,
    const double* __restrict__ alpha, const double* __restrict__ ut, int K, int Kp, int64_t Np, int64_t rb, int cbase, int jtg,
    double* __restrict__ bpart, char* smem_raw, double* __restrict__ mupart = nullptr) {
    typedef typename Cfg::T T;
    AccCoord<Cfg> co;
    if constexpr (VEC4) {
        static_assert(Cfg::TN == 4 && Cfg::MS == 16 && sizeof(T) == 4 && (EPI == 0 || EPI == 1 || EPI == 3 || EPI == 4), "VEC4 layout");
        const int c4 = co.wn0 + 4 * (co.lane & 15);               // first of this lane's four adjacent columns
        const int jg = cbase + c4;
        if constexpr (EPI == 4) {
#pragma unroll
            for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    *reinterpret_cast<v4f*>(V + (rb * Cfg::BM + co.row(tm, r)) * Kp + jg) = v4f{acc[tm][0][r], acc[tm][1][r], acc[tm][2][r], acc[tm][3][r]};
        } else if constexpr (EPI == 0 || EPI == 3) {
            double* red = reinterpret_cast<double*>(smem_raw);
            double* red2 = red + Cfg::WGN * Cfg::BM;
            const int wn = (threadIdx.x >> 6) % Cfg::WGN;
            double al[4], live[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { live[k] = jg + k < K ? 1.0 : 0.0; al[k] = MU && jg + k < K ? alpha[jg + k] : 0.0; }
            // the re-read of Phi is issued for the four rows of an accumulator row group at once: written row by row, every row's
            // load waited for the row before it (16 dependent round trips per tile; profiles/r03_tuning.md)
#pragma unroll
            for (int tm = 0; tm < Cfg::TM; ++tm) {
                v4f ph[4];
                int64_t off[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    off[r] = (rb * Cfg::BM + co.row(tm, r)) * Kp + jg;
#if SCFGP_DIAG_EPI0 < 2                                              // timing diagnostics (wrong numbers): 1 no row reduction, 2 stores only, 3 nothing
                    if (EPI == 0 || MU) ph[r] = *reinterpret_cast<const v4f*>(Phi + off[r]);
#endif
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = co.row(tm, r);
                    const v4f c = v4f{acc[tm][0][r], acc[tm][1][r], acc[tm][2][r], acc[tm][3][r]};
#if SCFGP_DIAG_EPI0 != 3
                    *reinterpret_cast<v4f*>(V + off[r]) = c;
#endif
#if SCFGP_DIAG_EPI0 >= 2
                    asm volatile("" :: "v"(c));
                    continue;
#endif
                    double part = 0, mup = 0;
                    if (EPI == 3) {
                        part = (double)(c[0] * c[0]) + (double)(c[1] * c[1]) + (double)(c[2] * c[2]) + (double)(c[3] * c[3]);
                        if (MU) mup = (double)ph[r][0] * al[0] + (double)ph[r][1] * al[1] + (double)ph[r][2] * al[2] + (double)ph[r][3] * al[3];
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) { part += (double)ph[r][k] * (double)c[k] * live[k]; if (MU) mup += (double)ph[r][k] * al[k]; }
                    }
#if SCFGP_DIAG_EPI0 == 1
                    asm volatile("" :: "v"(part), "v"(mup));
                    continue;
#endif
                    part = row16_sum(part);
                    if ((co.lane & 15) == 0) red[wn * Cfg::BM + row] = part;
                    if (MU) {
                        mup = row16_sum(mup);
                        if ((co.lane & 15) == 0) red2[wn * Cfg::BM + row] = mup;
                    }
                }
            }
            __syncthreads();
            if (threadIdx.x < Cfg::BM) {
                double s = 0, s2 = 0;
#pragma unroll
                for (int k = 0; k < Cfg::WGN; ++k) { s += red[k * Cfg::BM + threadIdx.x]; if (MU) s2 += red2[k * Cfg::BM + threadIdx.x]; }
                vpart[(int64_t)jtg * Np + rb * Cfg::BM + threadIdx.x] = s;
                if (MU && mupart) mupart[(int64_t)jtg * Np + rb * Cfg::BM + threadIdx.x] = s2;
            }
        } else {                                                    // EPI 1
            double bb = 0;
            double al[4], u4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { al[k] = alpha[jg + k]; u4[k] = ut[jg + k]; }
            float alf[4], u4f[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { alf[k] = (float)al[k]; u4f[k] = (float)u4[k]; }
            // per-row scalars (2 q, p, y) of the tile's rows through LDS (free after the loop's last barrier), and the re-reads of V
            // and Phi issued for the four rows of an accumulator row group at once: written row by row, the in-place store of a row
            // stood between the loads of the next one and its own (16 dependent round trips per tile; profiles/r03_tuning.md)
            float* rowsc = reinterpret_cast<float*>(smem_raw);        // [BM][3]
            for (int i = threadIdx.x; i < Cfg::BM; i += Cfg::THREADS) {
                const int64_t n = rb * Cfg::BM + i;
                rowsc[3 * i] = (float)(2.0 * q[n]); rowsc[3 * i + 1] = (float)p[n]; rowsc[3 * i + 2] = (float)y[n];
            }
            __syncthreads();
#pragma unroll
            for (int tm = 0; tm < Cfg::TM; ++tm) {
                v4f vv[4], ph[4];
                int64_t off[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    off[r] = (rb * Cfg::BM + co.row(tm, r)) * Kp + jg;
#if SCFGP_DIAG_EPI1 >= 2                                             // timing diagnostics (wrong numbers): 1 no Phi re-read / b-bar dot, 2 no V re-read either, 3 nothing
                    vv[r] = v4f{0, 0, 0, 0};
#else
                    vv[r] = *reinterpret_cast<const v4f*>(V + off[r]);
#endif
#if SCFGP_DIAG_EPI1 >= 1
                    ph[r] = v4f{0, 0, 0, 0};
#else
                    ph[r] = *reinterpret_cast<const v4f*>(Phi + off[r]);
#endif
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#if SCFGP_DIAG_EPI1 == 3
                    asm volatile("" :: "v"(acc[tm][0][r]), "v"(acc[tm][1][r]), "v"(acc[tm][2][r]), "v"(acc[tm][3][r]));
                    continue;
#endif
                    const int row = co.row(tm, r);
                    // Phibar is stored in fp32: its four terms are combined in fp32 FMAs (one rounding per term instead of one at
                    // the end; the accumulator itself carries ~1e-7 of the product), and the b-bar dot takes the four products of a
                    // lane in fp32 and everything above them in fp64 -- the fp64 conversions and FMAs of the all-fp64 form were
                    // ~1 ms of the launch (profiles/r03_tuning.md)
                    const float qn = rowsc[3 * row], pn = rowsc[3 * row + 1], yn = rowsc[3 * row + 2];
                    v4f o;
                    float dot = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        o[k] = fmaf(qn, vv[r][k], fmaf(pn, alf[k], fmaf(yn, u4f[k], 2.0f * acc[tm][k][r])));
#if SCFGP_DIAG_EPI1 == 0
                        dot = fmaf(o[k], jg + k < K ? ph[r][k] : 0.f, dot);
#endif
                    }
                    bb += (double)dot;
                    *reinterpret_cast<v4f*>(V + off[r]) = o;
                }
            }
            double* red = reinterpret_cast<double*>(rowsc + 4 * Cfg::BM);
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) bb += __shfl_xor(bb, m);
            if (co.lane == 0) red[threadIdx.x >> 6] = bb;
            __syncthreads();
            if (threadIdx.x == 0) {
                double s = 0;
                for (int k = 0; k < Cfg::THREADS / 64; ++k) s += red[k];
                bpart[blockIdx.x] = s;
            }
        }
        return;
    }
    if (EPI == 4) {
#pragma unroll
        for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
            for (int r = 0; r < Cfg::MTr::NACC; ++r) {
                const int64_t off = (rb * Cfg::BM + co.row(tm, r)) * Kp + cbase;
#pragma unroll
                for (int tn = 0; tn < Cfg::TN; ++tn) V[off + co.col(tn)] = acc[tm][tn][r];
            }
    } else if (EPI == 0 || EPI == 2 || EPI == 3) {
        double* red = reinterpret_cast<double*>(smem_raw);          // [WGN][BM] (MU: twice); main loop ended with a barrier
        double* red2 = red + Cfg::WGN * Cfg::BM;
        const int wn = (threadIdx.x >> 6) % Cfg::WGN;
        double al[Cfg::TN];
        if (MU) {
#pragma unroll
            for (int tn = 0; tn < Cfg::TN; ++tn) al[tn] = cbase + co.col(tn) < K ? alpha[cbase + co.col(tn)] : 0.0;
        }
#pragma unroll
        for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
            for (int r = 0; r < Cfg::MTr::NACC; ++r) {
                const int row = co.row(tm, r);
                const int64_t off = (rb * Cfg::BM + row) * Kp + cbase;
                double part = 0, mup = 0;
#pragma unroll
                for (int tn = 0; tn < Cfg::TN; ++tn) {
                    const T c = acc[tm][tn][r];
                    if (EPI != 2) V[off + co.col(tn)] = c;
                    if (EPI != 0) {                                   // v_n = || Li phi_n ||^2
                        part += (double)c * (double)c;
                        if (MU && cbase + co.col(tn) < K) mup += (double)Phi[off + co.col(tn)] * al[tn];
                    } else if (cbase + co.col(tn) < K) {              // v_n = phi_n . (B phi_n)
                        const double ph = (double)Phi[off + co.col(tn)];
                        part += ph * (double)c;
                        if (MU) mup += ph * al[tn];
                    }
                }
                if constexpr (Cfg::MS == 16) part = row16_sum(part);                     // lanes of one MFMA row group
                else {
#pragma unroll
                    for (int m = 1; m < Cfg::MS; m <<= 1) part += __shfl_xor(part, m);
                }
                if ((co.lane % Cfg::MS) == 0) red[wn * Cfg::BM + row] = part;
                if (MU) {
                    if constexpr (Cfg::MS == 16) mup = row16_sum(mup);
                    else {
#pragma unroll
                        for (int m = 1; m < Cfg::MS; m <<= 1) mup += __shfl_xor(mup, m);
                    }
                    if ((co.lane % Cfg::MS) == 0) red2[wn * Cfg::BM + row] = mup;
                }
            }
        __syncthreads();
        if (threadIdx.x < Cfg::BM) {
            double s = 0, s2 = 0;
#pragma unroll
            for (int k = 0; k < Cfg::WGN; ++k) { s += red[k * Cfg::BM + threadIdx.x]; if (MU) s2 += red2[k * Cfg::BM + threadIdx.x]; }
            vpart[(int64_t)jtg * Np + rb * Cfg::BM + threadIdx.x] = s;
            if (MU && mupart) mupart[(int64_t)jtg * Np + rb * Cfg::BM + threadIdx.x] = s2;
        }
    } else {
        double bb = 0;                                                  // bbar = sum Phibar o Phi  (d cost / d b)
        [[maybe_unused]] float alf[Cfg::TN], utf[Cfg::TN];
        if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int tn = 0; tn < Cfg::TN; ++tn) { alf[tn] = (float)alpha[cbase + co.col(tn)]; utf[tn] = (float)ut[cbase + co.col(tn)]; }
        }
#pragma unroll
        for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
            for (int r = 0; r < Cfg::MTr::NACC; ++r) {
                const int64_t n = rb * Cfg::BM + co.row(tm, r);
                const int64_t off = n * Kp + cbase;
                const double qn = 2.0 * q[n], pn = p[n], yn = y[n];
                if constexpr (sizeof(T) == 4) {                     // fp32 storage: fp32 FMAs, as in the VEC4 path above
                    const float qf = (float)qn, pf = (float)pn, yf = (float)yn;
                    float dot = 0.f;
#pragma unroll
                    for (int tn = 0; tn < Cfg::TN; ++tn) {
                        const float o = fmaf(qf, V[off + co.col(tn)], fmaf(pf, alf[tn], fmaf(yf, utf[tn], 2.0f * acc[tm][tn][r])));
                        V[off + co.col(tn)] = o;
                        if (cbase + co.col(tn) < K) dot = fmaf(o, Phi[off + co.col(tn)], dot);
                    }
                    bb += (double)dot;
                } else {
#pragma unroll
                    for (int tn = 0; tn < Cfg::TN; ++tn) {
                        const int j = cbase + co.col(tn);
                        const double v = 2.0 * (double)acc[tm][tn][r] + qn * (double)V[off + co.col(tn)] + pn * alpha[j] + yn * ut[j];
                        V[off + co.col(tn)] = (T)v;
                        if (j < K) bb += v * (double)Phi[off + co.col(tn)];
                    }
                }
            }
        double* red = reinterpret_cast<double*>(smem_raw);
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) bb += __shfl_xor(bb, m);
        if (co.lane == 0) red[threadIdx.x >> 6] = bb;
        __syncthreads();
        if (threadIdx.x == 0) {
            double s = 0;
            for (int k = 0; k < Cfg::THREADS / 64; ++k) s += red[k];
            bpart[blockIdx.x] = s;
        }
    }
}
typedef float v2f __attribute__((ext_vector_type(2)));
// --------------------------------------------------------------------------
// EPI 5: the Phibar product writes Zbar instead of Phibar ("pass 3 without the Phibar round trip").  The column tiles of the launch
// are tiles of j: a tile holds the cosine columns j and the sine columns J + j of the same j's, gathered through the operand's row
// addresses, so that one lane ends up with Phibar[n][j] AND Phibar[n][J + j]:
//   Zbar[n][j] = Phi[n][j] Phibar[n][J + j] - Phi[n][J + j] Phibar[n][j]          (zbar_kernel), written over V[n][j], j < J
// Phibar itself never reaches memory (8.45 GB less written, Phi and Phibar not re-read by a Zbar pass or by the X~^T Zbar loader);
// bbar = sum Phibar o Phi as in EPI 1.  fp32 storage only, J % 4 == 0.
//   VEC4 form (LDS-DMA kernels): MFMA tile tn of a 64-wide wave tile w, lane column i: tn = 0, 1 -> cosine column jb + 32 w + 2 i + tn,
//   tn = 2, 3 -> the sine column of the same j
// --------------------------------------------------------------------------
template <class Cfg>
__device__ __forceinline__ void zbar_epilogue_vec4(
    const typename Cfg::MTr::acc_t (&acc)[Cfg::TM][Cfg::TN], const float* __restrict__ Phi, float* V,
    const double* __restrict__ p, const double* __restrict__ q, const double* __restrict__ y,
    const double* __restrict__ alpha, const double* __restrict__ ut, int J, int Kp, int64_t rb, int jb,
    double* __restrict__ bpart, char* smem_raw) {
    static_assert(Cfg::TN == 4 && Cfg::MS == 16, "64-wide wave tiles");
    AccCoord<Cfg> co;
    const int j = jb + (co.wn0 >> 1) + 2 * (co.lane & 15);      // this lane's two j: j, j + 1 (full tiles only: j + 1 < J)
    float alc[2], als[2], uc[2], us[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        alc[t] = (float)alpha[j + t]; als[t] = (float)alpha[J + j + t];
        uc[t] = (float)ut[j + t]; us[t] = (float)ut[J + j + t];
    }
    float* rowsc = reinterpret_cast<float*>(smem_raw);          // [BM][3]: 2 q, p, y of the tile's rows (LDS is free after the loop)
    for (int i = threadIdx.x; i < Cfg::BM; i += Cfg::THREADS) {
        const int64_t n = rb * Cfg::BM + i;
        rowsc[3 * i] = (float)(2.0 * q[n]); rowsc[3 * i + 1] = (float)p[n]; rowsc[3 * i + 2] = (float)y[n];
    }
    __syncthreads();
    double bb = 0;
#pragma unroll
    for (int tm = 0; tm < Cfg::TM; ++tm) {
        v2f vc[4], vs[4], fc[4], fs[4];
        int64_t off[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            off[r] = (rb * Cfg::BM + co.row(tm, r)) * Kp + j;
            vc[r] = *reinterpret_cast<const v2f*>(V + off[r]); vs[r] = *reinterpret_cast<const v2f*>(V + off[r] + J);
            fc[r] = *reinterpret_cast<const v2f*>(Phi + off[r]); fs[r] = *reinterpret_cast<const v2f*>(Phi + off[r] + J);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = co.row(tm, r);
            const float qn = rowsc[3 * row], pn = rowsc[3 * row + 1], yn = rowsc[3 * row + 2];
            v2f z;
            float dot = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float bc = fmaf(qn, vc[r][t], fmaf(pn, alc[t], fmaf(yn, uc[t], 2.0f * acc[tm][t][r])));
                const float bs = fmaf(qn, vs[r][t], fmaf(pn, als[t], fmaf(yn, us[t], 2.0f * acc[tm][2 + t][r])));
                dot = fmaf(bc, fc[r][t], fmaf(bs, fs[r][t], dot));
                z[t] = fc[r][t] * bs - fs[r][t] * bc;
            }
            bb += (double)dot;
            *reinterpret_cast<v2f*>(V + off[r]) = z;
        }
    }
    double* red = reinterpret_cast<double*>(rowsc + 4 * Cfg::BM);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) bb += __shfl_xor(bb, m);
    if (co.lane == 0) red[threadIdx.x >> 6] = bb;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0;
        for (int k = 0; k < Cfg::THREADS / 64; ++k) s += red[k];
        bpart[blockIdx.x] = s;
    }
}
//   pair form (the loader-staged 64-wide remainder tile, 32-wide wave tiles of two MFMA tiles): wave column w, lane column i:
//   tn = 0 -> cosine column jb + 16 w + i, tn = 1 -> its sine column; j >= J (ragged last tile) contributes nothing
template <class Cfg>
__device__ __forceinline__ void zbar_epilogue_pair(
    const typename Cfg::MTr::acc_t (&acc)[Cfg::TM][Cfg::TN], const float* __restrict__ Phi, float* V,
    const double* __restrict__ p, const double* __restrict__ q, const double* __restrict__ y,
    const double* __restrict__ alpha, const double* __restrict__ ut, int J, int Kp, int64_t rb, int jb,
    double* __restrict__ bpart, char* smem_raw) {
    static_assert(Cfg::TN == 2 && Cfg::MS == 16, "32-wide wave tiles");
    AccCoord<Cfg> co;
    const int j = jb + (co.wn0 >> 1) + (co.lane & 15);
    const bool live = j < J;
    const float alc = live ? (float)alpha[j] : 0.f, als = live ? (float)alpha[J + j] : 0.f;
    const float uc = live ? (float)ut[j] : 0.f, us = live ? (float)ut[J + j] : 0.f;
    double bb = 0;
#pragma unroll
    for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
        for (int r = 0; r < Cfg::MTr::NACC; ++r) {
            const int64_t n = rb * Cfg::BM + co.row(tm, r);
            const int64_t off = n * Kp + j;
            if (live) {
                const float qn = (float)(2.0 * q[n]), pn = (float)p[n], yn = (float)y[n];
                const float fc = Phi[off], fs = Phi[off + J];
                const float bc = fmaf(qn, V[off], fmaf(pn, alc, fmaf(yn, uc, 2.0f * acc[tm][0][r])));
                const float bs = fmaf(qn, V[off + J], fmaf(pn, als, fmaf(yn, us, 2.0f * acc[tm][1][r])));
                bb += (double)fmaf(bc, fc, bs * fs);
                V[off] = fc * bs - fs * bc;
            }
        }
    double* red = reinterpret_cast<double*>(smem_raw);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) bb += __shfl_xor(bb, m);
    if (co.lane == 0) red[threadIdx.x >> 6] = bb;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0;
        for (int k = 0; k < Cfg::THREADS / 64; ++k) s += red[k];
        bpart[blockIdx.x] = s;
    }
}
#ifndef SCFGP_BF3_WAVES
#define SCFGP_BF3_WAVES 2        // waves per SIMD the split-precision apply kernel is compiled for (2: one workgroup per CU)
#endif
// one output tile (column tile jt of this launch, row block rb) of the apply product
template <class Cfg, int EPI>
__device__ __forceinline__ void apply_tile(
    const typename Cfg::T* __restrict__ Phi, const typename Cfg::T* __restrict__ Bm, typename Cfg::T* V,
    double* __restrict__ vpart, const double* __restrict__ p, const double* __restrict__ q, const double* __restrict__ y,
    const double* __restrict__ alpha, const double* __restrict__ ut, int K, int Kp, int64_t Np, int njt,
    double* __restrict__ bpart, int col0, int jt0, double* __restrict__ mu, int ntot, int jt, int64_t rb, char* smem_raw) {
    typedef typename Cfg::T T;
    T* smem = reinterpret_cast<T*>(smem_raw);
    const int cbase = col0 + jt * Cfg::BN;
    // EPI 0: column tile t of ntot also forms the slice kt % ntot == t of mu = Phi.alpha for its rows
    const bool want_mu = (EPI == 0 || EPI == 2 || EPI == 3) && mu != nullptr;
    // rows >= K of the operand matrix are zero padding, so the contraction stops at K rounded up to the k-tile;
    // EPI 2: the operand is lower-triangular-transposed, column tile jt needs k < cbase + BN only, and forms the slice
    // k in [cbase, cbase + BN) of mu (the k-tiles no earlier column tile visits)
    const int nkt_all = (K + Cfg::BK - 1) / Cfg::BK;
    const int nkt_tri = (cbase + Cfg::BN + Cfg::BK - 1) / Cfg::BK;
    const int kt0 = EPI == 4 ? cbase / Cfg::BK : 0;                 // EPI 4: Bm[k][j] = 0 for k < j
    const int nkt = ((EPI == 2 || EPI == 3) && nkt_tri < nkt_all ? nkt_tri : nkt_all) - kt0;
    if constexpr (IsBf3<Cfg>::value) {                         // split-precision tiles; Bm: the matrix pre-split by bf3_presplit()
        typename Cfg::MTr::acc_t acc[Cfg::TM][Cfg::TN];
        acc_zero<Cfg>(acc);
        Bf3TrLoader<Cfg::BM, Cfg::THREADS, EPI != 1> la(Phi + rb * Cfg::BM * Kp, Kp, threadIdx.x, want_mu ? alpha : nullptr, jt0 + jt, ntot);
        if (EPI == 2) la.dot_range(cbase / Cfg::BK, nkt);
        else if (ntot == 0) {                                   // beside DMA-fed tiles: mu slices are the tiles' own column bands
            const int hi = (cbase + Cfg::BN) / Cfg::BK;
            la.dot_range(cbase / Cfg::BK, hi < nkt ? hi : nkt);
        }
        Bf3CopyLoader<Cfg::BN, Cfg::THREADS> lb(Bm, Kp, cbase, threadIdx.x);
        bf3_mainloop<Cfg>(la, lb, nkt, acc, smem_raw);
        if (EPI != 1 && want_mu) la.dot_reduce(mu + (int64_t)(jt0 + jt) * Np + rb * Cfg::BM);
        apply_epilogue<Cfg, EPI>(acc, Phi, V, vpart, p, q, y, alpha, ut, K, Kp, Np, rb, cbase, jt0 + jt, bpart, smem_raw);
    } else {
        typename Cfg::MTr::acc_t acc[Cfg::TM][Cfg::TN];
        acc_zero<Cfg>(acc);
        TrLoader<T, T, Cfg::BM, Cfg::BK, Cfg::LDA, Cfg::THREADS, EPI != 1 && EPI != 4 && EPI != 5, Cfg::SWZA> la(
            Phi + rb * Cfg::BM * Kp + kt0 * Cfg::BK, Kp, threadIdx.x, want_mu ? alpha : nullptr, jt0 + jt, ntot);
        if (EPI == 2 || EPI == 3) la.dot_range(cbase / Cfg::BK, nkt);
        else if (ntot == 0) {                                   // beside DMA-fed tiles: mu slices are the tiles' own column bands
            const int hi = (cbase + Cfg::BN) / Cfg::BK;
            la.dot_range(cbase / Cfg::BK, hi < nkt ? hi : nkt);
        }
        NatLoader<T, T, Cfg::BN, Cfg::BK, Cfg::LDB, Cfg::THREADS, false, false> lb(Bm + (int64_t)kt0 * Cfg::BK * Kp + cbase, Kp, threadIdx.x);
        if constexpr (EPI == 5) {
            // col0 counts j here: tile jt covers j in [jb, jb + BN / 2); tile column x = 32 w + 16 tn + i is operand column
            // jb + 16 w + i (tn = 0, cosine) or J + that (tn = 1, sine); j >= J: a zero padding column (K < Kp whenever a tile is ragged)
            const int J = K / 2, jb = col0 + jt * (Cfg::BN / 2);
            lb.remap_columns(Bm, Kp, [=](int x) {
                const int jj = jb + 16 * (x >> 5) + (x & 15);
                return jj < J ? ((x & 16) ? J + jj : jj) : K + (x & 3);
            });
            tile_mainloop<Cfg>(la, lb, nkt, acc, smem);
            if constexpr (sizeof(T) == 4 && Cfg::TN == 2) zbar_epilogue_pair<Cfg>(acc, Phi, V, p, q, y, alpha, ut, J, Kp, rb, jb, bpart, smem_raw);
            return;
        }
        tile_mainloop<Cfg>(la, lb, nkt, acc, smem);
        if constexpr (EPI != 1 && EPI != 4 && EPI != 5) { if (want_mu) la.dot_reduce(mu + (int64_t)(jt0 + jt) * Np + rb * Cfg::BM); }
        apply_epilogue<Cfg, EPI>(acc, Phi, V, vpart, p, q, y, alpha, ut, K, Kp, Np, rb, cbase, jt0 + jt, bpart, smem_raw);
    }
}

template <class Cfg, int EPI>
__global__ __launch_bounds__(Cfg::THREADS)
__attribute__((amdgpu_waves_per_eu(IsBf3<Cfg>::value ? SCFGP_BF3_WAVES : (sizeof(typename Cfg::T) == 4 && SCFGP_APPLY_WAVES_F32 != 8 ? SCFGP_APPLY_WAVES_F32 : 1),
                                   IsBf3<Cfg>::value ? SCFGP_BF3_WAVES : (sizeof(typename Cfg::T) == 4 ? SCFGP_APPLY_WAVES_F32 : 8))))
void apply_kernel(
    const typename Cfg::T* __restrict__ Phi, const typename Cfg::T* __restrict__ Bm, typename Cfg::T* V,
    double* __restrict__ vpart, const double* __restrict__ p, const double* __restrict__ q, const double* __restrict__ y,
    const double* __restrict__ alpha, const double* __restrict__ ut, int K, int Kp, int64_t Np, int njt,
    double* __restrict__ bpart, int col0, int jt0, double* __restrict__ mu, int ntot) {
    // this launch covers columns [col0, col0 + njt*BN); jt0 = index of its first tile in vpart
    SMEM_DECL;
    const unsigned wid = xcd_remap(blockIdx.x, gridDim.x);
    if constexpr (EPI == 3 || EPI == 4) {
        // triangular operand: column tile jt contracts over (jt + 1) / njt (EPI 3) or (njt - jt) / njt (EPI 4) of the k range, so a
        // workgroup takes tile t AND tile njt - 1 - t -- every workgroup of the launch then does the same amount of work
        const int np = (njt + 1) / 2, t = (int)(wid % np), o = njt - 1 - t;
        const int64_t rb = wid / np;
        apply_tile<Cfg, EPI>(Phi, Bm, V, vpart, p, q, y, alpha, ut, K, Kp, Np, njt, bpart, col0, jt0, mu, ntot, t, rb, smem_raw);
        if (o != t) {
            __syncthreads();                                     // the first tile's epilogue is done with the LDS
            apply_tile<Cfg, EPI>(Phi, Bm, V, vpart, p, q, y, alpha, ut, K, Kp, Np, njt, bpart, col0, jt0, mu, ntot, o, rb, smem_raw);
        }
    } else
        apply_tile<Cfg, EPI>(Phi, Bm, V, vpart, p, q, y, alpha, ut, K, Kp, Np, njt, bpart, col0, jt0, mu, ntot, (int)(wid % njt), wid / njt, smem_raw);
}

// fp32 apply product with LDS-DMA staging (option apply_dma): both operands go global -> LDS by global_load_lds_dwordx4
// into a ring of three 24 KB stages (16 k of the 256 x 128 tile), counted vmcnt, one raw barrier per stage: no staging
// registers, no ds_write, and the fetch of stage s+2 is in flight while stage s is multiplied.  Bm must be SYMMETRIC
// (B and Abar are): row j of it supplies the k-contiguous column j.
//   LDS image of an operand: row x at x * 64 B = its 16 k as four 16-byte chunks, chunk c stored at position
//   c ^ ((x >> 2) & 3) (source-side swizzle, the DMA writes linearly): the 16 lanes of an MFMA row group read 64 banks.
//   Lane group q of the 16x16x4 shape takes chunk q -- k = 4q .. 4q+3 -- and feeds component e to k-step e: both
//   operands use the same permutation of the 16 k, so the sum is unchanged and a fragment is ONE ds_read_b128 per stage.
//   EPI 3 / 4 (factor form): Bm holds the k-contiguous COLUMNS of the triangular operand as its rows (Li for C = Phi Li^T,
//   Li^T for V = C Li); the stages run over k < cbase + BN (EPI 3) or k >= cbase (EPI 4) only.
//   WGM_ = 2 (experiment, BN = 256 only): 8 waves of 128 x 64 instead of 16 of 64 x 64 -- 12 fragment reads per 128 MFMAs instead of
//   8 per 64, one workgroup per CU at two waves per SIMD
template <int BN_, int WGM_ = 4>
struct ApplyDma {
    static constexpr int BM = 256, BN = BN_, WAVES = WGM_ * (BN / 64), STAGE = (BM + BN) * 64, STAGES = 3, LDS_BYTES = STAGES * STAGE,
                         DMA_PER_WAVE = STAGE / 1024 / WAVES;
    typedef TileCfg<float, BM, BN, 16, WGM_, BN / 64, 16, true> Cfg;      // wave grid / accumulator map of the epilogue: (256 / WGM_) x 64 wave tiles
    static_assert(STAGE / 1024 % WAVES == 0, "whole DMA instructions per wave");
};
// BN = 128: 8 waves, two workgroups per CU, 48 operand bytes per MFMA; BN = 256: 16 waves, one workgroup per CU, 32 bytes per MFMA
// one 8-byte LDS read at a literal offset, as inline asm: the compiler can neither merge two of them into a half-rate ds_read2st64_b64
// nor replace the counted waits of the pipelined loop by lgkmcnt(0).  The result register is only valid after the loop's own
// s_waitcnt lgkmcnt: the compiler does not know that, so the value must reach its MFMA without an intermediate copy (checked in the
// ISA of the three PIPE instantiations: the reads land in the registers the MFMAs name)
template <int OFF> __device__ __forceinline__ v2f lds_read_b64(unsigned addr) {
    v2f r;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
// in-place MFMA as inline asm: between the scheduling barriers of the pipelined loop the register allocator otherwise gives every
// MFMA a fresh destination tuple (two live copies of the 64 accumulator registers: spills)
__device__ __forceinline__ void mfma_inplace(v4f& c, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
#define SCFGP_RD4(dst, base, half)                                                                                 \
    do { dst[0] = lds_read_b64<0 + half>(base); dst[1] = lds_read_b64<1024 + half>(base);                        \
         dst[2] = lds_read_b64<2048 + half>(base); dst[3] = lds_read_b64<3072 + half>(base); } while (0)

//   PIPE (experiment, option apply_dma = 5): fragment reads software-pipelined at half-stage granularity -- the 16 k of a stage are
//   consumed as two halves (the first and the last 8 bytes of every lane's 16-byte chunk), each half is fetched while the other is
//   multiplied, the stage's barrier sits between the halves and the ring runs three stages ahead, so neither LDS latency nor
//   the DMA issue stands between a wave's MFMAs; same registers.  Reads and waits are inline asm (see lds_read_b64).
template <int EPI, int BN, int WGM = 4, int PIPE = 0>
__global__ __launch_bounds__((64 * ApplyDma<BN, WGM>::WAVES)) __attribute__((amdgpu_waves_per_eu(WGM == 2 ? 2 : 4, WGM == 2 ? 2 : 4)))
void apply_dma_kernel(const float* __restrict__ Phi, const float* __restrict__ Bm, float* V,
                      double* __restrict__ vpart, const double* __restrict__ p, const double* __restrict__ q,
                      const double* __restrict__ y, const double* __restrict__ alpha, const double* __restrict__ ut,
                      int K, int Kp, int64_t Np, int njt, double* __restrict__ bpart, double* __restrict__ mu, int col0, int slot0) {
    typedef ApplyDma<BN, WGM> D;
    typedef typename D::Cfg Cfg;
    SMEM_DECL;
    char* smem = smem_raw;
    const unsigned wid = xcd_remap(blockIdx.x, gridDim.x);
    const int jt = wid % njt;
    const int64_t rb = wid / njt;
    const int cbase = col0 + jt * D::BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // DMA instruction t = DMA_PER_WAVE wave + u of a stage: rows 16 t .. 16 t + 15 of the stacked (A: 256, then B: BN) operand
    // rows; lane l carries row + l / 4, position l % 4 <- chunk (l % 4) ^ ((row >> 2) & 3)
    const char* src[D::DMA_PER_WAVE]; int dst[D::DMA_PER_WAVE];
#pragma unroll
    for (int u = 0; u < D::DMA_PER_WAVE; ++u) {
        const int t = wave * D::DMA_PER_WAVE + u, x = 16 * t + (lane >> 2), c = (lane & 3) ^ ((x >> 2) & 3);
        // B rows in the VEC4 order (apply_epilogue): LDS row tn*16 + i of a 64-wide wave tile holds operand row 4 i + tn
        const int xb = x - D::BM, xcol = (xb & ~63) + 4 * (xb & 15) + ((xb >> 4) & 3);
        // EPI 5 (zbar_epilogue_vec4): col0 counts j, tile jt covers j in [jb, jb + BN / 2); LDS row 64 w + 16 tn + i holds operand row
        // jb + 32 w + 2 i + (tn & 1) for tn < 2 and J + that for tn >= 2
        const int zj = col0 + jt * (D::BN / 2) + 32 * (xb >> 6) + 2 * (xb & 15) + ((xb >> 4) & 1);
        const int brow = EPI == 5 ? (((xb >> 4) & 2) ? K / 2 + zj : zj) : cbase + xcol;
        const float* rowp = x < D::BM ? Phi + (rb * D::BM + x) * Kp : Bm + (int64_t)brow * Kp;
        src[u] = reinterpret_cast<const char*>(rowp) + c * 16 + (EPI == 4 ? (cbase / 16) * 64 : 0);
        dst[u] = t * 1024;
    }
    const auto issue = [&](int slot) {
#pragma unroll
        for (int u = 0; u < D::DMA_PER_WAVE; ++u) {
            __builtin_amdgcn_global_load_lds((gbl_void*)src[u], (lds_void*)(smem + slot * D::STAGE + dst[u]), 16, 0, 0);
            src[u] += 64;                                       // 16 k further
        }
    };
    const int i = lane & 15, qg = lane >> 4;
    const int wm0 = (wave / Cfg::WGN) * Cfg::WM, wn0 = (wave % Cfg::WGN) * Cfg::WN;
    const int sw = (qg ^ ((i >> 2) & 3)) << 4;
    const int aoff = (wm0 + i) * 64 + sw, boff = D::BM * 64 + (wn0 + i) * 64 + sw;
    typename Cfg::MTr::acc_t acc[Cfg::TM][Cfg::TN];
    acc_zero<Cfg>(acc);
    const int nst_all = (K + 15) / 16, nst_tri = (cbase + D::BN + 15) / 16;
    const int nst = (EPI == 3 && nst_tri < nst_all ? nst_tri : nst_all) - (EPI == 4 ? cbase / 16 : 0);
    if constexpr (PIPE) {
        static_assert(Cfg::TM == 4 && Cfg::TN == 4 && (D::DMA_PER_WAVE == 2 || D::DMA_PER_WAVE == 3), "pipelined loop: 64 x 64 wave tiles");
        const auto wait_landed = [&](int after) {                 // all but the youngest `after` stages of this wave's DMAs are done
            constexpr int P = D::DMA_PER_WAVE;
            if (after >= 2) { if (P == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
            else if (after == 1) { if (P == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem;
        issue(0);
        if (nst > 1) issue(1);
        if (nst > 2) issue(2);
        wait_landed(nst > 2 ? 2 : (nst > 1 ? 1 : 0));
        asm volatile("s_barrier" ::: "memory");
        v2f ax[4], bx[4], ay[4], by[4];
        SCFGP_RD4(ax, lds0 + aoff, 0); SCFGP_RD4(bx, lds0 + boff, 0);
        int slot = 0;
        for (int s = 0; s < nst; ++s) {
            const unsigned cur = lds0 + slot * D::STAGE;
            const int nslot = slot == 2 ? 0 : slot + 1;
            SCFGP_RD4(ay, cur + aoff, 8); SCFGP_RD4(by, cur + boff, 8);
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");   // the first half (issued half a stage ago) is in; the second is in flight
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn) mfma_inplace(acc[tm][tn], ax[tm][e], bx[tn][e]);
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < nst) {
                // stage s+1 has landed (stage s+2 may still be in flight); this wave's reads of slot `slot` are complete
                wait_landed(s + 2 < nst ? 1 : 0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("s_barrier" ::: "memory");
                if (s + 3 < nst) issue(slot);                    // everybody is done with stage s: its slot takes stage s+3
                const unsigned nxt = lds0 + nslot * D::STAGE;
                SCFGP_RD4(ax, nxt + aoff, 0); SCFGP_RD4(bx, nxt + boff, 0);
            } else
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn) mfma_inplace(acc[tm][tn], ay[tm][e], by[tn][e]);
            __builtin_amdgcn_sched_barrier(0);
            slot = nslot;
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // the last MFMA results before ordinary code reads them
    } else {
    issue(0);
    if (nst > 1) issue(1);
    int slot = 0, fill = 2;
    for (int s = 0; s < nst; ++s) {
        // this wave's share of stage s has landed when only the DMAs of stage s+1 are outstanding
        if (s + 1 < nst) {
            if (D::DMA_PER_WAVE == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (D::DMA_PER_WAVE == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // everybody's has; nobody reads the slot of stage s-1 any more
        asm volatile("" ::: "memory");
#ifndef SCFGP_DIAG_DMA_NODMA                                   // timing diagnostic only (wrong numbers): no operand traffic after the prologue
        if (s + 2 < nst) issue(fill);
#endif
        const char* base = smem + slot * D::STAGE;
        v4f a[Cfg::TM], b[Cfg::TN];
#pragma unroll
        for (int tm = 0; tm < Cfg::TM; ++tm) a[tm] = *reinterpret_cast<const v4f*>(base + aoff + tm * 1024);
#pragma unroll
        for (int tn = 0; tn < Cfg::TN; ++tn) b[tn] = *reinterpret_cast<const v4f*>(base + boff + tn * 1024);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < Cfg::TN; ++tn) Cfg::MTr::mfma(acc[tm][tn], a[tm][e], b[tn][e]);
        slot = slot == 2 ? 0 : slot + 1;
        fill = fill == 2 ? 0 : fill + 1;
    }
    }
    __syncthreads();
    constexpr int SLOTS = BN / 128;                            // vpart / mupart slots are 128 columns wide
    const int vslot = slot0 + SLOTS * jt;
    if constexpr (EPI == 5) zbar_epilogue_vec4<Cfg>(acc, Phi, V, p, q, y, alpha, ut, K / 2, Kp, rb, col0 + jt * (D::BN / 2), bpart, smem_raw);
    else apply_epilogue<Cfg, EPI, EPI == 0 || EPI == 3, true>(acc, Phi, V, vpart, p, q, y, alpha, ut, K, Kp, Np, rb, cbase, vslot, bpart, smem_raw, mu);
    if (SLOTS == 2 && (EPI == 0 || EPI == 3) && threadIdx.x < D::BM) {
        vpart[(int64_t)(vslot + 1) * Np + rb * D::BM + threadIdx.x] = 0.0;
        if (mu) mu[(int64_t)(vslot + 1) * Np + rb * D::BM + threadIdx.x] = 0.0;
    }
}

// DMA-fed split-precision product (tile_bf16x3_dma.h): column tiles [0, 256 njt) of C = Phi . Bm from the row planes of Phi
// and the matrix planes of Bm; epilogues as above, a tile reports through slot 2 jt of vpart / mupart (slot 2 jt + 1: zero)
template <int EPI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void apply_bf3dma_kernel(const char* __restrict__ Apl, const char* __restrict__ Bpl, const float* __restrict__ Phi, float* V,
                         double* __restrict__ vpart, const double* __restrict__ p, const double* __restrict__ q,
                         const double* __restrict__ y, const double* __restrict__ alpha, const double* __restrict__ ut,
                         int K, int Kp, int64_t Np, int njt, double* __restrict__ bpart, double* __restrict__ mu) {
    typedef Bf3DCfg Cfg;
    SMEM_DECL;
    const unsigned wid = xcd_remap(blockIdx.x, gridDim.x);
    const int jt = wid % njt;
    const int64_t rb = wid / njt;
    const int cbase = jt * Bf3D::BN;
    typename Cfg::MTr::acc_t acc[Cfg::TM][Cfg::TN];
    acc_zero<Cfg>(acc);
    bf3dma_mainloop(Apl + rb * Bf3D::BM * 32, Np * 32, Bpl + (int64_t)cbase * 32, (int64_t)Kp * 32, (K + 15) / 16, acc, smem_raw);
    apply_epilogue<Cfg, EPI, EPI == 0>(acc, Phi, V, vpart, p, q, y, alpha, ut, K, Kp, Np, rb, cbase, 2 * jt, bpart, smem_raw, mu);
    if (EPI == 0 && threadIdx.x < Bf3D::BM) {
        vpart[(int64_t)(2 * jt + 1) * Np + rb * Bf3D::BM + threadIdx.x] = 0.0;
        if (mu) mu[(int64_t)(2 * jt + 1) * Np + rb * Bf3D::BM + threadIdx.x] = 0.0;
    }
}

// Columns [0, K) of the output are covered by a cascade of launches of decreasing tile width
// (APPLY_BN, then 128, then 64-wide tiles for the ragged remainder; K = 2112 with 128-wide tiles:
// 16 x 128 + 1 x 64), so no workgroup multiplies a half-empty tile; columns >= K of the output buffer are
// never written and stay zero.
template <typename T> struct ApplyPlan {
    static constexpr int NW = 3;
    int width[NW], count[NW], col0[NW], jt0[NW], total;
    explicit ApplyPlan(int K) {
        const int w[NW] = {Tune<T>::APPLY_BN, 128, 64};
        int col = 0, jt = 0;
        // K <= 256: a handful of workgroups whatever the tiling, so ONE launch of 64-wide tiles (a launch costs more
        // than the half-empty tile there: Boston shape, K = 144, 80 -> 45 us per product)
        const bool small = K <= 256;
        for (int i = 0; i < NW; ++i) {
            width[i] = w[i];
            const bool last = i == NW - 1, dup = i > 0 && w[i] >= w[i - 1];
            count[i] = dup || (small && !last) ? 0 : (last ? (K - col + w[i] - 1) / w[i] : (K - col) / w[i]);
            col0[i] = col; jt0[i] = jt;
            col += count[i] * w[i]; jt += count[i];
        }
        total = jt;
    }
};
template <class Cfg, int EPI, typename T>
static int apply_launch_cfg(const Geom& g, int njt, int col0, int jt0, int boff, const T* Phi, const T* Bm, T* V, double* vpart,
                            const double* p, const double* q, const double* y, const double* alpha, const double* ut,
                            double* bpart, double* mu, hipStream_t st) {
    if (njt <= 0) return 0;
    const int64_t nrb = g.Np / Cfg::BM;
    const int wgs_per_rb = EPI == 3 || EPI == 4 ? (njt + 1) / 2 : njt;          // triangular products pair their column tiles
    allow_big_lds(apply_kernel<Cfg, EPI>, Cfg::LDS_BYTES);
    hipLaunchKernelGGL((apply_kernel<Cfg, EPI>), dim3((unsigned)(wgs_per_rb * nrb)), dim3(Cfg::THREADS), Cfg::LDS_BYTES, st,
                       Phi, Bm, V, vpart, p, q, y, alpha, ut, g.K, g.Kp, g.Np, njt, bpart ? bpart + boff : nullptr, col0, jt0, mu, ApplyPlan<T>(g.K).total);
    return (int)(njt * nrb);
}
template <int EPI, int BN, int WGM = 4, int PIPE = 0>
static int apply_dma_launch(const Geom& g, int njt, int col0, int slot0, int boff, const float* Phi, const float* Bm, float* V, double* vpart,
                            const double* p, const double* q, const double* y, const double* alpha, const double* ut,
                            double* bpart, double* mu, hipStream_t st) {
    if (njt <= 0) return 0;
    typedef ApplyDma<BN, WGM> D;
    const int64_t nrb = g.Np / D::BM;
    allow_big_lds(apply_dma_kernel<EPI, BN, WGM, PIPE>, D::LDS_BYTES);
    hipLaunchKernelGGL((apply_dma_kernel<EPI, BN, WGM, PIPE>), dim3((unsigned)(njt * nrb)), dim3(64 * D::WAVES), D::LDS_BYTES, st,
                       Phi, Bm, V, vpart, p, q, y, alpha, ut, g.K, g.Kp, g.Np, njt, bpart ? bpart + boff : nullptr, mu, col0, slot0);
    return (int)(njt * nrb);
}
// bf3: Bm is the matrix pre-split for Bf3CopyLoader; with planes (row planes of Phi + the 16-deep matrix planes of Bm) the
// 256-wide column tiles go through the DMA-fed kernel and only the remainder through the loader-split tiles
template <typename T, int EPI>
static int apply_launch(const Geom& g, const T* Phi, const T* Bm, T* V, double* vpart, const double* p, const double* q,
                        const double* y, const double* alpha, const double* ut, double* bpart, double* mu, hipStream_t st, bool bf3 = false,
                        const Bf3Planes* planes = nullptr, const T* BmT = nullptr) {
    // BmT: the operand with its k-contiguous columns as rows (what the DMA-fed tiles read); NULL: Bm is symmetric
    if (!BmT) BmT = Bm;
    const ApplyPlan<T> pl(g.K);
    int nb = 0;
    if constexpr (sizeof(T) == 4) {
        if (bf3) {                                             // split-precision tiles (same column plan: 128-wide, then 64)
            static_assert(Tune<T>::APPLY_BN == 128, "bf16x3 apply tiles are 128 and 64 wide");
            const int n256 = planes && planes->rows && planes->matrix16 && EPI != 2 && g.K > 256 ? g.K / 256 : 0;
            if (n256 > 0) {
                const int64_t nrb = g.Np / Bf3D::BM;
                allow_big_lds(apply_bf3dma_kernel<EPI>, Bf3D::LDS_BYTES);
                hipLaunchKernelGGL((apply_bf3dma_kernel<EPI>), dim3((unsigned)(n256 * nrb)), dim3(512), Bf3D::LDS_BYTES, st,
                                   (const char*)planes->rows, (const char*)planes->matrix16, Phi, V, vpart, p, q, y, alpha, ut,
                                   g.K, g.Kp, g.Np, n256, bpart, mu);
                nb += (int)(n256 * nrb);
                // remainder: at most one 128-wide tile, then 64-wide ones; slots continue after the 2 n256 of the wide tiles
                const int c0 = 256 * n256, n128 = (g.K - c0) / 128, c1 = c0 + 128 * n128, n64 = (g.K - c1 + 63) / 64;
                const auto rest = [&](auto cfg, int njt, int col0, int jt0) {
                    typedef typename decltype(cfg)::type Cfg;
                    if (njt <= 0) return;
                    const int64_t nr = g.Np / Cfg::BM;
                    allow_big_lds(apply_kernel<Cfg, EPI>, Cfg::LDS_BYTES);
                    hipLaunchKernelGGL((apply_kernel<Cfg, EPI>), dim3((unsigned)(njt * nr)), dim3(Cfg::THREADS), Cfg::LDS_BYTES, st,
                                       Phi, Bm, V, vpart, p, q, y, alpha, ut, g.K, g.Kp, g.Np, njt, bpart ? bpart + nb : nullptr, col0, jt0, mu,
                                       0);                      // ntot = 0: mu slices are column bands
                    nb += (int)(njt * nr);
                };
                rest(ApplyBf3Cfg<128>{}, n128, c0, 2 * n256);
                rest(ApplyBf3Cfg<64>{}, n64, c1, 2 * n256 + n128);
                return nb;
            }
            nb += apply_launch_cfg<typename ApplyBf3Cfg<128>::type, EPI, T>(g, pl.count[0], pl.col0[0], pl.jt0[0], nb, Phi, Bm, V, vpart, p, q, y, alpha, ut, bpart, mu, st);
            nb += apply_launch_cfg<typename ApplyBf3Cfg<64>::type, EPI, T>(g, pl.count[2], pl.col0[2], pl.jt0[2], nb, Phi, Bm, V, vpart, p, q, y, alpha, ut, bpart, mu, st);
            return nb;
        }
    }
    if constexpr (sizeof(T) == 4 && EPI != 2) {
        if (planes && planes->dma && pl.count[0] > 0 && Tune<T>::APPLY_BN == 128 && Tune<T>::APPLY_BM == 256) {
            // the full tiles by LDS-DMA (Bm symmetric): option value 1 = 128 wide, 2 = 256 wide with a 128-wide one for an odd
            // count; the 64-wide remainder by the loader-staged kernel, whose mu slices then are column bands too (ntot = 0)
            // 3: 256-wide for V = Phi.B (EPI 0), 128-wide for Phibar (EPI 1: its epilogue wants a second resident workgroup)
            const bool wide = planes->dma == 2 || planes->dma == 4 || planes->dma == 5 || (planes->dma == 3 && EPI != 1);
            const int n256 = wide ? pl.count[0] / 2 : 0, n128 = pl.count[0] - 2 * n256;
            bool done256 = false, done128 = false;
            if constexpr (EPI < 2) {
                if (planes->dma == 4) {                            // experiment: 8 waves of 128 x 64
                    nb += apply_dma_launch<EPI, 256, 2>(g, n256, 0, 0, nb, Phi, BmT, V, vpart, p, q, y, alpha, ut, bpart, mu, st); done256 = true;
                } else if (planes->dma == 5) {                     // experiment: half-stage pipelined fragment reads
                    nb += apply_dma_launch<EPI, 256, 4, 1>(g, n256, 0, 0, nb, Phi, BmT, V, vpart, p, q, y, alpha, ut, bpart, mu, st); done256 = true;
                    nb += apply_dma_launch<EPI, 128, 4, 1>(g, n128, 256 * n256, 2 * n256, nb, Phi, BmT, V, vpart, p, q, y, alpha, ut, bpart, mu, st); done128 = true;
                }
            }
            if (!done256) nb += apply_dma_launch<EPI, 256>(g, n256, 0, 0, nb, Phi, BmT, V, vpart, p, q, y, alpha, ut, bpart, mu, st);
            if (!done128) nb += apply_dma_launch<EPI, 128>(g, n128, 256 * n256, 2 * n256, nb, Phi, BmT, V, vpart, p, q, y, alpha, ut, bpart, mu, st);
            typedef typename ApplyCfg<T, 64>::type RCfg;
            if (pl.count[2] > 0) {
                const int64_t nr = g.Np / RCfg::BM;
                allow_big_lds(apply_kernel<RCfg, EPI>, RCfg::LDS_BYTES);
                hipLaunchKernelGGL((apply_kernel<RCfg, EPI>), dim3((unsigned)((EPI == 3 || EPI == 4 ? (pl.count[2] + 1) / 2 : pl.count[2]) * nr)),
                                   dim3(RCfg::THREADS), RCfg::LDS_BYTES, st,
                                   Phi, Bm, V, vpart, p, q, y, alpha, ut, g.K, g.Kp, g.Np, pl.count[2], bpart ? bpart + nb : nullptr,
                                   pl.col0[2], pl.jt0[2], mu, 0);
                nb += (int)(pl.count[2] * nr);
            }
            return nb;
        }
    }
    nb += apply_launch_cfg<typename ApplyCfg<T, Tune<T>::APPLY_BN>::type, EPI, T>(g, pl.count[0], pl.col0[0], pl.jt0[0], nb, Phi, Bm, V, vpart, p, q, y, alpha, ut, bpart, mu, st);
    nb += apply_launch_cfg<typename ApplyCfg<T, 128>::type, EPI, T>(g, pl.count[1], pl.col0[1], pl.jt0[1], nb, Phi, Bm, V, vpart, p, q, y, alpha, ut, bpart, mu, st);
    nb += apply_launch_cfg<typename ApplyCfg<T, 64>::type, EPI, T>(g, pl.count[2], pl.col0[2], pl.jt0[2], nb, Phi, Bm, V, vpart, p, q, y, alpha, ut, bpart, mu, st);
    return nb;
}
// fp32 K x K operand -> the plane layout Bf3CopyLoader reads (tile_bf16x3.h); out: Kp * Kp * 6 bytes
void bf3_presplit(const float* M, void* out, int Kp, hipStream_t st) {
    hipLaunchKernelGGL(bf3_presplit_kernel, dim3(2048), dim3(256), 0, st, M, reinterpret_cast<__bf16*>(out), Kp);
}
void bf3_presplit16(const float* M, void* out, int Kp, hipStream_t st) {
    hipLaunchKernelGGL(bf3_presplit16_kernel, dim3(2048), dim3(256), 0, st, M, reinterpret_cast<__bf16*>(out), Kp);
}
void bf3_split_rows(const float* S, int64_t ld, void* out, int64_t Np, int Kp, hipStream_t st) {
    hipLaunchKernelGGL(bf3_split_rows_kernel, dim3((unsigned)((Kp / 128) * (Np / 64))), dim3(256), 0, st, S, ld,
                       reinterpret_cast<__bf16*>(out), Np, Kp);
}

template <typename T>
void SweepKernels<T>::apply_v(const Geom& g, const T* Phi, const T* Bm, T* V, double* vpart, const double* alpha, double* mu,
                              hipStream_t st, bool bf3, const Bf3Planes* planes) {
#ifdef SCFGP_DIAG_NOMU
    mu = nullptr;                                              // timing diagnostic only: wrong numbers
#endif
    apply_launch<T, 0>(g, Phi, Bm, V, vpart, nullptr, nullptr, nullptr, alpha, nullptr, nullptr, mu, st, bf3, planes);
}
template <typename T>
void SweepKernels<T>::apply_predict(const Geom& g, const T* Phi, const T* LiT, double* vpart, const double* alpha, double* mu,
                                    hipStream_t st, bool bf3) {
    apply_launch<T, 2>(g, Phi, LiT, (T*)nullptr, vpart, nullptr, nullptr, nullptr, alpha, nullptr, nullptr, mu, st, bf3);
}
// Out[n][c] = sum_{k < Kc} A[n][k] Bm[k][c] for c < ncols (rounded up to 64-wide tiles), everything with leading dimension Kp;
// Bm[k][c] = 0 for k < c is assumed (the contraction of a column tile starts at its first column)
template <typename T>
void SweepKernels<T>::apply_plain(const Geom& g, const T* A, const T* Bm, T* Out, int Kc, int ncols, hipStream_t st) {
    Geom gk = g; gk.K = Kc;
    apply_launch_cfg<typename ApplyCfg<T, 64>::type, 4, T>(gk, (ncols + 63) / 64, 0, 0, 0, A, Bm, Out, nullptr, nullptr, nullptr, nullptr, nullptr,
                                                           nullptr, nullptr, nullptr, st);
}

template <typename T>
void SweepKernels<T>::apply_c(const Geom& g, const T* Phi, const T* LiT, const T* Li, T* C, double* vpart, const double* alpha,
                              double* mu, hipStream_t st, const Bf3Planes* planes) {
    apply_launch<T, 3>(g, Phi, LiT, C, vpart, nullptr, nullptr, nullptr, alpha, nullptr, nullptr, mu, st, false, planes, Li);
}
template <typename T>
void SweepKernels<T>::apply_vc(const Geom& g, const T* C, const T* Li, const T* LiT, T* V, hipStream_t st, const Bf3Planes* planes) {
    apply_launch<T, 4>(g, C, Li, V, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, st, false, planes, LiT);
}
template <typename T>
int SweepKernels<T>::apply_phibar(const Geom& g, const T* Phi, const T* Abar, T* V, const double* p, const double* q,
                                  const double* y, const double* alpha, const double* ut, double* bpart, hipStream_t st, bool bf3,
                                  const Bf3Planes* planes) {
    return apply_launch<T, 1>(g, Phi, Abar, V, nullptr, p, q, y, alpha, ut, bpart, nullptr, st, bf3, planes);
}

// Phibar product with the Zbar epilogue (EPI 5): Zbar[n][j], j < J, over V[n][j]; returns the number of bbar partials, or -1
// when the fused form does not apply (fp64, split-precision mode, no LDS-DMA tiles at this size, J % 4 != 0) and the caller runs
// apply_phibar + the Zbar pass instead.  j plan: 128 j per 256-wide LDS-DMA tile, 64 per 128-wide one, 32 per loader-staged
// 64-wide tile (the last of those may be ragged).
template <typename T>
int SweepKernels<T>::apply_zbar(const Geom& g, const T* Phi, const T* Abar, T* V, const double* p, const double* q,
                                const double* y, const double* alpha, const double* ut, double* bpart, hipStream_t st,
                                const Bf3Planes* planes) {
    if constexpr (sizeof(T) != 4) return -1;
    else {
        if (!planes || !planes->dma || planes->rows || g.J % 4 != 0 || g.K <= 256) return -1;
        const bool wide = planes->dma != 1;
        const int n256 = wide ? g.J / 128 : 0, n128 = (g.J - 128 * n256) / 64;
        const int jrest = g.J - 128 * n256 - 64 * n128, n64 = (jrest + 31) / 32;
        if (n256 + n128 == 0) return -1;
        int nb = 0;
        nb += apply_dma_launch<5, 256>(g, n256, 0, 0, nb, Phi, Abar, V, nullptr, p, q, y, alpha, ut, bpart, nullptr, st);
        nb += apply_dma_launch<5, 128>(g, n128, 128 * n256, 0, nb, Phi, Abar, V, nullptr, p, q, y, alpha, ut, bpart, nullptr, st);
        if (n64 > 0) {
            typedef typename ApplyCfg<T, 64>::type RCfg;
            const int64_t nr = g.Np / RCfg::BM;
            allow_big_lds(apply_kernel<RCfg, 5>, RCfg::LDS_BYTES);
            hipLaunchKernelGGL((apply_kernel<RCfg, 5>), dim3((unsigned)(n64 * nr)), dim3(RCfg::THREADS), RCfg::LDS_BYTES, st,
                               Phi, Abar, V, nullptr, p, q, y, alpha, ut, g.K, g.Kp, g.Np, n64, bpart + nb, 128 * n256 + 64 * n128, 0, nullptr, 0);
            nb += (int)(n64 * nr);
        }
        return nb;
    }
}
// number of column tiles of the apply kernel (vpart leading count)
template <typename T> static int apply_njt(const Geom& g) { return ApplyPlan<T>(g.K).total; }
template <typename T>
int SweepKernels<T>::apply_blocks(const Geom& g) {
    return (int)(apply_njt<T>(g) * (g.Np / Tune<T>::APPLY_BM));
}


// --------------------------------------------------------------------------
// per-row statistics from apply_v's per-tile by-products: one thread per row.
//   mu = sum_jt mupart, v = sum_jt vpart, d = kappa (v+1), r = mu - y
//   MODE 0 (train): p = 2r/d, e = 1/d - (r^2+v)/d^2, q = 1/d + kappa e; block partials of
//                   T2 = (r^2+v)/d + log(2 pi d) and kbar = e (v+1)
//   MODE 1 (predict): mu, sd = sqrt(kappa (1+v))
// --------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void rowstats_kernel(const double* __restrict__ mupart, const double* __restrict__ vpart, int njt,
                                                       const double* __restrict__ y, const Scal* __restrict__ sc,
                                                       double* __restrict__ o1, double* __restrict__ o2,
                                                       double* __restrict__ partial, int64_t N, int64_t Np) {
    __shared__ double red[2][256];
    const double kappa = sc->kappa;
    double t2 = 0, kb = 0;
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < Np; n += (int64_t)gridDim.x * 256) {
        double v = 0, mu = 0;
        for (int t = 0; t < njt; ++t) { v += vpart[(int64_t)t * Np + n]; mu += mupart[(int64_t)t * Np + n]; }
        const double d = kappa * (v + 1.0);
        if (MODE == 0) {
            double pn = 0, qn = 0;
            if (n < N) {
                const double r = mu - y[n];
                const double rv = r * r + v;
                const double e = 1.0 / d - rv / (d * d);
                pn = 2.0 * r / d;
                qn = 1.0 / d + kappa * e;
                t2 += rv / d + log(2.0 * M_PI * d);
                kb += e * (v + 1.0);
            }
            o1[n] = pn; o2[n] = qn;
        } else if (n < N) {
            o1[n] = mu; o2[n] = sqrt(d);
        }
    }
    if (MODE == 0) {
        red[0][threadIdx.x] = t2; red[1][threadIdx.x] = kb;
        __syncthreads();
        for (int w = 128; w >= 1; w >>= 1) {
            if ((int)threadIdx.x < w) { red[0][threadIdx.x] += red[0][threadIdx.x + w]; red[1][threadIdx.x] += red[1][threadIdx.x + w]; }
            __syncthreads();
        }
        if (threadIdx.x < 2) partial[blockIdx.x * 2 + threadIdx.x] = red[threadIdx.x][0];
    }
}

template <typename T>
void SweepKernels<T>::rowstats(const Geom& g, const double* mupart, const double* vpart, const double* y,
                               const Scal* sc, double* p, double* q, double* partial, int nblocks, hipStream_t st) {
    hipLaunchKernelGGL((rowstats_kernel<0>), dim3(nblocks), dim3(256), 0, st, mupart, vpart, apply_njt<T>(g), y, sc, p, q,
                       partial, g.N, g.Np);
}

template <typename T>
void SweepKernels<T>::rowpredict(const Geom& g, const double* mupart, const double* vpart, const Scal* sc, double* mu, double* sd,
                                 hipStream_t st) {
    const int nblocks = (int)((g.Np + 255) / 256 < 1024 ? (g.Np + 255) / 256 : 1024);
    hipLaunchKernelGGL((rowstats_kernel<1>), dim3(nblocks), dim3(256), 0, st, mupart, vpart, apply_njt<T>(g),
                       (const double*)nullptr, sc, mu, sd, (double*)nullptr, g.N, g.Np);
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
void SweepKernels<T>::convert_transposed(const double* src, T* dst, int K, int Kp, hipStream_t st) {
    hipLaunchKernelGGL(convert_t_kernel<T>, dim3(2048), dim3(256), 0, st, src, dst, K, Kp);
}
template <typename T>
void SweepKernels<T>::convert(const double* src, T* dst, int K, int Kp, hipStream_t st) {
    hipLaunchKernelGGL(convert_kernel<T>, dim3(2048), dim3(256), 0, st, src, dst, K, Kp);
}
template struct SweepKernels<double>;
template struct SweepKernels<float>;

// --------------------------------------------------------------------------
// reductions (deterministic: fixed order over splits)
// --------------------------------------------------------------------------
// packed lower-tile layout of a symmetric Kp x Kp matrix: tile (ti >= tj) number t = ti(ti+1)/2 + tj holds
// its B x B elements row-major at [t*B*B, (t+1)*B*B) -- what the all-reduce of a sharded run moves
// (K^2/2 instead of K^2 doubles).
__global__ __launch_bounds__(256) void reduce_tri_kernel(const double* __restrict__ slabs, int nsplit, int ntiles, int B,
                                                         double* __restrict__ packed) {
    const int t = blockIdx.x;
    for (int e = blockIdx.y * 256 + threadIdx.x; e < B * B; e += gridDim.y * 256) {
        double s = 0;
        for (int sp = 0; sp < nsplit; ++sp) s += slabs[((int64_t)sp * ntiles + t) * (B * B) + e];
        packed[(int64_t)t * B * B + e] = s;
    }
}
__global__ __launch_bounds__(256) void reduce_tri_cnt_kernel(const double* __restrict__ slabs, const int* __restrict__ cnt, int ntiles, int B,
                                                             double* __restrict__ packed) {
    const int t = blockIdx.x, n = cnt[t];
    for (int e = blockIdx.y * 256 + threadIdx.x; e < B * B; e += gridDim.y * 256) {
        double s = 0;
        for (int sp = 0; sp < n; ++sp) s += slabs[((int64_t)sp * ntiles + t) * (B * B) + e];
        packed[(int64_t)t * B * B + e] = s;
    }
}
void reduce_tri_tiles_cnt(const double* slabs, const int* cnt, int nts, int tile, double* packed, hipStream_t st) {
    const int ntiles = nts * (nts + 1) / 2;
    hipLaunchKernelGGL(reduce_tri_cnt_kernel, dim3(ntiles, 16), dim3(256), 0, st, slabs, cnt, ntiles, tile, packed);
}
void reduce_tri_tiles(const double* slabs, int nsplit, int nts, int tile, double* packed, hipStream_t st) {
    const int ntiles = nts * (nts + 1) / 2;
    hipLaunchKernelGGL(reduce_tri_kernel, dim3(ntiles, 16), dim3(256), 0, st, slabs, nsplit, ntiles, tile, packed);
}
// vec[j] = sum over splits of the Gram's side partials for j < ncov (columns covered by diagonal tiles), 0 beyond
__global__ void reduce_side_kernel(const double* __restrict__ sidepart, int nsplit, int Kp, int ncov, double* __restrict__ vec) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Kp) return;
    double s = 0;
    if (j < ncov)
        for (int sp = 0; sp < nsplit; ++sp) s += sidepart[(int64_t)sp * Kp + j];
    vec[j] = s;
}
void reduce_side(const double* sidepart, int nsplit, int Kp, int ncov, double* vec, hipStream_t st) {
    hipLaunchKernelGGL(reduce_side_kernel, dim3((Kp + 255) / 256), dim3(256), 0, st, sidepart, nsplit, Kp, ncov, vec);
}
// full symmetric matrix (ld = Kp) from the packed lower tiles; diagonal tiles carry both triangles
__global__ __launch_bounds__(256) void unpack_tri_kernel(const double* __restrict__ packed, int B, double* __restrict__ full, int64_t ld) {
    const int t = blockIdx.x;
    int ti = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    while (ti * (ti + 1) / 2 > t) --ti;
    const int tj = t - ti * (ti + 1) / 2;
    for (int e = blockIdx.y * 256 + threadIdx.x; e < B * B; e += gridDim.y * 256) {
        const double v = packed[(int64_t)t * B * B + e];
        const int i = ti * B + e / B, j = tj * B + e % B;
        full[(int64_t)i * ld + j] = v;
        if (ti != tj) full[(int64_t)j * ld + i] = v;
    }
}
void unpack_tri_tiles(const double* packed, int nts, int tile, double* full, int64_t ld, hipStream_t st) {
    hipLaunchKernelGGL(unpack_tri_kernel, dim3(nts * (nts + 1) / 2, 16), dim3(256), 0, st, packed, tile, full, ld);
}

__global__ __launch_bounds__(256) void reduce_full_kernel(const double* __restrict__ slabs, int nsplit, int ntiles, int ntn,
                                                          double* __restrict__ out, int64_t ldo) {
    constexpr int B = 128;
    const int t = blockIdx.x, ti = t / ntn, tj = t % ntn;
    for (int e = blockIdx.y * 256 + threadIdx.x; e < B * B; e += gridDim.y * 256) {
        double s = 0;
        for (int sp = 0; sp < nsplit; ++sp) s += slabs[((int64_t)sp * ntiles + t) * (B * B) + e];
        out[(int64_t)(ti * B + e / B) * ldo + tj * B + e % B] = s;
    }
}
void reduce_full_tiles(const double* slabs, int nsplit, int ntm, int ntn, double* out, int64_t ldo, hipStream_t st) {
    // few tiles, many splits: 64 workgroups per tile keep the 150 MB of slabs streaming (16: 210 us at the headline shape)
    hipLaunchKernelGGL(reduce_full_kernel, dim3(ntm * ntn, 64), dim3(256), 0, st, slabs, nsplit, ntm * ntn, ntn, out, ldo);
}

__global__ void reduce_rows_kernel(const double* __restrict__ partial, int nsplit, int64_t n, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0;
    for (int sp = 0; sp < nsplit; ++sp) s += partial[(int64_t)sp * n + i];
    out[i] = s;
}
void reduce_rows(const double* partial, int nsplit, int64_t n, double* out, hipStream_t st) {
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, partial, nsplit, n, out);
}

// single workgroup: fixed-order tree over block partials
__global__ __launch_bounds__(256) void reduce_scalars_kernel(const double* __restrict__ partial, int nblocks, int width,
                                                             double* __restrict__ scalars, int slot0) {
    __shared__ double red[256];
    for (int k = 0; k < width; ++k) {
        double s = 0;
        for (int b = threadIdx.x; b < nblocks; b += 256) s += partial[(int64_t)b * width + k];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int m = 128; m >= 1; m >>= 1) {
            if (threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
            __syncthreads();
        }
        if (threadIdx.x == 0) scalars[slot0 + k] = red[0];
        __syncthreads();
    }
}
// many partials of one scalar (one per workgroup of the apply product: 6.6e4 at the headline shape, 108 us for the single
// workgroup above): 128 workgroups first sum a contiguous chunk each, in a fixed order, into the chunk's first slot
__global__ __launch_bounds__(256) void reduce_chunks_kernel(double* __restrict__ partial, int n, int chunk) {
    __shared__ double red[256];
    const int64_t b0 = (int64_t)blockIdx.x * chunk;
    const int len = b0 + chunk <= n ? chunk : (int)(n - b0);
    double s = 0;
    for (int b = threadIdx.x; b < len; b += 256) s += partial[b0 + b];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if (threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0 && len > 0) partial[b0] = red[0];
}
__global__ __launch_bounds__(128) void reduce_strided_kernel(const double* __restrict__ partial, int nchunks, int chunk,
                                                             double* __restrict__ scalars, int slot0) {
    __shared__ double red[128];
    red[threadIdx.x] = (int)threadIdx.x < nchunks ? partial[(int64_t)threadIdx.x * chunk] : 0.0;
    __syncthreads();
    for (int m = 64; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) scalars[slot0] = red[0];
}
void reduce_scalars(double* partial, int nblocks, int width, double* scalars, int slot0, hipStream_t st) {
    if (width == 1 && nblocks > 8192) {                        // the partials are scratch: the first stage works in place
        const int chunk = (nblocks + 127) / 128, nchunks = (nblocks + chunk - 1) / chunk;
        hipLaunchKernelGGL(reduce_chunks_kernel, dim3(nchunks), dim3(256), 0, st, partial, nblocks, chunk);
        hipLaunchKernelGGL(reduce_strided_kernel, dim3(1), dim3(128), 0, st, partial, nchunks, chunk, scalars, slot0);
        return;
    }
    hipLaunchKernelGGL(reduce_scalars_kernel, dim3(1), dim3(256), 0, st, partial, nblocks, width, scalars, slot0);
}

__global__ __launch_bounds__(256) void sumsq_kernel(const double* __restrict__ y, int64_t n, double* __restrict__ partial) {
    __shared__ double red[256];
    double s = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += y[i] * y[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if (threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
void sum_squares(const double* y, int64_t n, double* scalars, int slot, double* scratch, hipStream_t st) {
    const int nb = 256;
    hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, st, y, n, scratch);
    reduce_scalars(scratch, nb, 1, scalars, slot, st);
}

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
