// Epilogues of the apply products (apply.hip: exact fp32 / fp64 tiles; apply_f16.hip: the three-term fp16 split's tiles): what happens to
// a workgroup's accumulators once its k loop is done.  EPI numbers as in apply.hip:
//   0  V = C and the row dots v_n = phi_n . V_n (mu_n beside them)     1  Phibar = 2 C + 2 q V + p alpha^T + y ut^T, in place over V
//   2  predict: row sums of C^2, nothing stored    3  factor form: C stored, row sums of C^2, mu = C . beta    4  plain store
#pragma once
#include "kernels.h"
#include "tile_cfgs.h"

// epilogues of the apply product: V and the row dots (EPI 0) or Phibar (EPI 1) from the accumulators
//   MU (EPI 0): also mupart[jtg][n] = sum_{j in tile} Phi[n][j] alpha[j] from the Phi values the row dot reads anyway
//   (the DMA-fed kernel has no operand values in registers for the loader-side dot); EPI 3 (C = Phi Li^T): the `alpha` argument is
//   BETA = Li Phi^T y and mupart = sum_j C[n][j] beta[j] -- the same mu_n = phi_n . alpha (alpha = Li^T beta) from the accumulators,
//   so the factor form's epilogue does not read Phi at all
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
// fp64 epilogue of the LDS-DMA tiles (64 x 32 wave tiles of two 16-column MFMA tiles): the B operand's rows were staged in the
// order that makes MFMA tile tn, lane column i output column 2 i + tn of the wave tile -- a lane holds two ADJACENT columns of
// each of its rows and moves V, Phi and Phibar 16 bytes at a time.  Same quantities as apply_epilogue below, all in fp64.
template <class Cfg, int EPI, bool MU>
__device__ __forceinline__ void apply_epilogue_vec2(
    const typename Cfg::MTr::acc_t (&acc)[Cfg::TM][Cfg::TN], const double* __restrict__ Phi, double* V,
    double* __restrict__ vpart, const double* __restrict__ p, const double* __restrict__ q, const double* __restrict__ y,
    const double* __restrict__ alpha, const double* __restrict__ ut, int K, int Kp, int64_t Np, int64_t rb, int cbase, int jtg,
    char* smem_raw, double* __restrict__ mupart, int tid) {
    static_assert(Cfg::TN == 2 && Cfg::MS == 16 && sizeof(typename Cfg::T) == 8 && (EPI == 0 || EPI == 1 || EPI == 3 || EPI == 4), "VEC2 layout");
    AccCoord<Cfg> co(tid);
    const int jg = cbase + co.wn0 + 2 * (co.lane & 15);         // first of this lane's two adjacent columns
    if constexpr (EPI == 4) {
#pragma unroll
        for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *reinterpret_cast<v2d*>(V + (rb * Cfg::BM + co.row(tm, r)) * Kp + jg) = v2d{acc[tm][0][r], acc[tm][1][r]};
    } else if constexpr (EPI == 0 || EPI == 3) {
        double* red = reinterpret_cast<double*>(smem_raw);
        double* red2 = red + Cfg::WGN * Cfg::BM;
        const int wn = (tid >> 6) % Cfg::WGN;
        double al[2], live[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) { live[k] = jg + k < K ? 1.0 : 0.0; al[k] = MU && jg + k < K ? alpha[jg + k] : 0.0; }
#pragma unroll
        for (int tm = 0; tm < Cfg::TM; ++tm) {
            v2d ph[4];
            int64_t off[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {                       // the four re-reads of Phi at once, not one dependent round trip per row
                off[r] = (rb * Cfg::BM + co.row(tm, r)) * Kp + jg;
                if (EPI == 0) ph[r] = *reinterpret_cast<const v2d*>(Phi + off[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = co.row(tm, r);
                const double c0 = acc[tm][0][r], c1 = acc[tm][1][r];
                *reinterpret_cast<v2d*>(V + off[r]) = v2d{c0, c1};
                double part, mup = 0;
                if (EPI == 3) part = fma(c0, c0, c1 * c1);
                else part = fma(ph[r][0] * live[0], c0, ph[r][1] * live[1] * c1);
                // EPI 3: mu_n = phi_n . alpha = (Li phi_n) . beta = C_n . beta (alpha = Li^T beta): from the accumulators, no re-read of Phi
                if (MU) mup = EPI == 3 ? fma(c0, al[0], c1 * al[1]) : fma(ph[r][0], al[0], ph[r][1] * al[1]);
                part = row16_sum(part);
                if ((co.lane & 15) == 0) red[wn * Cfg::BM + row] = part;
                if (MU) {
                    mup = row16_sum(mup);
                    if ((co.lane & 15) == 0) red2[wn * Cfg::BM + row] = mup;
                }
            }
        }
        __syncthreads();
        if (tid < Cfg::BM) {
            double s = 0, s2 = 0;
#pragma unroll
            for (int k = 0; k < Cfg::WGN; ++k) { s += red[k * Cfg::BM + tid]; if (MU) s2 += red2[k * Cfg::BM + tid]; }
            vpart[(int64_t)jtg * Np + rb * Cfg::BM + tid] = s;
            if (MU && mupart) mupart[(int64_t)jtg * Np + rb * Cfg::BM + tid] = s2;
        }
    } else {                                                    // EPI 1 (bbar is a K x K affair now: kstage_bbar)
        double al[2], u2[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) { al[k] = alpha[jg + k]; u2[k] = ut[jg + k]; }
        double* rowsc = reinterpret_cast<double*>(smem_raw);      // [BM][3]: 2 q, p, y of the tile's rows (LDS is free after the loop)
        for (int i = tid; i < Cfg::BM; i += Cfg::THREADS) {
            const int64_t n = rb * Cfg::BM + i;
            rowsc[3 * i] = 2.0 * q[n]; rowsc[3 * i + 1] = p[n]; rowsc[3 * i + 2] = y[n];
        }
        __syncthreads();
#pragma unroll
        for (int tm = 0; tm < Cfg::TM; ++tm) {
            v2d vv[4];
            int64_t off[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                off[r] = (rb * Cfg::BM + co.row(tm, r)) * Kp + jg;
                vv[r] = *reinterpret_cast<const v2d*>(V + off[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = co.row(tm, r);
                const double qn = rowsc[3 * row], pn = rowsc[3 * row + 1], yn = rowsc[3 * row + 2];
                v2d o;
#pragma unroll
                for (int k = 0; k < 2; ++k) o[k] = 2.0 * acc[tm][k][r] + qn * vv[r][k] + pn * al[k] + yn * u2[k];
                *reinterpret_cast<v2d*>(V + off[r]) = o;
            }
        }
    }
}
template <class Cfg, int EPI, bool MU = false, bool VEC4 = false>
__device__ __forceinline__ void apply_epilogue(
    const typename Cfg::MTr::acc_t (&acc)[Cfg::TM][Cfg::TN], const typename Cfg::T* __restrict__ Phi, typename Cfg::T* V,
    double* __restrict__ vpart, const double* __restrict__ p, const double* __restrict__ q, const double* __restrict__ y,
    const double* __restrict__ alpha, const double* __restrict__ ut, int K, int Kp, int64_t Np, int64_t rb, int cbase, int jtg,
    char* smem_raw, double* __restrict__ mupart = nullptr, int tid = (int)threadIdx.x) {
    typedef typename Cfg::T T;
    if constexpr (VEC4 && sizeof(T) == 8) {                        // the LDS-DMA tiles in fp64
        apply_epilogue_vec2<Cfg, EPI, MU>(acc, Phi, V, vpart, p, q, y, alpha, ut, K, Kp, Np, rb, cbase, jtg, smem_raw, mupart, tid);
        return;
    }
    AccCoord<Cfg> co(tid);
    if constexpr (VEC4 && sizeof(T) == 4) {
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
            const int wn = (tid >> 6) % Cfg::WGN;
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
                    if (EPI == 0) ph[r] = *reinterpret_cast<const v4f*>(Phi + off[r]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = co.row(tm, r);
                    const v4f c = v4f{acc[tm][0][r], acc[tm][1][r], acc[tm][2][r], acc[tm][3][r]};
                    *reinterpret_cast<v4f*>(V + off[r]) = c;
                    double part = 0, mup = 0;
                    if (EPI == 3) {
                        part = (double)(c[0] * c[0]) + (double)(c[1] * c[1]) + (double)(c[2] * c[2]) + (double)(c[3] * c[3]);
                        if (MU) mup = (double)c[0] * al[0] + (double)c[1] * al[1] + (double)c[2] * al[2] + (double)c[3] * al[3];     // C_n . beta
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) { part += (double)ph[r][k] * (double)c[k] * live[k]; if (MU) mup += (double)ph[r][k] * al[k]; }
                    }
                    part = row16_sum(part);
                    if ((co.lane & 15) == 0) red[wn * Cfg::BM + row] = part;
                    if (MU) {
                        mup = row16_sum(mup);
                        if ((co.lane & 15) == 0) red2[wn * Cfg::BM + row] = mup;
                    }
                }
            }
            __syncthreads();
            if (tid < Cfg::BM) {
                double s = 0, s2 = 0;
#pragma unroll
                for (int k = 0; k < Cfg::WGN; ++k) { s += red[k * Cfg::BM + tid]; if (MU) s2 += red2[k * Cfg::BM + tid]; }
                vpart[(int64_t)jtg * Np + rb * Cfg::BM + tid] = s;
                if (MU && mupart) mupart[(int64_t)jtg * Np + rb * Cfg::BM + tid] = s2;
            }
        } else {                                                    // EPI 1
            float alf[4], u4f[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { alf[k] = (float)alpha[jg + k]; u4f[k] = (float)ut[jg + k]; }
            // per-row scalars (2 q, p, y) of the tile's rows through LDS (free after the loop's last barrier), and the re-read of V
            // issued for the four rows of an accumulator row group at once: written row by row, the in-place store of a row
            // stood between the loads of the next one and its own (16 dependent round trips per tile; profiles/r03_tuning.md).
            // Phi is not read here any more: bbar = sum Phibar o Phi was its only use, and that sum needs neither matrix
            // (kernels_kstage.hip: kstage_bbar)
            float* rowsc = reinterpret_cast<float*>(smem_raw);        // [BM][3]
            for (int i = tid; i < Cfg::BM; i += Cfg::THREADS) {
                const int64_t n = rb * Cfg::BM + i;
                rowsc[3 * i] = (float)(2.0 * q[n]); rowsc[3 * i + 1] = (float)p[n]; rowsc[3 * i + 2] = (float)y[n];
            }
            __syncthreads();
#pragma unroll
            for (int tm = 0; tm < Cfg::TM; ++tm) {
                v4f vv[4];
                int64_t off[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    off[r] = (rb * Cfg::BM + co.row(tm, r)) * Kp + jg;
                    vv[r] = *reinterpret_cast<const v4f*>(V + off[r]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = co.row(tm, r);
                    // Phibar is stored in fp32: its four terms are combined in fp32 FMAs (one rounding per term instead of one at
                    // the end; the accumulator itself carries ~1e-7 of the product)
                    const float qn = rowsc[3 * row], pn = rowsc[3 * row + 1], yn = rowsc[3 * row + 2];
                    v4f o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = fmaf(qn, vv[r][k], fmaf(pn, alf[k], fmaf(yn, u4f[k], 2.0f * acc[tm][k][r])));
                    *reinterpret_cast<v4f*>(V + off[r]) = o;
                }
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
        const int wn = (tid >> 6) % Cfg::WGN;
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
                        // mu_n: predict (EPI 2) Phi* . alpha with the caller's alpha; factor form (EPI 3) C_n . beta
                        if (MU && cbase + co.col(tn) < K) mup += (EPI == 3 ? (double)c : (double)Phi[off + co.col(tn)]) * al[tn];
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
        if (tid < Cfg::BM) {
            double s = 0, s2 = 0;
#pragma unroll
            for (int k = 0; k < Cfg::WGN; ++k) { s += red[k * Cfg::BM + tid]; if (MU) s2 += red2[k * Cfg::BM + tid]; }
            vpart[(int64_t)jtg * Np + rb * Cfg::BM + tid] = s;
            if (MU && mupart) mupart[(int64_t)jtg * Np + rb * Cfg::BM + tid] = s2;
        }
    } else {                                                        // EPI 1: Phibar over V (bbar: kstage_bbar)
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
#pragma unroll
                    for (int tn = 0; tn < Cfg::TN; ++tn)
                        V[off + co.col(tn)] = fmaf(qf, V[off + co.col(tn)], fmaf(pf, alf[tn], fmaf(yf, utf[tn], 2.0f * acc[tm][tn][r])));
                } else {
#pragma unroll
                    for (int tn = 0; tn < Cfg::TN; ++tn) {
                        const int j = cbase + co.col(tn);
                        V[off + co.col(tn)] = (T)(2.0 * (double)acc[tm][tn][r] + qn * (double)V[off + co.col(tn)] + pn * alpha[j] + yn * ut[j]);
                    }
                }
            }
    }
}
