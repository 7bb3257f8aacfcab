// K x K stage (fp64): A = G + lam I, blocked Cholesky with L^{-1} formed beside it (augmented factorisation),
// B = Li^T Li, alpha, log det, and the K x K cotangent Abar of the backward pass.
// Replaces Theano's Cholesky / MatrixInverse ops (SCFGP/SCFGP.py:105-110) and the
// linear-algebra part of TT.grad (:129).  All matrices are Kp x Kp (Kp % 128 == 0)
// with the padding block kept at identity, so no kernel needs edge handling.
#include "kernels.h"
#include "tile_engine.h"

#ifndef SCFGP_KSTAGE_BK
#define SCFGP_KSTAGE_BK 16   // measured (profiles/r02_tuning.md): 32 and 64 are slower for the inverse's doubling levels
#endif
#ifndef SCFGP_KSTAGE_WGM
#define SCFGP_KSTAGE_WGM 2
#define SCFGP_KSTAGE_WGN 2
#endif
typedef TileCfg<double, 64, 64, SCFGP_KSTAGE_BK, SCFGP_KSTAGE_WGM, SCFGP_KSTAGE_WGN> KCfg;
#define SMEM_DECL extern __shared__ __attribute__((aligned(16))) char smem_raw[]

struct GemmArgs {
    const double* A; const double* B; double* C;
    int64_t lda, ldb, ldc;
    int M, N, K;
    double alpha, beta;
    int tri;        // 1: skip tiles strictly above the diagonal; 2: also store every tile's transpose (C symmetric)
    int kskip;      // contraction range by the operands' triangular shape: 0 all of K; 1 both Li^T-shaped: from max(m0, n0);
                    // 2 B lower triangular (B[k][n] = 0 for k < n): from n0; 3 A lower triangular (A[m][k] = 0 for k > m): up to m0 + 64;
                    // 4 A stored [k][m] lower triangular (A[k][m] = 0 for k < m): from m0
};

// C[m][n] = alpha * sum_k Aop[k][m] Bop[k][n] + beta * C[m][n]
//   TRA=false: A stored [k][m] (Nat)   TRA=true: A stored [m][k]
template <bool TRA, bool TRB>
__device__ __forceinline__ void gemm64_body(const GemmArgs& a, int tm, int tn, double* smem) {
    typedef KCfg Cfg;
    if (tm * Cfg::BM >= a.M || tn * Cfg::BN >= a.N) return;
    if (a.tri && tn > tm) return;
    const int k0 = a.kskip == 1 ? (tm > tn ? tm : tn) * Cfg::BM : (a.kskip == 2 ? tn * Cfg::BN : (a.kskip == 4 ? tm * Cfg::BM : 0));
    const int k1 = a.kskip == 3 && (tm + 1) * Cfg::BM < a.K ? (tm + 1) * Cfg::BM : a.K;
    const int nkt = (k1 - k0) / Cfg::BK;
    v4d acc[Cfg::TM][Cfg::TN];
    acc_zero<Cfg>(acc);
    if (TRA) {
        TrLoader<double, double, Cfg::BM, Cfg::BK, Cfg::LDA, Cfg::THREADS, false, false, 3> la(a.A + (int64_t)tm * Cfg::BM * a.lda + k0, a.lda, threadIdx.x);
        if (TRB) {
            TrLoader<double, double, Cfg::BN, Cfg::BK, Cfg::LDB, Cfg::THREADS, false, false, 3> lb(a.B + (int64_t)tn * Cfg::BN * a.ldb + k0, a.ldb, threadIdx.x);
            tile_mainloop_deep3<Cfg>(la, lb, nkt, acc, smem);
        } else {
            NatLoader<double, double, Cfg::BN, Cfg::BK, Cfg::LDB, Cfg::THREADS, false, false, false, 3> lb(a.B + (int64_t)k0 * a.ldb + tn * Cfg::BN, a.ldb, threadIdx.x);
            tile_mainloop_deep3<Cfg>(la, lb, nkt, acc, smem);
        }
    } else {
        NatLoader<double, double, Cfg::BM, Cfg::BK, Cfg::LDA, Cfg::THREADS, false, false, false, 3> la(a.A + (int64_t)k0 * a.lda + tm * Cfg::BM, a.lda, threadIdx.x);
        if (TRB) {
            TrLoader<double, double, Cfg::BN, Cfg::BK, Cfg::LDB, Cfg::THREADS, false, false, 3> lb(a.B + (int64_t)tn * Cfg::BN * a.ldb + k0, a.ldb, threadIdx.x);
            tile_mainloop_deep3<Cfg>(la, lb, nkt, acc, smem);
        } else {
            NatLoader<double, double, Cfg::BN, Cfg::BK, Cfg::LDB, Cfg::THREADS, false, false, false, 3> lb(a.B + (int64_t)k0 * a.ldb + tn * Cfg::BN, a.ldb, threadIdx.x);
            tile_mainloop_deep3<Cfg>(la, lb, nkt, acc, smem);
        }
    }
    AccCoord<Cfg> co;
#pragma unroll
    for (int t1 = 0; t1 < Cfg::TM; ++t1)
#pragma unroll
        for (int r = 0; r < Cfg::MTr::NACC; ++r) {
            double* c = a.C + (int64_t)(tm * Cfg::BM + co.row(t1, r)) * a.ldc + tn * Cfg::BN;
#pragma unroll
            for (int t2 = 0; t2 < Cfg::TN; ++t2) {
                const double v = a.alpha * acc[t1][t2][r];
                c[co.col(t2)] = a.beta == 0.0 ? v : v + a.beta * c[co.col(t2)];
                if (a.tri == 2 && tm != tn) a.C[(int64_t)(tn * Cfg::BN + co.col(t2)) * a.ldc + tm * Cfg::BM + co.row(t1, r)] = v;
            }
        }
}

template <bool TRA, bool TRB>
__global__ __launch_bounds__(KCfg::THREADS) void gemm64_kernel(GemmArgs a) {
    SMEM_DECL;
    gemm64_body<TRA, TRB>(a, blockIdx.y, blockIdx.x, reinterpret_cast<double*>(smem_raw));
}

template <bool TRA, bool TRB>
static void gemm64(const GemmArgs& a, hipStream_t st) {
    if (a.M <= 0 || a.N <= 0) return;
    allow_big_lds(gemm64_kernel<TRA, TRB>, KCfg::LDS_BYTES);
    hipLaunchKernelGGL((gemm64_kernel<TRA, TRB>), dim3(a.N / KCfg::BN, a.M / KCfg::BM), dim3(KCfg::THREADS), KCfg::LDS_BYTES, st, a);
}

// ---------------------------------------------------------------------------
// 64 x 64 diagonal block: Cholesky and triangular inverse by one 4-wave workgroup, the matrix in LDS.
//   Cholesky: left-to-right in four 16-column panels.  A panel (rows pb..63 x 16 columns) is factored by
//   wave 0 alone with one matrix row per lane in registers: the pivot and the multipliers L[k][j] reach the
//   other lanes as v_readlane broadcasts (scalar registers; the ds_bpermute form of __shfl cost a 100-cycle LDS
//   round trip per column on this 64-step critical path), so the 16 sequential column steps need neither LDS nor
//   barriers.  The rest of the matrix then gets one rank-16 update as fp64 MFMA tiles (16 x 16 x 4, four k-steps).
//   Inverse: the four 16 x 16 diagonal blocks by forward substitution in registers (one column per thread,
//   chains of <= 120 FMAs), then two doubling levels X21 = -X22 (L21 X11), again as MFMA tiles.
//   This kernel sits 34 times on the critical path of the K-stage at K = 2112.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double lane_bcast(double x, int lane) {           // lane must be wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane), hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}
// one 16 x 16 tile  C = sum_k A[i][k] B[k][j], k < 4*NS, on one wave:  A[i][k] = pa[i*lda + k], B[k][j] = pb[k*ldbk + j*ldbj]
template <int NS>
__device__ __forceinline__ v4d mfma_tile(const double* pa, int lda, const double* pb, int ldbk, int ldbj, v4d acc, double sign) {
    const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
#pragma unroll
    for (int s = 0; s < NS; ++s)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sign * pa[i * lda + 4 * s + q], pb[(4 * s + q) * ldbk + i * ldbj], acc, 0, 0, 0);
    return acc;
}
// Y (64 x 64, as 16 x 16 MFMA tiles: wave w owns tile row w, registers acc[b] = tile (w, b)) = X . W^T with W lower
// triangular, both in LDS with pitch LD: Y[i][j] = sum_{k <= j} X[i][k] W[j][k]
// (k-steps outermost: the X fragment is read once per step and the four column tiles are independent MFMA chains, so
// the LDS latency of a step hides under the other tiles' MFMAs; tile-by-tile it was exposed 64 times)
template <int LD>
__device__ __forceinline__ void block_xwt(const double* sX, const double* sW, v4d (&acc)[4]) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = v4d{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const double x = sX[(16 * wave + i) * LD + 4 * s + q];
#pragma unroll
        for (int b = s / 4; b < 4; ++b)                               // W[j][k] = 0 for k > j: tile b needs k < 16 (b + 1)
            acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, sW[(16 * b + i) * LD + 4 * s + q], acc[b], 0, 0, 0);
    }
}
// tiles (w, b) of a 64 x 64 block between the MFMA register layout and memory with pitch `pitch`
template <typename P>
__device__ __forceinline__ void block_store(P* dst, int64_t pitch, const v4d (&acc)[4]) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, ci = lane & 15, cq = lane >> 4;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(int64_t)(16 * wave + cq + 4 * r) * pitch + 16 * b + ci] = acc[b][r];
}
__device__ __forceinline__ void block_load(const double* src, int64_t pitch, v4d (&acc)[4]) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, ci = lane & 15, cq = lane >> 4;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[b][r] = src[(int64_t)(16 * wave + cq + 4 * r) * pitch + 16 * b + ci];
}
// acc (tiles (w, b)) -= P . Q^T, P and Q 64 x 64 in LDS
template <int LD>
__device__ __forceinline__ void block_sub_pqt(const double* sP, const double* sQ, v4d (&acc)[4]) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const double x = -sP[(16 * wave + i) * LD + 4 * s + q];
#pragma unroll
        for (int b = 0; b < 4; ++b)
            acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, sQ[(16 * b + i) * LD + 4 * s + q], acc[b], 0, 0, 0);
    }
}
// acc (tiles (w, b)) += P . Q^T
template <int LD>
__device__ __forceinline__ void block_add_pqt(const double* sP, const double* sQ, v4d (&acc)[4]) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const double x = sP[(16 * wave + i) * LD + 4 * s + q];
#pragma unroll
        for (int b = 0; b < 4; ++b)
            acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, sQ[(16 * b + i) * LD + 4 * s + q], acc[b], 0, 0, 0);
    }
}
__device__ __forceinline__ void block_to_lds(const double* src, int64_t pitch, double* dst, int LD) {
    for (int e = threadIdx.x; e < 64 * 64; e += 256) dst[(e / 64) * LD + e % 64] = src[(int64_t)(e / 64) * pitch + e % 64];
}

// Diagnostic build (-DSCFGP_TRACE): workgroup 0 of every Cholesky step stamps its phases (s_memrealtime, 100 MHz)
#ifdef SCFGP_TRACE
__device__ unsigned long long g_ctrace[64][16];
#define CSTAMP(k) do { if (threadIdx.x == 0 && p < 64) g_ctrace[p][k] = __builtin_amdgcn_s_memrealtime(); } while (0)
int64_t chol_trace_read(void* host, int64_t max_bytes) {
    const int64_t n = max_bytes < (int64_t)sizeof(g_ctrace) ? max_bytes : (int64_t)sizeof(g_ctrace);
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ctrace), n) == hipSuccess ? n : -2;
}
#else
#define CSTAMP(k)
int64_t chol_trace_read(void*, int64_t) { return -1; }
#endif

// One launch per 64-column step p of the blocked Cholesky (nb = Kp / 64 steps, ONE dependent launch each, plus a last one):
//   workgroup 0      the diagonal block of step p.  It brings the block up to date itself -- D = A[p][p] - L_p L_p^T with
//                    L_p = A[p][p-1] Inv(p-1)^T, the only part of step p-1's update it depends on -- then factors it,
//                    D = L L^T, and inverts L:  Lm[p][p] = L, Li[p][p] = L^-1.
//   workgroups 1..   the trailing update of step q = p-1, one 64 x 64 tile each, with the panel solve folded in:
//                    L_i = A[i][q] Inv(q)^T, L_j likewise, A[i][j] -= L_i L_j^T.  The working matrix A keeps its unsolved panel
//                    columns (every tile re-derives the L blocks it needs from them).
//   The INVERSE and B = A^-1 ride along (no separate triangular-inverse or Li^T Li launches): the factorisation of the
//   augmented matrix [[A, I], [I, 0]] gives [L; L^-T] and the Schur complement -A^-1, and its extra rows are updated by the same
//   tile code -- row e of the identity part lives in the unused upper blocks A[e][j], e < j, zeroed beforehand (its own diagonal
//   block I is implicit).  Per step q: identity-row tiles (e <= q, j > q) A[e][j] -= L_e L_j^T with L_e = A[e][q] Inv(q)^T
//   (e = q: Inv(q)^T itself); and the tiles (q >= e >= e') of B += L_e L_e'^T (first touched at step q = e; the mirror tile is
//   rewritten with it), of which the tiles (q, e < q) also store the finished block Li[q][e] = L_e^T of the inverse.  Nobody on the critical path waits
//   for them: they fill the shadow of workgroup 0.  Only the nbk = ceil(K / 64) live steps run: the padding blocks of A are
//   the identity and factor to themselves (Li and B carry 1 on the padding diagonal).
// So the critical path of the whole K x K factor-and-invert is the chain of diagonal blocks alone -- one ~22 us workgroup per
// step -- with the O(K^3) update work of the previous step running beside it.
__global__ __launch_bounds__(256) void chol_step_kernel(double* A, double* Lm, double* Li, double* Bm, int64_t ld, int p, int nb, int* flag) {
    constexpr int NB = 64, LD = NB + 1, PB = 16;
    __shared__ double sL[NB * LD];                                     // 2 x 33 KB + 9 KB: two workgroups per CU
    __shared__ double sI[NB * LD];
    __shared__ double sT[32 * 33];
    __shared__ double sD[NB];                                          // 1 / L[j][j]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ci = lane & 15, cq = lane >> 4;                          // MFMA C/D map (fp64): column ci, rows cq + 4 r
    v4d acc[4];
    if (blockIdx.x > 0) {
        // ---- work item of step q = p - 1: trailing tile (i >= j >= p), identity-row tile (i <= q < j) or tile (q >= i >= j) of B
        //      (the tiles (q, j < q) of B hold L_j = the finished block Li[q][j]^T of the inverse and store it too)
        const int q = p - 1, n = nb - p, T = n * (n + 1) / 2, U = p * n;
        int t = blockIdx.x - 1, i, j;
        bool b_tile = false;
        const auto tri_decode = [](int t, int& ti, int& tj) {
            ti = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);
            while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
            while (ti * (ti + 1) / 2 > t) --ti;
            tj = t - ti * (ti + 1) / 2;
        };
        if (t < T) {
            int ti, tj; tri_decode(t, ti, tj);
            i = p + ti; j = p + tj;
            if (i == p && j == p) return;                              // workgroup 0 updates and factors this block itself
        } else if (t < T + U) {
            t -= T; i = t / n; j = p + t % n;
        } else {
            tri_decode(t - T - U, i, j); b_tile = true;                // q >= i >= j >= 0
        }
        // every block this item needs is requested before anything is waited for (the launch starts cold)
        const double* gw = Li + ((int64_t)q * ld + q) * NB;            // W = Inv(q)
        const double* gi = A + ((int64_t)i * ld + q) * NB;
        const double* gj = A + ((int64_t)j * ld + q) * NB;
        double* c = (b_tile ? Bm : A) + ((int64_t)i * ld + j) * NB;
        const bool unit_i = i == q, unit_j = j == q;                   // row q of the identity part: its block (q, q) is I, L = W^T
        const bool first_b = b_tile && i == q;                         // B[i][j] is first touched at step q = i
        v2d rw[8], ri[8], rj[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = (tid + u * 256) * 2;
            const int64_t o = (int64_t)(e / NB) * ld + e % NB;
            rw[u] = *reinterpret_cast<const v2d*>(gw + o);
            if (!unit_i) ri[u] = *reinterpret_cast<const v2d*>(gi + o);
            if (i != j && !unit_j) rj[u] = *reinterpret_cast<const v2d*>(gj + o);
        }
        v4d accc[4];
        if (!first_b) block_load(c, ld, accc);
        else {
#pragma unroll
            for (int b = 0; b < 4; ++b) accc[b] = v4d{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = (tid + u * 256) * 2, o = (e / NB) * LD + e % NB;
            sL[o] = rw[u][0]; sL[o + 1] = rw[u][1];
            if (!unit_i) { sI[o] = ri[u][0]; sI[o + 1] = ri[u][1]; }
        }
        __syncthreads();
        const auto w_transposed = [&](v4d (&a)[4]) {                   // W^T straight into the accumulator layout
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) a[b][r] = sL[(16 * b + ci) * LD + 16 * wave + cq + 4 * r];
        };
        if (unit_i) w_transposed(acc);
        else block_xwt<LD>(sI, sL, acc);                               // L_i (registers)
        v4d accj[4];
        if (i != j) {
            if (unit_j) w_transposed(accj);
            else {
                __syncthreads();
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = (tid + u * 256) * 2, o = (e / NB) * LD + e % NB;
                    sI[o] = rj[u][0]; sI[o + 1] = rj[u][1];
                }
                __syncthreads();
                block_xwt<LD>(sI, sL, accj);                           // L_j
            }
            if (b_tile && unit_i) {                                    // Li[q][j] = L_j^T, j < q
                double* d = Li + ((int64_t)q * ld + j) * NB;
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) d[(int64_t)(16 * b + ci) * ld + 16 * wave + cq + 4 * r] = accj[b][r];
            }
        }
        __syncthreads();
        block_store(sL, LD, acc);                                      // W is done with: L_i takes its place, L_j stays in sI's
        if (i != j) block_store(sI, LD, accj);
        __syncthreads();
        if (b_tile) {
            block_add_pqt<LD>(sL, i != j ? sI : sL, accc);
            block_store(c, ld, accc);
            if (i != j) {                                              // B is symmetric: the mirror tile
                double* ct = Bm + ((int64_t)j * ld + i) * NB;
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ct[(int64_t)(16 * b + ci) * ld + 16 * wave + cq + 4 * r] = accc[b][r];
            }
            return;
        }
        block_sub_pqt<LD>(sL, i != j ? sI : sL, accc);
        block_store(c, ld, accc);
        return;
    }
    if (p >= nb) return;                                               // the last launch carries only the inverse blocks of step nb - 1
    // ---- diagonal block of step p
    CSTAMP(0);
    double* a = A + ((int64_t)p * ld + p) * NB;
    if (p > 0) {
        // all three blocks are fetched at once (the launch starts cold: every dependent fetch is ~2.5 us on this chain):
        // A[p][p-1] and Inv(p-1) as 16-byte vectors on their way to LDS, A[p][p] straight into the accumulator layout
        const double* gx = A + ((int64_t)p * ld + (p - 1)) * NB;
        const double* gw = Li + ((int64_t)(p - 1) * ld + (p - 1)) * NB;
        v2d rx[8], rw[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = (tid + i * 256) * 2;
            rx[i] = *reinterpret_cast<const v2d*>(gx + (int64_t)(e / NB) * ld + e % NB);
            rw[i] = *reinterpret_cast<const v2d*>(gw + (int64_t)(e / NB) * ld + e % NB);
        }
        v4d accd[4];
        block_load(a, ld, accd);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = (tid + i * 256) * 2, o = (e / NB) * LD + e % NB;
            sI[o] = rx[i][0]; sI[o + 1] = rx[i][1];
            sL[o] = rw[i][0]; sL[o + 1] = rw[i][1];
        }
        __syncthreads();
        CSTAMP(12);
        block_xwt<LD>(sI, sL, acc);                                    // L_p = A[p][p-1] Inv(p-1)^T
        __syncthreads();
        block_store(sI, LD, acc);
        __syncthreads();
        CSTAMP(13);
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[b] = accd[b];
        CSTAMP(14);
        block_sub_pqt<LD>(sI, sI, acc);                                // D = A[p][p] - L_p L_p^T
        __syncthreads();
        CSTAMP(15);
        block_store(sL, LD, acc);
        for (int e = tid; e < NB * NB; e += 256) sI[(e / NB) * LD + e % NB] = 0.0;
    } else {
        for (int e = tid; e < NB * NB; e += 256) {
            sL[(e / NB) * LD + e % NB] = a[(int64_t)(e / NB) * ld + e % NB];
            sI[(e / NB) * LD + e % NB] = 0.0;
        }
    }
    double* lm = Lm + ((int64_t)p * ld + p) * NB;
    double* li = Li + ((int64_t)p * ld + p) * NB;
    bool bad = false;
    CSTAMP(1);
    // rank-16 update of one 16 x 16 tile with the panel at column pp:  L[r0+i][c0+j] -= sum_c L[r0+i][pp+c] L[c0+j][pp+c]
    const auto tile_update = [&](int r0, int c0, int pp) {
        v4d acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = sL[(r0 + cq + 4 * r) * LD + c0 + ci];
        acc = mfma_tile<4>(sL + r0 * LD + pp, LD, sL + c0 * LD + pp, 1, LD, acc, -1.0);
#pragma unroll
        for (int r = 0; r < 4; ++r) sL[(r0 + cq + 4 * r) * LD + c0 + ci] = acc[r];
    };
    // Look-ahead: the update with panel pb-16 is split.  Wave 0 applies it to the NEXT panel's columns only and factors
    // them at once; waves 1..3 apply it to the columns right of that meanwhile.  One barrier per panel.
    for (int pb = 0; pb < NB; pb += PB) {
        __syncthreads();
        const int nt = (NB - pb) / PB;                                 // 16-row tiles from row pb down
        if (wave == 0) {
            if (pb > 0)
                for (int ti = 0; ti < nt; ++ti) tile_update(pb + PB * ti, pb, pb - PB);
            const int row = pb + lane;                                 // lanes past the last row idle along
            double v[PB], invs[PB];
#pragma unroll
            for (int c = 0; c < PB; ++c) v[c] = row < NB ? sL[row * LD + pb + c] : 0.0;
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const double d = lane_bcast(v[j], j);                  // pivot: row pb+j is lane j
                bad |= !(d > 0.0);
                // 1/sqrt(d): hardware estimate + two Newton steps (the library sqrt and divide are ~450 dependent
                // cycles per column on this 64-step critical path)
                double inv = __builtin_amdgcn_rsq(d);
                inv = fma(0.5 * inv, fma(-d * inv, inv, 1.0), inv);
                inv = fma(0.5 * inv, fma(-d * inv, inv, 1.0), inv);
                invs[j] = inv;                                         // d, and with it inv, is the same in every lane
                v[j] *= inv;                                           // lane j: d * inv = L[j][j]; entries above the diagonal are never used
#pragma unroll
                for (int k = j + 1; k < PB; ++k) v[k] = fma(-v[j], lane_bcast(v[j], k), v[k]);    // L[pb+k][pb+j] lives in lane k
            }
#pragma unroll
            for (int c = 0; c < PB; ++c)
                if (row < NB) sL[row * LD + pb + c] = v[c];
            if (lane == 0) {
#pragma unroll
                for (int c = 0; c < PB; ++c) sD[pb + c] = invs[c];
            }
            CSTAMP(2 + pb / PB * 2);
            CSTAMP(3 + pb / PB * 2);
        } else if (pb > 0) {
            // tiles (ti >= tj >= 1) of the region that starts at row / column pb
            int t = 0;
            for (int tj = 1; tj < nt; ++tj)
                for (int ti = tj; ti < nt; ++ti, ++t)
                    if (t % 3 == wave - 1) tile_update(pb + PB * ti, pb + PB * tj, pb - PB);
        }
    }
    if (bad && tid == 0) *flag = 1;                                    // not positive definite (or NaN)
    __syncthreads();
    // ---- inverse, level 0: the four 16 x 16 diagonal blocks, thread = (block, column)
    if (tid < NB) {
        // column c of the inverse of block o by forward substitution, column-oriented: once x[k] is known every later row
        // takes its term at once (16 short independent chains instead of one chain of 120 FMAs)
        const int o = (tid / PB) * PB, c = tid % PB;
        double sres[PB];
#pragma unroll
        for (int r = 0; r < PB; ++r) sres[r] = r == c ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < PB; ++k) {
            const double xk = k >= c ? sres[k] * sD[o + k] : 0.0;                   // x[k] = 0 above the diagonal
            sI[(o + k) * LD + o + c] = xk;
#pragma unroll
            for (int r = k + 1; r < PB; ++r) sres[r] = fma(-sL[(o + r) * LD + o + k], xk, sres[r]);
        }
    }
    __syncthreads();
    CSTAMP(10);
    // ---- doubling levels h = 16, 32: for each pair of diagonal blocks T = L21 X11, X21 = -X22 T (sI is zero above
    //      its diagonal blocks' diagonals, so the triangular factors multiply as full tiles)
    for (int h = PB; h < NB; h *= 2) {
        const int npair = NB / (2 * h), tpp = (h / PB) * (h / PB), LT = h + 1;       // tiles per pair
        for (int t = wave; t < npair * tpp; t += 4) {
            const int q = t / tpp, ti = (t % tpp) / (h / PB), tj = t % (h / PB), o = q * 2 * h;
            v4d acc = {0, 0, 0, 0};
            acc = h == PB ? mfma_tile<4>(sL + (o + h + PB * ti) * LD + o, LD, sI + o * LD + o + PB * tj, LD, 1, acc, 1.0)
                          : mfma_tile<8>(sL + (o + h + PB * ti) * LD + o, LD, sI + o * LD + o + PB * tj, LD, 1, acc, 1.0);
#pragma unroll
            for (int r = 0; r < 4; ++r) sT[q * 16 * 17 + (PB * ti + cq + 4 * r) * LT + PB * tj + ci] = acc[r];
        }
        __syncthreads();
        for (int t = wave; t < npair * tpp; t += 4) {
            const int q = t / tpp, ti = (t % tpp) / (h / PB), tj = t % (h / PB), o = q * 2 * h;
            v4d acc = {0, 0, 0, 0};
            acc = h == PB ? mfma_tile<4>(sI + (o + h + PB * ti) * LD + o + h, LD, sT + q * 16 * 17 + PB * tj, LT, 1, acc, -1.0)
                          : mfma_tile<8>(sI + (o + h + PB * ti) * LD + o + h, LD, sT + q * 16 * 17 + PB * tj, LT, 1, acc, -1.0);
#pragma unroll
            for (int r = 0; r < 4; ++r) sI[(o + h + PB * ti + cq + 4 * r) * LD + o + PB * tj + ci] = acc[r];
        }
        __syncthreads();
    }
    for (int e = tid; e < NB * NB; e += 256) {
        const int i = e / NB, k = e % NB;
        lm[(int64_t)i * ld + k] = k <= i ? sL[i * LD + k] : 0.0;
        li[(int64_t)i * ld + k] = sI[i * LD + k];
    }
    CSTAMP(11);
}

// ---------------------------------------------------------------------------
// small vector / diagonal kernels
// ---------------------------------------------------------------------------
// summed exchange buffer 1 (packed lower 128 x 128 tiles of G) -> working matrix of the factorisation: lower 64 x 64 blocks of
// G + lam I (1 on the padding diagonal); the strictly upper 64 x 64 blocks := 0 -- they hold the identity rows of the augmented
// factorisation (chol_step_kernel).  Blocks of the padding rows / columns of Li and B: identity on the diagonal.
__global__ __launch_bounds__(256) void kstage_unpack_kernel(const double* __restrict__ packed, int B, double* __restrict__ A, double* __restrict__ Li,
                                                            double* __restrict__ Bm, int64_t ld, int K, int nbk, const Scal* __restrict__ sc) {
    const int t = blockIdx.x;
    int ti = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    while (ti * (ti + 1) / 2 > t) --ti;
    const int tj = t - ti * (ti + 1) / 2;
    const double lam = sc->lam;
    for (int e = blockIdx.y * 256 + threadIdx.x; e < B * B; e += gridDim.y * 256) {
        const double v = packed[(int64_t)t * B * B + e];
        const int i = ti * B + e / B, j = tj * B + e % B;
        A[(int64_t)i * ld + j] = j / 64 > i / 64 ? 0.0 : (i == j ? v + (i < K ? lam : 1.0) : v);
        if (ti != tj) A[(int64_t)j * ld + i] = 0.0;
        if (i / 64 >= nbk || j / 64 >= nbk) {                          // padding blocks: never touched by the step launches
            const double id = i == j ? 1.0 : 0.0;
            Li[(int64_t)i * ld + j] = id; Bm[(int64_t)i * ld + j] = id;
            if (ti != tj) { Li[(int64_t)j * ld + i] = 0.0; Bm[(int64_t)j * ld + i] = 0.0; }
        }
    }
}

// out[i] = sum_k M[i][k] v[k]      one wave per row
__global__ __launch_bounds__(256) void gemv_rows_kernel(const double* __restrict__ M, int64_t ld, const double* __restrict__ v,
                                                        double* __restrict__ out, int n) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    double s = 0;
    for (int k = lane; k < n; k += 64) s += M[(int64_t)i * ld + k] * v[k];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if (lane == 0) out[i] = s;
}
// scalars: logdet = 2 sum_{i<K} log L_ii ; g.alpha ; and the condition estimate's ingredients: min / max of L_ii^2 and
// max_j B_jj, B = A^-1.  max L_ii^2 <= max A_ii <= lam_max(A) and max B_jj <= 1 / lam_min(A), so their product is a LOWER
// bound of cond_2(A) -- and at least the usual diagonal ratio max L_ii^2 / min L_ii^2, since B_jj >= 1 / L_jj^2.
__global__ __launch_bounds__(256) void factor_scalars_kernel(const double* __restrict__ L, const double* __restrict__ B, int64_t ld, int K,
                                                             const double* __restrict__ g, const double* __restrict__ alpha,
                                                             double* __restrict__ scalars) {
    __shared__ double r1[256], r2[256], r3[256], r4[256], r5[256];
    double s1 = 0, s2 = 0, lmin = 1.0 / 0.0, lmax = 0, bmax = 0;
    for (int i = threadIdx.x; i < K; i += 256) {
        const double l = L[(int64_t)i * ld + i], b = B[(int64_t)i * ld + i];
        s1 += log(l); s2 += g[i] * alpha[i];
        lmin = fmin(lmin, l * l); lmax = fmax(lmax, l * l); bmax = fmax(bmax, b);
    }
    r1[threadIdx.x] = s1; r2[threadIdx.x] = s2; r3[threadIdx.x] = lmin; r4[threadIdx.x] = lmax; r5[threadIdx.x] = bmax;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if (threadIdx.x < m) {
            r1[threadIdx.x] += r1[threadIdx.x + m]; r2[threadIdx.x] += r2[threadIdx.x + m];
            r3[threadIdx.x] = fmin(r3[threadIdx.x], r3[threadIdx.x + m]);
            r4[threadIdx.x] = fmax(r4[threadIdx.x], r4[threadIdx.x + m]);
            r5[threadIdx.x] = fmax(r5[threadIdx.x], r5[threadIdx.x + m]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        scalars[R_LOGDET] = 2.0 * r1[0]; scalars[R_GTALPHA] = r2[0];
        scalars[R_LMIN2] = r3[0]; scalars[R_LMAX2] = r4[0]; scalars[R_BMAX] = r5[0];
    }
}

// Abar = B - BWB - (u a^T + a u^T)/2 + e^{-2a} a a^T  on the K x K block (padding: B - BWB = I)
__global__ __launch_bounds__(256) void abar_kernel(const double* __restrict__ B, const double* __restrict__ BWB, double* __restrict__ Abar,
                                                   const double* __restrict__ u, const double* __restrict__ al, int Kp,
                                                   const Scal* __restrict__ sc) {
    const double em2a = sc->em2a;
    const int64_t total = (int64_t)Kp * Kp;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int i = (int)(e / Kp), j = (int)(e % Kp);
        Abar[e] = B[e] - BWB[e] - 0.5 * (u[i] * al[j] + al[i] * u[j]) + em2a * al[i] * al[j];
    }
}
__global__ __launch_bounds__(256) void adjoint_vec_kernel(const double* __restrict__ Abar, int64_t ld, int K, int Kp,
                                                          const double* __restrict__ u, const double* __restrict__ al,
                                                          double* __restrict__ ut, const Scal* __restrict__ sc,
                                                          double* __restrict__ scalars) {
    __shared__ double r1[256];
    const double em2a = sc->em2a;
    double s = 0;
    for (int i = threadIdx.x; i < Kp; i += 256) {
        ut[i] = u[i] - 2.0 * em2a * al[i];
        if (i < K) s += Abar[(int64_t)i * ld + i];
    }
    r1[threadIdx.x] = s;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if (threadIdx.x < m) r1[threadIdx.x] += r1[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) scalars[R_TRABAR] = r1[0];
}

// bbar = d(N cost)/db = sum_n Phibar_n . phi_n with Phibar_n = p_n alpha + y_n ut + 2 q_n B phi_n + 2 Abar phi_n is
//      2 tr(Abar G) + ut^T (Phi^T y) + 2 sum_n q_n v_n + sum_n p_n mu_n :
// a K x K trace over the summed Gram (still in exchange buffer 1 as packed lower 128 x 128 tiles: G and Abar are symmetric, so an
// off-diagonal tile counts twice; diagonal tiles carry both triangles), a K-dot, and two row sums rowstats_kernel forms beside p
// and q (exchange buffer 2).  Neither Phi nor Phibar is needed, and the K x K part is computed once per evaluation on the sums
// over ranks instead of per output tile of the Phibar product.  part[t] = tile t's share of tr(Abar G).
__global__ __launch_bounds__(256) void trace_ag_kernel(const double* __restrict__ packed, const double* __restrict__ Abar, int64_t ld,
                                                       int K, double* __restrict__ part) {
    constexpr int B = 128;
    __shared__ double r1[256];
    const int t = blockIdx.x;
    int ti = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    while (ti * (ti + 1) / 2 > t) --ti;
    const int tj = t - ti * (ti + 1) / 2;
    double s = 0;
    for (int e = threadIdx.x; e < B * B; e += 256) {
        const int i = ti * B + e / B, j = tj * B + e % B;
        if (i < K && j < K) s += packed[(int64_t)t * B * B + e] * Abar[(int64_t)i * ld + j];
    }
    r1[threadIdx.x] = s;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if (threadIdx.x < m) r1[threadIdx.x] += r1[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[t] = (ti == tj ? 1.0 : 2.0) * r1[0];
}
// scalars[R_TRAG] = sum of the tile shares (fixed order), scalars[R_UTG] = ut^T g
__global__ __launch_bounds__(256) void bbar_sums_kernel(const double* __restrict__ part, int ntiles, const double* __restrict__ ut,
                                                        const double* __restrict__ g, int K, double* __restrict__ scalars) {
    __shared__ double r1[256], r2[256];
    double s = 0, d = 0;
    for (int t = threadIdx.x; t < ntiles; t += 256) s += part[t];
    for (int i = threadIdx.x; i < K; i += 256) d += ut[i] * g[i];
    r1[threadIdx.x] = s; r2[threadIdx.x] = d;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if (threadIdx.x < m) { r1[threadIdx.x] += r1[threadIdx.x + m]; r2[threadIdx.x] += r2[threadIdx.x + m]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { scalars[R_TRAG] = r1[0]; scalars[R_UTG] = r2[0]; }
}
// packed: the summed exchange buffer 1 (G); part: nts (nts + 1) / 2 doubles of scratch; call after kstage_adjoint (Abar, k.ut)
void kstage_bbar(const KStage& k, const double* packed, const double* Abar, double* part, hipStream_t st) {
    const int nts = k.Kp / 128, ntiles = nts * (nts + 1) / 2;
    hipLaunchKernelGGL(trace_ag_kernel, dim3(ntiles), dim3(256), 0, st, packed, Abar, (int64_t)k.Kp, k.K, part);
    hipLaunchKernelGGL(bbar_sums_kernel, dim3(1), dim3(256), 0, st, part, ntiles, k.ut, k.g, k.K, k.scalars);
}

// ---------------------------------------------------------------------------
// host drivers
// ---------------------------------------------------------------------------
// A (k.A: lower blocks of G + lam I, upper blocks zero) -> the diagonal blocks of L in k.T2, L^-1 in k.Li, A^-1 in k.B
static void cholesky_inverse_gram(const KStage& k, hipStream_t st) {
    const int nbk = (k.K + 63) / 64;                                   // live steps: the padding blocks are the identity
    for (int p = 0; p <= nbk; ++p) {
        // step p's diagonal block + the items of step p-1: trailing tiles (i >= j >= p), identity-row tiles (i < p <= j), tiles of B
        // (p (p + 1) / 2): nbk (nbk + 1) / 2 items whatever p
        const int n = nbk - p, items = p > 0 ? n * (n + 1) / 2 + p * n + p * (p + 1) / 2 : 0;
        hipLaunchKernelGGL(chol_step_kernel, dim3(1 + items), dim3(256), 0, st, k.A, k.T2, k.Li, k.B, (int64_t)k.Kp, p, nbk, k.flag);
    }
}

// packed: the summed exchange buffer 1 (lower 128 x 128 tiles of G, then Phi^T y and the scalars, which the caller copies)
void kstage_factor(const KStage& k, const double* packed, const Scal* sc, hipStream_t st) {
    const int Kp = k.Kp, nts = Kp / 128, nbk = (k.K + 63) / 64;
    const int64_t ld = Kp;
    hipLaunchKernelGGL(kstage_unpack_kernel, dim3(nts * (nts + 1) / 2, 16), dim3(256), 0, st, packed, 128, k.A, k.Li, k.B, ld, k.K, nbk, sc);
    cholesky_inverse_gram(k, st);
    // alpha = Li^T (Li g) = B g  (SCFGP.py:108-110); B is symmetric, so one coalesced row-dot GEMV
    hipLaunchKernelGGL(gemv_rows_kernel, dim3((Kp + 3) / 4), dim3(256), 0, st, k.B, ld, k.g, k.alpha, Kp);
    // beta = Li g (SCFGP.py:109): what the factor form of pass 2 multiplies C = Phi Li^T with to get mu = Phi alpha = C beta
    hipLaunchKernelGGL(gemv_rows_kernel, dim3((Kp + 3) / 4), dim3(256), 0, st, k.Li, ld, k.g, k.beta, Kp);
    hipLaunchKernelGGL(factor_scalars_kernel, dim3(1), dim3(256), 0, st, k.T2, k.B, ld, k.K, k.g, k.alpha, k.scalars);
}

// out[i] = sum_k M[k][i] v[k]  (M^T v; M lower triangular: k >= i only).  blockIdx.y cuts k into gridDim.y parts whose partial
// sums go to part[y][i]; the second kernel adds them in order (deterministic, no atomics).
__global__ __launch_bounds__(256) void gemv_cols_part_kernel(const double* __restrict__ M, int64_t ld, const double* __restrict__ v,
                                                             double* __restrict__ part, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int per = (n + gridDim.y - 1) / gridDim.y, k0 = blockIdx.y * per, k1 = k0 + per < n ? k0 + per : n;
    if (i >= n) return;
    double s = 0;
    for (int k = k0 > i ? k0 : i; k < k1; ++k) s += M[(int64_t)k * ld + i] * v[k];
    part[(int64_t)blockIdx.y * n + i] = s;
}
__global__ __launch_bounds__(256) void gemv_cols_sum_kernel(const double* __restrict__ part, int nparts, double* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double s = 0;
    for (int p = 0; p < nparts; ++p) s += part[(int64_t)p * n + i];
    out[i] = s;
}

// Factor form of pass 2 (SweepKernels::apply_c): the sweep delivers Mc = C^T diag(q) C and s = C^T p with C = Phi Li^T, so
//   B W B = Li^T Mc Li   (two K x K products, triangular ranges)      u = B Phi^T p = Li^T s
// McBWB: Mc on entry, B W B on return; k.h: s on entry; u goes to k.u.  Then as kstage_adjoint.
void kstage_adjoint_factor_form(const KStage& k, double* McBWB, double* Abar, const Scal* sc, hipStream_t st) {
    const int Kp = k.Kp;
    const int64_t ld = Kp;
    GemmArgs a1 = {McBWB, k.Li, k.T1, ld, ld, ld, Kp, Kp, Kp, 1.0, 0.0, 0, 2};       // T1 = Mc Li (Mc symmetric: stored [k][m] too)
    gemm64<false, false>(a1, st);
    GemmArgs a2 = {k.Li, k.T1, McBWB, ld, ld, ld, Kp, Kp, Kp, 1.0, 0.0, 2, 4};        // B W B = Li^T T1: lower tiles, mirrored
    gemm64<false, false>(a2, st);
    constexpr int PARTS = 32;
    hipLaunchKernelGGL(gemv_cols_part_kernel, dim3((Kp + 255) / 256, PARTS), dim3(256), 0, st, k.Li, ld, k.h, k.T1, Kp);
    hipLaunchKernelGGL(gemv_cols_sum_kernel, dim3((Kp + 255) / 256), dim3(256), 0, st, k.T1, PARTS, k.u, Kp);
    hipLaunchKernelGGL(abar_kernel, dim3(1024), dim3(256), 0, st, k.B, McBWB, Abar, k.u, k.alpha, Kp, sc);
    hipLaunchKernelGGL(adjoint_vec_kernel, dim3(1), dim3(256), 0, st, Abar, ld, k.K, Kp, k.u, k.alpha, k.ut, sc, k.scalars);
}

// BWB = V^T diag(q) V and u = B h = V^T p arrive ready from the row sweep (V = Phi B is resident), so the adjoint
// of A needs no K^3 work: Abar = B - BWB - (u a^T + a u^T)/2 + e^{-2a} a a^T.   k.h holds u.
void kstage_adjoint(const KStage& k, const double* BWB, double* Abar, const Scal* sc, hipStream_t st) {
    const int Kp = k.Kp;
    const int64_t ld = Kp;
    hipLaunchKernelGGL(abar_kernel, dim3(1024), dim3(256), 0, st, k.B, BWB, Abar, k.h, k.alpha, Kp, sc);
    hipLaunchKernelGGL(adjoint_vec_kernel, dim3(1), dim3(256), 0, st, Abar, ld, k.K, Kp, k.h, k.alpha, k.ut, sc, k.scalars);
}
