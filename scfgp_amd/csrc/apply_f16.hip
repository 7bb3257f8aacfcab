// Compute mode SCFGP_F16X3 (include/scfgp_hip.h; a labelled SECONDARY mode, never the headline): the two square apply products
// V = Phi B and Phibar = 2 Phi Abar + ... as a THREE-TERM fp16 split on the fp16 matrix pipe; everything else of the evaluation
// is fp32 mode's.  No reference counterpart (the reference is float64 throughout, SCFGP/SCFGP.py:95-96); what it must equal is
// fp32 mode, whose parity tier it runs under (tests/test_gpu_round5.py).
//
// The split.  x = (h + l) 2^-e with h = fp16(x 2^e), l = fp16(x 2^e - h) and ONE power-of-two scale per operand matrix that puts its
// largest entry in [2^14, 2^15): l then stays in fp16's normal range for every entry within 2^-18 of the largest, and h + l carries
// 22-23 bits.  Products of two fp16 values are exact in fp32, so  a b ~ ah bh + al bh + ah bl  accumulated in fp32 loses only the
// l.l term (2^-22 relative) against an exact-fp32 product: measured errors within 1.0-3.5x of fp32 mode's
// (tests/cpu_f16x3_emulation.py, profiles/r05_f16x3_emulation.txt).
//
// The operands.  An element of Phi stays 4 bytes: the packed pair (h, l) -- so the LDS image of the A panel, its DMA and its HBM
// traffic are exactly the fp32 tile's (apply.hip: apply_dma_kernel<float, ., 256>).  The small operand (B or Abar, K x K, cache
// resident) is stored per element as 8 bytes, the derived pairs (bh, bh) and (bl, 0): with a lane's A vector
// (ah0, al0, ah1, al1, ...) the two matrix instructions
//       a . (bh0, bh0, bh1, bh1, ...) = sum_k (ah_k + al_k) bh_k          a . (bl0, 0, bl1, 0, ...) = sum_k ah_k bl_k
// are the three terms.  v_mfma_f32_16x16x32_f16 takes 32 slots = 16 k: TWO instructions of 16 cycles per 16 x 16 output tile and
// 16-k stage where exact fp32 needs four v_mfma_f32_16x16x4_f32 of 32 cycles.  The k loop is the plain staged one (wait, barrier,
// fetch of stage s+2, fragment reads, MFMAs; compiler-scheduled): tools/f16x3_probe.hip measured it at 2.6x the fp32 loop before this
// file existed; the LDS array, not the matrix pipe, is what it leans on (192 B per lane and stage against 32 MFMAs of 16 cycles).
#include "apply_epilogue.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

namespace {
constexpr int BM = 256, ROWA = 64, ROWB = 128;                              // bytes of one row's 16 k: packed pairs / derived pairs
// tile widths as in fp32 mode's launch plan (apply.hip): 256 (16 waves), 128 (8 waves: an odd 128-column block, the thin last round's
// row blocks), 64 (4 waves: the ragged remainder, K = 2112 = 8 x 256 + 64); wave tiles 64 x 64 throughout
template <int BN_> struct F16Tile {
    static constexpr int BN = BN_, WAVES = 4 * (BN / 64), STAGE = BM * ROWA + BN * ROWB, STAGES = 3, LDS_BYTES = STAGES * STAGE,
                         DPW = STAGE / 1024 / WAVES, WAVES_PER_EU = WAVES == 4 ? 2 : 4;
    static_assert(STAGE / 1024 % WAVES == 0 && LDS_BYTES <= 160 * 1024, "whole DMA instructions per wave; the ring fits the LDS");
    typedef TileCfg<float, BM, BN, 16, 4, BN / 64, 16, true> Cfg;           // the fp32 tile's accumulator map (apply_epilogue.h)
};
// position swizzles of the two images (apply.hip): 64-byte rows f[(x >> 2) & 3], f = (0, 2, 3, 1); 128-byte rows f[(x >> 1) & 7]
__device__ __forceinline__ int swz4(int x) { return (0x78 >> (2 * ((x >> 2) & 3))) & 3; }
__device__ __forceinline__ int swz8(int x) { return (int)((0x6BEB08u >> (3 * ((x >> 1) & 7))) & 7); }
}

// One 256 x BN tile: column tile jt of the launch, row block rb0 + wid / njt.  Phi16: Np x Kp packed pairs; B16: Kp rows (= output
// columns) of Kp derived elements, 128 bytes per 16 k: [16 x (bh, bh) | 16 x (bl, 0)]; scale[0] = 2^-(e_Phi + e_B).
template <int EPI, int BN>
__global__ __launch_bounds__((64 * F16Tile<BN>::WAVES))
__attribute__((amdgpu_waves_per_eu(F16Tile<BN>::WAVES_PER_EU, F16Tile<BN>::WAVES_PER_EU)))
void apply_f16_kernel(const float* __restrict__ Phi, const unsigned* __restrict__ Phi16, const char* __restrict__ B16, const float* __restrict__ scale,
                      float* V, double* __restrict__ vpart, const double* __restrict__ p, const double* __restrict__ q,
                      const double* __restrict__ y, const double* __restrict__ alpha, const double* __restrict__ ut,
                      int K, int Kp, int64_t Np, int njt, double* __restrict__ mu, int col0, int slot0, int64_t rb0) {
    typedef F16Tile<BN> D;
    typedef typename D::Cfg Cfg;
    constexpr int DPW = D::DPW, STAGE = D::STAGE;
    SMEM_DECL;
    char* smem = smem_raw;
    const unsigned wid = xcd_remap(blockIdx.x, gridDim.x);
    const int jt = wid % njt;
    const int64_t rb = rb0 + wid / njt;
    const int cbase = col0 + jt * BN;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // DMA instruction t = DPW wave + u of a stage: 1 KiB of the stacked image -- t < 16: A rows 16 t .. (4 chunks each), else B
    // rows 8 (t - 16) .. (8 chunks each); chunk c of row x goes to position c ^ swz(x) (source-side swizzle, linear LDS writes).
    // B rows are staged in the order that leaves an MFMA lane 4 ADJACENT output columns (apply.hip): LDS row tn 16 + i of a wave
    // tile holds operand row 4 i + tn.
    const char* src[DPW]; int dst[DPW], adv[DPW];
#pragma unroll
    for (int u = 0; u < DPW; ++u) {
        const int t = DPW * wave + u;
        if (t < BM * ROWA / 1024) {
            const int x = 16 * t + lane / 4, c = (lane % 4) ^ swz4(x);
            src[u] = reinterpret_cast<const char*>(Phi16 + (rb * BM + x) * Kp) + (c << 4); adv[u] = ROWA;
        } else {
            const int xb = 8 * (t - BM * ROWA / 1024) + lane / 8, c = (lane % 8) ^ swz8(xb);
            const int xcol = (xb & ~63) + 4 * (xb & 15) + ((xb >> 4) & 3);
            src[u] = B16 + (int64_t)(cbase + xcol) * Kp * 8 + (c << 4); adv[u] = ROWB;
        }
        dst[u] = t * 1024;
    }
    const auto fetch = [&](int slot) {
#pragma unroll
        for (int u = 0; u < DPW; ++u) {
            __builtin_amdgcn_global_load_lds((gbl_void*)src[u], (lds_void*)(smem + slot * STAGE + dst[u]), 16, 0, 0);
            src[u] += adv[u];
        }
    };
    const int i = lane & 15, qg = lane >> 4;
    const int wm0 = (wave / Cfg::WGN) * Cfg::WM, wn0 = (wave % Cfg::WGN) * Cfg::WN;
    // this lane's 16 bytes (k = 4 qg .. 4 qg + 3) in the first fragment row of the wave's A tile; the (bh, bh) and (bl, 0) chunks of B
    const int offa = (wm0 + i) * ROWA + ((qg ^ swz4(i)) << 4);
    const int offd = BM * ROWA + (wn0 + i) * ROWB + ((qg ^ swz8(i)) << 4), offl = BM * ROWA + (wn0 + i) * ROWB + (((4 + qg) ^ swz8(i)) << 4);
    typename Cfg::MTr::acc_t acc[Cfg::TM][Cfg::TN];
    acc_zero<Cfg>(acc);
    const int nst = (K + 15) / 16;
    fetch(0);
    if (nst > 1) fetch(1);
    int slot = 0;
    for (int s = 0; s < nst; ++s) {
        // this wave's share of stage s has landed when only the fetches of stage s+1 are outstanding
        if (s + 1 < nst) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // everybody's has; nobody reads the slot of stage s-1 any more
        asm volatile("" ::: "memory");
        if (s + 2 < nst) fetch(slot == 0 ? 2 : slot - 1);
        const char* base = smem + slot * STAGE;
        h8 fa[4], fd[4], fl[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            fa[t] = *reinterpret_cast<const h8*>(base + offa + t * 16 * ROWA);
            fd[t] = *reinterpret_cast<const h8*>(base + offd + t * 16 * ROWB);
            fl[t] = *reinterpret_cast<const h8*>(base + offl + t * 16 * ROWB);
        }
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) {
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[tm], fd[tn], acc[tm][tn], 0, 0, 0);
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[tm], fl[tn], acc[tm][tn], 0, 0, 0);
            }
        slot = slot == 2 ? 0 : slot + 1;
    }
    __syncthreads();                                           // the epilogue reuses the LDS
    const float sc = scale[0];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc[tm][tn] *= sc;
    constexpr int SLOTS = BN >= 128 ? BN / 128 : 1;            // vpart / mupart slots: one per 128 columns of the full tiles, one per remainder tile
    const int vslot = slot0 + SLOTS * jt;
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));
    apply_epilogue<Cfg, EPI, EPI == 0, true>(acc, Phi, V, vpart, p, q, y, alpha, ut, K, Kp, Np, rb, cbase, vslot, smem_raw, mu, tid);
    if (SLOTS == 2 && EPI == 0 && tid < BM) {
        vpart[(int64_t)(vslot + 1) * Np + rb * BM + tid] = 0.0;
        if (mu) mu[(int64_t)(vslot + 1) * Np + rb * BM + tid] = 0.0;
    }
}

// part[b] = max |M[i][j]|, i, j < K, of block b's share
__global__ __launch_bounds__(256) void maxabs_kernel(const double* __restrict__ M, int K, int Kp, double* __restrict__ part) {
    __shared__ double r1[256];
    double m = 0;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < (int64_t)K * Kp; e += (int64_t)gridDim.x * 256) {
        const int j = (int)(e % Kp);
        if (j < K) m = fmax(m, fabs(M[e]));
    }
    r1[threadIdx.x] = m;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) r1[threadIdx.x] = fmax(r1[threadIdx.x], r1[threadIdx.x + w]);
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = r1[0];
}
// M (fp64, symmetric K x K in a Kp x Kp array) -> derived pairs, row j = column j of M; scale[0] = 2^-(e_Phi + e_M); scale[1] = 2^e_M
__global__ __launch_bounds__(256) void split_operand_kernel(const double* __restrict__ M, int K, int Kp, const double* __restrict__ part, int nparts,
                                                            const Scal* __restrict__ sc, char* __restrict__ out, float* __restrict__ scale) {
    double m = 0;
    for (int b = 0; b < nparts; ++b) m = fmax(m, part[b]);
    const int em = m > 0 ? 14 - ilogb(m) : 0, ephi = 14 - ilogbf((float)sc->s);
    if (blockIdx.x == 0 && threadIdx.x == 0) { scale[0] = ldexpf(1.0f, -(em + ephi)); scale[1] = ldexpf(1.0f, em); }
    const int64_t total = (int64_t)Kp * Kp;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int j = (int)(e / Kp), k = (int)(e % Kp);
        const double x = j < K && k < K ? ldexp(M[(int64_t)k * Kp + j], em) : 0.0;       // M^T = M; read along k for the write's sake
        const _Float16 h = (_Float16)x, l = (_Float16)(x - (double)h);
        const h2 dup = h2{h, h}, lo = h2{l, (_Float16)0};
        char* row = out + (int64_t)j * Kp * 8 + (int64_t)(k / 16) * 128 + (k % 16) * 4;
        *reinterpret_cast<unsigned*>(row) = *reinterpret_cast<const unsigned*>(&dup);
        *reinterpret_cast<unsigned*>(row + 64) = *reinterpret_cast<const unsigned*>(&lo);
    }
}

void F16x3Kernels::split_operand(const Geom& g, const double* M, char* B16, float* scale, double* part, const Scal* sc, hipStream_t st) {
    constexpr int NP = 512;
    hipLaunchKernelGGL(maxabs_kernel, dim3(NP), dim3(256), 0, st, M, g.K, g.Kp, part);
    hipLaunchKernelGGL(split_operand_kernel, dim3(2048), dim3(256), 0, st, M, g.K, g.Kp, (const double*)part, NP, sc, B16, scale);
}
template <int EPI, int BN>
int F16x3Kernels::apply(const Geom& g, int njt, int col0, int slot0, const float* Phi, const F16Operands& f, float* V,
                        double* vpart, const double* p, const double* q, const double* y, const double* alpha, const double* ut, double* mu,
                        hipStream_t st, int64_t rb0, int64_t nrb) {
    typedef F16Tile<BN> D;
    if (njt <= 0 || nrb <= 0) return 0;
    allow_big_lds(apply_f16_kernel<EPI, BN>, D::LDS_BYTES);
    hipLaunchKernelGGL((apply_f16_kernel<EPI, BN>), dim3((unsigned)(njt * nrb)), dim3(64 * D::WAVES), D::LDS_BYTES, st,
                       Phi, f.Phi16, f.B16, f.scale, V, vpart, p, q, y, alpha, ut, g.K, g.Kp, g.Np, njt, mu, col0, slot0, rb0);
    return (int)(njt * nrb);
}
#define SCFGP_F16_INST(EPI, BN)                                                                                                                       \
    template int F16x3Kernels::apply<EPI, BN>(const Geom&, int, int, int, const float*, const F16Operands&, float*, double*, const double*,          \
                                              const double*, const double*, const double*, const double*, double*, hipStream_t, int64_t, int64_t)
SCFGP_F16_INST(0, 256); SCFGP_F16_INST(1, 256); SCFGP_F16_INST(0, 128); SCFGP_F16_INST(1, 128); SCFGP_F16_INST(0, 64); SCFGP_F16_INST(1, 64);
#undef SCFGP_F16_INST
