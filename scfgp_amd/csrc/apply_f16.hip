// Compute mode SCFGP_F16X3 (include/scfgp_hip.h; a labelled SECONDARY mode, never the headline): the two square apply products
// V = Phi B and Phibar = 2 Phi Abar + ... as a THREE-TERM fp16 split on the fp16 matrix pipe (gram_f16.hip: the two Gram products);
// everything else of the evaluation is fp32 mode's.  No reference counterpart (the reference is float64 throughout,
// SCFGP/SCFGP.py:95-96); what it must equal is fp32 mode, whose parity tier it runs under (tests/test_gpu_round5.py).
//
// The split.  x = (h + l) 2^-e with h = fp16(x 2^e), l = fp16(x 2^e - h) and ONE power-of-two scale per operand matrix that puts its
// largest entry in [2^14, 2^15): h + l is x to 2^-23 of x for every entry within about 2^-15 of the largest, to l's subnormal step
// (2^-40 of the largest) below (tests/test_f16x3_split.py).  Products of two fp16 values are exact in fp32, so
// a b ~ ah bh + al bh + ah bl  loses only the l.l term (at most 2^-22 of the operands' bounds) against an exact-fp32 product; what the mode really pays is the fp16 matrix instruction's accumulation, which truncates
// (gram_f16.hip, profiles/r05_tuning.md).  Measured errors of the two products here: within 1.0-3.5x of fp32 mode's.
//
// The operands.  Both come in "plane form" (kernels.h): 4 bytes per element, per 16 consecutive k the 16 h's and then the 16 l's.  Phi's
// is the array the fp16 Gram reads too (one split pass writes it, gram_f16.hip: split_rows_kernel); the small operand (B or Abar, K x K,
// cache resident) is written row by row = output column by output column by split_operand.  A lane's 8 k of v_mfma_f32_16x16x32_f16 are
// 16 bytes of ONE plane, so the three terms are three instructions into one accumulator, Ah.Bh + Al.Bh + Ah.Bl, per 16 x 16 output tile and
// 32 k: 48 cycles where exact fp32 spends 256.  (The first version of this file kept Phi as packed (h, l) pairs against the derived
// operand pairs (bh, bh), (bl, 0): two instructions per 16 k, i.e. four per 32 -- a third more matrix work and twice the operand bytes.)
// Stage = 32 k: 128-byte rows of both images, ring of two stages (64 KiB each at BN = 256), fragment reads by hand (inline assembly with
// counted lgkmcnt, three of the four fragment sets live at a time), the fetch of stage s+1 in flight while stage s is multiplied.
// Both f16x3 kernels run the matrix pipe at 0.7-0.76 busy with the clock at ~1.65 GHz: the fp16 pipe at this rate is power-bound
// (profiles/r05_tuning.md).
#include "apply_epilogue.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

namespace {
constexpr int BM = 256, ROW = 128;                                          // bytes of one row's 32 k in plane form: [16 h | 16 l] [16 h | 16 l]
// tile widths as in fp32 mode's launch plan (apply.hip): 256 (16 waves), 128 (8 waves: an odd 128-column block, the thin last round's
// row blocks), 64 (4 waves: the ragged remainder, K = 2112 = 8 x 256 + 64); wave tiles 64 x 64 throughout
template <int BN_> struct F16Tile {
    static constexpr int BN = BN_, WAVES = 4 * (BN / 64), STAGE = (BM + BN) * ROW, STAGES = 2, LDS_BYTES = STAGES * STAGE,
                         DPW = STAGE / 1024 / WAVES, WAVES_PER_EU = WAVES == 16 ? 4 : 2;   // one 16- or 8-wave workgroup per CU, two of 4 waves
    static_assert(STAGE / 1024 % WAVES == 0 && LDS_BYTES <= 160 * 1024, "whole DMA instructions per wave; the ring fits the LDS");
    typedef TileCfg<float, BM, BN, 16, 4, BN / 64, 16, true> Cfg;           // the fp32 tile's accumulator map (apply_epilogue.h)
};
// position swizzle of the images (apply.hip): 16-byte chunk c of a 128-byte row x lies at position c ^ f[(x >> 1) & 7]
__device__ __forceinline__ int swz8(int x) { return (int)((0x6BEB08u >> (3 * ((x >> 1) & 7))) & 7); }
// fragment read by hand: through plain loads the compiler may put an s_waitcnt vmcnt(0) in front (it cannot tell the LDS-DMA of the NEXT
// stage from the data being read: gram_f16.hip), and the release of the fragments is counted here (SCFGP_F16_WAIT)
template <int OFF> __device__ __forceinline__ void lds_read(h8& out, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(out) : "v"(addr), "n"(OFF));
}
#define SCFGP_F16_WAIT(n, x) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) :: "memory")
}

// One 256 x BN tile: column tile jt of the launch, row block rb0 + wid / njt.  Phi16: Np x Kp in plane form; B16: Kp rows (= output
// columns) of Kp elements in plane form; scale[0] = 2^-(e_Phi + e_B).
template <int EPI, int BN>
__global__ __launch_bounds__((64 * F16Tile<BN>::WAVES))
__attribute__((amdgpu_waves_per_eu(F16Tile<BN>::WAVES_PER_EU, F16Tile<BN>::WAVES_PER_EU)))
void apply_f16_kernel(const float* __restrict__ Phi, const unsigned* __restrict__ Phi16, const char* __restrict__ B16, const float* __restrict__ scale,
                      unsigned* __restrict__ V16, const float* __restrict__ vbound, float* V, double* __restrict__ vpart, const double* __restrict__ p, const double* __restrict__ q,
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
    // DMA instruction t = DPW wave + u of a stage: 1 KiB = rows 8 t .. 8 t + 7 of the stacked image (A's 256 rows, then B's BN), 8 chunks
    // each; chunk c of row x goes to position c ^ swz8(x) (source-side swizzle, linear LDS writes).  B rows are staged in the order
    // that leaves an MFMA lane 4 ADJACENT output columns (apply.hip): LDS row tn 16 + i of a wave tile holds operand row 4 i + tn.
    const char* src[DPW]; int dst[DPW];
#pragma unroll
    for (int u = 0; u < DPW; ++u) {
        const int t = DPW * wave + u, x = 8 * t + lane / 8;
        if (x < BM) {
            src[u] = reinterpret_cast<const char*>(Phi16 + (rb * BM + x) * Kp) + (((lane % 8) ^ swz8(x)) << 4);
        } else {
            const int xb = x - BM, xcol = (xb & ~63) + 4 * (xb & 15) + ((xb >> 4) & 3);
            src[u] = B16 + (int64_t)(cbase + xcol) * Kp * 4 + (((lane % 8) ^ swz8(xb)) << 4);
        }
        dst[u] = t * 1024;
    }
    const auto fetch = [&](int slot) {
#pragma unroll
        for (int u = 0; u < DPW; ++u) {
            __builtin_amdgcn_global_load_lds((gbl_void*)src[u], (lds_void*)(smem + slot * STAGE + dst[u]), 16, 0, 0);
            src[u] += ROW;
        }
    };
    const int i = lane & 15, qg = lane >> 4;
    const int wm0 = (wave / Cfg::WGN) * Cfg::WM, wn0 = (wave % Cfg::WGN) * Cfg::WN;
    // this lane's 8 k (8 qg .. 8 qg + 7) of a row: chunk 4 (qg >> 1) + (qg & 1) of the h's, two chunks further the l's
    const int ch = 4 * (qg >> 1) + (qg & 1), sw = swz8(i);
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const unsigned oah = (wm0 + i) * ROW + ((ch ^ sw) << 4), oal = (wm0 + i) * ROW + (((ch + 2) ^ sw) << 4);
    const unsigned obh = (BM + wn0 + i) * ROW + ((ch ^ sw) << 4), obl = (BM + wn0 + i) * ROW + (((ch + 2) ^ sw) << 4);
    typename Cfg::MTr::acc_t acc[Cfg::TM][Cfg::TN];
    acc_zero<Cfg>(acc);
    const int nst = (K + 31) / 32;
    fetch(0);
    int slot = 0;
    for (int s = 0; s < nst; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's share of stage s (issued a stage ago) has landed
        __builtin_amdgcn_s_barrier();                          // everybody's has; nobody reads the other slot any more
        asm volatile("" ::: "memory");
        if (s + 1 < nst) fetch(slot ^ 1);
        const unsigned base = lds0 + slot * STAGE;
        // at most three of the four fragment sets are live (48 registers beside the 64 accumulators): Ah, Bh and Al are read up front, Bl
        // moves into Al's registers while the middle term is multiplied
        h8 ah[4], al[4], bh[4], bl[4];
        static_for<4>([&](auto tc) { lds_read<decltype(tc)::value * 16 * ROW>(ah[decltype(tc)::value], base + oah); });
        static_for<4>([&](auto tc) { lds_read<decltype(tc)::value * 16 * ROW>(bh[decltype(tc)::value], base + obh); });
        static_for<4>([&](auto tc) { lds_read<decltype(tc)::value * 16 * ROW>(al[decltype(tc)::value], base + oal); });
        SCFGP_F16_WAIT(4, ah);
        SCFGP_F16_WAIT(4, bh);
        __builtin_amdgcn_sched_barrier(0);
        static_for<16>([&](auto ic) { constexpr int tm = decltype(ic)::value / 4, tn = decltype(ic)::value % 4;
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0); });
        __builtin_amdgcn_sched_barrier(0);
        SCFGP_F16_WAIT(0, al);
        __builtin_amdgcn_sched_barrier(0);
        static_for<4>([&](auto mc) { constexpr int tm = decltype(mc)::value;
            static_for<4>([&](auto nc) { constexpr int tn = decltype(nc)::value;
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0); });
            __builtin_amdgcn_sched_barrier(0);
            lds_read<tm * 16 * ROW>(bl[tm], base + obl);         // Al[tm] is dead: its registers are free
            __builtin_amdgcn_sched_barrier(0);
        });
        SCFGP_F16_WAIT(0, bl);
        __builtin_amdgcn_sched_barrier(0);
        static_for<16>([&](auto ic) { constexpr int tm = decltype(ic)::value / 4, tn = decltype(ic)::value % 4;
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0); });
        __builtin_amdgcn_sched_barrier(0);
        slot ^= 1;
    }
    __syncthreads();                                           // the epilogue reuses the LDS
    const float sc = scale[0];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc[tm][tn] *= sc;
    if constexpr (EPI == 0) {
        // V in plane form for the weighted Gram (gram_f16.hip), straight from the accumulators: what split_v's pass over V would write,
        // bit for bit (same values, same scale 2^e, e = 14 - ilogb(bound of |V|)).  A lane holds 4 adjacent columns of 16 rows, the four
        // lanes of a quad one 64-byte block [16 h | 16 l] of a row; they swap halves (two quad permutes) so that each lane stores 16
        // contiguous bytes of it -- as 8-byte stores of its own h's and l's the launch was 1.9 ms longer (four times the cache-line
        // requests of the fp32 V it writes anyway).
        if (V16) {
            const float bnd = vbound[0], up = bnd > 0.f ? ldexpf(1.0f, 14 - ilogbf(bnd)) : 1.0f;
            AccCoord<Cfg> co((int)threadIdx.x);
            const int jg = cbase + co.wn0 + 4 * (co.lane & 15);
            const bool low = (co.lane & 2) == 0;                   // lanes 0, 1 of a quad store the h's, lanes 2, 3 the l's
            const int64_t poff = 64 * (jg >> 4) + 16 * (co.lane & 3);
#pragma unroll
            for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    _Float16 h[4], l[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float x = (jg + k < K ? acc[tm][k][r] : 0.f) * up;
                        h[k] = (_Float16)x; l[k] = (_Float16)(x - (float)h[k]);
                    }
                    const h2 h01 = h2{h[0], h[1]}, h23 = h2{h[2], h[3]}, l01 = h2{l[0], l[1]}, l23 = h2{l[2], l[3]};
                    int w[4] = {__builtin_bit_cast(int, h01), __builtin_bit_cast(int, h23), __builtin_bit_cast(int, l01), __builtin_bit_cast(int, l23)};
                    int o[4];
                    // first 8 bytes: the h's (lanes 0, 1) or l's (lanes 2, 3) of quad lane 0 / 2 / 0 / 2; second 8 bytes: of quad lane 1 / 3 / 1 / 3
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        const int ha = __builtin_amdgcn_update_dpp(0, w[d], 0x88, 0xF, 0xF, true), la = __builtin_amdgcn_update_dpp(0, w[2 + d], 0x88, 0xF, 0xF, true);
                        const int hb = __builtin_amdgcn_update_dpp(0, w[d], 0xDD, 0xF, 0xF, true), lb = __builtin_amdgcn_update_dpp(0, w[2 + d], 0xDD, 0xF, 0xF, true);
                        o[d] = low ? ha : la; o[2 + d] = low ? hb : lb;
                    }
                    char* row = reinterpret_cast<char*>(V16 + (rb * BM + co.row(tm, r)) * Kp) + poff;
                    *reinterpret_cast<int4*>(row) = int4{o[0], o[1], o[2], o[3]};
                }
        }
    }
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
// M (fp64, symmetric K x K in a Kp x Kp array) -> plane form, row j = column j of M; scale[0] = 2^-(e_Phi + e_M); scale[1] = 2^e_M
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
        _Float16* row = reinterpret_cast<_Float16*>(out + (int64_t)j * Kp * 4 + (int64_t)(k / 16) * 64) + k % 16;
        row[0] = h; row[16] = l;
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
                       Phi, f.Phi16, f.B16, f.scale, f.V16, f.vbound, V, vpart, p, q, y, alpha, ut, g.K, g.Kp, g.Np, njt, mu, col0, slot0, rb0);
    return (int)(njt * nrb);
}
#define SCFGP_F16_INST(EPI, BN)                                                                                                                       \
    template int F16x3Kernels::apply<EPI, BN>(const Geom&, int, int, int, const float*, const F16Operands&, float*, double*, const double*,          \
                                              const double*, const double*, const double*, const double*, double*, hipStream_t, int64_t, int64_t)
SCFGP_F16_INST(0, 256); SCFGP_F16_INST(1, 256); SCFGP_F16_INST(0, 128); SCFGP_F16_INST(1, 128); SCFGP_F16_INST(0, 64); SCFGP_F16_INST(1, 64);
#undef SCFGP_F16_INST
