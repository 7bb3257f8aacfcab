// Split-precision MFMA tile engine for gfx950 (experimental compute mode SCFGP_BF16X3):
//   C[m][n] += sum_k A[m][k] B[k][n]  on fp32 operands through v_mfma_f32_32x32x16_bf16.
//
// Every fp32 value is split EXACTLY into three bf16 pieces, x = h + m + l (h = bf16(x), m = bf16(x - h),
// l = bf16(x - h - m): 3 x 8 significand bits cover fp32's 24), while it is staged into LDS; a product a.b is then the
// six MFMA terms  h.h + (h.m + m.h) + (h.l + m.m + l.h)  accumulated in the fp32 accumulators -- bf16 products are
// exact in fp32, and the three dropped terms (m.l, l.m, l.l) are <= 2^-23 |a||b|, the size of fp32's own product
// rounding.  bf16 MFMA runs at 16x the fp32-input MFMA rate, so six of them are 2.67x faster than one exact-fp32 MFMA;
// operands stay fp32 in HBM (no extra bytes), the split costs ~6 VALU instructions per staged element.
//
// LDS image of an operand tile (32 k deep = two MFMA steps per barrier): three planes [x][32 k] of bf16, 64 bytes per
// row x, the four 16-byte chunks of a row XOR-swizzled with bits 2..3 of x so that the 16-lane groups of a
// ds_read_b128 fragment read hit 64 distinct banks.  A lane's MFMA fragment of step s (row x = lane & 31,
// k = 16 s + 8 (lane >> 5) .. +7) is ONE ds_read_b128 per plane.
#pragma once
#include "tile_engine.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <int BM_, int BN_, int WGM_, int WGN_>
struct Bf3Cfg {
    typedef float T;
    typedef MT<float, 32> MTr;                                  // accumulator tile and C/D map of the 32x32 shapes
    static constexpr int BM = BM_, BN = BN_, BK = 32, WGM = WGM_, WGN = WGN_, MS = 32;
    static constexpr bool SWZA = false, PERM = false;
    static constexpr int THREADS = 64 * WGM * WGN;
    static constexpr int WM = BM / WGM, WN = BN / WGN;
    static constexpr int TM = WM / MS, TN = WN / MS;
    static constexpr int PITCH = 64;                            // bytes per row of a plane
    static constexpr int PA = BM * PITCH, PB = BN * PITCH;      // bytes per plane
    static constexpr int BUF = 3 * (PA + PB);                   // one k-tile: A planes h, m, l then B planes h, m, l
    static constexpr int LDS_BYTES = 2 * BUF;
    static_assert(BM % (MS * WGM) == 0 && BN % (MS * WGN) == 0, "tile shape");
};

// byte offset of the 8-byte piece holding k = 4 kq .. 4 kq + 3 (kq < 8) of row x inside a plane
__device__ __forceinline__ int bf3_piece(int x, int kq) { return x * 64 + (((kq >> 1) ^ ((x >> 2) & 3)) << 4) + ((kq & 1) << 3); }
// byte offset of the 16-byte chunk c (k = 8 c .. 8 c + 7) of row x
__device__ __forceinline__ int bf3_chunk(int x, int c) { return x * 64 + ((c ^ ((x >> 2) & 3)) << 4); }

// exact three-way split of 4 fp32 values into bf16 pieces
__device__ __forceinline__ void bf3_split4(const float (&x)[4], bf16x4& h, bf16x4& m, bf16x4& l) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 hh = (__bf16)x[e];
        const float r1 = x[e] - (float)hh;
        const __bf16 mm = (__bf16)r1;
        const float r2 = r1 - (float)mm;
        h[e] = hh; m[e] = mm; l[e] = (__bf16)r2;
    }
}

// ---------------------------------------------------------------------------
// Bf3TrLoader: source S[x][k] (fp32, row-major, k contiguous): each 16-byte vector is 4 consecutive k of one row
// and becomes one 8-byte piece per plane.  DOT as TrLoader: the row dots dot[x] = sum_k S[x][k] d[k] in fp64 for the
// k-tiles kt % nparts == part.
// ---------------------------------------------------------------------------
template <int BX, int THREADS, bool DOT>
struct Bf3TrLoader {
    static constexpr int VPR = 8;                             // vectors per row of a 32-deep k-tile
    static constexpr int NV = (BX * VPR + THREADS - 1) / THREADS;
    const float* ptr[NV]; int tid;
    v4f r[2][NV];                                             // two k-tiles in flight (prefetch distance 2)
    const double* dptr = nullptr; double dv[2][4]; double dacc[NV]; bool dot_on = false, dot_now[2] = {false, false};
    int dpart = 0, dnparts = 1, dphase = 0;
    int dlo = 0, dhi = -1, dkt = 0;                             // range mode (dot_range): k-tiles [dlo, dhi) instead of the modulo rule
    __device__ __forceinline__ void dot_range(int lo, int hi) { dlo = lo; dhi = hi; }
    __device__ __forceinline__ Bf3TrLoader(const float* b, int64_t l, int t, const double* d_ = nullptr, int part = 0, int nparts = 1)
        : tid(t), dpart(part), dnparts(nparts) {
        dot_on = DOT && d_ != nullptr;
        if (dot_on) dptr = d_ + (t % VPR) * 4;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * THREADS;
            ptr[i] = b + (int64_t)(v / VPR) * l + (v % VPR) * 4;
            dacc[i] = 0;
        }
    }
    template <int SET>
    __device__ __forceinline__ void load() {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * THREADS;
            const bool ok = (BX * VPR) % THREADS == 0 || v < BX * VPR;
            v4f val = {0, 0, 0, 0};
            if (ok) val = *reinterpret_cast<const v4f*>(ptr[i]);
            r[SET][i] = val;
            ptr[i] += 32;
        }
        if (DOT && dot_on) {
            dot_now[SET] = dhi >= 0 ? (dkt >= dlo && dkt < dhi) : dphase == dpart;
            ++dkt;
            if (dot_now[SET]) {
#pragma unroll
                for (int e = 0; e < 4; ++e) dv[SET][e] = dptr[e];
            }
            dptr += 32;
            dphase = dphase + 1 == dnparts ? 0 : dphase + 1;
        }
    }
    template <int SET>
    __device__ __forceinline__ void store(char* planes, int plane_bytes) {      // planes: base of this operand's plane h
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * THREADS;
            if ((BX * VPR) % THREADS != 0 && v >= BX * VPR) continue;
            const int x = v / VPR, kq = v % VPR;
            const float xs[4] = {r[SET][i][0], r[SET][i][1], r[SET][i][2], r[SET][i][3]};
            bf16x4 h, m, l;
#ifdef SCFGP_DIAG_BF3_NOSPLIT                                   // timing diagnostic only (wrong numbers): no split arithmetic
            { const float2 lo = {xs[0], xs[1]}, hi = {xs[2], xs[3]};
              h = __builtin_bit_cast(bf16x4, lo); m = __builtin_bit_cast(bf16x4, hi); l = h; }
#else
            bf3_split4(xs, h, m, l);
#endif
            char* d = planes + bf3_piece(x, kq);
            *reinterpret_cast<bf16x4*>(d) = h;
            *reinterpret_cast<bf16x4*>(d + plane_bytes) = m;
            *reinterpret_cast<bf16x4*>(d + 2 * plane_bytes) = l;
            if (DOT && dot_on && dot_now[SET]) {                // fp64 work in this loop is expensive: only in the tile's own k-tiles
#pragma unroll
                for (int e = 0; e < 4; ++e) dacc[i] = fma((double)r[SET][i][e], dv[SET][e], dacc[i]);
            }
        }
    }
    __device__ __forceinline__ void dot_reduce(double* __restrict__ out) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            double s = dacc[i];
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
            const int v = tid + i * THREADS;
            if (((BX * VPR) % THREADS == 0 || v < BX * VPR) && v % VPR == 0) out[v / VPR] = s;
        }
    }
};

// ---------------------------------------------------------------------------
// Bf3CopyLoader: an operand that was split ONCE in global memory (bf3_presplit: the K x K matrices B and Abar, which
// every workgroup of the apply product re-reads) arrives as ready-made plane rows and is copied, 16 bytes per lane,
// straight into the image: no conversion work in the loop.
//   global layout: [k-tile][plane][column j < ld][32 k] bf16 = 64 bytes per (k-tile, plane, j), chunks swizzled with
//   bits 2..3 of j (tile origins are multiples of 16 columns, so it is the image's swizzle).
// ---------------------------------------------------------------------------
template <int BX, int THREADS>
struct Bf3CopyLoader {
    static constexpr int VPP = BX * 4;                        // 16-byte vectors per plane
    static constexpr int NV = (3 * VPP + THREADS - 1) / THREADS;
    const char* ptr[NV]; int64_t step; int tid;
    v4f r[2][NV];
    __device__ __forceinline__ Bf3CopyLoader(const void* split, int64_t ldj, int col0, int t) : step(3 * ldj * 64), tid(t) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * THREADS, pl = v / VPP, off = v % VPP;
            ptr[i] = reinterpret_cast<const char*>(split) + ((int64_t)pl * ldj + col0) * 64 + off * 16;
        }
    }
    template <int SET>
    __device__ __forceinline__ void load() {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * THREADS;
            if ((3 * VPP) % THREADS == 0 || v < 3 * VPP) r[SET][i] = *reinterpret_cast<const v4f*>(ptr[i]);
            ptr[i] += step;
        }
    }
    template <int SET>
    __device__ __forceinline__ void store(char* planes, int plane_bytes) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * THREADS, pl = v / VPP, off = v % VPP;
            if ((3 * VPP) % THREADS != 0 && v >= 3 * VPP) continue;
            *reinterpret_cast<v4f*>(planes + pl * plane_bytes + off * 16) = r[SET][i];
        }
    }
};
// out (layout above, Kp/32 k-tiles x 3 planes x Kp columns) from the fp32 matrix M (Kp x Kp, element (k, j) = M[k*Kp + j])
__global__ void bf3_presplit_kernel(const float* __restrict__ M, __bf16* __restrict__ out, int Kp) {
    const int64_t n = (int64_t)Kp * Kp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i / Kp), j = (int)(i % Kp);       // consecutive threads: consecutive j (coalesced read)
        const float x = M[i];
        const __bf16 h = (__bf16)x; const float r1 = x - (float)h;
        const __bf16 m = (__bf16)r1; const float r2 = r1 - (float)m;
        const int kt = k >> 5, kk = k & 31;
        const int64_t row = ((int64_t)kt * 3 * Kp + j) * 32 + ((((kk >> 3) ^ ((j >> 2) & 3)) << 3) | (kk & 7));
        out[row] = h; out[row + (int64_t)Kp * 32] = m; out[row + (int64_t)2 * Kp * 32] = (__bf16)r2;
    }
}

// MFMAs of one k-tile (32 deep = two 32x32x16 steps) out of LDS buffer `buf`
template <class Cfg>
__device__ __forceinline__ void bf3_compute(const char* buf, typename Cfg::MTr::acc_t (&acc)[Cfg::TM][Cfg::TN]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm0 = (wave / Cfg::WGN) * Cfg::WM, wn0 = (wave % Cfg::WGN) * Cfg::WN;
    const int r = lane & 31, hk = lane >> 5;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        bf16x8 b[Cfg::TN][3], a[Cfg::TM][3];
#pragma unroll
        for (int tn = 0; tn < Cfg::TN; ++tn) {
            const char* p = buf + 3 * Cfg::PA + bf3_chunk(wn0 + tn * 32 + r, 2 * st + hk);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) b[tn][pl] = *reinterpret_cast<const bf16x8*>(p + pl * Cfg::PB);
        }
#pragma unroll
        for (int tm = 0; tm < Cfg::TM; ++tm) {
            const char* p = buf + bf3_chunk(wm0 + tm * 32 + r, 2 * st + hk);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) a[tm][pl] = *reinterpret_cast<const bf16x8*>(p + pl * Cfg::PA);
        }
        // six terms per accumulator tile, smallest first; consecutive MFMAs go to different accumulators
        constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};      // (plane of A, plane of B): l.h h.l m.m m.h h.m h.h
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < Cfg::TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][TA[t]], b[tn][TB[t]], acc[tm][tn], 0, 0, 0);
    }
}

// main loop: nkt k-tiles into acc; smem = Cfg::LDS_BYTES.  LDS double buffered, one barrier per k-tile; the global
// fetches run TWO k-tiles ahead in two register sets: iteration kt issues the fetch of tile kt+2, multiplies tile kt
// out of LDS and splits / stores tile kt+1 (fetched a whole iteration ago).  Measured and not kept
// (profiles/r02_tuning.md): the split's VALU work forced between the MFMAs with sched_group_barrier (+35 % time), a
// branch-free form of the fp64 row dots (+30 %), a peeled steady-state iteration without the two conditions (+7 %).
template <class Cfg, int SET, class LA, class LB>
__device__ __forceinline__ void bf3_step(LA& la, LB& lb, int kt, int nkt, typename Cfg::MTr::acc_t (&acc)[Cfg::TM][Cfg::TN], char* smem) {
    // SET = kt & 1: registers of tile kt+1 are set SET^1, tile kt+2 goes into set SET (tile kt's set, stored last iteration)
    const int cur = kt & 1;
    if (kt + 2 < nkt) { la.template load<SET>(); lb.template load<SET>(); }
    bf3_compute<Cfg>(smem + cur * Cfg::BUF, acc);
    if (kt + 1 < nkt) {
        la.template store<SET ^ 1>(smem + (cur ^ 1) * Cfg::BUF, Cfg::PA);
        lb.template store<SET ^ 1>(smem + (cur ^ 1) * Cfg::BUF + 3 * Cfg::PA, Cfg::PB);
    }
    __syncthreads();
}
template <class Cfg, class LA, class LB>
__device__ __forceinline__ void bf3_mainloop(LA& la, LB& lb, int nkt, typename Cfg::MTr::acc_t (&acc)[Cfg::TM][Cfg::TN], char* smem) {
    la.template load<0>(); lb.template load<0>();
    if (nkt > 1) { la.template load<1>(); lb.template load<1>(); }
    la.template store<0>(smem, Cfg::PA); lb.template store<0>(smem + 3 * Cfg::PA, Cfg::PB);
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nkt; kt += 2) {
        bf3_step<Cfg, 0>(la, lb, kt, nkt, acc, smem);
        bf3_step<Cfg, 1>(la, lb, kt + 1, nkt, acc, smem);
    }
    if (kt < nkt) bf3_step<Cfg, 0>(la, lb, kt, nkt, acc, smem);
}
