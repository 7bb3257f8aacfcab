// MFMA tile engine for gfx950: C[m][n] += sum_k Aop[k][m] * Bop[k][n]
//
// One code path serves fp64 (v_mfma_f64_16x16x4_f64) and exact fp32
// (v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32): all take ONE scalar per lane for A and B
// with lane l supplying A[i=l%MS][k=l/MS] and B[k=l/MS][j=l%MS], so the LDS image is
// k-major -- sA[k][m], sB[k][n] -- and every fragment read is 16 consecutive
// elements per k row: bank-conflict free with a row stride == 16 (mod 32)
// elements (LD = B? + 16).  The two shapes differ only in the C/D row map.
//
// Operands reach LDS through register-staged loaders (double buffered, one
// barrier per k-tile): `NatLoader` for sources that are already k-major
// (S[k][x], x contiguous: coalesced 16-byte loads, 16-byte LDS stores) and
// `TrLoader` for x-major sources (S[x][k], k contiguous) which transposes on the
// LDS store.  At 32-64 cycles per MFMA the matrix pipe, not LDS or staging, is
// the bound of every kernel built on this engine.
#pragma once
#include "common.h"

// MFMA traits: element type T and instruction shape MS (16: 16x16x4, 32: 32x32x2, f32 only).
//   lane l supplies A[i = l % MS][k = l / MS] and B[k = l / MS][j = l % MS]
template <typename T, int MS> struct MT;
template <> struct MT<double, 16> {
    typedef v4d acc_t;
    static constexpr int M = 16, KS = 4, NACC = 4;
    static __device__ __forceinline__ void mfma(acc_t& c, double a, double b) {
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // C/D row held in accumulator register r of lane `lane` (f64 map, guide sec. 3)
    static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <> struct MT<float, 16> {
    typedef v4f acc_t;
    static constexpr int M = 16, KS = 4, NACC = 4;
    static __device__ __forceinline__ void mfma(acc_t& c, float a, float b) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int crow(int lane, int r) { return 4 * (lane >> 4) + r; }
};
typedef float v16f __attribute__((ext_vector_type(16)));
template <> struct MT<float, 32> {
    typedef v16f acc_t;
    static constexpr int M = 32, KS = 2, NACC = 16;
    static __device__ __forceinline__ void mfma(acc_t& c, float a, float b) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int crow(int lane, int r) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
};

template <typename T_, int BM_, int BN_, int BK_, int WGM_, int WGN_, int MS_ = 16>
struct TileCfg {
    typedef T_ T;
    typedef MT<T_, MS_> MTr;
    static constexpr int BM = BM_, BN = BN_, BK = BK_, WGM = WGM_, WGN = WGN_, MS = MS_;
    static constexpr int THREADS = 64 * WGM * WGN;
    static constexpr int WM = BM / WGM, WN = BN / WGN;
    static constexpr int TM = WM / MS, TN = WN / MS;
    static constexpr int LDA = BM + 16, LDB = BN + 16;
    static constexpr int SA = BK * LDA, SB = BK * LDB;        // elements per buffer
    static constexpr int LDS_BYTES = 2 * (SA + SB) * (int)sizeof(T);
    static_assert(BM % (MS * WGM) == 0 && BN % (MS * WGN) == 0 && BK % MTr::KS == 0, "tile shape");
};

// 16-byte global vector of source type
template <typename S> struct Vec16;
template <> struct Vec16<double> { typedef v2d type; static constexpr int N = 2; };
template <> struct Vec16<float> { typedef v4f type; static constexpr int N = 4; };

// ---------------------------------------------------------------------------
// NatLoader: source S[k][x] (row-major, leading dimension ld, x contiguous).
//   tile kt = rows [kt*BK, kt*BK+BK), columns [0, BX) relative to `base`.
//   Optional per-k-row weight w[k] (Gram with row weights) and x-limit guard.
// ---------------------------------------------------------------------------
template <typename S, typename T, int BX, int BK, int LD, int THREADS, bool WEIGHT, bool GUARD>
struct NatLoader {
    typedef typename Vec16<S>::type vec_t;
    static constexpr int VS = Vec16<S>::N;
    static constexpr int VPR = BX / VS;                       // vectors per k row
    static constexpr int NV = (BK * VPR + THREADS - 1) / THREADS;
    const S* base; int64_t ld; const double* w; int xlim;     // xlim: first invalid x (GUARD)
    vec_t r[NV]; T wr[NV];
    int tid;
    __device__ __forceinline__ NatLoader(const S* b, int64_t l, int t, const double* w_ = nullptr, int xl = 0)
        : base(b), ld(l), w(w_), xlim(xl), tid(t) {}
    __device__ __forceinline__ void load(int kt) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * THREADS;
            const int k = v / VPR, xv = v % VPR;
            const bool ok = (BK * VPR) % THREADS == 0 || v < BK * VPR;
            vec_t val;
#pragma unroll
            for (int e = 0; e < VS; ++e) val[e] = 0;
            if (ok && (!GUARD || xv * VS < xlim))
                val = *reinterpret_cast<const vec_t*>(base + (int64_t)(kt * BK + k) * ld + xv * VS);
            r[i] = val;
            if (WEIGHT) wr[i] = ok ? (T)w[kt * BK + k] : (T)0;
        }
    }
    __device__ __forceinline__ void store(T* s) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * THREADS;
            if ((BK * VPR) % THREADS != 0 && v >= BK * VPR) continue;
            const int k = v / VPR, xv = v % VPR;
            T* d = s + k * LD + xv * VS;
            constexpr int TV = 16 / (int)sizeof(T) < VS ? 16 / (int)sizeof(T) : VS;   // elements per LDS store
            typedef T tv_t __attribute__((ext_vector_type(TV)));
#pragma unroll
            for (int e0 = 0; e0 < VS; e0 += TV) {
                tv_t o;
#pragma unroll
                for (int e = 0; e < TV; ++e) o[e] = WEIGHT ? (T)r[i][e0 + e] * wr[i] : (T)r[i][e0 + e];
                *reinterpret_cast<tv_t*>(d + e0) = o;
            }
        }
    }
};

// ---------------------------------------------------------------------------
// TrLoader: source S[x][k] (row-major, k contiguous); transposes into s[k][x].
// ---------------------------------------------------------------------------
template <typename S, typename T, int BX, int BK, int LD, int THREADS>
struct TrLoader {
    typedef typename Vec16<S>::type vec_t;
    static constexpr int VS = Vec16<S>::N;
    static constexpr int VPR = BK / VS;                       // vectors per x row
    static constexpr int NV = (BX * VPR + THREADS - 1) / THREADS;
    const S* base; int64_t ld; int tid;
    vec_t r[NV];
    __device__ __forceinline__ TrLoader(const S* b, int64_t l, int t) : base(b), ld(l), tid(t) {}
    __device__ __forceinline__ void load(int kt) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * THREADS;
            const int x = v / VPR, kv = v % VPR;
            const bool ok = (BX * VPR) % THREADS == 0 || v < BX * VPR;
            vec_t val;
#pragma unroll
            for (int e = 0; e < VS; ++e) val[e] = 0;
            if (ok) val = *reinterpret_cast<const vec_t*>(base + (int64_t)x * ld + kt * BK + kv * VS);
            r[i] = val;
        }
    }
    __device__ __forceinline__ void store(T* s) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * THREADS;
            if ((BX * VPR) % THREADS != 0 && v >= BX * VPR) continue;
            const int x = v / VPR, kv = v % VPR;
#pragma unroll
            for (int e = 0; e < VS; ++e) s[(kv * VS + e) * LD + x] = (T)r[i][e];
        }
    }
};

// ---------------------------------------------------------------------------
// main loop: accumulates nkt k-tiles into acc[TM][TN]; smem = 2*(SA+SB) elements
// ---------------------------------------------------------------------------
template <class Cfg, class LA, class LB>
__device__ __forceinline__ void tile_mainloop(LA& la, LB& lb, int nkt,
                                              typename Cfg::MTr::acc_t (&acc)[Cfg::TM][Cfg::TN],
                                              typename Cfg::T* smem) {
    typedef typename Cfg::T T;
    typedef typename Cfg::MTr M;
    constexpr int MS = Cfg::MS, KS = M::KS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm0 = (wave / Cfg::WGN) * Cfg::WM, wn0 = (wave % Cfg::WGN) * Cfg::WN;
    T* sA = smem;
    T* sB = smem + 2 * Cfg::SA;
    la.load(0); lb.load(0);
    la.store(sA); lb.store(sB);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nkt;
        if (more) { la.load(kt + 1); lb.load(kt + 1); }
        const T* a_s = sA + cur * Cfg::SA + (lane / MS) * Cfg::LDA + wm0 + (lane % MS);
        const T* b_s = sB + cur * Cfg::SB + (lane / MS) * Cfg::LDB + wn0 + (lane % MS);
#pragma unroll
        for (int kk = 0; kk < Cfg::BK / KS; ++kk) {
            T a[Cfg::TM], b[Cfg::TN];
#pragma unroll
            for (int tm = 0; tm < Cfg::TM; ++tm) a[tm] = a_s[kk * KS * Cfg::LDA + tm * MS];
#pragma unroll
            for (int tn = 0; tn < Cfg::TN; ++tn) b[tn] = b_s[kk * KS * Cfg::LDB + tn * MS];
#pragma unroll
            for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < Cfg::TN; ++tn) M::mfma(acc[tm][tn], a[tm], b[tn]);
        }
        if (more) { la.store(sA + (cur ^ 1) * Cfg::SA); lb.store(sB + (cur ^ 1) * Cfg::SB); }
        __syncthreads();
    }
}

template <class Cfg>
__device__ __forceinline__ void acc_zero(typename Cfg::MTr::acc_t (&acc)[Cfg::TM][Cfg::TN]) {
#pragma unroll
    for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < Cfg::TN; ++tn)
#pragma unroll
            for (int r = 0; r < Cfg::MTr::NACC; ++r) acc[tm][tn][r] = 0;
}

// coordinates of accumulator element (tm,tn,r) inside the workgroup tile
template <class Cfg>
struct AccCoord {
    int lane, wm0, wn0;
    __device__ __forceinline__ AccCoord() {
        lane = threadIdx.x & 63;
        const int wave = threadIdx.x >> 6;
        wm0 = (wave / Cfg::WGN) * Cfg::WM; wn0 = (wave % Cfg::WGN) * Cfg::WN;
    }
    __device__ __forceinline__ int row(int tm, int r) const { return wm0 + tm * Cfg::MS + Cfg::MTr::crow(lane, r); }
    __device__ __forceinline__ int col(int tn) const { return wn0 + tn * Cfg::MS + (lane % Cfg::MS); }
};
