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
#include "sincos.h"
#include <utility>

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>), in this order (hand-scheduled instruction sequences whose
// inline-assembly operands must be compile-time constants)
template <int... I, class F> __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>()), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>(), f); }

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

// SWZA: the A image is written by a TrLoader<..., SWZ = true> (column index XOR-swizzled, see there)
template <typename T_, int BM_, int BN_, int BK_, int WGM_, int WGN_, int MS_ = 16, bool SWZA_ = false>
struct TileCfg {
    typedef T_ T;
    typedef MT<T_, MS_> MTr;
    static constexpr int BM = BM_, BN = BN_, BK = BK_, WGM = WGM_, WGN = WGN_, MS = MS_;
    static constexpr bool SWZA = SWZA_;
    static_assert(!SWZA_ || (BK_ == 16 && MS_ == 16), "swizzle is laid out for BK = 16 and the 16x16x4 shapes");
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
//   Optional per-k-row weight w[k] (Gram with row weights), x-limit guard, and a "side" product:
//   with SIDE and side_on the loader also accumulates, from the unweighted values it stages anyway,
//   side[x] = sum_k s[k] * S[k][x] (Phi^T y, Phi^T p on the Gram's diagonal tiles): in the compute type
//   per thread between side_flush() calls (one row in BK: a chain 1/BK as long as the MFMA's), fp64 across.
// ---------------------------------------------------------------------------
//   NSETS: register sets for fetches more than one k-tile ahead (tile_mainloop_deep3); load<SET> / store<SET> name the set
template <typename S, typename T, int BX, int BK, int LD, int THREADS, bool WEIGHT, bool GUARD, bool SIDE = false, int NSETS = 1>
struct NatLoader {
    typedef typename Vec16<S>::type vec_t;
    static constexpr int VS = Vec16<S>::N;
    static constexpr int VPR = BX / VS;                       // vectors per k row
    static constexpr int NV = (BK * VPR + THREADS - 1) / THREADS;
    // Loads are issued for consecutive k-tiles (0, 1, 2, ...): each vector keeps a running pointer
    // that advances by BK rows per call, so the loop carries no 64-bit multiplies.
    const S* ptr[NV]; const double* wptr[NV]; int64_t step; int xlim;     // xlim: first invalid x (GUARD)
    vec_t r[NSETS][NV]; double wr[NSETS][NV];                 // weights stay raw until store(): converting in
    int tid;                                                  // load() would wait on the fetch before the MFMAs
    // side accumulators: when THREADS is a multiple of VPR every vector of a thread has the same columns -> one set
    static constexpr bool SAMEX = THREADS % VPR == 0;
    static constexpr int NS = SAMEX ? 1 : NV;
    const double* sptr[NV]; double sr[NV]; T sacc[NS][VS]; double stot[NS][VS]; bool side_on = false;
    __device__ __forceinline__ NatLoader(const S* b, int64_t l, int t, const double* w_ = nullptr, int xl = 0,
                                         const double* s_ = nullptr)
        : step((int64_t)BK * l), xlim(xl), tid(t) {
        side_on = SIDE && s_ != nullptr;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * THREADS;
            const int k = v / VPR, xv = v % VPR;
            ptr[i] = b + (int64_t)k * l + xv * VS;
            wptr[i] = WEIGHT ? w_ + k : nullptr;
            sptr[i] = side_on ? s_ + k : nullptr;
            sr[i] = 0;
        }
#pragma unroll
        for (int i = 0; i < NS; ++i)
#pragma unroll
            for (int e = 0; e < VS; ++e) { sacc[i][e] = 0; stot[i][e] = 0; }
    }
    __device__ __forceinline__ void advance(int64_t delta) {   // move the source window (segmented main loop)
#pragma unroll
        for (int i = 0; i < NV; ++i) ptr[i] += delta;
    }
    template <int SET = 0>
    __device__ __forceinline__ void load(int) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * THREADS;
            const int xv = v % VPR;
            const bool ok = (BK * VPR) % THREADS == 0 || v < BK * VPR;
            vec_t val;
#pragma unroll
            for (int e = 0; e < VS; ++e) val[e] = 0;
            if (ok && (!GUARD || xv * VS < xlim)) val = *reinterpret_cast<const vec_t*>(ptr[i]);
            r[SET][i] = val;
            if (WEIGHT) { wr[SET][i] = ok ? *wptr[i] : 0.0; wptr[i] += BK; }
            if (SIDE && side_on) { sr[i] = ok ? *sptr[i] : 0.0; sptr[i] += BK; }
            ptr[i] += step;
        }
    }
    template <int SET = 0>
    __device__ __forceinline__ void store(T* s) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * THREADS;
            if ((BK * VPR) % THREADS != 0 && v >= BK * VPR) continue;
            const int k = v / VPR, xv = v % VPR;
            T* d = s + k * LD + xv * VS;
            constexpr int TV = 16 / (int)sizeof(T) < VS ? 16 / (int)sizeof(T) : VS;   // elements per LDS store
            typedef T tv_t __attribute__((ext_vector_type(TV)));
            const T wt = WEIGHT ? (T)wr[SET][i] : (T)1;
            T val[VS];
#pragma unroll
            for (int e = 0; e < VS; ++e) val[e] = (T)r[SET][i][e];
#pragma unroll
            for (int e0 = 0; e0 < VS; e0 += TV) {
                tv_t o;
#pragma unroll
                for (int e = 0; e < TV; ++e) o[e] = WEIGHT ? val[e0 + e] * wt : val[e0 + e];
                *reinterpret_cast<tv_t*>(d + e0) = o;
            }
            if (SIDE && side_on) {
                const T sv = (T)sr[i];
#pragma unroll
                for (int e = 0; e < VS; ++e) sacc[SAMEX ? 0 : i][e] += sv * val[e];
            }
        }
    }
    __device__ __forceinline__ void side_flush() {
#pragma unroll
        for (int i = 0; i < NS; ++i)
#pragma unroll
            for (int e = 0; e < VS; ++e) { stot[i][e] += (double)sacc[i][e]; sacc[i][e] = 0; }
    }
    // out[x] = side[x], x < BX, summed over this workgroup's BK row groups through lds (BK*BX doubles).
    // Every thread of the workgroup must call; lds must not be in use by the main loop any more.
    __device__ __forceinline__ void side_reduce(double* lds, double* __restrict__ out) const {
        constexpr int ROWS = SAMEX ? (THREADS / VPR < BK ? THREADS / VPR : BK) : BK;      // partial sums per column
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int v = tid + i * THREADS;
            if (v >= ROWS * VPR) continue;
            const int k = v / VPR, xv = v % VPR;
#pragma unroll
            for (int e = 0; e < VS; ++e) lds[k * BX + xv * VS + e] = stot[i][e];
        }
        __syncthreads();
        for (int x = tid; x < BX; x += THREADS) {
            double sum = 0;
#pragma unroll
            for (int k = 0; k < ROWS; ++k) sum += lds[k * BX + x];
            out[x] = sum;
        }
        __syncthreads();
    }
};

// ---------------------------------------------------------------------------
// TrLoader: source S[x][k] (row-major, k contiguous); transposes into s[k][x].
//   With DOT and a vector d the loader also forms, in fp64 and from the values it stages anyway, its share of
//   the row dots dot[x] = sum_k S[x][k] d[k]: k-tiles kt with kt % nparts == part (mu = Phi.alpha rides along
//   with Phi.B, one slice per column tile -- equal work in every workgroup keeps the workgroups that share an
//   operand panel in step, which is what makes them share it in L2).
// ---------------------------------------------------------------------------
//   SWZ (BK = 16): element (k, x) is stored at column x ^ SZ*(k/VS), SZ = 8 (fp32, VS = 4) or 2 (fp64, VS = 2).
//   A thread holds VS consecutive k of one row, so the threads of a row write k/VS = 0, 1, ...: without the
//   swizzle their banks coincide (the row stride is a multiple of 16 elements) and every transposing store is
//   a 4-way (fp32) or 8-way (fp64) bank conflict; with it the lanes of a store group spread over all banks.
//   Fragment reads XOR the same constant (tile_compute): within a 16-lane group it only permutes the group's 16
//   columns, so reads stay conflict-free.
template <typename S, typename T, int BX, int BK, int LD, int THREADS, bool DOT = false, bool SWZ = false, int NSETS = 1>
struct TrLoader {
    typedef typename Vec16<S>::type vec_t;
    static constexpr int VS = Vec16<S>::N;
    static constexpr int VPR = BK / VS;                       // vectors per x row
    static constexpr int NV = (BX * VPR + THREADS - 1) / THREADS;
    static_assert(!DOT || (THREADS % VPR == 0 && (VPR & (VPR - 1)) == 0 && VPR <= 64), "DOT: a row's vectors sit in adjacent lanes");
    const S* ptr[NV]; int tid;
    vec_t r[NSETS][NV];
    const double* dptr = nullptr; double dv[VS]; double dacc[NV]; bool dot_on = false, dot_now = false;
    int dpart = 0, dnparts = 1, dphase = 0;                     // dphase: (index of the k-tile being loaded) % dnparts
    int dlo = 0, dhi = -1, dkt = 0;                             // range mode (dot_range): k-tiles [dlo, dhi) instead of the modulo rule
    __device__ __forceinline__ void dot_range(int lo, int hi) { dlo = lo; dhi = hi; }
    __device__ __forceinline__ TrLoader(const S* b, int64_t l, int t, const double* d_ = nullptr, int part = 0, int nparts = 1)
        : tid(t), dpart(part), dnparts(nparts) {
        dot_on = DOT && d_ != nullptr;
        if (dot_on) dptr = d_ + (t % VPR) * VS;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * THREADS;
            ptr[i] = b + (int64_t)(v / VPR) * l + (v % VPR) * VS;
            dacc[i] = 0;
        }
    }
    __device__ __forceinline__ void advance(int64_t delta) {   // move the source window (segmented main loop)
#pragma unroll
        for (int i = 0; i < NV; ++i) ptr[i] += delta;
    }
    template <int SET = 0>
    __device__ __forceinline__ void load(int) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * THREADS;
            const bool ok = (BX * VPR) % THREADS == 0 || v < BX * VPR;
            vec_t val;
#pragma unroll
            for (int e = 0; e < VS; ++e) val[e] = 0;
            if (ok) val = *reinterpret_cast<const vec_t*>(ptr[i]);
            r[SET][i] = val;
            ptr[i] += BK;
        }
        if (DOT && dot_on) {
            dot_now = dhi >= 0 ? (dkt >= dlo && dkt < dhi) : dphase == dpart;
            ++dkt;
            if (dot_now) {
#pragma unroll
                for (int e = 0; e < VS; ++e) dv[e] = dptr[e];
            }
            dptr += BK;
            dphase = dphase + 1 == dnparts ? 0 : dphase + 1;
        }
    }
    template <int SET = 0>
    __device__ __forceinline__ void store(T* s) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * THREADS;
            if ((BX * VPR) % THREADS != 0 && v >= BX * VPR) continue;
            const int x = v / VPR, kv = v % VPR;
#pragma unroll
            for (int e = 0; e < VS; ++e) s[(kv * VS + e) * LD + (SWZ ? x ^ ((sizeof(T) == 4 ? 8 : 2) * kv) : x)] = (T)r[SET][i][e];
            if (DOT && dot_on && dot_now) {
#pragma unroll
                for (int e = 0; e < VS; ++e) dacc[i] = fma((double)r[SET][i][e], dv[e], dacc[i]);
            }
        }
    }
    // out[x] = dot[x] for the BX rows of this tile (all threads call)
    __device__ __forceinline__ void dot_reduce(double* __restrict__ out) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            double s = dacc[i];
#pragma unroll
            for (int m = 1; m < VPR; m <<= 1) s += __shfl_xor(s, m);
            const int v = tid + i * THREADS;
            if (((BX * VPR) % THREADS == 0 || v < BX * VPR) && v % VPR == 0) out[v / VPR] = s;
        }
    }
};

// ---------------------------------------------------------------------------
// ZbarLoader: B operand of X~^T Zbar, formed on the fly from the resident Phi and Phibar:
//   s[k=n][x=j] = Phi[n][j] * Phibar[n][J+j] - Phi[n][J+j] * Phibar[n][j]      (0 for j >= J)
// the cotangent of the phase argument (cos' = -sin, sin' = cos), so no N x J buffer exists.
// ---------------------------------------------------------------------------
//   NSETS: register sets for fetches more than one k-tile ahead (tile_mainloop_deep3), as in NatLoader
template <typename S, typename T, int BX, int BK, int LD, int THREADS, int NSETS = 1>
struct ZbarLoader {
    static constexpr int VS = Vec16<S>::N;
    static constexpr int VPR = BX / VS;
    static constexpr int NV = (BK * VPR + THREADS - 1) / THREADS;
    const S* phi; const S* pb; int64_t ld; int J, j0, tid; bool vec; int kt = 0;   // consecutive k-tiles
    typedef typename Vec16<S>::type vec_t;
    vec_t raw[NSETS][NV][4];                                  // fc, fs, bc, bs: combined in store(), after the MFMAs
    __device__ __forceinline__ ZbarLoader(const S* phi_, const S* pb_, int64_t ld_, int J_, int j0_, int t)
        : phi(phi_), pb(pb_), ld(ld_), J(J_), j0(j0_), tid(t), vec(J_ % VS == 0) {}
    template <int SET = 0>
    __device__ __forceinline__ void load(int) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * THREADS;
            const int k = v / VPR, j = j0 + (v % VPR) * VS;
            const bool ok = (BK * VPR) % THREADS == 0 || v < BK * VPR;
            const S* f = phi + (int64_t)(kt * BK + k) * ld;
            const S* b = pb + (int64_t)(kt * BK + k) * ld;
            if (ok && vec && j + VS <= J) {
                raw[SET][i][0] = *reinterpret_cast<const vec_t*>(f + j); raw[SET][i][1] = *reinterpret_cast<const vec_t*>(f + J + j);
                raw[SET][i][2] = *reinterpret_cast<const vec_t*>(b + j); raw[SET][i][3] = *reinterpret_cast<const vec_t*>(b + J + j);
            } else {
#pragma unroll
                for (int e = 0; e < VS; ++e) {
                    const bool in = ok && j + e < J;
                    raw[SET][i][0][e] = in ? f[j + e] : (S)0; raw[SET][i][1][e] = in ? f[J + j + e] : (S)0;
                    raw[SET][i][2][e] = in ? b[j + e] : (S)0; raw[SET][i][3][e] = in ? b[J + j + e] : (S)0;
                }
            }
        }
        ++kt;
    }
    template <int SET = 0>
    __device__ __forceinline__ void store(T* s) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * THREADS;
            if ((BK * VPR) % THREADS != 0 && v >= BK * VPR) continue;
            T* d = s + (v / VPR) * LD + (v % VPR) * VS;
#pragma unroll
            for (int e = 0; e < VS; ++e) d[e] = (T)raw[SET][i][0][e] * (T)raw[SET][i][3][e] - (T)raw[SET][i][1][e] * (T)raw[SET][i][2][e];
        }
    }
};

// MFMAs of one k-tile out of LDS buffer `cur`
//   tid: the thread's id; a caller that runs many tiles in one launch passes its own (opaque) copy so that the addresses derived
//   from it are not hoisted out of its job loop (gram.hip)
template <class Cfg>
__device__ __forceinline__ void tile_compute(const typename Cfg::T* sA, const typename Cfg::T* sB,
                                             typename Cfg::MTr::acc_t (&acc)[Cfg::TM][Cfg::TN], int tid) {
    typedef typename Cfg::T T;
    typedef typename Cfg::MTr M;
    constexpr int MS = Cfg::MS, KS = M::KS;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / Cfg::WGN) * Cfg::WM, wn0 = (wave % Cfg::WGN) * Cfg::WN;
    const T* a_s = sA + (lane / MS) * Cfg::LDA + wm0 + (lane % MS);
    const T* b_s = sB + (lane / MS) * Cfg::LDB + wn0 + (lane % MS);
    // SWZA: lane (q = lane/16, i = lane%16) reads row k = 4 kk + q of k-step kk at column (..+i) ^ SZ*(k/VS):
    //   fp32: 8 kk          -> i ^ 8 for odd kk, neighbouring 16-column group for kk >= 2
    //   fp64: 4 kk + 2(q/2) -> stays inside the lane's own 16-column group
    const int q = lane / MS;
    const T* a_x = sA + q * Cfg::LDA + wm0 + ((lane % MS) ^ 8);                  // fp32, odd kk
#pragma unroll
    for (int kk = 0; kk < Cfg::BK / KS; ++kk) {
        T a[Cfg::TM], b[Cfg::TN];
#pragma unroll
        for (int tm = 0; tm < Cfg::TM; ++tm) {
            if (!Cfg::SWZA) a[tm] = a_s[kk * KS * Cfg::LDA + tm * MS];
            else if (sizeof(T) == 4) a[tm] = ((kk & 1) ? a_x : a_s)[kk * KS * Cfg::LDA + (tm ^ ((kk >> 1) & 1)) * MS];
            else a[tm] = sA[(kk * KS + q) * Cfg::LDA + wm0 + tm * MS + (((lane % MS) ^ (2 * (q >> 1))) ^ (4 * kk))];
        }
#pragma unroll
        for (int tn = 0; tn < Cfg::TN; ++tn) b[tn] = b_s[kk * KS * Cfg::LDB + tn * MS];
#pragma unroll
        for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < Cfg::TN; ++tn) M::mfma(acc[tm][tn], a[tm], b[tn]);
    }
}

// ---------------------------------------------------------------------------
// main loop: accumulates nkt k-tiles into acc[TM][TN]; smem = 2*(SA+SB) elements.
// LDS is double buffered (buffer = k-tile parity), one barrier per k-tile.
// ---------------------------------------------------------------------------
template <class Cfg, class LA, class LB>
__device__ __forceinline__ void tile_mainloop(LA& la, LB& lb, int nkt,
                                              typename Cfg::MTr::acc_t (&acc)[Cfg::TM][Cfg::TN],
                                              typename Cfg::T* smem, int tid) {
    typedef typename Cfg::T T;
    T* sA = smem;
    T* sB = smem + 2 * Cfg::SA;
    la.template load<0>(0); lb.template load<0>(0);
    la.template store<0>(sA); lb.template store<0>(sB);
    __syncthreads();
    // Rolled loop, run-time buffer index.  Measured alternatives that were equal or slower on MI355X
    // (profiles/r01_tuning.md): global loads two k-tiles ahead with two register sets, LDS stores
    // placed mid-tile, scheduler interleave hints, BK = 32, parity-unrolled body.
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const bool stage = kt + 1 < nkt;
        if (stage) { la.template load<0>(kt + 1); lb.template load<0>(kt + 1); }
        tile_compute<Cfg>(sA + cur * Cfg::SA, sB + cur * Cfg::SB, acc, tid);
        if (stage) { la.template store<0>(sA + (cur ^ 1) * Cfg::SA); lb.template store<0>(sB + (cur ^ 1) * Cfg::SB); }
        __syncthreads();
    }
}
template <class Cfg, class LA, class LB>
__device__ __forceinline__ void tile_mainloop(LA& la, LB& lb, int nkt, typename Cfg::MTr::acc_t (&acc)[Cfg::TM][Cfg::TN],
                                              typename Cfg::T* smem) {
    tile_mainloop<Cfg>(la, lb, nkt, acc, smem, (int)threadIdx.x);
}

// Main loop for products that run ONE workgroup per CU (the K x K stage): the operand fetch of k-tile t+3 is issued while
// tile t is multiplied (three register sets in the loaders, NSETS = 3), so three fetch latencies overlap instead of one;
// LDS stays double buffered, one barrier per k-tile.
template <class Cfg, int S, class LA, class LB>
__device__ __forceinline__ void tile_deep3_step(LA& la, LB& lb, int t, int nkt, typename Cfg::MTr::acc_t (&acc)[Cfg::TM][Cfg::TN],
                                                typename Cfg::T* sA, typename Cfg::T* sB) {
    const int cur = t & 1;
    tile_compute<Cfg>(sA + cur * Cfg::SA, sB + cur * Cfg::SB, acc, (int)threadIdx.x);
    if (t + 1 < nkt) { la.template store<(S + 1) % 3>(sA + (cur ^ 1) * Cfg::SA); lb.template store<(S + 1) % 3>(sB + (cur ^ 1) * Cfg::SB); }
    if (t + 3 < nkt) { la.template load<S>(t + 3); lb.template load<S>(t + 3); }     // set S: tile t's, stored an iteration ago
    __syncthreads();
}
template <class Cfg, class LA, class LB>
__device__ __forceinline__ void tile_mainloop_deep3(LA& la, LB& lb, int nkt, typename Cfg::MTr::acc_t (&acc)[Cfg::TM][Cfg::TN],
                                                    typename Cfg::T* smem) {
    typedef typename Cfg::T T;
    T* sA = smem;
    T* sB = smem + 2 * Cfg::SA;
    if (nkt <= 0) return;
    la.template load<0>(0); lb.template load<0>(0);
    if (nkt > 1) { la.template load<1>(1); lb.template load<1>(1); }
    if (nkt > 2) { la.template load<2>(2); lb.template load<2>(2); }
    la.template store<0>(sA); lb.template store<0>(sB);
    __syncthreads();
    for (int t = 0; t < nkt; t += 3) {
        tile_deep3_step<Cfg, 0>(la, lb, t, nkt, acc, sA, sB);
        if (t + 1 < nkt) tile_deep3_step<Cfg, 1>(la, lb, t + 1, nkt, acc, sA, sB);
        if (t + 2 < nkt) tile_deep3_step<Cfg, 2>(la, lb, t + 2, nkt, acc, sA, sB);
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

// Segmented main loop: nseg products of nkt k-tiles each run as ONE stream of k-tiles, so the operand fetch of a
// segment's first k-tile is in flight while the previous segment is multiplied and flushed (short contractions would
// otherwise be all prologue).  Before the first k-tile of segment s > 0 is fetched the loaders move by (dA, dB)
// elements; flush(s, acc) consumes the finished accumulators (registers and global memory only: the LDS buffers
// already belong to the next segment) and the accumulators restart from zero.
template <class Cfg, class LA, class LB, class F>
__device__ __forceinline__ void tile_mainloop_segments(LA& la, LB& lb, int nkt, int nseg, int64_t dA, int64_t dB,
                                                       typename Cfg::MTr::acc_t (&acc)[Cfg::TM][Cfg::TN],
                                                       typename Cfg::T* smem, F&& flush) {
    typedef typename Cfg::T T;
    T* sA = smem;
    T* sB = smem + 2 * Cfg::SA;
    la.template load<0>(0); lb.template load<0>(0);
    la.template store<0>(sA); lb.template store<0>(sB);
    __syncthreads();
    const int total = nkt * nseg;
    int kin = 0, seg = 0;                                       // k-tile inside the segment, segment
    for (int t = 0; t < total; ++t) {
        const int cur = t & 1;
        const bool stage = t + 1 < total, last = kin + 1 == nkt;
        if (stage) {
            if (last) { la.advance(dA); lb.advance(dB); }
            la.template load<0>(t + 1); lb.template load<0>(t + 1);
        }
        tile_compute<Cfg>(sA + cur * Cfg::SA, sB + cur * Cfg::SB, acc, (int)threadIdx.x);
        if (last) { flush(seg, acc); acc_zero<Cfg>(acc); ++seg; kin = 0; } else ++kin;
        if (stage) { la.template store<0>(sA + (cur ^ 1) * Cfg::SA); lb.template store<0>(sB + (cur ^ 1) * Cfg::SB); }
        __syncthreads();
    }
}

// coordinates of accumulator element (tm,tn,r) inside the workgroup tile
template <class Cfg>
struct AccCoord {
    int lane, wm0, wn0;
    __device__ __forceinline__ explicit AccCoord(int tid = (int)threadIdx.x) {
        lane = tid & 63;
        const int wave = tid >> 6;
        wm0 = (wave / Cfg::WGN) * Cfg::WM; wn0 = (wave % Cfg::WGN) * Cfg::WN;
    }
    __device__ __forceinline__ int row(int tm, int r) const {
        return wm0 + tm * Cfg::MS + Cfg::MTr::crow(lane, r);
    }
    __device__ __forceinline__ int col(int tn) const {
        return wn0 + tn * Cfg::MS + (lane % Cfg::MS);
    }
};

// XCD-aware work-item id (guide T1): hardware deals consecutive workgroup ids round-robin over
// the 8 XCDs, so ids b and b+8 share an L2.  Mapping id b to the b/8-th item of a contiguous
// chunk owned by XCD b%8 makes workgroups that share operand panels (same rows, different
// column tiles) run on ONE XCD: the panel crosses the fabric once instead of up to 8 times.
// Placement is only a speed matter; the map is a bijection for any n.
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned n) {
#if SCFGP_NO_XCD_REMAP
    return b;
#else
    const unsigned q = n / 8, r = n % 8, xcd = b % 8, local = b / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
#endif
}
