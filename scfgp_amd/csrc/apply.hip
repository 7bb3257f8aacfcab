// NT products of the SCFGP objective (contraction over the feature columns):  V = Phi B, Phibar = 2 Phi Abar + ...,
// the triangular products of the factor form and of predict (SCFGP/SCFGP.py:111-113, :143-144 and their backward),
// the per-row statistics that follow them, and the scalar reductions.  Built on tile_engine.h.
#include "kernels.h"
#include "tile_cfgs.h"

#include <algorithm>

// apply tiles: 256 rows x {256, 128, 64} columns (tile_cfgs.h)
// Diagnostic build (-DSCFGP_TRACE): every workgroup of the LDS-DMA apply kernel records [start, end] on the 100 MHz constant
// clock and its XCC id (tools/apply_trace.py); the product build contains none of this
#ifdef SCFGP_TRACE
constexpr int ATRACE_CAP = 1 << 16;                         // per epilogue kind (EPI 0 .. 4): the launches of one stage do not overwrite each other's
__device__ unsigned long long g_atrace[5 * ATRACE_CAP][5];  // start, end, xcc | column tile << 8 | stages << 20, first barrier passed, k loop left
int64_t apply_trace_read(void* host, int64_t max_bytes) {
    const int64_t n = max_bytes < (int64_t)sizeof(g_atrace) ? max_bytes : (int64_t)sizeof(g_atrace);
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_atrace), n) == hipSuccess ? n : -2;
}
#else
int64_t apply_trace_read(void*, int64_t) { return -1; }
#endif
template <typename T, int TILE> struct ApplyCfg {
    typedef TileCfg<T, Tune<T>::APPLY_BM, TILE, SCFGP_BK, Tune<T>::APPLY_WGM, Tune<T>::apply_wgn(TILE), Tune<T>::MS,
                    SCFGP_BK == 16 && Tune<T>::MS == 16> type;                 // swizzled Phi image (TrLoader)
};

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
#include "apply_epilogue.h"

// one output tile (column tile jt of this launch, row block rb) of the apply product, operands staged through registers
template <class Cfg, int EPI>
__device__ __forceinline__ void apply_tile(
    const typename Cfg::T* __restrict__ Phi, const typename Cfg::T* __restrict__ Bm, typename Cfg::T* V,
    double* __restrict__ vpart, const double* __restrict__ p, const double* __restrict__ q, const double* __restrict__ y,
    const double* __restrict__ alpha, const double* __restrict__ ut, int K, int Kp, int64_t Np, int njt,
    int col0, int jt0, double* __restrict__ mu, int ntot, int jt, int64_t rb, char* smem_raw) {
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
    typename Cfg::MTr::acc_t acc[Cfg::TM][Cfg::TN];
    acc_zero<Cfg>(acc);
    TrLoader<T, T, Cfg::BM, Cfg::BK, Cfg::LDA, Cfg::THREADS, EPI != 1 && EPI != 4, Cfg::SWZA> la(
        Phi + rb * Cfg::BM * Kp + kt0 * Cfg::BK, Kp, threadIdx.x, want_mu ? alpha : nullptr, jt0 + jt, ntot);
    if (EPI == 2 || EPI == 3) la.dot_range(cbase / Cfg::BK, nkt);
    else if (ntot == 0) {                                   // beside DMA-fed tiles: mu slices are the tiles' own column bands
        const int hi = (cbase + Cfg::BN) / Cfg::BK;
        la.dot_range(cbase / Cfg::BK, hi < nkt ? hi : nkt);
    }
    NatLoader<T, T, Cfg::BN, Cfg::BK, Cfg::LDB, Cfg::THREADS, false, false> lb(Bm + (int64_t)kt0 * Cfg::BK * Kp + cbase, Kp, threadIdx.x);
    tile_mainloop<Cfg>(la, lb, nkt, acc, smem);
    if constexpr (EPI != 1 && EPI != 4) { if (want_mu) la.dot_reduce(mu + (int64_t)(jt0 + jt) * Np + rb * Cfg::BM); }
    apply_epilogue<Cfg, EPI>(acc, Phi, V, vpart, p, q, y, alpha, ut, K, Kp, Np, rb, cbase, jt0 + jt, smem_raw);
}

template <class Cfg, int EPI>
__global__ __launch_bounds__(Cfg::THREADS) __attribute__((amdgpu_waves_per_eu(1, 8)))
void apply_kernel(
    const typename Cfg::T* __restrict__ Phi, const typename Cfg::T* __restrict__ Bm, typename Cfg::T* V,
    double* __restrict__ vpart, const double* __restrict__ p, const double* __restrict__ q, const double* __restrict__ y,
    const double* __restrict__ alpha, const double* __restrict__ ut, int K, int Kp, int64_t Np, int njt,
    int col0, int jt0, double* __restrict__ mu, int ntot) {
    // this launch covers columns [col0, col0 + njt*BN); jt0 = index of its first tile in vpart
    SMEM_DECL;
    const unsigned wid = xcd_remap(blockIdx.x, gridDim.x);
    if constexpr (EPI == 3 || EPI == 4) {
        // triangular operand: column tile jt contracts over (jt + 1) / njt (EPI 3) or (njt - jt) / njt (EPI 4) of the k range, so a
        // workgroup takes tile t AND tile njt - 1 - t -- every workgroup of the launch then does the same amount of work
        const int np = (njt + 1) / 2, t = (int)(wid % np), o = njt - 1 - t;
        const int64_t rb = wid / np;
        apply_tile<Cfg, EPI>(Phi, Bm, V, vpart, p, q, y, alpha, ut, K, Kp, Np, njt, col0, jt0, mu, ntot, t, rb, smem_raw);
        if (o != t) {
            __syncthreads();                                     // the first tile's epilogue is done with the LDS
            apply_tile<Cfg, EPI>(Phi, Bm, V, vpart, p, q, y, alpha, ut, K, Kp, Np, njt, col0, jt0, mu, ntot, o, rb, smem_raw);
        }
    } else
        apply_tile<Cfg, EPI>(Phi, Bm, V, vpart, p, q, y, alpha, ut, K, Kp, Np, njt, col0, jt0, mu, ntot, (int)(wid % njt), wid / njt, smem_raw);
}

// Apply product with LDS-DMA staging: both operands go global -> LDS by global_load_lds_dwordx4 into a ring of three stages of
// 16 k each, counted vmcnt, one raw barrier per stage: no staging registers, no ds_write, and the fetches of stages s+2 and s+3
// are in flight while stage s is multiplied.  Bm must hold the k-contiguous COLUMNS of the operand as its rows: the matrix itself
// when it is symmetric (B and Abar are), else its transpose (factor form: Li for C = Phi Li^T, Li^T for V = C Li).
//   LDS image of an operand: row x at x * ROWB bytes (ROWB = 16 k: 64 B in fp32, 128 B in fp64) = its 16 k as CPR 16-byte
//   chunks, chunk c stored at position c ^ swz(x) (source-side swizzle: the DMA itself writes linearly, 1 KiB per wave
//   instruction).  The 16 k of a stage are used as two HALVES of 8: in half h lane group q of the 16x16x4 shape holds 2 k and
//   feeds component j to k-step 2h + j -- both operands use the same permutation of the 16 k, so the sum is unchanged.
//     fp32: the 8 bytes (k = 8h + 2q, +1) at offset 8 (q & 1) of chunk 2h + (q >> 1): one ds_read_b64, served in two 32-lane
//           halves that touch 64 distinct banks each (16 rows x 16 bytes: rows 4 apart share banks and differ in swz);
//     fp64: chunk 2q + h (k = 4q + 2h, +1): one ds_read_b128, served in four 16-lane groups ({0-3,12-15,20-27},
//           {4-11,16-19,28-31} and the same + 32: MI355X_MICROARCH.md, LDS) that touch 16 distinct 16-byte slots each.
//   B rows are staged in the order that leaves an MFMA lane TN ADJACENT output columns (LDS row tn*16 + i of a wave tile holds
//   operand row TN i + tn), so the epilogue moves V, Phi and Phibar 16 bytes per lane.
//   EPI 3 / 4 (factor form): the stages run over k < cbase + BN (EPI 3) or k >= cbase (EPI 4) only.
// Software pipeline across the barrier (apply_dma_kernel's k loop): the barrier that opens stage s+1 stands in the MIDDLE of
// the arithmetic -- a wave reaches it with the first half of stage s multiplied and the second half's fragments in registers,
// so it leaves the barrier with 2 TM TN MFMAs that need nothing from the LDS.  Between those MFMAs, one instruction per MFMA,
// ride the fetch of stage s+3 (into the slot the barrier just freed) and the reads of the first half of stage s+1; between the
// MFMAs of that half ride the reads of its second half.  Every read has >= 20 MFMAs between issue and first use, the matrix
// pipe never sees all waves of the CU waiting for the LDS at once (measured: k loop of a 256 x 256 fp32 tile 494 -> 466 us,
// profiles/r04_tuning.md), and the first SCFGP-PRE MFMAs of the second half are issued in front of the barrier so the pipe
// has work while it resolves.
//   The LDS reads are inline assembly with hand-counted waits: the compiler's own lgkmcnt bookkeeping gives up on a load that
//   is in flight across a loop's back edge (it waits for ALL outstanding reads at the first use, the just-issued ones too).
//   Each wait statement names the fragment registers it releases as in/out operands, so neither an MFMA nor a register copy
//   of them can be placed above it; sched_barrier(0) keeps the written order.  tools/isa_inflight.py replays the compiled
//   instruction stream and fails if anything touches a destination of a read not yet covered by a wait (tests/test_isa_lint.py).
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
template <typename T, int BN_>
struct ApplyDma {
    static constexpr int ES = (int)sizeof(T), BM = 256, BN = BN_, ROWB = 16 * ES, CPR = ROWB / 16;
    static constexpr int WN = ES == 4 ? 64 : 32;               // wave tile 64 x 64 (fp32) / 64 x 32 (fp64): 64 accumulator registers
    static constexpr int WAVES = 4 * (BN / WN), STAGE = (BM + BN) * ROWB, STAGES = 3, LDS_BYTES = STAGES * STAGE,
                         DMA_PER_WAVE = STAGE / 1024 / WAVES, ROWS_PER_DMA = 1024 / ROWB;
    static constexpr int PRE = 2;                              // MFMAs of a stage's second half issued in front of the barrier
    // resident waves per SIMD the register budget is cut for: one 16-wave workgroup or two 8-wave ones per CU, two 4-wave ones
    // of the 64-column remainder tile (the LDS holds no more)
    static constexpr int WAVES_PER_EU = WAVES == 4 ? 2 : 4;
    typedef TileCfg<T, BM, BN, 16, 4, BN / WN, 16, true> Cfg;      // wave grid / accumulator map of the epilogue
    static_assert(STAGE / 1024 % WAVES == 0 && LDS_BYTES <= 160 * 1024, "whole DMA instructions per wave; the ring fits the LDS");
    // position swizzle of row x (see above): fp32 f[(x >> 2) & 3], f = (0, 2, 3, 1); fp64 f[(x >> 1) & 7], f = (0, 1, 4, 5, 6, 7, 2, 3)
    static __device__ __forceinline__ int swz(int x) {
        if constexpr (ES == 4) return (0x78 >> (2 * ((x >> 2) & 3))) & 3;
        else return (int)((0x6BEB08u >> (3 * ((x >> 1) & 7))) & 7);
    }
};
static_assert(((0x6BEB08u >> 0) & 7) == 0 && ((0x6BEB08u >> 3) & 7) == 1 && ((0x6BEB08u >> 6) & 7) == 4 && ((0x6BEB08u >> 9) & 7) == 5 &&
              ((0x6BEB08u >> 12) & 7) == 6 && ((0x6BEB08u >> 15) & 7) == 7 && ((0x6BEB08u >> 18) & 7) == 2 && ((0x6BEB08u >> 21) & 7) == 3,
              "fp64 swizzle table");

// fp32 BN = 256: 16 waves, one workgroup per CU, 32 operand bytes per MFMA; BN = 128: 8 waves, two workgroups per CU, 48 bytes;
// fp64 BN = 128: 16 waves of 64 x 32, one workgroup per CU (3 x 48 KB of LDS), 96 bytes per MFMA of twice the duration;
// BN = 64 (the ragged last columns): 4 (fp32) / 8 (fp64) waves, 80 / 160 bytes per MFMA
template <typename T, int EPI, int BN>
__global__ __launch_bounds__((64 * ApplyDma<T, BN>::WAVES)) __attribute__((amdgpu_waves_per_eu(ApplyDma<T, BN>::WAVES_PER_EU, ApplyDma<T, BN>::WAVES_PER_EU)))
void apply_dma_kernel(const T* __restrict__ Phi, const T* __restrict__ Bm, T* V,
                      double* __restrict__ vpart, const double* __restrict__ p, const double* __restrict__ q,
                      const double* __restrict__ y, const double* __restrict__ alpha, const double* __restrict__ ut,
                      int K, int Kp, int64_t Np, int njt, double* __restrict__ mu, int col0, int slot0, int64_t rb0) {
    typedef ApplyDma<T, BN> D;
    typedef typename D::Cfg Cfg;
    typedef T half_t __attribute__((ext_vector_type(2)));      // a lane's 2 k of one half of a stage
    typedef std::integral_constant<int, 0> H0;
    typedef std::integral_constant<int, 1> H1;
    constexpr int DPW = D::DMA_PER_WAVE, NRD = Cfg::TM + Cfg::TN, NM = 2 * Cfg::TM * Cfg::TN, PRE = D::PRE;   // per half: reads, MFMAs
    static_assert(Cfg::TM == 4 && (Cfg::TN == 4 || Cfg::TN == 2), "fragment lists of the wait statements");
    static_assert(PRE + DPW + NRD <= NM, "one fetch or read per MFMA behind the barrier");
    SMEM_DECL;
    char* smem = smem_raw;
#ifdef SCFGP_TRACE
    const unsigned long long tr_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    const unsigned wid = xcd_remap(blockIdx.x, gridDim.x);
    // Triangular products (EPI 3 / 4): a column tile's k range, and with it the workgroup's duration, grows with jt.  The
    // hardware deals the workgroups of an XCD to its four shader engines round-robin in launch order and balances only inside
    // an engine, so with jt = wid % njt (njt a multiple of 4) one engine got every longest tile and one every shortest -- a
    // quarter of the workgroup slots idle for the whole launch, and a k loop per stage that is periodic in jt mod 4
    // (profiles/r05_apply_tri_trace.txt).  Rotating the column tile by the row block hands every engine every length in turn.
    const int jt = EPI == 3 || EPI == 4 ? (int)((wid % njt + wid / njt) % njt) : (int)(wid % njt);
    const int64_t rb = rb0 + wid / njt;                        // the launch covers row blocks rb0 ..
    const int cbase = col0 + jt * D::BN;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // DMA instruction t = DMA_PER_WAVE wave + u of a stage: rows ROWS_PER_DMA t .. of the stacked (A: 256, then B: BN) operand
    // rows; lane l carries row + l / CPR, position l % CPR <- chunk (l % CPR) ^ swz(row).  Its address is a wave-uniform row
    // pointer + a 32-bit lane offset; the LDS address (M0) is scalar arithmetic.
    const char* src[DPW]; unsigned loff[DPW]; int dst[DPW];
#pragma unroll
    for (int u = 0; u < DPW; ++u) {
        const int t = wave * DPW + u, x0 = D::ROWS_PER_DMA * t, l = lane / D::CPR, c = (lane % D::CPR) ^ D::swz(x0 + l);
        const int xb = x0 - D::BM, xcol = (xb & ~(D::WN - 1)) + Cfg::TN * (xb & 15) + ((xb >> 4) & (Cfg::TN - 1));
        // B rows: LDS row xb + l holds operand row xcol + TN l (a DMA instruction's rows lie inside one group of 16)
        const T* rowp = x0 < D::BM ? Phi + (rb * D::BM + x0) * Kp : Bm + (int64_t)(cbase + xcol) * Kp;
        src[u] = reinterpret_cast<const char*>(rowp) + (EPI == 4 ? (cbase / 16) * D::ROWB : 0);
        loff[u] = (unsigned)((x0 < D::BM ? l : Cfg::TN * l) * Kp * D::ES + c * 16);
        dst[u] = t * 1024;
    }
    const auto fetch_one = [&](auto uc, int slot) {
        constexpr int u = decltype(uc)::value;
        __builtin_amdgcn_global_load_lds((gbl_void*)(src[u] + loff[u]), (lds_void*)(smem + slot * D::STAGE + dst[u]), 16, 0, 0);
        src[u] += D::ROWB;                                      // 16 k further
    };
    const int i = lane & 15, qg = lane >> 4;
    const int wm0 = (wave / Cfg::WGN) * Cfg::WM, wn0 = (wave % Cfg::WGN) * Cfg::WN;
    // LDS byte addresses of this lane's 8 (fp32) / 16 (fp64) bytes of half 0 / 1 in the first fragment row of the wave's A and B
    // tiles (the wave tile bases are multiples of 16 rows: swz(row) = swz(i))
    const int hoff0 = sizeof(T) == 4 ? (((qg >> 1) ^ D::swz(i)) << 4) + 8 * (qg & 1) : ((2 * qg) ^ D::swz(i)) << 4;
    const int hoff1 = sizeof(T) == 4 ? (((2 + (qg >> 1)) ^ D::swz(i)) << 4) + 8 * (qg & 1) : ((2 * qg + 1) ^ D::swz(i)) << 4;
    const int ring = (int)(uintptr_t)smem, aoff = ring + (wm0 + i) * D::ROWB, boff = ring + D::BM * D::ROWB + (wn0 + i) * D::ROWB;
    const int ra[2] = {aoff + hoff0, aoff + hoff1}, rbb[2] = {boff + hoff0, boff + hoff1};
    half_t ha[2][Cfg::TM], hb[2][Cfg::TN];
    typename Cfg::MTr::acc_t acc[Cfg::TM][Cfg::TN];
    acc_zero<Cfg>(acc);
    const int nst_all = (K + 15) / 16, nst_tri = (cbase + D::BN + 15) / 16;
    const int nst = (EPI == 3 && nst_tri < nst_all ? nst_tri : nst_all) - (EPI == 4 ? cbase / 16 : 0);
#define SCFGP_LDS_READ(dst, addr, off)                                                                              \
    do {                                                                                                             \
        if constexpr (sizeof(T) == 4) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off));   \
        else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off));                           \
    } while (0)
    // read R (A fragments 0 .. TM-1, then B fragments) of half h of the stage at byte `stage` of the ring
    const auto read_one = [&](auto hc, auto rc, int stage) {
        constexpr int h = decltype(hc)::value, R = decltype(rc)::value;
        if constexpr (R < Cfg::TM) SCFGP_LDS_READ(ha[h][R], stage + ra[h], R * 16 * D::ROWB);
        else SCFGP_LDS_READ(hb[h][R - Cfg::TM], stage + rbb[h], (R - Cfg::TM) * 16 * D::ROWB);
    };
#undef SCFGP_LDS_READ
    // "s_waitcnt <what>" that hands out the fragments of half h
#define SCFGP_WAIT_FRAGS(what, h)                                                                                                   \
    do {                                                                                                                            \
        if constexpr (Cfg::TN == 4)                                                                                                 \
            asm volatile("s_waitcnt " what : "+v"(ha[h][0]), "+v"(ha[h][1]), "+v"(ha[h][2]), "+v"(ha[h][3]),                        \
                                             "+v"(hb[h][0]), "+v"(hb[h][1]), "+v"(hb[h][2]), "+v"(hb[h][3]) :: "memory");           \
        else                                                                                                                        \
            asm volatile("s_waitcnt " what : "+v"(ha[h][0]), "+v"(ha[h][1]), "+v"(ha[h][2]), "+v"(ha[h][3]),                        \
                                             "+v"(hb[h][0]), "+v"(hb[h][1]) :: "memory");                                           \
    } while (0)
    // MFMA I of half h: k-step 2h + j of accumulator tile (tm, tn), I = (j TM + tm) TN + tn
    const auto mfma_one = [&](auto hc, auto ic) {
        constexpr int h = decltype(hc)::value, I = decltype(ic)::value, j = I / (Cfg::TM * Cfg::TN), tm = I / Cfg::TN % Cfg::TM, tn = I % Cfg::TN;
        Cfg::MTr::mfma(acc[tm][tn], ha[h][tm][j], hb[h][tn][j]);
        __builtin_amdgcn_sched_barrier(0);
    };
    // the first PRE MFMAs of the second half of stage s (in front of the barrier), then the rest with the fetch of stage s+3
    // (FETCH; into slot fslot) and the reads of the first half of stage s+1 (at byte `next` of the ring) between them
    const auto second_half_pre = [&]() { static_for<PRE>([&](auto ic) { mfma_one(H1(), ic); }); };
    const auto second_half = [&](auto fc, int fslot, int next) {
        static_for<NM - PRE>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            mfma_one(H1(), std::integral_constant<int, I + PRE>());
            if constexpr (I < DPW) {
                if constexpr (decltype(fc)::value) { fetch_one(ic, fslot); __builtin_amdgcn_sched_barrier(0); }
            } else if constexpr (I - DPW < NRD) {
                read_one(H0(), std::integral_constant<int, I - DPW>(), next);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };
    // the first half of stage s+1 with the reads of its second half between the MFMAs
    const auto first_half = [&](int next) {
        static_for<NM>([&](auto ic) {
            mfma_one(H0(), ic);
            if constexpr (decltype(ic)::value < NRD) { read_one(H1(), ic, next); __builtin_amdgcn_sched_barrier(0); }
        });
    };
    static_for<DPW>([&](auto uc) { fetch_one(uc, 0); });
    if (nst > 1) static_for<DPW>([&](auto uc) { fetch_one(uc, 1); });
    if (nst > 2) static_for<DPW>([&](auto uc) { fetch_one(uc, 2); });
    // this wave's share of stage 0 has landed when only the fetches of stages 1 and 2 are outstanding
    if (nst > 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * DPW) : "memory");
    else if (nst > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#ifdef SCFGP_TRACE
    const unsigned long long tr_l0 = __builtin_amdgcn_s_memrealtime();
#endif
    static_for<NRD>([&](auto rc) { read_one(H0(), rc, 0); });
    static_for<NRD>([&](auto rc) { read_one(H1(), rc, 0); });
    SCFGP_WAIT_FRAGS("lgkmcnt(0)", 0);
    __builtin_amdgcn_sched_barrier(0);
    static_for<NM>([&](auto ic) { mfma_one(H0(), ic); });
    int slot = 0, s = 0;                                       // slot of stage s; the first half of stage s is multiplied
    for (; s + 3 < nst; ++s) {                                 // steady state: nothing to decide
        // this wave's share of stage s+1 has landed when only the fetches of stage s+2 are outstanding; its reads of stage s are done
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DPW) : "memory");
        SCFGP_WAIT_FRAGS("lgkmcnt(0)", 1);
        __builtin_amdgcn_sched_barrier(0);
        second_half_pre();
        __builtin_amdgcn_s_barrier();                          // everybody's has; nobody reads the slot of stage s any more
        asm volatile("" ::: "memory");
        const int fslot = slot;
        slot = slot == 2 ? 0 : slot + 1;
        __builtin_amdgcn_sched_barrier(0);
        second_half(std::true_type(), fslot, slot * D::STAGE);
        SCFGP_WAIT_FRAGS("lgkmcnt(0)", 0);
        __builtin_amdgcn_sched_barrier(0);
        first_half(slot * D::STAGE);
    }
    // (a register copy the allocator places where one loop hands over to the next must not read a fragment still in flight)
    SCFGP_WAIT_FRAGS("lgkmcnt(0)", 1);
    for (; s + 1 < nst; ++s) {                                 // the last stages: nothing left to fetch
        if (s + 2 < nst) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SCFGP_WAIT_FRAGS("lgkmcnt(0)", 1);
        __builtin_amdgcn_sched_barrier(0);
        second_half_pre();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        slot = slot == 2 ? 0 : slot + 1;
        __builtin_amdgcn_sched_barrier(0);
        second_half(std::false_type(), 0, slot * D::STAGE);
        SCFGP_WAIT_FRAGS("lgkmcnt(0)", 0);
        __builtin_amdgcn_sched_barrier(0);
        first_half(slot * D::STAGE);
    }
    SCFGP_WAIT_FRAGS("lgkmcnt(0)", 1);
    __builtin_amdgcn_sched_barrier(0);
    static_for<NM>([&](auto ic) { mfma_one(H1(), ic); });
#undef SCFGP_WAIT_FRAGS
    __syncthreads();
#ifdef SCFGP_TRACE
    const unsigned long long tr_l1 = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr int SLOTS = BN >= 128 ? BN / 128 : 1;            // vpart / mupart slots: one per 128 columns of the full tiles, one per remainder tile
    const int vslot = slot0 + SLOTS * jt;
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));                              // the epilogue's lane arithmetic stays behind the k loop (registers)
    apply_epilogue<Cfg, EPI, EPI == 0 || EPI == 3, true>(acc, Phi, V, vpart, p, q, y, alpha, ut, K, Kp, Np, rb, cbase, vslot, smem_raw, mu, tid);
    if (SLOTS == 2 && (EPI == 0 || EPI == 3) && tid < D::BM) {
        vpart[(int64_t)(vslot + 1) * Np + rb * D::BM + tid] = 0.0;
        if (mu) mu[(int64_t)(vslot + 1) * Np + rb * D::BM + tid] = 0.0;
    }
#ifdef SCFGP_TRACE
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x < ATRACE_CAP) {
        unsigned long long* rec = g_atrace[EPI * ATRACE_CAP + blockIdx.x];
        rec[0] = tr_t0; rec[1] = __builtin_amdgcn_s_memrealtime();
        rec[2] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xF) | ((unsigned long long)jt << 8) |
                                  ((unsigned long long)nst << 20);
        rec[3] = tr_l0; rec[4] = tr_l1;
    }
#endif
}

// Columns [0, K) of the output are covered by a cascade of launches of decreasing tile width
// (128-wide tiles, then 64-wide tiles for the ragged remainder; K = 2112: 16 x 128 + 1 x 64), so no workgroup multiplies a
// half-empty tile; columns >= K of the output buffer are never written and stay zero.
template <typename T> struct ApplyPlan {
    static constexpr int NW = 2;
    int width[NW], count[NW], col0[NW], jt0[NW], total;
    explicit ApplyPlan(int K) {
        const int w[NW] = {128, 64};
        int col = 0, jt = 0;
        // K <= 256: a handful of workgroups whatever the tiling, so ONE launch of 64-wide tiles (a launch costs more
        // than the half-empty tile there: Boston shape, K = 144, 80 -> 45 us per product)
        const bool small = K <= 256;
        for (int i = 0; i < NW; ++i) {
            width[i] = w[i];
            const bool last = i == NW - 1;
            count[i] = small && !last ? 0 : (last ? (K - col + w[i] - 1) / w[i] : (K - col) / w[i]);
            col0[i] = col; jt0[i] = jt;
            col += count[i] * w[i]; jt += count[i];
        }
        total = jt;
    }
};
template <class Cfg, int EPI, typename T>
static int apply_launch_cfg(const Geom& g, int njt, int col0, int jt0, const T* Phi, const T* Bm, T* V, double* vpart,
                            const double* p, const double* q, const double* y, const double* alpha, const double* ut,
                            double* mu, int ntot, hipStream_t st) {
    if (njt <= 0) return 0;
    const int64_t nrb = g.Np / Cfg::BM;
    const int wgs_per_rb = EPI == 3 || EPI == 4 ? (njt + 1) / 2 : njt;          // triangular products pair their column tiles
    allow_big_lds(apply_kernel<Cfg, EPI>, Cfg::LDS_BYTES);
    hipLaunchKernelGGL((apply_kernel<Cfg, EPI>), dim3((unsigned)(wgs_per_rb * nrb)), dim3(Cfg::THREADS), Cfg::LDS_BYTES, st,
                       Phi, Bm, V, vpart, p, q, y, alpha, ut, g.K, g.Kp, g.Np, njt, col0, jt0, mu, ntot);
    return (int)(njt * nrb);
}
template <typename T, int EPI, int BN>
static int apply_dma_launch(const Geom& g, int njt, int col0, int slot0, const T* Phi, const T* Bm, T* V, double* vpart,
                            const double* p, const double* q, const double* y, const double* alpha, const double* ut,
                            double* mu, hipStream_t st, int64_t rb0 = 0, int64_t nrb = -1) {
    typedef ApplyDma<T, BN> D;
    if (nrb < 0) nrb = g.Np / D::BM;                           // row blocks rb0 .. rb0 + nrb - 1 (default: all)
    if (njt <= 0 || nrb <= 0) return 0;
    allow_big_lds(apply_dma_kernel<T, EPI, BN>, D::LDS_BYTES);
    hipLaunchKernelGGL((apply_dma_kernel<T, EPI, BN>), dim3((unsigned)(njt * nrb)), dim3(64 * D::WAVES), D::LDS_BYTES, st,
                       Phi, Bm, V, vpart, p, q, y, alpha, ut, g.K, g.Kp, g.Np, njt, mu, col0, slot0, rb0);
    return (int)(njt * nrb);
}
static int apply_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}
// dma: 0 = every tile by the register-staged kernel; 1 / 2 = LDS-DMA tiles, the full 128-column blocks 128 wide / 256 wide (fp32
//      only; an odd 128-column block and fp64 stay 128 wide), the ragged 64-column remainder 64 wide
// BmT: the operand with its k-contiguous columns as rows (what the DMA-fed tiles read); NULL: Bm is symmetric
template <typename T, int EPI>
static int apply_launch(const Geom& g, const T* Phi, const T* Bm, T* V, double* vpart, const double* p, const double* q,
                        const double* y, const double* alpha, const double* ut, double* mu, hipStream_t st,
                        int dma = 0, const T* BmT = nullptr, const F16Operands* f16 = nullptr) {
    if (!BmT) BmT = Bm;
    const ApplyPlan<T> pl(g.K);
    int nb = 0;
    if constexpr (EPI != 2) {
        if (dma && pl.count[0] > 0) {
            int n256 = 0;
            const int64_t nrb = g.Np / 256;
            int64_t tail_rb = 0;
            if constexpr (sizeof(T) == 4) {
                n256 = dma == 2 ? pl.count[0] / 2 : 0;
                // The 256-wide tiles run one per CU, so a launch is whole rounds of num_cus tiles plus a last, partly filled one that
                // costs a full tile time.  When that last round would be less than 0.45 full, its row blocks go to 128-wide tiles
                // instead (at most one 8-wave workgroup per CU: ~0.55 of the tile time): H, 3907 row blocks x 8 = 122.09 rounds.
                if (n256 > 0 && EPI != 3 && EPI != 4) {
                    const int64_t ncu = apply_num_cus(), tiles = nrb * n256, rem = tiles % ncu;
                    if (tiles > ncu && rem > 0 && 20 * rem <= 9 * ncu) tail_rb = (rem + n256 - 1) / n256;
                }
                if (f16 && (EPI == 0 || EPI == 1)) {             // compute mode SCFGP_F16X3: the same launch plan on the three-term fp16 split
                    constexpr int E = EPI == 1 ? 1 : 0;
                    nb += F16x3Kernels::apply<E, 256>(g, n256, 0, 0, Phi, *f16, V, vpart, p, q, y, alpha, ut, mu, st, 0, nrb - tail_rb);
                    nb += F16x3Kernels::apply<E, 128>(g, pl.count[0] - 2 * n256, 256 * n256, 2 * n256, Phi, *f16, V, vpart, p, q, y, alpha, ut, mu, st, 0, nrb - tail_rb);
                    nb += F16x3Kernels::apply<E, 64>(g, pl.count[1], pl.col0[1], pl.jt0[1], Phi, *f16, V, vpart, p, q, y, alpha, ut, mu, st, 0, nrb - tail_rb);
                    if (tail_rb > 0) {
                        nb += F16x3Kernels::apply<E, 128>(g, pl.count[0], 0, 0, Phi, *f16, V, vpart, p, q, y, alpha, ut, mu, st, nrb - tail_rb, tail_rb);
                        nb += F16x3Kernels::apply<E, 64>(g, pl.count[1], pl.col0[1], pl.jt0[1], Phi, *f16, V, vpart, p, q, y, alpha, ut, mu, st, nrb - tail_rb, tail_rb);
                    }
                    return nb;
                }
                nb += apply_dma_launch<T, EPI, 256>(g, n256, 0, 0, Phi, BmT, V, vpart, p, q, y, alpha, ut, mu, st, 0, nrb - tail_rb);
            }
            nb += apply_dma_launch<T, EPI, 128>(g, pl.count[0] - 2 * n256, 256 * n256, 2 * n256, Phi, BmT, V, vpart, p, q, y, alpha, ut, mu, st, 0, nrb - tail_rb);
            // the ragged 64-column remainder: the same kernel with 256 x 64 tiles (4 / 8 waves, two workgroups per CU)
            nb += apply_dma_launch<T, EPI, 64>(g, pl.count[1], pl.col0[1], pl.jt0[1], Phi, BmT, V, vpart, p, q, y, alpha, ut, mu, st, 0, nrb - tail_rb);
            if (tail_rb > 0) {
                nb += apply_dma_launch<T, EPI, 128>(g, pl.count[0], 0, 0, Phi, BmT, V, vpart, p, q, y, alpha, ut, mu, st, nrb - tail_rb, tail_rb);
                nb += apply_dma_launch<T, EPI, 64>(g, pl.count[1], pl.col0[1], pl.jt0[1], Phi, BmT, V, vpart, p, q, y, alpha, ut, mu, st, nrb - tail_rb, tail_rb);
            }
            return nb;
        }
    }
    nb += apply_launch_cfg<typename ApplyCfg<T, 128>::type, EPI, T>(g, pl.count[0], pl.col0[0], pl.jt0[0], Phi, Bm, V, vpart, p, q, y, alpha, ut, mu, pl.total, st);
    nb += apply_launch_cfg<typename ApplyCfg<T, 64>::type, EPI, T>(g, pl.count[1], pl.col0[1], pl.jt0[1], Phi, Bm, V, vpart, p, q, y, alpha, ut, mu, pl.total, st);
    return nb;
}

template <typename T>
void ApplyKernels<T>::apply_v(const Geom& g, const T* Phi, const T* Bm, T* V, double* vpart, const double* alpha, double* mu,
                              hipStream_t st, int dma, const F16Operands* f16) {
    apply_launch<T, 0>(g, Phi, Bm, V, vpart, nullptr, nullptr, nullptr, alpha, nullptr, mu, st, dma, nullptr, f16);
}
template <typename T>
void ApplyKernels<T>::apply_predict(const Geom& g, const T* Phi, const T* LiT, double* vpart, const double* alpha, double* mu,
                                    hipStream_t st) {
    apply_launch<T, 2>(g, Phi, LiT, (T*)nullptr, vpart, nullptr, nullptr, nullptr, alpha, nullptr, mu, st);
}
// Out[n][c] = sum_{k < Kc} A[n][k] Bm[k][c] for c < ncols (rounded up to 64-wide tiles), everything with leading dimension Kp;
// Bm[k][c] = 0 for k < c is assumed (the contraction of a column tile starts at its first column)
template <typename T>
void ApplyKernels<T>::apply_plain(const Geom& g, const T* A, const T* Bm, T* Out, int Kc, int ncols, hipStream_t st) {
    Geom gk = g; gk.K = Kc;
    apply_launch_cfg<typename ApplyCfg<T, 64>::type, 4, T>(gk, (ncols + 63) / 64, 0, 0, A, Bm, Out, nullptr, nullptr, nullptr, nullptr, nullptr,
                                                           nullptr, nullptr, ApplyPlan<T>(Kc).total, st);
}

template <typename T>
void ApplyKernels<T>::apply_c(const Geom& g, const T* Phi, const T* LiT, const T* Li, T* C, double* vpart, const double* alpha,
                              const double* beta, double* mu, hipStream_t st, int dma) {
    // mu = Phi alpha: the loader-staged tiles form it from the Phi values they stage (alpha); the LDS-DMA tiles have no operand
    // values in registers and take it from their accumulators instead, mu = C beta (apply_epilogue, EPI 3)
    const bool by_dma = dma && ApplyPlan<T>(g.K).count[0] > 0;
    apply_launch<T, 3>(g, Phi, LiT, C, vpart, nullptr, nullptr, nullptr, by_dma ? beta : alpha, nullptr, mu, st, dma, Li);
}
template <typename T>
void ApplyKernels<T>::apply_vc(const Geom& g, const T* C, const T* Li, const T* LiT, T* V, hipStream_t st, int dma) {
    apply_launch<T, 4>(g, C, Li, V, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, st, dma, LiT);
}
template <typename T>
void ApplyKernels<T>::apply_phibar(const Geom& g, const T* Phi, const T* Abar, T* V, const double* p, const double* q,
                                   const double* y, const double* alpha, const double* ut, hipStream_t st, int dma, const F16Operands* f16) {
    apply_launch<T, 1>(g, Phi, Abar, V, nullptr, p, q, y, alpha, ut, nullptr, st, dma, nullptr, f16);
}

// number of column tiles of the apply kernel (vpart leading count)
template <typename T> static int apply_njt(const Geom& g) { return ApplyPlan<T>(g.K).total; }


// --------------------------------------------------------------------------
// per-row statistics from apply_v's per-tile by-products: one thread per row.
//   mu = sum_jt mupart, v = sum_jt vpart, d = kappa (v+1), r = mu - y
//   MODE 0 (train): p = 2r/d, e = 1/d - (r^2+v)/d^2, q = 1/d + kappa e; block partials of
//                   T2 = (r^2+v)/d + log(2 pi d), kbar = e (v+1) and the two row sums of bbar = sum_n Phibar_n . phi_n that
//                   are not K x K work: sum_n q_n v_n and sum_n p_n mu_n (kernels_kstage.hip: kstage_bbar)
//   MODE 1 (predict): mu, sd = sqrt(kappa (1+v))
// --------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void rowstats_kernel(const double* __restrict__ mupart, const double* __restrict__ vpart, int njt,
                                                       const double* __restrict__ y, const Scal* __restrict__ sc,
                                                       double* __restrict__ o1, double* __restrict__ o2,
                                                       double* __restrict__ partial, int64_t N, int64_t Np) {
    __shared__ double red[4][256];
    const double kappa = sc->kappa;
    double t2 = 0, kb = 0, qv = 0, pm = 0;
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
                qv += qn * v; pm += pn * mu;
            }
            o1[n] = pn; o2[n] = qn;
        } else if (n < N) {
            o1[n] = mu; o2[n] = sqrt(d);
        }
    }
    if (MODE == 0) {
        red[0][threadIdx.x] = t2; red[1][threadIdx.x] = kb; red[2][threadIdx.x] = qv; red[3][threadIdx.x] = pm;
        __syncthreads();
        for (int w = 128; w >= 1; w >>= 1) {
            if ((int)threadIdx.x < w)
                for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + w];
            __syncthreads();
        }
        if (threadIdx.x < 4) partial[blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
    }
}

template <typename T>
void ApplyKernels<T>::rowstats(const Geom& g, const double* mupart, const double* vpart, const double* y,
                               const Scal* sc, double* p, double* q, double* partial, int nblocks, hipStream_t st) {
    hipLaunchKernelGGL((rowstats_kernel<0>), dim3(nblocks), dim3(256), 0, st, mupart, vpart, apply_njt<T>(g), y, sc, p, q,
                       partial, g.N, g.Np);
}

template <typename T>
void ApplyKernels<T>::rowpredict(const Geom& g, const double* mupart, const double* vpart, const Scal* sc, double* mu, double* sd,
                                 hipStream_t st) {
    const int nblocks = (int)((g.Np + 255) / 256 < 1024 ? (g.Np + 255) / 256 : 1024);
    hipLaunchKernelGGL((rowstats_kernel<1>), dim3(nblocks), dim3(256), 0, st, mupart, vpart, apply_njt<T>(g),
                       (const double*)nullptr, sc, mu, sd, (double*)nullptr, g.N, g.Np);
}
template struct ApplyKernels<double>;
template struct ApplyKernels<float>;

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
