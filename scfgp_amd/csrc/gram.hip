// TN products of the SCFGP objective (contraction over the rows):  Phi^T diag(w) Phi with its side vector Phi^T s
// (SCFGP/SCFGP.py:104,108 and the backward of :111-113) and X~^T Zbar (the reverse sweep of :98-102), plus the
// reductions of their per-split fp64 slabs.  Built on tile_engine.h; templated on the compute type T (double | float).
#include "kernels.h"
#include "tile_cfgs.h"

#include <algorithm>

// Diagnostic build (-DSCFGP_TRACE, library variant "_trace"): every workgroup of the Gram kernel records
// [start, end] on the 100 MHz constant clock, its XCC id and its job kind, so the tail and the spread of job lengths
// can be read off (tests/gpu_gram_trace.py).  No stamp reaches any output; the product build contains none of this.
#ifdef SCFGP_TRACE
constexpr int TRACE_CAP = 1 << 16;
__device__ unsigned long long g_trace[TRACE_CAP][4];
#define TRACE_BEGIN() const unsigned long long tr_t0 = __builtin_amdgcn_s_memrealtime()   /* TRACE_END reads the job id `j` */
#define TRACE_END(kind)                                                                                     \
    do {                                                                                                    \
        __syncthreads();                                                                                    \
        if (threadIdx.x == 0 && j < TRACE_CAP) {                                                            \
            g_trace[j][0] = tr_t0; g_trace[j][1] = __builtin_amdgcn_s_memrealtime();                        \
            g_trace[j][2] = (unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) |             \
                            ((unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) << 8);       \
            g_trace[j][3] = (unsigned long long)(kind);                                                     \
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

// Gram tiles: 256 x 128 (fp32 only) and 128 x 128 output tiles plus a 64 x 128 strip (tile_cfgs.h)
template <typename T, int TILE> struct GramCfg {
    typedef TileCfg<T, TILE, TILE, Tune<T>::GRAM_BK, Tune<T>::GRAM_WGM, Tune<T>::GRAM_WGN, Tune<T>::MS> type;
};
// the 64-high strip below the square tiles: same workgroup size (one launch), 32 x 32 wave tiles
template <typename T> struct GramStripCfg {
    typedef TileCfg<T, 64, 128, Tune<T>::GRAM_BK, Tune<T>::GRAM_WGM / 2, Tune<T>::GRAM_WGN * 2, Tune<T>::MS> type;
};
// fp32 only: the tall Gram tile (64 x 64 wave tiles); fp64 would need 16 waves for the same tile
template <typename T> struct GramBigCfg {
    typedef TileCfg<T, 256, 128, SCFGP_BK, Tune<T>::GRAM_WGM, Tune<T>::GRAM_WGN, Tune<T>::MS> type;
};
template <typename T> struct XtzCfg { typedef TileCfg<T, 128, 128, 16, 4, 2, Tune<T>::MS> type; };
// row tiles of X~^T Zbar that hold at most 96 / 64 live rows of X~^T (D + 1 = 65 at the headline shape): same 128-wide
// slabs, fewer MFMA rows
template <typename T> struct Xtz96Cfg { typedef TileCfg<T, 96, 128, 16, 2, 4, 16> type; };     // 48-row wave tiles: 16 x 16 MFMA shape only
template <typename T> struct Xtz64Cfg { typedef TileCfg<T, 64, 128, 16, 2, 4, Tune<T>::MS> type; };

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
                                           double* slab_hi = nullptr, int tid = (int)threadIdx.x) {
    AccCoord<Cfg> co(tid);
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
// DIAG is a compile-time property of the instantiation (the side sums cost a dozen registers that only the diagonal jobs need:
// with them in every job the 64 x 64 wave tiles spilled inside the k-loop); gram_body dispatches on the job's flag
template <class Cfg, bool WEIGHT, bool STRIP, bool DIAG>
__device__ __forceinline__ void gram_body_impl(
    const typename Cfg::T* __restrict__ Phi, int64_t ld, const double* __restrict__ w, const double* __restrict__ side,
    int64_t r0, int64_t r1, int64_t chunk, int acol, int bcol, double* __restrict__ sideout,
    double* __restrict__ slab, double* __restrict__ slab_hi, char* smem_raw) {
    typedef typename Cfg::T T;
    T* smem = reinterpret_cast<T*>(smem_raw);
    typename Cfg::MTr::acc_t acc[Cfg::TM][Cfg::TN];
    // the thread id behind an opaque move: inside the persistent launch the compiler would otherwise hoist every lane-dependent
    // address of every tile shape out of the job loop and keep them all in registers (spills in the k-loops)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    // consecutive chunks are consecutive k-tiles, so one pair of loaders walks the whole row range
    NatLoader<T, T, Cfg::BM, Cfg::BK, Cfg::LDA, Cfg::THREADS, WEIGHT, false, DIAG> la(
        Phi + r0 * ld + acol, ld, tid, WEIGHT ? w + r0 : nullptr, 0, DIAG ? side + r0 : nullptr);
    NatLoader<T, T, Cfg::BN, Cfg::BK, Cfg::LDB, Cfg::THREADS, false, false> lb(Phi + r0 * ld + bcol, ld, tid);
    bool first = true;
    for (int64_t c0 = r0; c0 < r1 || first; c0 += chunk) {
        const int64_t c1 = c0 + chunk < r1 ? c0 + chunk : r1;
        acc_zero<Cfg>(acc);
        if (c0 < r1) tile_mainloop<Cfg>(la, lb, (int)((c1 - c0) / Cfg::BK), acc, smem, tid);
        slab_flush<Cfg>(acc, slab, first, slab_hi, tid);
        if constexpr (DIAG) la.side_flush();
        first = false;
    }
    if constexpr (STRIP) {                                     // lower half of the 128 x 128 slab(s): rows the strip does not have
        constexpr int NB = Cfg::BN / 128, REST = (128 - Cfg::BM) * 128;
        for (int e = threadIdx.x; e < NB * REST; e += Cfg::THREADS) slab[(e / REST) * (128 * 128) + Cfg::BM * 128 + e % REST] = 0.0;
    }
    if constexpr (DIAG) la.side_reduce(reinterpret_cast<double*>(smem_raw), sideout);
}
template <class Cfg, bool WEIGHT, bool STRIP>
__device__ __forceinline__ void gram_body(
    const typename Cfg::T* __restrict__ Phi, int64_t ld, const double* __restrict__ w, const double* __restrict__ side,
    int64_t r0, int64_t r1, int64_t chunk, int acol, int bcol, bool diag, double* __restrict__ sideout,
    double* __restrict__ slab, double* __restrict__ slab_hi, char* smem_raw) {
    if (diag) gram_body_impl<Cfg, WEIGHT, STRIP, true>(Phi, ld, w, side, r0, r1, chunk, acol, bcol, sideout, slab, slab_hi, smem_raw);
    else gram_body_impl<Cfg, WEIGHT, STRIP, false>(Phi, ld, w, side, r0, r1, chunk, acol, bcol, sideout, slab, slab_hi, smem_raw);
}

// fp32 tall tile (256 x 128, 8 waves of 64 x 64) with LDS-DMA staging: both operand panels go global -> LDS by
// global_load_lds_dwordx4 into a ring of three stages of 16 rows, counted vmcnt, one raw barrier per stage -- no staging registers,
// no ds_write, the fetches of the next stages in flight while stage s is multiplied (the structure of apply.hip's apply_dma_kernel).
//   LDS image of a stage: the 16 rows of the A panel (256 floats = 1 KiB each: ONE DMA instruction per row), then the 16 rows of
//   the B panel (128 floats: one instruction per two rows), k-major and unpadded: a lane of MFMA tile column i reads the FOUR
//   adjacent floats 4 i .. 4 i + 3 of its k row with one ds_read_b128 -- MFMA tile tm, tile row rho IS output row 4 rho + tm of
//   the wave tile (columns likewise), so a fragment of four tiles is one read, the 16 lanes of a k row cover one 256-byte bank
//   row, and the four lane groups a ds_read_b128 is served in ({0-3,12-15 | 20-27}, ...: two k rows each) touch disjoint slots
//   because every row starts on a bank-row boundary.  A lane ends up with a 4 x 4 block of the output: 32-byte slab updates.
//   Row weights (WEIGHT: q_n) and the side vector's multipliers (DIAG: y_n or p_n) of the stage's 16 rows ride in the ring as
//   one more DMA instruction (wave 0) from an array of (weight, multiplier) float pairs packed before the launch
//   (gram_pack_ws); the weight multiplies the A fragment after the LDS read, and the side sums sum_k s_k Phi[k][acol + m] are
//   formed from the same (unweighted) fragments, fp32 over a chunk of rows and fp64 across chunks.
struct GramDma {
    static constexpr int BM = 256, BN = 128, A_BYTES = 16 * BM * 4, B_BYTES = 16 * BN * 4, W_OFF = A_BYTES + B_BYTES,   // W_OFF: the 16 (weight, multiplier) pairs
                         STAGE = W_OFF + 256, STAGES = 3, LDS_BYTES = STAGES * STAGE, DMA_PER_WAVE = (A_BYTES + B_BYTES) / 1024 / 8;
    static_assert(DMA_PER_WAVE == 3, "8 waves, 24 KiB of operands per stage");
};
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
// generic address of an LDS byte -> LDS pointer: the low 32 bits are the LDS offset (the address-space cast proper carries a null
// check that this compiler mis-selects in one of the kernels below: "V_CMP_NE_U32 0, $src_shared_base")
__device__ __forceinline__ lds_void* lds_ptr(const char* p) { return (lds_void*)(uintptr_t)((unsigned)(uintptr_t)p); }
// The k loop is software-pipelined across the stage barrier like apply.hip's apply_dma_kernel (see there): a stage's four
// k-steps are two HALVES (k-steps 0,1 and 2,3) with their own fragment registers; the barrier that opens stage s+1 stands between
// the MFMAs of the first and the second half of stage s, the fetch of stage s+3 and the reads of the next first half ride between
// the second half's MFMAs, the reads of the next second half between the first half's.  LDS reads are inline assembly with
// hand-counted waits that name the registers they release (tools/isa_inflight.py checks the compiled stream).
// Three tile shapes share the loop (8 waves; diagonal jobs add the side sums, DIAG):
//   GramDma      256 x 128 (wave grid 4 x 2 of 64 x 64), ring of three 24 KiB stages
//   GramDmaWide   64 x 512 (wave grid 1 x 8 of 64 x 64: four strip tiles side by side), ring of TWO 36 KiB stages (two workgroups per CU
//                 share the LDS): the fetch of stage s+2 goes into the slot the barrier of stage s+1 frees and has one stage
//                 to land.  Its 36 operand instructions per stage do not divide by 8 waves: waves 0-3 issue five, 4-7 four
//                 (every vmcnt wait of the two-slot ring is vmcnt(0), so nothing counts them).
struct GramDmaWide {
    static constexpr int BM = 64, BN = 512, A_BYTES = 16 * BM * 4, B_BYTES = 16 * BN * 4, W_OFF = A_BYTES + B_BYTES,   // W_OFF: the 16 (weight, multiplier) pairs
                         STAGE = W_OFF + 256, STAGES = 2, LDS_BYTES = STAGES * STAGE, NINSTR = (A_BYTES + B_BYTES) / 1024, DMA_PER_WAVE = 5;
    static_assert(NINSTR == 36 && 2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
};
//   GramDmaSq    128 x 128 (wave grid 4 x 2 of 32 x 64: two MFMA tiles along the rows; the diagonal blocks the tall tiles leave),
//                 ring of three 16 KiB stages
struct GramDmaSq {
    static constexpr int BM = 128, BN = 128, A_BYTES = 16 * BM * 4, B_BYTES = 16 * BN * 4, W_OFF = A_BYTES + B_BYTES,   // W_OFF: the 16 (weight, multiplier) pairs
                         STAGE = W_OFF + 256, STAGES = 3, LDS_BYTES = STAGES * STAGE, DMA_PER_WAVE = (A_BYTES + B_BYTES) / 1024 / 8;
    static_assert(DMA_PER_WAVE == 2, "8 waves, 16 KiB of operands per stage");
};
//   GramDmaPair  TWO diagonal 128 x 128 blocks side by side as one 256-row tile (wave grid 4 x 2 of 64 x 64: four waves per block):
//                 a diagonal block's B panel IS its A panel, so the stage holds only the 256-column A image (columns of block 1,
//                 then of block 2) and the B fragments are read from it; 64 MFMAs per wave and barrier like the tall tile
struct GramDmaPair {
    static constexpr int BM = 256, BN = 128, A_BYTES = 16 * BM * 4, B_BYTES = 0, W_OFF = A_BYTES,   // W_OFF: the 16 (weight, multiplier) pairs
                         STAGE = W_OFF + 256, STAGES = 3, LDS_BYTES = STAGES * STAGE, DMA_PER_WAVE = A_BYTES / 1024 / 8;
    static_assert(DMA_PER_WAVE == 2, "8 waves, 16 KiB of operands per stage");
};
//   TRANS (tall tiles only): the accumulators are flushed TRANSPOSED -- output row = B-panel column, output column = A-panel column,
//   A-panel columns 0..127 into `slab`, 128..255 into `slab_hi` -- which lets a 256 x 128 tile compute two blocks of a block ROW:
//   the unpaired last block row of an odd block count (K = 4224: 32 of its 33 tiles) as 16 tall tiles with A = two column blocks
//   and B = the row's own block, instead of 32 square tiles at 0.73 of the matrix peak (profiles/r05_gram_trace_C5.txt).  A lane's
//   accumulators (tm = 0..3, tn, r) are four consecutive A-panel columns of one B-panel column: still one 32-byte slab update.
template <class D, bool WEIGHT, bool DIAG, bool TRANS = false>
__device__ __forceinline__ void gram_pipe_dma(
    const float* __restrict__ Phi, int64_t ld, const float* __restrict__ ws2,
    int64_t r0, int64_t r1, int64_t chunk, int acol, int bcol, double* __restrict__ sideout,
    double* __restrict__ slab, double* __restrict__ slab_hi, char* smem) {
    constexpr bool WIDE = D::BM == 64, SQ = D::BM == 128, PAIR = D::B_BYTES == 0;      // PAIR: acol / bcol are the two blocks' columns
    constexpr int RING = D::STAGES, TM = SQ ? 2 : 4;            // MFMA tiles of a wave along the rows (four along the columns)
    typedef std::integral_constant<int, 0> H0;
    typedef std::integral_constant<int, 1> H1;
    constexpr bool WS = WEIGHT || DIAG;
    constexpr int DPW = D::DMA_PER_WAVE, NRD = 4 + (WEIGHT || DIAG ? 2 : 0), NM = 8 * TM, PRE = 2;          // per half: reads, MFMAs
    static_assert(PRE + DPW + 1 + NRD <= NM, "one fetch or read per MFMA behind the barrier");
    static_assert(!(PAIR && DIAG), "the paired diagonal blocks carry no side sums");
    static_assert(!TRANS || (D::BM == 256 && D::B_BYTES != 0 && !DIAG), "transposed flush: off-diagonal tall tiles");
    // the thread id behind an opaque move (see gram_body_impl)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), i = lane & 15, q = lane >> 4;
    const int wm0 = WIDE ? 0 : (wave >> 1) * (16 * TM), wn0 = WIDE ? wave * 64 : (wave & 1) * 64;
    // DMA instruction t of a stage (1 KiB of the stacked A | B image each)
    //   tall: t = 3 wave + u; t < 16: row t of the A panel; else rows 2 (t - 16), 2 (t - 16) + 1 of the B panel
    //   wide: t = wave + 8 u (< 36); t < 4: rows 4t .. 4t+3 of the A panel (256 bytes each); else half (t - 4) & 1 of row (t - 4) / 2 of the B panel
    //   pair: t = 2 wave + u: row t of the A image, lanes 0-31 from block 1's columns, lanes 32-63 from block 2's
    //   square: t = 2 wave + u; t < 8: rows 2t, 2t+1 of the A panel, the odd row ROTATED by 32 floats (position p holds column
    //           p - 32 mod 128: the 8-byte A reads of an even and an odd k row then fall into different halves of the 256-byte bank
    //           row); else rows 2 (t - 8), 2 (t - 8) + 1 of the B panel
    const char* src[DPW]; int dst[DPW];
#pragma unroll
    for (int u = 0; u < DPW; ++u) {
        const int t = WIDE ? wave + 8 * u : DPW * wave + u;
        const float* g;
        if constexpr (PAIR) g = Phi + (r0 + t) * ld + (lane < 32 ? acol : bcol - 128) + 4 * lane;
        else if constexpr (SQ) g = t < 8 ? Phi + (r0 + 2 * t + (lane >> 5)) * ld + acol + ((4 * (lane & 31) - 32 * (lane >> 5)) & 127)
                                    : Phi + (r0 + 2 * (t - 8) + (lane >> 5)) * ld + bcol + 4 * (lane & 31);
        else if constexpr (WIDE) g = t < 4 ? Phi + (r0 + 4 * t + (lane >> 4)) * ld + acol + 4 * (lane & 15) : Phi + (r0 + ((t - 4) >> 1)) * ld + bcol + 256 * ((t - 4) & 1) + 4 * lane;
        else g = t < 16 ? Phi + (r0 + t) * ld + acol + 4 * lane : Phi + (r0 + 2 * (t - 16) + (lane >> 5)) * ld + bcol + 4 * (lane & 31);
        src[u] = reinterpret_cast<const char*>(g);
        dst[u] = t * 1024;
    }
    const bool last_u = !WIDE || wave < 4;                      // wide: instruction u = 4 exists on waves 0-3 only
    const int64_t step = 16 * ld * (int64_t)sizeof(float);
    // ws2[n] = (weight of row n or 1, side multiplier of row n or 0) as two floats (gram_pack_ws): a stage's 16 pairs are 128
    // bytes, fetched by lanes 0..31 of one more instruction on wave 0 (lanes 32..63 bring the next stage's pairs to the 128
    // bytes behind them, where nobody looks; the array is padded by that much)
    const char* wsrc = WS ? reinterpret_cast<const char*>(ws2 + 2 * r0) + 4 * lane : nullptr;
    const bool ws_wave = WS && wave == 0;
    const auto fetch_one = [&](auto uc, int slot) {            // u < DPW: operand instruction u; u == DPW: the weights / multipliers
        constexpr int u = decltype(uc)::value;
        char* base = smem + slot * D::STAGE;
        if constexpr (u < DPW) {
            if (u < DPW - 1 || last_u) {
                __builtin_amdgcn_global_load_lds((gbl_void*)src[u], lds_ptr(base + dst[u]), 16, 0, 0);
                src[u] += step;
            }
        } else if (ws_wave) {
            __builtin_amdgcn_global_load_lds((gbl_void*)wsrc, lds_ptr(base + D::W_OFF), 4, 0, 0);
            wsrc += 128;
        }
    };
    // LDS byte addresses (first stage) of this lane's fragment of k row q: 4 adjacent floats of the A / B panel, its weight
    const int ring = (int)(uintptr_t)smem;
    const int pa0 = SQ ? ring + q * (D::BM * 4) + ((wm0 + 2 * i + 32 * (q & 1)) & 127) * 4 : ring + q * (D::BM * 4) + (wm0 + 4 * i) * 4, pb0 = PAIR ? ring + q * (D::BM * 4) + ((wm0 & 128) + wn0 + 4 * i) * 4 : ring + D::A_BYTES + q * (D::BN * 4) + (wn0 + 4 * i) * 4, pw0 = ring + D::W_OFF + q * 8;
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef float afrag_t __attribute__((ext_vector_type(TM)));   // TM adjacent floats of the A panel's k row: MFMA tile tm, tile row rho = output row TM rho + tm
    afrag_t fa[2][2]; v4f fb[2][2];                            // [half][k-step of the half]
    v2f fws[2][2];                                             // (weight, side multiplier) of the lane's k row
    v4f acc[TM][4];
    afrag_t sacc = 0.f;                                        // side sums of a chunk: fp32 chains a quarter as long as the MFMAs'
    // read R of half h of the stage at byte `stage` of the ring: A, A, B, B, (pair, pair)
    const auto read_one = [&](auto hc, auto rc, int stage) {
        constexpr int h = decltype(hc)::value, R = decltype(rc)::value, kk = 2 * h + (R & 1);
        (void)&fws; (void)&pw0; (void)&fa; (void)&fb; (void)&pa0; (void)&pb0;      // (named outside the discarded branches: the capture is decided here)
        if constexpr (R < 2 && SQ) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(fa[h][R & 1]) : "v"(stage + pa0), "n"(kk * 4 * D::BM * 4));
        else if constexpr (R < 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[h][R & 1]) : "v"(stage + pa0), "n"(kk * 4 * D::BM * 4));
        else if constexpr (R < 4) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[h][R & 1]) : "v"(stage + pb0), "n"(kk * 4 * (PAIR ? D::BM : D::BN) * 4));
        else asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(fws[h][R & 1]) : "v"(stage + pw0), "n"(kk * 32));
    };
    // "s_waitcnt <what>" that hands out the fragments of half h, then the row weights / side sums on them (VALU).  The fragments
    // are in-out operands ("+v", as in apply.hip): their values are formally PRODUCED by the wait, so no use, copy or spill of
    // them can be ordered before it at the IR level either
#define SCFGP_WAIT_FRAGS(what, h)                                                                                                        \
    do {                                                                                                                                 \
        if constexpr (WS)                                                                                                                \
            asm volatile("s_waitcnt " what : "+v"(fa[h][0]), "+v"(fa[h][1]), "+v"(fb[h][0]), "+v"(fb[h][1]), "+v"(fws[h][0]), "+v"(fws[h][1]) :: "memory"); \
        else asm volatile("s_waitcnt " what : "+v"(fa[h][0]), "+v"(fa[h][1]), "+v"(fb[h][0]), "+v"(fb[h][1]) :: "memory");              \
        __builtin_amdgcn_sched_barrier(0);                                                                                               \
        if constexpr (DIAG) { sacc += fws[h][0][1] * fa[h][0]; sacc += fws[h][1][1] * fa[h][1]; }       /* from the unweighted fragments */ \
        if constexpr (WEIGHT) { fa[h][0] *= fws[h][0][0]; fa[h][1] *= fws[h][1][0]; }                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                                               \
    } while (0)
    // MFMA I of half h: k-step 2h + I / (4 TM) of accumulator tile (tm, tn) = (I / 4 % TM, I % 4)
    const auto mfma_one = [&](auto hc, auto ic) {
        constexpr int h = decltype(hc)::value, I = decltype(ic)::value, k2 = I / (4 * TM), tm = I / 4 % TM, tn = I % 4;
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[h][k2][tm], fb[h][k2][tn], acc[tm][tn], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    const auto second_half_pre = [&]() { static_for<PRE>([&](auto ic) { mfma_one(H1(), ic); }); };
    // the rest of the second half of stage s with the fetch of stage s+3 (FETCH; slot fslot) and the reads of the first half of
    // stage s+1 (READ; at byte `next` of the ring) between its MFMAs
    const auto second_half = [&](auto fc, auto rdc, int fslot, int next) {
        static_for<NM - PRE>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            mfma_one(H1(), std::integral_constant<int, I + PRE>());
            if constexpr (I < DPW + 1) {
                if constexpr (decltype(fc)::value && (I < DPW || WS)) { fetch_one(ic, fslot); __builtin_amdgcn_sched_barrier(0); }
            } else if constexpr (I - DPW - 1 < NRD) {
                if constexpr (decltype(rdc)::value) { read_one(H0(), std::integral_constant<int, I - DPW - 1>(), next); __builtin_amdgcn_sched_barrier(0); }
            }
        });
    };
    const auto first_half = [&](int next) {
        static_for<NM>([&](auto ic) {
            mfma_one(H0(), ic);
            if constexpr (decltype(ic)::value < NRD) { read_one(H1(), ic, next); __builtin_amdgcn_sched_barrier(0); }
        });
    };
    // accumulator (tm, tn, r) of lane (i, q) is output row wm0 + TM (4 q + r) + tm, column wn0 + 4 i + tn; slabs are 128 x 128:
    // tall: rows >= 128 in slab_hi; wide: four consecutive slabs, one per 128 columns, rows 0 .. 63 of each
    double* sl = TRANS ? (wm0 >= 128 ? slab_hi : slab) + (int64_t)wn0 * 128 + (wm0 & 127)
               : WIDE  ? slab + (int64_t)(wn0 >> 7) * (128 * 128) + (wn0 & 127)
                       : (wm0 >= 128 ? slab_hi + (int64_t)(wm0 - 128) * 128 : slab + (int64_t)wm0 * 128) + wn0;
    bool first = true;
    const auto flush = [&]() {
        int lf = lane;                                         // (opaque: the 16 slab addresses are formed here, not kept across the k loop)
        asm volatile("" : "+v"(lf));
        const int i = lf & 15, q = lf >> 4;
        if constexpr (TRANS) {                                 // slab row 4 i + tn (B-panel column), columns 4 (4 q + r) .. + 3 (A-panel columns)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v4d* d = reinterpret_cast<v4d*>(sl + (4 * i + tn) * 128 + 4 * (4 * q + r));
                    const v4d v = v4d{(double)acc[0][tn][r], (double)acc[1][tn][r], (double)acc[2][tn][r], (double)acc[3][tn][r]};
                    *d = first ? v : *d + v;
                }
        } else {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v4d* d = reinterpret_cast<v4d*>(sl + (TM * (4 * q + r) + tm) * 128 + 4 * i);
                const v4d v = v4d{(double)acc[tm][0][r], (double)acc[tm][1][r], (double)acc[tm][2][r], (double)acc[tm][3][r]};
                *d = first ? v : *d + v;
            }
        }
        if constexpr (DIAG) {                                  // the 4 k rows of a k-step live in the 4 lane groups: sum over q, fp64 across chunks
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                double x = (double)sacc[tm];
                x += __shfl_xor(x, 16); x += __shfl_xor(x, 32);
                if (q == 0 && wn0 == 0) { double* d = sideout + wm0 + TM * i + tm; *d = first ? x : *d + x; }
            }
            sacc = 0.f;
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = v4f{0.f, 0.f, 0.f, 0.f};
        first = false;
    };
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = v4f{0.f, 0.f, 0.f, 0.f};
    const auto strip_rest = [&]() {                             // wide: rows 64 .. 127 of the four slabs, which the strip does not have
        if constexpr (WIDE)
            for (int e = tid; e < 4 * 64 * 128; e += 512) slab[(e / (64 * 128)) * (128 * 128) + 64 * 128 + e % (64 * 128)] = 0.0;
    };
    const int nst = (int)((r1 - r0) / 16), per_chunk = (int)(chunk / 16);
    if (nst == 0) { flush(); strip_rest(); return; }            // an empty split still owns its slab
    // the fetch of stage s + RING - 1 is issued behind the barrier that opens stage s; before the loop: stages 0 .. RING - 2
    static_for<DPW + 1>([&](auto uc) { fetch_one(uc, 0); });
    if (RING == 3 && nst > 1) static_for<DPW + 1>([&](auto uc) { fetch_one(uc, 1); });
    const auto wait_landed = [&](bool more) {                   // this wave's share of a stage; `more`: a later stage's fetch may stay outstanding
        if (RING == 3 && more) {                                // one later stage in flight: DPW instructions, one more on the wave that fetches the pairs
            if (ws_wave) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DPW + 1) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DPW) : "memory");
        }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    const auto next_slot = [&](int sl_) { return sl_ == RING - 1 ? 0 : sl_ + 1; };
    int slot = 0, s = 0;                                        // slot of stage s
    // A chunk starts with an empty pipeline and drains it before its flush: no fragment is live (or in flight) across the flush,
    // whose register appetite would otherwise have the allocator spill fragment registers whose data has not arrived yet.
    while (s < nst) {
        const int cend = s + per_chunk < nst ? s + per_chunk : nst;
        // prime: stage s has landed when only the fetches of stage s+1 are outstanding
        wait_landed(s + 1 < nst);
        __builtin_amdgcn_s_barrier();                          // everybody's has; nobody reads the slot of stage s-1 any more
        asm volatile("" ::: "memory");
        if (s + RING - 1 < nst) { const int fslot = slot == 0 ? RING - 1 : slot - 1; static_for<DPW + 1>([&](auto uc) { fetch_one(uc, fslot); }); }
        static_for<NRD>([&](auto rc) { read_one(H0(), rc, slot * D::STAGE); });
        static_for<NRD>([&](auto rc) { read_one(H1(), rc, slot * D::STAGE); });
        SCFGP_WAIT_FRAGS("lgkmcnt(0)", 0);
        static_for<NM>([&](auto ic) { mfma_one(H0(), ic); });
        for (; s + 1 < cend && s + RING < nst; ++s) {          // steady state: the first half of stage s is multiplied
            // this wave's share of stage s+1 has landed when only the fetches of stage s+2 are outstanding; its reads of stage s are done
            // (the two counts apart from the statement that releases the fragments: one such statement per point of the loop, or
            // the allocator joins the branches with copies of registers still in flight)
            wait_landed(true);
            SCFGP_WAIT_FRAGS("lgkmcnt(0)", 1);
            second_half_pre();
            __builtin_amdgcn_s_barrier();                      // everybody's has; nobody reads the slot of stage s any more
            asm volatile("" ::: "memory");
            const int fslot = slot;
            slot = next_slot(slot);
            __builtin_amdgcn_sched_barrier(0);
            second_half(std::true_type(), std::true_type(), fslot, slot * D::STAGE);
            SCFGP_WAIT_FRAGS("lgkmcnt(0)", 0);
            first_half(slot * D::STAGE);
        }
        for (; s + 1 < cend; ++s) {                            // the last stages of the row range: nothing left to fetch
            wait_landed(s + 2 < nst);
            SCFGP_WAIT_FRAGS("lgkmcnt(0)", 1);
            second_half_pre();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            slot = next_slot(slot);
            __builtin_amdgcn_sched_barrier(0);
            second_half(std::false_type(), std::true_type(), 0, slot * D::STAGE);
            SCFGP_WAIT_FRAGS("lgkmcnt(0)", 0);
            first_half(slot * D::STAGE);
        }
        SCFGP_WAIT_FRAGS("lgkmcnt(0)", 1);                      // the last stage of the chunk: its second half, nothing issued for the next
        static_for<NM>([&](auto ic) { mfma_one(H1(), ic); });
        flush();
        ++s;
        slot = next_slot(slot);
    }
    strip_rest();
#undef SCFGP_WAIT_FRAGS
}

// fp64 square tile (128 x 128, 8 waves of 32 x 64) with LDS-DMA staging: the same structure as the fp32 tall tile with 16-row
// stages of 32 KiB (A panel 16 x 128 doubles, B panel likewise, one DMA instruction per row) in a ring of TWO (two workgroups per
// CU share the 160 KiB): the fetch of stage s+1 is issued behind the barrier of stage s and lands while stage s is multiplied.
//   A fragment: the two adjacent doubles 2 i, 2 i + 1 of the k row (MFMA tile tm, tile row rho = output row 2 rho + tm);
//   B fragment: doubles 2 i, 2 i + 1 and 32 + 2 i, 32 + 2 i + 1 (MFMA tile tn, tile column i = output column
//   32 (tn >> 1) + 2 i + (tn & 1)): every read is one ds_read_b128 whose 16 lanes cover one 256-byte bank row.
struct GramDma64 {
    static constexpr int B = 128, ROWB = B * 8, A_BYTES = 16 * ROWB, W_OFF = 2 * A_BYTES, S_OFF = W_OFF + 128, STAGE = W_OFF + 256,
                         STAGES = 2, LDS_BYTES = STAGES * STAGE, DMA_PER_WAVE = 2 * A_BYTES / 1024 / 8;
    static_assert(DMA_PER_WAVE == 4, "8 waves, 32 KiB of operands per stage");
};
template <bool WEIGHT, bool DIAG>
__device__ __forceinline__ void gram_sq_dma64(
    const double* __restrict__ Phi, int64_t ld, const double* __restrict__ w, const double* __restrict__ side,
    int64_t r0, int64_t r1, int acol, int bcol, double* __restrict__ sideout, double* __restrict__ slab, char* smem) {
    typedef GramDma64 D;
    constexpr bool WS = WEIGHT || DIAG;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                              // opaque thread id (see gram_body_impl)
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), i = lane & 15, q = lane >> 4;
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 64;
    // DMA instruction t = 4 wave + u of a stage: t < 16: row t of the A panel, else row t - 16 of the B panel; a lane carries 16 bytes
    const char* src[D::DMA_PER_WAVE]; int dst[D::DMA_PER_WAVE];
#pragma unroll
    for (int u = 0; u < D::DMA_PER_WAVE; ++u) {
        const int t = D::DMA_PER_WAVE * wave + u;
        const double* g = Phi + (r0 + (t & 15)) * ld + (t < 16 ? acol : bcol) + 2 * lane;
        src[u] = reinterpret_cast<const char*>(g);
        dst[u] = t * 1024;
    }
    const int64_t step = 16 * ld * (int64_t)sizeof(double);
    const double* ws_lo = WEIGHT ? w : side; const double* ws_hi = DIAG ? side : w;
    const char* wsrc = WS ? reinterpret_cast<const char*>((lane < 32 ? ws_lo : ws_hi) + r0) + 4 * (lane & 31) : nullptr;
    const bool ws_wave = WS && wave == 0;
    const auto issue = [&](int slot) {
        char* base = smem + slot * D::STAGE;
#pragma unroll
        for (int u = 0; u < D::DMA_PER_WAVE; ++u) {
            __builtin_amdgcn_global_load_lds((gbl_void*)src[u], lds_ptr(base + dst[u]), 16, 0, 0);
            src[u] += step;
        }
        if (ws_wave) {
            __builtin_amdgcn_global_load_lds((gbl_void*)wsrc, lds_ptr(base + D::W_OFF), 4, 0, 0);
            wsrc += 128;
        }
    };
    const int aoff = q * D::ROWB + (wm0 + 2 * i) * 8, boff = D::A_BYTES + q * D::ROWB + (wn0 + 2 * i) * 8;
    v4d acc[2][4];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = v4d{0, 0, 0, 0};
    v2d sacc = v2d{0, 0};
    const int nst = (int)((r1 - r0) / 16);
    if (nst > 0) issue(0);
    int slot = 0;
    for (int s = 0; s < nst; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's share of stage s (issued a stage ago) has landed
        __builtin_amdgcn_s_barrier();                          // everybody's has; nobody reads the other slot any more
        asm volatile("" ::: "memory");
        if (s + 1 < nst) issue(slot ^ 1);
        const char* base = smem + slot * D::STAGE;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            v2d a = *reinterpret_cast<const v2d*>(base + aoff + kk * (4 * D::ROWB));
            const v2d b0 = *reinterpret_cast<const v2d*>(base + boff + kk * (4 * D::ROWB));
            const v2d b1 = *reinterpret_cast<const v2d*>(base + boff + kk * (4 * D::ROWB) + 32 * 8);
            if (DIAG) sacc += *reinterpret_cast<const double*>(base + D::S_OFF + (4 * kk + q) * 8) * a;
            if (WEIGHT) a *= *reinterpret_cast<const double*>(base + D::W_OFF + (4 * kk + q) * 8);
            const double b[4] = {b0[0], b0[1], b1[0], b1[1]};
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
        }
        slot ^= 1;
    }
    // accumulator (tm, tn, r) of lane (i, q) is output row wm0 + 2 (q + 4 r) + tm, column wn0 + 32 (tn >> 1) + 2 i + (tn & 1);
    // fp64 runs one chunk: the slab is written once
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double* d = slab + (wm0 + 2 * (q + 4 * r) + tm) * D::B + wn0 + 2 * i;
            *reinterpret_cast<v2d*>(d) = v2d{acc[tm][0][r], acc[tm][1][r]};
            *reinterpret_cast<v2d*>(d + 32) = v2d{acc[tm][2][r], acc[tm][3][r]};
        }
    if constexpr (DIAG) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            double x = sacc[tm];
            x += __shfl_xor(x, 16); x += __shfl_xor(x, 32);
            if (q == 0 && wn0 == 0) sideout[wm0 + 2 * i + tm] = x;
        }
    }
}

// Job list of one row split (all kernels of the launch have 8 waves):
//   !BIG  diagonal 128 x 128 tiles, strictly lower tiles row by row, strip tiles
//    BIG  (fp32) pairs of 128-row blocks are covered by 256 x 128 tiles (64 x 64 wave tiles, the shape of the apply
//         product): tile (a, b), b <= 2a, is blocks (2a, b) and (2a+1, b); the diagonal blocks (2a+1, 2a+1) and the strip stay
//         128- / 64-row tiles.  Tall tiles with b == 2a hold the diagonal block of both of their column blocks' rows and carry
//         the side vector for all 256 columns.  An unpaired last block row i (odd block count) runs as tall tiles too, TRANSPOSED:
//         A = column blocks (2b', 2b'+1), B = block i, flushed into tiles (i, 2b') and (i, 2b'+1); its diagonal block is a
//         128 x 128 tile with the side sums of its columns.
// Jobs are split-major and the XCD map hands each XCD a contiguous range of them (whole splits), so the workgroups
// running together on one L2 work on the same rows and share operand panels; inside a split the longest jobs come
// first and the short strip jobs last.
template <bool BIG> __host__ __device__ inline int gram_jobs_per_split(int nfull, int nstrip) {
    if (!BIG) return nfull * (nfull + 1) / 2 + nstrip * (nfull + 1);
    const int R = nfull / 2, odd = nfull & 1, nsb = nstrip * (nfull + 1);
    return R * R + nsb / 4 + (R + 1) / 2 + odd * (R + 1) + nsb % 4;  // tall, wide (4 strip tiles each), diagonal blocks two by two, unpaired row (R transposed tall + its diagonal block), single strips
}
// one job (row split, output tile) of the list
template <class Cfg, class SCfg, class BCfg, bool WEIGHT, bool BIG>
__device__ __forceinline__ void gram_job(
    const int j, const typename Cfg::T* __restrict__ Phi, int64_t ld, const double* __restrict__ w, const double* __restrict__ side,
    const float* __restrict__ ws2, const RowSplits& rs, int64_t chunk, int nfull, int nstrip, double* __restrict__ sidepart,
    double* __restrict__ slabs, char* smem_raw) {
    static_assert(Cfg::THREADS == SCfg::THREADS && Cfg::THREADS == BCfg::THREADS && Cfg::THREADS == 512 &&
                  Cfg::BN == SCfg::BN && Cfg::BN == BCfg::BN && Cfg::BM == Cfg::BN && GramDmaWide::BM == SCfg::BM && GramDmaWide::BN == 4 * Cfg::BN,
                  "one launch, five tile shapes");
    TRACE_BEGIN();
    constexpr int B = Cfg::BN;
    const int nall = nfull + nstrip, ntile_all = nall * (nall + 1) / 2;
    const int per_split = gram_jobs_per_split<BIG>(nfull, nstrip);
    const int split = j / per_split;
    int u = j % per_split, acol, bcol, slab_t, slab_t2 = 0, kind = 0;     // kind 0: 128-row tile, 1: strip, 2: 256-row tile, 3: wide strip, 4: two diagonal blocks, 5: transposed 256-row tile
    bool diag = false;
    const auto tri = [](int ti, int tj) { return ti * (ti + 1) / 2 + tj; };
    if (BIG) {
        const int R = nfull / 2, Rp = (R + 1) / 2, nbig = R * R, nsmall = Rp + (nfull & 1) * (R + 1), nwide = nstrip * (nfull + 1) / 4;
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
        } else if (u < nbig + nwide + Rp) {                    // diagonal blocks of the second rows of two pairs (or of the last one)
            const int m = u - nbig - nwide, i = 2 * (2 * m) + 1;
            acol = bcol = i * B; slab_t = tri(i, i);
            if (2 * m + 1 < R) { const int i2 = 2 * (2 * m + 1) + 1; bcol = i2 * B; slab_t2 = tri(i2, i2); kind = 4; }
        } else if (u < nbig + nwide + nsmall) {                // unpaired last block row i: R transposed tall tiles, then its diagonal block
            const int i = nfull - 1, b = u - nbig - nwide - Rp;
            if (b < R) { acol = 2 * b * B; bcol = i * B; slab_t = tri(i, 2 * b); slab_t2 = tri(i, 2 * b + 1); kind = 5; }
            else { acol = bcol = i * B; slab_t = tri(i, i); diag = side != nullptr; }
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
    if (kind == 1) { gram_body<SCfg, WEIGHT, true>(Phi, ld, w, side, r0, r1, chunk, acol, bcol, diag, sideout, slab, nullptr, smem_raw); TRACE_END(kind); return; }
    if constexpr (BIG) {
        if (kind == 2) {
            if (diag) gram_pipe_dma<GramDma, WEIGHT, true>(Phi, ld, ws2, r0, r1, chunk, acol, bcol, sideout, slab, slab2, smem_raw);
            else gram_pipe_dma<GramDma, WEIGHT, false>(Phi, ld, ws2, r0, r1, chunk, acol, bcol, sideout, slab, slab2, smem_raw);
            TRACE_END(kind); return;
        }
        if (kind == 5) {
            gram_pipe_dma<GramDma, WEIGHT, false, true>(Phi, ld, ws2, r0, r1, chunk, acol, bcol, sideout, slab, slab2, smem_raw);
            TRACE_END(kind); return;
        }
        if (kind == 4) {                                       // two diagonal blocks (their columns' side sums come from the tall diagonal tiles)
            gram_pipe_dma<GramDmaPair, WEIGHT, false>(Phi, ld, ws2, r0, r1, chunk, acol, bcol, sideout, slab, slab2, smem_raw);
            TRACE_END(kind); return;
        }
        if (kind == 0) {
            if (diag) gram_pipe_dma<GramDmaSq, WEIGHT, true>(Phi, ld, ws2, r0, r1, chunk, acol, bcol, sideout, slab, nullptr, smem_raw);
            else gram_pipe_dma<GramDmaSq, WEIGHT, false>(Phi, ld, ws2, r0, r1, chunk, acol, bcol, sideout, slab, nullptr, smem_raw);
            TRACE_END(kind); return;
        }
        if (kind == 3) {
            if (diag) gram_pipe_dma<GramDmaWide, WEIGHT, true>(Phi, ld, ws2, r0, r1, chunk, acol, bcol, sideout, slab, nullptr, smem_raw);
            else gram_pipe_dma<GramDmaWide, WEIGHT, false>(Phi, ld, ws2, r0, r1, chunk, acol, bcol, sideout, slab, nullptr, smem_raw);
            TRACE_END(kind); return;
        }
    }
    if constexpr (!BIG && sizeof(typename Cfg::T) == 8) {      // fp64 square tiles by LDS-DMA
        if (diag) gram_sq_dma64<WEIGHT, true>(Phi, ld, w, side, r0, r1, acol, bcol, sideout, slab, smem_raw);
        else gram_sq_dma64<WEIGHT, false>(Phi, ld, w, side, r0, r1, acol, bcol, sideout, slab, smem_raw);
    } else
        gram_body<Cfg, WEIGHT, false>(Phi, ld, w, side, r0, r1, chunk, acol, bcol, diag, sideout, slab, nullptr, smem_raw);
    TRACE_END(kind);
}

// Persistent launch (long job lists): as many workgroups as the chip holds (two per CU), each pulling jobs until the list is empty.  The list is
// cut into 8 contiguous queues, one per XCD (split-major: the workgroups of an XCD work on the same rows and share operand
// panels in its L2, as the static XCD map did); a workgroup serves the queue of the XCD it runs on (XCC_ID) and, once that is
// empty, steals from the others -- the hardware deals workgroups to XCDs statically, so with one job per workgroup an XCD that
// ran 2 % faster idled while the others finished (0.8 ms spread of the last job starts in a 37 ms launch,
// profiles/r04_gram_trace_H.txt).  head[x] counts the jobs handed out from queue x (zeroed by the host before the launch).
// Every workgroup leaves when all eight queues are empty: nobody waits for anybody.
template <class Cfg, class SCfg, class BCfg, bool WEIGHT, bool BIG>
__global__ __launch_bounds__(Cfg::THREADS)
__attribute__((amdgpu_waves_per_eu(4, 4)))       // two 8-wave workgroups per CU: the compiler would take up to 256 VGPRs
void gram_kernel(
    const typename Cfg::T* __restrict__ Phi, int64_t ld, const double* __restrict__ w, const double* __restrict__ side,
    const float* __restrict__ ws2, RowSplits rs, int64_t chunk, int nfull, int nstrip, double* __restrict__ sidepart,
    double* __restrict__ slabs, int njobs, int* __restrict__ head) {
    SMEM_DECL;
    if ((int)gridDim.x >= njobs) {                             // short job lists: one job per workgroup, static XCD map
        gram_job<Cfg, SCfg, BCfg, WEIGHT, BIG>((int)xcd_remap(blockIdx.x, gridDim.x), Phi, ld, w, side, ws2, rs, chunk, nfull, nstrip, sidepart, slabs, smem_raw);
        return;
    }
    __shared__ int s_job;
    const int q8 = njobs / 8, r8 = njobs % 8;
    const int xcd = (int)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7;      // HW_REG_XCC_ID
    for (;;) {
        if (threadIdx.x == 0) {
            int job = -1;
            for (int a = 0; a < 8 && job < 0; ++a) {
                const int x = (xcd + a) & 7, len = q8 + (x < r8 ? 1 : 0);
                if (len <= 0) continue;
                // the unlocked read spares the atomics on queues known to be empty; the count itself decides
                if (__atomic_load_n(&head[x], __ATOMIC_RELAXED) >= len) continue;
                const int t = atomicAdd(&head[x], 1);
                if (t < len) job = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + t;
            }
            s_job = job;
        }
        __syncthreads();                                       // also: every wave is done with the previous job's LDS
        const int j = s_job;
        __syncthreads();                                       // s_job may be rewritten
        if (j < 0) return;
        gram_job<Cfg, SCfg, BCfg, WEIGHT, BIG>(j, Phi, ld, w, side, ws2, rs, chunk, nfull, nstrip, sidepart, slabs, smem_raw);
    }
}

// Row tile ti0 + (t / ntn) of the output (tiles are 128 rows apart whatever Cfg::BM is: a narrower Cfg multiplies only the
// first BM rows of its tile and zeroes the rest of the 128 x 128 slab)
//   PLAIN: the B operand is a stored matrix (Phi[n][j], j < J; Pb unused) instead of Zbar formed from Phi and Phibar
#ifndef SCFGP_XTZ_DEEP
#define SCFGP_XTZ_DEEP 0           // measured, not shipped: see the comment in the kernel
#endif
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
            // The product is bound by the stream of Phi and Phibar (44 KB per 16-row k-tile and workgroup against 0.4 MFLOP).  Fetching
            // three k-tiles ahead (three register sets in the loaders, tile_mainloop_deep3; -DSCFGP_XTZ_DEEP=1) was measured and is NOT
            // what it lacks: 180 VGPRs leave one workgroup per CU and the launch went 3.6-3.9 -> 4.05 ms at the headline shape (C5:
            // equal).  At 4.3-4.7 TB/s it already runs at what a read-only stream reaches on these boxes (scfgp_box_probe out[6];
            // profiles/r05_tuning.md).
            constexpr bool DEEP = SCFGP_XTZ_DEEP && sizeof(T) == 4;      // fp64: three sets of doubles do not fit the registers
            constexpr int NS = DEEP ? 3 : 1;
            NatLoader<double, T, Cfg::BM, Cfg::BK, Cfg::LDA, Cfg::THREADS, false, true, false, NS> la(
                Xt + c0 * Dp + (int64_t)ti * 128, Dp, threadIdx.x, nullptr, Dp - ti * 128);
            const auto run = [&](auto& lb) {
                if constexpr (DEEP) tile_mainloop_deep3<Cfg>(la, lb, (int)((c1 - c0) / Cfg::BK), acc, smem);
                else tile_mainloop<Cfg>(la, lb, (int)((c1 - c0) / Cfg::BK), acc, smem);
            };
            if constexpr (PLAIN) {
                NatLoader<S, T, Cfg::BN, Cfg::BK, Cfg::LDB, Cfg::THREADS, false, true, false, NS> lb(Phi + c0 * ld + tj * Cfg::BN, ld, threadIdx.x, nullptr,
                                                                                                    J - tj * Cfg::BN);
                run(lb);
            } else {
                ZbarLoader<S, T, Cfg::BN, Cfg::BK, Cfg::LDB, Cfg::THREADS, NS> lb(Phi + c0 * ld, Pb + c0 * ld, ld, J, tj * Cfg::BN, threadIdx.x);
                run(lb);
            }
        }
        slab_flush<Cfg>(acc, slab, first);
        first = false;
    }
    if (Cfg::BM < 128)
        for (int e = threadIdx.x; e < (128 - Cfg::BM) * 128; e += Cfg::THREADS) slab[Cfg::BM * 128 + e] = 0.0;
}

static int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}
template <typename T>
int GramKernels<T>::gram_jobs(const Geom& g) { return gram_jobs_per_split<sizeof(T) == 4>(g.gfull, g.gstrip); }

// Row splits: enough workgroups (>> 512 resident) that faster CUs can take more of them, but long jobs -- at least 3072
// rows per unit (fp32) -- so that the slab traffic (nsplit x K^2/2 x 8 B written and re-read by the reduction) and the
// per-job prologue stay small.  Measured: the fp32 job list (77 jobs per split at K = 2112) is fastest at ~55 units for
// N = 2.5e5..1e6 (profiles/r01_tuning.md); a 125 000-row shard at 40 units (28.3 ms per evaluation against 29.1 at the 24 the
// former 5120-row floor gave it: the jobs fill 3.6 rounds of the resident workgroups there, and the last round's tail is a whole
// tall job; profiles/r04_tuning.md).  From 16 units on they are dealt to the 8 XCD groups and the last unit of each group is
// tapered (kernels.h: RowSplits).
RowSplits gram_row_splits(int jobs, int64_t Np, bool f32, int nsplit_override, int taper) {
    int64_t s = ((f32 ? 4224 : 6144) + jobs - 1) / jobs;
    // ... but never fewer than 56 units where the rows allow it: a long job list per split (K = 4224: 297 jobs) would otherwise get
    // 15 units -- below the 16 from which the units are dealt to the XCD groups and tapered -- i.e. 8.7 rounds of 11 ms jobs whose
    // ragged end left the launch 14 ms of tail (profiles/r05_gram_trace_C5.txt: slot occupancy 0.939); with 56 units (80 splits
    // after the taper, the structure of the headline shape) C5's gram + gram_w go 282.6 -> 265.0 ms for 2.3 ms more slab
    // reduction (profiles/r05_tuning.md)
    s = std::max<int64_t>(s, 56);
    const int64_t smax = std::max<int64_t>(Np / (f32 ? 3072 : 2048), 1);
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

// ws2[n] = ((float) w[n] or 1, (float) side[n] or 0), n < Np: what the pipelined fp32 tiles read beside their operand panels
__global__ __launch_bounds__(256) void gram_pack_ws(const double* __restrict__ w, const double* __restrict__ side, float* __restrict__ ws2, int64_t Np) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < Np; n += (int64_t)gridDim.x * 256)
        reinterpret_cast<v2f*>(ws2)[n] = v2f{w ? (float)w[n] : 1.f, side ? (float)side[n] : 0.f};
}
template <typename T>
void GramKernels<T>::gram(const Geom& g, const T* Phi, const double* w, const double* side, const RowSplits& rs, int64_t chunk,
                          double* slabs, double* sidepart, int* qhead, float* ws2, hipStream_t st) {
    typedef typename GramCfg<T, 128>::type Cfg;
    typedef typename GramStripCfg<T>::type SCfg;
    typedef typename GramBigCfg<T>::type BCfg;
    constexpr bool BIG = sizeof(T) == 4;
    const int njobs = gram_jobs(g) * rs.nsplit;
    if (chunk <= 0 || chunk > g.Np) chunk = g.Np;
    chunk = round_up(chunk, 256);                              // splits start and end on 64- or 256-row granules
    constexpr int L0 = Cfg::LDS_BYTES > SCfg::LDS_BYTES ? Cfg::LDS_BYTES : SCfg::LDS_BYTES;
    constexpr int L1 = !BIG && GramDma64::LDS_BYTES > L0 ? GramDma64::LDS_BYTES : L0;
    constexpr int L2 = BIG && GramDma::LDS_BYTES > L1 ? GramDma::LDS_BYTES : L1;
    constexpr int LDS = BIG && GramDmaWide::LDS_BYTES > L2 ? GramDmaWide::LDS_BYTES : L2;
    static_assert(2 * LDS <= 160 * 1024, "two workgroups per CU");
    // Persistent launch with per-XCD queues from 6 rounds of jobs up; below that one job per workgroup.  Measured on one box
    // (profiles/r04_tuning.md): H (12 rounds) gram + gram_w 72.9-73.1 against 73.0-73.5 ms, C5 293.3 against 297.0, but a
    // 125 000-row shard (3.6 rounds) 10.2 against 9.9 ms
    const int resident = 2 * num_cus();                        // two workgroups per CU
    const bool persistent = njobs >= 6 * resident;
    if (persistent) (void)hipMemsetAsync(qhead, 0, sizeof(int) * 8, st);       // the eight queue heads
    if (BIG && (w || side)) hipLaunchKernelGGL(gram_pack_ws, dim3((unsigned)std::min<int64_t>((g.Np + 255) / 256, 2048)), dim3(256), 0, st, w, side, ws2, g.Np);
    const auto launch = [&](auto kernel) {
        allow_big_lds(kernel, LDS);
#ifdef SCFGP_GRAM_LD0
        // diagnostic build only (tools/ab_bench.sh with VARIANT=_ld0): every operand row is row 0, so the launch runs out of the
        // caches -- what the Gram would cost if its panels never crossed the fabric.  Results are meaningless, timing only.
        const int64_t ld = 0;
#else
        const int64_t ld = g.Kp;
#endif
        hipLaunchKernelGGL(kernel, dim3(persistent ? resident : njobs), dim3(Cfg::THREADS), LDS, st, Phi, ld, w, side, (const float*)ws2, rs, chunk,
                           g.gfull, g.gstrip, sidepart, slabs, njobs, qhead);
    };
    if (w) launch(gram_kernel<Cfg, SCfg, BCfg, true, BIG>);
    else launch(gram_kernel<Cfg, SCfg, BCfg, false, BIG>);
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
void GramKernels<T>::tn_plain(const double* A, int Dp, const T* Bm, int64_t ldb, int J, int64_t Np, int nsplit, int64_t chunk, double* slabs,
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
void GramKernels<T>::zbar_inplace(const Geom& g, const T* Phi, T* Phibar, hipStream_t st) {
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
void GramKernels<T>::rsel(const Geom& g, const double* params, T* out, hipStream_t st) {
    hipLaunchKernelGGL(rsel_kernel<T>, dim3(1024), dim3(256), 0, st, params, g.D, g.S, g.M, out, g.Kp, (int)round_up(g.S, 64));
}
template <typename T>
void GramKernels<T>::xtz(const Geom& g, const double* Xt, const T* Phi, const T* Phibar, int nsplit, int64_t chunk, double* slabs,
                          hipStream_t st) {
    tn_product<T, false>(Xt, g.Dp, Phi, Phibar, (int64_t)g.Kp, g.J, g.Np, nsplit, chunk, slabs, st);
}
template struct GramKernels<double>;
template struct GramKernels<float>;

// --------------------------------------------------------------------------
// reductions (deterministic: fixed order over splits)
// --------------------------------------------------------------------------
// packed lower-tile layout of a symmetric Kp x Kp matrix: tile (ti >= tj) number t = ti(ti+1)/2 + tj holds
// its B x B elements row-major at [t*B*B, (t+1)*B*B) -- what the all-reduce of a sharded run moves
// (K^2/2 instead of K^2 doubles).
__global__ __launch_bounds__(256) void reduce_tri_kernel(const double* __restrict__ slabs, int nsplit, int ntiles, int B,
                                                         double* __restrict__ packed) {
    const int t = blockIdx.x;
    // two doubles per thread and eight splits' loads in flight before the first add (the adds themselves stay in split order):
    // the launch streams 1.6 GB at the headline shape and was bound by one dependent load per iteration
    for (int e = 2 * (blockIdx.y * 256 + threadIdx.x); e < B * B; e += 2 * gridDim.y * 256) {
        const double* src = slabs + (int64_t)t * (B * B) + e;
        const int64_t step = (int64_t)ntiles * (B * B);
        v2d s = v2d{0, 0};
        int sp = 0;
        for (; sp + 8 <= nsplit; sp += 8) {
            v2d x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const v2d*>(src + (sp + u) * step);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += x[u];
        }
        for (; sp < nsplit; ++sp) s += *reinterpret_cast<const v2d*>(src + sp * step);
        *reinterpret_cast<v2d*>(packed + (int64_t)t * B * B + e) = s;
    }
}
void reduce_tri_tiles(const double* slabs, int nsplit, int nts, int tile, double* packed, hipStream_t st) {
    const int ntiles = nts * (nts + 1) / 2;
    hipLaunchKernelGGL(reduce_tri_kernel, dim3(ntiles, 16), dim3(256), 0, st, slabs, nsplit, ntiles, tile, packed);
}
// vec[j] = sum over splits of the Gram's side partials for j < ncov (columns covered by diagonal tiles), 0 beyond
// (64 columns x 16 interleaved runs of splits per block, the runs joined in a fixed order: the f16x3 split passes leave a thousand partials)
__global__ __launch_bounds__(1024) void reduce_side_kernel(const double* __restrict__ sidepart, int nsplit, int Kp, int ncov, double* __restrict__ vec) {
    __shared__ double run[16][64];
    const int j = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
    double s = 0;
    if (j < ncov)
        for (int sp = sub; sp < nsplit; sp += 16) s += sidepart[(int64_t)sp * Kp + j];
    run[sub][threadIdx.x & 63] = s;
    __syncthreads();
    if (sub == 0 && j < Kp) {
#pragma unroll
        for (int k = 1; k < 16; ++k) s += run[k][threadIdx.x];
        vec[j] = s;
    }
}
void reduce_side(const double* sidepart, int nsplit, int Kp, int ncov, double* vec, hipStream_t st) {
    hipLaunchKernelGGL(reduce_side_kernel, dim3((Kp + 63) / 64), dim3(1024), 0, st, sidepart, nsplit, Kp, ncov, vec);
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
