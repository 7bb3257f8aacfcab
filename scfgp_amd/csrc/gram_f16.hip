// Compute mode SCFGP_F16X3, second half (apply_f16.hip is the first): the two TN products G = Phi^T Phi and
// B W B = V^T diag(q) V as a three-term fp16 split.  A labelled SECONDARY mode; what it must equal is fp32 mode (its parity tier).
//
// The contraction runs over the ROWS n, so the operand image in the LDS is k-major as in the fp32 Gram (gram.hip): a stage is 32 rows
// of the A panel (256 elements = 1 KiB per row = ONE global_load_lds instruction) and 32 of the B panel (128).  Elements come in the
// "plane per 16 columns" form [16 x h | 16 x l] (64 bytes per 16 columns, written by the split passes below): gfx950's transposing LDS
// read ds_read_b64_tr_b16 hands lane i of a 16-lane group column i of a block of 4 rows x 16 halves -- the four consecutive-n values of
// ONE element's h (or l); two such reads are a lane's 8 k of v_mfma_f32_16x16x32_f16.  Both operands arrive as separate planes, so the
// three terms are three instructions into ONE accumulator, Ah.Bh + Al.Bh + Ah.Bl -- 48 cycles per output tile and 32 rows where exact
// fp32 spends 256.  Rows lie 1056 bytes apart in the LDS: the rows 4 G + q (G = lane group) and 16 + 4 G + q a 32-lane half touches then
// fall into 64 distinct banks.  tools/f16x3_probe.hip measured this loop at 403 fp32-equivalent TFLOP/s (executed) before it was built.
//
// The accumulation has TWO levels.  The fp16 matrix instruction truncates when it adds into a large fp32 accumulator: over a 4096-row chain
// (the fp32 Gram's flush interval) G came out 4e-8 from fp64's, negative on the diagonal, where the fp32 Gram's is 5e-9, and the error grew
// linearly with the chain (profiles/r05_tuning.md: a first version with 256 x 256 tiles and one accumulator set, 13.3 ms per launch at the
// headline shape against this one's 14.9).  Here a chain ends after FOLD stages = 512 rows and the accumulators are folded into a second
// set by ordinary fp32 additions (round to nearest; 16 per 8192 rows): G is 7e-9 from fp64's, alpha 1.5 x fp32 mode's error.  What is
// left on the diagonal (-6e-8 relative) is the split's dropped l.l term, which is a sum of squares there.
//
// The second set costs the registers of half the wave tile: 64 x 32 per wave, workgroup tile 256 x 128 on 16 waves (A panel 256 columns,
// B panel 128), 24 matrix instructions and 24 transposing reads per wave and stage, ring of three 49.5 KB stages (two fetches in flight),
// one workgroup per CU.  B rows are 512 bytes: LDS line rho holds rows rho and rho + 16 of the stage side by side, one DMA instruction
// fills it (32 lanes per row).  A job is (row chunk, 256-block ti of the A side, 128-tile tj <= 2 ti + 1 of the B side); after `chunk`
// rows the sums are WRITTEN, as doubles, to the chunk's own 128 x 128 slabs -- a chunk is a "split" of the shared reduction
// (reduce_tri_tiles), which sums the chunks in fp64.  (Read-modify-write flushes into per-split slabs, as the fp32 Gram does them, would
// cost this kernel 40 %.)  Row weights cannot ride along (an MFMA sums 32 rows at once): the weighted product takes V as one operand and
// q o V as the other -- V's planes come out of the epilogue of V = Phi B (apply_f16.hip), q o V's out of split_v's pass over V once the
// row statistics have made q.  The side vectors Phi^T y and V^T p are by-products of the split passes (fp64 block partials), not of
// this kernel.
#include "kernels.h"
#include "tile_cfgs.h"
#include "tile_engine.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

namespace {
constexpr int RS = 1056, FSTAGE = 48 * RS, FSLOTS = 3, FLDS_BYTES = FSLOTS * FSTAGE, FOLD = 16, SIDE_ROWS = 512;      // SIDE_ROWS: rows per block of the split passes
// Transposing read of 4 rows x 16 halves at LDS byte address `addr` + OFF: lane i of a 16-lane group receives column i.  Inline assembly
// with hand-counted lgkmcnt waits (SCFGP_F16_WAIT), as the fp32 Gram reads its fragments: through the builtin the compiler puts an
// s_waitcnt vmcnt(0) in front of the reads (it cannot tell the LDS-DMA of the NEXT stage from the data being read), which serialises
// every stage's fetch with its multiplication (15.1 instead of 13.4 ms per launch at the headline shape, measured on the first version).
template <int OFF> __device__ __forceinline__ void tr_read(h4& out, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(out) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ h8 cat(const h4& a, const h4& b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }
// "s_waitcnt lgkmcnt(n)" that formally PRODUCES the fragment halves it releases (in-out operands: no use can be ordered before it)
#define SCFGP_F16_WAIT(n, x) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[1][0]), "+v"(x[1][1]),         \
                                          "+v"(x[2][0]), "+v"(x[2][1]), "+v"(x[3][0]), "+v"(x[3][1]) :: "memory")
#define SCFGP_F16_WAIT2(n, x) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[1][0]), "+v"(x[1][1]) :: "memory")
__device__ __forceinline__ void split2(float x, _Float16& h, _Float16& l) { h = (_Float16)x; l = (_Float16)(x - (float)h); }
}

// A16 / B16: Np x ld16 elements in plane form (the same array for the plain product); job = blockIdx: chunk-major, then the tiles (ti, tj);
// slabs: [chunk][tri(128-tile)][128 x 128] doubles, every one written.
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4)))
void gram_f16_kernel(const unsigned* __restrict__ A16, const unsigned* __restrict__ B16, int64_t ld16, int64_t Np, int64_t chunk, int ntile,
                          int nts128, const float* __restrict__ scale, double* __restrict__ slabs) {
    SMEM_DECL;
    char* smem = smem_raw;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned wid = xcd_remap(blockIdx.x, gridDim.x);
    const int t = (int)(wid % ntile), ch = (int)(wid / ntile);
    int ti = 0, tj = t;
    for (int cnt = 2 < nts128 ? 2 : nts128; tj >= cnt; cnt = 2 * ti + 2 < nts128 ? 2 * ti + 2 : nts128) { tj -= cnt; ++ti; }
    const int64_t r0 = (int64_t)ch * chunk, r1 = r0 + chunk < Np ? r0 + chunk : Np;
    const int wm0 = (wave >> 2) * 64, wn0 = (wave & 3) * 32;
    // the wave's 64 x 32 outputs lie in the 128 x 128 tile (i128, tj) of the packed lower triangle; not stored: the upper tile of a
    // diagonal pair (its transpose is) and anything beyond Kp (the last 256-block may stick out: its operand reads run into the next row --
    // finite fp16 data -- and, on the last rows, into the arrays' padding)
    const int i128 = 2 * ti + (wm0 >> 7);
    const bool stored = i128 >= tj && i128 < nts128;
    const int ntile128 = nts128 * (nts128 + 1) / 2;
    double* slab = slabs + ((int64_t)ch * ntile128 + (stored ? i128 * (i128 + 1) / 2 + tj : 0)) * (128 * 128) + (int64_t)(wm0 & 127) * 128 + wn0;
    // DMA instruction u of this wave fills line 3 wave + u of the stage's 48: lines 0..31 = the A panel's rows; line 32 + rho = rows rho
    // (lanes 0..31) and rho + 16 (lanes 32..63) of the B panel
    // Lanes 0..31 of an A line carry the columns of 128-tile 2 ti, lanes 32..63 those of 2 ti + 1: the half nobody multiplies (see `stored`)
    // is not fetched -- a third of the stage's bytes in 25 of the 89 tiles of a chunk at the headline shape.  Every instruction keeps
    // some active lanes, so a wave's count of outstanding fetches is 3 per stage whatever the tile.
    const char* src[3]; int dst[3]; bool need[3];
    const int a128 = 2 * ti + (lane >> 5);
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int ln = 3 * wave + u;
        if (ln < 32) src[u] = reinterpret_cast<const char*>(A16 + (r0 + ln) * ld16 + (int64_t)ti * 256) + lane * 16;
        else src[u] = reinterpret_cast<const char*>(B16 + (r0 + (ln - 32) + 16 * (lane >> 5)) * ld16 + (int64_t)tj * 128) + (lane & 31) * 16;
        dst[u] = ln * RS;
        need[u] = ln >= 32 || (a128 >= tj && a128 < nts128);
    }
    const int64_t step = 32 * ld16 * 4;
    const auto fetch = [&](int slot) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            if (need[u]) __builtin_amdgcn_global_load_lds((gbl_void*)src[u], (lds_void*)(smem + slot * FSTAGE + dst[u]), 16, 0, 0);
            src[u] += step;
        }
    };
    const int G = lane >> 4, q = (lane >> 2) & 3, p = lane & 3, i = lane & 15;
    const int offa = (4 * G + q) * RS + 64 * (wm0 / 16) + 8 * p, offb = (32 + 4 * G + q) * RS + 64 * (wn0 / 16) + 8 * p;
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    v4f acc[4][2], sum[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) { acc[a][b] = v4f{0.f, 0.f, 0.f, 0.f}; sum[a][b] = v4f{0.f, 0.f, 0.f, 0.f}; }
    const int nst = (int)((r1 - r0) / 32);
    if (nst > 0) fetch(0);
    if (nst > 1) fetch(1);
    int slot = 0, since = 0;
    for (int s = 0; s < nst; ++s) {
        // this wave's share of stage s has landed when only the fetches of stage s+1 are outstanding
        if (s + 1 < nst) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // everybody's has; nobody reads the slot of stage s-1 any more
        asm volatile("" ::: "memory");
        if (s + 2 < nst) fetch(slot == 0 ? 2 : slot - 1);
        // a wave whose outputs nobody stores keeps its share of the fetches and the barriers but multiplies nothing: 14 % of the wave tiles
        // at the headline shape (the upper half of a diagonal pair, the half of the last 256-block beyond Kp).  The launch is no shorter for
        // it (15.2 ms either way: what bounds it is the rate at which the stages arrive, 8.8 TB/s from the L2s into the LDS, not the matrix
        // pipe at 0.63 busy); the idle instructions just are not executed
        if (!stored) { slot = slot == 2 ? 0 : slot + 1; continue; }
        const unsigned pa = lds0 + slot * FSTAGE + offa, pb = lds0 + slot * FSTAGE + offb;
        h4 ah[4][2], al[4][2], bh[2][2], bl[2][2];
        static_for<4>([&](auto kc) { constexpr int k = decltype(kc)::value;
            tr_read<64 * k>(ah[k][0], pa); tr_read<64 * k + 16 * RS>(ah[k][1], pa); });
        static_for<2>([&](auto kc) { constexpr int k = decltype(kc)::value;
            tr_read<64 * k>(bh[k][0], pb); tr_read<64 * k + 512>(bh[k][1], pb); });
        static_for<4>([&](auto kc) { constexpr int k = decltype(kc)::value;
            tr_read<64 * k + 32>(al[k][0], pa); tr_read<64 * k + 32 + 16 * RS>(al[k][1], pa); });
        static_for<2>([&](auto kc) { constexpr int k = decltype(kc)::value;
            tr_read<64 * k + 32>(bl[k][0], pb); tr_read<64 * k + 32 + 512>(bl[k][1], pb); });
        SCFGP_F16_WAIT(12, ah);                                // in issue order: 8 reads of Ah, 4 of Bh, 8 of Al, 4 of Bl
        SCFGP_F16_WAIT2(12, bh);
        __builtin_amdgcn_sched_barrier(0);
        static_for<8>([&](auto ic) { constexpr int a = decltype(ic)::value / 2, b = decltype(ic)::value % 2;
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cat(ah[a][0], ah[a][1]), cat(bh[b][0], bh[b][1]), acc[a][b], 0, 0, 0); });
        __builtin_amdgcn_sched_barrier(0);
        SCFGP_F16_WAIT(4, al);
        __builtin_amdgcn_sched_barrier(0);
        static_for<8>([&](auto ic) { constexpr int a = decltype(ic)::value / 2, b = decltype(ic)::value % 2;
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cat(al[a][0], al[a][1]), cat(bh[b][0], bh[b][1]), acc[a][b], 0, 0, 0); });
        __builtin_amdgcn_sched_barrier(0);
        SCFGP_F16_WAIT2(0, bl);
        __builtin_amdgcn_sched_barrier(0);
        static_for<8>([&](auto ic) { constexpr int a = decltype(ic)::value / 2, b = decltype(ic)::value % 2;
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cat(ah[a][0], ah[a][1]), cat(bl[b][0], bl[b][1]), acc[a][b], 0, 0, 0); });
        __builtin_amdgcn_sched_barrier(0);
        slot = slot == 2 ? 0 : slot + 1;
        if (++since == FOLD) {                                 // the chain ends here: fold it away by ordinary fp32 additions
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) { sum[a][b] += acc[a][b]; acc[a][b] = v4f{0.f, 0.f, 0.f, 0.f}; }
            since = 0;
        }
    }
    if (!stored) return;
    const double sc = (double)scale[0];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int b = 0; b < 2; ++b) slab[(16 * a + 4 * G + r) * 128 + 16 * b + i] = (double)(sum[a][b][r] + acc[a][b][r]) * sc;
}

// One pass over a resident fp32 matrix M (Np x Kp): its elements in plane form (out_pl; may be NULL) and, times rw[n], in plane form again (out_w;
// with rw, or neither); the fp64 block partials of the side vector
// sum_n sw[n] M[n][j] -- part[block][Kp], SIDE_ROWS rows per block.  Scaled by 2^e, e = 14 - ilogb(bound): bnd[0] bounds |M|, bnd[0] bnd[1]
// bounds |rw M|; es[0] = 2^-e, es[1] = 2^-e_w, es[2] = the scale of the Gram product that follows: 2^-(e + e_w), or 2^-2e without rw.
__global__ __launch_bounds__(1024) void split_rows_kernel(const float* __restrict__ M, int64_t Np, int Kp, const double* __restrict__ rw,
                                                         const double* __restrict__ sw, const float* __restrict__ bnd,
                                                         unsigned* __restrict__ out_pl, unsigned* __restrict__ out_w,
                                                         double* __restrict__ part, float* __restrict__ es) {
    const float b0 = bnd[0], b1 = rw ? b0 * bnd[1] : 0.f;
    const int e0 = b0 > 0.f ? 14 - ilogbf(b0) : 0, e1 = b1 > 0.f ? 14 - ilogbf(b1) : 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { es[0] = ldexpf(1.0f, -e0); es[1] = ldexpf(1.0f, -e1); es[2] = ldexpf(1.0f, rw ? -e0 - e1 : -2 * e0); }
    const float up0 = ldexpf(1.0f, e0);
    const double up1 = ldexp(1.0, e1);
    const int64_t n0 = (int64_t)blockIdx.x * SIDE_ROWS, n1 = n0 + SIDE_ROWS < Np ? n0 + SIDE_ROWS : Np;
    // The four threads of a quad own one 64-byte block [16 h | 16 l] of a row; they swap halves (two quad permutes per dword) so that each
    // stores 16 contiguous bytes of it instead of its own 8 bytes of h's and 8 of l's: a quarter of the cache-line requests (apply_f16.hip)
    const bool low = (threadIdx.x & 2) == 0;                        // lanes 0, 1 of a quad store the h's, lanes 2, 3 the l's
    const auto put = [&](char* row, const _Float16 (&h)[4], const _Float16 (&l)[4]) {
        const h2 h01 = h2{h[0], h[1]}, h23 = h2{h[2], h[3]}, l01 = h2{l[0], l[1]}, l23 = h2{l[2], l[3]};
        const int w[4] = {__builtin_bit_cast(int, h01), __builtin_bit_cast(int, h23), __builtin_bit_cast(int, l01), __builtin_bit_cast(int, l23)};
        int o[4];
#pragma unroll
        for (int d = 0; d < 2; ++d) {                              // first 8 bytes: of quad lane 0 / 2 / 0 / 2; second: of quad lane 1 / 3 / 1 / 3
            const int ha = __builtin_amdgcn_update_dpp(0, w[d], 0x88, 0xF, 0xF, true), la = __builtin_amdgcn_update_dpp(0, w[2 + d], 0x88, 0xF, 0xF, true);
            const int hb = __builtin_amdgcn_update_dpp(0, w[d], 0xDD, 0xF, 0xF, true), lb = __builtin_amdgcn_update_dpp(0, w[2 + d], 0xDD, 0xF, 0xF, true);
            o[d] = low ? ha : la; o[2 + d] = low ? hb : lb;
        }
        *reinterpret_cast<int4*>(row) = int4{o[0], o[1], o[2], o[3]};
    };
    for (int c4 = threadIdx.x; c4 < Kp / 4; c4 += blockDim.x) {             // one sweep: the block has a thread per column quad up to Kp = 4096
        const int c = 4 * c4;
        const int64_t poff = 64 * (c >> 4) + 16 * (c4 & 3);                   // this thread's 16 bytes inside a row of plane form
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        for (int64_t nb = n0; nb < n1; nb += 8) {                  // (n1 - n0 is a multiple of 256) 8 rows' loads in flight: the quad permutes keep
            v4f xr[8]; double wr[8];                               // the compiler from unrolling the row loop itself
#pragma unroll
            for (int u = 0; u < 8; ++u) { xr[u] = *reinterpret_cast<const v4f*>(M + (nb + u) * Kp + c); wr[u] = sw[nb + u]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t n = nb + u;
                const v4f x = xr[u];
                const double w = wr[u];
                s0 += w * x[0]; s1 += w * x[1]; s2 += w * x[2]; s3 += w * x[3];
                const v4f xs = x * up0;
                _Float16 h[4], l[4];
                if (out_pl) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) split2(xs[k], h[k], l[k]);
                    put(reinterpret_cast<char*>(out_pl + n * Kp) + poff, h, l);
                }
                if (rw) {
                    const float f = (float)(rw[n] * up1);
#pragma unroll
                    for (int k = 0; k < 4; ++k) split2(x[k] * f, h[k], l[k]);
                    put(reinterpret_cast<char*>(out_w + n * Kp) + poff, h, l);
                }
            }
        }
        double* d = part + (int64_t)blockIdx.x * Kp + c;
        d[0] = s0; d[1] = s1; d[2] = s2; d[3] = s3;
    }
}
// out[0] (as int bits, zeroed before) = max |v[n]|, n < N
__global__ __launch_bounds__(256) void maxabs_vec_kernel(const double* __restrict__ v, int64_t N, float* __restrict__ out) {
    __shared__ double r1[256];
    double m = 0;
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < N; n += (int64_t)gridDim.x * 256) m = fmax(m, fabs(v[n]));
    r1[threadIdx.x] = m;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) r1[threadIdx.x] = fmax(r1[threadIdx.x], r1[threadIdx.x + w]);
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicMax(reinterpret_cast<int*>(out), __float_as_int((float)(r1[0] * (1.0 + 1e-6))));
}
// out[0] = s: |Phi| <= s = e^b sqrt(2/M) (SCFGP/SCFGP.py:98,102) -- the same number split_operand (apply_f16.hip) derives e_Phi from
__global__ void phi_bound_kernel(float* out, const Scal* sc) { if (threadIdx.x == 0) out[0] = (float)sc->s; }
// out[0] (as int bits, zeroed before) = max over rows j of s sqrt(M) |B_j|: every row of Phi has norm s sqrt(M) (cos^2 + sin^2 = 1 per
// feature), so by Cauchy-Schwarz this bounds |V| = |Phi B|.  It may be loose by up to sqrt(K); that costs the split no accuracy that shows
// (elements 2^-12 of the bound and above keep all 22 bits, the floor is 2^-40 of it).  One wave per row of the symmetric B.
__global__ __launch_bounds__(256) void v_bound_kernel(const double* __restrict__ B, int K, int Kp, const Scal* __restrict__ sc, double rootM, float* out) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= K) return;
    double s = 0;
    for (int k = lane; k < K; k += 64) { const double b = B[(int64_t)j * Kp + k]; s += b * b; }
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) atomicMax(reinterpret_cast<int*>(out), __float_as_int((float)(sc->s * rootM * sqrt(s) * (1.0 + 1e-6))));
}

static int split_threads(const Geom& g) { return (int)std::min<int64_t>(1024, round_up(g.Kp / 4, 64)); }
int F16x3Kernels::side_blocks(const Geom& g) { return (int)((g.Np + SIDE_ROWS - 1) / SIDE_ROWS); }

void F16x3Kernels::split_phi(const Geom& g, const float* Phi, const double* y, const Scal* sc, unsigned* Phi16, double* sidepart, float* tmp,
                             hipStream_t st) {
    hipLaunchKernelGGL(phi_bound_kernel, dim3(1), dim3(64), 0, st, tmp, sc);
    hipLaunchKernelGGL(split_rows_kernel, dim3(side_blocks(g)), dim3(split_threads(g)), 0, st, Phi, g.Np, g.Kp, (const double*)nullptr, y, (const float*)tmp,
                       Phi16, (unsigned*)nullptr, sidepart, tmp + 2);
}
void F16x3Kernels::v_bound(const Geom& g, const double* B, const Scal* sc, float* tmp, hipStream_t st) {
    (void)hipMemsetAsync(tmp + 5, 0, 2 * sizeof(float), st);         // tmp[5] = bound of |V|, tmp[6] = max |q| (split_v)
    hipLaunchKernelGGL(v_bound_kernel, dim3((g.K + 3) / 4), dim3(256), 0, st, B, g.K, g.Kp, sc, sqrt((double)g.M), tmp + 5);
}
void F16x3Kernels::split_v(const Geom& g, const float* V, const double* q, const double* p, unsigned* V16g, unsigned* qV16g, double* sidepart,
                           float* tmp, hipStream_t st) {
    hipLaunchKernelGGL(maxabs_vec_kernel, dim3(256), dim3(256), 0, st, q, g.N, tmp + 6);
    hipLaunchKernelGGL(split_rows_kernel, dim3(side_blocks(g)), dim3(split_threads(g)), 0, st, V, g.Np, g.Kp, q, p, (const float*)(tmp + 5), V16g, qV16g,
                       sidepart, tmp + 2);
}
int F16x3Kernels::gram_chunks(const Geom& g, int64_t chunk) {
    if (chunk <= 0 || chunk > g.Np) chunk = g.Np;
    chunk = round_up(chunk, 256);
    return (int)((g.Np + chunk - 1) / chunk);
}
int F16x3Kernels::gram_tiles(const Geom& g) {
    const int nts128 = g.Kp / 128;
    int ntile = 0;
    for (int ti = 0; 2 * ti < nts128; ++ti) ntile += std::min(2 * ti + 2, nts128);
    return ntile;
}
void F16x3Kernels::gram(const Geom& g, const unsigned* A16g, const unsigned* B16g, const float* scale, int64_t chunk, double* slabs, hipStream_t st) {
    if (chunk <= 0 || chunk > g.Np) chunk = g.Np;
    chunk = round_up(chunk, 256);
    const int nts128 = g.Kp / 128, ntile = gram_tiles(g);
    allow_big_lds(gram_f16_kernel, FLDS_BYTES);
    hipLaunchKernelGGL(gram_f16_kernel, dim3((unsigned)(ntile * gram_chunks(g, chunk))), dim3(1024), FLDS_BYTES, st, A16g, B16g, (int64_t)g.Kp, g.Np, chunk,
                       ntile, nts128, scale, slabs);
}
