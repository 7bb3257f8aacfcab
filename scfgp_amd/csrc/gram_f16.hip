// Compute mode SCFGP_F16X3, second half (apply_f16.hip is the first): the two TN products G = Phi^T Phi and
// B W B = V^T diag(q) V as a three-term fp16 split.  A labelled SECONDARY mode; what it must equal is fp32 mode (its parity tier).
//
// The contraction runs over the ROWS n, so the operand image in the LDS is k-major as in the fp32 Gram (gram.hip): a stage is 32 rows
// of the A panel (256 elements = 1 KiB per row = ONE global_load_lds instruction) and 32 of the B panel.  Elements come in the
// "plane per 16 columns" form [16 x h | 16 x l] (64 bytes per 16 columns, written by the split passes below): gfx950's transposing LDS
// read ds_read_b64_tr_b16 hands lane i of a 16-lane group column i of a block of 4 rows x 16 halves -- the four consecutive-n values of
// ONE element's h (or l); two such reads are a lane's 8 k of v_mfma_f32_16x16x32_f16.  Both operands arrive as separate planes, so the
// three terms are three instructions into ONE accumulator, Ah.Bh + Al.Bh + Ah.Bl -- 48 cycles per output tile and 32 rows where exact
// fp32 spends 256.  Rows lie 1056 bytes apart in the LDS: the rows 4 G + q (G = lane group) and 16 + 4 G + q a 32-lane half touches then
// fall into 64 distinct banks.  tools/f16x3_probe.hip measured this loop at 403 fp32-equivalent TFLOP/s (executed) before it was built.
//
// Tile 256 x 256 on 16 waves of 64 x 64, ring of two 66 KB stages, one workgroup per CU.  A job is (row chunk, lower 256-block pair): the
// fp32 accumulators live for `chunk` rows (the fp32 Gram's flush interval, so the same accumulation error) and are then WRITTEN, as
// doubles, to the chunk's own 128 x 128 slabs -- a chunk is a "split" of the shared reduction (reduce_tri_tiles), which sums the chunks in
// fp64.  (Read-modify-write flushes into per-split slabs, as the fp32 Gram does them, would cost this kernel 40 % -- it runs through
// 4096 rows in 0.16 ms.)  Row weights cannot ride along (an MFMA sums 32 rows at once): the weighted product takes V as one operand and
// q o V as the other, both written by split_v in one pass over V.  The side vectors Phi^T y and V^T p are by-products of the split
// passes (fp64 block partials), not of this kernel.
#include "kernels.h"
#include "tile_cfgs.h"
#include "tile_engine.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

namespace {
constexpr int RS = 1056, STAGE = 64 * RS, LDS_BYTES = 2 * STAGE, SIDE_ROWS = 1024;     // SIDE_ROWS: rows per block of the split passes
// Transposing read of 4 rows x 16 halves at LDS byte address `addr` + OFF: lane i of a 16-lane group receives column i.  Inline assembly
// with hand-counted lgkmcnt waits (SCFGP_F16_WAIT), as the fp32 Gram reads its fragments: through the builtin the compiler puts an
// s_waitcnt vmcnt(0) in front of the reads (it cannot tell the LDS-DMA of the NEXT stage from the data being read), which serialises
// every stage's fetch with its multiplication -- 15.1 ms per launch at the headline shape instead of what stands in profiles/r05_tuning.md.
template <int OFF> __device__ __forceinline__ void tr_read(h4& out, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(out) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ h8 cat(const h4& a, const h4& b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }
// "s_waitcnt lgkmcnt(n)" that formally PRODUCES the fragment halves it releases (in-out operands: no use can be ordered before it)
#define SCFGP_F16_WAIT(n, x) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[1][0]), "+v"(x[1][1]),         \
                                          "+v"(x[2][0]), "+v"(x[2][1]), "+v"(x[3][0]), "+v"(x[3][1]) :: "memory")
__device__ __forceinline__ void split2(float x, _Float16& h, _Float16& l) { h = (_Float16)x; l = (_Float16)(x - (float)h); }
}

// A16 / B16: Np x ld16 elements in plane form (the same array for the plain product); job = blockIdx: chunk-major, tiles (ti >= tj) of
// 256 x 256; nb256 = number of 256-column blocks covering Kp; slabs: [chunk][tri(128-tile)][128 x 128] doubles, every one written.
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4)))
void gram_f16_kernel(const unsigned* __restrict__ A16, const unsigned* __restrict__ B16, int64_t ld16, int64_t Np, int64_t chunk, int nb256,
                     int nts128, const float* __restrict__ scale, double* __restrict__ slabs) {
    SMEM_DECL;
    char* smem = smem_raw;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntile = nb256 * (nb256 + 1) / 2;
    const unsigned wid = xcd_remap(blockIdx.x, gridDim.x);
    const int t = (int)(wid % ntile), ch = (int)(wid / ntile);
    int ti = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    while (ti * (ti + 1) / 2 > t) --ti;
    const int tj = t - ti * (ti + 1) / 2;
    const int64_t r0 = (int64_t)ch * chunk, r1 = r0 + chunk < Np ? r0 + chunk : Np;
    const int wm0 = (wave >> 2) * 64, wn0 = (wave & 3) * 64;
    // the wave's 64 x 64 outputs lie in ONE 128 x 128 tile (i128, j128) of the packed lower triangle; a diagonal 256-tile's upper 128-tile
    // is not stored (its transpose is), nor is anything beyond Kp (the last 256-block may stick out: its operand reads run into the next
    // row -- finite fp16 data -- and, on the last rows, into the arrays' padding)
    const int i128 = 2 * ti + (wm0 >> 7), j128 = 2 * tj + (wn0 >> 7);
    const bool stored = i128 >= j128 && i128 < nts128 && j128 < nts128;
    const int ntile128 = nts128 * (nts128 + 1) / 2;
    double* slab = slabs + ((int64_t)ch * ntile128 + (stored ? i128 * (i128 + 1) / 2 + j128 : 0)) * (128 * 128) + (int64_t)(wm0 & 127) * 128 + (wn0 & 127);
    // DMA instruction u of this wave: row r = 4 wave + u of the stage's 64 (0..31 A panel, 32..63 B panel), 16 bytes per lane
    const char* src[4]; int dst[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = 4 * wave + u;
        const unsigned* base = r < 32 ? A16 + (int64_t)ti * 256 : B16 + (int64_t)tj * 256;
        src[u] = reinterpret_cast<const char*>(base + (r0 + (r & 31)) * ld16) + lane * 16;
        dst[u] = r * RS;
    }
    const int64_t step = 32 * ld16 * 4;
    const auto fetch = [&](int slot) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            __builtin_amdgcn_global_load_lds((gbl_void*)src[u], (lds_void*)(smem + slot * STAGE + dst[u]), 16, 0, 0);
            src[u] += step;
        }
    };
    // transposing read of lane 16 G + 4 q + p: row 4 G + q of the stage, bytes 8 p .. of the 32-byte plane block
    const int G = lane >> 4, q = (lane >> 2) & 3, p = lane & 3, i = lane & 15;
    const int offa = (4 * G + q) * RS + 64 * (wm0 / 16) + 8 * p, offb = (32 + 4 * G + q) * RS + 64 * (wn0 / 16) + 8 * p;
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    v4f acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = v4f{0.f, 0.f, 0.f, 0.f};
    const int nst = (int)((r1 - r0) / 32);
    if (nst > 0) fetch(0);
    int slot = 0;
    for (int s = 0; s < nst; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's share of stage s (issued a stage ago) has landed
        __builtin_amdgcn_s_barrier();                          // everybody's has; nobody reads the other slot any more
        asm volatile("" ::: "memory");
        if (s + 1 < nst) fetch(slot ^ 1);
        // the wave's fragments of stage s: per 16 columns the h plane, then (+32 bytes) the l plane; rows 4 G + q and (+16 RS) 16 + 4 G + q.
        // At most three of the four fragment sets are live (48 registers beside the 64 accumulators): Ah, Bh and Al are read up front,
        // Bl moves into Al's registers while the middle term is multiplied.
        const unsigned pa = lds0 + slot * STAGE + offa, pb = lds0 + slot * STAGE + offb;
        h4 ah[4][2], al[4][2], bh[4][2], bl[4][2];
        static_for<4>([&](auto kc) { constexpr int k = decltype(kc)::value;
            tr_read<64 * k>(ah[k][0], pa); tr_read<64 * k + 16 * RS>(ah[k][1], pa); });
        static_for<4>([&](auto kc) { constexpr int k = decltype(kc)::value;
            tr_read<64 * k>(bh[k][0], pb); tr_read<64 * k + 16 * RS>(bh[k][1], pb); });
        static_for<4>([&](auto kc) { constexpr int k = decltype(kc)::value;
            tr_read<64 * k + 32>(al[k][0], pa); tr_read<64 * k + 32 + 16 * RS>(al[k][1], pa); });
        SCFGP_F16_WAIT(8, ah);                                 // all but the 8 reads of Al (the counter's field ends at 15)
        SCFGP_F16_WAIT(8, bh);
        __builtin_amdgcn_sched_barrier(0);
        static_for<16>([&](auto ic) { constexpr int a = decltype(ic)::value / 4, b = decltype(ic)::value % 4;
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cat(ah[a][0], ah[a][1]), cat(bh[b][0], bh[b][1]), acc[a][b], 0, 0, 0); });
        __builtin_amdgcn_sched_barrier(0);
        SCFGP_F16_WAIT(0, al);
        __builtin_amdgcn_sched_barrier(0);
        static_for<4>([&](auto ac) { constexpr int a = decltype(ac)::value;
            static_for<4>([&](auto bc) { constexpr int b = decltype(bc)::value;
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cat(al[a][0], al[a][1]), cat(bh[b][0], bh[b][1]), acc[a][b], 0, 0, 0); });
            __builtin_amdgcn_sched_barrier(0);
            tr_read<64 * a + 32>(bl[a][0], pb); tr_read<64 * a + 32 + 16 * RS>(bl[a][1], pb);       // Al[a] is dead: its registers are free
            __builtin_amdgcn_sched_barrier(0);
        });
        SCFGP_F16_WAIT(0, bl);
        __builtin_amdgcn_sched_barrier(0);
        static_for<16>([&](auto ic) { constexpr int a = decltype(ic)::value / 4, b = decltype(ic)::value % 4;
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cat(ah[a][0], ah[a][1]), cat(bl[b][0], bl[b][1]), acc[a][b], 0, 0, 0); });
        __builtin_amdgcn_sched_barrier(0);
        slot ^= 1;
    }
    if (!stored) return;
    // accumulator (a, b, r) of lane (i, G): output row wm0 + 16 a + 4 G + r, column wn0 + 16 b + i -- IF the A operand's tile index runs
    // over the MFMA's rows; it is passed as the instruction's first operand, whose index the MFMA puts on the output ROWS
    const double sc = (double)scale[0];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int b = 0; b < 4; ++b) slab[(16 * a + 4 * G + r) * 128 + 16 * b + i] = (double)acc[a][b][r] * sc;
}

// One pass over a resident fp32 matrix M (Np x Kp): its elements in plane form (out_pl) and, times rw[n], in plane form again (out_w;
// with rw, or neither); the fp64 block partials of the side vector
// sum_n sw[n] M[n][j] -- part[block][Kp], SIDE_ROWS rows per block.  Scaled by 2^e, e = 14 - ilogb(bound): bnd[0] bounds |M|, bnd[0] bnd[1]
// bounds |rw M|; es[0] = 2^-e, es[1] = 2^-e_w, es[2] = the scale of the Gram product that follows: 2^-(e + e_w), or 2^-2e without rw.
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ M, int64_t Np, int Kp, const double* __restrict__ rw,
                                                         const double* __restrict__ sw, const float* __restrict__ bnd,
                                                         unsigned* __restrict__ out_pl, unsigned* __restrict__ out_w,
                                                         double* __restrict__ part, float* __restrict__ es) {
    const float b0 = bnd[0], b1 = rw ? b0 * bnd[1] : 0.f;
    const int e0 = b0 > 0.f ? 14 - ilogbf(b0) : 0, e1 = b1 > 0.f ? 14 - ilogbf(b1) : 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { es[0] = ldexpf(1.0f, -e0); es[1] = ldexpf(1.0f, -e1); es[2] = ldexpf(1.0f, rw ? -e0 - e1 : -2 * e0); }
    const float up0 = ldexpf(1.0f, e0);
    const double up1 = ldexp(1.0, e1);
    const int64_t n0 = (int64_t)blockIdx.x * SIDE_ROWS, n1 = n0 + SIDE_ROWS < Np ? n0 + SIDE_ROWS : Np;
    for (int c4 = threadIdx.x; c4 < Kp / 4; c4 += 256) {
        const int c = 4 * c4;
        const int64_t poff = 64 * (c >> 4) + 2 * (c & 15);                    // byte offset of the 4 h's inside a row of plane form
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll 4
        for (int64_t n = n0; n < n1; ++n) {
            const v4f x = *reinterpret_cast<const v4f*>(M + n * Kp + c);
            const double w = sw[n];
            s0 += w * x[0]; s1 += w * x[1]; s2 += w * x[2]; s3 += w * x[3];
            const v4f xs = x * up0;
            _Float16 h[4], l[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) split2(xs[k], h[k], l[k]);
            char* row = reinterpret_cast<char*>(out_pl + n * Kp) + poff;
            *reinterpret_cast<h4*>(row) = h4{h[0], h[1], h[2], h[3]};
            *reinterpret_cast<h4*>(row + 32) = h4{l[0], l[1], l[2], l[3]};
            if (rw) {
                const float f = (float)(rw[n] * up1);
#pragma unroll
                for (int k = 0; k < 4; ++k) split2(x[k] * f, h[k], l[k]);
                row = reinterpret_cast<char*>(out_w + n * Kp) + poff;
                *reinterpret_cast<h4*>(row) = h4{h[0], h[1], h[2], h[3]};
                *reinterpret_cast<h4*>(row + 32) = h4{l[0], l[1], l[2], l[3]};
            }
        }
        double* d = part + (int64_t)blockIdx.x * Kp + c;
        d[0] = s0; d[1] = s1; d[2] = s2; d[3] = s3;
    }
}
// out[0] = max |v[n]|, n < N (one workgroup)
__global__ __launch_bounds__(1024) void maxabs_vec_kernel(const double* __restrict__ v, int64_t N, float* __restrict__ out) {
    __shared__ double r1[1024];
    double m = 0;
    for (int64_t n = threadIdx.x; n < N; n += 1024) m = fmax(m, fabs(v[n]));
    r1[threadIdx.x] = m;
    __syncthreads();
    for (int w = 512; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) r1[threadIdx.x] = fmax(r1[threadIdx.x], r1[threadIdx.x + w]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(r1[0] * (1.0 + 1e-6));
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

int F16x3Kernels::side_blocks(const Geom& g) { return (int)((g.Np + SIDE_ROWS - 1) / SIDE_ROWS); }

void F16x3Kernels::split_phi(const Geom& g, const float* Phi, const double* y, const Scal* sc, unsigned* Phi16, double* sidepart, float* tmp,
                             hipStream_t st) {
    hipLaunchKernelGGL(phi_bound_kernel, dim3(1), dim3(64), 0, st, tmp, sc);
    hipLaunchKernelGGL(split_rows_kernel, dim3(side_blocks(g)), dim3(256), 0, st, Phi, g.Np, g.Kp, (const double*)nullptr, y, (const float*)tmp,
                       Phi16, (unsigned*)nullptr, sidepart, tmp + 2);
}
void F16x3Kernels::split_v(const Geom& g, const float* V, const double* B, const double* q, const double* p, const Scal* sc, unsigned* V16g,
                           unsigned* qV16g, double* sidepart, float* tmp, hipStream_t st) {
    hipMemsetAsync(tmp, 0, sizeof(float), st);
    hipLaunchKernelGGL(v_bound_kernel, dim3((g.K + 3) / 4), dim3(256), 0, st, B, g.K, g.Kp, sc, sqrt((double)g.M), tmp);
    hipLaunchKernelGGL(maxabs_vec_kernel, dim3(1), dim3(1024), 0, st, q, g.N, tmp + 1);
    hipLaunchKernelGGL(split_rows_kernel, dim3(side_blocks(g)), dim3(256), 0, st, V, g.Np, g.Kp, q, p, (const float*)tmp, V16g, qV16g, sidepart,
                       tmp + 2);
}
int F16x3Kernels::gram_chunks(const Geom& g, int64_t chunk) {
    if (chunk <= 0 || chunk > g.Np) chunk = g.Np;
    chunk = round_up(chunk, 256);
    return (int)((g.Np + chunk - 1) / chunk);
}
void F16x3Kernels::gram(const Geom& g, const unsigned* A16g, const unsigned* B16g, const float* scale, int64_t chunk, double* slabs, hipStream_t st) {
    const int nb256 = (g.Kp + 255) / 256, ntile = nb256 * (nb256 + 1) / 2;
    if (chunk <= 0 || chunk > g.Np) chunk = g.Np;
    chunk = round_up(chunk, 256);
    allow_big_lds(gram_f16_kernel, LDS_BYTES);
    hipLaunchKernelGGL(gram_f16_kernel, dim3((unsigned)(ntile * gram_chunks(g, chunk))), dim3(1024), LDS_BYTES, st, A16g, B16g, (int64_t)g.Kp, g.Np,
                       chunk, nb256, g.Kp / 128, scale, slabs);
}
