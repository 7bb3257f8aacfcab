// Diagnostic, not part of the product library: what the k loop of an apply tile would deliver if the N-sized products ran as the
// THREE-TERM fp16 split of DESIGN.md section 9 (tests/cpu_f16x3_emulation.py shows its accuracy) -- measured before anything is built.
//
// The tile is the shipped fp32 apply tile's (apply.hip: apply_dma_kernel<float, ., 256>): 256 x 256 outputs, 16 waves of 64 x 64, ring of three
// LDS stages of 16 k filled by global_load_lds_dwordx4, one barrier per stage; grid = (N / 256 row blocks) x (8 column tiles), every
// workgroup 132 stages (K = 2112), A streamed from an N x K array of 4-byte elements, B from a K x K one -- the real launch's memory
// traffic.  What differs is what an element IS and what multiplies it:
//   A element: the packed pair (h, l) of two fp16 (same 4 bytes as the fp32 element: the image, the DMA and the traffic are unchanged);
//   B element: 8 bytes, the two derived pairs (bh, bh) and (bl, 0) -- with A's lane vector (ah0, al0, ah1, al1, ...) the two products
//       a . (bh, bh, ...) = sum (ah + al) bh     and     a . (bl, 0, ...) = sum ah bl
//     are the three terms h.h + l.h + h.l of the split (B is the small, cache-resident operand: its doubled bytes cost LDS, not HBM);
//   MFMA: v_mfma_f32_16x16x32_f16 (32 slots = 16 k): 2 per output tile and stage instead of four v_mfma_f32_16x16x4_f32.
// MODE 0: that loop.  MODE 1: the fp32 loop on the same skeleton (compiler-scheduled here, so its rate is a little under the shipped,
// hand-pipelined kernel's: the ratio MODE 0 / MODE 1 is the figure of interest).  MODE 2: MODE 0 without the MFMAs (DMA + LDS reads +
// barriers), MODE 3: MODE 0 without the DMA after the prologue (MFMAs + LDS reads on stale stages).  Results are not checked: timing only.
//   hipcc -O3 --offload-arch=gfx950 tools/f16x3_probe.hip -o tools/f16x3_probe && tools/f16x3_probe [rows]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

constexpr int BM = 256, BN = 256, KTOT = 2112, NST = KTOT / 16;

template <int MODE>
__global__ __launch_bounds__(1024) void tile_loop(const char* __restrict__ A, const char* __restrict__ B, float* __restrict__ out, int64_t lda, int64_t ldb) {
    constexpr bool F16 = MODE != 1;
    constexpr int ROWA = 64, ROWB = F16 ? 128 : 64;                     // bytes per row and 16 k
    constexpr int STAGE = BM * ROWA + BN * ROWB, NDMA = STAGE / 1024, DPW = NDMA / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), i = lane & 15, q = lane >> 4;
    const int jt = blockIdx.x % 8; const int64_t rb = blockIdx.x / 8;
    const int wm0 = (wave >> 2) * 64, wn0 = (wave & 3) * 64;
    // DMA instruction t = DPW * wave + u: 1 KiB of the stacked (A rows, then B rows) image; 16 bytes per lane
    const char* src[DPW]; int dst[DPW]; int adv[DPW];
#pragma unroll
    for (int u = 0; u < DPW; ++u) {
        const int t = DPW * wave + u, byte0 = t * 1024;
        if (byte0 < BM * ROWA) {                                        // A: 16 rows of 64 bytes per instruction
            const int r = byte0 / ROWA + lane / 4, c = lane % 4;
            src[u] = A + (rb * BM + r) * lda + ((c ^ ((0x78 >> (2 * ((r >> 2) & 3))) & 3)) << 4); adv[u] = 64;
        } else {
            const int off = byte0 - BM * ROWA, per = 1024 / ROWB, r = off / ROWB + lane / (64 / per), c = lane % (64 / per);
            src[u] = B + (int64_t)(jt * BN + r) * ldb + (c << 4); adv[u] = ROWB;
        }
        dst[u] = byte0;
    }
    const auto fetch = [&](int slot) {
#pragma unroll
        for (int u = 0; u < DPW; ++u) {
            __builtin_amdgcn_global_load_lds((gbl_void*)src[u], (lds_void*)(smem + slot * STAGE + dst[u]), 16, 0, 0);
            src[u] += adv[u];
        }
    };
    v4f acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = v4f{0.f, 0.f, 0.f, 0.f};
    const int swa = (0x78 >> (2 * ((i >> 2) & 3))) & 3;                   // the fp32 image's position swizzle (apply.hip)
    const int swb = F16 ? (int)((0x6BEB08u >> (3 * ((i >> 1) & 7))) & 7) : swa;
    fetch(0); fetch(1);
    int slot = 0;
    for (int s = 0; s < NST; ++s) {
        if (MODE == 3 && s >= 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        else if (s + 1 < NST) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (MODE != 3 && s + 2 < NST) fetch(slot == 0 ? 2 : slot - 1);
        const char* base = smem + slot * STAGE;
        const char* pa = base + (wm0 + i) * ROWA, *pb = base + BM * ROWA + (wn0 + i) * ROWB;
        if constexpr (F16) {
            h8 fa[4], fd[4], fl[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                fa[t] = *reinterpret_cast<const h8*>(pa + t * 16 * ROWA + ((q ^ swa) << 4));
                fd[t] = *reinterpret_cast<const h8*>(pb + t * 16 * ROWB + ((q ^ swb) << 4));
                fl[t] = *reinterpret_cast<const h8*>(pb + t * 16 * ROWB + (((4 + q) ^ swb) << 4));
            }
            if constexpr (MODE != 2) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[a], fd[b], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[a], fl[b], acc[a][b], 0, 0, 0);
                    }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t][0][0] += (float)fa[t][0] + (float)fd[t][1] + (float)fl[t][2];
            }
        } else {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {                              // two halves of 8 k, as the shipped loop reads them
                v2f fa[4], fb[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    fa[t] = *reinterpret_cast<const v2f*>(pa + t * 16 * ROWA + (((2 * hh + (q >> 1)) ^ swa) << 4) + 8 * (q & 1));
                    fb[t] = *reinterpret_cast<const v2f*>(pb + t * 16 * ROWB + (((2 * hh + (q >> 1)) ^ swb) << 4) + 8 * (q & 1));
                }
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a][j], fb[b][j], acc[a][b], 0, 0, 0);
            }
        }
        slot = slot == 2 ? 0 : slot + 1;
    }
    float sum = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) sum += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
    if (sum == 123.456f) out[blockIdx.x] = sum;                           // never true: keeps the loop alive
}

// ---- the Gram's k loop (contraction over the ROWS n): G tile = sum_n Phi[n][acol ..]^T Phi[n][bcol ..] -------------------------------------
// k-major LDS image, as the fp32 Gram's (gram.hip): a stage is 32 rows n of the A panel (256 elements = 1 KiB per row = ONE DMA instruction)
// and of the B panel; rows 1056 bytes apart, so that the rows 4 G + q and 16 + 4 G + q a 32-lane half touches fall into 64 distinct banks.
// Elements in the "plane per 16 columns" form [16 x h | 16 x l] (64 bytes): ds_read_b64_tr_b16 (the transposing LDS read of gfx950) hands
// lane i of a 16-lane group column i of a block of 4 rows x 16 halves, i.e. the four consecutive-n values of ONE element's h (or l) -- two
// such reads are a lane's 8 k of v_mfma_f32_16x16x32_f16, and since both operands come as separate planes the three terms are three
// instructions into one accumulator: Ah.Bh + Al.Bh + Ah.Bl.  256 x 256 tile, 16 waves of 64 x 64, ring of two 66 KB stages.
typedef __fp16 f4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f4 lds_f4;
constexpr int GRS = 1056, GSTAGE = 64 * GRS, GROWS = 12544;                 // rows per job: the headline's row split
__device__ __forceinline__ h8 tr8(const char* p) {                          // rows r0 .. r0+3 (p points at them) and r0+16 .. r0+19
    const f4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_f4*)(uintptr_t)(unsigned)(uintptr_t)p);
    const f4 b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_f4*)(uintptr_t)(unsigned)(uintptr_t)(p + 16 * GRS));
    return __builtin_shufflevector(__builtin_bit_cast(h4, a), __builtin_bit_cast(h4, b), 0, 1, 2, 3, 4, 5, 6, 7);
}
template <int MODE>
__global__ __launch_bounds__(1024) void gram_loop(const char* __restrict__ A, float* __restrict__ out, int64_t lda, int ntile) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = blockIdx.x % ntile; const int64_t split = blockIdx.x / ntile;
    int ti = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    while (ti * (ti + 1) / 2 > t) --ti;
    const int tj = t - ti * (ti + 1) / 2;
    const int wm0 = (wave >> 2) * 64, wn0 = (wave & 3) * 64;
    const char* src[4]; int dst[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = 4 * wave + u;                                         // 0..31: A rows, 32..63: B rows
        src[u] = A + (split * GROWS + (r & 31)) * lda + (int64_t)(r < 32 ? ti : tj) * 1024 + lane * 16;
        dst[u] = r * GRS;
    }
    const auto fetch = [&](int slot) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            __builtin_amdgcn_global_load_lds((gbl_void*)src[u], (lds_void*)(smem + slot * GSTAGE + dst[u]), 16, 0, 0);
            src[u] += 32 * lda;
        }
    };
    const int G = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int offa = (4 * G + q) * GRS + 64 * (wm0 / 16) + 8 * p, offb = 32 * GRS + (4 * G + q) * GRS + 64 * (wn0 / 16) + 8 * p;
    v4f acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = v4f{0.f, 0.f, 0.f, 0.f};
    constexpr int NS = GROWS / 32;
    fetch(0);
    int slot = 0;
    for (int s = 0; s < NS; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (MODE != 3 && s + 1 < NS) fetch(slot ^ 1);
        const char* base = smem + slot * GSTAGE;
        h8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ah[k] = tr8(base + offa + 64 * k); al[k] = tr8(base + offa + 64 * k + 32);
            bh[k] = tr8(base + offb + 64 * k); bl[k] = tr8(base + offb + 64 * k + 32);
        }
        if constexpr (MODE != 2) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a], bh[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[a], bh[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a], bl[b], acc[a][b], 0, 0, 0);
                }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k][0][0] += (float)ah[k][0] + (float)al[k][1] + (float)bh[k][2] + (float)bl[k][3];
        }
        slot ^= 1;
    }
    float sum = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) sum += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
    if (sum == 123.456f) out[blockIdx.x] = sum;
}
template <int MODE> static void run_gram(const char* A, float* out, int64_t N, const char* what) {
    constexpr int LDS = 2 * GSTAGE;
    hipFuncSetAttribute(reinterpret_cast<const void*>(gram_loop<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    const int nb = 8, ntile = nb * (nb + 1) / 2;                          // K = 2048: the lower triangle of 8 x 8 blocks of 256
    const int64_t nsplit = N / GROWS;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(gram_loop<MODE>, dim3((unsigned)(ntile * nsplit)), dim3(1024), LDS, 0, A, out, (int64_t)KTOT * 4, ntile);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double flop = 2.0 * nsplit * GROWS * 256.0 * 256.0 * ntile;      // executed (lower tiles only), as the shipped Gram is priced
    printf("%-62s %8.2f ms   %7.1f fp32-equivalent TFLOP/s executed (LDS %d KB)\n", what, best, flop / (best * 1e-3) / 1e12, LDS / 1024);
    if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); exit(1); }
}

__global__ void fill_kernel(unsigned* p, int64_t n) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
        unsigned x = (unsigned)k * 2654435761u + 12345u; x = x * 1664525u + 1013904223u;
        p[k] = 0x3C003C00u ^ ((x >> 7) & 0x03FF03FFu);
    }
}

template <int MODE> static double run(const char* A, const char* B, float* out, int64_t N, const char* what) {
    constexpr bool F16 = MODE != 1;
    constexpr int STAGE = BM * 64 + BN * (F16 ? 128 : 64), LDS = 3 * STAGE;
    hipFuncSetAttribute(reinterpret_cast<const void*>(tile_loop<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    const int64_t lda = (int64_t)KTOT * 4, ldb = (int64_t)KTOT * (F16 ? 8 : 4);
    const unsigned grid = (unsigned)(N / BM * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(tile_loop<MODE>, dim3(grid), dim3(1024), LDS, 0, A, B, out, lda, ldb);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double flop = 2.0 * N * 2048.0 * KTOT;                          // the fp32 product these tiles stand for (8 x 256 columns)
    printf("%-62s %8.2f ms   %7.1f fp32-equivalent TFLOP/s  (LDS %d KB)\n", what, best, flop / (best * 1e-3) / 1e12, LDS / 1024);
    if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); exit(1); }
    return best;
}

int main(int argc, char** argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) / 256 * 256 : 1000192;
    char *A, *B; float* out;
    const size_t abytes = (size_t)N * KTOT * 4, bbytes = (size_t)KTOT * KTOT * 8;
    if (hipMalloc(&A, abytes) != hipSuccess || hipMalloc(&B, bbytes) != hipSuccess || hipMalloc(&out, 4 * (N / BM * 8)) != hipSuccess) { printf("alloc failed\n"); return 1; }
    // fp16 values in [1, 2) with random mantissas (as fp32 the same bytes are ~1e-2): constant operands would let the chip hold a
    // higher clock than real data does
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (unsigned*)A, (int64_t)(abytes / 4));
    hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, 0, (unsigned*)B, (int64_t)(bbytes / 4));
    hipDeviceSynchronize();
    printf("apply tile k loop, %lld rows x K = %d, 256 x 256 tiles on 16 waves, ring of three 16-k stages\n", (long long)N, KTOT);
    const double f32 = run<1>(A, B, out, N, "fp32 (v_mfma_f32_16x16x4_f32, compiler-scheduled skeleton)");
    const double f16 = run<0>(A, B, out, N, "f16x3 (2 x v_mfma_f32_16x16x32_f16 per tile and stage)");
    run<2>(A, B, out, N, "f16x3 without the MFMAs (DMA + LDS reads + barriers)");
    run<3>(A, B, out, N, "f16x3 without the DMA after the prologue (MFMAs + LDS reads)");
    printf("ratio fp32 / f16x3 on the same skeleton: %.2f\n", f32 / f16);
    printf("Gram k loop (K = 2048: 36 lower tiles of 256 x 256, %d-row jobs, stages of 32 rows, transposing LDS reads); the shipped fp32 Gram runs at 0.87-0.91 x 157.3 executed\n", GROWS);
    run_gram<0>(A, out, N, "f16x3 (3 x v_mfma_f32_16x16x32_f16 per tile and 32 rows)");
    run_gram<2>(A, out, N, "f16x3 without the MFMAs (DMA + transposing reads + barriers)");
    run_gram<3>(A, out, N, "f16x3 without the DMA after the prologue");
    return 0;
}
