// Split-precision apply product, second structure (experimental compute mode SCFGP_BF16X3, option bf3_dma; measured
// equal to the loader-split tiles end to end, profiles/r02_tuning.md, so it is off by default): BOTH operands arrive
// pre-split as bf16 planes and go global -> LDS by LDS-DMA (global_load_lds_dwordx4), so the k-loop holds nothing but
// the DMA issue, fragment reads (one ds_read_b128 per fragment and plane) and MFMAs.
//
//   tile 256 x 256, 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 = 4 x 2 MFMA tiles of 32 x 32: 18 fragment reads for
//   48 v_mfma_f32_32x32x16_bf16 per 16-deep stage (the 256 x 128 / 64 x 64 structure of tile_bf16x3.h reads 24)
//   LDS: ring of 3 stages, one stage = 16 k of both operands = 2 x 3 planes x 256 rows x 32 B = 48 KB (144 KB in all);
//        stage s+2 is in flight while stage s is multiplied; one barrier per stage
//   plane layouts in global memory (k16 = 16 consecutive k):
//        rows    [k16][plane][Np rows][16 k] bf16     (bf3_split_rows: Phi)
//        matrix  [k16][plane][Kp cols][16 k] bf16     (bf3_presplit16: B^T / Abar^T, element (k, j) of the fp32 matrix)
//   LDS image of a plane: row x at x * 32 B, its two 16-byte chunks (k 0..7, 8..15) swapped when bit 3 of x is set, so the
//        16-lane groups of a fragment read (rows x .. x+15, one chunk each) touch all 64 banks; the DMA writes LDS
//        linearly (lane l -> byte 16 l of its 1 KB), so the swap is applied to the lane's SOURCE address.
#pragma once
#include "tile_bf16x3.h"

struct Bf3D {
    static constexpr int BM = 256, BN = 256, STAGES = 3;
    static constexpr int PLANE = 256 * 32;                      // one plane of one operand in a stage
    static constexpr int OPER = 3 * PLANE;
    static constexpr int STAGE = 2 * OPER;                      // A planes h, m, l then B planes h, m, l
    static constexpr int LDS_BYTES = STAGES * STAGE;
    static constexpr int DMA_PER_WAVE = STAGE / 1024 / 8;       // 1 KB per wave instruction, 8 waves: 6
};
typedef Bf3Cfg<256, 256, 2, 4> Bf3DCfg;                          // wave grid / accumulator map for the epilogues

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// rows: fp32 source S[n][ld] (n < Np, the first Kp columns) -> row planes.  One block: 64 rows x 128 columns through LDS,
// coalesced 512-byte row reads, 2 KB contiguous plane writes.
__global__ __launch_bounds__(256) void bf3_split_rows_kernel(const float* __restrict__ S, int64_t ld, __bf16* __restrict__ out,
                                                             int64_t Np, int Kp) {
    __shared__ float tile[64][132];
    const int nct = Kp / 128;
    const int ct = blockIdx.x % nct;
    const int64_t n0 = (int64_t)(blockIdx.x / nct) * 64;
    for (int e = threadIdx.x; e < 64 * 32; e += 256) {
        const int r = e / 32, c4 = e % 32;
        const v4f v = *reinterpret_cast<const v4f*>(S + (n0 + r) * ld + ct * 128 + c4 * 4);
        *reinterpret_cast<v4f*>(&tile[r][c4 * 4]) = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 8 * 128; e += 256) {          // (k16 inside the tile, row, 8-k half)
        const int kl = e / 128, w = e % 128, r = w / 2, half = w % 2;
        const float* src = &tile[r][kl * 16 + half * 8];
        bf16x8 h, m, l;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float x = src[i];
            const __bf16 hh = (__bf16)x; const float r1 = x - (float)hh;
            const __bf16 mm = (__bf16)r1; const float r2 = r1 - (float)mm;
            h[i] = hh; m[i] = mm; l[i] = (__bf16)r2;
        }
        const int64_t k16 = ct * 8 + kl;
        __bf16* d = out + ((k16 * 3) * Np + n0 + r) * 16 + half * 8;
        *reinterpret_cast<bf16x8*>(d) = h;
        *reinterpret_cast<bf16x8*>(d + Np * 16) = m;
        *reinterpret_cast<bf16x8*>(d + 2 * Np * 16) = l;
    }
}

// matrix planes from the fp32 matrix M (Kp x Kp, element (k, j) = M[k * Kp + j])
__global__ void bf3_presplit16_kernel(const float* __restrict__ M, __bf16* __restrict__ out, int Kp) {
    const int64_t n = (int64_t)Kp * Kp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i / Kp), j = (int)(i % Kp);       // consecutive threads: consecutive j (coalesced read)
        const float x = M[i];
        const __bf16 h = (__bf16)x; const float r1 = x - (float)h;
        const __bf16 m = (__bf16)r1; const float r2 = r1 - (float)m;
        const int64_t at = (((int64_t)(k >> 4) * 3) * Kp + j) * 16 + (k & 15);
        out[at] = h; out[at + (int64_t)Kp * 16] = m; out[at + (int64_t)2 * Kp * 16] = (__bf16)r2;
    }
}

// The k-loop.  Apl / Bpl: plane bases already offset to the tile's first row / column; a_plane / b_plane: bytes between
// the planes of one k16 (= rows x 32).  nst stages (k16 units).  On return every wave has passed a barrier after its last
// fragment read.
//
// Schedule of stage s (three MFMA groups; fragments are always fetched one group ahead of their use, so that neither the
// LDS latency nor the burst of 8 waves reading at once is exposed):
//   G1  l.h           8 MFMAs   } meanwhile: fragments A_h, B_l of stage s
//   G2  m.h, m.m     16 MFMAs   }
//   -- own DMAs of stage s+1 counted down, barrier (stage s+1 complete, nobody reads slot s any more), DMA of stage s+3
//      into slot s --
//   G3  h.l, h.m, h.h 24 MFMAs    meanwhile: fragments A_l, A_m, B_h, B_m of stage s+1
__device__ __forceinline__ void bf3dma_mainloop(const char* __restrict__ Apl, int64_t a_plane, const char* __restrict__ Bpl,
                                                int64_t b_plane, int nst, Bf3DCfg::MTr::acc_t (&acc)[Bf3DCfg::TM][Bf3DCfg::TN],
                                                char* smem) {
    typedef Bf3DCfg Cfg;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // this wave's six DMA instructions per stage: waves 0..3 fill the A planes, 4..7 the B planes; instruction t of the
    // operand (t = 0..23) is plane t / 8, rows 32 (t % 8) .. +31; lane l carries row +l/2, chunk (l & 1) ^ bit 3 of the row
    const bool isB = wave >= 4;
    const int64_t plane = isB ? b_plane : a_plane;
    const char* src[Bf3D::DMA_PER_WAVE]; int dst[Bf3D::DMA_PER_WAVE];
#pragma unroll
    for (int u = 0; u < Bf3D::DMA_PER_WAVE; ++u) {
        const int t = (wave & 3) * Bf3D::DMA_PER_WAVE + u, pl = t >> 3, blk = t & 7;
        const int x = blk * 32 + (lane >> 1), c = (lane & 1) ^ ((x >> 3) & 1);
        src[u] = (isB ? Bpl : Apl) + (int64_t)pl * plane + x * 32 + c * 16;
        dst[u] = (isB ? Bf3D::OPER : 0) + pl * Bf3D::PLANE + blk * 1024;
    }
    const int64_t step = 3 * plane;                             // one k16 further
    const auto issue = [&](int slot) {
#ifdef SCFGP_DIAG_DMA_NOLOAD                                    // timing diagnostic only (wrong numbers): no operand traffic
        return;
#endif
#pragma unroll
        for (int u = 0; u < Bf3D::DMA_PER_WAVE; ++u) {
            __builtin_amdgcn_global_load_lds((gbl_void*)src[u], (lds_void*)(smem + slot * Bf3D::STAGE + dst[u]), 16, 0, 0);
            src[u] += step;
        }
    };
    const int r = lane & 31, hk = lane >> 5;
    const int wm0 = (wave / Cfg::WGN) * Cfg::WM, wn0 = (wave % Cfg::WGN) * Cfg::WN;
    const int sw = (hk ^ ((r >> 3) & 1)) << 4;
    const int aoff = (wm0 + r) * 32 + sw, boff = Bf3D::OPER + (wn0 + r) * 32 + sw;
    typedef bf16x8 afrag[Cfg::TM];
    typedef bf16x8 bfrag[Cfg::TN];
    const auto ldA = [&](const char* base, int pl, afrag& d) {
#pragma unroll
        for (int tm = 0; tm < Cfg::TM; ++tm) d[tm] = *reinterpret_cast<const bf16x8*>(base + aoff + pl * Bf3D::PLANE + tm * 1024);
    };
    const auto ldB = [&](const char* base, int pl, bfrag& d) {
#pragma unroll
        for (int tn = 0; tn < Cfg::TN; ++tn) d[tn] = *reinterpret_cast<const bf16x8*>(base + boff + pl * Bf3D::PLANE + tn * 1024);
    };
    const auto mm = [&](const afrag& a, const bfrag& b) {
#ifdef SCFGP_DIAG_DMA_NOMFMA                                    // timing diagnostic only: operand traffic and fragment reads alone
        asm volatile("" :: "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]));
        return;
#endif
#pragma unroll
        for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < Cfg::TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
    };

    issue(0);
    if (nst > 1) issue(1);
    if (nst > 2) issue(2);
    if (nst > 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (nst > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    afrag a_h, a_m, a_l;
    bfrag b_h, b_m, b_l, n_h, n_m;
    ldA(smem, 2, a_l); ldA(smem, 1, a_m); ldB(smem, 0, b_h); ldB(smem, 1, b_m);
    int slot = 0;
    for (int s = 0; s + 1 < nst; ++s) {                         // the last stage is peeled: no branch around the fragment reads
        const char* cur = smem + slot * Bf3D::STAGE;
        const int slot1 = slot == 2 ? 0 : slot + 1;
        ldA(cur, 0, a_h); ldB(cur, 2, b_l);
        __builtin_amdgcn_sched_barrier(0);                      // the scheduler would sink the reads to their first use
        mm(a_l, b_h);
        mm(a_m, b_h); mm(a_m, b_m);
        // fragment reads of slot `slot` are complete (lgkmcnt) before anybody's DMA may overwrite it
        if (s + 2 < nst) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + 3 < nst) issue(slot);
        const char* nxt = smem + slot1 * Bf3D::STAGE;
        ldA(nxt, 2, a_l); ldA(nxt, 1, a_m); ldB(nxt, 0, n_h); ldB(nxt, 1, n_m);
        __builtin_amdgcn_sched_barrier(0);
        mm(a_h, b_l); mm(a_h, b_m); mm(a_h, b_h);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tn = 0; tn < Cfg::TN; ++tn) { b_h[tn] = n_h[tn]; b_m[tn] = n_m[tn]; }
        slot = slot1;
    }
    {
        const char* cur = smem + slot * Bf3D::STAGE;
        ldA(cur, 0, a_h); ldB(cur, 2, b_l);
        mm(a_l, b_h);
        mm(a_m, b_h); mm(a_m, b_m);
        mm(a_h, b_l); mm(a_h, b_m); mm(a_h, b_h);
    }
    __syncthreads();
}
