// Tile shapes shared by the row-sweep kernels (gram.hip, apply.hip).
// Measured on MI355X (N = 5e5..1e6, K = 2112; tests/gpu_tune.py, profiles/r01_tuning.md):
//   * every MFMA kernel wants two workgroups per CU: 8-wave workgroups within a 128-VGPR budget (the compiler
//     takes up to 256 unless told: amdgpu_waves_per_eu on the Gram kernel); fp64 apply: one 16-wave workgroup;
//   * 64 x 64 wave tiles and one barrier per 64 MFMAs per wave where the registers allow it (apply: BK = 16 on a
//     256 x 128 tile; fp32 Gram: the same tall tile, BK = 32 on its 128 x 128 tiles);
//   * the 32x32x2 fp32 MFMA shape, 192- / 256-wide square tiles and 16-wave 256 x 256 register-staged tiles are slower.
#pragma once
#include "tile_engine.h"

#ifndef SCFGP_BK
#define SCFGP_BK 16
#endif
#ifndef SCFGP_GRAM_BK_F32
#define SCFGP_GRAM_BK_F32 32       // one barrier per 64 MFMAs per wave; fits 128 VGPRs only in fp32
#endif
template <typename T> struct Tune;
template <> struct Tune<float>  {
    static constexpr int MS = 16, GRAM_WGM = 4, GRAM_WGN = 2, GRAM_BK = SCFGP_GRAM_BK_F32, APPLY_BM = 256, APPLY_BN = 128, APPLY_WGM = 4;
    static constexpr int apply_wgn(int bn) { return bn >= 256 ? 4 : 2; }
};
template <> struct Tune<double> {
    static constexpr int MS = 16, GRAM_WGM = 4, GRAM_WGN = 2, GRAM_BK = SCFGP_BK, APPLY_BM = 256, APPLY_BN = 128, APPLY_WGM = 4;
    static constexpr int apply_wgn(int) { return 4; }
};

#define SMEM_DECL extern __shared__ __attribute__((aligned(16))) char smem_raw[]
