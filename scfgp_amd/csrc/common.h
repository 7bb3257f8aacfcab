// Shared declarations for the scfgp HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));


static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// device-side scalars derived from (a,b,c) once per parameter update
struct Scal {
    double a, b, c;
    double s;        // e^b * sqrt(2/M)          SCFGP.py:98,102
    double lam;      // e^{2a} + 1e-6            SCFGP.py:93,98,105
    double kappa;    // log(1+e^c)               SCFGP.py:103
    double em2a;     // e^{-2a}
    double e2a;      // e^{2a}
    double sigc;     // sigmoid(c) = d kappa / dc
};

// slots of the small scalar reduction vector kept on the device
enum {
    R_YY = 0, R_LOGDET, R_T2, R_KBAR, R_BBAR, R_TRABAR, R_GTALPHA, R_PEN, R_COST, R_FLAG,
    R_LMIN2, R_LMAX2, R_BMAX,      // min / max of L_ii^2 and max_j (A^-1)_jj of the last factorisation (condition estimate)
    R_TRAG, R_UTG,                 // tr(Abar G) and ut^T (Phi^T y): the K x K part of bbar (kstage_bbar)
    R_COUNT
};

// Status word of a row-sharded evaluation: slots of the 8 scalars that close every exchange buffer ([0..3] carry row sums: y^T y;
// T2, kbar, sum q v, sum p mu; nothing in exchange 3).  Every rank writes its own 0 / 1 before the sum over ranks, so the summed slots COUNT ranks and every
// rank reads the same numbers (include/scfgp_hip.h, "ranks decide together"):
//   XS_RAN1  ranks whose pass 1 ran at precision level >= 1 (fp64 Gram)          exchange 1
//   XS_CAP1  ranks that cannot reach level 1 (its row buffer was refused)        exchange 1
//   XS_CAP2  ranks that cannot reach level 2                                     exchange 1
//   XS_FAIL  ranks that could not compute this stage (scfgp_fail_stage)          exchanges 1, 2, 3
enum { XS_RAN1 = 4, XS_CAP1 = 5, XS_CAP2 = 6, XS_FAIL = 7 };

// Dynamic LDS above 64 KiB needs an explicit opt-in per kernel on HIP.
extern thread_local bool g_scfgp_capturing;      // set while a hipGraph is being captured (scfgp_api.hip)
template <typename Kern>
static inline void allow_big_lds(Kern kernel, int bytes) {
    if (bytes > 64 * 1024 && !g_scfgp_capturing)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
