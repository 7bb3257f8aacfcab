// Branch-free sin/cos for the phases of the feature map (SCFGP/SCFGP.py:101 / :141).
#pragma once
#include "common.h"

// --------------------------------------------------------------------------
// sin/cos for |z| up to ~1e5 rad: two-term Cody-Waite reduction by pi/2 in fp64 (fdlibm's
// medium-argument constants: exact n*pio2_1 for |n| < 2^20) and fdlibm's kernel polynomials on
// [-pi/4, pi/4] (< 1 ulp).  Branch-free; the library sincos carries a Payne-Hanek path the
// phases of this model never need (|FF| is tens to hundreds of radians).
// --------------------------------------------------------------------------
__device__ __forceinline__ void fast_sincos(double z, double& sn, double& cs) {
    const double fn = rint(z * 6.36619772367581382433e-01);              // 2/pi
    const double r = fma(-fn, 1.57079632673412561417e+00, z);            // pio2_1 (33 bits)
    const double w = fn * 6.07710050650619224932e-11;                    // pio2_1t
    const double x = r - w;
    const double y = (r - x) - w;                                        // tail of x
    const double z2 = x * x;
    // __kernel_sin(x, y, 1)
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double v = z2 * x;
    const double rs = S2 + z2 * (S3 + z2 * (S4 + z2 * (S5 + z2 * S6)));
    const double s = x - ((z2 * (0.5 * y - v * rs) - y) - v * S1);
    // __kernel_cos(x, y)
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double rc = z2 * (C1 + z2 * (C2 + z2 * (C3 + z2 * (C4 + z2 * (C5 + z2 * C6)))));
    const double hz = 0.5 * z2;
    const double wc = 1.0 - hz;
    const double c = wc + (((1.0 - wc) - hz) + (z2 * rc - x * y));
    const int q = (int)fn & 3;
    const double ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
    sn = (q & 2) ? -ss : ss;
    cs = ((q + 1) & 2) ? -cc : cc;
}

// fp32 outputs: the same fp64 reduction (the phase itself needs it: |z| ~ 1e2 rad at 2^-24 would already be
// 1e-5), then the reduced argument in fp32 with the cephes sinf / cosf kernel polynomials on [-pi/4, pi/4]
// (~1 ulp of fp32, the precision Phi is stored in): a third of the VALU work of the fp64 kernels.
__device__ __forceinline__ void fast_sincos(double z, float& sn, float& cs) {
    const double fn = rint(z * 6.36619772367581382433e-01);
    const double r = fma(-fn, 6.07710050650619224932e-11, fma(-fn, 1.57079632673412561417e+00, z));
    const float x = (float)r, z2 = x * x;
    const float s = fmaf(x * z2, fmaf(z2, fmaf(z2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), x);
    const float c = fmaf(z2 * z2, fmaf(z2, fmaf(z2, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f),
                         fmaf(-0.5f, z2, 1.0f));
    const int q = (int)fn & 3;
    const float ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
    sn = (q & 2) ? -ss : ss;
    cs = ((q + 1) & 2) ? -cc : cc;
}
