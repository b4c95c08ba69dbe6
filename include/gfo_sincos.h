/* gfo_sincos.h -- one deterministic sin/cos shared by the CPU oracle and the HIP kernels.
 *
 * The reference computes  a = (float)cos(angle), b = (float)sin(angle)  with the host libm
 * (src/ORBextractor.cc:110-112).  libm's cosf/sinf are not specified to the last ulp and
 * ROCm's ocml versions are a different implementation, so a descriptor kernel that called
 * the device libm could not be bit-identical to any CPU run by construction.
 *
 * This header evaluates sin and cos of a float argument in IEEE double with an explicit,
 * fully ordered sequence of fma/mul/add (every one of them exactly specified by IEEE-754,
 * so gcc on x86 and hipcc on gfx950 produce the same bits) and rounds once to float.  The
 * result is the correctly rounded float for all but a vanishing fraction of arguments
 * (double evaluation error ~1e-16 relative vs. a float half-ulp of 3e-8); the oracle test
 * suite measures the disagreement with the host libm over the whole [0,360) degree range.
 *
 * Build both sides with -ffp-contract=off so that nothing outside the explicit fma() calls
 * is fused.
 */
#ifndef GFO_SINCOS_H
#define GFO_SINCOS_H

#include <math.h>

#ifdef __HIPCC__
#define GFO_HD __host__ __device__ static inline
#else
#define GFO_HD static inline
#endif

/* sin and cos of t (radians, |t| < ~1e4), each rounded once from double to float. */
GFO_HD void gfo_sincosf(float t, float* s_out, float* c_out)
{
    const double x = (double)t;
    const double two_over_pi = 6.36619772367581382433e-01;
    const double pio2_hi = 1.57079632679489655800e+00; /* 0x3FF921FB54442D18 */
    const double pio2_lo = 6.12323399573676603587e-17; /* 0x3C91A62633145C07 */
    const double kd = rint(x * two_over_pi);
    const int k = (int)kd;
    double r = fma(-kd, pio2_hi, x);
    r = fma(-kd, pio2_lo, r);
    const double z = r * r;
    /* minimax polynomials on [-pi/4, pi/4] (the classic fdlibm kernel coefficients) */
    double ps = 1.58969099521155010221e-10;
    ps = fma(ps, z, -2.50507602534068634195e-08);
    ps = fma(ps, z, 2.75573137070700676789e-06);
    ps = fma(ps, z, -1.98412698298579493134e-04);
    ps = fma(ps, z, 8.33333333332248946124e-03);
    ps = fma(ps, z, -1.66666666666666324348e-01);
    const double sn = fma(r * z, ps, r);
    double pc = -1.13596475577881948265e-11;
    pc = fma(pc, z, 2.08757232129817482790e-09);
    pc = fma(pc, z, -2.75573143513906633035e-07);
    pc = fma(pc, z, 2.48015872894767294178e-05);
    pc = fma(pc, z, -1.38888888888741095749e-03);
    pc = fma(pc, z, 4.16666666666666019037e-02);
    const double cs = fma(z * z, pc, fma(z, -0.5, 1.0));
    double s, c;
    switch (k & 3) {
    case 0: s = sn; c = cs; break;
    case 1: s = cs; c = -sn; break;
    case 2: s = -sn; c = -cs; break;
    default: s = -cs; c = sn; break;
    }
    *s_out = (float)s;
    *c_out = (float)c;
}

/* OpenCV 3.4.x cv::fastAtan2(y, x) restated (scalar path of modules/core mathfuncs_core):
 * degree-7 odd polynomial in plain float, result in degrees in [0, 360).
 * Call site in the reference: src/ORBextractor.cc:102.  [OCV] recalled, see DESIGN.md. */
GFO_HD float gfo_fast_atan2f(float y, float x)
{
    const float scale = (float)(180.0 / 3.141592653589793238462643383279502884197169399375);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    const float eps = (float)2.2204460492503131e-016; /* (float)DBL_EPSILON */
    const float ax = fabsf(x), ay = fabsf(y);
    /* the smaller over the larger magnitude: one division serves both octant cases (the operands and the operations are those
     * of the two-branch form, so are the bits) */
    const int steep = !(ax >= ay);
    const float c = (steep ? ax : ay) / ((steep ? ay : ax) + eps);
    const float c2 = c * c;
    const float pa = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    float a = steep ? 90.f - pa : pa;
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

#endif /* GFO_SINCOS_H */
