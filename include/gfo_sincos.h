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
#include "gfo_sincos_coef.h"

/* the constants: literals, or -- where the includer defines GFO_SINCOS_TABLE (the HIP kernel) -- entries of a table in memory */
#ifdef GFO_SINCOS_TABLE
#define GFO_SC(i, lit) (GFO_SINCOS_TABLE[i])
#else
#define GFO_SC(i, lit) (lit)
#endif
/* fma(a, b, c) with a constant addend c.  In the HIP kernel c sits in a scalar register pair (loaded from the table) and the
 * three-operand v_fma_f64 reads it there; left to itself the compiler picks the accumulating form, which wants the addend in
 * the destination and pays two v_mov per constant.  Same IEEE operation either way. */
#if defined(__HIP_DEVICE_COMPILE__) && defined(GFO_SINCOS_TABLE)
static __device__ __forceinline__ double gfo_fma_c(double a, double b, double c)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}
#else
#define gfo_fma_c(a, b, c) fma((a), (b), (c))
#endif

#ifdef __HIPCC__
#define GFO_HD __host__ __device__ static inline
#else
#define GFO_HD static inline
#endif

/* sin and cos of t (radians, |t| < ~1e4), each rounded once from double to float. */
GFO_HD void gfo_sincosf(float t, float* s_out, float* c_out)
{
    const double x = (double)t;
    const double two_over_pi = GFO_SC(0, GFO_SC_2OPI);
    const double pio2_hi = GFO_SC(1, GFO_SC_PIO2_HI);
    const double pio2_lo = GFO_SC(2, GFO_SC_PIO2_LO);
    const double kd = rint(x * two_over_pi);
    const int k = (int)kd;
    double r = fma(-kd, pio2_hi, x);
    r = fma(-kd, pio2_lo, r);
    const double z = r * r;
    /* minimax polynomials on [-pi/4, pi/4] (the classic fdlibm kernel coefficients) */
    double ps = GFO_SC(3, GFO_SC_S6);
    ps = gfo_fma_c(ps, z, GFO_SC(4, GFO_SC_S5));
    ps = gfo_fma_c(ps, z, GFO_SC(5, GFO_SC_S4));
    ps = gfo_fma_c(ps, z, GFO_SC(6, GFO_SC_S3));
    ps = gfo_fma_c(ps, z, GFO_SC(7, GFO_SC_S2));
    ps = gfo_fma_c(ps, z, GFO_SC(8, GFO_SC_S1));
    const double sn = fma(r * z, ps, r);
    double pc = GFO_SC(9, GFO_SC_C6);
    pc = gfo_fma_c(pc, z, GFO_SC(10, GFO_SC_C5));
    pc = gfo_fma_c(pc, z, GFO_SC(11, GFO_SC_C4));
    pc = gfo_fma_c(pc, z, GFO_SC(12, GFO_SC_C3));
    pc = gfo_fma_c(pc, z, GFO_SC(13, GFO_SC_C2));
    pc = gfo_fma_c(pc, z, GFO_SC(14, GFO_SC_C1));
    const double cs = fma(z * z, pc, fma(z, -0.5, 1.0));
    /* quadrant: (s, c) = (sn, cs), (cs, -sn), (-sn, -cs), (-cs, sn).  Rounding commutes with negation and with the choice, so
     * both are done on the rounded floats (two conversions, 32-bit selects) instead of on the doubles. */
    {
        const float snf = (float)sn, csf = (float)cs;
        float s = (k & 1) ? csf : snf;
        float c = (k & 1) ? snf : csf;
        if (k & 2) s = -s;
        if ((k + 1) & 2) c = -c;
        *s_out = s;
        *c_out = c;
    }
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
#if defined(GFO_OCV_ATAN_FMA) && GFO_OCV_ATAN_FMA == 1
    const float pa = fmaf(fmaf(fmaf(p7, c2, p5), c2, p3), c2, p1) * c;   /* a build of OpenCV whose v_fma fuses the Horner steps */
#else
    const float pa = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
#endif
    float a = steep ? 90.f - pa : pa;
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

#endif /* GFO_SINCOS_H */
