/* gfo_sincos_coef.h -- the double constants of gfo_sincosf (include/gfo_sincos.h), listed once: the function spells them as
 * literals on the host and reads them from a table on the device (GFO_SINCOS_TABLE), and both must be the same fifteen numbers.
 *   0: 2/pi   1, 2: pi/2 high and low part   3..8: sine kernel, highest degree first   9..14: cosine kernel
 * (the classic fdlibm kernel coefficients on [-pi/4, pi/4]) */
#ifndef GFO_SINCOS_COEF_H
#define GFO_SINCOS_COEF_H
#define GFO_SC_2OPI 6.36619772367581382433e-01
#define GFO_SC_PIO2_HI 1.57079632679489655800e+00 /* 0x3FF921FB54442D18 */
#define GFO_SC_PIO2_LO 6.12323399573676603587e-17 /* 0x3C91A62633145C07 */
#define GFO_SC_S6 1.58969099521155010221e-10
#define GFO_SC_S5 -2.50507602534068634195e-08
#define GFO_SC_S4 2.75573137070700676789e-06
#define GFO_SC_S3 -1.98412698298579493134e-04
#define GFO_SC_S2 8.33333333332248946124e-03
#define GFO_SC_S1 -1.66666666666666324348e-01
#define GFO_SC_C6 -1.13596475577881948265e-11
#define GFO_SC_C5 2.08757232129817482790e-09
#define GFO_SC_C4 -2.75573143513906633035e-07
#define GFO_SC_C3 2.48015872894767294178e-05
#define GFO_SC_C2 -1.38888888888741095749e-03
#define GFO_SC_C1 4.16666666666666019037e-02
#define GFO_SINCOS_NCOEF 15
#define GFO_SINCOS_COEF_LIST GFO_SC_2OPI, GFO_SC_PIO2_HI, GFO_SC_PIO2_LO, GFO_SC_S6, GFO_SC_S5, GFO_SC_S4, GFO_SC_S3, GFO_SC_S2, GFO_SC_S1, \
                             GFO_SC_C6, GFO_SC_C5, GFO_SC_C4, GFO_SC_C3, GFO_SC_C2, GFO_SC_C1
#endif
