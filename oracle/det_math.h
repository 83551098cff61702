/*
 * det_math.h — TEST INFRASTRUCTURE (part of the CPU oracle, never shipped).
 *
 * Single-precision sin/cos/acos/atan2/exp/log/pow5 written with nothing but
 * IEEE-754 +,-,*,/ and sqrt on floats plus exact integer/bit steps, so that a
 * CPU build without FMA contraction and a GPU build with -ffp-contract=off
 * return the same bits for the same input.  The reference itself calls the
 * platform's libm/libdevice (sinf, cosf, acosf, atan2f, expf, logf, powf:
 * src/pathtrace.cu:1039-1040,1097-1106,1164-1168,1308,1586-1596,
 * src/interactions.cu:13-15,43-44) whose last bits differ between platforms;
 * the oracle therefore has two arithmetic modes: "libm" (what a host build of
 * the reference computes) and "det" (this file; what the GPU kernels compute).
 * Polynomials are the classic Cephes single-precision minimax sets.
 */
#ifndef ORACLE_DET_MATH_H
#define ORACLE_DET_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

namespace detm {

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* Cody-Waite split of pi/2: kf*C1 is exact for |kf| < 2^15. */
static const float PIO2_1 = 1.5703125f;
static const float PIO2_2 = 4.837512969970703125e-4f;
static const float PIO2_3 = 7.54978995489188216e-8f;
static const float TWO_OVER_PI = 0.63661977236758134308f;

static inline float sin_poly(float r)
{
    float z = r * r;
    float p = (-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f;
    return r + r * z * p;
}
static inline float cos_poly(float r)
{
    float z = r * r;
    float p = (2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f;
    return (1.0f - 0.5f * z) + z * z * p;
}
/* Reduce x to r in [-pi/4, pi/4] and quadrant q = k mod 4 (k = nearest integer to x*2/pi). */
static inline float reduce_pio2(float x, int* q)
{
    float kf = floorf(x * TWO_OVER_PI + 0.5f);
    float r = ((x - kf * PIO2_1) - kf * PIO2_2) - kf * PIO2_3;
    *q = ((int)kf) & 3;
    return r;
}
static inline float sinf_det(float x)
{
    int q;
    float r = reduce_pio2(x, &q);
    float s = (q & 1) ? cos_poly(r) : sin_poly(r);
    return (q & 2) ? -s : s;
}
static inline float cosf_det(float x)
{
    int q;
    float r = reduce_pio2(x, &q);
    float c = (q & 1) ? sin_poly(r) : cos_poly(r);
    return ((q + 1) & 2) ? -c : c;
}

static inline float asin_core(float x) /* |x| <= 0.5 */
{
    float z = x * x;
    float p = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z
                + 7.4953002686e-2f) * z + 1.6666752422e-1f);
    return p * z * x + x;
}
static const float PI_F = 3.14159265358979323846f;
static const float PIO2_F = 1.57079632679489661923f;
static const float PIO4_F = 0.78539816339744830962f;

static inline float acosf_det(float x)
{
    if (x < -0.5f) {
        float s = sqrtf(0.5f * (1.0f + x));
        return PI_F - 2.0f * asin_core(s);
    }
    if (x > 0.5f) {
        float s = sqrtf(0.5f * (1.0f - x));
        return 2.0f * asin_core(s);
    }
    return PIO2_F - asin_core(x);
}

static inline float atan_pos(float x) /* x >= 0 */
{
    float y;
    if (x > 2.414213562373095f) { y = PIO2_F; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = PIO4_F; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    float z = x * x;
    float p = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z
               - 3.33329491539e-1f);
    return y + (p * z * x + x);
}
static inline float atan2f_det(float y, float x)
{
    if (x != x || y != y) return x + y;
    if (y == 0.0f) {
        bool xneg = (f2u(x) >> 31) != 0;
        float r = xneg ? PI_F : 0.0f;
        return (f2u(y) >> 31) ? -r : r;
    }
    if (x == 0.0f) return y > 0.0f ? PIO2_F : -PIO2_F;
    float ax = fabsf(x), ay = fabsf(y);
    float a = atan_pos(ay / ax);
    if (x < 0.0f) a = PI_F - a;
    return y < 0.0f ? -a : a;
}

static inline float expf_det(float x)
{
    if (x != x) return x;
    if (x > 88.72283905206835f) return u2f(0x7f800000u);
    if (x < -87.33654475055310f) return 0.0f;
    float n = floorf(1.44269504088896341f * x + 0.5f);
    float r = x - n * 0.693359375f;
    r = r - n * -2.12194440e-4f;
    float z = r * r;
    float p = ((((1.9875691500e-4f * r + 1.3981999507e-3f) * r + 8.3334519073e-3f) * r
                + 4.1665795894e-2f) * r + 1.6666665459e-1f) * r + 5.0000001201e-1f;
    float y = p * z + r + 1.0f;
    int e = (int)n;                     /* in [-126, 128] */
    if (e > 127) { y = y * 2.0f; e -= 1; }
    return y * u2f((uint32_t)(e + 127) << 23);
}

static inline float logf_det(float x)
{
    if (x != x) return x;
    if (x <= 0.0f) return x == 0.0f ? -u2f(0x7f800000u) : u2f(0x7fc00000u);
    uint32_t u = f2u(x);
    if (u == 0x7f800000u) return x;
    int e = (int)(u >> 23) - 126;       /* frexp: x = m * 2^e, m in [0.5,1); normals only */
    float m = u2f((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = (m + m) - 1.0f; }
    else m = m - 1.0f;
    float z = m * m;
    float p = ((((((((7.0376836292e-2f * m - 1.1514610310e-1f) * m + 1.1676998740e-1f) * m
                    - 1.2420140846e-1f) * m + 1.4249322787e-1f) * m - 1.6668057665e-1f) * m
                 + 2.0000714765e-1f) * m - 2.4999993993e-1f) * m + 3.3333331174e-1f);
    float y = m * z * p;
    float fe = (float)e;
    y = y + -2.12194440e-4f * fe;
    y = y + -0.5f * z;
    float r = m + y;
    return r + 0.693359375f * fe;
}

static inline float pow5_det(float x)
{
    float x2 = x * x;
    return (x2 * x2) * x;
}

} // namespace detm
#endif
