/*
 * pt_device.h — device-side types and arithmetic for the gfx950 kernels.
 *
 * Everything that decides a result bit is spelled out here: vector operations
 * keep the operation order of the reference's GLM 0.9.6.3 build
 * (dot = x*x' + y*y' + z*z' left to right, normalize = v * (1/sqrt(dot)),
 * mat4*vec4 = (c0*x + c1*y) + (c2*z + c3*w), min/max as compare+select), the
 * transcendental functions are fixed polynomial sequences instead of OCML calls,
 * and the translation unit is compiled with -ffp-contract=off so no FMA is formed.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_DEV __device__ __forceinline__

namespace ptd {

/* ------------------------------------------------------------------ vectors */
struct F2 { float x, y; };
struct F3 { float x, y, z; };
struct F4 { float x, y, z, w; };

PT_DEV F3 f3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_DEV F3 f3(float s) { return f3(s, s, s); }
PT_DEV F3 f3(const float* p) { return f3(p[0], p[1], p[2]); }
PT_DEV F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_DEV F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_DEV F3 operator*(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_DEV F3 operator/(F3 a, F3 b) { return f3(a.x / b.x, a.y / b.y, a.z / b.z); }
PT_DEV F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
PT_DEV F3 operator*(float s, F3 a) { return f3(s * a.x, s * a.y, s * a.z); }
PT_DEV F3 operator/(F3 a, float s) { return f3(a.x / s, a.y / s, a.z / s); }
PT_DEV F3 operator/(float s, F3 a) { return f3(s / a.x, s / a.y, s / a.z); }
PT_DEV F3 operator+(F3 a, float s) { return f3(a.x + s, a.y + s, a.z + s); }
PT_DEV F3 operator+(float s, F3 a) { return f3(s + a.x, s + a.y, s + a.z); }
PT_DEV F3 operator-(F3 a) { return f3(-a.x, -a.y, -a.z); }
PT_DEV F4 operator*(F4 a, float s) { F4 r; r.x = a.x * s; r.y = a.y * s; r.z = a.z * s; r.w = a.w * s; return r; }
PT_DEV F4 operator+(F4 a, F4 b) { F4 r; r.x = a.x + b.x; r.y = a.y + b.y; r.z = a.z + b.z; r.w = a.w + b.w; return r; }
PT_DEV float comp(F3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

/* compare + select, never v_min/v_max: a NaN in x yields y (GLM's x<y?x:y / x>y?x:y) */
PT_DEV float selMin(float x, float y) { return x < y ? x : y; }
PT_DEV float selMax(float x, float y) { return x > y ? x : y; }
PT_DEV float selClamp(float x, float lo, float hi) { return selMin(selMax(x, lo), hi); }
/* fminf/fmaxf semantics with an explicit NaN rule: a NaN operand loses */
PT_DEV float nanMin(float a, float b) { return (b != b) ? a : ((a != a) ? b : (a < b ? a : b)); }
PT_DEV float nanMax(float a, float b) { return (b != b) ? a : ((a != a) ? b : (a > b ? a : b)); }

PT_DEV float dot(F3 a, F3 b) { F3 t = a * b; return t.x + t.y + t.z; }
PT_DEV F3 cross(F3 x, F3 y) { return f3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
PT_DEV float length(F3 v) { return __builtin_sqrtf(dot(v, v)); }
PT_DEV F3 normalize(F3 v) { return v * (1.0f / __builtin_sqrtf(dot(v, v))); }
PT_DEV F3 reflect(F3 I, F3 N) { return I - N * dot(N, I) * 2.0f; }
PT_DEV F3 mix(F3 x, F3 y, float a) { return x + a * (y - x); }
PT_DEV F3 vmin(F3 a, F3 b) { return f3(selMin(a.x, b.x), selMin(a.y, b.y), selMin(a.z, b.z)); }
PT_DEV F3 vmax(F3 a, F3 b) { return f3(selMax(a.x, b.x), selMax(a.y, b.y), selMax(a.z, b.z)); }
PT_DEV F3 vclamp(F3 v, F3 lo, F3 hi) { return vmin(vmax(v, lo), hi); }
PT_DEV F3 vsqrt(F3 v) { return f3(__builtin_sqrtf(v.x), __builtin_sqrtf(v.y), __builtin_sqrtf(v.z)); }

/* column-major 4x4 times (v, w) */
PT_DEV F3 mulMV(const float* m, F3 v, float w)
{
    float r[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float mul0 = m[0 + i] * v.x;
        float mul1 = m[4 + i] * v.y;
        float add0 = mul0 + mul1;
        float mul2 = m[8 + i] * v.z;
        float mul3 = m[12 + i] * w;
        float add1 = mul2 + mul3;
        r[i] = add0 + add1;
    }
    return f3(r[0], r[1], r[2]);
}

/* ------------------------------------------------------- fixed-sequence math */
namespace dm {
PT_DEV uint32_t f2u(float f) { return __float_as_uint(f); }
PT_DEV float u2f(uint32_t u) { return __uint_as_float(u); }

constexpr float PIO2_1 = 1.5703125f;
constexpr float PIO2_2 = 4.837512969970703125e-4f;
constexpr float PIO2_3 = 7.54978995489188216e-8f;
constexpr float TWO_OVER_PI = 0.63661977236758134308f;
constexpr float PI_F = 3.14159265358979323846f;
constexpr float PIO2_F = 1.57079632679489661923f;
constexpr float PIO4_F = 0.78539816339744830962f;

PT_DEV float sinPoly(float r)
{
    float z = r * r;
    float p = (-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f;
    return r + r * z * p;
}
PT_DEV float cosPoly(float r)
{
    float z = r * r;
    float p = (2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f;
    return (1.0f - 0.5f * z) + z * z * p;
}
PT_DEV float reduce(float x, int& q)
{
    float kf = __builtin_floorf(x * TWO_OVER_PI + 0.5f);
    float r = ((x - kf * PIO2_1) - kf * PIO2_2) - kf * PIO2_3;
    q = ((int)kf) & 3;
    return r;
}
PT_DEV float sin(float x)
{
    int q;
    float r = reduce(x, q);
    float s = (q & 1) ? cosPoly(r) : sinPoly(r);
    return (q & 2) ? -s : s;
}
PT_DEV float cos(float x)
{
    int q;
    float r = reduce(x, q);
    float c = (q & 1) ? sinPoly(r) : cosPoly(r);
    return ((q + 1) & 2) ? -c : c;
}
PT_DEV float asinCore(float x)
{
    float z = x * x;
    float p = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z
               + 1.6666752422e-1f);
    return p * z * x + x;
}
PT_DEV float acos(float x)
{
    if (x < -0.5f) {
        float s = __builtin_sqrtf(0.5f * (1.0f + x));
        return PI_F - 2.0f * asinCore(s);
    }
    if (x > 0.5f) {
        float s = __builtin_sqrtf(0.5f * (1.0f - x));
        return 2.0f * asinCore(s);
    }
    return PIO2_F - asinCore(x);
}
PT_DEV float atanPos(float x)
{
    float y;
    if (x > 2.414213562373095f) { y = PIO2_F; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = PIO4_F; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    float z = x * x;
    float p = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f);
    return y + (p * z * x + x);
}
PT_DEV float atan2(float y, float x)
{
    if (x != x || y != y) return x + y;
    if (y == 0.0f) {
        bool xneg = (f2u(x) >> 31) != 0;
        float r = xneg ? PI_F : 0.0f;
        return (f2u(y) >> 31) ? -r : r;
    }
    if (x == 0.0f) return y > 0.0f ? PIO2_F : -PIO2_F;
    float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    float a = atanPos(ay / ax);
    if (x < 0.0f) a = PI_F - a;
    return y < 0.0f ? -a : a;
}
PT_DEV float exp(float x)
{
    if (x != x) return x;
    if (x > 88.72283905206835f) return u2f(0x7f800000u);
    if (x < -87.33654475055310f) return 0.0f;
    float n = __builtin_floorf(1.44269504088896341f * x + 0.5f);
    float r = x - n * 0.693359375f;
    r = r - n * -2.12194440e-4f;
    float z = r * r;
    float p = ((((1.9875691500e-4f * r + 1.3981999507e-3f) * r + 8.3334519073e-3f) * r + 4.1665795894e-2f) * r
               + 1.6666665459e-1f) * r + 5.0000001201e-1f;
    float y = p * z + r + 1.0f;
    int e = (int)n;
    if (e > 127) { y = y * 2.0f; e -= 1; }
    return y * u2f((uint32_t)(e + 127) << 23);
}
PT_DEV float log(float x)
{
    if (x != x) return x;
    if (x <= 0.0f) return x == 0.0f ? -u2f(0x7f800000u) : u2f(0x7fc00000u);
    uint32_t u = f2u(x);
    if (u == 0x7f800000u) return x;
    int e = (int)(u >> 23) - 126;
    float m = u2f((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = (m + m) - 1.0f; }
    else m = m - 1.0f;
    float z = m * m;
    float p = ((((((((7.0376836292e-2f * m - 1.1514610310e-1f) * m + 1.1676998740e-1f) * m - 1.2420140846e-1f) * m
                   + 1.4249322787e-1f) * m - 1.6668057665e-1f) * m + 2.0000714765e-1f) * m - 2.4999993993e-1f) * m
               + 3.3333331174e-1f);
    float y = m * z * p;
    float fe = (float)e;
    y = y + -2.12194440e-4f * fe;
    y = y + -0.5f * z;
    float r = m + y;
    return r + 0.693359375f * fe;
}
PT_DEV float pow5(float x)
{
    float x2 = x * x;
    return (x2 * x2) * x;
}
} // namespace dm

/* ---------------------------------------------------------------------- RNG
 * minstd_rand seeded per (iteration, queue slot, remaining bounces), the stream the
 * reference builds with makeSeededRandomEngine (src/pathtrace.cu:138-143) on top of
 * thrust::default_random_engine + uniform_real_distribution<float>(0,1). */
PT_DEV uint32_t hashU32(uint32_t a)
{
    a = (a + 0x7ed55d16u) + (a << 12);
    a = (a ^ 0xc761c23cu) ^ (a >> 19);
    a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9);
    a = (a + 0xfd7046c5u) + (a << 3);
    a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}
struct Lcg {
    uint32_t x;
    PT_DEV void seed(int iter, int index, int depth)
    {
        uint32_t h = hashU32((1u << 31) | ((uint32_t)depth << 22) | (uint32_t)iter) ^ hashU32((uint32_t)index);
        x = h % 2147483647u;
        if (x == 0) x = 1;
    }
    PT_DEV float u01()
    {
        /* x * 48271 mod (2^31 - 1) without a 64-bit division: fold the high part (2^31 == 1 mod m) */
        uint64_t p = (uint64_t)x * 48271ull;
        uint32_t lo = (uint32_t)(p & 0x7fffffffull), hi = (uint32_t)(p >> 31);
        uint32_t s = lo + hi;
        if (s >= 2147483647u) s -= 2147483647u;
        x = s;
        float r = (float)(x - 1u);
        return r / 2147483648.0f;
    }
};

} // namespace ptd
