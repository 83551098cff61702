// VALU issue-rate probe for gfx950: how many cycles a wave64 v_fma_f32 / v_pk_fma_f32 / v_min_f32 occupies a SIMD.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) probe(float* out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = { a0, a1 }, p1 = { a2, a3 }, p2 = { a4, a5 }, p3 = { a6, a7 }, p4 = { a1, a0 }, p5 = { a3, a2 }, p6 = { a5, a4 }, p7 = { a7, a6 };
    const float m = 1.0000001f, c = 1e-9f;
    const v2f pm = { m, m }, pc = { c, c };
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
#define FMA1(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c));
                FMA1(a0) FMA1(a1) FMA1(a2) FMA1(a3) FMA1(a4) FMA1(a5) FMA1(a6) FMA1(a7)
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
#define FMA2(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(pm), "v"(pc));
                FMA2(p0) FMA2(p1) FMA2(p2) FMA2(p3) FMA2(p4) FMA2(p5) FMA2(p6) FMA2(p7)
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                a0 = __builtin_fminf(a0, a1 + c); a1 = __builtin_fminf(a1, a2 + c); a2 = __builtin_fminf(a2, a3 + c); a3 = __builtin_fminf(a3, a4 + c);
                a4 = __builtin_fminf(a4, a5 + c); a5 = __builtin_fminf(a5, a6 + c); a6 = __builtin_fminf(a6, a7 + c); a7 = __builtin_fminf(a7, a0 + c);
            }
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
    if (r == 12345.678f) out[0] = r;
}

template <int MODE>
static void run(const char* name, int wavesPerSimd, int instrPerIter)
{
    float* out; hipMalloc(&out, 4);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * wavesPerSimd;            // 256 threads = 4 waves = one per SIMD
    const int iters = 20000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<MODE><<<blocks, 256>>>(out, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<MODE><<<blocks, 256>>>(out, iters, 1.0f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double clk = prop.clockRate * 1e3;          // Hz
    const double instrPerSimd = (double)iters * instrPerIter * wavesPerSimd;
    printf("%-14s waves/SIMD %d: %.3f ms, %.2f cycles per wave-instruction per SIMD (clock %.0f MHz)\n", name, wavesPerSimd, ms,
           ms * 1e-3 * clk / instrPerSimd, clk / 1e6);
    hipFree(out);
}

int main()
{
    for (int w : { 1, 2, 4 }) {
        run<0>("v_fma_f32", w, 64);
        run<1>("v_pk_fma_f32", w, 64);
        run<2>("v_add+v_min", w, 128);
    }
    return 0;
}
