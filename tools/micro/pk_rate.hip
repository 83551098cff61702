// Issue rate of packed against plain single-precision multiply-add pairs on one SIMD-filling launch.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/pk_rate.hip -o gpurun_out/pk_rate && gpurun_out/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float pk2 __attribute__((ext_vector_type(2)));
#define KEEP(x) asm volatile("" : "+v"(x))
__global__ void __launch_bounds__(256) plain(float* out, float a, float b, int n)
{
    float x[16];
    for (int k = 0; k < 16; ++k) x[k] = threadIdx.x + k;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) { x[k] = x[k] - a; KEEP(x[k]); }
#pragma unroll
        for (int k = 0; k < 16; ++k) { x[k] = x[k] * b; KEEP(x[k]); }
    }
    float s = 0;
    for (int k = 0; k < 16; ++k) s += x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) packed(float* out, float a, float b, int n)
{
    pk2 x[8];
    for (int k = 0; k < 8; ++k) x[k] = (pk2){ (float)threadIdx.x + 2 * k, (float)threadIdx.x + 2 * k + 1 };
    const pk2 pa = { a, a }, pb = { b, b };
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { x[k] = x[k] - pa; KEEP(x[k]); }
#pragma unroll
        for (int k = 0; k < 8; ++k) { x[k] = x[k] * pb; KEEP(x[k]); }
    }
    float s = 0;
    for (int k = 0; k < 8; ++k) s += x[k].x + x[k].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    float* d; hipMalloc(&d, 1024 * 256 * 8 * sizeof(float));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int n = 20000;
    for (int waves = 1; waves <= 4; waves *= 2) {
        dim3 grid(256 * 4 * waves / 4 * 1), block(256);      // 4 waves per workgroup: `waves` waves per SIMD
        grid.x = 256 * waves;
        for (int k = 0; k < 2; ++k) {
            float ms = 0;
            hipLaunchKernelGGL(k ? packed : plain, grid, block, 0, 0, d, 0.5f, 1.0001f, 10);
            hipEventRecord(a);
            hipLaunchKernelGGL(k ? packed : plain, grid, block, 0, 0, d, 0.5f, 1.0001f, n);
            hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
            // per lane: 32 operations per loop iteration
            double ops = (double)grid.x * 256 * 32.0 * n;
            printf("%s, %d wave(s) per SIMD: %.3f ms, %.1f G lane-operations/s\n", k ? "packed" : "plain ", waves, ms, ops / ms / 1e6);
        }
    }
    return 0;
}
