/*
 * pt_denoise.hip — guide buffers and an open denoiser for gfx950.
 *
 * k_gbuffer is captureNormalsAndAlbedo (src/pathtrace.cu:94-135): per pixel the normalised first-hit normal and
 * the hit material's base colour, zeros where the camera ray leaves the scene.
 *
 * The reference hands its accumulated image and those two buffers to the OptiX AI denoiser
 * (src/pathtrace.cu:2885-2910, src/optixDenoiser.cpp:127-257), which is closed.  k_atrous is an open stand-in with
 * the same inputs and output: the edge-avoiding a-trous wavelet filter (five passes of a 5x5 B3-spline kernel with
 * holes of 1, 2, 4, ... pixels; a tap's weight falls off with its colour, normal and albedo distance from the
 * centre, the colour tolerance halving every pass).  Its results are its own — they are checked against a CPU
 * statement of the same filter, not against the reference.
 *
 * One thread per pixel, 25 taps through the cache hierarchy; bandwidth bound and far from the hot path.
 */
#include "pt_device.h"
#include "pt_scene_dev.h"

#include <hip/hip_runtime.h>

namespace ptd {

__global__ void __launch_bounds__(256) k_gbuffer(PathState st, const PtMaterial* materials, int n, float* normals, float* albedo)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 ha = st.hitA[i], hb = st.hitB[i];
    F3 nn = f3(0.0f), al = f3(0.0f);
    if (ha.x > 0.0f) {
        nn = normalize(f3(ha.y, ha.z, ha.w));
        const int mat = __float_as_int(hb.x);
        if (mat >= 0) { const PtMaterial& m = materials[mat]; al = f3(m.color[0], m.color[1], m.color[2]); }
    }
    normals[3 * (size_t)i] = nn.x; normals[3 * (size_t)i + 1] = nn.y; normals[3 * (size_t)i + 2] = nn.z;
    albedo[3 * (size_t)i] = al.x; albedo[3 * (size_t)i + 1] = al.y; albedo[3 * (size_t)i + 2] = al.z;
}

/* sum / samples -> colour */
__global__ void __launch_bounds__(256) k_average(const float* sum, float samples, size_t n3, float* out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n3) out[i] = sum[i] / samples;
}

__global__ void __launch_bounds__(256) k_atrous(const float* in, const float* normals, const float* albedo, int W, int H, int step,
                                                float invSigmaC2, float invSigmaN2, float invSigmaA2, float* out)
{
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= W || y >= H) return;
    const float kern[5] = { 1.0f / 16, 1.0f / 4, 3.0f / 8, 1.0f / 4, 1.0f / 16 };
    const size_t c = 3 * ((size_t)y * W + x);
    const F3 c0 = f3(in[c], in[c + 1], in[c + 2]);
    const F3 n0 = normals ? f3(normals[c], normals[c + 1], normals[c + 2]) : f3(0.0f);
    const F3 a0 = albedo ? f3(albedo[c], albedo[c + 1], albedo[c + 2]) : f3(0.0f);
    F3 acc = f3(0.0f);
    float wsum = 0.0f;
    for (int dy = -2; dy <= 2; ++dy) {
        const int yy = y + dy * step;
        if (yy < 0 || yy >= H) continue;
        for (int dx = -2; dx <= 2; ++dx) {
            const int xx = x + dx * step;
            if (xx < 0 || xx >= W) continue;
            const size_t t = 3 * ((size_t)yy * W + xx);
            const F3 ct = f3(in[t], in[t + 1], in[t + 2]);
            F3 d = ct - c0;
            float e = dot(d, d) * invSigmaC2;
            if (normals) { d = f3(normals[t], normals[t + 1], normals[t + 2]) - n0; e += dot(d, d) * invSigmaN2; }
            if (albedo) { d = f3(albedo[t], albedo[t + 1], albedo[t + 2]) - a0; e += dot(d, d) * invSigmaA2; }
            const float w = kern[dy + 2] * kern[dx + 2] * __builtin_expf(-e);
            acc = acc + ct * w;
            wsum += w;
        }
    }
    const float inv = 1.0f / wsum;          /* the centre tap keeps wsum >= 9/64 */
    out[c] = acc.x * inv; out[c + 1] = acc.y * inv; out[c + 2] = acc.z * inv;
}

} // namespace ptd

using namespace ptd;

void ptl_gbuffer(hipStream_t s, PathState st, const PtMaterial* materials, int n, float* normals, float* albedo)
{
    hipLaunchKernelGGL(k_gbuffer, dim3((n + 255) / 256), dim3(256), 0, s, st, materials, n, normals, albedo);
}
void ptl_average(hipStream_t s, const float* sum, float samples, size_t n3, float* out)
{
    hipLaunchKernelGGL(k_average, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, s, sum, samples, n3, out);
}
void ptl_atrous(hipStream_t s, const float* in, const float* normals, const float* albedo, int W, int H, int step,
                float invSigmaC2, float invSigmaN2, float invSigmaA2, float* out)
{
    hipLaunchKernelGGL(k_atrous, dim3((W + 15) / 16, (H + 15) / 16), dim3(256), 0, s, in, normals, albedo, W, H, step, invSigmaC2, invSigmaN2, invSigmaA2, out);
}
