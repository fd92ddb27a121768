// Glue of the relight / eval frame (relighting.py:114-170 -> gaussian_renderer/neilf.py:74-209, is_training=False) for
// gfx950: the two streaming passes either side of the shading integral and the rasterizer that the reference runs as a few
// dozen PyTorch elementwise launches per frame.
//
//   relight_pack_features_kernel   the S=28 eval feature row (neilf.py:115-130): depth, depth^2, pbr, normal, base colour,
//                                  roughness, diffuse light, specular, incident / local / global light means, mean
//                                  visibility -- from the shading kernel's 19 outputs, one pass
//   relight_compose_kernel         per pixel: camera ray in world space (Camera.get_world_directions, cameras.py:79-91),
//                                  lat-long lookup of the HDR environment (EnvLight.direct_light, envmap.py:35-53:
//                                  arccos / atan2, grid_sample bilinear, align_corners, zero padding, optional light
//                                  rotation) and the three composited outputs of neilf.py:203-207:
//                                    pbr_env    = srgb(pbr * opacity + (1 - opacity) * env)
//                                    render_env = image + (1 - opacity) * srgb(env)
//                                    env_only   = srgb(env)
// Parity target: the PyTorch restatement in relightable3dgaussian_amd/relight.py (frame_reference).
#include "common.hpp"
#include "r3dg_hip.h"

namespace r3dg {

__global__ void __launch_bounds__(256)
relight_pack_features_kernel(int P, const float* __restrict__ xyz, const float* __restrict__ viewmatrix,
                             const float* __restrict__ normal, const float* __restrict__ base_color,
                             const float* __restrict__ roughness, const float* __restrict__ shade_out,
                             float* __restrict__ features)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const size_t i3 = 3 * (size_t)i;
    const float depth = xyz[i3] * viewmatrix[2] + xyz[i3 + 1] * viewmatrix[6] + xyz[i3 + 2] * viewmatrix[10] +
                        viewmatrix[14];
    const float* so = shade_out + 19 * (size_t)i;      // pbr3 diffuse3 specular3 lights3 local3 global3 vis1
    float f[28];
    f[0] = depth;
    f[1] = depth * depth;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        f[2 + c] = so[c];
        f[5 + c] = normal[i3 + c];
        f[8 + c] = base_color[i3 + c];
        f[12 + c] = so[3 + c];
    }
    f[11] = roughness[i];
#pragma unroll
    for (int c = 0; c < 13; c++) f[15 + c] = so[6 + c];
    float4* out = reinterpret_cast<float4*>(features + 28 * (size_t)i);      // 112-byte rows: 16-byte aligned
#pragma unroll
    for (int q = 0; q < 7; q++) out[q] = make_float4(f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]);
}

__device__ __forceinline__ float srgb_of(float x)
{
    // rgb_to_srgb (utils/graphics_utils.py:207-213), clip=True
    // (x^(1/2.4) as v_log_f32 * y -> v_exp_f32, ~4 ulp: the library powf is ~155 instructions per channel of every pixel)
    const float p = __builtin_amdgcn_exp2f((1.0f / 2.4f) * __builtin_amdgcn_logf(fmaxf(x, 0.0031308f)));
    const float y = x > 0.0031308f ? p * 1.055f - 0.055f : 12.92f * x;
    return fminf(fmaxf(y, 0.f), 1.f);
}

struct RelightCam {
    float fx, fy, cx, cy;
};

// viewmatrix: the reference's world_view_transform (W2C transposed, row-major 4x4, device memory); its upper-left 3x3 is
// the camera-to-world rotation.  tr: row-major 3x3 light rotation (dirs @ tr^T) in device memory, or NULL.
__global__ void __launch_bounds__(256)
relight_compose_kernel(int W, int H, RelightCam cam, const float* __restrict__ viewmatrix,
                       const float* __restrict__ tr, const float* __restrict__ env, int He, int We,
                       const float* __restrict__ image, const float* __restrict__ opacity,
                       const float* __restrict__ feature, const int* __restrict__ n_contrib,
                       float* __restrict__ pbr_env, float* __restrict__ render_env, float* __restrict__ env_only)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int HW = W * H;
    if (i >= HW) return;
    const int v = i / W, u = i - v * W;
    // camera ray, normalised (F.normalize eps 1e-12), rotated to world space
    float d[3] = {((float)u - cam.cx) / cam.fx, ((float)v - cam.cy) / cam.fy, 1.f};
    const float inv = 1.f / fmaxf(sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), 1e-12f);
    d[0] *= inv; d[1] *= inv; d[2] *= inv;
    float w[3];
#pragma unroll
    for (int r = 0; r < 3; r++)
        w[r] = viewmatrix[4 * r] * d[0] + viewmatrix[4 * r + 1] * d[1] + viewmatrix[4 * r + 2] * d[2];
    if (tr != nullptr) {
        float t[3];
#pragma unroll
        for (int r = 0; r < 3; r++) t[r] = w[0] * tr[3 * r] + w[1] * tr[3 * r + 1] + w[2] * tr[3 * r + 2];
        w[0] = t[0]; w[1] = t[1]; w[2] = t[2];
    }
    // lat-long lookup
    const float kPi = 3.14159265358979323846f;
    const float phi = acosf(fminf(fmaxf(w[2], -1.f), 1.f)) - 1e-6f;     // clamp: a unit vector may round to |z| > 1
    const float theta = atan2f(w[1], w[0]);
    const float qy = (phi / kPi) * 2.f - 1.f, qx = -theta / kPi;
    const float ix = (qx + 1.f) * 0.5f * (float)(We - 1), iy = (qy + 1.f) * 0.5f * (float)(He - 1);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float wx1 = ix - x0f, wy1 = iy - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    float e[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int yy = y0 + a, xx = x0 + b;
            if (xx >= 0 && xx <= We - 1 && yy >= 0 && yy <= He - 1) {
                const float wt = (a ? wy1 : 1.f - wy1) * (b ? wx1 : 1.f - wx1);
                const float* px = env + 3 * ((size_t)yy * We + xx);
                e[0] += px[0] * wt; e[1] += px[1] * wt; e[2] += px[2] * wt;
            }
        }
    const float op = opacity[i];
    const float scale = n_contrib[i] > 0 ? 1.f / fmaxf(op, 1e-5f) : 0.f;     // rendered_feature / opacity * mask
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float pbr = feature[(size_t)(2 + c) * HW + i] * scale;
        if (pbr_env) pbr_env[(size_t)c * HW + i] = srgb_of(pbr * op + (1.f - op) * e[c]);
        const float se = srgb_of(e[c]);
        if (render_env) render_env[(size_t)c * HW + i] = image[(size_t)c * HW + i] + (1.f - op) * se;
        if (env_only) env_only[(size_t)c * HW + i] = se;
    }
}

void launch_relight_pack(hipStream_t s, int P, const float* xyz, const float* viewmatrix, const float* normal,
                         const float* base_color, const float* roughness, const float* shade_out, float* features)
{
    relight_pack_features_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, xyz, viewmatrix, normal, base_color, roughness,
                                                                 shade_out, features);
    check_launch(s, false, "relight_pack_features_kernel");
}

void launch_relight_compose(hipStream_t s, int W, int H, float fx, float fy, float cx, float cy,
                            const float* viewmatrix, const float* tr, const float* env, int He, int We,
                            const float* image, const float* opacity, const float* feature, const int* n_contrib,
                            float* pbr_env, float* render_env, float* env_only)
{
    const RelightCam cam = {fx, fy, cx, cy};
    const int HW = W * H;
    relight_compose_kernel<<<(HW + 255) / 256, 256, 0, s>>>(W, H, cam, viewmatrix, tr, env, He, We, image, opacity,
                                                            feature, n_contrib, pbr_env, render_env, env_only);
    check_launch(s, false, "relight_compose_kernel");
}

}  // namespace r3dg
