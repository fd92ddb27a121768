// Fused glue of the stage-2 (neilf) training iteration for gfx950 -- SURVEY.md 8(f) rows n1/n2: the elementwise code the
// reference runs as ~250 tiny PyTorch kernels per iteration either side of the hot ops, restated as seven HBM-bound
// kernels (every one a single streaming pass, one thread per Gaussian / pixel / parameter):
//   s2_activate_kernel            GaussianModel.get_* activations (scene/gaussian_model.py:183-232) + view directions
//                                 (gaussian_renderer/neilf.py:74-76)
//   s2_pack_features_kernel       the S=16 feature row (neilf.py:115-122) + the light-smoothness L1 (neilf.py:286-292)
//   s2_unpack_kernel              its backward: rasterizer feature gradients -> shading-op upstream gradients
//   s2_activate_backward_kernel   chain rule of every activation, summed with the rasterizer / shading gradients,
//                                 straight into the raw-parameter gradient buffers
//   s2_loss_kernel                image-space loss terms AND their gradients in one pass (neilf.py:212-318: L1 on the
//                                 SH image, L1 on the sRGB-mapped PBR image, normal-vs-pseudo-normal MSE)
//   adam_kernel                   multi-group Adam step, all parameter groups in one launch (gaussian_model.py:465-497)
// Parity target: the plain-PyTorch restatement in relightable3dgaussian_amd/train_step.py (Stage2Step), fp32 tolerance.
#include <type_traits>

#include "common.hpp"
#include "pseudo_normal.hpp"
#include "r3dg_hip.h"

namespace r3dg {

// x^y for the sRGB curve (x >= 0.0031308): v_log_f32 * y -> v_exp_f32, 3 instructions and ~4 ulp.  HIP's __powf is the
// full-precision library routine (~155 instructions, a software logarithm): 18 of them per pixel were three quarters of the
// smoothness kernels' instructions and most of s2_pbr_srgb_kernel.  (torch.pow in the reference's rgb_to_srgb,
// utils/graphics_utils.py, is itself good to ~2 ulp; the parity tolerances are 1e-5 and wider.)
__device__ __forceinline__ float srgb_pow(float x, float y) { return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }


__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// F.normalize(v, eps): v / max(|v|, eps)
__device__ __forceinline__ void normalize3(const float v[3], float eps, float out[3], float& inv)
{
    const float n = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    inv = 1.f / fmaxf(n, eps);
    out[0] = v[0] * inv; out[1] = v[1] * inv; out[2] = v[2] * inv;
}
// backward of v / max(|v|, eps): (g - n (n.g)) / |v| when |v| >= eps, g / eps below it (clamp passes no gradient)
__device__ __forceinline__ void normalize3_backward(const float v[3], float eps, const float g[3], float out[3])
{
    const float n = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (n > eps) {
        const float inv = 1.f / n;
        const float u[3] = {v[0] * inv, v[1] * inv, v[2] * inv};
        const float d = u[0] * g[0] + u[1] * g[1] + u[2] * g[2];
#pragma unroll
        for (int c = 0; c < 3; c++) out[c] = (g[c] - u[c] * d) * inv;
    } else {
        const float inv = 1.f / eps;
#pragma unroll
        for (int c = 0; c < 3; c++) out[c] = g[c] * inv;
    }
}

// Side jobs of the activation kernel: microseconds of work that would otherwise be launches of their own on the iteration's critical
// stream (5-20 us each there).  They ride as EXTRA WORKGROUPS behind the nb_main workgroups of the Gaussians:
//   * env[i] = softplus(env_raw[i]) for the n_env floats of the environment texture (DirectLightMap.get_env,
//     scene/direct_light_map.py:18-23; torch's softplus: beta 1, threshold 20);
//   * zero[0..n_zero) = 0 (the iteration's loss sums).
struct ActivateSide {
    int nb_main;
    int n_env;
    const float* env_raw;
    float* env;
    float* zero;
    int n_zero;
};

__global__ void __launch_bounds__(256)
s2_activate_kernel(int P, const float* __restrict__ xyz, const float* __restrict__ scaling_raw,
                   const float* __restrict__ rotation_raw, const float* __restrict__ opacity_raw,
                   const float* __restrict__ normal_raw, const float* __restrict__ base_raw,
                   const float* __restrict__ rough_raw, const float* __restrict__ campos,
                   float* __restrict__ scales, float* __restrict__ rot, float* __restrict__ opacity,
                   float* __restrict__ normal, float* __restrict__ base_color, float* __restrict__ roughness,
                   float* __restrict__ viewdirs, const float* __restrict__ viewmatrix, float* __restrict__ features,
                   ActivateSide side)
{
    if ((int)blockIdx.x >= side.nb_main) {
        const int nb_side = (int)gridDim.x - side.nb_main;
        for (int k = ((int)blockIdx.x - side.nb_main) * 256 + (int)threadIdx.x; k < side.n_env; k += nb_side * 256) {
            const float r = side.env_raw[k];
            side.env[k] = r > 20.f ? r : log1pf(expf(r));
        }
        for (int k = ((int)blockIdx.x - side.nb_main) * 256 + (int)threadIdx.x; k < side.n_zero; k += nb_side * 256)
            side.zero[k] = 0.f;
        return;
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const size_t i3 = 3 * (size_t)i, i4 = 4 * (size_t)i;
#pragma unroll
    for (int c = 0; c < 3; c++) scales[i3 + c] = __expf(scaling_raw[i3 + c]);
    {
        const float q[4] = {rotation_raw[i4], rotation_raw[i4 + 1], rotation_raw[i4 + 2], rotation_raw[i4 + 3]};
        const float inv = 1.f / fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
#pragma unroll
        for (int c = 0; c < 4; c++) rot[i4 + c] = q[c] * inv;
    }
    opacity[i] = sigmoidf_(opacity_raw[i]);
    float inv, nrm[3];
    {
        const float v[3] = {normal_raw[i3], normal_raw[i3 + 1], normal_raw[i3 + 2]};
        normalize3(v, 1e-3f, nrm, inv);
        normal[i3] = nrm[0]; normal[i3 + 1] = nrm[1]; normal[i3 + 2] = nrm[2];
    }
    if (base_raw != nullptr) {
        float bc[3];
#pragma unroll
        for (int c = 0; c < 3; c++) base_color[i3 + c] = bc[c] = 0.03f + 0.77f * sigmoidf_(base_raw[i3 + c]);
        const float rg = 0.09f + 0.9f * sigmoidf_(rough_raw[i]);
        roughness[i] = rg;
        const float p[3] = {xyz[i3], xyz[i3 + 1], xyz[i3 + 2]};
        const float d[3] = {campos[0] - p[0], campos[1] - p[1], campos[2] - p[2]};
        float o[3];
        normalize3(d, 1e-12f, o, inv);
        viewdirs[i3] = o[0]; viewdirs[i3 + 1] = o[1]; viewdirs[i3 + 2] = o[2];
        if (features != nullptr) {
            // the columns of the S=16 feature row that do not wait for the shading integral (s2_pack_features_kernel's, value for
            // value): depth, depth^2 | . . . | normal | base colour | roughness | . . . .   The shading kernels fill in the rest
            const float depth = p[0] * viewmatrix[2] + p[1] * viewmatrix[6] + p[2] * viewmatrix[10] + viewmatrix[14];
            float* f = features + 16 * (size_t)i;
            *reinterpret_cast<float2*>(f) = make_float2(depth, depth * depth);
            f[5] = nrm[0]; f[6] = nrm[1]; f[7] = nrm[2];
            *reinterpret_cast<float4*>(f + 8) = make_float4(bc[0], bc[1], bc[2], rg);
        }
    }
}

__device__ __forceinline__ float block_sum_256(float v, float* s_part)
{
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    return s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

// features[P,16] = depth, depth^2, pbr(3), normal(3), base_color(3), roughness, diffuse_light(3), mean visibility
__global__ void __launch_bounds__(256)
s2_pack_features_kernel(int P, const float* __restrict__ xyz, const float* __restrict__ viewmatrix,
                        const float* __restrict__ normal, const float* __restrict__ base_color,
                        const float* __restrict__ roughness, const float* __restrict__ shade_out,
                        float* __restrict__ features, float* __restrict__ light_l1_sum)
{
    __shared__ float s_part[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float l1 = 0.f;
    if (i < P) {
        const size_t i3 = 3 * (size_t)i;
        const float depth = xyz[i3] * viewmatrix[2] + xyz[i3 + 1] * viewmatrix[6] + xyz[i3 + 2] * viewmatrix[10] +
                            viewmatrix[14];
        const float* so = shade_out + 19 * (size_t)i;
        const float dl[3] = {so[3], so[4], so[5]};
        float4* f = reinterpret_cast<float4*>(features + 16 * (size_t)i);
        f[0] = make_float4(depth, depth * depth, so[0], so[1]);
        f[1] = make_float4(so[2], normal[i3], normal[i3 + 1], normal[i3 + 2]);
        f[2] = make_float4(base_color[i3], base_color[i3 + 1], base_color[i3 + 2], roughness[i]);
        f[3] = make_float4(dl[0], dl[1], dl[2], so[18]);
        const float m = (dl[0] + dl[1] + dl[2]) / 3.f;
        l1 = fabsf(dl[0] - m) + fabsf(dl[1] - m) + fabsf(dl[2] - m);
    }
    const float tot = block_sum_256(l1, s_part);
    if (threadIdx.x == 0 && light_l1_sum != nullptr) atomicAdd(sum_slot(light_l1_sum), tot);
}

__device__ __forceinline__ float signf_(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

// dL_dpbr / dL_ddiffuse_light for the shading op: the rasterizer's feature gradients (cols 2-4 / 12-14) plus the gradient of
// light_weight * sum_c |dl_c - mean(dl)|
// block_absmax (may be NULL): [gridDim.x] floats, max |upstream gradient| of the block's rows, +inf if any is not finite --
// the scale of the shading backward's fixed-point texture accumulation (shading.hip), so that op needs no reduction pass
__global__ void __launch_bounds__(256)
s2_unpack_kernel(int P, const float* __restrict__ dL_dfeatures, const float* __restrict__ shade_out, float light_weight,
                 float* __restrict__ dL_dpbr, float* __restrict__ dL_ddiffuse, float* __restrict__ block_absmax,
                 float* __restrict__ light_l1_sum)
{
    __shared__ float s_m[4];
    __shared__ float s_l1[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float m = 0.f, l1 = 0.f;
    bool bad = false;
    if (i < P) {
        const float4* g = reinterpret_cast<const float4*>(dL_dfeatures + 16 * (size_t)i);
        const float4 g0 = g[0], g1 = g[1], g3 = g[3];
        const size_t i3 = 3 * (size_t)i;
        dL_dpbr[i3] = g0.z; dL_dpbr[i3 + 1] = g0.w; dL_dpbr[i3 + 2] = g1.x;
        const float* so = shade_out + 19 * (size_t)i;
        const float dl[3] = {so[3], so[4], so[5]};
        const float mean = (dl[0] + dl[1] + dl[2]) / 3.f;
        const float s[3] = {signf_(dl[0] - mean), signf_(dl[1] - mean), signf_(dl[2] - mean)};
        const float sm = (s[0] + s[1] + s[2]) / 3.f;
        l1 = fabsf(dl[0] - mean) + fabsf(dl[1] - mean) + fabsf(dl[2] - mean);
        const float d0 = g3.x + light_weight * (s[0] - sm), d1 = g3.y + light_weight * (s[1] - sm),
                    d2 = g3.z + light_weight * (s[2] - sm);
        dL_ddiffuse[i3] = d0; dL_ddiffuse[i3 + 1] = d1; dL_ddiffuse[i3 + 2] = d2;
        const float v[6] = {fabsf(g0.z), fabsf(g0.w), fabsf(g1.x), fabsf(d0), fabsf(d1), fabsf(d2)};
#pragma unroll
        for (int c = 0; c < 6; c++) {
            bad = bad || !(v[c] <= 3.0e38f);
            m = fmaxf(m, v[c]);
        }
    }
    if (light_l1_sum != nullptr) {      // the term's VALUE, when no s2_pack_features_kernel ran in front (it adds the same sum)
        const float tot = block_sum_256(l1, s_l1);
        if (threadIdx.x == 0) atomicAdd(sum_slot(light_l1_sum), tot);
    }
    if (block_absmax == nullptr) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (__ballot(bad) != 0ull) m = __uint_as_float(0x7f800000u);
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) block_absmax[blockIdx.x] = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
}

// DirectLightMap: env = softplus(raw) (direct_light_map.py:18-23) and the total-variation smoothness term on it
// (neilf.py:294-300): g_raw = (dL_denv + w_tv * dTV/denv) * softplus'(raw); *tv_sum += TV(env) (unweighted).
// env layout [He,We,3]; TV = mean (d/dh)^2 + mean (d/dw)^2 over the 3 x He x We image (the reference's tv_loss,
// utils/loss_utils.py:113-117, pinned by tests/golden/ssim_reference.npz).
// One workgroup of 256 threads per 256 texture floats (`block` = its index); a launch of its own (s2_env_backward_kernel) or the
// extra workgroups of s2_activate_backward_kernel (the texture is 1536 floats: six workgroups that took 11-20 us as a launch on
// the critical stream between the activation chain rule and Adam).
struct EnvBackward {
    int He, We;
    const float* raw;
    const float* env;
    float* dL_denv;
    float w_tv;
    float* g_raw;
    float* tv_sum;
    int consume;
};

__device__ __forceinline__ void env_backward_block(int block, const EnvBackward& a)
{
    __shared__ float s_part[4];
    const int He = a.He, We = a.We;
    const int n = He * We * 3;
    const int i = block * 256 + threadIdx.x;
    float tv = 0.f;
    if (i < n) {
        const int w = (i / 3) % We, h = i / (3 * We);
        const float inv_v = He > 1 ? 1.f / (3.f * (He - 1) * We) : 0.f, inv_h = We > 1 ? 1.f / (3.f * He * (We - 1)) : 0.f;
        const float x = a.env[i];
        float g = 0.f;
        // tv_loss (utils/loss_utils.py:113-117): mean SQUARED forward difference along h plus the same along w
        if (h > 0) g += inv_v * 2.f * (x - a.env[i - 3 * We]);
        if (h < He - 1) { const float d = a.env[i + 3 * We] - x; g -= inv_v * 2.f * d; tv += inv_v * d * d; }
        if (w > 0) g += inv_h * 2.f * (x - a.env[i - 3]);
        if (w < We - 1) { const float d = a.env[i + 3] - x; g -= inv_h * 2.f * d; tv += inv_h * d * d; }
        const float r = a.raw[i];
        const float dsoft = r > 20.f ? 1.f : sigmoidf_(r);
        a.g_raw[i] = (a.dL_denv[i] + a.w_tv * g) * dsoft;
        if (a.consume) a.dL_denv[i] = 0.f;        // the accumulator is handed back zeroed for the next shading backward
    }
    const float tot = block_sum_256(tv, s_part);
    if (threadIdx.x == 0 && a.tv_sum != nullptr) atomicAdd(sum_slot(a.tv_sum), tot);
}

__global__ void __launch_bounds__(256)
s2_env_backward_kernel(EnvBackward a)
{
    env_backward_block((int)blockIdx.x, a);
}

__global__ void __launch_bounds__(256)
s2_activate_backward_kernel(int P, const float* __restrict__ xyz, const float* __restrict__ scaling_raw,
                            const float* __restrict__ rotation_raw, const float* __restrict__ opacity_raw,
                            const float* __restrict__ normal_raw, const float* __restrict__ base_raw,
                            const float* __restrict__ rough_raw, const float* __restrict__ viewmatrix,
                            const float* __restrict__ campos, const float* __restrict__ dL_dfeatures,
                            const float* __restrict__ dL_dbase_shade, const float* __restrict__ dL_drough_shade,
                            const float* __restrict__ dL_dviewdirs, const float* __restrict__ dL_dscales,
                            const float* __restrict__ dL_drot, const float* __restrict__ dL_dopacity,
                            const float* __restrict__ dL_dmeans3D, float* __restrict__ g_xyz,
                            float* __restrict__ g_scaling, float* __restrict__ g_rotation,
                            float* __restrict__ g_opacity, float* __restrict__ g_normal, float* __restrict__ g_base,
                            float* __restrict__ g_rough, int nb_main, EnvBackward env_job)
{
    if ((int)blockIdx.x >= nb_main) {             // side job: the environment texture's chain rule (EnvBackward)
        env_backward_block((int)blockIdx.x - nb_main, env_job);
        return;
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const size_t i3 = 3 * (size_t)i, i4 = 4 * (size_t)i;
    const float4* gf = reinterpret_cast<const float4*>(dL_dfeatures + 16 * (size_t)i);
    const float4 f0 = gf[0], f1 = gf[1], f2 = gf[2];
    // base_color = 0.03 + 0.77 sigmoid(raw), roughness = 0.09 + 0.9 sigmoid(raw): feature row + shading op
    {
        const float gfeat[3] = {f2.x, f2.y, f2.z};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float s = sigmoidf_(base_raw[i3 + c]);
            g_base[i3 + c] = (gfeat[c] + dL_dbase_shade[i3 + c]) * 0.77f * s * (1.f - s);
        }
        const float s = sigmoidf_(rough_raw[i]);
        g_rough[i] = (f2.w + dL_drough_shade[i]) * 0.9f * s * (1.f - s);
    }
    // frozen geometry (g_xyz == NULL, wave-uniform): base colour and roughness are the only per-Gaussian groups that train
    if (g_xyz == nullptr) return;

    // scales = exp(raw)
#pragma unroll
    for (int c = 0; c < 3; c++) g_scaling[i3 + c] = dL_dscales[i3 + c] * __expf(scaling_raw[i3 + c]);
    // rotation = q / max(|q|, 1e-12)
    {
        const float q[4] = {rotation_raw[i4], rotation_raw[i4 + 1], rotation_raw[i4 + 2], rotation_raw[i4 + 3]};
        const float g[4] = {dL_drot[i4], dL_drot[i4 + 1], dL_drot[i4 + 2], dL_drot[i4 + 3]};
        const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        if (n > 1e-12f) {
            const float inv = 1.f / n;
            const float d = (q[0] * g[0] + q[1] * g[1] + q[2] * g[2] + q[3] * g[3]) * inv * inv;
#pragma unroll
            for (int c = 0; c < 4; c++) g_rotation[i4 + c] = (g[c] - q[c] * d) * inv;
        } else {
#pragma unroll
            for (int c = 0; c < 4; c++) g_rotation[i4 + c] = g[c] * 1e12f;
        }
    }
    // opacity = sigmoid(raw)
    {
        const float s = sigmoidf_(opacity_raw[i]);
        g_opacity[i] = dL_dopacity[i] * s * (1.f - s);
    }
    // normal = raw / max(|raw|, 1e-3); only the feature row carries its gradient (the shading op sees normal.detach())
    {
        const float v[3] = {normal_raw[i3], normal_raw[i3 + 1], normal_raw[i3 + 2]};
        const float g[3] = {f1.y, f1.z, f1.w};
        float o[3];
        normalize3_backward(v, 1e-3f, g, o);
        g_normal[i3] = o[0]; g_normal[i3 + 1] = o[1]; g_normal[i3 + 2] = o[2];
    }
    // xyz: rasterizer + depth / depth^2 feature columns + view direction
    {
        const float p[3] = {xyz[i3], xyz[i3 + 1], xyz[i3 + 2]};
        const float depth = p[0] * viewmatrix[2] + p[1] * viewmatrix[6] + p[2] * viewmatrix[10] + viewmatrix[14];
        const float gd = f0.x + 2.f * depth * f0.y;
        const float d[3] = {campos[0] - p[0], campos[1] - p[1], campos[2] - p[2]};
        const float gv[3] = {dL_dviewdirs[i3], dL_dviewdirs[i3 + 1], dL_dviewdirs[i3 + 2]};
        float gdd[3];
        normalize3_backward(d, 1e-12f, gv, gdd);
        g_xyz[i3] = dL_dmeans3D[i3] + gd * viewmatrix[2] - gdd[0];
        g_xyz[i3 + 1] = dL_dmeans3D[i3 + 1] + gd * viewmatrix[6] - gdd[1];
        g_xyz[i3 + 2] = dL_dmeans3D[i3 + 2] + gd * viewmatrix[10] - gdd[2];
    }
}

// sRGB-mapped PBR image (neilf.py:190-200: feat = feature / max(opacity,1e-5) * mask, pbr_img = feat[2:5]*op + (1-op)*bg,
// linear -> sRGB), materialised for the SSIM term
__global__ void __launch_bounds__(256)
s2_pbr_srgb_kernel(int HW, const float* __restrict__ opacity, const float* __restrict__ feature,
                   const int* __restrict__ n_contrib, const float* __restrict__ bg, float* __restrict__ srgb)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const float op = opacity[i];
    const float scale = n_contrib[i] > 0 ? 1.f / fmaxf(op, 1e-5f) : 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float x = feature[(size_t)(2 + c) * HW + i] * scale * op + (1.f - op) * bg[c];
        const float v = x <= 0.0031308f ? 12.92f * x : 1.055f * srgb_pow(fmaxf(x, 0.0031308f), 1.f / 2.4f) - 0.055f;
        srgb[(size_t)c * HW + i] = fminf(fmaxf(v, 0.f), 1.f);      // rgb_to_srgb clips (utils/graphics_utils.py:211-212)
    }
}

// pseudo_normal_kernel (the rasterizer forward's K9 + K10, pseudo_normal.hpp) and s2_pbr_srgb_kernel as ONE launch, pixel by pixel:
// the two are independent per-pixel passes over the tile forward's outputs that followed each other on the critical stream (10 + 7
// us at 800x800).  Same values as the two kernels (same device functions).
__global__ void __launch_bounds__(256)
s2_normals_srgb_kernel(int W, int H, float focal_x, float focal_y, float cx, float cy, const float* __restrict__ vm,
                       const float* __restrict__ opacity, const float* __restrict__ depths, float* __restrict__ normals,
                       float* __restrict__ surface_xyz, const float* __restrict__ feature, const int* __restrict__ n_contrib,
                       const float* __restrict__ bg, float* __restrict__ srgb)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const int HW = W * H, i = y * W + x;
    {
        const float op = opacity[i];
        const float scale = n_contrib[i] > 0 ? 1.f / fmaxf(op, 1e-5f) : 0.f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float v0 = feature[(size_t)(2 + c) * HW + i] * scale * op + (1.f - op) * bg[c];
            const float v = v0 <= 0.0031308f ? 12.92f * v0 : 1.055f * srgb_pow(fmaxf(v0, 0.0031308f), 1.f / 2.4f) - 0.055f;
            srgb[(size_t)c * HW + i] = fminf(fmaxf(v, 0.f), 1.f);
        }
    }
    pseudo_normal_pixel(x, y, W, H, focal_x, focal_y, cx, cy, vm, opacity, depths, normals, surface_xyz);
}

// Image-space loss + gradient in one pass.  sums[0..2] += sum|image-gt|, sum|srgb(pbr)-gt|, sum (n_render - n_pseudo)^2
// (unweighted); gradients carry the weights w_* (already divided by the element counts).
// SPARSE: only the feature-gradient maps that carry a loss term are written (2-4; 5-7 when w_normal != 0) -- for callers
// whose rasterizer backward reads exactly those (active_features); with w_normal == 0 the normal maps are not even read.
template <bool SPARSE>
__global__ void __launch_bounds__(256)
s2_loss_kernel(int HW, const float* __restrict__ image, const float* __restrict__ opacity,
               const float* __restrict__ feature, const float* __restrict__ pseudo_normal,
               const int* __restrict__ n_contrib, const float* __restrict__ gt, const float* __restrict__ bg,
               const float* __restrict__ image_mask, float w_l1, float w_pbr, float w_normal,
               const float* __restrict__ extra_dimage,
               const float* __restrict__ extra_dsrgb, float* __restrict__ dL_dimage,
               float* __restrict__ dL_dopacity, float* __restrict__ dL_dfeature, float* __restrict__ sums)
{
    __shared__ float s_part[4];
    float s_l1 = 0.f, s_pbr = 0.f, s_n = 0.f;
    // grid-stride: a few hundred blocks, so the three same-address atomics per block do not serialise the kernel
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        // every load of the pixel first (up to 20 of them, no branch around any: as written per channel -- load, use, store, next
        // channel -- the compiler kept that order and the kernel waited for memory eight times per pixel: tools/isa_waits.py)
        const float op = opacity[i];
        const int nc = n_contrib[i];
        float v_gt[3], v_im[3], v_ei[3], v_F[3], v_es[3], v_Fn[3], v_pn[3];
        const bool normal_on = !SPARSE || w_normal != 0.f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            v_gt[c] = gt[(size_t)c * HW + i];
            v_im[c] = image[(size_t)c * HW + i];
            v_F[c] = feature[(size_t)(2 + c) * HW + i];
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            v_ei[c] = extra_dimage ? extra_dimage[(size_t)c * HW + i] : 0.f;
            v_es[c] = extra_dsrgb ? extra_dsrgb[(size_t)c * HW + i] : 0.f;
        }
        float mk = 1.f;
        if (normal_on) {
            mk = image_mask ? image_mask[i] : 1.f;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                v_Fn[c] = feature[(size_t)(5 + c) * HW + i];
                v_pn[c] = pseudo_normal[(size_t)c * HW + i];
            }
        }
        const bool mask = nc > 0;
        const float opc = fmaxf(op, 1e-5f);
        const float scale = mask ? 1.f / opc : 0.f;                 // feat = feature * scale
        const float dscale_dop = (mask && op >= 1e-5f) ? -1.f / (opc * opc) : 0.f;
        float g_op = 0.f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float g = v_gt[c];
            // L1 on the SH image
            const float d0 = v_im[c] - g;
            s_l1 += fabsf(d0);
            // extra_*: gradients of further terms on the same two images (the SSIM terms, r3dg_ssim_backward)
            dL_dimage[(size_t)c * HW + i] = w_l1 * signf_(d0) + v_ei[c];
            // L1 on the sRGB-mapped PBR image: pbr_img = r_pbr * op + (1 - op) * bg
            const float F = v_F[c];
            const float r = F * scale;
            const float x = r * op + (1.f - op) * bg[c];
            const bool lin = x <= 0.0031308f;
            const float xs = fmaxf(x, 0.0031308f);
            // rgb_to_srgb with clip=True (utils/graphics_utils.py:207-213): the curve, then clamp to [0,1]; the clamp passes
            // the gradient only where 0 <= curve <= 1 (torch.clamp), so saturated HDR highlights stop pulling
            const float curve = lin ? 12.92f * x : 1.055f * srgb_pow(xs, 1.f / 2.4f) - 0.055f;
            const bool unclipped = curve >= 0.f && curve <= 1.f;
            const float srgb = fminf(fmaxf(curve, 0.f), 1.f);
            const float d1 = srgb - g;
            s_pbr += fabsf(d1);
            const float dsrgb = !unclipped ? 0.f : (lin ? 12.92f : 1.055f / 2.4f * srgb_pow(xs, 1.f / 2.4f - 1.f));
            const float gx = (w_pbr * signf_(d1) + v_es[c]) * dsrgb;   // dL/dx
            dL_dfeature[(size_t)(2 + c) * HW + i] = gx * op * scale;
            g_op += gx * (r - bg[c] + op * F * dscale_dop);
            // normal consistency: mse(r_normal, pseudo_normal)
            if (normal_on) {
                // mse(normal * m, pseudo_normal * m), m = the view's object mask (neilf.py:258-264; NULL = all ones)
                const float Fn = v_Fn[c];
                const float dn = (Fn * scale - v_pn[c]) * mk;
                s_n += dn * dn;
                const float gn = 2.f * w_normal * dn * mk;
                dL_dfeature[(size_t)(5 + c) * HW + i] = gn * scale;
                g_op += gn * Fn * dscale_dop;
            }
        }
        dL_dopacity[i] = g_op;
        if (!SPARSE) {
            dL_dfeature[i] = 0.f;
            dL_dfeature[(size_t)HW + i] = 0.f;
#pragma unroll
            for (int c = 8; c < 16; c++) dL_dfeature[(size_t)c * HW + i] = 0.f;
        }
    }
    const float t0 = block_sum_256(s_l1, s_part);
    __syncthreads();
    const float t1 = block_sum_256(s_pbr, s_part);
    __syncthreads();
    const float t2 = block_sum_256(s_n, s_part);
    if (threadIdx.x == 0) {
        atomicAdd(sum_slot(sums + 0 * R3DG_SUM_SLOTS), t0);
        atomicAdd(sum_slot(sums + 1 * R3DG_SUM_SLOTS), t1);
        atomicAdd(sum_slot(sums + 2 * R3DG_SUM_SLOTS), t2);
    }
}

// ---- stage 1 (plain 3DGS + normals, gaussian_renderer/render.py:15-130): S = 5 feature row [normal, depth, depth^2] ---
__global__ void __launch_bounds__(256)
s1_pack_features_kernel(int P, const float* __restrict__ xyz, const float* __restrict__ viewmatrix,
                        const float* __restrict__ normal, float* __restrict__ features)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const size_t i3 = 3 * (size_t)i;
    const float depth = xyz[i3] * viewmatrix[2] + xyz[i3 + 1] * viewmatrix[6] + xyz[i3 + 2] * viewmatrix[10] +
                        viewmatrix[14];
    float* f = features + 5 * (size_t)i;
    f[0] = normal[i3]; f[1] = normal[i3 + 1]; f[2] = normal[i3 + 2];
    f[3] = depth; f[4] = depth * depth;
}

// ---- stage-1 objective (gaussian_renderer/render.py:137-223 with the flags of script/run_nerf.sh:7-14) ---------------------
//   L = (1-l)*L1(image, gt) [+ l*(1-SSIM): csrc/ssim.hip, gradient arrives in extra_dimage]
//     + lambda_mask_entropy        * -mean(m log o + (1-m) log(1-o)),  o = clamp(opacity, 1e-6, 1-1e-6)      (:156-160)
//     + lambda_normal_render_depth * mse(normal*m, pseudo_normal*m)                                           (:162-167)
//     + lambda_normal_smooth       * first_order_edge_aware_loss(normal, gt)                                  (:169-173)
//     + lambda_depth_var(iter)     * mean sqrt(max(depth2 - depth^2, 1e-6))                                   (:199-205)
// with [normal, depth, depth2] = feature / max(opacity, 1e-5) * (n_contrib > 0) (:107-112) and m the view's object mask.
// first_order_edge_aware_loss (utils/loss_utils.py:104-105) = mean_{c,y,x} sum_{d in {x,y}} |G_d normal_c| exp(-|G_d gt_c|) with
// G = kornia.filters.spatial_gradient(order=1) of kornia 0.6.12 (readme.md:31-32; the package is not in this image, its
// published algorithm is restated): 3x3 Sobel cross-correlation, kernels [[-1,0,1],[-2,0,2],[-1,0,1]] and its transpose,
// normalised by the sum of absolute values (/8), replicate padding.
// Pass A (s1_edge_kernel): per pixel the six values sign(G_d normal_c) * exp(-|G_d gt_c|) and the loss sum.
// Pass B (inside s1_loss_kernel): the adjoint of the replicate-padded stencil, gathered (no atomics).
__device__ __forceinline__ float s1_rendered(const float* __restrict__ feature, const float* __restrict__ opacity,
                                             const int* __restrict__ n_contrib, size_t HW, int ch, size_t pix)
{
    const float opc = fmaxf(opacity[pix], 1e-5f);
    return n_contrib[pix] > 0 ? feature[(size_t)ch * HW + pix] / opc : 0.f;
}

__global__ void __launch_bounds__(256)
s1_edge_kernel(int W, int H, const float* __restrict__ feature, const float* __restrict__ opacity,
               const int* __restrict__ n_contrib, const float* __restrict__ gt, float* __restrict__ edge_g /*[3][2][HW]*/,
               float* __restrict__ sum_out)
{
    __shared__ float s_part[4];
    const size_t HW = (size_t)W * H;
    float acc = 0.f;
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < HW; i += (size_t)gridDim.x * 256) {
        const int y = (int)(i / W), x = (int)(i % W);
        const int ys[3] = {y > 0 ? y - 1 : 0, y, y < H - 1 ? y + 1 : H - 1};
        const int xs[3] = {x > 0 ? x - 1 : 0, x, x < W - 1 ? x + 1 : W - 1};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float n[3][3], g[3][3];
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int b = 0; b < 3; b++) {
                    const size_t q = (size_t)ys[a] * W + xs[b];
                    n[a][b] = s1_rendered(feature, opacity, n_contrib, HW, c, q);
                    g[a][b] = gt[(size_t)c * HW + q];
                }
            const float nx = ((n[0][2] - n[0][0]) + 2.f * (n[1][2] - n[1][0]) + (n[2][2] - n[2][0])) * 0.125f;
            const float ny = ((n[2][0] - n[0][0]) + 2.f * (n[2][1] - n[0][1]) + (n[2][2] - n[0][2])) * 0.125f;
            const float gx = ((g[0][2] - g[0][0]) + 2.f * (g[1][2] - g[1][0]) + (g[2][2] - g[2][0])) * 0.125f;
            const float gy = ((g[2][0] - g[0][0]) + 2.f * (g[2][1] - g[0][1]) + (g[2][2] - g[0][2])) * 0.125f;
            const float ex = __expf(-fabsf(gx)), ey = __expf(-fabsf(gy));
            acc += fabsf(nx) * ex + fabsf(ny) * ey;
            edge_g[(size_t)(2 * c) * HW + i] = signf_(nx) * ex;
            edge_g[(size_t)(2 * c + 1) * HW + i] = signf_(ny) * ey;
        }
    }
    const float t = block_sum_256(acc, s_part);
    if (threadIdx.x == 0) atomicAdd(sum_slot(sum_out), t);
}

// sum_d [clamp(q + d, 0, n-1) == p] * k[d+1]: weight with which position q's replicate-padded 1-D stencil reads position p
__device__ __forceinline__ float s1_adj1(int q, int p, int n, float km, float k0, float kp)
{
    float w = (q == p) ? k0 : 0.f;
    const int qm = q > 0 ? q - 1 : 0, qp = q < n - 1 ? q + 1 : n - 1;
    w += (qm == p) ? km : 0.f;
    w += (qp == p) ? kp : 0.f;
    return w;
}

// sums[0] += sum|image-gt|, [1] += sum m^2 (normal - pseudo)^2, [2] += sum -(m log o + (1-m) log(1-o)), [5] += sum sqrt(var)
// (sums[3] is the SSIM slot, sums[4] the edge-aware sum of s1_edge_kernel); image_mask == nullptr means all ones.
__global__ void __launch_bounds__(256)
s1_loss_kernel(int W, int H, const float* __restrict__ image, const float* __restrict__ opacity,
               const float* __restrict__ feature, const float* __restrict__ pseudo_normal,
               const int* __restrict__ n_contrib, const float* __restrict__ gt, const float* __restrict__ image_mask,
               float w_l1, float w_entropy, float w_normal, float w_smooth, float w_var,
               const float* __restrict__ extra_dimage, const float* __restrict__ edge_g, float* __restrict__ dL_dimage,
               float* __restrict__ dL_dopacity, float* __restrict__ dL_dfeature, float* __restrict__ sums)
{
    __shared__ float s_part[4];
    const size_t HW = (size_t)W * H;
    float s_l1 = 0.f, s_n = 0.f, s_e = 0.f, s_v = 0.f;
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < HW; i += (size_t)gridDim.x * 256) {
        const float op = opacity[i];
        const bool mask = n_contrib[i] > 0;
        const float opc = fmaxf(op, 1e-5f);
        const float scale = mask ? 1.f / opc : 0.f;
        const float dscale_dop = (mask && op >= 1e-5f) ? -1.f / (opc * opc) : 0.f;
        const float m = image_mask ? image_mask[i] : 1.f;
        // mask entropy
        const float o = fminf(fmaxf(op, 1e-6f), 1.f - 1e-6f);
        s_e -= m * __logf(o) + (1.f - m) * __logf(1.f - o);
        float g_op = (op >= 1e-6f && op <= 1.f - 1e-6f) ? -w_entropy * (m / o - (1.f - m) / (1.f - o)) : 0.f;
        // adjoint of the edge-aware stencil: weights of the (up to) 9 neighbours q whose stencil reads this pixel
        float dsm[3] = {0.f, 0.f, 0.f};
        if (edge_g != nullptr && w_smooth != 0.f) {
            const int y = (int)(i / W), x = (int)(i % W);
#pragma unroll
            for (int a = -1; a <= 1; a++) {
                const int qy = y + a;
                if (qy < 0 || qy >= H) continue;
                const float sy = s1_adj1(qy, y, H, 1.f, 2.f, 1.f), dy = s1_adj1(qy, y, H, -1.f, 0.f, 1.f);
#pragma unroll
                for (int b = -1; b <= 1; b++) {
                    const int qx = x + b;
                    if (qx < 0 || qx >= W) continue;
                    const float sx = s1_adj1(qx, x, W, 1.f, 2.f, 1.f), dx = s1_adj1(qx, x, W, -1.f, 0.f, 1.f);
                    const float wx = sy * dx * 0.125f, wy = dy * sx * 0.125f;
                    const size_t q = (size_t)qy * W + qx;
#pragma unroll
                    for (int c = 0; c < 3; c++)
                        dsm[c] += wx * edge_g[(size_t)(2 * c) * HW + q] + wy * edge_g[(size_t)(2 * c + 1) * HW + q];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float d0 = image[(size_t)c * HW + i] - gt[(size_t)c * HW + i];
            s_l1 += fabsf(d0);
            dL_dimage[(size_t)c * HW + i] = w_l1 * signf_(d0) + (extra_dimage ? extra_dimage[(size_t)c * HW + i] : 0.f);
            const float Fn = feature[(size_t)c * HW + i];
            const float dn = (Fn * scale - pseudo_normal[(size_t)c * HW + i]) * m;
            s_n += dn * dn;
            const float gn = 2.f * w_normal * dn * m + w_smooth * dsm[c];       // dL / d rendered_normal_c
            dL_dfeature[(size_t)c * HW + i] = gn * scale;
            g_op += gn * Fn * dscale_dop;
        }
        // depth variance
        const float F3 = feature[(size_t)3 * HW + i], F4 = feature[(size_t)4 * HW + i];
        const float D = F3 * scale, D2 = F4 * scale;
        const float var = D2 - D * D;
        const float sd = sqrtf(fmaxf(var, 1e-6f));
        s_v += sd;
        const float dvar = var >= 1e-6f ? w_var * 0.5f / sd : 0.f;
        const float gD = -2.f * D * dvar;
        dL_dfeature[(size_t)3 * HW + i] = gD * scale;
        dL_dfeature[(size_t)4 * HW + i] = dvar * scale;
        g_op += (gD * F3 + dvar * F4) * dscale_dop;
        dL_dopacity[i] = g_op;
    }
    const float t0 = block_sum_256(s_l1, s_part);
    __syncthreads();
    const float t1 = block_sum_256(s_n, s_part);
    __syncthreads();
    const float t2 = block_sum_256(s_e, s_part);
    __syncthreads();
    const float t3 = block_sum_256(s_v, s_part);
    if (threadIdx.x == 0) {
        atomicAdd(sum_slot(sums + 0 * R3DG_SUM_SLOTS), t0);
        atomicAdd(sum_slot(sums + 1 * R3DG_SUM_SLOTS), t1);
        atomicAdd(sum_slot(sums + 2 * R3DG_SUM_SLOTS), t2);
        atomicAdd(sum_slot(sums + 5 * R3DG_SUM_SLOTS), t3);
    }
}

// ---- stage-2 edge-aware smoothness terms (Synthetic4Relight / DTU objective) ------------------------------------------
// neilf.py:275-292 with the flags of script/run_syn4.sh:34-36 / run_dtu.sh:36-38:
//   lambda_base_color_smooth * first_order_edge_aware_loss(base_color * m, gt)
// + lambda_roughness_smooth  * first_order_edge_aware_loss(roughness  * m, gt)        (1 channel against 3: broadcast)
// + lambda_light_smooth      * first_order_edge_aware_loss(diffuse    * m, rendered_normal)   (the guide is NOT detached)
// first_order_edge_aware_loss(data, img) = mean_{c,y,x} sum_d |G_d data_c| exp(-|G_d img_c|), G = Sobel / 8 with replicate
// padding (see the stage-1 comment above); X = feature_X / max(opacity, 1e-5) * (n_contrib > 0), m = the view's object mask.
// Three passes: (0) s2_smooth_maps_kernel materialises the ten divided (and masked) maps the stencils read; (A)
// s2_smooth_edge_kernel evaluates the stencils, adds the three sums and writes, per pixel, the 20 values the adjoint needs
// (weights folded in); (B) s2_smooth_backward_kernel gathers the adjoint of the replicate-padded stencil (no atomics) and
// applies the chain rule of the division into the feature-gradient maps and the opacity gradient r3dg_stage2_loss left.
// The base-colour and diffuse-light maps enter through the sRGB curve and its clip to [0,1] -- the loss reads
// results["base_color"] = rgb_to_srgb(rendered_base_color), results["diffuse"] likewise (neilf.py:153-155); the clip passes no
// gradient outside [0,1] -- the roughness map as rendered.
// rend layout [10][HW]: srgb(base_color)*m 0..2 | roughness*m 3 | srgb(diffuse)*m 4..6 | normal 7..9
// edge layout [20][HW]: base 2c+d (0..5) | roughness d (6,7) | diffuse 8+2c+d | normal guide 14+2c+d      (d: 0 = x, 1 = y)
// rgb_to_srgb with clip=True (utils/graphics_utils.py:207-213) and its derivative (0 where the clamp is active)
__device__ __forceinline__ float srgb_clip(float x)
{
    const float curve = x <= 0.0031308f ? 12.92f * x : 1.055f * srgb_pow(fmaxf(x, 0.0031308f), 1.f / 2.4f) - 0.055f;
    return fminf(fmaxf(curve, 0.f), 1.f);
}
__device__ __forceinline__ float srgb_clip_derivative(float x)
{
    const bool lin = x <= 0.0031308f;
    const float xs = fmaxf(x, 0.0031308f);
    const float curve = lin ? 12.92f * x : 1.055f * srgb_pow(xs, 1.f / 2.4f) - 0.055f;
    if (!(curve >= 0.f && curve <= 1.f)) return 0.f;
    return lin ? 12.92f : 1.055f / 2.4f * srgb_pow(xs, 1.f / 2.4f - 1.f);
}

// One tap of the stencils' adjoint and the chain rule behind it, shared by the three formulations of the smoothness terms below
// (three passes / LDS tiles / streamed): written with contraction off, so that the formulations agree bit for bit whatever
// contraction the compiler would pick for a * b + c * d in each context.
__device__ __forceinline__ float smooth_tap(float d, float wx, float ex, float wy, float ey)
{
#pragma clang fp contract(off)          // (HIP's __fmul_rn / __fadd_rn are inline operators that still fuse: this pins the roundings)
    const float t = wx * ex;
    const float u = __builtin_fmaf(wy, ey, t);
    return d + u;
}
// d[0..9]: dL / d (rend 0..9) of s2_smooth_maps_kernel at one pixel; f[0..9]: feature 5..14 there.  gf[0..9]: dL / d feature 5..14
// (normal 3 | base colour 3 | roughness | diffuse light 3; entries of absent terms untouched), returns dL / d opacity.
__device__ __forceinline__ float smooth_chain(bool base, bool rough, bool light, const float (&d)[10], float op, int nc, float m,
                                              const float (&f)[10], float (&gf)[10])
{
#pragma clang fp contract(off)
    const bool mask = nc > 0;
    const float opc = fmaxf(op, 1e-5f);
    const float scale = mask ? 1.f / opc : 0.f;
    const float dscale_dop = (mask && op >= 1e-5f) ? -1.f / (opc * opc) : 0.f;
    float g_op = 0.f;
    if (base) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float F = f[3 + c];
            const float g = d[c] * m * srgb_clip_derivative(F * scale);       // dL / d (linear base colour map)
            gf[3 + c] = g * scale;
            const float t = g * F * dscale_dop;
            g_op = g_op + t;
        }
    }
    if (rough) {
        const float g = d[3] * m;
        gf[6] = g * scale;
        const float t = g * f[6] * dscale_dop;
        g_op = g_op + t;
    }
    if (light) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float Fd = f[7 + c];
            const float g = d[4 + c] * m * srgb_clip_derivative(Fd * scale);   // dL / d (linear diffuse map)
            gf[7 + c] = g * scale;
            const float t = g * Fd * dscale_dop;
            g_op = g_op + t;
            const float gn = d[7 + c];
            gf[c] = gn * scale;
            const float u = gn * f[c] * dscale_dop;
            g_op = g_op + u;
        }
    }
    return g_op;
}

// `old_normal`: the three values the normal maps' gradient already holds at this pixel when the caller has read them ahead of
// time (the streamed kernel requests them before its nine taps: a load used straight away costs a wave its full latency), else NULL
__device__ __forceinline__ void smooth_store(bool base, bool rough, bool light, int accumulate_normal, const float (&gf)[10],
                                             size_t HW, size_t i, float* dL_dfeature,
                                             const float* old_normal = nullptr)
{
#pragma clang fp contract(off)
    if (base) {
#pragma unroll
        for (int c = 0; c < 3; c++) dL_dfeature[(size_t)(8 + c) * HW + i] = gf[3 + c];
    }
    if (rough) dL_dfeature[(size_t)11 * HW + i] = gf[6];
    if (light) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            dL_dfeature[(size_t)(12 + c) * HW + i] = gf[7 + c];
            const size_t o = (size_t)(5 + c) * HW + i;
            const float old = accumulate_normal ? (old_normal != nullptr ? old_normal[c] : dL_dfeature[o]) : 0.f;
            dL_dfeature[o] = old + gf[c];
        }
    }
}

__global__ void __launch_bounds__(256)
s2_smooth_maps_kernel(int HW, const float* __restrict__ opacity, const float* __restrict__ feature,
                      const int* __restrict__ n_contrib, const float* __restrict__ image_mask, float* __restrict__ rend)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const float scale = n_contrib[i] > 0 ? 1.f / fmaxf(opacity[i], 1e-5f) : 0.f;
    const float m = image_mask ? image_mask[i] : 1.f;
#pragma unroll
    for (int c = 0; c < 7; c++) {
        const float x = feature[(size_t)(8 + c) * HW + i] * scale;
        // results["base_color"] / results["diffuse"] are rgb_to_srgb(.) with its clip to [0,1] (neilf.py:153-155); roughness is not
        rend[(size_t)c * HW + i] = (c == 3 ? x : srgb_clip(x)) * m;
    }
#pragma unroll
    for (int c = 0; c < 3; c++) rend[(size_t)(7 + c) * HW + i] = feature[(size_t)(5 + c) * HW + i] * scale;
}

__device__ __forceinline__ void sobel3(const float* __restrict__ map, int W, const int (&ys)[3], const int (&xs)[3],
                                       float& gx, float& gy)
{
    float v[3][3];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) v[a][b] = map[(size_t)ys[a] * W + xs[b]];
    gx = ((v[0][2] - v[0][0]) + 2.f * (v[1][2] - v[1][0]) + (v[2][2] - v[2][0])) * 0.125f;
    gy = ((v[2][0] - v[0][0]) + 2.f * (v[2][1] - v[0][1]) + (v[2][2] - v[0][2])) * 0.125f;
}

__global__ void __launch_bounds__(256)
s2_smooth_edge_kernel(int W, int H, const float* __restrict__ rend, const float* __restrict__ gt, float w_base,
                      float w_rough, float w_light, float* __restrict__ edge, float* __restrict__ sums3)
{
    __shared__ float s_part[4];
    const size_t HW = (size_t)W * H;
    float a_base = 0.f, a_rough = 0.f, a_light = 0.f;
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < HW; i += (size_t)gridDim.x * 256) {
        const int y = (int)(i / W), x = (int)(i % W);
        const int ys[3] = {y > 0 ? y - 1 : 0, y, y < H - 1 ? y + 1 : H - 1};
        const int xs[3] = {x > 0 ? x - 1 : 0, x, x < W - 1 ? x + 1 : W - 1};
        float ex[3], ey[3];                       // exp(-|G_d gt_c|)
        if (w_base != 0.f || w_rough != 0.f) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                float gx, gy;
                sobel3(gt + (size_t)c * HW, W, ys, xs, gx, gy);
                ex[c] = __expf(-fabsf(gx));
                ey[c] = __expf(-fabsf(gy));
            }
        }
        if (w_base != 0.f) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                float dx, dy;
                sobel3(rend + (size_t)c * HW, W, ys, xs, dx, dy);
                a_base += fabsf(dx) * ex[c] + fabsf(dy) * ey[c];
                edge[(size_t)(2 * c) * HW + i] = w_base * signf_(dx) * ex[c];
                edge[(size_t)(2 * c + 1) * HW + i] = w_base * signf_(dy) * ey[c];
            }
        }
        if (w_rough != 0.f) {
            float dx, dy;
            sobel3(rend + (size_t)3 * HW, W, ys, xs, dx, dy);
            const float sx = ex[0] + ex[1] + ex[2], sy = ey[0] + ey[1] + ey[2];
            a_rough += fabsf(dx) * sx + fabsf(dy) * sy;
            edge[(size_t)6 * HW + i] = w_rough * signf_(dx) * sx;
            edge[(size_t)7 * HW + i] = w_rough * signf_(dy) * sy;
        }
        if (w_light != 0.f) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                float dx, dy, nx, ny;
                sobel3(rend + (size_t)(4 + c) * HW, W, ys, xs, dx, dy);
                sobel3(rend + (size_t)(7 + c) * HW, W, ys, xs, nx, ny);
                const float gx = __expf(-fabsf(nx)), gy = __expf(-fabsf(ny));
                a_light += fabsf(dx) * gx + fabsf(dy) * gy;
                edge[(size_t)(8 + 2 * c) * HW + i] = w_light * signf_(dx) * gx;
                edge[(size_t)(9 + 2 * c) * HW + i] = w_light * signf_(dy) * gy;
                // d/d(G_d n_c) of |G_d dl_c| exp(-|G_d n_c|)
                edge[(size_t)(14 + 2 * c) * HW + i] = -w_light * fabsf(dx) * gx * signf_(nx);
                edge[(size_t)(15 + 2 * c) * HW + i] = -w_light * fabsf(dy) * gy * signf_(ny);
            }
        }
    }
    const float t0 = block_sum_256(a_base, s_part);
    __syncthreads();
    const float t1 = block_sum_256(a_rough, s_part);
    __syncthreads();
    const float t2 = block_sum_256(a_light, s_part);
    if (threadIdx.x == 0) {
        atomicAdd(sum_slot(sums3 + 0 * R3DG_SUM_SLOTS), t0);
        atomicAdd(sum_slot(sums3 + 1 * R3DG_SUM_SLOTS), t1);
        atomicAdd(sum_slot(sums3 + 2 * R3DG_SUM_SLOTS), t2);
    }
}

__global__ void __launch_bounds__(256)
s2_smooth_backward_kernel(int W, int H, const float* __restrict__ opacity, const float* __restrict__ feature,
                          const int* __restrict__ n_contrib, const float* __restrict__ image_mask,
                          const float* __restrict__ edge, int has_base, int has_rough, int has_light,
                          int accumulate_normal, float* __restrict__ dL_dopacity, float* __restrict__ dL_dfeature)
{
    const size_t HW = (size_t)W * H;
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < HW; i += (size_t)gridDim.x * 256) {
        const int y = (int)(i / W), x = (int)(i % W);
        // adjoint of the replicate-padded stencils: d[k] = dL / d rend_k at this pixel
        float d[10];
#pragma unroll
        for (int k = 0; k < 10; k++) d[k] = 0.f;
#pragma unroll
        for (int a = -1; a <= 1; a++) {
            const int qy = y + a;
            if (qy < 0 || qy >= H) continue;
            const float sy = s1_adj1(qy, y, H, 1.f, 2.f, 1.f), dy = s1_adj1(qy, y, H, -1.f, 0.f, 1.f);
#pragma unroll
            for (int b = -1; b <= 1; b++) {
                const int qx = x + b;
                if (qx < 0 || qx >= W) continue;
                const float sx = s1_adj1(qx, x, W, 1.f, 2.f, 1.f), dx = s1_adj1(qx, x, W, -1.f, 0.f, 1.f);
                const float wx = sy * dx * 0.125f, wy = dy * sx * 0.125f;
                const size_t q = (size_t)qy * W + qx;
                auto E = [&](int k) { return edge[(size_t)k * HW + q]; };
                if (has_base) {
#pragma unroll
                    for (int c = 0; c < 3; c++) d[c] = smooth_tap(d[c], wx, E(2 * c), wy, E(2 * c + 1));
                }
                if (has_rough) d[3] = smooth_tap(d[3], wx, E(6), wy, E(7));
                if (has_light) {
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        d[4 + c] = smooth_tap(d[4 + c], wx, E(8 + 2 * c), wy, E(9 + 2 * c));
                        d[7 + c] = smooth_tap(d[7 + c], wx, E(14 + 2 * c), wy, E(15 + 2 * c));
                    }
                }
            }
        }
        float f[10], gf[10];
#pragma unroll
        for (int c = 0; c < 10; c++) {
            const bool used = (c < 3) ? has_light : (c < 6) ? has_base : (c == 6) ? has_rough : has_light;
            f[c] = used ? feature[(size_t)(5 + c) * HW + i] : 0.f;
        }
        const float g_op = smooth_chain(has_base, has_rough, has_light, d, opacity[i], n_contrib[i],
                                        image_mask ? image_mask[i] : 1.f, f, gf);
        smooth_store(has_base, has_rough, has_light, accumulate_normal, gf, HW, i, dL_dfeature);
        dL_dopacity[i] += g_op;
    }
}

// ---- the three passes above as ONE kernel, streamed through registers (round 4) -------------------------------------------------
// maps -> edge -> backward move, per pixel, 10 divided maps out and back in, 20 adjoint inputs out and -- nine times, through the
// caches -- back in: 0.41 ms of the 2.0 ms DTU iteration (1600x1200).  None of that has to exist in HBM.  (A
// first fused version staged 32 x 8 pixel tiles + halo in 50 KB of LDS: 1.69x the loads for the halo of 2, two thirds empty
// second rounds, three barriers per 256 pixels, three workgroups per CU -- 0.148 ms where this one takes 0.102; deleted.)
// Nothing of the stencil needs a tile: lane = COLUMN, the wave walks DOWN the image.  The left / right neighbours of a value are one DPP lane shift away (wave_shr:1 / wave_shl:1 fold
// into the consuming instruction), the rows above live in registers:
//   per input row   13 maps of this column (divided / masked / sRGB-mapped as in s2_smooth_maps_kernel), their horizontal
//                   differences; with the two rows before: the Sobel pair of the row in the middle (gx from the three rows'
//                   differences, gy from the lane-shifted difference of the outer rows -- the reference expression, term by term),
//                   its 20 adjoint inputs;
//   two rows later  the adjoint of the row whose three edge rows are now complete: nine (row, lane shift) taps in the order of
//                   s2_smooth_backward_kernel, then the chain rule, written once.
// A wave owns 60 columns (lanes 2..61; the two lanes either side are the halo of the two stencil levels) x `rows` rows (+4 rows
// of run-in); replicate padding = clamped load coordinates.  No LDS, no barrier, every load a full 256-byte row segment.
constexpr int SS_OWN = 60;

__device__ __forceinline__ float lane_left(float v)      // the value of lane - 1 (wave_shr:1)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_right(float v)     // the value of lane + 1 (wave_shl:1)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xF, 0xF, true));
}

struct SmoothRow {            // what one pixel of a row contributes, as loaded
    float op, mk;
    int nc;
    float f[10];              // feature 5..14: normal 3 | base colour 3 | roughness | diffuse light 3
    float t[3];               // target
};

template <bool BASE, bool ROUGH, bool LIGHT, bool TARGET>
__device__ __forceinline__ SmoothRow smooth_load_row(size_t i, size_t HW, const float* __restrict__ opacity,
                                                     const float* __restrict__ feature, const int* __restrict__ n_contrib,
                                                     const float* __restrict__ gt, const float* __restrict__ image_mask)
{
    SmoothRow r;
    r.op = opacity[i];
    r.nc = n_contrib[i];
    r.mk = image_mask ? image_mask[i] : 1.f;
#pragma unroll
    for (int c = 0; c < 10; c++) {
        const bool used = (c < 3) ? LIGHT : (c < 6) ? BASE : (c == 6) ? ROUGH : LIGHT;
        r.f[c] = used ? feature[(size_t)(5 + c) * HW + i] : 0.f;
    }
#pragma unroll
    for (int c = 0; c < 3; c++) r.t[c] = (TARGET && (BASE || ROUGH)) ? gt[(size_t)c * HW + i] : 0.f;
    return r;
}

template <bool BASE, bool ROUGH, bool LIGHT>
__global__ void __launch_bounds__(256, 2)      // (two waves per SIMD: all three terms together hold ~250 registers)
s2_smooth_stream_kernel(int W, int H, int rows, int strips_x, const float* __restrict__ opacity,
                        const float* __restrict__ feature, const int* __restrict__ n_contrib, const float* __restrict__ gt,
                        const float* __restrict__ image_mask, float w_base, float w_rough, float w_light, int accumulate_normal,
                        float* dL_dopacity, float* dL_dfeature, float* __restrict__ sums3)
{
    // dL_dopacity and dL_dfeature (planes 5-7 with accumulate_normal) are READ-MODIFY-WRITE: every element is read before it is
    // written, by the one lane that owns the pixel, and never read again.  They carry no __restrict__ (rounds 4-5 passed each buffer
    // a second time under a second restrict-qualified name for the reads -- outside the language's guarantees): the
    // compiler must now keep a row's stores in front of the next row's loads from the same array, which is the order the source
    // has them in anyway -- the old values are requested at the top of part (3), a full row of arithmetic ahead of their use.
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int strip = wid % strips_x, y0 = (wid / strips_x) * rows;
    if (y0 >= H) return;                                   // (whole waves; nothing below synchronises a workgroup)
    const int y1 = min(y0 + rows, H);
    const int x = strip * SS_OWN - 2 + lane;               // this lane's column; lanes 0, 1, 62, 63 are halo
    const int xc = min(max(x, 0), W - 1);
    const bool own_col = lane >= 2 && lane < 2 + SS_OWN && x < W;
    const size_t HW = (size_t)W * H;
    constexpr int NM = 13;                                 // rend 0..9 (layout of s2_smooth_maps_kernel), target 10..12
    auto used = [](int m) { return m < 3 ? BASE : m == 3 ? ROUGH : m < 10 ? LIGHT : (BASE || ROUGH); };
    // weights with which the stencils of columns x-1, x, x+1 read column x (replicate padding folded in); 0 outside the image
    float sxw[3], dxw[3];
#pragma unroll
    for (int b = -1; b <= 1; b++) {
        const int qx = x + b;
        const bool in = qx >= 0 && qx < W;
        sxw[b + 1] = in ? s1_adj1(qx, x, W, 1.f, 2.f, 1.f) : 0.f;
        dxw[b + 1] = in ? s1_adj1(qx, x, W, -1.f, 0.f, 1.f) : 0.f;
    }
    // three rows in flight: values, right - left, adjoint inputs.  Row r lives in slot r % 3; the loop is unrolled by three so
    // that the slots are compile-time names (registers), not copies
    float V[3][NM], D[3][NM], E[3][20];
#pragma unroll
    for (int q = 0; q < 3; q++) {
#pragma unroll
        for (int m = 0; m < NM; m++) V[q][m] = D[q][m] = 0.f;
#pragma unroll
        for (int k = 0; k < 20; k++) E[q][k] = 0.f;
    }
    float a_base = 0.f, a_rough = 0.f, a_light = 0.f;
    const int nrows = y1 - y0;
    SmoothRow nxt = smooth_load_row<BASE, ROUGH, LIGHT, true>((size_t)min(max(y0 - 2, 0), H - 1) * W + xc, HW, opacity, feature,
                                                              n_contrib, gt, image_mask);
    auto step = [&](auto slot, int r) {
        constexpr int P2 = decltype(slot)::value, P1 = (P2 + 2) % 3, P0 = (P2 + 1) % 3;
        float (&v2)[NM] = V[P2], (&v0)[NM] = V[P0];
        float (&h2)[NM] = D[P2], (&h1)[NM] = D[P1], (&h0)[NM] = D[P0];
        float (&E2)[20] = E[P2], (&E1)[20] = E[P1], (&E0)[20] = E[P0];
        const int yy = y0 - 2 + r;                         // incoming row; edge row yy - 1; adjoint row yy - 2
        const SmoothRow in = nxt;
        // (unconditionally -- the row index is clamped anyway: under `if (r + 1 < total)` the compiler merges the loaded
        // registers with the untouched ones INSIDE the branch, i.e. waits for the loads it has just issued: 44 % of the
        // wave-cycles of the first version)
        nxt = smooth_load_row<BASE, ROUGH, LIGHT, true>((size_t)min(max(yy + 1, 0), H - 1) * W + xc, HW, opacity, feature,
                                                        n_contrib, gt, image_mask);
        const int py = yy - 2;
        const bool adjoint = r >= 4 && own_col && py < y1;
        // (1) the incoming row's maps
        {
            const float scale = in.nc > 0 ? 1.f / fmaxf(in.op, 1e-5f) : 0.f;
#pragma unroll
            for (int c = 0; c < 7; c++) {
                const float xv = in.f[3 + c] * scale;
                v2[c] = (c == 3 ? xv : srgb_clip(xv)) * in.mk;
            }
#pragma unroll
            for (int c = 0; c < 3; c++) {
                v2[7 + c] = in.f[c] * scale;
                v2[10 + c] = in.t[c];
            }
#pragma unroll
            for (int m = 0; m < NM; m++) h2[m] = used(m) ? lane_right(v2[m]) - lane_left(v2[m]) : 0.f;
        }
        // (2) the edge row in the middle of the three
        const int cy = yy - 1;
        if (r >= 2 && cy >= 0 && cy < H) {
            const bool own = own_col && cy >= y0 && cy < y1;
            auto sobel = [&](int m, float& gx_, float& gy_) {
                const float vd = v2[m] - v0[m];
                gx_ = ((h0[m]) + 2.f * (h1[m]) + (h2[m])) * 0.125f;
                gy_ = ((lane_left(vd)) + 2.f * (vd) + (lane_right(vd))) * 0.125f;
            };
            float ex[3], ey3[3];
            if (BASE || ROUGH) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    float gx1, gy1;
                    sobel(10 + c, gx1, gy1);
                    ex[c] = __expf(-fabsf(gx1));
                    ey3[c] = __expf(-fabsf(gy1));
                }
            }
            if (BASE) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    float dx, dy;
                    sobel(c, dx, dy);
                    if (own) a_base += fabsf(dx) * ex[c] + fabsf(dy) * ey3[c];
                    E2[2 * c] = w_base * signf_(dx) * ex[c];
                    E2[2 * c + 1] = w_base * signf_(dy) * ey3[c];
                }
            }
            if (ROUGH) {
                float dx, dy;
                sobel(3, dx, dy);
                const float sx = ex[0] + ex[1] + ex[2], sy = ey3[0] + ey3[1] + ey3[2];
                if (own) a_rough += fabsf(dx) * sx + fabsf(dy) * sy;
                E2[6] = w_rough * signf_(dx) * sx;
                E2[7] = w_rough * signf_(dy) * sy;
            }
            if (LIGHT) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    float dx, dy, nx, ny;
                    sobel(4 + c, dx, dy);
                    sobel(7 + c, nx, ny);
                    const float gx1 = __expf(-fabsf(nx)), gy1 = __expf(-fabsf(ny));
                    if (own) a_light += fabsf(dx) * gx1 + fabsf(dy) * gy1;
                    E2[8 + 2 * c] = w_light * signf_(dx) * gx1;
                    E2[9 + 2 * c] = w_light * signf_(dy) * gy1;
                    E2[14 + 2 * c] = -w_light * fabsf(dx) * gx1 * signf_(nx);
                    E2[15 + 2 * c] = -w_light * fabsf(dy) * gy1 * signf_(ny);
                }
            }
        }
        // (3) the adjoint of row yy - 2 and the chain rule
        if (r >= 4 && py < y1) {
            // the adjoint row's own pixel again (cache hits; requested here, used after the nine taps)
            const SmoothRow pr = smooth_load_row<BASE, ROUGH, LIGHT, false>((size_t)py * W + xc, HW, opacity, feature, n_contrib,
                                                                            gt, image_mask);
            // ... and what the two read-modify-write outputs hold there (requested now, used after the nine taps: read where
            // they are used, these loads were 44 % of the wave-cycles -- `s_waitcnt` with two waves per SIMD to cover it)
            const float old_dop = dL_dopacity[(size_t)py * W + xc];
            float old_n[3] = {0.f, 0.f, 0.f};
            if (LIGHT && accumulate_normal) {
#pragma unroll
                for (int c = 0; c < 3; c++) old_n[c] = dL_dfeature[(size_t)(5 + c) * HW + (size_t)py * W + xc];
            }
            float d[10];
#pragma unroll
            for (int k = 0; k < 10; k++) d[k] = 0.f;
#pragma unroll
            for (int a = -1; a <= 1; a++) {
                const int qy = py + a;
                if (qy < 0 || qy >= H) continue;
                const float sy = s1_adj1(qy, py, H, 1.f, 2.f, 1.f), dy = s1_adj1(qy, py, H, -1.f, 0.f, 1.f);
                const float (&Er)[20] = a < 0 ? E0 : a == 0 ? E1 : E2;
#pragma unroll
                for (int b = -1; b <= 1; b++) {
                    const float wx = sy * dxw[b + 1] * 0.125f, wy = dy * sxw[b + 1] * 0.125f;
                    auto tap = [&](int k) { return b < 0 ? lane_left(Er[k]) : b == 0 ? Er[k] : lane_right(Er[k]); };
                    if (BASE) {
#pragma unroll
                        for (int c = 0; c < 3; c++) d[c] = smooth_tap(d[c], wx, tap(2 * c), wy, tap(2 * c + 1));
                    }
                    if (ROUGH) d[3] = smooth_tap(d[3], wx, tap(6), wy, tap(7));
                    if (LIGHT) {
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            d[4 + c] = smooth_tap(d[4 + c], wx, tap(8 + 2 * c), wy, tap(9 + 2 * c));
                            d[7 + c] = smooth_tap(d[7 + c], wx, tap(14 + 2 * c), wy, tap(15 + 2 * c));
                        }
                    }
                }
            }
            if (adjoint) {
                const size_t i = (size_t)py * W + x;
                float gf[10];
                const float g_op = smooth_chain(BASE, ROUGH, LIGHT, d, pr.op, pr.nc, pr.mk, pr.f, gf);
                smooth_store(BASE, ROUGH, LIGHT, accumulate_normal, gf, HW, i, dL_dfeature, old_n);
                dL_dopacity[i] = old_dop + g_op;
            }
        }
    };
    // nrows + 4 steps, rounded up to whole rounds of three: a step past the end loads clamped rows and stores nothing, and a
    // straight-line round has no control-flow merge in it -- at a merge the compiler copies the prefetched row's registers, which
    // means waiting for everything in flight including the stores just issued (a write round trip per step)
    const int total = (nrows + 4 + 2) / 3 * 3;
    for (int r = 0; r < total; r += 3) {
        step(std::integral_constant<int, 0>{}, r);
        step(std::integral_constant<int, 1>{}, r + 1);
        step(std::integral_constant<int, 2>{}, r + 2);
    }
    a_base = wave_sum_to_lane63(a_base);
    a_rough = wave_sum_to_lane63(a_rough);
    a_light = wave_sum_to_lane63(a_light);
    if (lane == 63) {
        if (BASE) atomicAdd(sum_slot(sums3 + 0 * R3DG_SUM_SLOTS), a_base);
        if (ROUGH) atomicAdd(sum_slot(sums3 + 1 * R3DG_SUM_SLOTS), a_rough);
        if (LIGHT) atomicAdd(sum_slot(sums3 + 2 * R3DG_SUM_SLOTS), a_light);
    }
}

__global__ void __launch_bounds__(256)
s1_activate_backward_kernel(int P, const float* __restrict__ xyz, const float* __restrict__ scaling_raw,
                            const float* __restrict__ rotation_raw, const float* __restrict__ opacity_raw,
                            const float* __restrict__ normal_raw, const float* __restrict__ viewmatrix,
                            const float* __restrict__ dL_dfeatures, const float* __restrict__ dL_dscales,
                            const float* __restrict__ dL_drot, const float* __restrict__ dL_dopacity,
                            const float* __restrict__ dL_dmeans3D, float* __restrict__ g_xyz,
                            float* __restrict__ g_scaling, float* __restrict__ g_rotation,
                            float* __restrict__ g_opacity, float* __restrict__ g_normal)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const size_t i3 = 3 * (size_t)i, i4 = 4 * (size_t)i;
    const float* gf = dL_dfeatures + 5 * (size_t)i;
#pragma unroll
    for (int c = 0; c < 3; c++) g_scaling[i3 + c] = dL_dscales[i3 + c] * __expf(scaling_raw[i3 + c]);
    {
        const float q[4] = {rotation_raw[i4], rotation_raw[i4 + 1], rotation_raw[i4 + 2], rotation_raw[i4 + 3]};
        const float g[4] = {dL_drot[i4], dL_drot[i4 + 1], dL_drot[i4 + 2], dL_drot[i4 + 3]};
        const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        if (n > 1e-12f) {
            const float inv = 1.f / n;
            const float d = (q[0] * g[0] + q[1] * g[1] + q[2] * g[2] + q[3] * g[3]) * inv * inv;
#pragma unroll
            for (int c = 0; c < 4; c++) g_rotation[i4 + c] = (g[c] - q[c] * d) * inv;
        } else {
#pragma unroll
            for (int c = 0; c < 4; c++) g_rotation[i4 + c] = g[c] * 1e12f;
        }
    }
    {
        const float sg = sigmoidf_(opacity_raw[i]);
        g_opacity[i] = dL_dopacity[i] * sg * (1.f - sg);
    }
    {
        const float v[3] = {normal_raw[i3], normal_raw[i3 + 1], normal_raw[i3 + 2]};
        const float g[3] = {gf[0], gf[1], gf[2]};
        float o[3];
        normalize3_backward(v, 1e-3f, g, o);
        g_normal[i3] = o[0]; g_normal[i3 + 1] = o[1]; g_normal[i3 + 2] = o[2];
    }
    {
        const float depth = xyz[i3] * viewmatrix[2] + xyz[i3 + 1] * viewmatrix[6] + xyz[i3 + 2] * viewmatrix[10] +
                            viewmatrix[14];
        const float gd = gf[3] + 2.f * depth * gf[4];
        g_xyz[i3] = dL_dmeans3D[i3] + gd * viewmatrix[2];
        g_xyz[i3 + 1] = dL_dmeans3D[i3 + 1] + gd * viewmatrix[6];
        g_xyz[i3 + 2] = dL_dmeans3D[i3 + 2] + gd * viewmatrix[10];
    }
}

// ---- multi-group Adam -----------------------------------------------------------------------------------------------
struct AdamTable {
    r3dg_adam_group g[R3DG_ADAM_MAX_GROUPS];
    unsigned int first_block[R3DG_ADAM_MAX_GROUPS + 1];
    int n_groups;
};

// One float4 per thread and array (ADAM_UNROLL = 1) with NONTEMPORAL accesses for what is touched once per iteration -- the
// gradient and both moments: measured cold (tools/kbench_adam.py: the last-level cache evicted between launches, 958 MB per launch
// at 300k Gaussians): 0.180 ms = 5.33 TB/s = 0.67 of the HBM peak with plain accesses (rounds 1-5), 0.154 ms = 6.21 TB/s = 0.78
// with nontemporal ones -- the rate the guide measures as achievable for a streaming kernel.  More floats per thread do NOT help
// (2 / 4 / 8 float4 per thread and array, all loads issued first: 0.164 / 0.168 / 0.178 ms -- fewer, longer workgroups leave a
// longer tail); the knob stays for the A/B (tools/build_variant.py, -DR3DG_ADAM_UNROLL=n).
#ifndef R3DG_ADAM_UNROLL
#define R3DG_ADAM_UNROLL 1
#endif
#ifndef R3DG_ADAM_NT
#define R3DG_ADAM_NT 1              // the moments are read and written ONCE per iteration: nontemporal accesses
#endif
constexpr int ADAM_UNROLL = R3DG_ADAM_UNROLL;
constexpr int ADAM_BLOCK_FLOATS = 1024 * ADAM_UNROLL;

__device__ __forceinline__ float4 adam_load(const float* p, bool nt)
{
    if (nt && R3DG_ADAM_NT) {
        const float4* q = reinterpret_cast<const float4*>(p);
        return make_float4(__builtin_nontemporal_load(&q->x), __builtin_nontemporal_load(&q->y), __builtin_nontemporal_load(&q->z),
                           __builtin_nontemporal_load(&q->w));
    }
    return *reinterpret_cast<const float4*>(p);
}
__device__ __forceinline__ void adam_store(float* p, const float (&v)[4], bool nt)
{
    if (nt && R3DG_ADAM_NT) {
        float4* q = reinterpret_cast<float4*>(p);
        __builtin_nontemporal_store(v[0], &q->x);
        __builtin_nontemporal_store(v[1], &q->y);
        __builtin_nontemporal_store(v[2], &q->z);
        __builtin_nontemporal_store(v[3], &q->w);
        return;
    }
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}

__global__ void __launch_bounds__(256)
adam_kernel(AdamTable t, float beta1, float beta2, float eps, float bias1, float inv_sqrt_bias2, float grad_scale,
            const float* __restrict__ skip_flag)
{
    // the gradients belong to a frame the bounded forward dropped on the device (r3dg_rasterize_forward_begin_bounded):
    // no update; the host hears about it later and does not count the step
    if (skip_flag != nullptr && *skip_flag != 0.0f) return;
    int gi = 0;
#pragma unroll 1
    while (gi + 1 < t.n_groups && blockIdx.x >= t.first_block[gi + 1]) gi++;
    const r3dg_adam_group grp = t.g[gi];
    const size_t block_base = (size_t)(blockIdx.x - t.first_block[gi]) * ADAM_BLOCK_FLOATS + threadIdx.x * 4;
    float* __restrict__ p = grp.param;
    const float* __restrict__ g = grp.grad;
    float* __restrict__ m = grp.exp_avg;
    float* __restrict__ v = grp.exp_avg_sq;
    float pv[ADAM_UNROLL][4], gv[ADAM_UNROLL][4], mv[ADAM_UNROLL][4], vv[ADAM_UNROLL][4];
    // every full float4 of this thread first (four arrays x ADAM_UNROLL loads in flight), the ragged tail element-wise
#pragma unroll
    for (int u = 0; u < ADAM_UNROLL; u++) {
        const size_t base = block_base + (size_t)u * 1024;
        if (base + 4 <= grp.n) {
            *reinterpret_cast<float4*>(pv[u]) = *reinterpret_cast<const float4*>(p + base);
            *reinterpret_cast<float4*>(gv[u]) = adam_load(g + base, true);
            *reinterpret_cast<float4*>(mv[u]) = adam_load(m + base, true);
            *reinterpret_cast<float4*>(vv[u]) = adam_load(v + base, true);
        } else {
            for (int k = 0; k < 4; k++)
                if (base + k < grp.n) { pv[u][k] = p[base + k]; gv[u][k] = g[base + k]; mv[u][k] = m[base + k]; vv[u][k] = v[base + k]; }
        }
    }
#pragma unroll
    for (int u = 0; u < ADAM_UNROLL; u++) {
        const size_t base = block_base + (size_t)u * 1024;
        if (base >= grp.n) continue;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            // two learning rates per group: elements whose index modulo `period` is below `split` use lr, the rest lr_tail
            // (one [P,16,3] SH tensor = dc columns + rest columns with different rates, gaussian_model.py:470-471)
            float lr = grp.lr;
            if (grp.period != 0 && ((unsigned int)(base + k) % grp.period) >= grp.split) lr = grp.lr_tail;   // n < 2^32
            float gk = gv[u][k] * grad_scale;                  // e.g. 1 / world_size after a sum all-reduce
            mv[u][k] = mv[u][k] + (gk - mv[u][k]) * (1.f - beta1);
            vv[u][k] = beta2 * vv[u][k] + (1.f - beta2) * gk * gk;
            const float denom = sqrtf(vv[u][k]) * inv_sqrt_bias2 + eps;
            pv[u][k] -= (lr / bias1) * (mv[u][k] / denom);
        }
        if (base + 4 <= grp.n) {
            *reinterpret_cast<float4*>(p + base) = *reinterpret_cast<float4*>(pv[u]);
            adam_store(m + base, mv[u], true);
            adam_store(v + base, vv[u], true);
        } else {
            for (int k = 0; k < 4; k++)
                if (base + k < grp.n) { p[base + k] = pv[u][k]; m[base + k] = mv[u][k]; v[base + k] = vv[u][k]; }
        }
    }
}

// ---- launchers ------------------------------------------------------------------------------------------------------
void launch_s2_activate(hipStream_t s, int P, const float* xyz, const float* scaling_raw, const float* rotation_raw,
                        const float* opacity_raw, const float* normal_raw, const float* base_raw,
                        const float* rough_raw, const float* campos, float* scales, float* rot, float* opacity,
                        float* normal, float* base_color, float* roughness, float* viewdirs, const float* viewmatrix,
                        float* features, int n_env, const float* env_raw, float* env, float* zero, int n_zero)
{
    ActivateSide side;
    side.nb_main = (P + 255) / 256;
    side.n_env = env_raw != nullptr && env != nullptr ? n_env : 0;
    side.env_raw = env_raw;
    side.env = env;
    side.zero = zero;
    side.n_zero = zero != nullptr ? n_zero : 0;
    const int work = side.n_env > side.n_zero ? side.n_env : side.n_zero;
    const int nb_side = work > 0 ? (work + 255) / 256 < 8 ? (work + 255) / 256 : 8 : 0;
    s2_activate_kernel<<<side.nb_main + nb_side, 256, 0, s>>>(P, xyz, scaling_raw, rotation_raw, opacity_raw, normal_raw,
                                                              base_raw, rough_raw, campos, scales, rot, opacity, normal,
                                                              base_color, roughness, viewdirs, viewmatrix, features, side);
    check_launch(s, false, "s2_activate_kernel");
}

void launch_s2_pack(hipStream_t s, int P, const float* xyz, const float* viewmatrix, const float* normal,
                    const float* base_color, const float* roughness, const float* shade_out, float* features,
                    float* light_l1_sum)
{
    s2_pack_features_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, xyz, viewmatrix, normal, base_color, roughness,
                                                            shade_out, features, light_l1_sum);
    check_launch(s, false, "s2_pack_features_kernel");
}

void launch_s2_unpack(hipStream_t s, int P, const float* dL_dfeatures, const float* shade_out, float light_weight,
                      float* dL_dpbr, float* dL_ddiffuse, float* block_absmax, float* light_l1_sum)
{
    s2_unpack_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, dL_dfeatures, shade_out, light_weight, dL_dpbr, dL_ddiffuse,
                                                    block_absmax, light_l1_sum);
    check_launch(s, false, "s2_unpack_kernel");
}

void launch_s2_activate_backward(hipStream_t s, int P, const float* xyz, const float* scaling_raw,
                                 const float* rotation_raw, const float* opacity_raw, const float* normal_raw,
                                 const float* base_raw, const float* rough_raw, const float* viewmatrix,
                                 const float* campos, const float* dL_dfeatures, const float* dL_dbase_shade,
                                 const float* dL_drough_shade, const float* dL_dviewdirs, const float* dL_dscales,
                                 const float* dL_drot, const float* dL_dopacity, const float* dL_dmeans3D, float* g_xyz,
                                 float* g_scaling, float* g_rotation, float* g_opacity, float* g_normal, float* g_base,
                                 float* g_rough, int He, int We, const float* env_raw, const float* env, float* dL_denv,
                                 float w_tv, float* g_env_raw, float* tv_sum, int consume)
{
    EnvBackward job;
    job.He = He; job.We = We; job.raw = env_raw; job.env = env; job.dL_denv = dL_denv; job.w_tv = w_tv; job.g_raw = g_env_raw;
    job.tv_sum = tv_sum; job.consume = consume;
    const int nb_main = (P + 255) / 256;
    const int nb_env = env_raw != nullptr ? (He * We * 3 + 255) / 256 : 0;
    s2_activate_backward_kernel<<<nb_main + nb_env, 256, 0, s>>>(
        P, xyz, scaling_raw, rotation_raw, opacity_raw, normal_raw, base_raw, rough_raw, viewmatrix, campos,
        dL_dfeatures, dL_dbase_shade, dL_drough_shade, dL_dviewdirs, dL_dscales, dL_drot, dL_dopacity, dL_dmeans3D, g_xyz,
        g_scaling, g_rotation, g_opacity, g_normal, g_base, g_rough, nb_main, job);
    check_launch(s, false, "s2_activate_backward_kernel");
}

void launch_s2_normals_srgb(hipStream_t s, int W, int H, const float* vm, float focal_x, float focal_y, float cx, float cy,
                            const float* opacity, const float* depths, float* normals, float* surface_xyz, const float* feature,
                            const int* n_contrib, const float* bg, float* srgb)
{
    dim3 grid((W + 63) / 64, (H + 3) / 4);
    s2_normals_srgb_kernel<<<grid, 256, 0, s>>>(W, H, focal_x, focal_y, cx, cy, vm, opacity, depths, normals, surface_xyz,
                                                feature, n_contrib, bg, srgb);
    check_launch(s, false, "s2_normals_srgb_kernel");
}

void launch_s2_pbr_srgb(hipStream_t s, int HW, const float* opacity, const float* feature, const int* n_contrib,
                        const float* bg, float* srgb)
{
    s2_pbr_srgb_kernel<<<(HW + 255) / 256, 256, 0, s>>>(HW, opacity, feature, n_contrib, bg, srgb);
    check_launch(s, false, "s2_pbr_srgb_kernel");
}

// grid of the stage-2 loss kernel: 768 workgroups measured best (1536: +4 us, 2500: +5 us even with the slot-spread sums)
constexpr int LOSS_BLOCKS = 768;

void launch_s2_loss(hipStream_t s, int HW, const float* image, const float* opacity, const float* feature,
                    const float* pseudo_normal, const int* n_contrib, const float* gt, const float* bg,
                    const float* image_mask, float w_l1, float w_pbr, float w_normal, const float* extra_dimage,
                    const float* extra_dsrgb, float* dL_dimage, float* dL_dopacity, float* dL_dfeature, float* sums,
                    int sparse)
{
    if (sparse)
        s2_loss_kernel<true><<<min((HW + 255) / 256, LOSS_BLOCKS), 256, 0, s>>>(
            HW, image, opacity, feature, pseudo_normal, n_contrib, gt, bg, image_mask, w_l1, w_pbr, w_normal, extra_dimage,
            extra_dsrgb, dL_dimage, dL_dopacity, dL_dfeature, sums);
    else
        s2_loss_kernel<false><<<min((HW + 255) / 256, LOSS_BLOCKS), 256, 0, s>>>(
            HW, image, opacity, feature, pseudo_normal, n_contrib, gt, bg, image_mask, w_l1, w_pbr, w_normal, extra_dimage,
            extra_dsrgb, dL_dimage, dL_dopacity, dL_dfeature, sums);
    check_launch(s, false, "s2_loss_kernel");
}

void launch_s2_env_backward(hipStream_t s, int He, int We, const float* raw, const float* env, float* dL_denv,
                            float w_tv, float* g_raw, float* tv_sum, int consume)
{
    EnvBackward job;
    job.He = He; job.We = We; job.raw = raw; job.env = env; job.dL_denv = dL_denv; job.w_tv = w_tv; job.g_raw = g_raw;
    job.tv_sum = tv_sum; job.consume = consume;
    s2_env_backward_kernel<<<(He * We * 3 + 255) / 256, 256, 0, s>>>(job);
    check_launch(s, false, "s2_env_backward_kernel");
}

void launch_s2_smooth_forward(hipStream_t s, int W, int H, const float* opacity, const float* feature, const int* n_contrib,
                              const float* gt, const float* image_mask, float w_base, float w_rough, float w_light,
                              float* scratch, float* sums3)
{
    const long long HW = (long long)W * H;
    float* rend = scratch;
    float* edge = scratch + 10 * HW;
    s2_smooth_maps_kernel<<<(int)((HW + 255) / 256), 256, 0, s>>>((int)HW, opacity, feature, n_contrib, image_mask, rend);
    check_launch(s, false, "s2_smooth_maps_kernel");
    s2_smooth_edge_kernel<<<(int)min((HW + 255) / 256, (long long)2048), 256, 0, s>>>(W, H, rend, gt, w_base, w_rough, w_light,
                                                                                 edge, sums3);
    check_launch(s, false, "s2_smooth_edge_kernel");
}

void launch_s2_smooth_backward(hipStream_t s, int W, int H, const float* opacity, const float* feature, const int* n_contrib,
                               const float* image_mask, const float* scratch, int has_base, int has_rough, int has_light,
                               int accumulate_normal, float* dL_dopacity, float* dL_dfeature)
{
    const long long HW = (long long)W * H;
    s2_smooth_backward_kernel<<<(int)min((HW + 255) / 256, (long long)2048), 256, 0, s>>>(
        W, H, opacity, feature, n_contrib, image_mask, scratch + 10 * HW, has_base, has_rough, has_light, accumulate_normal,
        dL_dopacity, dL_dfeature);
    check_launch(s, false, "s2_smooth_backward_kernel");
}

template <bool BASE, bool ROUGH, bool LIGHT>
static void launch_s2_smooth_stream_t(hipStream_t s, int W, int H, int rows, const float* opacity, const float* feature,
                                      const int* n_contrib, const float* gt, const float* image_mask, float w_base, float w_rough,
                                      float w_light, int accumulate_normal, float* dL_dopacity, float* dL_dfeature, float* sums3)
{
    const int strips_x = (W + SS_OWN - 1) / SS_OWN, strips_y = (H + rows - 1) / rows;
    const int waves = strips_x * strips_y;
    s2_smooth_stream_kernel<BASE, ROUGH, LIGHT><<<(waves + 3) / 4, 256, 0, s>>>(
        W, H, rows, strips_x, opacity, feature, n_contrib, gt, image_mask, w_base, w_rough, w_light, accumulate_normal,
        dL_dopacity, dL_dfeature, sums3);
}

void launch_s2_smooth_fused(hipStream_t s, int W, int H, const float* opacity, const float* feature, const int* n_contrib,
                            const float* gt, const float* image_mask, float w_base, float w_rough, float w_light,
                            int accumulate_normal, float* dL_dopacity, float* dL_dfeature, float* sums3)
{
    // rows per wave: every wave pays 4 rows of run-in; one full round of two waves per SIMD is ~2048 waves
    static const int rows_env = getenv("R3DG_SMOOTH_ROWS") ? atoi(getenv("R3DG_SMOOTH_ROWS")) : 0;
    const int strips_x = (W + SS_OWN - 1) / SS_OWN;
    int rows = rows_env > 0 ? rows_env : (int)std::min<long long>(32, std::max<long long>(8, ((long long)strips_x * H + 2047) / 2048));
    rows = std::max(1, std::min(rows, H));
    const int sel = (w_base != 0.f ? 1 : 0) | (w_rough != 0.f ? 2 : 0) | (w_light != 0.f ? 4 : 0);
#define R3DG_SMOOTH_CASE(n, B, R, L)                                                                                        \
    case n:                                                                                                                 \
        launch_s2_smooth_stream_t<B, R, L>(s, W, H, rows, opacity, feature, n_contrib, gt, image_mask, w_base, w_rough,     \
                                           w_light, accumulate_normal, dL_dopacity, dL_dfeature, sums3);                    \
        break;
    switch (sel) {
        R3DG_SMOOTH_CASE(1, true, false, false)
        R3DG_SMOOTH_CASE(2, false, true, false)
        R3DG_SMOOTH_CASE(3, true, true, false)
        R3DG_SMOOTH_CASE(4, false, false, true)
        R3DG_SMOOTH_CASE(5, true, false, true)
        R3DG_SMOOTH_CASE(6, false, true, true)
        R3DG_SMOOTH_CASE(7, true, true, true)
        default: return;                             // no term: nothing to add
    }
#undef R3DG_SMOOTH_CASE
    check_launch(s, false, "s2_smooth_stream_kernel");
}

void launch_s1_pack(hipStream_t s, int P, const float* xyz, const float* viewmatrix, const float* normal, float* features)
{
    s1_pack_features_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, xyz, viewmatrix, normal, features);
    check_launch(s, false, "s1_pack_features_kernel");
}

void launch_s1_edge(hipStream_t s, int W, int H, const float* feature, const float* opacity, const int* n_contrib,
                    const float* gt, float* edge_g, float* sum_out)
{
    const long long HW = (long long)W * H;
    s1_edge_kernel<<<(int)min((HW + 255) / 256, (long long)4096), 256, 0, s>>>(W, H, feature, opacity, n_contrib, gt, edge_g,
                                                                           sum_out);
    check_launch(s, false, "s1_edge_kernel");
}

void launch_s1_loss(hipStream_t s, int W, int H, const float* image, const float* opacity, const float* feature,
                    const float* pseudo_normal, const int* n_contrib, const float* gt, const float* image_mask, float w_l1,
                    float w_entropy, float w_normal, float w_smooth, float w_var, const float* extra_dimage,
                    const float* edge_g, float* dL_dimage, float* dL_dopacity, float* dL_dfeature, float* sums)
{
    const long long HW = (long long)W * H;
    s1_loss_kernel<<<(int)min((HW + 255) / 256, (long long)2048), 256, 0, s>>>(
        W, H, image, opacity, feature, pseudo_normal, n_contrib, gt, image_mask, w_l1, w_entropy, w_normal, w_smooth, w_var,
        extra_dimage, edge_g, dL_dimage, dL_dopacity, dL_dfeature, sums);
    check_launch(s, false, "s1_loss_kernel");
}

void launch_s1_activate_backward(hipStream_t s, int P, const float* xyz, const float* scaling_raw,
                                 const float* rotation_raw, const float* opacity_raw, const float* normal_raw,
                                 const float* viewmatrix, const float* dL_dfeatures, const float* dL_dscales,
                                 const float* dL_drot, const float* dL_dopacity, const float* dL_dmeans3D, float* g_xyz,
                                 float* g_scaling, float* g_rotation, float* g_opacity, float* g_normal)
{
    s1_activate_backward_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, xyz, scaling_raw, rotation_raw, opacity_raw,
                                                              normal_raw, viewmatrix, dL_dfeatures, dL_dscales, dL_drot,
                                                              dL_dopacity, dL_dmeans3D, g_xyz, g_scaling, g_rotation,
                                                              g_opacity, g_normal);
    check_launch(s, false, "s1_activate_backward_kernel");
}

void launch_adam(hipStream_t s, int n_groups, const r3dg_adam_group* groups, float beta1, float beta2, float eps,
                 int step, float grad_scale, const float* skip_flag)
{
    AdamTable t;
    t.n_groups = n_groups;
    unsigned int blocks = 0;
    for (int i = 0; i < n_groups; i++) {
        t.g[i] = groups[i];
        t.first_block[i] = blocks;
        blocks += (unsigned int)((groups[i].n + ADAM_BLOCK_FLOATS - 1) / ADAM_BLOCK_FLOATS);
    }
    t.first_block[n_groups] = blocks;
    if (blocks == 0) return;
    const double b1 = 1.0 - pow((double)beta1, (double)step), b2 = 1.0 - pow((double)beta2, (double)step);
    adam_kernel<<<blocks, 256, 0, s>>>(t, beta1, beta2, eps, (float)b1, (float)(1.0 / sqrt(b2)), grad_scale, skip_flag);
    check_launch(s, false, "adam_kernel");
}

}  // namespace r3dg
