// Fused per-Gaussian shading integral (the LIVE stage-2 model) for gfx950, forward and backward.
// Reference semantics: rendering_equation neilf.py:339-371, GGX_specular :374-407, eval_sh sh_utils.py:71-128,
// DirectLightMap.direct_light direct_light_map.py:70-83 / EnvLight.direct_light envmap.py:35-53.
//
// The reference evaluates this with dozens of unfused elementwise passes over [P,K,3] temporaries (230 MB each at
// K=64) plus autograd saved tensors.  Here ONE kernel reads each cached sample once -- dirs [P,K,3], visibility
// [P,K], areas [P,K], 16 B per sample -- and writes 19 floats per Gaussian:
//   * one wave per Gaussian, lanes across the K samples (K=64: exactly one sample per lane; K=384: six), so the
//     three [P,K,*] caches are read with fully coalesced 256-768 B wave accesses;
//   * the environment texture (activated, lat-long) is staged in LDS when it fits (16x32x3 fp32 = 6 KB) and
//     sampled bilinearly from there; in the backward its gradient is accumulated in a second LDS copy with
//     ds_add_f32 and flushed with one global atomic per texel per block (blocks are persistent / grid-strided);
//   * the K-mean of the 19 forward outputs / the 55 per-Gaussian gradients (48 SH + 3 albedo + 1 roughness +
//     3 view direction) is a transposing wave reduction (see rasterizer_render_bwd.hip) ending in one store.
#include "common.hpp"
#include "wave_reduce.hpp"

namespace r3dg {

constexpr float kPi = 3.14159265358979323846f;
constexpr int SHADE_WAVES = 4;               // Gaussians in flight per block
constexpr int ENV_LDS_MAX = 12288;           // floats (48 KB) -- larger maps are sampled from global/L2
constexpr int SHADE_NOUT = 19;               // pbr3 diffuse3 specular3 lights3 local3 global3 vis1

// ---- real SH basis, degree 3, reference sign convention (sh_utils.py:92-127) ----
__device__ __forceinline__ void sh_basis16(float x, float y, float z, int M, float (&Y)[16])
{
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    Y[0] = C0;
#pragma unroll
    for (int i = 1; i < 16; i++) Y[i] = 0.f;
    if (M > 1) {
        Y[1] = -C1 * y; Y[2] = C1 * z; Y[3] = -C1 * x;
        if (M > 4) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            Y[4] = 1.0925484305920792f * xy;
            Y[5] = -1.0925484305920792f * yz;
            Y[6] = 0.31539156525252005f * (2.0f * zz - xx - yy);
            Y[7] = -1.0925484305920792f * xz;
            Y[8] = 0.5462742152960396f * (xx - yy);
            if (M > 9) {
                Y[9] = -0.5900435899266435f * y * (3.f * xx - yy);
                Y[10] = 2.890611442640554f * xy * z;
                Y[11] = -0.4570457994644658f * y * (4.f * zz - xx - yy);
                Y[12] = 0.3731763325901154f * z * (2.f * zz - 3.f * xx - 3.f * yy);
                Y[13] = -0.4570457994644658f * x * (4.f * zz - xx - yy);
                Y[14] = 1.445305721320277f * z * (xx - yy);
                Y[15] = -0.5900435899266435f * x * (xx - 3.f * yy);
            }
        }
    }
}

struct EnvTap {
    int idx[4];      // texel index (y*We + x), -1 when out of range (zero padding)
    float w[4];
};

// lat-long lookup coordinates + bilinear taps (direct_light_map.py:70-83; grid_sample align_corners=True, zeros)
__device__ __forceinline__ EnvTap env_taps(float dx, float dy, float dz, const float* __restrict__ tr, int He, int We)
{
    if (tr != nullptr) {
        const float tx = dx * tr[0] + dy * tr[1] + dz * tr[2];
        const float ty = dx * tr[3] + dy * tr[4] + dz * tr[5];
        const float tz = dx * tr[6] + dy * tr[7] + dz * tr[8];
        dx = tx; dy = ty; dz = tz;
    }
    const float phi = acosf(dz) - 1e-6f;
    const float theta = atan2f(dy, dx);
    const float qy = (phi / kPi) * 2.f - 1.f;
    const float qx = -theta / kPi;
    const float ix = (qx + 1.f) * 0.5f * (float)(We - 1);
    const float iy = (qy + 1.f) * 0.5f * (float)(He - 1);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float wx1 = ix - x0f, wy1 = iy - y0f, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    const int x0 = (int)x0f, y0 = (int)y0f;
    EnvTap t;
    const int xs[2] = {x0, x0 + 1}, ys[2] = {y0, y0 + 1};
    const float wxs[2] = {wx0, wx1}, wys[2] = {wy0, wy1};
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const bool ok = xs[b] >= 0 && xs[b] <= We - 1 && ys[a] >= 0 && ys[a] <= He - 1;
            t.idx[a * 2 + b] = ok ? ys[a] * We + xs[b] : -1;
            t.w[a * 2 + b] = wys[a] * wxs[b];
        }
    return t;
}

struct SampleFwd {
    float local[3], glob[3], lin[3], transport[3];
    float spec, ndi, area_ndi;
    // intermediates kept for the backward
    float Y[16], shsum[3];
    float L[3], Hh[3], ulen, NoL, NoH, VoH, rawNoH, rawVoH, nom0, nom1, nom2, nomr, frac0, p2;
    EnvTap taps;
    float vis;
};

struct GaussFwd {            // wave-uniform per-Gaussian quantities
    float base[3], r, n[3], V[3], vlen, N[3], NoV, rawNoV, a, a2, kk;
    float v_raw[3];
};

__device__ __forceinline__ void gauss_setup(GaussFwd& G, const float* __restrict__ base_color,
                                            const float* __restrict__ roughness, const float* __restrict__ normals,
                                            const float* __restrict__ viewdirs, int g)
{
#pragma unroll
    for (int c = 0; c < 3; c++) {
        G.base[c] = base_color[3 * g + c];
        G.n[c] = normals[3 * g + c];
        G.v_raw[c] = viewdirs[3 * g + c];
    }
    G.r = roughness[g];
    G.vlen = fmaxf(sqrtf(G.v_raw[0] * G.v_raw[0] + G.v_raw[1] * G.v_raw[1] + G.v_raw[2] * G.v_raw[2]), 1e-12f);
    const float nlen = fmaxf(sqrtf(G.n[0] * G.n[0] + G.n[1] * G.n[1] + G.n[2] * G.n[2]), 1e-12f);
    float N0[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        G.V[c] = G.v_raw[c] / G.vlen;
        N0[c] = G.n[c] / nlen;
    }
    const float d0 = G.V[0] * N0[0] + G.V[1] * N0[1] + G.V[2] * N0[2];
    const float sgn = d0 > 0.f ? 1.f : (d0 < 0.f ? -1.f : 0.f);
#pragma unroll
    for (int c = 0; c < 3; c++) G.N[c] = N0[c] * sgn;
    G.rawNoV = G.N[0] * G.V[0] + G.N[1] * G.V[1] + G.N[2] * G.V[2];
    G.NoV = fminf(fmaxf(G.rawNoV, 1e-6f), 1.f);
    G.a = G.r * G.r;
    G.a2 = G.a * G.a;
    G.kk = (G.a + 2.f * G.r + 1.0f) / 8.0f;
}

template <bool ENV_LDS>
__device__ __forceinline__ void shade_sample(SampleFwd& s, const GaussFwd& G, const float* __restrict__ sh /*[M*3]*/,
                                             int M, float dx, float dy, float dz, float vis, float area,
                                             const float* __restrict__ env, const float* s_env,
                                             const float* __restrict__ tr, int He, int We)
{
    // environment light (global) * visibility
    s.taps = env_taps(dx, dy, dz, tr, He, We);
    s.vis = vis;
    float e[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; t++) {
        if (s.taps.idx[t] >= 0) {
            const float* px = ENV_LDS ? (s_env + 3 * s.taps.idx[t]) : (env + 3 * (size_t)s.taps.idx[t]);
            e[0] += px[0] * s.taps.w[t];
            e[1] += px[1] * s.taps.w[t];
            e[2] += px[2] * s.taps.w[t];
        }
    }
    // local incident light: max(SH(d), 0)
    sh_basis16(dx, dy, dz, M, s.Y);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (i < M) acc += s.Y[i] * sh[i * 3 + c];
        s.shsum[c] = acc;
        s.local[c] = fmaxf(acc, 0.f);
        s.glob[c] = e[c] * vis;
        s.lin[c] = s.local[c] + s.glob[c];
    }
    s.ndi = fmaxf(G.n[0] * dx + G.n[1] * dy + G.n[2] * dz, 0.f);
    s.area_ndi = area * s.ndi;
    // GGX specular (neilf.py:374-407)
    const float dlen = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-12f);
    s.L[0] = dx / dlen; s.L[1] = dy / dlen; s.L[2] = dz / dlen;
    float u[3];
#pragma unroll
    for (int c = 0; c < 3; c++) u[c] = (s.L[c] + G.V[c]) / 2.0f;
    s.ulen = fmaxf(sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1e-12f);
#pragma unroll
    for (int c = 0; c < 3; c++) s.Hh[c] = u[c] / s.ulen;
    s.NoL = fminf(fmaxf(G.N[0] * s.L[0] + G.N[1] * s.L[1] + G.N[2] * s.L[2], 1e-6f), 1.f);
    s.rawNoH = G.N[0] * s.Hh[0] + G.N[1] * s.Hh[1] + G.N[2] * s.Hh[2];
    s.NoH = fminf(fmaxf(s.rawNoH, 1e-6f), 1.f);
    s.rawVoH = G.V[0] * s.Hh[0] + G.V[1] * s.Hh[1] + G.V[2] * s.Hh[2];
    s.VoH = fminf(fmaxf(s.rawVoH, 1e-6f), 1.f);
    const float FMi = (-5.55473f * s.VoH - 6.98316f) * s.VoH;
    s.p2 = exp2f(FMi);
    s.frac0 = 0.04f + 0.96f * s.p2;
    const float frac = s.frac0 * G.a2;
    s.nom0 = s.NoH * s.NoH * (G.a2 - 1.f) + 1.f;
    s.nom1 = G.NoV * (1.f - G.kk) + G.kk;
    s.nom2 = s.NoL * (1.f - G.kk) + G.kk;
    s.nomr = 4.f * kPi * s.nom0 * s.nom0 * s.nom1 * s.nom2;
    const float nom = fminf(fmaxf(s.nomr, 1e-6f), 4.f * kPi);
    s.spec = frac / nom;
#pragma unroll
    for (int c = 0; c < 3; c++) s.transport[c] = s.lin[c] * s.area_ndi;
}

template <bool ENV_LDS>
__global__ void __launch_bounds__(64 * SHADE_WAVES)
shade_forward_kernel(int P, int K, int M, const float* __restrict__ base_color, const float* __restrict__ roughness,
                     const float* __restrict__ normals, const float* __restrict__ viewdirs,
                     const float* __restrict__ incidents, const float* __restrict__ env, int He, int We,
                     const float* __restrict__ tr, const float* __restrict__ visibility,
                     const float* __restrict__ dirs, const float* __restrict__ areas, float* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    float* s_env = s_mem;
    if (ENV_LDS) {
        for (int i = threadIdx.x; i < He * We * 3; i += blockDim.x) s_env[i] = env[i];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chan = transposed_channel<32>(lane);
    const bool owner = transposed_owner<32>(lane) && chan < SHADE_NOUT;
    const float invK = 1.0f / (float)K;
    for (int g0 = blockIdx.x * SHADE_WAVES + wave; g0 < P; g0 += gridDim.x * SHADE_WAVES) {
        const int g = __builtin_amdgcn_readfirstlane(g0);
        GaussFwd G;
        gauss_setup(G, base_color, roughness, normals, viewdirs, g);
        const float* sh = incidents + (size_t)g * M * 3;
        float v[32];
#pragma unroll
        for (int i = 0; i < 32; i++) v[i] = 0.f;
        for (int k = lane; k < K; k += 64) {
            const size_t o = (size_t)g * K + k;
            SampleFwd s;
            shade_sample<ENV_LDS>(s, G, sh, M, dirs[3 * o], dirs[3 * o + 1], dirs[3 * o + 2], visibility[o], areas[o],
                                  env, s_env, tr, He, We);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float fd = G.base[c] / kPi;
                v[c] += (fd + s.spec) * s.transport[c];      // pbr
                v[3 + c] += s.transport[c];                  // diffuse_light
                v[6 + c] += s.spec * s.transport[c];         // specular
                v[9 + c] += s.lin[c];                        // incident_lights mean
                v[12 + c] += s.local[c];
                v[15 + c] += s.glob[c];
            }
            v[18] += s.vis;
        }
        const float total = transpose_reduce<32, true>(v);
        if (owner) out[(size_t)g * SHADE_NOUT + chan] = total * invK;
    }
}

// Backward: gradients of sum(pbr*g_pbr) + sum(diffuse_light*g_diff) w.r.t. base_color, roughness, viewdirs,
// incidents and the (activated) environment texture.  normals / dirs / visibility carry no gradient in the
// reference (normal.detach(), cached samples).
template <bool ENV_LDS>
__global__ void __launch_bounds__(64 * SHADE_WAVES)
shade_backward_kernel(int P, int K, int M, const float* __restrict__ base_color, const float* __restrict__ roughness,
                      const float* __restrict__ normals, const float* __restrict__ viewdirs,
                      const float* __restrict__ incidents, const float* __restrict__ env, int He, int We,
                      const float* __restrict__ tr, const float* __restrict__ visibility,
                      const float* __restrict__ dirs, const float* __restrict__ areas,
                      const float* __restrict__ g_pbr, const float* __restrict__ g_diff, float* __restrict__ d_base,
                      float* __restrict__ d_rough, float* __restrict__ d_view, float* __restrict__ d_inc,
                      float* __restrict__ d_env)
{
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    float* s_env = s_mem;
    float* s_denv = s_mem + (ENV_LDS ? He * We * 3 : 0);
    const int ntex = He * We * 3;
    if (ENV_LDS) {
        for (int i = threadIdx.x; i < ntex; i += blockDim.x) {
            s_env[i] = env[i];
            s_denv[i] = 0.f;
        }
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // channel map of the 64-wide transposed reduction: 0..47 incidents (i*3+c), 48..50 base, 51 roughness, 52..54 view
    const int chan = transposed_channel<64>(lane);
    const float invK = 1.0f / (float)K;
    for (int g0 = blockIdx.x * SHADE_WAVES + wave; g0 < P; g0 += gridDim.x * SHADE_WAVES) {
        const int g = __builtin_amdgcn_readfirstlane(g0);
        GaussFwd G;
        gauss_setup(G, base_color, roughness, normals, viewdirs, g);
        const float* sh = incidents + (size_t)g * M * 3;
        const float gp[3] = {g_pbr[3 * g] * invK, g_pbr[3 * g + 1] * invK, g_pbr[3 * g + 2] * invK};
        const float gd[3] = {g_diff[3 * g] * invK, g_diff[3 * g + 1] * invK, g_diff[3 * g + 2] * invK};
        float v[64];
#pragma unroll
        for (int i = 0; i < 64; i++) v[i] = 0.f;
        for (int k = lane; k < K; k += 64) {
            const size_t o = (size_t)g * K + k;
            const float dx = dirs[3 * o], dy = dirs[3 * o + 1], dz = dirs[3 * o + 2];
            SampleFwd s;
            shade_sample<ENV_LDS>(s, G, sh, M, dx, dy, dz, visibility[o], areas[o], env, s_env, tr, He, We);
            float gspec = 0.f;
            float dlin[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float fd = G.base[c] / kPi;
                const float dT = gp[c] * (fd + s.spec) + gd[c];       // dL/dtransport_c
                gspec += gp[c] * s.transport[c];
                v[48 + c] += gp[c] * s.transport[c] / kPi;            // base_color
                dlin[c] = dT * s.area_ndi;                            // dL/d(incident light)_c
            }
            // incident SH (clamp_min(0): gradient where the SH sum >= 0) and environment texels
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float dl = s.shsum[c] >= 0.f ? dlin[c] : 0.f;
#pragma unroll
                for (int i = 0; i < 16; i++)
                    if (i < M) v[i * 3 + c] += dl * s.Y[i];
            }
#pragma unroll
            for (int t = 0; t < 4; t++) {
                if (s.taps.idx[t] >= 0) {
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const float val = dlin[c] * s.vis * s.taps.w[t];
                        if (ENV_LDS) atomicAdd(&s_denv[3 * s.taps.idx[t] + c], val);
                        else atomicAdd(&d_env[3 * (size_t)s.taps.idx[t] + c], val);
                    }
                }
            }
            // specular -> roughness, view direction
            const float frac = s.frac0 * G.a2;
            const bool nom_free = s.nomr >= 1e-6f && s.nomr <= 4.f * kPi;
            const float nom = fminf(fmaxf(s.nomr, 1e-6f), 4.f * kPi);
            const float dfrac = gspec / nom;
            const float dnom = nom_free ? -gspec * frac / (nom * nom) : 0.f;
            float da2 = dfrac * s.frac0;
            const float dfrac0 = dfrac * G.a2;
            const float dFMi = dfrac0 * 0.96f * 0.6931471805599453f * s.p2;
            float dVoH = dFMi * (-2.f * 5.55473f * s.VoH - 6.98316f);
            const float c4 = 4.f * kPi;
            const float dnom0 = dnom * c4 * 2.f * s.nom0 * s.nom1 * s.nom2;
            const float dnom1 = dnom * c4 * s.nom0 * s.nom0 * s.nom2;
            const float dnom2 = dnom * c4 * s.nom0 * s.nom0 * s.nom1;
            float dNoH = dnom0 * 2.f * s.NoH * (G.a2 - 1.f);
            da2 += dnom0 * s.NoH * s.NoH;
            float dNoV = dnom1 * (1.f - G.kk);
            const float dkk = dnom1 * (1.f - G.NoV) + dnom2 * (1.f - s.NoL);
            float da = dkk / 8.f + da2 * 2.f * G.a;
            const float dr = dkk * 2.f / 8.f + da * 2.f * G.r;
            v[51] += dr;
            if (!(s.rawNoH >= 1e-6f && s.rawNoH <= 1.f)) dNoH = 0.f;
            if (!(s.rawVoH >= 1e-6f && s.rawVoH <= 1.f)) dVoH = 0.f;
            if (!(G.rawNoV >= 1e-6f && G.rawNoV <= 1.f)) dNoV = 0.f;
            float dH[3], dV[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                dH[c] = dNoH * G.N[c] + dVoH * G.V[c];
                dV[c] = dVoH * s.Hh[c] + dNoV * G.N[c];
            }
            const float hd = s.Hh[0] * dH[0] + s.Hh[1] * dH[1] + s.Hh[2] * dH[2];
#pragma unroll
            for (int c = 0; c < 3; c++) dV[c] += 0.5f * (dH[c] - s.Hh[c] * hd) / s.ulen;
            const float vd = G.V[0] * dV[0] + G.V[1] * dV[1] + G.V[2] * dV[2];
#pragma unroll
            for (int c = 0; c < 3; c++) v[52 + c] += (dV[c] - G.V[c] * vd) / G.vlen;
        }
        const float total = transpose_reduce<64, true>(v);
        if (chan < 48) {
            if (chan < M * 3) d_inc[(size_t)g * M * 3 + chan] = total;
        } else if (chan < 51) d_base[3 * g + (chan - 48)] = total;
        else if (chan == 51) d_rough[g] = total;
        else if (chan < 55) d_view[3 * g + (chan - 52)] = total;
    }
    if (ENV_LDS) {
        __syncthreads();
        for (int i = threadIdx.x; i < ntex; i += blockDim.x) {
            const float val = s_denv[i];
            if (val != 0.f) atomicAdd(&d_env[i], val);
        }
    }
}

static int shade_grid(int P)
{
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int want = (P + SHADE_WAVES - 1) / SHADE_WAVES;
    const int cap = cus * 8;
    return want < cap ? (want > 0 ? want : 1) : cap;
}

void launch_shade_forward(hipStream_t s, int P, int K, int M, const float* base_color, const float* roughness,
                          const float* normals, const float* viewdirs, const float* incidents, const float* env, int He,
                          int We, const float* tr, const float* visibility, const float* dirs, const float* areas,
                          float* out)
{
    const int ntex = He * We * 3;
    const int grid = shade_grid(P);
    if (ntex <= ENV_LDS_MAX)
        shade_forward_kernel<true><<<grid, 64 * SHADE_WAVES, ntex * sizeof(float), s>>>(
            P, K, M, base_color, roughness, normals, viewdirs, incidents, env, He, We, tr, visibility, dirs, areas, out);
    else
        shade_forward_kernel<false><<<grid, 64 * SHADE_WAVES, 0, s>>>(
            P, K, M, base_color, roughness, normals, viewdirs, incidents, env, He, We, tr, visibility, dirs, areas, out);
}

void launch_shade_backward(hipStream_t s, int P, int K, int M, const float* base_color, const float* roughness,
                           const float* normals, const float* viewdirs, const float* incidents, const float* env,
                           int He, int We, const float* tr, const float* visibility, const float* dirs,
                           const float* areas, const float* g_pbr, const float* g_diff, float* d_base, float* d_rough,
                           float* d_view, float* d_inc, float* d_env)
{
    const int ntex = He * We * 3;
    // persistent blocks so the LDS-privatised env gradient is flushed once per block, not once per Gaussian
    int grid = shade_grid(P);
    if (2 * ntex <= ENV_LDS_MAX)
        shade_backward_kernel<true><<<grid, 64 * SHADE_WAVES, 2 * ntex * sizeof(float), s>>>(
            P, K, M, base_color, roughness, normals, viewdirs, incidents, env, He, We, tr, visibility, dirs, areas, g_pbr,
            g_diff, d_base, d_rough, d_view, d_inc, d_env);
    else
        shade_backward_kernel<false><<<grid, 64 * SHADE_WAVES, 0, s>>>(
            P, K, M, base_color, roughness, normals, viewdirs, incidents, env, He, We, tr, visibility, dirs, areas, g_pbr,
            g_diff, d_base, d_rough, d_view, d_inc, d_env);
}

}  // namespace r3dg
