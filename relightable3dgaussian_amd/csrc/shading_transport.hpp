// The relight frame's opt-in transport cache: builder + per-frame kernel (see the banner below).  Included by shading.hip
// (after shading_math.hpp) and, unchanged, by the CPU emulation in tests/emu.
#pragma once

namespace r3dg {

// =====================================================================================================================
// Relighting under a FIXED light ("transport" cache; opt-in, relight.RelightRenderer(cache="transport")).
// While neither the Gaussians nor the light change -- a camera flying through a relit scene, relighting.py with
// configs/teaser or configs/nerf_syn -- everything of the integral that does not depend on the view is a constant:
//     transport_k = (max(SH_incident(d_k), 0) + radiance(d_k) * visibility_k) * area_k * max(n . d_k, 0)     per sample,
//     diffuse_light, mean incident / local / global light, mean visibility                                       per Gaussian,
// and per frame only the GGX lobe is left:  specular = mean_k f_s(n, v, d_k) transport_k,  pbr = albedo / pi * diffuse_light
// + specular (rendering_equation, neilf.py:339-371 with the sums regrouped).  shade_build_transport_kernel turns the cached
// RADIANCE of every sample (r3dg_shade_build_taps with a radiance map) into its transport IN PLACE and writes the 13
// per-Gaussian constants; shade_forward_transport_kernel reads 12 bytes per sample (the transport) -- the direction is
// regenerated from the Gaussian's normal and the K-entry Fibonacci table (rotation_between_z, utils/sh_utils.py:36-68;
// graphics_utils.py:9-37), or read from the cache when the caller passes it -- and evaluates ~80 instead of ~290
// instructions per sample.  One wave per Gaussian, lane = sample; plain (non-persistent) launches.
// =====================================================================================================================
constexpr int TR_WAVES = 4;
constexpr int TR_CONSTS = 16;     // floats per Gaussian: diffuse_light 3 | incident light 3 | local 3 | global 3 | visibility 1 | pad

__device__ __forceinline__ float wave_sum64(float x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

__global__ void __launch_bounds__(64 * TR_WAVES)
shade_build_transport_kernel(int P, int K, int M, const float* __restrict__ normals, const float* __restrict__ incidents,
                             const float* __restrict__ visibility, const float* __restrict__ dirs,
                             const float* __restrict__ areas, float uniform_area, float* radiance_to_transport,
                             float* __restrict__ consts)
{
    __shared__ __attribute__((aligned(16))) float s_sh[TR_WAVES][48];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = blockIdx.x * TR_WAVES + wave;
    if (g >= P) return;                                       // whole waves leave; no block-wide barrier below
    float* sh = s_sh[wave];
    if (lane < 48) sh[lane] = lane < 3 * M ? incidents[(size_t)g * 3 * M + lane] : 0.f;
    __builtin_amdgcn_wave_barrier();      // one wave: its LDS operations execute in order (no instruction; the emulation's cue)
    const float nx = normals[3 * (size_t)g], ny = normals[3 * (size_t)g + 1], nz = normals[3 * (size_t)g + 2];
    const size_t row = (size_t)g * (size_t)K;
    float acc[13];
#pragma unroll
    for (int i = 0; i < 13; i++) acc[i] = 0.f;
    const int kend = (K + 63) & ~63;
    for (int k = lane; k < kend; k += 64) {
        if (k < K) {
            const float dx = dirs[3 * (row + k)], dy = dirs[3 * (row + k) + 1], dz = dirs[3 * (row + k) + 2];
            const float vis = visibility[row + k];
            const float area = areas != nullptr ? areas[row + k] : uniform_area;
            float* e = radiance_to_transport + 3 * (row + k);
            float Y[16];
            sh_basis16(dx, dy, dz, M, Y);
            float l[3];
            sh_local_sum(sh, Y, l);
            const float area_ndi = area * fmaxf(nx * dx + ny * dy + nz * dz, 0.f);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float loc = fmaxf(l[c], 0.f), glob = e[c] * vis, lin = loc + glob, t = lin * area_ndi;
                e[c] = t;
                acc[c] += t;
                acc[3 + c] += lin;
                acc[6 + c] += loc;
                acc[9 + c] += glob;
            }
            acc[12] += vis;
        }
    }
    const float invK = 1.0f / (float)K;
#pragma unroll
    for (int i = 0; i < 13; i++) acc[i] = wave_sum64(acc[i]) * invK;
    if (lane == 0) {
        float* o = consts + (size_t)g * TR_CONSTS;
#pragma unroll
        for (int i = 0; i < TR_CONSTS; i++) o[i] = i < 13 ? acc[i] : 0.f;
    }
}

__global__ void __launch_bounds__(64 * TR_WAVES)
shade_forward_transport_kernel(int P, int K, const float* __restrict__ base_color, const float* __restrict__ roughness,
                               const float* __restrict__ normals, const float* __restrict__ viewdirs,
                               const float* __restrict__ transport, const float* __restrict__ consts,
                               const float* __restrict__ zsamples, const float* __restrict__ dirs, float* __restrict__ out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = blockIdx.x * TR_WAVES + wave;
    if (g >= P) return;
    float u[64];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        u[48 + c] = base_color[3 * (size_t)g + c];
        u[52 + c] = normals[3 * (size_t)g + c];
        u[55 + c] = viewdirs[3 * (size_t)g + c];
    }
    u[51] = roughness[g];
    GaussFwd G;
    gauss_setup(G, u);
    // rotation_between_z(normal): the rotation that takes +z to the normal (the identity's negative when n_z + 1 <= 0)
    float R[9];
    {
        const float v1 = -G.n[1], v2 = G.n[0], cp = fmaxf(G.n[2] + 1.f, 1e-7f);
        const bool regular = G.n[2] + 1.f > 0.f;
        R[0] = regular ? 1.f + (-v2 * v2) / cp : -1.f;
        R[1] = regular ? v1 * v2 / cp : 0.f;
        R[2] = regular ? v2 : 0.f;
        R[3] = R[1];
        R[4] = regular ? 1.f + (-v1 * v1) / cp : -1.f;
        R[5] = regular ? -v1 : 0.f;
        R[6] = regular ? -v2 : 0.f;
        R[7] = regular ? v1 : 0.f;
        R[8] = regular ? 1.f + (-v2 * v2 - v1 * v1) / cp : -1.f;
    }
    const float a2 = G.a2, kk = G.kk;
    const float nom1 = G.NoV * (1.f - kk) + kk;
    const size_t row = (size_t)g * (size_t)K;
    float S[3] = {0.f, 0.f, 0.f};
    const int kend = (K + 63) & ~63;
    for (int k = lane; k < kend; k += 64) {
        if (k < K) {
            float rx, ry, rz;
            if (dirs != nullptr) {
                rx = dirs[3 * (row + k)]; ry = dirs[3 * (row + k) + 1]; rz = dirs[3 * (row + k) + 2];
            } else {
                const float zx = zsamples[3 * k], zy = zsamples[3 * k + 1], zz = zsamples[3 * k + 2];
                rx = R[0] * zx + R[1] * zy + R[2] * zz;
                ry = R[3] * zx + R[4] * zy + R[5] * zz;
                rz = R[6] * zx + R[7] * zy + R[8] * zz;
            }
            const float* t = transport + 3 * (row + k);
            const float t0 = t[0], t1 = t[1], t2 = t[2];
            // GGX lobe exactly as in the row kernel (neilf.py:374-407)
            const float dinv = __builtin_amdgcn_rsqf(fmaxf(rx * rx + ry * ry + rz * rz, 1e-24f));
            const float Lx = rx * dinv, Ly = ry * dinv, Lz = rz * dinv;
            const float ux = (Lx + G.V[0]) / 2.0f, uy = (Ly + G.V[1]) / 2.0f, uz = (Lz + G.V[2]) / 2.0f;
            const float uinv = __builtin_amdgcn_rsqf(fmaxf(ux * ux + uy * uy + uz * uz, 1e-24f));
            const float Hx = ux * uinv, Hy = uy * uinv, Hz = uz * uinv;
            const float NoL = fminf(fmaxf(G.N[0] * Lx + G.N[1] * Ly + G.N[2] * Lz, 1e-6f), 1.f);
            const float NoH = fminf(fmaxf(G.N[0] * Hx + G.N[1] * Hy + G.N[2] * Hz, 1e-6f), 1.f);
            const float VoH = fminf(fmaxf(G.V[0] * Hx + G.V[1] * Hy + G.V[2] * Hz, 1e-6f), 1.f);
            const float p2 = exp2f((-5.55473f * VoH - 6.98316f) * VoH);
            const float frac = (0.04f + 0.96f * p2) * a2;
            const float nom0 = NoH * NoH * (a2 - 1.f) + 1.f;
            const float nom2 = NoL * (1.f - kk) + kk;
            const float nom = fminf(fmaxf(4.f * kPi * nom0 * nom0 * nom1 * nom2, 1e-6f), 4.f * kPi);
            const float spec = frac / nom;
            S[0] += spec * t0;
            S[1] += spec * t1;
            S[2] += spec * t2;
        }
    }
    const float invK = 1.0f / (float)K;
#pragma unroll
    for (int c = 0; c < 3; c++) S[c] = wave_sum64(S[c]) * invK;
    if (lane == 0) {
        const float* cst = consts + (size_t)g * TR_CONSTS;
        float* o = out + (size_t)g * SHADE_NOUT;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            o[c] = G.base[c] / kPi * cst[c] + S[c];     // pbr
            o[3 + c] = cst[c];                          // diffuse_light
            o[6 + c] = S[c];                            // specular
            o[9 + c] = cst[3 + c];                      // mean incident light
            o[12 + c] = cst[6 + c];                     // local
            o[15 + c] = cst[9 + c];                     // global
        }
        o[18] = cst[12];                                // mean visibility
    }
}

}  // namespace r3dg
