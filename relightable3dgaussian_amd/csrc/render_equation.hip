// The reference's render_equation.{cu,h} contract model (metallic BRDF with a spherical-Gaussian D, SH environment
// light, SH visibility, SH local light) for gfx950: forward, forward_complex, backward.
// Reference semantics: render_equation.cu:555-666 (forward), :55-190 (forward_complex), :280-463 (backward); the
// backward's quirks Q1-Q4 listed in oracle/shading_oracle.c are reproduced, the cross-thread race on dL_ddirect_shs
// (Q5) is replaced by the well-defined sum (LDS-privatised per block, one global atomic per coefficient per block).
//
// The reference runs one THREAD per Gaussian with a serial loop over the samples.  Here one WAVE owns a Gaussian and
// the 64 lanes take the samples, so the per-sample outputs ([P,K,3] directions / lights) are written coalesced and the
// K-sums are transposing wave reductions (wave_reduce.hpp).
#include "common.hpp"
#include "wave_reduce.hpp"

namespace r3dg {

constexpr float kPiRef = 3.14159f;      // the reference's literal
constexpr int RE_WAVES = 4;

__device__ __forceinline__ void re_sh_coef3(const float d[3], float (&coef)[16])
{
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    const float x = d[0], y = d[1], z = d[2];
    coef[0] = C0;
    coef[1] = -C1 * y; coef[2] = C1 * z; coef[3] = -C1 * x;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    coef[4] = 1.0925484305920792f * xy;
    coef[5] = -1.0925484305920792f * yz;
    coef[6] = 0.31539156525252005f * (2.0f * zz - xx - yy);
    coef[7] = -1.0925484305920792f * xz;
    coef[8] = 0.5462742152960396f * (xx - yy);
    coef[9] = -0.5900435899266435f * y * (3.0f * xx - yy);
    coef[10] = 2.890611442640554f * xy * z;
    coef[11] = -0.4570457994644658f * y * (4.0f * zz - xx - yy);
    coef[12] = 0.3731763325901154f * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
    coef[13] = -0.4570457994644658f * x * (4.0f * zz - xx - yy);
    coef[14] = 1.445305721320277f * z * (xx - yy);
    coef[15] = -0.5900435899266435f * x * (xx - 3.0f * yy);
}

// Fibonacci direction rotated z -> normal (render_equation.cu:583-610)
__device__ __forceinline__ void re_sample_dir(const float n[3], int ray_id, int sample_num, bool has_rand, float rnd,
                                              float (&dir)[3])
{
    const float delta = kPiRef * (3.0f - sqrtf(5.0f));
    const float z = 1 - 2 * (float)ray_id / (2 * (float)sample_num - 1);
    const float rad = sqrtf(1 - z * z);
    float theta = delta * ray_id;
    if (has_rand) theta = rnd * 2 * kPiRef + theta;
    const float y = cosf(theta) * rad;
    const float x = sinf(theta) * rad;
    const float v1 = -n[1], v2 = n[0];
    const float v11 = v1 * v1, v22 = v2 * v2, v12 = v1 * v2;
    const float cp = fmaxf(n[2] + 1, 0.0000001f);
    const float zx = (1 + (-v22) / cp) * x + (v12 / cp) * y + v2 * z;
    const float zy = (v12 / cp) * x + (1 + (-v11) / cp) * y + (-v1) * z;
    const float zz = (-v2) * x + v1 * y + (1 + (-v22 - v11) / cp) * z;
    const float norm = sqrtf(fmaxf(0.0000001f, zx * zx + zy * zy + zz * zz));
    dir[0] = zx / norm; dir[1] = zy / norm; dir[2] = zz / norm;
}

struct ReSample {
    float coef[16], local[3], glob_raw[3], glob[3], vis, light[3];
    float half_n[3], half_norm, h_d_n, h_d_o, n_d_i, n_d_o;
    float f_d[3], r2, amp, sharp, expf_amp, D, F0[3], F[3], r2v, denom1, denom2, g1, g2, V, f_s[3];
};

__device__ __forceinline__ float re_dot(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

__device__ __forceinline__ void re_eval(ReSample& s, int Si, int Sd, int Sv, const float base[3], float rough, float metal,
                                        const float normal[3], const float viewdir[3], const float* __restrict__ inc,
                                        const float* __restrict__ direct, const float* __restrict__ vis, const float dir[3])
{
    re_sh_coef3(dir, s.coef);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float l = 0.f, g = 0.5f;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (i < Si) l += inc[i * 3 + c] * s.coef[i];
            if (i < Sd) g += direct[i * 3 + c] * s.coef[i];
        }
        s.local[c] = fmaxf(l, 0.0f);
        s.glob_raw[c] = fmaxf(g, 0.0f);
    }
    float v = 0.5f;
#pragma unroll
    for (int i = 0; i < 16; i++)
        if (i < Sv) v += vis[i] * s.coef[i];
    s.vis = fmaxf(0.0f, fminf(v, 1.0f));
    float hd[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        s.glob[c] = s.vis * s.glob_raw[c];
        s.light[c] = s.glob[c] + s.local[c];
        hd[c] = dir[c] + viewdir[c];
    }
    s.half_norm = fmaxf(sqrtf(re_dot(hd, hd)), 0.0000001f);
#pragma unroll
    for (int c = 0; c < 3; c++) s.half_n[c] = hd[c] / s.half_norm;
    s.h_d_n = fmaxf(re_dot(s.half_n, normal), 0.0f);
    s.h_d_o = fmaxf(re_dot(s.half_n, viewdir), 0.0f);
    s.n_d_i = fmaxf(re_dot(normal, dir), 0.0f);
    s.n_d_o = fmaxf(re_dot(normal, viewdir), 0.0f);
#pragma unroll
    for (int c = 0; c < 3; c++) s.f_d[c] = (1 - metal) * base[c] / kPiRef;
    s.r2 = fmaxf(rough * rough, 0.0000001f);
    s.amp = 1.0f / (s.r2 * kPiRef);
    s.sharp = 2.0f / s.r2;
    s.expf_amp = expf(s.sharp * (s.h_d_n - 1.0f));
    s.D = s.amp * s.expf_amp;
    const float om = 1.0f - s.h_d_o;
    const float p5 = om * om * om * om * om;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        s.F0[c] = 0.04f * (1.0f - metal) + base[c] * metal;
        s.F[c] = s.F0[c] + (1.0f - s.F0[c]) * p5;
    }
    s.r2v = (1.0f + rough) * (1.0f + rough) / 8.0f;
    s.denom1 = fmaxf(s.n_d_i * (1 - s.r2v) + s.r2v, 0.0000001f);
    s.denom2 = fmaxf(s.n_d_o * (1 - s.r2v) + s.r2v, 0.0000001f);
    s.g1 = 0.5f / s.denom1;
    s.g2 = 0.5f / s.denom2;
    s.V = s.g1 * s.g2;
#pragma unroll
    for (int c = 0; c < 3; c++) s.f_s[c] = s.D * s.F[c] * s.V;
}

// forward (COMPLEX=false: pbr, incident_dirs, diffuse_light) and forward_complex (COMPLEX=true: all 11 outputs)
template <bool COMPLEX>
__global__ void __launch_bounds__(64 * RE_WAVES)
re_forward_kernel(int P, int Si, int Sd, int Sv, const float* __restrict__ base_color, const float* __restrict__ roughness,
                  const float* __restrict__ metallic, const float* __restrict__ normals,
                  const float* __restrict__ viewdirs, const float* __restrict__ inc, const float* __restrict__ direct,
                  const float* __restrict__ vis, int K, const float* __restrict__ rand_float,
                  float* __restrict__ incident_dirs, float* __restrict__ out_pbr, float* __restrict__ out_lights,
                  float* __restrict__ out_local, float* __restrict__ out_global, float* __restrict__ out_vis,
                  float* __restrict__ out_diffuse, float* __restrict__ out_local_diffuse, float* __restrict__ out_accum,
                  float* __restrict__ out_rgb_d, float* __restrict__ out_rgb_s)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int g0 = blockIdx.x * RE_WAVES + wave; g0 < P; g0 += gridDim.x * RE_WAVES) {
        const int g = __builtin_amdgcn_readfirstlane(g0);
        const float base[3] = {base_color[3 * g], base_color[3 * g + 1], base_color[3 * g + 2]};
        const float normal[3] = {normals[3 * g], normals[3 * g + 1], normals[3 * g + 2]};
        const float viewdir[3] = {viewdirs[3 * g], viewdirs[3 * g + 1], viewdirs[3 * g + 2]};
        const float rough = roughness[g], metal = metallic[g];
        // channels: 0..2 rgb_d, 3..5 rgb_s, 6..8 diffuse, 9..11 local diffuse
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = 0.f;
        for (int k = lane; k < K; k += 64) {
            const size_t w = (size_t)g * K + k;
            float dir[3];
            re_sample_dir(normal, k, K, rand_float != nullptr, rand_float ? rand_float[w] : 0.f, dir);
            ReSample s;
            re_eval(s, Si, Sd, Sv, base, rough, metal, normal, viewdir, inc + (size_t)g * Si * 3, direct,
                    vis + (size_t)g * Sv, dir);
            const float tmp = 2.0f * kPiRef * s.n_d_i / (float)K;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float tr = s.light[c] * tmp;
                v[c] += s.f_d[c] * tr;
                v[3 + c] += s.f_s[c] * tr;
                v[6 + c] += tr;
                v[9 + c] += s.local[c] * tmp;
                incident_dirs[3 * w + c] = dir[c];
                if (COMPLEX) {
                    out_lights[3 * w + c] = s.light[c];
                    out_local[3 * w + c] = s.local[c];
                    out_global[3 * w + c] = s.glob[c];
                }
            }
            if (COMPLEX) out_vis[w] = s.vis;
        }
        const float total = transpose_reduce<16, true>(v);
        // every lane holds the total of ITS channel; the derived outputs (pbr = rgb_d + rgb_s, accum) need several
        // channels, so broadcast the 12 sums from their owner lanes (constant lane ids -> v_readlane)
        // lane -> channel map is static: find the owner lane of channel q
        auto owner_lane = [](int q) {       // inverse of transposed_channel<16> restricted to owners (low 2 bits 0)
            int l = 0;
            if (q & 8) l |= 32;
            if (q & 4) l |= 16;
            if (q & 2) l |= 8;
            if (q & 1) l |= 4;
            return l;
        };
        float s12[12];
#pragma unroll
        for (int q = 0; q < 12; q++) s12[q] = __shfl(total, owner_lane(q), 64);
        if (lane < 3) {
            const int c = lane;
            out_pbr[3 * g + c] = s12[c] + s12[3 + c];
            out_diffuse[3 * g + c] = s12[6 + c];
            if (COMPLEX) {
                out_rgb_d[3 * g + c] = s12[c];
                out_rgb_s[3 * g + c] = s12[3 + c];
                out_local_diffuse[3 * g + c] = s12[9 + c];
            }
        }
        if (COMPLEX && lane == 0) {
            const float a0 = s12[6] / kPiRef + s12[3], a1 = s12[7] / kPiRef + s12[4], a2 = s12[8] / kPiRef + s12[5];
            out_accum[g] = (a0 + a1 + a2) / 3;
        }
    }
}

__global__ void __launch_bounds__(64 * RE_WAVES)
re_backward_kernel(int P, int Si, int Sd, int Sv, const float* __restrict__ base_color,
                   const float* __restrict__ roughness, const float* __restrict__ metallic,
                   const float* __restrict__ normals, const float* __restrict__ viewdirs, const float* __restrict__ inc,
                   const float* __restrict__ direct, const float* __restrict__ vis, int K,
                   const float* __restrict__ incident_dirs, const float* __restrict__ dL_dpbrs,
                   const float* __restrict__ dL_ddls, float* __restrict__ dL_dbase, float* __restrict__ dL_drough,
                   float* __restrict__ dL_dmetal, float* __restrict__ dL_dnormals, float* __restrict__ dL_dviewdirs,
                   float* __restrict__ dL_dinc, float* __restrict__ dL_ddirect, float* __restrict__ dL_dvis)
{
    __shared__ float s_ddirect[48];
    if (threadIdx.x < 48) s_ddirect[threadIdx.x] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chanA = transposed_channel<32>(lane);
    const bool ownerA = transposed_owner<32>(lane);
    const int chanB = transposed_channel<64>(lane);
    const int Sinc_loop = Sd < Si ? Sd : Si;     // Q2 (clamped so it cannot run past the row)
    for (int g0 = blockIdx.x * RE_WAVES + wave; g0 < P; g0 += gridDim.x * RE_WAVES) {
        const int g = __builtin_amdgcn_readfirstlane(g0);
        const float base[3] = {base_color[3 * g], base_color[3 * g + 1], base_color[3 * g + 2]};
        const float normal[3] = {normals[3 * g], normals[3 * g + 1], normals[3 * g + 2]};
        const float viewdir[3] = {viewdirs[3 * g], viewdirs[3 * g + 1], viewdirs[3 * g + 2]};
        const float gp[3] = {dL_dpbrs[3 * g], dL_dpbrs[3 * g + 1], dL_dpbrs[3 * g + 2]};
        const float gd[3] = {dL_ddls[3 * g], dL_ddls[3 * g + 1], dL_ddls[3 * g + 2]};
        const float rough = roughness[g], metal = metallic[g];
        float rA = 0.f, rB = 0.f, rC = 0.f;
        for (int k0 = 0; k0 < K; k0 += 64) {
            const int k = k0 + lane;
            const bool live = k < K;
            float dlight[3] = {0.f, 0.f, 0.f}, dglob[3] = {0.f, 0.f, 0.f};
            float coef[16];
            // group A: 0..2 base, 3 rough, 4 metal, 5..7 normal, 8..10 view, 11..26 visibility SH
            float vA[32];
#pragma unroll
            for (int i = 0; i < 32; i++) vA[i] = 0.f;
#pragma unroll
            for (int i = 0; i < 16; i++) coef[i] = 0.f;
            if (live) {
                const size_t w = (size_t)g * K + k;
                const float dir[3] = {incident_dirs[3 * w], incident_dirs[3 * w + 1], incident_dirs[3 * w + 2]};
                ReSample s;
                re_eval(s, Si, Sd, Sv, base, rough, metal, normal, viewdir, inc + (size_t)g * Si * 3, direct,
                        vis + (size_t)g * Sv, dir);
#pragma unroll
                for (int i = 0; i < 16; i++) coef[i] = s.coef[i];
                const float tw = 2.0f * kPiRef * s.n_d_i / (float)K;
                float dfd[3], dfs[3];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    dfd[c] = gp[c] * s.light[c] * tw;
                    dfs[c] = dfd[c];
                    dlight[c] = gp[c] * (s.f_d[c] + s.f_s[c]) * tw + gd[c] * tw;
                }
                float dbase[3];
#pragma unroll
                for (int c = 0; c < 3; c++) dbase[c] = dfd[c] * (1 - metal) / kPiRef;
                float dmetal = -re_dot(dfd, base) / kPiRef;
                float t3[3] = {dfs[0] * s.V, dfs[1] * s.V, dfs[2] * s.V};
                const float dD = re_dot(t3, s.F);
                float dF[3];
#pragma unroll
                for (int c = 0; c < 3; c++) { dF[c] = dfs[c] * s.D * s.V; t3[c] = dfs[c] * s.D; }
                const float dV = re_dot(t3, s.F);
                const float damp = dD * s.expf_amp;
                const float dexp = dD * s.amp;
                const float dsharp = (s.h_d_n - 1.0f) * s.expf_amp * dexp;
                const float dh_d_n = s.sharp * s.expf_amp * dexp;
                const float dr2 = -2.0f / (s.r2 * s.r2) * dsharp - 1.0f / (s.r2 * s.r2 * kPiRef) * damp;
                float drough = dr2 * 2.0f * rough;
                const float om = 1.0f - s.h_d_o;
                const float p4 = om * om * om * om, p5 = p4 * om;
                float dF0[3], omF0[3], bm[3];
#pragma unroll
                for (int c = 0; c < 3; c++) { dF0[c] = (1.0f - p5) * dF[c]; omF0[c] = 1.0f - s.F0[c]; bm[c] = base[c] - 0.04f; }
                const float dh_d_o = re_dot(omF0, dF) * -5.0f * p4;
#pragma unroll
                for (int c = 0; c < 3; c++) dbase[c] += metal * dF0[c];
                dmetal += re_dot(bm, dF0);
                const float dg1 = dV * s.g2, dg2 = dV * s.g1;
                const float dden1 = -0.5f / (s.denom1 * s.denom1) * dg1;
                const float dden2 = -0.5f / (s.denom2 * s.denom2) * dg2;
                const float dn_d_i = dden1 * (1 - s.r2v);                    // Q1: overwrites the transport path
                const float dn_d_o = dden2 * (1 - s.r2v);
                const float dr2v = (1.0f - s.n_d_i) * dden1 + (1.0f - s.n_d_o) * dden2;
                drough += (1.0f + rough) / 4.0f * dr2v;
                float dh[3] = {0.f, 0.f, 0.f}, dn[3] = {0.f, 0.f, 0.f}, dv[3] = {0.f, 0.f, 0.f};
                if (s.h_d_n > 0.0f)
#pragma unroll
                    for (int c = 0; c < 3; c++) { dh[c] += normal[c] * dh_d_n; dn[c] += s.half_n[c] * dh_d_n; }
                if (s.h_d_o > 0.0f)
#pragma unroll
                    for (int c = 0; c < 3; c++) { dh[c] += viewdir[c] * dh_d_o; dv[c] += s.half_n[c] * dh_d_o; }
                if (s.n_d_i > 0.0f)
#pragma unroll
                    for (int c = 0; c < 3; c++) dn[c] += dir[c] * dn_d_i;
                if (s.n_d_o > 0.0f)
#pragma unroll
                    for (int c = 0; c < 3; c++) { dn[c] += viewdir[c] * dn_d_o; dv[c] += normal[c] * dn_d_o; }
#pragma unroll
                for (int c = 0; c < 3; c++) dv[c] += dh[c] / s.half_norm;  // Q4: normalisation not differentiated
#pragma unroll
                for (int c = 0; c < 3; c++) dglob[c] = dlight[c] * s.vis;  // Q3: no clamp masks
                const float dvisib = re_dot(dlight, s.glob_raw);
#pragma unroll
                for (int c = 0; c < 3; c++) { vA[c] = dbase[c]; vA[5 + c] = dn[c]; vA[8 + c] = dv[c]; }
                vA[3] = drough;
                vA[4] = dmetal;
                if (s.vis <= 1.0f && s.vis >= 0.0f)
#pragma unroll
                    for (int i = 0; i < 16; i++)
                        if (i < Sv) vA[11 + i] = dvisib * s.coef[i];
            }
            rA += transpose_reduce<32, true>(vA);
            float vB[64];
#pragma unroll
            for (int i = 0; i < 64; i++) vB[i] = 0.f;
#pragma unroll
            for (int i = 0; i < 16; i++)
                if (i < Sinc_loop)
#pragma unroll
                    for (int c = 0; c < 3; c++) vB[i * 3 + c] = dlight[c] * coef[i];
            rB += transpose_reduce<64, true>(vB);
#pragma unroll
            for (int i = 0; i < 64; i++) vB[i] = 0.f;
#pragma unroll
            for (int i = 0; i < 16; i++)
                if (i < Sd)
#pragma unroll
                    for (int c = 0; c < 3; c++) vB[i * 3 + c] = dglob[c] * coef[i];
            rC += transpose_reduce<64, true>(vB);
        }
        if (ownerA) {
            if (chanA < 3) dL_dbase[3 * g + chanA] = rA;
            else if (chanA == 3) dL_drough[g] = rA;
            else if (chanA == 4) dL_dmetal[g] = rA;
            else if (chanA < 8) dL_dnormals[3 * g + chanA - 5] = rA;
            else if (chanA < 11) dL_dviewdirs[3 * g + chanA - 8] = rA;
            else if (chanA - 11 < Sv) dL_dvis[(size_t)g * Sv + chanA - 11] = rA;
        }
        if (chanB < Si * 3 && chanB < 48) dL_dinc[(size_t)g * Si * 3 + chanB] = rB;
        if (chanB < Sd * 3 && chanB < 48) atomicAdd(&s_ddirect[chanB], rC);
    }
    __syncthreads();
    if (threadIdx.x < 48 && threadIdx.x < Sd * 3) {
        const float val = s_ddirect[threadIdx.x];
        if (val != 0.f) atomicAdd(&dL_ddirect[threadIdx.x], val);
    }
}

static int re_grid(int P)
{
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int want = (P + RE_WAVES - 1) / RE_WAVES;
    const int cap = cus * 8;
    return want < cap ? (want > 0 ? want : 1) : cap;
}

void launch_re_forward(hipStream_t s, bool complex_, int P, int Si, int Sd, int Sv, const float* base_color,
                       const float* roughness, const float* metallic, const float* normals, const float* viewdirs,
                       const float* inc, const float* direct, const float* vis, int K, const float* rand_float,
                       float* incident_dirs, float* out_pbr, float* out_lights, float* out_local, float* out_global,
                       float* out_vis, float* out_diffuse, float* out_local_diffuse, float* out_accum, float* out_rgb_d,
                       float* out_rgb_s)
{
    const int grid = re_grid(P);
    if (complex_)
        re_forward_kernel<true><<<grid, 64 * RE_WAVES, 0, s>>>(P, Si, Sd, Sv, base_color, roughness, metallic, normals,
                                                              viewdirs, inc, direct, vis, K, rand_float, incident_dirs,
                                                              out_pbr, out_lights, out_local, out_global, out_vis,
                                                              out_diffuse, out_local_diffuse, out_accum, out_rgb_d,
                                                              out_rgb_s);
    else
        re_forward_kernel<false><<<grid, 64 * RE_WAVES, 0, s>>>(P, Si, Sd, Sv, base_color, roughness, metallic, normals,
                                                               viewdirs, inc, direct, vis, K, rand_float, incident_dirs,
                                                               out_pbr, out_lights, out_local, out_global, out_vis,
                                                               out_diffuse, out_local_diffuse, out_accum, out_rgb_d,
                                                               out_rgb_s);
}

void launch_re_backward(hipStream_t s, int P, int Si, int Sd, int Sv, const float* base_color, const float* roughness,
                        const float* metallic, const float* normals, const float* viewdirs, const float* inc,
                        const float* direct, const float* vis, int K, const float* incident_dirs, const float* dL_dpbr,
                        const float* dL_ddl, float* dL_dbase, float* dL_drough, float* dL_dmetal, float* dL_dnormals,
                        float* dL_dviewdirs, float* dL_dinc, float* dL_ddirect, float* dL_dvis)
{
    re_backward_kernel<<<re_grid(P), 64 * RE_WAVES, 0, s>>>(P, Si, Sd, Sv, base_color, roughness, metallic, normals,
                                                           viewdirs, inc, direct, vis, K, incident_dirs, dL_dpbr, dL_ddl,
                                                           dL_dbase, dL_drough, dL_dmetal, dL_dnormals, dL_dviewdirs,
                                                           dL_dinc, dL_ddirect, dL_dvis);
}

}  // namespace r3dg
