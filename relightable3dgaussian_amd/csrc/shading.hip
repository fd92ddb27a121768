// Fused per-Gaussian shading integral (the LIVE stage-2 model) for gfx950, forward and backward.
// Reference semantics: rendering_equation neilf.py:339-371, GGX_specular :374-407, eval_sh sh_utils.py:71-128,
// DirectLightMap.direct_light direct_light_map.py:70-83 / EnvLight.direct_light envmap.py:35-53.
//
// The reference evaluates this with dozens of unfused elementwise passes over [P,K,3] temporaries (230 MB each at
// K=64) plus autograd saved tensors.  Here ONE kernel reads each cached sample once -- dirs [P,K,3], visibility
// [P,K], areas [P,K], 20 B per sample -- and writes 19 floats per Gaussian (forward) / 55 gradients (backward):
//   * 16 lanes per Gaussian, 4 Gaussians per wave, the K samples strided over the 16 lanes; per-Gaussian setup and the
//     final cross-lane reduction (a transposing butterfly that never leaves a 16-lane DPP row) are shared by four
//     Gaussians per instruction;
//   * the environment texture (activated, lat-long) is staged in LDS when it fits (16x32x3 fp32 = 6 KB) and sampled
//     bilinearly from there; in the backward its gradient is accumulated in a second LDS copy in 64-bit fixed point
//     (integer ds_add_u64: LDS float atomics retire ~1 lane / 3 cycles on gfx950) and flushed with one global atomic
//     per texel per persistent block;
//   * the per-Gaussian uniform record (48 SH coefficients, albedo, roughness, normal, view direction, upstream
//     gradients = 64 floats) is fetched by ONE coalesced wave load (lane l reads element l) into a per-wave LDS slot and
//     read back as broadcasts;
//   * backward only: record + samples of the wave's NEXT 64-sample block arrive by LDS-DMA (global_load_lds_dwordx4),
//     double buffered, while the current block is computed (three passes per block, 12 parked floats per lane).
#include "common.hpp"
#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>
#include <stdexcept>
#include <tuple>
#include "wave_reduce.hpp"
#include "shading_math.hpp"

namespace r3dg {

constexpr int SHADE_WAVES = 4;               // Gaussians in flight per block
constexpr int ENV_LDS_MAX = 12288;           // floats (48 KB) -- larger maps are sampled from global/L2

// acos / atan2 for the lat-long lookup, branch-free (Cephes single-precision minimax polynomials, ~1 ulp like the libm
// versions they replace at about a third of the instructions: the lookup runs once per cached sample, forward and
// backward).  acos: |x| <= 0.5 -> pi/2 - asin(x), else 2 asin(sqrt((1-|x|)/2)) reflected; atan: argument reduced to
// [0, tan(pi/8)] by the octant identities.
__device__ __forceinline__ float fast_acosf(float x)
{
    const float ax = fabsf(x);
    const bool big = ax > 0.5f;
    const float z = big ? 0.5f * (1.0f - ax) : x * x;
    const float s = big ? sqrtf(z) : ax;
    float p = 4.2163199048e-2f;
    p = p * z + 2.4181311049e-2f;
    p = p * z + 4.5470025998e-2f;
    p = p * z + 7.4953002686e-2f;
    p = p * z + 1.6666752422e-1f;
    const float a = s + s * z * p;                         // asin(s)
    const float pos = big ? 2.0f * a : 1.5707963267948966f - a;      // acos(|x|)
    return x >= 0.f ? pos : 3.14159265358979323846f - pos;
}

__device__ __forceinline__ float fast_atan2f(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    float a = mx > 0.f ? mn / mx : 0.f;                    // in [0,1]
    const bool hi = a > 0.4142135623730950f;               // tan(pi/8)
    a = hi ? (a - 1.0f) / (a + 1.0f) : a;
    const float z = a * a;
    float p = 8.05374449538e-2f;
    p = p * z - 1.38776856032e-1f;
    p = p * z + 1.99777106478e-1f;
    p = p * z - 3.33329491539e-1f;
    float r = p * z * a + a;
    r = hi ? r + 0.7853981633974483f : r;
    r = ay > ax ? 1.5707963267948966f - r : r;
    r = x < 0.f ? 3.14159265358979323846f - r : r;
    return __uint_as_float(__float_as_uint(r) | (__float_as_uint(y) & 0x80000000u));     // copysign: atan2(-0, x<0) = -pi
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
    const float phi = fast_acosf(dz) - 1e-6f;
    const float theta = fast_atan2f(dy, dx);
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

struct PackedTap {           // 12 bytes per cached sample
    uint32_t xy;             // (x0 + 1) | (y0 + 1) << 16, (x0, y0) = floor of the lat-long pixel coordinate (>= -1)
    float wx1, wy1;          // bilinear weights of column x0 + 1 / row y0 + 1
};

__device__ __forceinline__ PackedTap make_tap(float dx, float dy, float dz, const float* __restrict__ tr, int He, int We)
{
    if (tr != nullptr) {
        const float tx = dx * tr[0] + dy * tr[1] + dz * tr[2];
        const float ty = dx * tr[3] + dy * tr[4] + dz * tr[5];
        const float tz = dx * tr[6] + dy * tr[7] + dz * tr[8];
        dx = tx; dy = ty; dz = tz;
    }
    const float phi = fast_acosf(dz) - 1e-6f;
    const float theta = fast_atan2f(dy, dx);
    const float qy = (phi / kPi) * 2.f - 1.f;
    const float qx = -theta / kPi;
    const float ix = (qx + 1.f) * 0.5f * (float)(We - 1);
    const float iy = (qy + 1.f) * 0.5f * (float)(He - 1);
    const float x0f = floorf(ix), y0f = floorf(iy);
    PackedTap t;
    t.wx1 = ix - x0f;
    t.wy1 = iy - y0f;
    const int x0 = max((int)x0f, -1), y0 = max((int)y0f, -1);
    t.xy = (uint32_t)(x0 + 1) | ((uint32_t)(y0 + 1) << 16);
    return t;
}

// cached lookup record -> the four (texel, weight) taps of env_taps (texel -1 = zero padding)
__device__ __forceinline__ EnvTap taps_from_packed(const PackedTap& t, int He, int We)
{
    const int x0 = (int)(t.xy & 0xffffu) - 1, y0 = (int)(t.xy >> 16) - 1;
    const float wx[2] = {1.f - t.wx1, t.wx1}, wy[2] = {1.f - t.wy1, t.wy1};
    EnvTap o;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int x = x0 + b, y = y0 + a;
            const bool ok = x >= 0 && x <= We - 1 && y >= 0 && y <= He - 1;
            o.idx[a * 2 + b] = ok ? __mul24(y, We) + x : -1;
            o.w[a * 2 + b] = wy[a] * wx[b];
        }
    return o;
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

// Layout of the per-wave uniform record u[64]: 0..47 SH coefficients (i*3+c), 48..50 albedo, 51 roughness,
// 52..54 normal, 55..57 view direction, 58..60 dL_dpbr, 61..63 dL_ddiffuse_light (the last six only in the backward).
template <bool ENV_LDS, bool HAVE_SHSUM = false, bool HAVE_TAP = false>
__device__ __forceinline__ void shade_sample(SampleFwd& s, const GaussFwd& G, const float* sh /*[48] in LDS, zero padded*/,
                                             int M, float dx, float dy, float dz, float vis, float area,
                                             const float* __restrict__ env, const float* s_env,
                                             const float* __restrict__ tr, int He, int We, const PackedTap* cached = nullptr)
{
    // environment light (global) * visibility
    if (HAVE_TAP) s.taps = taps_from_packed(*cached, He, We);
    else s.taps = env_taps(dx, dy, dz, tr, He, We);
    s.vis = vis;
    float e[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; t++) {
        if (s.taps.idx[t] >= 0) {
            float3 px;
            if (ENV_LDS) {
                const float* q = s_env + 3 * s.taps.idx[t];
                px = make_float3(q[0], q[1], q[2]);
            } else {
                // large maps live in L2: one 12-byte load per tap (global_load_dwordx3), not three 4-byte ones
                px = *reinterpret_cast<const float3*>(env + 3 * (size_t)s.taps.idx[t]);
            }
            e[0] += px.x * s.taps.w[t];
            e[1] += px.y * s.taps.w[t];
            e[2] += px.z * s.taps.w[t];
        }
    }
    // local incident light: max(SH(d), 0)
    {
        float acc[3];
        if (HAVE_SHSUM) {                        // the caller evaluated the SH sum in an earlier pass (passed via s.shsum)
            acc[0] = s.shsum[0]; acc[1] = s.shsum[1]; acc[2] = s.shsum[2];
        } else {
            sh_basis16(dx, dy, dz, M, s.Y);      // Y[i] = 0 for i >= M, and the LDS record is zero padded
            sh_local_sum(sh, s.Y, acc);
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            s.shsum[c] = acc[c];
            s.local[c] = fmaxf(acc[c], 0.f);
            s.glob[c] = e[c] * vis;
            s.lin[c] = s.local[c] + s.glob[c];
        }
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

// ---- 16 lanes per Gaussian, 4 Gaussians per wave ----------------------------------------------------------
// One Gaussian occupies one 16-lane DPP row; its K samples are strided over the 16 lanes (k = l, l+16, ...), so each
// wave load touches four contiguous 192-byte runs.  Per-Gaussian setup and the final cross-lane reduction are shared
// by 4 Gaussians per wave instruction and the reduction never leaves a DPP row (distances 8,4,2,1 only).
constexpr int SH_L = 16;
constexpr int SH_GW = 64 / SH_L;                  // Gaussians per wave
constexpr int SH_GB = SH_GW * SHADE_WAVES;        // Gaussians per block step

// 16 partial values per lane -> lane l (within its 16-lane row) returns the row total of channel l
__device__ __forceinline__ float row_transpose_reduce16(float (&v)[16])
{
    const int lane = lane_id();
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = transpose_step<8, true>(v[k], v[k + 8], (lane & 8) != 0);
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = transpose_step<4, true>(v[k], v[k + 4], (lane & 4) != 0);
#pragma unroll
    for (int k = 0; k < 2; k++) v[k] = transpose_step<2, true>(v[k], v[k + 2], (lane & 2) != 0);
    return transpose_step<1, true>(v[0], v[1], (lane & 1) != 0);
}

// ---- register-free prefetch: global -> LDS DMA (global_load_lds), double buffered per wave -----------------------------
// Every wave owns two copies of {4 uniform records (4x64 floats), its 4 Gaussians' next 64 sample directions (4x64x3),
// visibilities (4x64), areas (4x64)}.  While the wave computes on one copy, the loads of its NEXT (Gaussian group,
// 64-sample block) are in flight into the other: the HBM/L2 latency of the three [P,K,*] caches and of the per-Gaussian
// record is hidden without spending VGPRs or extra waves (the kernels run at 2 waves/SIMD).  The LDS image of a DMA
// load is wave base + lane * size, so the copies keep the global layout: dirs [grp][k][3], vis/area [grp][k],
// record [grp][64].  With K % 4 == 0 every lane moves 16 bytes per instruction (9 DMA instructions per block),
// otherwise 4 bytes (24 instructions).
// The DMA is issued from inline assembly on purpose: the compiler's wait-count pass does not tell which LDS-DMA load
// feeds which LDS read and drains the whole vector-memory queue (s_waitcnt vmcnt(0)) in front of the first LDS access
// after a __builtin_amdgcn_global_load_lds -- including the prefetch that was just issued.  Loads it does not know
// about can only make its own waits stricter, never too weak (vmcnt completes in order), and the one true dependency
// -- "my previous prefetch has landed" -- is a single explicit s_waitcnt at the top of each block.  m0 (LDS base of
// the DMA) is saved and restored inside the statement.
template <int BYTES>
__device__ __forceinline__ void lds_dma(const float* gptr, float* lds_base /* wave-uniform */)
{
    const unsigned int off = __builtin_amdgcn_readfirstlane(
        (unsigned int)(size_t)(__attribute__((address_space(3))) float*)lds_base);
    unsigned int saved;
    if (BYTES == 16)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                     "s_mov_b32 m0, %0" : "=&s"(saved) : "v"(gptr), "s"(off) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\t"
                     "s_mov_b32 m0, %0" : "=&s"(saved) : "v"(gptr), "s"(off) : "memory");
}
#define R3DG_GLDS(gp, lp, sz) lds_dma<sz>(gp, lp)
// the same with the LDS byte address already in an SGPR (lds_address_of, computed once per wave): the generic-pointer form above
// costs a 64-bit VGPR pair + a null test per destination, which the compiler hoists out of loops and keeps live
__device__ __forceinline__ unsigned int lds_address_of(const float* lds_ptr /* wave-uniform */)
{
    return __builtin_amdgcn_readfirstlane((unsigned int)(size_t)(__attribute__((address_space(3))) const float*)lds_ptr);
}
template <int BYTES>
__device__ __forceinline__ void lds_dma_at(const float* gptr, unsigned int lds_byte_address /* SGPR */)
{
    unsigned int saved;
    if (BYTES == 16)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                     "s_mov_b32 m0, %0" : "=&s"(saved) : "v"(gptr), "s"(lds_byte_address) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\t"
                     "s_mov_b32 m0, %0" : "=&s"(saved) : "v"(gptr), "s"(lds_byte_address) : "memory");
}

constexpr int SB_USTRIDE = 68;
constexpr int SB_U = 0, SB_DIRS = 272, SB_VIS = 1040, SB_AREA = 1296, SB_FLOATS = 1552;   // one buffer of one wave

struct ShadeSrc {
    const float *base_color, *roughness, *normals, *viewdirs, *incidents, *g_pbr, *g_diff, *zero;
    const float *dirs, *vis, *areas;
};

__device__ __forceinline__ const float* uniform_src(int e, int g, int M, const ShadeSrc& p)
{
    const float* q = p.zero;
    if (e < 48) { if (e < 3 * M) q = p.incidents + (size_t)g * M * 3 + e; }
    else if (e < 51) q = p.base_color + 3 * (size_t)g + (e - 48);
    else if (e == 51) q = p.roughness + g;
    else if (e < 55) q = p.normals + 3 * (size_t)g + (e - 52);
    else if (e < 58) q = p.viewdirs + 3 * (size_t)g + (e - 55);
    else if (e < 61) { if (p.g_pbr) q = p.g_pbr + 3 * (size_t)g + (e - 58); }
    else { if (p.g_diff) q = p.g_diff + 3 * (size_t)g + (e - 61); }
    return q;
}

template <bool VEC16>
__device__ __forceinline__ void issue_block_loads(int lane, int gb, int k0, int P, int K, int M, const ShadeSrc& p,
                                                  float* sb /* one SB_FLOATS buffer of this wave */, size_t total /* samples in the arrays */)
{
#pragma unroll
    for (int t = 0; t < SH_GW; t++)
        R3DG_GLDS(uniform_src(lane, min(gb + t, P - 1), M, p), sb + SB_U + SB_USTRIDE * t, 4);
    if (VEC16) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int f = (j * 64 + lane) * 4, grp = f / 192, w = f % 192;
            size_t idx = ((size_t)min(gb + grp, P - 1) * K + k0) * 3 + w;
            idx = idx < 3 * total - 4 ? idx : 3 * total - 4;          // ragged last block: stay inside the array
            R3DG_GLDS(p.dirs + idx, sb + SB_DIRS + 256 * j, 16);
        }
        const int f = lane * 4, grp = f / 64, w = f % 64;
        size_t idx = (size_t)min(gb + grp, P - 1) * K + k0 + w;
        idx = idx < total - 4 ? idx : total - 4;
        R3DG_GLDS(p.vis + idx, sb + SB_VIS, 16);
        R3DG_GLDS(p.areas + idx, sb + SB_AREA, 16);
    } else {
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const int f = j * 64 + lane, grp = f / 192, w = f % 192;
            size_t idx = ((size_t)min(gb + grp, P - 1) * K + k0) * 3 + w;
            idx = idx < 3 * total - 1 ? idx : 3 * total - 1;
            R3DG_GLDS(p.dirs + idx, sb + SB_DIRS + 64 * j, 4);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int f = j * 64 + lane, grp = f / 64, w = f % 64;
            size_t idx = (size_t)min(gb + grp, P - 1) * K + k0 + w;
            idx = idx < total - 1 ? idx : total - 1;
            R3DG_GLDS(p.vis + idx, sb + SB_VIS + 64 * j, 4);
            R3DG_GLDS(p.areas + idx, sb + SB_AREA + 64 * j, 4);
        }
    }
}

// all DMA loads of this wave have landed (they are the only vector-memory loads in the steady-state loop)
__device__ __forceinline__ void wait_block_loads() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// =====================================================================================================================
// Forward, second formulation ("row" kernels): ONE WAVE PER GAUSSIAN, lane = sample.
//   * everything that is uniform per Gaussian -- the 48 SH coefficients and the view/normal/roughness-derived factors of
//     the GGX term -- is precomputed into a 64-float record by shade_prepare_kernel (thread per Gaussian); the wave fetches
//     it with ONE coalesced load (lane l reads element l, prefetched with the samples), parks it in LDS and reads it back as
//     wave-uniform broadcasts: 16 ds_read_b128 per 64 SAMPLES where the 16-lane kernel above issues 12 per sample, and no
//     redundant per-lane setup.  (Scalar loads into SGPRs were tried first: the three s_load_dwordx16 sit behind the same
//     lgkmcnt as the LDS texture reads and their latency was exposed once per row -- slower than the 16-lane kernel.)
//   * the K samples of the Gaussian are one coalesced row per array (lane k reads sample kb+k);
//   * the lat-long lookup of a cached direction never changes between visibility updates, so its result -- texel corner
//     and the two bilinear weights, 12 bytes per sample -- can be cached by the caller (r3dg_shade_build_taps, one pass
//     per visibility update): acos / atan2 / floor, a fifth of the 16-lane kernel's instructions, disappear;
//   * the training iteration reads only pbr, diffuse_light and the mean visibility of the 19 outputs (neilf.py:120-122):
//     NOUT = 7 accumulates and reduces just those;
//   * the environment texture is staged as one float4 per texel: a tap is ONE ds_read_b128 instead of three ds_read_b32.
// Measured (P=300k): K=64 0.27 -> see DESIGN.md; K=384 (relight) 2.4 ms -> see DESIGN.md.
// =====================================================================================================================
constexpr int REC = 64;      // floats per Gaussian record
// record layout: 0..47 SH coefficients (i*3+c, zero padded beyond M), 48..50 albedo, 51 roughness, 52..54 normal (as given),
// 55..57 V = normalize(viewdir), 58..60 N = normalize(normal) * sign(N.V), 61 NoV (clamped), 62 alpha^2, 63 k
// thread per Gaussian: the 16 derived entries 48..63 of the record ([P][16] floats).  The 48 coefficient entries are NOT
// copied: the row kernel's lanes 0..47 read them straight from `incidents` (one coalesced 192-byte run per Gaussian),
// lanes 48..63 read these 16 (a copying version of this kernel cost 0.05 ms per iteration for 115 MB of pure copy).
__global__ void __launch_bounds__(256)
shade_prepare_kernel(int n, const float* __restrict__ base_color, const float* __restrict__ roughness,
                     const float* __restrict__ normals, const float* __restrict__ viewdirs, float* __restrict__ rec16)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= n) return;
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
    float4* o = reinterpret_cast<float4*>(rec16 + (size_t)g * 16);
    o[0] = make_float4(u[48], u[49], u[50], u[51]);
    o[1] = make_float4(G.n[0], G.n[1], G.n[2], G.V[0]);
    o[2] = make_float4(G.V[1], G.V[2], G.N[0], G.N[1]);
    o[3] = make_float4(G.N[2], G.NoV, G.a2, G.kk);
}

// env == nullptr: the 12-byte lookup record (texel corner + weights).  env != nullptr: the looked-up RADIANCE itself (three
// floats) -- for a light that is not being trained (relighting under a fixed HDR map) the bilinear sample of every cached
// direction is a constant too, and the shading kernel then needs no texture access at all.
__global__ void __launch_bounds__(256)
shade_build_taps_kernel(size_t n, const float* __restrict__ dirs, const float* __restrict__ tr, int He, int We,
                        const float* __restrict__ env, uint32_t* __restrict__ taps)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const PackedTap t = make_tap(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], tr, He, We);
    if (env == nullptr) {
        taps[3 * i] = t.xy;
        taps[3 * i + 1] = __float_as_uint(t.wx1);
        taps[3 * i + 2] = __float_as_uint(t.wy1);
        return;
    }
    const int x0 = (int)(t.xy & 0xffffu) - 1, y0 = (int)(t.xy >> 16) - 1;
    const float wx[2] = {1.f - t.wx1, t.wx1}, wy[2] = {1.f - t.wy1, t.wy1};
    float e[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int x = x0 + b, y = y0 + a;
            if (x >= 0 && x <= We - 1 && y >= 0 && y <= He - 1) {
                const float* px = env + 3 * ((size_t)y * We + x);
                const float w = wy[a] * wx[b];
                e[0] += px[0] * w; e[1] += px[1] * w; e[2] += px[2] * w;
            }
        }
    taps[3 * i] = __float_as_uint(e[0]);
    taps[3 * i + 1] = __float_as_uint(e[1]);
    taps[3 * i + 2] = __float_as_uint(e[2]);
}

// bilinear sample with zero padding (grid_sample align_corners=True, padding_mode zeros) from a packed tap; the texture
// holds one float4 per texel (LDS or global)
__device__ __forceinline__ void env_fetch(const PackedTap& t, const float4* tex4, int He, int We, float (&e)[3],
                                          int (&tex)[4], float (&w)[4])
{
    const int x0 = (int)(t.xy & 0xffffu) - 1, y0 = (int)(t.xy >> 16) - 1;
    const float wx0 = 1.f - t.wx1, wy0 = 1.f - t.wy1;
    const bool xa = x0 >= 0, xb = x0 + 1 <= We - 1, ya = y0 >= 0, yb = y0 + 1 <= He - 1;    // x0 <= We-1, y0 <= He-1 always
    const int xc0 = xa ? x0 : 0, xc1 = xb ? x0 + 1 : We - 1, yc0 = ya ? y0 : 0, yc1 = yb ? y0 + 1 : He - 1;
    const float fx0 = xa ? wx0 : 0.f, fx1 = xb ? t.wx1 : 0.f, fy0 = ya ? wy0 : 0.f, fy1 = yb ? t.wy1 : 0.f;
    const int r0 = __mul24(yc0, We), r1 = __mul24(yc1, We);          // 24-bit operands (He, We <= 32767): full-rate multiply
    tex[0] = r0 + xc0; tex[1] = r0 + xc1; tex[2] = r1 + xc0; tex[3] = r1 + xc1;
    w[0] = fy0 * fx0; w[1] = fy0 * fx1; w[2] = fy1 * fx0; w[3] = fy1 * fx1;
    e[0] = e[1] = e[2] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float4 v = tex4[tex[q]];
        e[0] += v.x * w[q]; e[1] += v.y * w[q]; e[2] += v.z * w[q];
    }
}

constexpr int ROW_WAVES = 4;

struct RowSample {            // one lane's share of a (Gaussian, 64-sample block): prefetched one block ahead (registers)
    float dx, dy, dz, vis, area;
    PackedTap t;
    float rec;                // element `lane` of the Gaussian's 64-float record
    int row;                  // the Gaussian's row in every array (wave-uniform)
};

// `g` is wave-uniform (callers derive it from readfirstlane'd values): the row bases are computed on the scalar unit and the
// loads address SGPR base + 32-bit lane offset
template <int TAPS>
__device__ __forceinline__ RowSample load_row_sample(bool live, int g, int lane, unsigned k, int K,
                                                     const float* __restrict__ rec, const float* __restrict__ dirs,
                                                     const float* __restrict__ visibility,
                                                     const float* __restrict__ areas, float uniform_area,
                                                     const uint32_t* __restrict__ taps, int rec_stride = REC,
                                                     const float* __restrict__ incidents = nullptr, int M = 16)
{
    RowSample r;
    r.row = g;
    r.dx = 0.f; r.dy = 0.f; r.dz = 1.f; r.vis = 0.f; r.area = 0.f;
    r.t.xy = 0x00010001u; r.t.wx1 = 0.f; r.t.wy1 = 0.f;
    const size_t row = (size_t)g * (size_t)K;
    const float* __restrict__ drow = dirs + 3 * row;
    const float* __restrict__ vrow = visibility + row;
    if (incidents != nullptr) {
        // forward: lanes 0..47 <- the Gaussian's SH coefficients where they live, lanes 48..63 <- its 16 derived floats
        const int row3 = 3 * M;
        const float* __restrict__ src = lane < 48 ? incidents + (size_t)g * row3 + lane : rec + (size_t)g * 16 + (lane - 48);
        r.rec = (lane < 48 && lane >= row3) ? 0.f : *src;
    } else {
        r.rec = (rec + (size_t)g * rec_stride)[(unsigned)lane];
    }
    if (live) {
        const float3 d = *reinterpret_cast<const float3*>(drow + 3u * k);
        r.dx = d.x; r.dy = d.y; r.dz = d.z;
        r.vis = vrow[k];
        r.area = areas != nullptr ? (areas + row)[k] : uniform_area;
        if (TAPS) {
            const uint3 t = *reinterpret_cast<const uint3*>(taps + 3 * row + 3u * k);
            r.t.xy = t.x; r.t.wx1 = __uint_as_float(t.y); r.t.wy1 = __uint_as_float(t.z);
        }
    }
    return r;
}

// env4: the environment texture as one float4 per texel (launch_shade_forward pads it) -- in LDS when it fits (ENV_LDS),
// else read from global / L2 with ONE aligned 16-byte load per tap.
// Software pipeline per wave: [wait for block i's samples] -> [issue the loads of block i+1: they fly during the ~250
// instructions below] -> [record of block i: one ds_write_b32 per lane, read back as wave-uniform broadcasts] -> compute ->
// (last block of the Gaussian) transposing wave reduction + store.
template <int NOUT, bool ENV_LDS, int TAPS /* 0 lookup in kernel, 1 cached lookup, 2 cached radiance */, bool M16>
__global__ void __launch_bounds__(64 * ROW_WAVES)
shade_forward_row_kernel(int P, int K, int M_, const float* __restrict__ rec, const float* __restrict__ incidents,
                         const float4* __restrict__ env4, int He, int We,
                         const float* __restrict__ tr, const float* __restrict__ visibility,
                         const float* __restrict__ dirs, const float* __restrict__ areas, float uniform_area,
                         const uint32_t* __restrict__ taps, float* __restrict__ out)
{
    static_assert(NOUT == 7 || NOUT == 19, "training (pbr, diffuse_light, mean visibility) or all 19 outputs");
    constexpr int NV = NOUT == 7 ? 8 : 32;
    const int M = M16 ? 16 : M_;                 // degree-3 incident light (the reference's only configuration) folds the
                                                 // degree branches of the basis away
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    __shared__ __attribute__((aligned(16))) float s_rec[ROW_WAVES][REC];
    float4* s_env4 = reinterpret_cast<float4*>(s_mem);
    if (ENV_LDS && TAPS != 2) {
        for (int i = threadIdx.x; i < He * We; i += blockDim.x) s_env4[i] = env4[i];
        __syncthreads();
    }
    const float4* tex4 = ENV_LDS ? s_env4 : env4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* u = s_rec[wave];
    const float invK = 1.0f / (float)K;
    const int nblk = (K + 63) / 64;
    const int g_stride = gridDim.x * ROW_WAVES;
    // the wave walks (Gaussian, 64-sample block) pairs; two blocks are in flight ahead of the one being computed
    int g = blockIdx.x * ROW_WAVES + wave, kb = 0;
    auto advance = [&](int& ag, int& akb) {
        if (++akb == nblk) { akb = 0; ag += g_stride; }
    };
    auto fetch = [&](int ag, int akb) {
        const int k = akb * 64 + lane;
        const int gg = min(ag, P - 1);
        return load_row_sample<TAPS>(ag < P && k < K, gg, lane, (unsigned)k, K, rec, dirs, visibility, areas,
                                     uniform_area, taps, 16, incidents, M);
    };
    int g1 = g, kb1 = kb;
    advance(g1, kb1);
    RowSample cur = fetch(g, kb);
    RowSample nx1 = fetch(g1, kb1);
    float v[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) v[i] = 0.f;
    while (g < P) {
        // the current block's registers are complete past this point (its loads were issued two iterations ago); the
        // statement also keeps the compiler from hoisting the prefetch below above the wait
        asm volatile("" : "+v"(cur.dx), "+v"(cur.dy), "+v"(cur.dz), "+v"(cur.vis), "+v"(cur.area), "+v"(cur.t.xy),
                     "+v"(cur.t.wx1), "+v"(cur.t.wy1), "+v"(cur.rec) :: "memory");
        int g2 = g1, kb2 = kb1;
        advance(g2, kb2);
        const RowSample nx2 = fetch(g2, kb2);
        u[lane] = cur.rec;                          // same wave, in-order LDS: no barrier needed
        const float4 ga = *reinterpret_cast<const float4*>(u + 48);     // albedo, roughness
        const float4 gb = *reinterpret_cast<const float4*>(u + 52);     // n, V.x
        const float4 gc = *reinterpret_cast<const float4*>(u + 56);     // V.yz, N.xy
        const float4 gd = *reinterpret_cast<const float4*>(u + 60);     // N.z, NoV, a2, kk
        const float nx = gb.x, ny = gb.y, nz = gb.z, Vx = gb.w, Vy = gc.x, Vz = gc.y, Nx = gc.z, Ny = gc.w, Nz = gd.x;
        const float NoV = gd.y, a2 = gd.z, kk = gd.w;
        const float fd[3] = {ga.x / kPi, ga.y / kPi, ga.z / kPi};
        const float nom1 = NoV * (1.f - kk) + kk;
        {
            const float dx = cur.dx, dy = cur.dy, dz = cur.dz, vis = cur.vis, area = cur.area;
            const bool live = kb * 64 + lane < K;
            PackedTap t = cur.t;
            if (TAPS == 0) t = make_tap(dx, dy, dz, tr, He, We);
            float e[3], w4[4];
            int tex[4];
            if (TAPS == 2) {
                e[0] = __uint_as_float(t.xy); e[1] = t.wx1; e[2] = t.wy1;
            } else {
                env_fetch(t, tex4, He, We, e, tex, w4);
            }
            // local incident light: max(sum_i Y_i(d) c_i, 0); coefficients = wave-uniform broadcast reads (12 per 64 samples)
            float Y[16];
            sh_basis16(dx, dy, dz, M, Y);
            float l[3];
            sh_local_sum(u, Y, l);
            const float lv = live ? 1.f : 0.f;               // lanes beyond K contribute nothing (their area is 0 as well)
            const float loc[3] = {fmaxf(l[0], 0.f) * lv, fmaxf(l[1], 0.f) * lv, fmaxf(l[2], 0.f) * lv};
            const float glob[3] = {e[0] * vis, e[1] * vis, e[2] * vis};
            const float ndi = fmaxf(nx * dx + ny * dy + nz * dz, 0.f);
            const float area_ndi = area * ndi;
            // GGX specular (neilf.py:374-407)
            // x / max(|x|, 1e-12) (F.normalize) as x * rsq(max(|x|^2, 1e-24)): one transcendental instead of two
            const float dinv = __builtin_amdgcn_rsqf(fmaxf(dx * dx + dy * dy + dz * dz, 1e-24f));
            const float Lx = dx * dinv, Ly = dy * dinv, Lz = dz * dinv;
            const float ux = (Lx + Vx) / 2.0f, uy = (Ly + Vy) / 2.0f, uz = (Lz + Vz) / 2.0f;
            const float uinv = __builtin_amdgcn_rsqf(fmaxf(ux * ux + uy * uy + uz * uz, 1e-24f));
            const float Hx = ux * uinv, Hy = uy * uinv, Hz = uz * uinv;
            const float NoL = fminf(fmaxf(Nx * Lx + Ny * Ly + Nz * Lz, 1e-6f), 1.f);
            const float NoH = fminf(fmaxf(Nx * Hx + Ny * Hy + Nz * Hz, 1e-6f), 1.f);
            const float VoH = fminf(fmaxf(Vx * Hx + Vy * Hy + Vz * Hz, 1e-6f), 1.f);
            const float p2 = exp2f((-5.55473f * VoH - 6.98316f) * VoH);
            const float frac = (0.04f + 0.96f * p2) * a2;
            const float nom0 = NoH * NoH * (a2 - 1.f) + 1.f;
            const float nom2 = NoL * (1.f - kk) + kk;
            const float nom = fminf(fmaxf(4.f * kPi * nom0 * nom0 * nom1 * nom2, 1e-6f), 4.f * kPi);
            const float spec = frac / nom;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float lin = loc[c] + glob[c];
                const float transport = lin * area_ndi;
                v[c] += (fd[c] + spec) * transport;        // pbr
                v[3 + c] += transport;                     // diffuse_light
                if (NOUT == 19) {
                    v[6 + c] += spec * transport;          // specular
                    v[9 + c] += lin;                       // mean incident light
                    v[12 + c] += loc[c];
                    v[15 + c] += glob[c];
                }
            }
            v[NOUT == 19 ? 18 : 6] += vis;
        }
        if (kb == nblk - 1) {
            const float r = transpose_reduce<NV, true>(v);
            const int ch = transposed_channel<NV>(lane);
            if (transposed_owner<NV>(lane)) {
                if (NOUT == 19) {
                    if (ch < 19) out[(size_t)cur.row * SHADE_NOUT + ch] = r * invK;
                } else if (ch < 7) {
                    out[(size_t)cur.row * SHADE_NOUT + (ch < 6 ? ch : 18)] = r * invK;
                }
            }
#pragma unroll
            for (int i = 0; i < NV; i++) v[i] = 0.f;
        }
        cur = nx1;
        nx1 = nx2;
        g = g1; kb = kb1;
        g1 = g2; kb1 = kb2;
    }
}

__global__ void __launch_bounds__(256)
shade_pad_env_kernel(int n, const float* __restrict__ env, float4* __restrict__ env4)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) env4[i] = make_float4(env[3 * i], env[3 * i + 1], env[3 * i + 2], 0.f);
}

// Backward: gradients of sum(pbr*g_pbr) + sum(diffuse_light*g_diff) w.r.t. base_color, roughness, viewdirs,
// incidents and the (activated) environment texture.  normals / dirs / visibility carry no gradient in the
// reference (normal.detach(), cached samples).
// the largest of `n` non-negative floats (the block maxima of max|upstream gradient|; n == 1: grad_absmax_kernel's word),
// +inf = "some upstream gradient is not finite"; uniform over the wave.  This file is compiled with -ffast-math, which lets the
// compiler assume that no float is inf / nan and fold `x <= FLT_MAX` to true: everything that DECIDES on finiteness works on
// the bit patterns (non-negative floats order like unsigned integers; exponent all ones = inf / nan).
__device__ __forceinline__ bool not_finite_bits(unsigned int bits) { return (bits & 0x7f800000u) == 0x7f800000u; }
__device__ __forceinline__ unsigned int wave_gmax_bits(const unsigned int* __restrict__ gmax_bits, int n)
{
    unsigned int m = 0u;
    for (int i = threadIdx.x & 63; i < n; i += 64) {
        const unsigned int b = gmax_bits[i] & 0x7fffffffu;
        m = not_finite_bits(b) ? 0x7f800000u : (b > m ? b : m);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned int other = (unsigned int)__shfl_xor((int)m, o, 64);
        m = other > m ? other : m;
    }
    return m;
}
// fixed-point accumulation is possible for a positive, finite maximum
__device__ __forceinline__ bool gmax_usable(unsigned int bits) { return bits != 0u && bits < 0x7f800000u; }

// max(|g_pbr|, |g_diff|) over all Gaussians -> *out (as float bits; non-negative floats order like unsigned ints), +inf if any
// element is inf / nan.  Integer arithmetic on the bit patterns from the load on: under -ffast-math every float operation carries
// "no nan, no inf" flags and LLVM folds even the exponent-all-ones idiom on a float-derived value to false (measured: a NaN
// upstream gradient left a finite texture gradient).
__global__ void __launch_bounds__(256)
grad_absmax_kernel(int n, const float* __restrict__ a, const float* __restrict__ b, unsigned int* __restrict__ out)
{
    __shared__ unsigned int s_m[4];
    const unsigned int* ua = reinterpret_cast<const unsigned int*>(a);
    const unsigned int* ub = reinterpret_cast<const unsigned int*>(b);
    unsigned int m = 0u;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const unsigned int x = ua[i] & 0x7fffffffu, y = ub[i] & 0x7fffffffu;      // |.|; inf = 0x7f800000 < every nan
        m = max(m, max(x, y));
    }
    if (m > 0x7f800000u) m = 0x7f800000u;                                           // nan -> the "not finite" marker
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned int)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    // one atomic per block (a few hundred same-address atomics, not thousands)
    if (threadIdx.x == 0) atomicMax(out, max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3])));
}

// The environment-texture gradient is a scatter of 12 values per sample into a few hundred texels.  LDS *float*
// atomics (ds_add_f32) retire about one lane every 3 cycles on gfx950 -- measured: they alone were 60 % of this
// kernel -- while LDS *integer* atomics run at full rate.  The block therefore accumulates in signed 64-bit fixed
// point, scale = 2^35 / max|upstream gradient| (one small max-reduction kernel in front): a single contribution is
// bounded by max|g| * (1 + spec) * 2*pi <= max|g| * 2^13 for spec <= 1024 (larger ones are clamped), a block adds
// fewer than 2^14 of them per texel, so the sum stays below 2^62; the resolution is 3e-11 * max|g| -- finer than the
// fp32 accumulation it replaces -- and the per-block sum is order-independent.  Non-finite upstream gradients fall
// back to float atomics so NaN/inf still propagate.
template <bool ENV_LDS, bool VEC16, bool TAPS>
__global__ void __launch_bounds__(64 * SHADE_WAVES)
shade_backward_kernel(int P, int K, int M, ShadeSrc src, const float* __restrict__ env, int He, int We,
                      const float* __restrict__ tr, float* __restrict__ d_base, float* __restrict__ d_rough,
                      float* __restrict__ d_view, float* __restrict__ d_inc, float* __restrict__ d_env,
                      const unsigned int* __restrict__ gmax_bits, int gmax_n, const uint32_t* __restrict__ taps,
                      size_t total_samples)
{
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    const int ntex_raw = He * We * 3;
    const int ntex = ENV_LDS ? ((ntex_raw + 3) & ~3) : 0;
    float* s_env = s_mem;
    long long* s_denv = reinterpret_cast<long long*>(s_mem + ntex);      // 64-bit fixed-point accumulators
    const unsigned int gmax_word = wave_gmax_bits(gmax_bits, gmax_n);
    const float gmax = __uint_as_float(gmax_word);
    const bool fixed = ENV_LDS && gmax_usable(gmax_word);
    const float fx_scale = fixed ? 34359738368.0f / gmax : 0.f;          // 2^35 / max|g|
    const float fx_clamp = gmax * 8192.0f;                               // |contribution| <= max|g| * 2^13
    if (ENV_LDS) {
        // (four loads in flight per thread: as a plain loop every load is waited for before the next one is issued, and with a
        // few hundred listed Gaussians -- one pass per workgroup -- this prologue is a visible part of the kernel)
        for (int i0 = threadIdx.x; i0 < ntex_raw; i0 += 4 * (int)blockDim.x) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int i = i0 + j * (int)blockDim.x;
                v[j] = env[i < ntex_raw ? i : 0];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int i = i0 + j * (int)blockDim.x;
                if (i < ntex_raw) {
                    s_env[i] = v[j];
                    s_denv[i] = 0;
                }
            }
        }
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, l = lane & 15;
    __shared__ __attribute__((aligned(16))) float s_buf[SHADE_WAVES][2][SB_FLOATS];      // DMA double buffers
    // 12 floats per thread: the 3 SH sums of each of the lane's 4 samples (pass 0 -> pass 1), overwritten in place by the
    // 3 "dL/d local light" values (pass 1 -> pass 2); [slot][thread] so every access is lane-contiguous
    constexpr int PARK = 64 * SHADE_WAVES;
    __shared__ float s_park_all[12 * PARK];
    float* s_park = s_park_all + threadIdx.x;
    const float invK = 1.0f / (float)K;
    const int nblk = (K + 63) / 64;
    const int g_stride = gridDim.x * SH_GB;
    int gb = (blockIdx.x * SHADE_WAVES + wave) * SH_GW, kb = 0, buf = 0;
    if (gb < P) issue_block_loads<VEC16>(lane, gb, 0, P, K, M, src, s_buf[wave][0], total_samples);
    // per-lane accumulators over this lane's samples: 48 SH gradient channels (f = i*3 + c), albedo, roughness, view.
    // Each 64-sample block (4 samples per lane) is walked three times to keep the live register set small: pass 0
    // evaluates the SH sums of the local light, pass 1 the full sample + BRDF / view / env gradients, pass 2 rebuilds
    // the SH basis (40 instructions) and does the 48 SH-gradient FMAs.
    float acc[48];
    float accb[8];       // 0..2 albedo, 3 roughness, 4..6 view direction
#pragma unroll
    for (int i = 0; i < 48; i++) acc[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) accb[i] = 0.f;
    while (gb < P) {
        wait_block_loads();
        int ngb = gb, nkb = kb + 1;
        if (nkb == nblk) { nkb = 0; ngb = gb + g_stride; }
        if (ngb < P) issue_block_loads<VEC16>(lane, ngb, nkb * 64, P, K, M, src, s_buf[wave][buf ^ 1], total_samples);
        const float* sb = s_buf[wave][buf];
        const float* s_u = sb + SB_U + grp * SB_USTRIDE;
        const bool live = gb + grp < P;
        const int g = min(gb + grp, P - 1);                // row of this 16-lane group's Gaussian in every array
        GaussFwd G;
        gauss_setup(G, s_u);
        const float gp[3] = {s_u[58] * invK, s_u[59] * invK, s_u[60] * invK};
        const float gd[3] = {s_u[61] * invK, s_u[62] * invK, s_u[63] * invK};
        // cached lat-long lookups of this lane's four samples (r3dg_shade_build_taps): straight from global into registers,
        // issued here so that pass 0 below covers their latency
        PackedTap ct[4];
        if (TAPS) {
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int k = kb * 64 + l + SH_L * t;
                ct[t].xy = 0x00010001u; ct[t].wx1 = 0.f; ct[t].wy1 = 0.f;
                if (live && k < K) {
                    const uint3 q = *reinterpret_cast<const uint3*>(taps + 3 * ((size_t)g * K + k));
                    ct[t].xy = q.x; ct[t].wx1 = __uint_as_float(q.y); ct[t].wy1 = __uint_as_float(q.z);
                }
            }
        }
        // pass 0: SH sums of the local incident light (basis + 48 coefficient FMAs), parked in LDS
#pragma unroll 1
        for (int t = 0; t < 4; t++) {
            const int kl = l + SH_L * t, k = kb * 64 + kl;
            float sum[3] = {0.f, 0.f, 0.f};
            if (live && k < K) {
                const float* d = sb + SB_DIRS + (grp * 64 + kl) * 3;
                float Y[16];
                sh_basis16(d[0], d[1], d[2], M, Y);
                sh_local_sum(s_u, Y, sum);
            }
            s_park[(3 * t) * PARK] = sum[0];
            s_park[(3 * t + 1) * PARK] = sum[1];
            s_park[(3 * t + 2) * PARK] = sum[2];
        }
#pragma unroll 1
        for (int t = 0; t < 4; t++) {
            const int kl = l + SH_L * t, k = kb * 64 + kl;
            float dl[3] = {0.f, 0.f, 0.f};
            if (live && k < K) {
                const float* d = sb + SB_DIRS + (grp * 64 + kl) * 3;
                SampleFwd s;
                s.shsum[0] = s_park[(3 * t) * PARK];
                s.shsum[1] = s_park[(3 * t + 1) * PARK];
                s.shsum[2] = s_park[(3 * t + 2) * PARK];
                PackedTap mine = ct[0];                  // t is a run-time loop index (unroll 1): select, no scratch
                if (TAPS) {
                    if (t == 1) mine = ct[1];
                    if (t == 2) mine = ct[2];
                    if (t == 3) mine = ct[3];
                }
                shade_sample<ENV_LDS, true, TAPS>(s, G, s_u, M, d[0], d[1], d[2], sb[SB_VIS + grp * 64 + kl],
                                                         sb[SB_AREA + grp * 64 + kl], env, s_env, tr, He, We, &mine);
                float gspec = 0.f;
                float dlin[3];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float fd = G.base[c] / kPi;
                    const float dT = gp[c] * (fd + s.spec) + gd[c];       // dL/dtransport_c
                    gspec += gp[c] * s.transport[c];
                    accb[c] += gp[c] * s.transport[c] / kPi;              // albedo
                    dlin[c] = dT * s.area_ndi;                            // dL/d(incident light)_c
                    dl[c] = s.shsum[c] >= 0.f ? dlin[c] : 0.f;            // clamp_min(0): gradient where SH sum >= 0
                }
                // environment-texture gradient: 4 taps x 3 channels.  One (wave-divergent) branch per sample on the
                // visibility, nothing per tap: an out-of-range tap contributes weight 0 to texel 0.
                if (s.vis != 0.f) {
                    const float ev[3] = {dlin[0] * s.vis, dlin[1] * s.vis, dlin[2] * s.vis};
                    if (fixed) {
                        // float -> 64-bit fixed point in three instructions: in double, x * scale + 1.5 * 2^52 has the
                        // integer round(x * scale) in its low mantissa bits (two's complement, |.| < 2^51), and the
                        // bit pattern of the magic constant (0x4338 << 48) only occupies the high dword
                        const double scale_d = (double)fx_scale;
#pragma unroll
                        for (int tt = 0; tt < 4; tt++) {
                            const int tex = s.taps.idx[tt] >= 0 ? s.taps.idx[tt] : 0;
                            const float w = s.taps.idx[tt] >= 0 ? s.taps.w[tt] : 0.f;
#pragma unroll
                            for (int c = 0; c < 3; c++) {
                                const float cl = __builtin_amdgcn_fmed3f(ev[c] * w, -fx_clamp, fx_clamp);
                                const double dsum = __builtin_fma((double)cl, scale_d, 6755399441055744.0);
                                const unsigned long long bits =
                                    (unsigned long long)__double_as_longlong(dsum) - 0x4338000000000000ull;
                                atomicAdd(reinterpret_cast<unsigned long long*>(&s_denv[3 * tex + c]), bits);
                            }
                        }
                    } else {
#pragma unroll
                        for (int tt = 0; tt < 4; tt++) {
                            if (s.taps.idx[tt] >= 0) {
#pragma unroll
                                for (int c = 0; c < 3; c++)
                                    atomicAdd(&d_env[3 * (size_t)s.taps.idx[tt] + c], ev[c] * s.taps.w[tt]);
                            }
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
                const float da = dkk / 8.f + da2 * 2.f * G.a;
                accb[3] += dkk * 2.f / 8.f + da * 2.f * G.r;               // roughness
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
                for (int c = 0; c < 3; c++) accb[4 + c] += (dV[c] - G.V[c] * vd) / G.vlen;   // view direction
            }
            s_park[(3 * t) * PARK] = dl[0];
            s_park[(3 * t + 1) * PARK] = dl[1];
            s_park[(3 * t + 2) * PARK] = dl[2];
        }
#pragma unroll 1
        for (int t = 0; t < 4; t++) {
            const int kl = l + SH_L * t, k = kb * 64 + kl;
            if (live && k < K) {
                const float* d = sb + SB_DIRS + (grp * 64 + kl) * 3;
                float Y[16];
                sh_basis16(d[0], d[1], d[2], M, Y);
                const float dl[3] = {s_park[(3 * t) * PARK], s_park[(3 * t + 1) * PARK], s_park[(3 * t + 2) * PARK]};
#pragma unroll
                for (int f = 0; f < 48; f++) acc[f] += dl[f % 3] * Y[f / 3];
            }
        }
        if (kb == nblk - 1) {
            // four row reductions of 16 channels each: lane l ends with channel 16*pass + l
            float r[4];
#pragma unroll
            for (int pass = 0; pass < 3; pass++) {
                float v[16];
#pragma unroll
                for (int i = 0; i < 16; i++) v[i] = acc[16 * pass + i];
                r[pass] = row_transpose_reduce16(v);
            }
            {
                float v[16];
#pragma unroll
                for (int i = 0; i < 16; i++) v[i] = i < 8 ? accb[i] : 0.f;
                r[3] = row_transpose_reduce16(v);
            }
            if (live) {
#pragma unroll
                for (int pass = 0; pass < 3; pass++) {
                    const int f = 16 * pass + l;
                    if (f < 3 * M) d_inc[(size_t)g * M * 3 + f] = r[pass];
                }
                if (l < 3) d_base[3 * (size_t)g + l] = r[3];
                else if (l == 3) d_rough[g] = r[3];
                else if (l < 7) d_view[3 * (size_t)g + (l - 4)] = r[3];
            }
#pragma unroll
            for (int i = 0; i < 48; i++) acc[i] = 0.f;
#pragma unroll
            for (int i = 0; i < 8; i++) accb[i] = 0.f;
        }
        gb = ngb; kb = nkb; buf ^= 1;
    }
    if (fixed) {
        __syncthreads();
        const float inv = 1.0f / fx_scale;
        for (int i = threadIdx.x; i < ntex_raw; i += blockDim.x) {
            const long long v64 = s_denv[i];
            if (v64 != 0) atomicAdd(&d_env[i], (float)((double)v64 * (double)inv));
        }
    }
}

static int shade_cus()
{
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    // persistent grids leave g_reserve_cus CUs to a collective running beside them (common.hpp)
    return cus > opt(R3DG_OPT_RESERVE_CUS) ? cus - opt(R3DG_OPT_RESERVE_CUS) : 1;
}

static int shade_grid(int P, int blocks_per_cu = 2)
{
    const int cus = shade_cus();
    const int want = (P + SH_GB - 1) / SH_GB;
    const int cap = cus * blocks_per_cu;      // persistent blocks, all resident: nothing queued behind them
    return want < cap ? (want > 0 ? want : 1) : cap;
}

// per-device 256-byte scratch, allocated once and never freed: word 0 = max|upstream gradient| of the backward (reset per
// launch), bytes 64.. stay zero (the DMA source for absent elements of the uniform record)
static unsigned int* shade_scratch()
{
    static std::mutex mu;
    static std::map<int, unsigned int*> scratch;
    int dev = 0;
    R3DG_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    unsigned int*& p = scratch[dev];
    if (p == nullptr) {
        R3DG_HIP(hipMalloc((void**)&p, 256));
        R3DG_HIP(hipMemset(p, 0, 256));
    }
    return p;
}

// scratch of the row forward kernel ([P][16] derived floats + the float4-padded texture): one grow-only buffer per
// (device, stream), so launches on different streams or threads never share it (common.hpp stream_scratch)
static float* shade_records(hipStream_t s, size_t floats) { return (float*)stream_scratch(s, 0, floats * sizeof(float)); }

int g_shade_row_blocks_per_cu = 0;   // R3DG_OPT_SHADE_FWD_BLOCKS_PER_CU: persistent row blocks per CU, 0 = all that fit

void launch_shade_build_taps(hipStream_t s, size_t n, const float* dirs, const float* tr, int He, int We, const float* env,
                             uint32_t* taps)
{
    if (n == 0) return;
    shade_build_taps_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(n, dirs, tr, He, We, env, taps);
    check_launch(s, false, "shade_build_taps_kernel");
}

// `taps`: optional cache of r3dg_shade_build_taps for THESE dirs / env size / transform; `train_outputs`: write only
// pbr (0..2), diffuse_light (3..5) and the mean visibility (18) of the 19 outputs.
void launch_shade_forward(hipStream_t s, int P, int K, int M, const float* base_color, const float* roughness,
                          const float* normals, const float* viewdirs, const float* incidents, const float* env, int He,
                          int We, const float* tr, const float* visibility, const float* dirs, const float* areas,
                          float* out, const uint32_t* taps, bool train_outputs, float uniform_area, bool taps_are_radiance,
                          bool leave_room)
{
    if (P == 0) return;
    const size_t ntexel = (size_t)He * We;
    float* rec = shade_records(s, (((size_t)P * 16 + 3) & ~(size_t)3) + ntexel * 4);   // [P][16] derived floats, then the padded texture
    float4* env4 = reinterpret_cast<float4*>(rec + (((size_t)P * 16 + 3) & ~(size_t)3));
    const int n = P;
    shade_prepare_kernel<<<(n + 255) / 256, 256, 0, s>>>(n, base_color, roughness, normals, viewdirs, rec);
    shade_pad_env_kernel<<<(int)((ntexel + 255) / 256), 256, 0, s>>>((int)ntexel, env, env4);
    const int mode = taps == nullptr ? 0 : (taps_are_radiance ? 2 : 1);
    const bool lds = mode != 2 && He * We * 4 <= ENV_LDS_MAX;          // float4 per texel
    const size_t smem = lds ? ntexel * sizeof(float4) : 0;
    const int want = (n + ROW_WAVES - 1) / ROW_WAVES;
#define R3DG_ROW(N, L, T)                                                                                             \
    do {                                                                                                              \
        /* persistent blocks: exactly as many as are resident at once (a larger grid runs a second, nearly empty wave of \
           blocks; the API can be one block per CU high at this SGPR count -- a spare block only costs a short tail) */ \
        static int per_cu[2] = {0, 0};                                                                                \
        if (per_cu[L] == 0) {                                                                                         \
            int nb = 0;                                                                                               \
            R3DG_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, shade_forward_row_kernel<N, L, T, true>,       \
                                                                  64 * ROW_WAVES, smem));                            \
            hipFuncAttributes fa;                                                                                     \
            R3DG_HIP(hipFuncGetAttributes(&fa, (const void*)shade_forward_row_kernel<N, L, T, true>));                \
            const int by_vgpr = 512 / (((fa.numRegs + 7) / 8) * 8);                                                   \
            nb = nb < by_vgpr ? nb : by_vgpr;                                                                         \
            per_cu[L] = nb > 0 ? (nb < 8 ? nb : 8) : 1;                                                               \
        }                                                                                                             \
        int bpc = opt(R3DG_OPT_SHADE_FWD_BLOCKS_PER_CU) > 0 && opt(R3DG_OPT_SHADE_FWD_BLOCKS_PER_CU) < per_cu[L] ? opt(R3DG_OPT_SHADE_FWD_BLOCKS_PER_CU)   \
                                                                                         : per_cu[L];                 \
        /* beside the instance ordering (fused iteration): 3 of the ~6 resident blocks per CU measured best for the     \
           iteration as a whole (2.05 -> 1.97 ms: the ordering kernels get CU time earlier) */                        \
        if (leave_room && opt(R3DG_OPT_SHADE_FWD_BLOCKS_PER_CU) == 0 && bpc > 3) bpc = 3;                                         \
        const int cap = shade_cus() * bpc;                                                                            \
        const int grid = want < cap ? want : cap;                                                                     \
        if (M == 16)                                                                                                  \
            shade_forward_row_kernel<N, L, T, true><<<grid, 64 * ROW_WAVES, smem, s>>>(                               \
                n, K, M, rec, incidents, env4, He, We, tr, visibility, dirs, areas, uniform_area, taps, out);         \
        else                                                                                                          \
            shade_forward_row_kernel<N, L, T, false><<<grid, 64 * ROW_WAVES, smem, s>>>(                              \
                n, K, M, rec, incidents, env4, He, We, tr, visibility, dirs, areas, uniform_area, taps, out);         \
    } while (0)
#define R3DG_ROW_MODE(N, L)                                                                                           \
    do {                                                                                                              \
        if (mode == 0) R3DG_ROW(N, L, 0); else if (mode == 1) R3DG_ROW(N, L, 1); else R3DG_ROW(N, L, 2);              \
    } while (0)
    if (train_outputs) { if (lds) R3DG_ROW_MODE(7, true); else R3DG_ROW_MODE(7, false); }
    else { if (lds) R3DG_ROW_MODE(19, true); else R3DG_ROW_MODE(19, false); }
#undef R3DG_ROW_MODE
#undef R3DG_ROW
}

void launch_shade_backward(hipStream_t s, int P, int K, int M, const float* base_color, const float* roughness,
                           const float* normals, const float* viewdirs, const float* incidents, const float* env,
                           int He, int We, const float* tr, const float* visibility, const float* dirs,
                           const float* areas, const float* g_pbr, const float* g_diff, float* d_base, float* d_rough,
                           float* d_view, float* d_inc, float* d_env, const uint32_t* taps, const float* block_absmax,
                           int n_block_absmax)
{
    unsigned int* scratch = shade_scratch();
    // scale of the fixed-point texture accumulation: max |upstream gradient|, either handed over as block maxima by the
    // producer of g_pbr / g_diff (r3dg_stage2_unpack_gradients) or reduced here
    const unsigned int* gmax = scratch;
    int gmax_n = 1;
    if (block_absmax != nullptr && n_block_absmax > 0) {
        gmax = reinterpret_cast<const unsigned int*>(block_absmax);
        gmax_n = n_block_absmax;
    } else {
        R3DG_HIP(hipMemsetAsync(scratch, 0, 4, s));
        const int nb = (3 * P + 255) / 256;
        grad_absmax_kernel<<<nb < 256 ? nb : 256, 256, 0, s>>>(3 * P, g_pbr, g_diff, scratch);
    }
    const int ntex = He * We * 3;
    const int n = P;
    const ShadeSrc src = {base_color, roughness, normals, viewdirs, incidents, g_pbr, g_diff,
                          reinterpret_cast<const float*>(scratch + 16), dirs, visibility, areas};
    // persistent blocks so the LDS-privatised env gradient is flushed once per block, not once per Gaussian
    const int grid = shade_grid(n);
    const bool lds = 3 * ntex <= ENV_LDS_MAX, vec = (K % 4) == 0 && (size_t)P * K >= 4;
    const size_t smem = lds ? 3 * ((ntex + 3) & ~3) * sizeof(float) : 0;  // + the static DMA buffers and parking slots
#define R3DG_SB3(L, V, T)                                                                                             \
    do {                                                                                                              \
        if (smem > 65536)                                                                                             \
            R3DG_HIP(hipFuncSetAttribute((const void*)shade_backward_kernel<L, V, T>,                                 \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));                     \
        shade_backward_kernel<L, V, T><<<grid, 64 * SHADE_WAVES, smem, s>>>(n, K, M, src, env, He, We, tr, d_base,    \
                                                                          d_rough, d_view, d_inc, d_env, gmax,        \
                                                                          gmax_n, taps, (size_t)P * K);               \
    } while (0)
#define R3DG_SB(L, V)                                                                                                 \
    do {                                                                                                              \
        if (taps != nullptr) R3DG_SB3(L, V, true); else R3DG_SB3(L, V, false);                                        \
    } while (0)
    if (lds) { if (vec) R3DG_SB(true, true); else R3DG_SB(true, false); }
    else { if (vec) R3DG_SB(false, true); else R3DG_SB(false, false); }
#undef R3DG_SB
#undef R3DG_SB3
    check_launch(s, false, "shade_backward_kernel");
}

}  // namespace r3dg
#include "shading_transport.hpp"
#include "shading_split.hpp"
#include "shading_frs.hpp"
namespace r3dg {

// ---- fixed ray set (shading_frs.hpp) -------------------------------------------------------------------------------------
size_t shade_frs_table_floats(int K) { return (size_t)((K + 15) / 16) * 512; }

void launch_shade_frs_build_tables(hipStream_t s, int K, const float* zsamples, float* tables)
{
    const int nblk = (K + 15) / 16;
    frs_build_tables_kernel<<<(nblk * 512 + 255) / 256, 256, 0, s>>>(K, nblk, zsamples, tables);
    check_launch(s, false, "frs_build_tables_kernel");
}

void launch_shade_frs_classify(hipStream_t s, int P, const float* ray_normals, uint8_t* valid)
{
    if (P == 0) return;
    frs_classify_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, ray_normals, valid);
    check_launch(s, false, "frs_classify_kernel");
}

void launch_shade_frs_build_taps(hipStream_t s, int P, int K, const float* ray_normals, const float* zsamples, int He, int We,
                                 uint32_t* taps)
{
    const size_t n = (size_t)P * K;
    if (n == 0) return;
    frs_build_taps_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(P, K, ray_normals, zsamples, He, We, taps);
    check_launch(s, false, "frs_build_taps_kernel");
}

// can the fixed-ray-set kernels take this configuration?  (everything else goes through the general kernels)
bool shade_frs_supported(int K, int M, int He, int We)
{
    return M == 16 && K >= 4 && (K % 4) == 0 && He <= 511 && We <= 511 &&
           (size_t)He * We * (16 + 24) <= (size_t)ENV_LDS_MAX * 4;
}

// dynamic LDS of the two kernels: the texture as float4 texels (+ its 3 x 64-bit gradient accumulators), the per-wave staging
// areas of the next group's per-Gaussian data, the backward's table words when they fit
static size_t frs_forward_lds_bytes(int He, int We)
{
    return ((size_t)He * We * 4 + (size_t)FRS_WAVES * FRS_ST_FWD) * sizeof(float);
}
static size_t frs_backward_lds_bytes(int K, int He, int We)
{
    const size_t ntexel = (size_t)He * We, nblk = (size_t)(K + 15) / 16;
    return (((10 * ntexel + 3) & ~(size_t)3) + (size_t)FRS_WAVES * FRS_ST_BWD + (K <= FRS_TAB_LDS_MAX_K ? nblk * 512 : 0)) *
           sizeof(float);
}

static int frs_grid(int P, const void* kernel, size_t smem)
{
    // resident workgroups per CU of (kernel, LDS size): asked once per device (the attribute / occupancy calls take the
    // runtime's locks on every launch otherwise)
    static std::mutex mu;
    static std::map<std::tuple<int, const void*, size_t>, int> cache;
    int dev = 0;
    R3DG_HIP(hipGetDevice(&dev));
    int nb = 0;
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = cache.find(std::make_tuple(dev, kernel, smem));
        if (it != cache.end()) {
            nb = it->second;
        } else {
            if (smem > 65536)
                R3DG_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            R3DG_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, 64 * FRS_WAVES, smem));
            hipFuncAttributes fa;
            R3DG_HIP(hipFuncGetAttributes(&fa, kernel));
            const int by_vgpr = 512 / (((fa.numRegs + 7) / 8) * 8);   // waves per SIMD = 256-thread blocks per CU
            nb = nb < by_vgpr ? nb : by_vgpr;
            nb = nb > 0 ? (nb < 8 ? nb : 8) : 1;
            cache[std::make_tuple(dev, kernel, smem)] = nb;
        }
    }
    const int want = ((P + FRS_G - 1) / FRS_G + FRS_WAVES - 1) / FRS_WAVES;
    const int cap = shade_cus() * nb;
    return want < cap ? (want > 0 ? want : 1) : cap;
}

// every sample of the Fibonacci set carries the same area, 2 pi (fibonacci_sphere_sampling, utils/graphics_utils.py:26-37);
// uniform_area == 0 means that value
static inline float frs_area(float uniform_area) { return uniform_area > 0.f ? uniform_area : 6.283185307179586f; }

// A fixed-ray-set call is three groups of launches, timed as three stages by the C ABI (capi.hip) so that the profile's
// "shade_forward" / "shade_backward" rows are ONE kernel each:
//   aux     the coefficient rotation (forward: incidents -> cprime, kept for the backward; backward: dcprime -> d_inc), the
//           max |upstream gradient| reduction when the caller has none
//   main    the MFMA kernel for the Gaussians on the rotated path
//   listed  the wave-per-Gaussian kernels for the listed rest
void launch_shade_frs_forward_aux(hipStream_t s, int P, const float* incidents, const float* ray_normals, float* cprime)
{
    if (P == 0) return;
    frs_rotate_kernel<false><<<(P + 255) / 256, 256, 0, s>>>(P, ray_normals, incidents, cprime, nullptr);
    check_launch(s, false, "frs_rotate_kernel");
}

void launch_shade_frs_forward_main(hipStream_t s, int P, int K, const float* base_color, const float* roughness,
                                   const float* normals, const float* viewdirs, const float* env, int He, int We,
                                   const float* visibility, float uniform_area, const uint32_t* taps, const float* ray_normals,
                                   const float* tables, const uint8_t* valid, const float* cprime, bool leave_room, float* out,
                                   float* feat)
{
    if (P == 0) return;
    const size_t smem = frs_forward_lds_bytes(He, We);
    int grid = frs_grid(P, (const void*)shade_forward_frs_kernel, smem);
    // beside the instance ordering (fused iteration) ONE workgroup per CU: the ordering chain (projection -> binning -> tile sort) is
    // the longer of the two concurrent paths and every wave this kernel keeps resident slows it -- measured per CU cap: 1 -> 618-627,
    // 2 -> 598-610, 3 -> 597-607 it/s (this kernel alone 0.21 / 0.195 / 0.21 ms; a high-priority ordering stream: no effect)
    // That holds while this kernel is the SHORTER path.  With more samples it becomes the longer one and the cap costs more
    // than it buys: 300k x 384 samples 339 (one per CU) / 365 (two) / 365 (three) it/s, 2M x 64 samples 152 / 158 / 155 -- two
    // per CU above 40 M samples per launch.  (R3DG_OPT_SHADE_FWD_BLOCKS_PER_CU > 0 replaces the choice for A/B runs)
    if (leave_room) {
        const int by_size = (long long)P * K > 40000000ll ? 2 : 1;
        const int per_cu = opt(R3DG_OPT_SHADE_FWD_BLOCKS_PER_CU) > 0 ? opt(R3DG_OPT_SHADE_FWD_BLOCKS_PER_CU) : by_size;
        grid = grid > per_cu * shade_cus() ? per_cu * shade_cus() : grid;
    }
    const FrsSrc src = {base_color, roughness, normals, viewdirs, ray_normals, cprime, nullptr, nullptr, visibility, taps};
    shade_forward_frs_kernel<<<grid, 64 * FRS_WAVES, smem, s>>>(P, K, src, env, He, We, frs_area(uniform_area), tables, valid, out,
                                                                feat);
    check_launch(s, false, "shade_forward_frs_kernel");
}

// grid of the listed kernels: one wave per Gaussian up to a few hundred waves, grid-stride beyond
static int frs_listed_grid(int n_list)
{
    const int want = (n_list + FRS_LISTED_WAVES - 1) / FRS_LISTED_WAVES;
    const int cap = shade_cus();
    return want < cap ? (want > 0 ? want : 1) : cap;
}

void launch_shade_frs_forward_listed(hipStream_t s, int K, const float* base_color, const float* roughness,
                                     const float* normals, const float* viewdirs, const float* incidents, const float* env, int He,
                                     int We, const float* visibility, const float* ray_normals, const float* zsamples,
                                     float uniform_area, const int* invalid_list, int n_invalid, float* out, float* feat)
{
    if (n_invalid <= 0) return;
    const FrsSrc src = {base_color, roughness, normals, viewdirs, ray_normals, nullptr, nullptr, nullptr, visibility, nullptr};
    const size_t smem = ((((size_t)3 * He * We + 3) & ~(size_t)3) + 64 * FRS_LISTED_WAVES) * sizeof(float);
    shade_forward_frs_listed_kernel<<<frs_listed_grid(n_invalid), 64 * FRS_LISTED_WAVES, smem, s>>>(
        n_invalid, invalid_list, K, src, incidents, env, He, We, zsamples, frs_area(uniform_area), out, feat);
    check_launch(s, false, "shade_forward_frs_listed_kernel");
}

// (before _main) -> the words the main kernel scales its fixed-point texture accumulation by
const unsigned int* launch_shade_frs_backward_aux(hipStream_t s, int P, const float* g_pbr, const float* g_diff,
                                                  const float* block_absmax, int n_block_absmax, int* gmax_n)
{
    unsigned int* scratch = shade_scratch();
    const unsigned int* gmax = scratch;
    *gmax_n = 1;
    if (block_absmax != nullptr && n_block_absmax > 0) {
        gmax = reinterpret_cast<const unsigned int*>(block_absmax);
        *gmax_n = n_block_absmax;
    } else {
        R3DG_HIP(hipMemsetAsync(scratch, 0, 4, s));
        const int nb = (3 * P + 255) / 256;
        grad_absmax_kernel<<<nb < 256 ? nb : 256, 256, 0, s>>>(3 * P, g_pbr, g_diff, scratch);
    }
    return gmax;
}

void launch_shade_frs_backward_main(hipStream_t s, int P, int K, const float* base_color, const float* roughness,
                                    const float* normals, const float* viewdirs, const float* env, int He, int We,
                                    const float* visibility, float uniform_area, const uint32_t* taps, const float* ray_normals,
                                    const float* tables, const uint8_t* valid, const float* cprime, float* dcp, const float* g_pbr,
                                    const float* g_diff, float* d_base, float* d_rough, float* d_view, float* d_env,
                                    const unsigned int* gmax, int gmax_n)
{
    if (P == 0) return;
    const bool tab_lds = K <= FRS_TAB_LDS_MAX_K;
    const size_t smem = frs_backward_lds_bytes(K, He, We);
    const FrsSrc src = {base_color, roughness, normals, viewdirs, ray_normals, cprime, g_pbr, g_diff, visibility, taps};
    if (tab_lds) {
        const int grid = frs_grid(P, (const void*)shade_backward_frs_kernel<true>, smem);
        shade_backward_frs_kernel<true><<<grid, 64 * FRS_WAVES, smem, s>>>(
            P, K, src, env, He, We, frs_area(uniform_area), tables, valid, d_base, d_rough, d_view, dcp, d_env, gmax, gmax_n);
    } else {
        const int grid = frs_grid(P, (const void*)shade_backward_frs_kernel<false>, smem);
        shade_backward_frs_kernel<false><<<grid, 64 * FRS_WAVES, smem, s>>>(
            P, K, src, env, He, We, frs_area(uniform_area), tables, valid, d_base, d_rough, d_view, dcp, d_env, gmax, gmax_n);
    }
    check_launch(s, false, "shade_backward_frs_kernel");
}

// gradient back to the unrotated coefficients.  valid == nullptr: every row of d_inc is written; valid != nullptr: only the rows
// on the rotated path are written (the listed Gaussians' rows come from their own kernel, possibly on another stream)
void launch_shade_frs_backward_rotate(hipStream_t s, int P, const float* ray_normals, const float* dcp, float* d_inc,
                                      const uint8_t* valid)
{
    if (P == 0) return;
    frs_rotate_kernel<true><<<(P + 255) / 256, 256, 0, s>>>(P, ray_normals, dcp, d_inc, valid);
    check_launch(s, false, "frs_rotate_kernel");
}

void launch_shade_frs_incident_chain(hipStream_t s, int P, const float* ray_normals, const uint8_t* valid, const float* dcp,
                                     float* d_inc, float* incidents, float* exp_avg, float* exp_avg_sq, float* cprime, float lr,
                                     float lr_tail, float beta1, float beta2, float eps, int step, float grad_scale,
                                     const float* skip_flag, int listed_in_dcprime)
{
    if (P == 0) return;
    // (bias corrections exactly as launch_adam forms them, stage2_glue.hip)
    const double b1 = 1.0 - pow((double)beta1, (double)step), b2 = 1.0 - pow((double)beta2, (double)step);
    FrsAdam a = {lr, lr_tail, beta1, beta2, eps, (float)b1, (float)(1.0 / sqrt(b2)), grad_scale, 1.f - beta1, 1.f - beta2};
    // 49 KB of LDS per workgroup = three per CU: measured against two (56 KB) and one (81 KB) per CU, which are gentler on the
    // activation / projection kernels running beside it but make the chain the longer path: 805 / 797 / 771 it/s
    const size_t lds = 4 * 64 * FRS_CHAIN_LD * sizeof(float);
    // (49 KB of dynamic LDS: under the 64 KB every launch may ask for, so no per-device function attribute is needed)
    frs_incident_chain_kernel<<<(P + 255) / 256, 256, lds, s>>>(P, ray_normals, valid, dcp, d_inc, incidents, exp_avg, exp_avg_sq,
                                                                cprime, a, skip_flag, listed_in_dcprime);
    check_launch(s, false, "frs_incident_chain_kernel");
}

void launch_shade_frs_backward_listed(hipStream_t s, int K, const float* base_color, const float* roughness,
                                      const float* normals, const float* viewdirs, const float* incidents, const float* env, int He,
                                      int We, const float* visibility, const float* ray_normals, const float* zsamples,
                                      float uniform_area, const int* invalid_list, int n_invalid, const float* g_pbr,
                                      const float* g_diff, float* d_base, float* d_rough, float* d_view, float* d_inc, float* d_env,
                                      const unsigned int* gmax, int gmax_n)
{
    if (n_invalid <= 0) return;
    const FrsSrc src = {base_color, roughness, normals, viewdirs, ray_normals, nullptr, g_pbr, g_diff, visibility, nullptr};
    const size_t smem = (3 * (((size_t)3 * He * We + 3) & ~(size_t)3) + 64 * FRS_LISTED_WAVES) * sizeof(float);
    // (the texture gradient is flushed once per workgroup: a quarter of the forward's grid keeps that under the kernel's own time)
    int grid = frs_listed_grid(n_invalid);
    grid = grid > 64 ? 64 + (grid - 64) / 4 : grid;
    if (smem > 65536)
        R3DG_HIP(hipFuncSetAttribute((const void*)shade_backward_frs_listed_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)smem));
    shade_backward_frs_listed_kernel<<<grid, 64 * FRS_LISTED_WAVES, smem, s>>>(
        n_invalid, invalid_list, K, src, incidents, env, He, We, zsamples, frs_area(uniform_area), d_base, d_rough, d_view, d_inc,
        d_env, gmax, gmax_n);
    check_launch(s, false, "shade_backward_frs_listed_kernel");
}

void launch_shade_build_split(hipStream_t s, int P, int K, const int* perm, const float* normals, const float* incidents,
                              const float* visibility, const float* dirs, const float* zsamples, float uniform_area, float* lt,
                              float* vis_t, float* consts)
{
    if (P == 0) return;
    shade_build_split_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, K, perm, normals, incidents, visibility, dirs, zsamples,
                                                           frs_area(uniform_area), reinterpret_cast<float4*>(lt), vis_t, consts);
    check_launch(s, false, "shade_build_split_kernel");
}

void launch_shade_forward_split(hipStream_t s, int P, int K, const int* perm, const float* base_color, const float* roughness,
                                const float* normals, const float* viewdirs, const float* lt, const float* vis_t,
                                const float* consts, const float* zsamples, const float* tr, const float* env_fp, int He, int We,
                                float* out)
{
    if (P == 0) return;
    // sample range split into parts (shading_split.hpp): ~12+ waves per SIMD in the launch, parts a multiple of 4 samples long
    const int waves = (P + 63) / 64;
    int parts = (12 * 4 * shade_cus() + waves - 1) / waves;
    parts = parts < 1 ? 1 : (parts > 8 ? 8 : parts);
    int Kp = ((K + parts - 1) / parts + 3) & ~3;
    parts = (K + Kp - 1) / Kp;
    float4* partial = reinterpret_cast<float4*>(stream_scratch(s, 1, (size_t)parts * P * 3 * sizeof(float4)));
    const dim3 grid((P + 255) / 256, parts);
    shade_forward_split_kernel<<<grid, 256, 0, s>>>(P, K, Kp, perm, base_color, roughness, normals, viewdirs,
                                                   reinterpret_cast<const float4*>(lt), vis_t, zsamples, tr,
                                                   reinterpret_cast<const float4*>(env_fp), He, We, partial);
    shade_split_combine_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, K, parts, perm, base_color, partial, consts, out);
    check_launch(s, false, "shade_forward_split_kernel");
}

void launch_shade_env_footprints(hipStream_t s, int He, int We, const float* env, float* fp)
{
    const int n = (He + 1) * (We + 1);
    shade_env_footprints_kernel<<<(n + 255) / 256, 256, 0, s>>>(He, We, env, reinterpret_cast<float4*>(fp));
    check_launch(s, false, "shade_env_footprints_kernel");
}

void launch_shade_build_transport(hipStream_t s, int P, int K, int M, const float* normals, const float* incidents,
                                  const float* visibility, const float* dirs, const float* areas, float uniform_area,
                                  float* radiance_to_transport, float* consts)
{
    if (P == 0) return;
    shade_build_transport_kernel<<<(P + TR_WAVES - 1) / TR_WAVES, 64 * TR_WAVES, 0, s>>>(
        P, K, M, normals, incidents, visibility, dirs, areas, uniform_area, radiance_to_transport, consts);
    check_launch(s, false, "shade_build_transport_kernel");
}

void launch_shade_forward_transport(hipStream_t s, int P, int K, const float* base_color, const float* roughness,
                                    const float* normals, const float* viewdirs, const float* transport, const float* consts,
                                    const float* zsamples, const float* dirs, float* out)
{
    if (P == 0) return;
    shade_forward_transport_kernel<<<(P + TR_WAVES - 1) / TR_WAVES, 64 * TR_WAVES, 0, s>>>(
        P, K, base_color, roughness, normals, viewdirs, transport, consts, zsamples, dirs, out);
    check_launch(s, false, "shade_forward_transport_kernel");
}


}  // namespace r3dg
