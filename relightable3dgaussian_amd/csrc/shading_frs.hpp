// Shading integral over a FIXED RAY SET ("frs"): forward and backward kernels for callers whose cached incident directions
// are the Fibonacci set rotated to each Gaussian's normal -- what GaussianModel.update_visibility produces
// (scene/gaussian_model.py:312-342: sample_incident_rays(normal, False, K) = fibonacci_sphere_sampling(random_rotate=False),
// utils/graphics_utils.py:9-37; d_k = normalize(R(n) z_k), R = rotation_between_z, utils/sh_utils.py:36-68).
// Included by shading.hip (after the general kernels, whose per-sample code it reuses).
//
// Why.  The general kernels evaluate the 16 SH basis functions at every cached direction and contract them with the 48
// incident-light coefficients -- forward: once; backward: twice (value and sign of the local light, then the gradient) --
// 80 of the forward's 260 and 225 of the backward's 540 VALU instructions per sample, on kernels that are bound by VALU issue.
// With d_k = R z_k the local light is   sum_i c_i Y_i(R z_k) = sum_i c'_i Y_i(z_k)   for the coefficients c' of the rotated
// function (an orthogonal map per SH band), and Y_i(z_k) is ONE [K x 16] table for all Gaussians.  Both contractions
//       l[k][c]   = sum_i Yz[k][i] c'[i][c]          (local light of sample k, channel c)
//       dc'[i][c] = sum_k Yz[k][i] dl[k][c]          (gradient w.r.t. the rotated coefficients)
// are then small dense products against a CONSTANT matrix and run on the matrix cores: v_mfma_f32_16x16x4_f32 -- exact fp32
// FMA chains at the vector rate, on the MFMA pipe, BESIDE the VALU work of the other waves -- with the 16 columns of the
// MFMA tile = 16 Gaussians.  That fixes the work split: lane (g = lane & 15, q = lane >> 4) owns samples 16 b + 4 q + v
// (v < 4) of Gaussian g0 + g in every 16-sample block b, i.e. 4 lanes per Gaussian and 16 samples per lane per 64 samples
// -- which also amortises the per-Gaussian set-up and the final cross-lane sums over 4x more samples than the 16-lane
// kernel (they were a sixth of its time: profiles/r03_ablate_shade_backward.json).
//
// The rotation itself is done per Gaussian by two streaming kernels (coefficients there, gradient back) with the sampling
// construction of tools/gen_sh_rotation_tables.py.  R is evaluated in fp32 exactly as the ray set was generated; where
// cancellation makes it measurably non-orthonormal (normals within ~2.5 degrees of -z: 0.3 % of uniformly distributed
// normals) the directions are not a rigid copy of the z set and the Gaussian is left to wave-per-Gaussian kernels at the end of
// this file, which the launcher runs on the list of those Gaussians (same entry point).
//
// Round 4: NO per-sample direction is read any more (the [P,K,3] cache is replaced by the 12-byte ray-normal
// snapshot).  Round 3's kernels still streamed the cached direction (12 B) and a 12-byte lookup record beside the visibility
// (4 B): 28 B per sample, 1.85x / 1.81x of their algorithmic bytes in HBM traffic, on kernels that were latency-bound.  With
// d_k = R z_k / |R z_k| every per-sample dot product is a product against the z table (see FrsFrame below), the half vector
// drops out of the GGX terms, and the lookup record is 8 bytes (frs_pack_axis): 12 B per sample, ~20 VALU instructions fewer
// per sample in the forward and ~45 in the backward.
#pragma once
#include "sh_rotation_tables.hpp"

namespace r3dg {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int FRS_WAVES = 4;                 // waves per workgroup
constexpr int FRS_G = 16;                    // Gaussians per wave step (the N dimension of the MFMA tile)
constexpr float FRS_MAX_DEFECT = 2e-5f;      // max |R R^T - I| for which a Gaussian takes the rotated path
constexpr int FRS_TAB_LDS_MAX_K = 128;       // backward: the Y_i(z_k) tables live in LDS up to this many samples (16 KB)

// rotation_between_z(n) (utils/sh_utils.py:36-68), fp32 operation for operation as sampling.rotation_between_z
__device__ __forceinline__ void frs_rotation(const float n0, const float n1, const float n2, float (&R)[9])
{
    const float v1 = -n1, v2 = n0, cp = fmaxf(n2 + 1.f, 1e-7f);
    const bool regular = n2 + 1.f > 0.f;
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

// valid[g] = 1 when R(n_g) is orthonormal to FRS_MAX_DEFECT (then normalize(R z_k) is a rigid copy of the z set)
__global__ void __launch_bounds__(256)
frs_classify_kernel(int P, const float* __restrict__ ray_normals, uint8_t* __restrict__ valid)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= P) return;
    float R[9];
    frs_rotation(ray_normals[3 * (size_t)g], ray_normals[3 * (size_t)g + 1], ray_normals[3 * (size_t)g + 2], R);
    float defect = 0.f;
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) {
            const float d = R[3 * a] * R[3 * b] + R[3 * a + 1] * R[3 * b + 1] + R[3 * a + 2] * R[3 * b + 2] - (a == b ? 1.f : 0.f);
            defect = fmaxf(defect, fabsf(d));
        }
    valid[g] = defect <= FRS_MAX_DEFECT ? 1 : 0;            // (NaN normals compare false: invalid)
}

// Table of Y_i(z_k) in the two MFMA operand layouts.  Per 16-sample block b (samples 16 b .. 16 b + 15) eight 64-float
// slots: slot s < 4 = A operand of the local-light product, lane (r = lane & 15, q = lane >> 4) holds Yz[16 b + r][4 s + q];
// slot 4 + v = A operand of the gradient product, lane (i, q) holds Yz[16 b + 4 q + v][i].  Rows k >= K are zero.
__global__ void __launch_bounds__(256)
frs_build_tables_kernel(int K, int nblk, const float* __restrict__ zsamples /*[K,3]*/, float* __restrict__ tables)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nblk * 512) return;
    const int b = idx >> 9, slot = (idx >> 6) & 7, lane = idx & 63;
    const int lo = lane & 15, q = lane >> 4;
    int k, i;
    if (slot < 4) { k = 16 * b + lo; i = 4 * slot + q; }
    else { k = 16 * b + 4 * q + (slot - 4); i = lo; }
    float val = 0.f;
    if (k < K) {
        float Y[16];
        sh_basis16(zsamples[3 * k], zsamples[3 * k + 1], zsamples[3 * k + 2], 16, Y);
        val = Y[0];
#pragma unroll
        for (int j = 1; j < 16; j++) val = i == j ? Y[j] : val;
    }
    tables[idx] = val;
}

// ---- coefficient rotation (thread per Gaussian, the row in registers) -----------------------------------------------------------
// band l of the function is sampled at the rotated points R p_j and re-expanded: c'_l = A_l^{-1} [sum_i c_{l,i} Y_{l,i}(R p_j)]_j
template <int A0, int N, bool BACK>
__device__ __forceinline__ void frs_rotate_band(const float (&R)[9], const float (*pts)[3], const float (*ainv)[N], float* row)
{
    float f[N][3];
    if (!BACK) {
        // f_j = band-l part of the function at R p_j
#pragma unroll
        for (int j = 0; j < N; j++) {
            const float px = pts[j][0], py = pts[j][1], pz = pts[j][2];
            float Y[16];
            sh_basis16(R[0] * px + R[1] * py + R[2] * pz, R[3] * px + R[4] * py + R[5] * pz, R[6] * px + R[7] * py + R[8] * pz,
                       16, Y);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < N; i++) acc += Y[A0 + i] * row[(A0 + i) * 3 + c];
                f[j][c] = acc;
            }
        }
#pragma unroll
        for (int i = 0; i < N; i++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < N; j++) acc += ainv[i][j] * f[j][c];
                row[(A0 + i) * 3 + c] = acc;
            }
    } else {
        // transpose of the map above: t = A^{-T} dc', dc_i = sum_j Y_{l,i}(R p_j) t_j
#pragma unroll
        for (int j = 0; j < N; j++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < N; i++) acc += ainv[i][j] * row[(A0 + i) * 3 + c];
                f[j][c] = acc;
            }
        float out[N][3];
#pragma unroll
        for (int i = 0; i < N; i++) out[i][0] = out[i][1] = out[i][2] = 0.f;
#pragma unroll
        for (int j = 0; j < N; j++) {
            const float px = pts[j][0], py = pts[j][1], pz = pts[j][2];
            float Y[16];
            sh_basis16(R[0] * px + R[1] * py + R[2] * pz, R[3] * px + R[4] * py + R[5] * pz, R[6] * px + R[7] * py + R[8] * pz,
                       16, Y);
#pragma unroll
            for (int i = 0; i < N; i++)
#pragma unroll
                for (int c = 0; c < 3; c++) out[i][c] += Y[A0 + i] * f[j][c];
        }
#pragma unroll
        for (int i = 0; i < N; i++)
#pragma unroll
            for (int c = 0; c < 3; c++) row[(A0 + i) * 3 + c] = out[i][c];
    }
}

// BACK == false: src = incidents [P,16,3] -> dst = rotated coefficients c' [P,16,3]
// BACK == true : src = dL/dc' [P,16,3]    -> dst = dL/d incidents [P,16,3]  (every row is written; rows of Gaussians that are
//                not on the rotated path hold garbage and are overwritten by the general kernel afterwards)
template <bool BACK>
__global__ void __launch_bounds__(256)
frs_rotate_kernel(int P, const float* __restrict__ ray_normals, const float* __restrict__ src, float* __restrict__ dst,
                  const uint8_t* __restrict__ only_valid /* optional: rows with only_valid[g] == 0 are left untouched */)
{
    // Thread per Gaussian, its 192-byte row in registers: twelve 16-byte loads in flight per lane, twelve 16-byte stores.  (The
    // first version staged 256 rows through 50 KB of LDS for coalescing: 3 workgroups per CU, two barriers, 44 us for 115 MB.
    // A wave's 16-byte accesses to 64 different rows cost the texture-address unit 64 cycles each -- 12 us for the whole array --
    // and the lines are used completely by the other 11 accesses of the same lanes.)
    const int g = blockIdx.x * 256 + (int)threadIdx.x;
    if (g >= P) return;
    if (only_valid != nullptr && only_valid[g] == 0) return;
    float row[48];
    const float4* s4 = reinterpret_cast<const float4*>(src + (size_t)g * 48);
#pragma unroll
    for (int q = 0; q < 12; q++) {
        const float4 v = s4[q];
        row[4 * q] = v.x; row[4 * q + 1] = v.y; row[4 * q + 2] = v.z; row[4 * q + 3] = v.w;
    }
    float R[9];
    frs_rotation(ray_normals[3 * (size_t)g], ray_normals[3 * (size_t)g + 1], ray_normals[3 * (size_t)g + 2], R);
    // band 0 (the constant) is rotation invariant
    frs_rotate_band<1, 3, BACK>(R, kShRotPoints1, kShRotAinv1, row);
    frs_rotate_band<4, 5, BACK>(R, kShRotPoints2, kShRotAinv2, row);
    frs_rotate_band<9, 7, BACK>(R, kShRotPoints3, kShRotAinv3, row);
    float4* d4 = reinterpret_cast<float4*>(dst + (size_t)g * 48);
#pragma unroll
    for (int q = 0; q < 12; q++) d4[q] = make_float4(row[4 * q], row[4 * q + 1], row[4 * q + 2], row[4 * q + 3]);
}

// ---- incident-light chain (round 5): rotation back + Adam + rotation forward in ONE pass ---------------------------------------
// A whole single-GPU iteration ends, for the incident-light coefficients, with three passes over [P,16,3] rows in a row:
//   frs_rotate_kernel<true>   dL/dc' -> dL/d incidents            (read 192, write 192 bytes per Gaussian)
//   adam_kernel               p, g, m, v -> p, m, v               (read 768, write 576)
//   frs_rotate_kernel<false>  incidents -> c' for the NEXT forward (read 192, write 192)
// = 2112 bytes per Gaussian and three launches on the chain that decides when the next shading forward can start (48 + 94 + 36 us
// at 300k Gaussians beside the other groups' Adam: the critical path of the iteration, profiles/r05_*_sequence.txt).  One kernel:
// 1728 bytes per Gaussian, no launch boundaries.
// FIRST VERSION (thread per Gaussian, the row in registers, p / m / v walked row-wise by every lane): 300 us -- eight 192-byte row
// streams per lane with 192-byte strides between the lanes of every load: the address unit, not HBM, was the limit.  THIS VERSION
// keeps the thread-per-Gaussian ROTATIONS (they need a whole row per lane) but makes every global access of the wave a contiguous
// run: a wave owns 64 consecutive Gaussians = 12 KB of each array; rows travel through 12.25 KB of wave-private LDS (row stride 49
// words: a lane walking its row and the wave walking a float4 column are both conflict-free), the Adam update runs on the
// COALESCED layout (lane = float4 index inside the wave's run, exactly adam_kernel's access pattern).  No workgroup barrier: the
// four waves of a workgroup only share the LDS allocation.
// The Adam arithmetic is adam_kernel's, statement for statement -- but this translation unit is built with -ffast-math (neither
// `#pragma float_control` nor __fdiv_rn / __fsqrt_rn switch that off on this target: checked in the ISA), so its division and
// square root are the 1-ulp v_rcp_f32 / v_sqrt_f32: the UPDATE term (lr x O(1)) differs from adam_kernel's by a few ulp of
// itself, i.e. by ~1e-7 x lr on the parameter -- far below the run-to-run differences the float atomics of the backward already
// cause.  Single-GPU whole iterations only (a data-parallel run applies the group's update with adam_kernel: replicas must
// stay bit-identical).  The rotations are the very functions the two kernels above call.
// Gaussians off the rotated path (valid[g] == 0): their gradient row was written by the listed kernel in the world frame already
// (read here instead of rotated), their c' row is not used by anybody (written anyway: same arithmetic as everywhere else).
struct FrsAdam {
    float lr, lr_tail, beta1, beta2, eps, bias1, inv_sqrt_bias2, grad_scale;
    float omb1, omb2;          // 1 - beta1, 1 - beta2, formed in fp32 on the host: under -ffast-math the compiler rewrites
                               // (1 - b) * x as x - b * x, whose cancellation costs 6e-5 of the (1 - beta2) g^2 term
};

__device__ __forceinline__ void frs_adam_update(const FrsAdam& a, float lr, float& p, float g, float& m, float& v)
{
    g *= a.grad_scale;
    m = m + (g - m) * a.omb1;
    v = a.beta2 * v + a.omb2 * g * g;
    const float denom = sqrtf(v) * a.inv_sqrt_bias2 + a.eps;
    p -= (lr / a.bias1) * (m / denom);
}

constexpr int FRS_CHAIN_LD = 49;           // LDS row stride in words

__device__ __forceinline__ float4 frs_nt_load4(const float4* q)
{
    return make_float4(__builtin_nontemporal_load(&q->x), __builtin_nontemporal_load(&q->y), __builtin_nontemporal_load(&q->z),
                       __builtin_nontemporal_load(&q->w));
}
__device__ __forceinline__ void frs_nt_store4(float4* q, const float4& v)
{
    __builtin_nontemporal_store(v.x, &q->x);
    __builtin_nontemporal_store(v.y, &q->y);
    __builtin_nontemporal_store(v.z, &q->z);
    __builtin_nontemporal_store(v.w, &q->w);
}

__global__ void __launch_bounds__(256)
frs_incident_chain_kernel(int P, const float* __restrict__ ray_normals, const uint8_t* __restrict__ valid,
                          const float* __restrict__ dcprime, float* __restrict__ dL_dincidents, float* __restrict__ incidents,
                          float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq, float* __restrict__ cprime, FrsAdam adam,
                          const float* __restrict__ skip_flag, int listed_in_dcprime)
{
    // a frame the bounded forward dropped: no update (adam_kernel's rule) -- and then nothing here is needed: the parameters and
    // therefore c' are unchanged, the gradient of a dropped frame is nobody's input
    if (skip_flag != nullptr && *skip_flag != 0.0f) return;
    extern __shared__ float s_rows[];             // 4 x 64 x FRS_CHAIN_LD floats (+ padding that limits the workgroups per CU)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* s = s_rows + wave * 64 * FRS_CHAIN_LD;
    const int g0 = (blockIdx.x * 4 + wave) * 64;                  // this wave's 64 consecutive Gaussians
    if (g0 >= P) return;
    const int ng = min(64, P - g0);
    const int n4 = ng * 12;                                       // float4s in the wave's run of every [P,48] array
    const int g = g0 + lane;
    const bool live = lane < ng;
    const size_t run = (size_t)g0 * 48;
    // position of float4 number e4 of the run inside the LDS rows (a float4 never straddles a row: 48 = 12 x 4)
    auto lds_of = [&](int e4) -> float* { return s + (e4 / 12) * FRS_CHAIN_LD + 4 * (e4 % 12); };
    // ---- the coefficient gradient in the rotated frame, coalesced, into the rows
    {
        const float4* src = reinterpret_cast<const float4*>(dcprime + run);
        float4 t[12];
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const int e4 = i * 64 + lane;
            t[i] = src[e4 < n4 ? e4 : 0];
        }
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const int e4 = i * 64 + lane;
            if (e4 < n4) {
                float* d = lds_of(e4);
                d[0] = t[i].x; d[1] = t[i].y; d[2] = t[i].z; d[3] = t[i].w;
            }
        }
    }
    float R[9];
    const int gc = live ? g : g0;
    frs_rotation(ray_normals[3 * (size_t)gc], ray_normals[3 * (size_t)gc + 1], ray_normals[3 * (size_t)gc + 2], R);
    const bool rotated = valid == nullptr || valid[gc] != 0;
    __builtin_amdgcn_wave_barrier();
    // ---- rotation back, thread per Gaussian; the world-frame gradient row replaces the rotated one in the LDS row
    if (live) {
        float row[48];
        float* mine = s + lane * FRS_CHAIN_LD;
        if (rotated) {
#pragma unroll
            for (int c = 0; c < 48; c++) row[c] = mine[c];
            frs_rotate_band<1, 3, true>(R, kShRotPoints1, kShRotAinv1, row);
            frs_rotate_band<4, 5, true>(R, kShRotPoints2, kShRotAinv2, row);
            frs_rotate_band<9, 7, true>(R, kShRotPoints3, kShRotAinv3, row);
        } else if (listed_in_dcprime) {
            // (data parallel, round 6: the listed kernel wrote its world-frame row into the Gaussian's row of `dcprime` -- ONE buffer
            // is all-reduced -- and that row sits in the LDS row already)
#pragma unroll
            for (int c = 0; c < 48; c++) row[c] = mine[c];
        } else {
            // (a few hundred Gaussians per launch: the listed kernel's world-frame row, read in place)
            const float4* s4 = reinterpret_cast<const float4*>(dL_dincidents + (size_t)g * 48);
#pragma unroll
            for (int q = 0; q < 12; q++) {
                const float4 v = s4[q];
                row[4 * q] = v.x; row[4 * q + 1] = v.y; row[4 * q + 2] = v.z; row[4 * q + 3] = v.w;
            }
        }
#pragma unroll
        for (int c = 0; c < 48; c++) mine[c] = row[c];
    }
    __builtin_amdgcn_wave_barrier();
    // ---- Adam on the coalesced layout (adam_kernel's access pattern); the gradient row goes out, the NEW parameters take its
    // place in the LDS rows.  Columns 0..2 of a row are the dc coefficient (learning rate lr), the other 45 the rest (lr_tail)
    {
        float4* p4 = reinterpret_cast<float4*>(incidents + run);
        float4* m4 = reinterpret_cast<float4*>(exp_avg + run);
        float4* v4 = reinterpret_cast<float4*>(exp_avg_sq + run);
        float4* g4 = reinterpret_cast<float4*>(dL_dincidents + run);
#pragma unroll 4
        for (int i = 0; i < 12; i++) {
            const int e4 = i * 64 + lane;
            if (e4 < n4) {
                // (the moments are touched once per iteration: nontemporal, as in adam_kernel)
                float4 p = p4[e4], m = frs_nt_load4(m4 + e4), v = frs_nt_load4(v4 + e4);
                float* d = lds_of(e4);
                const float4 gr = make_float4(d[0], d[1], d[2], d[3]);
                const bool dc = (e4 % 12) == 0;                          // this float4 holds columns 0..3 of its row
                frs_adam_update(adam, dc ? adam.lr : adam.lr_tail, p.x, gr.x, m.x, v.x);
                frs_adam_update(adam, dc ? adam.lr : adam.lr_tail, p.y, gr.y, m.y, v.y);
                frs_adam_update(adam, dc ? adam.lr : adam.lr_tail, p.z, gr.z, m.z, v.z);
                frs_adam_update(adam, adam.lr_tail, p.w, gr.w, m.w, v.w);
                p4[e4] = p;
                frs_nt_store4(m4 + e4, m);
                frs_nt_store4(v4 + e4, v);
                g4[e4] = gr;
                d[0] = p.x; d[1] = p.y; d[2] = p.z; d[3] = p.w;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- rotation forward of the new coefficients, thread per Gaussian
    if (live) {
        float row[48];
        float* mine = s + lane * FRS_CHAIN_LD;
#pragma unroll
        for (int c = 0; c < 48; c++) row[c] = mine[c];
        frs_rotate_band<1, 3, false>(R, kShRotPoints1, kShRotAinv1, row);
        frs_rotate_band<4, 5, false>(R, kShRotPoints2, kShRotAinv2, row);
        frs_rotate_band<9, 7, false>(R, kShRotPoints3, kShRotAinv3, row);
#pragma unroll
        for (int c = 0; c < 48; c++) mine[c] = row[c];
    }
    __builtin_amdgcn_wave_barrier();
    {
        float4* dst = reinterpret_cast<float4*>(cprime + run);
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const int e4 = i * 64 + lane;
            if (e4 < n4) {
                const float* d = lds_of(e4);
                dst[e4] = make_float4(d[0], d[1], d[2], d[3]);
            }
        }
    }
}

// texture [ntexel,3] -> LDS as float4 texels; two texels (six loads) in flight per thread -- as a plain loop the loads of one
// texel are waited for before the next texel's are issued
__device__ __forceinline__ void frs_stage_texture(const float* __restrict__ env, int ntexel, float4* s_env4)
{
    for (int i0 = threadIdx.x; i0 < ntexel; i0 += 2 * (int)blockDim.x) {
        const int i1 = i0 + (int)blockDim.x, j1 = i1 < ntexel ? i1 : i0;
        const float a0 = env[3 * i0], a1 = env[3 * i0 + 1], a2 = env[3 * i0 + 2];
        const float b0 = env[3 * j1], b1 = env[3 * j1 + 1], b2 = env[3 * j1 + 2];
        s_env4[i0] = make_float4(a0, a1, a2, 0.f);
        if (i1 < ntexel) s_env4[i1] = make_float4(b0, b1, b2, 0.f);
    }
}

// ---- the 8-byte lookup record of a sample -------------------------------------------------------------------------------------
// The lat-long lookup of a fixed direction is a constant between visibility updates (direct_light_map.py:70-83: texel corner +
// two bilinear weights).  Round 3 cached it as 12 bytes (corner, fp32 weight, fp32 weight) next to the 12-byte direction; the
// direction is gone (see the kernels) and the record is two dwords:
//     dword 0 = (x0 + 1) << 23 | mantissa of (1 + wx1)        dword 1 = (y0 + 1) << 23 | mantissa of (1 + wy1)
// A weight w in [0, 1) is stored as the 23 mantissa bits of 1 + w -- fixed point with 2^-23 steps, the resolution fp32 itself has
// on [0.5, 1) -- and comes back as (0x3f800000 | bits) - 1 in two instructions; x0 + 1 in [0, We], y0 + 1 in [0, He] take the 9
// bits above (He, We <= 511: far beyond the textures that fit LDS).
__device__ __forceinline__ uint32_t frs_pack_axis(int corner /* >= -1 */, float w1 /* [0, 1) */)
{
    const float f = fminf(1.0f + w1, __uint_as_float(0x3fffffffu));                 // (1 + w rounds to 2.0 for w > 1 - 2^-24)
    return ((uint32_t)(corner + 1) << 23) | (__float_as_uint(f) & 0x7fffffu);
}
__device__ __forceinline__ PackedTap frs_unpack_tap(uint32_t lo, uint32_t hi)
{
    PackedTap t;
    t.xy = (lo >> 23) | ((hi >> 23) << 16);
    t.wx1 = __uint_as_float(0x3f800000u | (lo & 0x7fffffu)) - 1.0f;
    t.wy1 = __uint_as_float(0x3f800000u | (hi & 0x7fffffu)) - 1.0f;
    return t;
}

// the records of all P x K samples from the ray normals alone: d_k = normalize(R(n) z_k) evaluated as sampling.py /
// graphics_utils.py:9-37 evaluate it (matrix product, then x / max(|x|, 1e-12)), then the lookup of make_tap
__global__ void __launch_bounds__(256)
frs_build_taps_kernel(int P, int K, const float* __restrict__ ray_normals, const float* __restrict__ zsamples /*[K,3]*/, int He,
                      int We, uint32_t* __restrict__ taps /*[P,K,2]*/)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)P * K) return;
    const int g = (int)(i / (size_t)K), k = (int)(i - (size_t)g * K);
    float R[9];
    frs_rotation(ray_normals[3 * (size_t)g], ray_normals[3 * (size_t)g + 1], ray_normals[3 * (size_t)g + 2], R);
    const float zx = zsamples[3 * k], zy = zsamples[3 * k + 1], zz = zsamples[3 * k + 2];
    float dx = R[0] * zx + R[1] * zy + R[2] * zz, dy = R[3] * zx + R[4] * zy + R[5] * zz, dz = R[6] * zx + R[7] * zy + R[8] * zz;
    const float len = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-12f);
    dx /= len; dy /= len; dz /= len;
    const PackedTap t = make_tap(dx, dy, dz, nullptr, He, We);
    taps[2 * i] = frs_pack_axis((int)(t.xy & 0xffffu) - 1, t.wx1);
    taps[2 * i + 1] = frs_pack_axis((int)(t.xy >> 16) - 1, t.wy1);
}

// ---- per-lane sample block: 4 consecutive samples of one Gaussian: 16 bytes of visibility + 32 bytes of lookup records ---------
struct FrsBlock {
    float4 vis;
    uint4 t0, t1;           // 4 records (8 dwords)
};

__device__ __forceinline__ FrsBlock frs_load_block(size_t row /* g * K */, int k0, int K, const float* __restrict__ visibility,
                                                   const uint32_t* __restrict__ taps)
{
    FrsBlock b;
    const int kc = k0 + 4 <= K ? k0 : K - 4;                         // (K % 4 == 0, K >= 4): clamped loads stay inside the row
    const size_t o = row + (size_t)kc;
    b.vis = *reinterpret_cast<const float4*>(visibility + o);
    const uint4* tp = reinterpret_cast<const uint4*>(taps + 2 * o);
    b.t0 = tp[0]; b.t1 = tp[1];
    return b;
}

__device__ __forceinline__ void frs_sample_of(const FrsBlock& b, int v, float& vis, PackedTap& t)
{
    const uint32_t u[8] = {b.t0.x, b.t0.y, b.t0.z, b.t0.w, b.t1.x, b.t1.y, b.t1.z, b.t1.w};
    const float vs[4] = {b.vis.x, b.vis.y, b.vis.z, b.vis.w};
    vis = vs[v];
    t = frs_unpack_tap(u[2 * v], u[2 * v + 1]);
}

__device__ __forceinline__ float frs_sum4(float x)       // sum over the 4 lanes of a Gaussian: lanes l, l^16, l^32, l^48
{
    x += __shfl_xor(x, 16, 64);
    x += __shfl_xor(x, 32, 64);
    return x;
}

// ---- the NEXT group's per-Gaussian data, staged per wave by LDS-DMA -----------------------------------------------------------
// A group of 16 Gaussians is only K / 16 sample blocks long (4 at K = 64), and its per-Gaussian data -- the material record, the
// ray normal and the lane's 12 rotated coefficients -- sit in front of every sample of it: loaded at the top of the group, one
// memory latency per group is exposed on a dependent chain.  Instead the wave issues them for the NEXT group during the LAST sample
// block of the current one, straight into LDS (global_load_lds, no registers: the backward has none to spare), together with the
// next group's first sample block (into the `nxt` registers the block loop already owns): the top of a group is one
// s_waitcnt + a handful of LDS reads.  Issued there and not earlier on purpose: vmcnt completes in order and the compiler,
// which does not see the DMA, waits with vmcnt(0) for its own sample-block loads at the top of every block -- anything issued
// before that wait has to land by then.
constexpr int FRS_ST_CP = 0, FRS_ST_TAPS = 768, FRS_ST_VIS = 1280, FRS_ST_BASE = 1536, FRS_ST_NRM = 1600, FRS_ST_VIEW = 1664,
              FRS_ST_RGH = 1728, FRS_ST_RNRM = 1792, FRS_ST_GP = 1856, FRS_ST_GD = 1920;
constexpr int FRS_ST_FWD = 1856, FRS_ST_BWD = 1984;          // floats per wave (7.25 / 7.75 KB)

struct FrsSrc {              // the per-Gaussian and per-sample arrays of one call
    const float *base_color, *roughness, *normals, *viewdirs, *ray_normals, *cprime, *g_pbr, *g_diff, *visibility;
    const uint32_t* taps;
};

template <bool BWD>
__device__ __forceinline__ void frs_stage_group(unsigned int sb /* LDS byte address of the wave's area, SGPR */, int grp, int P,
                                                int K, int lane, const FrsSrc& p)
{
    const int g0 = grp * FRS_G;
    // the group's FIRST sample block (16 samples of 16 rows): 32 lookup dwords and 16 visibilities per row, same [j][row][16] image
    // as the coefficients.  K < 16: the chunks past the row's end belong to the next row (masked by k < K in the kernels); the very
    // last rows of the arrays are clamped to stay inside them
    {
        const size_t total = (size_t)P * (size_t)K;
        const size_t r0 = (size_t)min(g0 + (lane >> 2), P - 1) * (size_t)K;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const size_t o = min(2 * r0 + 16 * j + 4 * (lane & 3), 2 * total - 4);
            lds_dma_at<16>(reinterpret_cast<const float*>(p.taps) + o, sb + 4u * (FRS_ST_TAPS + 256 * j));
        }
        lds_dma_at<16>(p.visibility + min(r0 + 4 * (lane & 3), total - 4), sb + 4u * FRS_ST_VIS);
    }
    // 16 rows of 48 coefficients: DMA j moves floats [16 j, 16 j + 16) of every row, lane = (row, 16-byte chunk), so the LDS image
    // is [j][row][16 floats]: coefficient float f of row t sits at (f >> 4) * 256 + t * 16 + (f & 15)
    const float* cp = p.cprime + (size_t)min(g0 + (lane >> 2), P - 1) * 48 + 4 * (lane & 3);
#pragma unroll
    for (int j = 0; j < 3; j++) lds_dma_at<16>(cp + 16 * j, sb + 4u * (FRS_ST_CP + 256 * j));
    const int l3 = lane < 48 ? lane : 47, t3 = l3 / 3, c3 = l3 - 3 * t3;
    const size_t o3 = 3 * (size_t)min(g0 + t3, P - 1) + c3;
    lds_dma_at<4>(p.base_color + o3, sb + 4u * FRS_ST_BASE);
    lds_dma_at<4>(p.normals + o3, sb + 4u * FRS_ST_NRM);
    lds_dma_at<4>(p.viewdirs + o3, sb + 4u * FRS_ST_VIEW);
    lds_dma_at<4>(p.ray_normals + o3, sb + 4u * FRS_ST_RNRM);
    lds_dma_at<4>(p.roughness + min(g0 + (lane & 15), P - 1), sb + 4u * FRS_ST_RGH);
    if (BWD) {
        lds_dma_at<4>(p.g_pbr + o3, sb + 4u * FRS_ST_GP);
        lds_dma_at<4>(p.g_diff + o3, sb + 4u * FRS_ST_GD);
    }
}

// the staged first block as the register image frs_load_block gives (lane (gl, q): samples 4 q .. 4 q + 3 = dwords 8 q .. of row gl)
__device__ __forceinline__ FrsBlock frs_staged_block(const float* st, int gl, int q, int K)
{
    FrsBlock b;
    const int kq = 4 * q + 4 <= K ? q : (K >> 2) - 1;                 // K < 16: the same clamp as frs_load_block
    uint4 t[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int c = 2 * kq + i, o = (c >> 2) * 256 + gl * 16 + (c & 3) * 4;
        t[i] = *reinterpret_cast<const uint4*>(st + FRS_ST_TAPS + o);
    }
    b.t0 = t[0]; b.t1 = t[1];
    b.vis = *reinterpret_cast<const float4*>(st + FRS_ST_VIS + gl * 16 + 4 * kq);
    return b;
}

// ---- a Gaussian in its ray frame ------------------------------------------------------------------------------------------------
// The sample directions are d_k = R z_k / |R z_k| with ONE z set for everybody, so every dot product of d_k with a per-Gaussian
// vector w is (R^T w) . z_k -- a [K x 3] x [3 x 16 Gaussians] product against a constant matrix, and the three direction rows of
// the Y_i(z_k) table (Y_1 = -C1 y, Y_2 = C1 z, Y_3 = -C1 x: slot 0 of a block) already ARE that matrix: one more
// v_mfma_f32_16x16x4_f32 per vector and block against the B operand (0, -w'_y / C1, w'_z / C1, -w'_x / C1)[q].  Two vectors are
// needed: the shading normal n (n . d_k; N . L is the same number times sign / |n|) and the unit view vector V (L . V).  The
// half vector never appears: with unit L and V, |(L + V) / 2|^2 = (1 + L.V) / 2, N.H = (N.L + N.V) / (2 |u|), V.H = |u|.
// R(n) is evaluated in fp32 and is orthonormal only to FRS_MAX_DEFECT on this path; that matters for ONE thing, the length of
// R z_k (a 1e-5 error of N.H moves the GGX lobe of a smooth Gaussian by per cents): 1 / |R z_k| = 1 - q_k / 2 to 1e-9 with
// q_k = z_k^T (R^T R - I) z_k, a quadratic form in z_k and therefore a combination of Y_0 and Y_4..8 -- two more products
// (slots 1 and 2) with the constant term as the accumulator's start value.
struct FrsFrame {
    float bn, bv;            // B operands of the two dot products (this lane's q)
    float bq1, bq2, q0;      // B operands / start value of the length correction
};

__device__ __forceinline__ void frs_frame(FrsFrame& f, const float (&R)[9], const float nx, const float ny, const float nz,
                                          const float Vx, const float Vy, const float Vz, int q)
{
    constexpr float C1 = 0.4886025119029199f, K4 = 1.0925484305920792f, K6 = 0.31539156525252005f, K8 = 0.5462742152960396f;
    constexpr float C0 = 0.28209479177387814f;
    (void)C0;
    // R^T w: columns of R
    const float npx = R[0] * nx + R[3] * ny + R[6] * nz, npy = R[1] * nx + R[4] * ny + R[7] * nz,
                npz = R[2] * nx + R[5] * ny + R[8] * nz;
    const float vpx = R[0] * Vx + R[3] * Vy + R[6] * Vz, vpy = R[1] * Vx + R[4] * Vy + R[7] * Vz,
                vpz = R[2] * Vx + R[5] * Vy + R[8] * Vz;
    f.bn = q == 1 ? -npy / C1 : (q == 2 ? npz / C1 : (q == 3 ? -npx / C1 : 0.f));
    f.bv = q == 1 ? -vpy / C1 : (q == 2 ? vpz / C1 : (q == 3 ? -vpx / C1 : 0.f));
    // E = R^T R - I
    const float e00 = R[0] * R[0] + R[3] * R[3] + R[6] * R[6] - 1.f, e11 = R[1] * R[1] + R[4] * R[4] + R[7] * R[7] - 1.f,
                e22 = R[2] * R[2] + R[5] * R[5] + R[8] * R[8] - 1.f;
    const float e01 = R[0] * R[1] + R[3] * R[4] + R[6] * R[7], e02 = R[0] * R[2] + R[3] * R[5] + R[6] * R[8],
                e12 = R[1] * R[2] + R[4] * R[5] + R[7] * R[8];
    // z^T E z = tr E / 3 + [Y4 2 e01 - Y5 2 e12 - Y7 2 e02] / K4 + Y6 (2 e22 - e00 - e11) / (6 K6) + Y8 (e00 - e11) / (2 K8)
    f.bq1 = q == 0 ? 2.f * e01 / K4 : (q == 1 ? -2.f * e12 / K4 : (q == 2 ? (2.f * e22 - e00 - e11) / (6.f * K6) : -2.f * e02 / K4));
    f.bq2 = q == 0 ? (e00 - e11) / (2.f * K8) : 0.f;
    f.q0 = (e00 + e11 + e22) / 3.f;
}

// the per-sample geometry of a block: n . d, L . V for the lane's four samples from the block's first three table words
struct FrsGeom {
    f32x4 dn, lov, sk;       // n . d_k, L_k . V, 1 / |R z_k|
};
__device__ __forceinline__ FrsGeom frs_block_geometry(const FrsFrame& f, float a0, float a1, float a2)
{
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    FrsGeom o;
    o.dn = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, f.bn, zero, 0, 0, 0);
    o.lov = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, f.bv, zero, 0, 0, 0);
    f32x4 qf = {f.q0, f.q0, f.q0, f.q0};
    qf = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, f.bq1, qf, 0, 0, 0);
    qf = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, f.bq2, qf, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < 4; v++) {
        o.sk[v] = 1.f - 0.5f * qf[v];
        o.dn[v] *= o.sk[v];
        o.lov[v] *= o.sk[v];
    }
    return o;
}

// =====================================================================================================================
// Forward (training outputs: pbr, diffuse_light, mean visibility -- columns 0..5 and 18 of the 19; neilf.py:120-122)
// =====================================================================================================================
__global__ void __launch_bounds__(64 * FRS_WAVES, 3)
shade_forward_frs_kernel(int P, int K, FrsSrc src, const float* __restrict__ env /* [He*We][3] */, int He, int We,
                         float uniform_area, const float* __restrict__ tables, const uint8_t* __restrict__ valid,
                         float* __restrict__ out, float* __restrict__ feat /* NULL, or the [P,16] feature rows */)
{
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    float4* s_env4 = reinterpret_cast<float4*>(s_mem);
    frs_stage_texture(env, He * We, s_env4);              // (a tap is one ds_read_b128: texels padded to float4 here)
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gl = lane & 15, q = lane >> 4;
    const int nblk = (K + 15) >> 4;
    const float invK = 1.0f / (float)K;
    const int ngroups = (P + FRS_G - 1) / FRS_G;
    float* st = s_mem + 4 * He * We + wave * FRS_ST_FWD;                 // this wave's staging area
    const unsigned int st_addr = lds_address_of(st);
    // Static split of the groups over the persistent waves.  (A device-wide queue -- one atomicAdd per wave and group -- was
    // measured: 0.283 instead of 0.154 ms.  22 000 atomics on one address retire ~35 ns apart, and each one, being older than the
    // next block's table loads, is waited for in order.  The static split's tail -- 18750 groups over 3072 waves: 318 waves run a
    // seventh group -- is cheaper than it looks: those waves then have their SIMD to themselves.)
    const int gstride = gridDim.x * FRS_WAVES;
    int grp = blockIdx.x * FRS_WAVES + wave;
    uint8_t nvalid = 0;
    if (grp < ngroups) {
        nvalid = valid[min(grp * FRS_G + gl, P - 1)];
        frs_stage_group<false>(st_addr, grp, P, K, lane, src);
    }
    for (; grp < ngroups; grp += gstride) {
        const int g = grp * FRS_G + gl;
        const int gc = min(g, P - 1);
        const bool live_g = g < P && nvalid != 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this group's staged data have landed
        // per-Gaussian record
        float u[64];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            u[48 + c] = st[FRS_ST_BASE + 3 * gl + c];
            u[52 + c] = st[FRS_ST_NRM + 3 * gl + c];
            u[55 + c] = st[FRS_ST_VIEW + 3 * gl + c];
        }
        u[51] = st[FRS_ST_RGH + gl];
        const float rn0 = st[FRS_ST_RNRM + 3 * gl], rn1 = st[FRS_ST_RNRM + 3 * gl + 1], rn2 = st[FRS_ST_RNRM + 3 * gl + 2];
        // B operand of the local-light product: rotated coefficients 4 s + q of the lane's Gaussian
        float bc[4][3];
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int f = (4 * s + q) * 3 + c;
                bc[s][c] = st[FRS_ST_CP + (f >> 4) * 256 + gl * 16 + (f & 15)];
            }
        FrsBlock nxt = frs_staged_block(st, gl, q, K);
        GaussFwd G;
        gauss_setup(G, u);
        FrsFrame F;
        {
            float R[9];
            frs_rotation(rn0, rn1, rn2, R);
            frs_frame(F, R, G.n[0], G.n[1], G.n[2], G.V[0], G.V[1], G.V[2], q);
        }
        const float fd[3] = {G.base[0] / kPi, G.base[1] / kPi, G.base[2] / kPi};
        const float nom1 = G.NoV * (1.f - G.kk) + G.kk;
        float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const size_t row = (size_t)gc * (size_t)K;
        int ngrp_l = min(grp + gstride, ngroups - 1);                    // (past the end: a harmless reload of the last group)
        int ngc_l = min(ngrp_l * FRS_G + gl, P - 1);
        for (int b = 0; b < nblk; b++) {
            const FrsBlock cur = nxt;
            // Issue order matters (vmcnt completes in order): the block's four table words FIRST, then the prefetch of the next
            // sample block -- the wait for the table words in front of the matrix product then leaves the prefetch in flight
            // (the other way round it drained it: the "prefetch" was waited for ~100 instructions after its issue).
            const float* tb = tables + (size_t)b * 512 + lane;
            float a[4];
#pragma unroll
            for (int s = 0; s < 4; s++) a[s] = tb[64 * s];
            const bool last = b + 1 == nblk;
            if (!last) {
                nxt = frs_load_block(row, 16 * (b + 1) + 4 * q, K, src.visibility, src.taps);
            } else {
                // (the empty asm keeps the compiler from hoisting the next group's address computations out of the block loop,
                // where they would be live -- 2 VGPRs each -- through every block of the group)
                asm volatile("" : "+v"(ngrp_l), "+v"(ngc_l));
                nvalid = valid[ngc_l];
            }
            __builtin_amdgcn_sched_barrier(0);       // (the prefetch is issued HERE, not sunk into the block to save registers)
            // local light of the 16 samples x 16 Gaussians of this block: l[c] = sum_i Yz[k][i] c'[i][c]
            f32x4 l[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int c = 0; c < 3; c++) l[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], bc[s][c], l[c], 0, 0, 0);
            const FrsGeom geo = frs_block_geometry(F, a[0], a[1], a[2]);
            // the next group's per-Gaussian data: behind the last compiler-visible wait of the group (see frs_stage_group)
            if (last) frs_stage_group<false>(st_addr, ngrp_l, P, K, lane, src);
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int k = 16 * b + 4 * q + v;
                const bool ok = live_g && k < K;
                float vis;
                PackedTap t;
                frs_sample_of(cur, v, vis, t);
                float e[3], w4[4];
                int tex[4];
                env_fetch(t, s_env4, He, We, e, tex, w4);
                const float lv = ok ? 1.f : 0.f;
                const float dn = geo.dn[v], lov = geo.lov[v];
                const float area_ndi = uniform_area * fmaxf(dn, 0.f) * lv;
                const float rawNoL = dn * G.nscale;
                const float uu = fmaxf(0.5f * lov + 0.5f, 1e-24f);                   // |(L + V) / 2|^2
                const float uinv = __builtin_amdgcn_rsqf(uu);
                const float NoL = fminf(fmaxf(rawNoL, 1e-6f), 1.f);
                const float NoH = fminf(fmaxf((rawNoL + G.rawNoV) * (0.5f * uinv), 1e-6f), 1.f);
                const float VoH = fminf(fmaxf(uu * uinv, 1e-6f), 1.f);
                const float p2 = exp2f((-5.55473f * VoH - 6.98316f) * VoH);
                const float frac = (0.04f + 0.96f * p2) * G.a2;
                const float nom0 = NoH * NoH * (G.a2 - 1.f) + 1.f;
                const float nom2 = NoL * (1.f - G.kk) + G.kk;
                const float nom = fminf(fmaxf(4.f * kPi * nom0 * nom0 * nom1 * nom2, 1e-6f), 4.f * kPi);
                const float spec = frac / nom;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float lin = fmaxf(l[c][v], 0.f) + e[c] * vis;
                    const float transport = lin * area_ndi;
                    acc[c] += (fd[c] + spec) * transport;      // pbr
                    acc[3 + c] += transport;                   // diffuse_light
                }
                acc[6] += vis * lv;
            }
        }
#pragma unroll
        for (int i = 0; i < 7; i++) acc[i] = frs_sum4(acc[i]) * invK;
        if (live_g && q == 0) {
            float* o = out + (size_t)g * SHADE_NOUT;
#pragma unroll
            for (int i = 0; i < 6; i++) o[i] = acc[i];
            o[18] = acc[6];
            if (feat != nullptr) {
                // straight into the rasterizer's feature row (columns pbr 2..4 | diffuse light 12..14 | mean visibility 15: what
                // s2_pack_features_kernel copies there; the other columns were written with the activations)
                float* f = feat + (size_t)g * 16;
                f[2] = acc[0]; f[3] = acc[1]; f[4] = acc[2];
                *reinterpret_cast<float4*>(f + 12) = make_float4(acc[3], acc[4], acc[5], acc[6]);
            }
        }
    }
}

// =====================================================================================================================
// Backward: gradients of <pbr, g_pbr> + <diffuse_light, g_diff> w.r.t. base colour, roughness, view direction, the ROTATED
// incident-light coefficients (dcp [P,16,3]; frs_rotate_kernel<true> takes them back) and the environment texture.
// Per-sample arithmetic = the general kernels' (neilf.py:339-407 differentiated) in the ray frame; the view gradient is
//     dL/dV = sum_k gLoV_k L_k + (sum_k gNoV_k) N,      gLoV = dVoH / (4 |u|) - dNoH N.H / (4 |u|^2),   gNoV = dNoV + dNoH / (2 |u|)
// (projected onto the tangent plane of V afterwards, as normalize() does): sum_k gLoV_k s_k z_k is one more product against the
// table -- rows 1..3 of the gradient layout -- and is rotated to world space once per Gaussian.  The texture gradient goes
// through the same 64-bit fixed-point LDS accumulators as in the general backward.
// =====================================================================================================================
template <bool TAB_LDS>
__global__ void __launch_bounds__(64 * FRS_WAVES, 2)
shade_backward_frs_kernel(int P, int K, FrsSrc src, const float* __restrict__ env /* [He*We][3] */, int He, int We,
                          float uniform_area, const float* __restrict__ tables, const uint8_t* __restrict__ valid,
                          float* __restrict__ d_base, float* __restrict__ d_rough, float* __restrict__ d_view,
                          float* __restrict__ dcp, float* __restrict__ d_env, const unsigned int* __restrict__ gmax_bits,
                          int gmax_n)
{
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    const int ntexel = He * We;
    float4* s_env4 = reinterpret_cast<float4*>(s_mem);
    long long* s_denv = reinterpret_cast<long long*>(s_mem + 4 * ntexel);        // [texel][3] 64-bit fixed point
    const unsigned int gmax_word = wave_gmax_bits(gmax_bits, gmax_n);
    const float gmax = __uint_as_float(gmax_word);
    const bool fixed = gmax_usable(gmax_word);
    const float fx_scale = fixed ? 34359738368.0f / gmax : 0.f;                 // 2^35 / max|g|
    const float fx_clamp = gmax * 8192.0f;
    const int nblk = (K + 15) >> 4;
    float* s_stage = s_mem + ((10 * ntexel + 3) & ~3);                           // FRS_WAVES x FRS_ST_BWD floats (16-byte aligned)
    float* s_tab = s_stage + FRS_WAVES * FRS_ST_BWD;                             // TAB_LDS: the nblk x 512 table words
    frs_stage_texture(env, ntexel, s_env4);
    for (int i = threadIdx.x; i < 3 * ntexel; i += blockDim.x) s_denv[i] = 0;
    if (TAB_LDS)
        for (int i = threadIdx.x; i < nblk * 512; i += blockDim.x) s_tab[i] = tables[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gl = lane & 15, q = lane >> 4;
    const float invK = 1.0f / (float)K;
    const int ngroups = (P + FRS_G - 1) / FRS_G;
    float* st = s_stage + wave * FRS_ST_BWD;                                     // this wave's staging area (frs_stage_group)
    const unsigned int st_addr = lds_address_of(st);
    const int gstride = gridDim.x * FRS_WAVES;
    int grp = blockIdx.x * FRS_WAVES + wave;
    uint8_t nvalid = 0;
    if (grp < ngroups) {
        nvalid = valid[min(grp * FRS_G + gl, P - 1)];
        frs_stage_group<true>(st_addr, grp, P, K, lane, src);
    }
    for (; grp < ngroups; grp += gstride) {
        const int g = grp * FRS_G + gl;
        const int gc = min(g, P - 1);
        const bool live_g = g < P && nvalid != 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // this group's staged data have landed
        float u[64];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            u[48 + c] = st[FRS_ST_BASE + 3 * gl + c];
            u[52 + c] = st[FRS_ST_NRM + 3 * gl + c];
            u[55 + c] = st[FRS_ST_VIEW + 3 * gl + c];
        }
        u[51] = st[FRS_ST_RGH + gl];
        const float rn0 = st[FRS_ST_RNRM + 3 * gl], rn1 = st[FRS_ST_RNRM + 3 * gl + 1], rn2 = st[FRS_ST_RNRM + 3 * gl + 2];
        const float gp[3] = {st[FRS_ST_GP + 3 * gl] * invK, st[FRS_ST_GP + 3 * gl + 1] * invK, st[FRS_ST_GP + 3 * gl + 2] * invK};
        const float gd[3] = {st[FRS_ST_GD + 3 * gl] * invK, st[FRS_ST_GD + 3 * gl + 1] * invK, st[FRS_ST_GD + 3 * gl + 2] * invK};
        float bc[4][3];
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int f = (4 * s + q) * 3 + c;
                bc[s][c] = st[FRS_ST_CP + (f >> 4) * 256 + gl * 16 + (f & 15)];
            }
        FrsBlock nxt = frs_staged_block(st, gl, q, K);
        GaussFwd G;
        gauss_setup(G, u);
        FrsFrame F;
        {
            float R[9];
            frs_rotation(rn0, rn1, rn2, R);
            frs_frame(F, R, G.n[0], G.n[1], G.n[2], G.V[0], G.V[1], G.V[2], q);
        }
        const float fd[3] = {G.base[0] / kPi, G.base[1] / kPi, G.base[2] / kPi};
        const float nom1 = G.NoV * (1.f - G.kk) + G.kk;
        f32x4 dcq[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // dL/dc'[4 q + v'][c]
        f32x4 dvq = {0.f, 0.f, 0.f, 0.f};          // q == 0: (., -C1 sum y g, C1 sum z g, -C1 sum x g), g = gLoV_k s_k
        float accb[5] = {0.f, 0.f, 0.f, 0.f, 0.f};               // albedo 3, roughness, sum_k gNoV_k
        const size_t row = (size_t)gc * (size_t)K;
        int ngrp_l = min(grp + gstride, ngroups - 1);                            // (past the end: a harmless reload of the last group)
        int ngc_l = min(ngrp_l * FRS_G + gl, P - 1);
        for (int b = 0; b < nblk; b++) {
            const FrsBlock cur = nxt;
            // table words first, then the prefetch (see the forward); with the tables in LDS (K <= FRS_TAB_LDS_MAX_K) no vector
            // memory load at all sits between the prefetch and the end of the block
            const float* tb = (TAB_LDS ? s_tab : tables) + (size_t)b * 512 + lane;
            float a[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
            if (!TAB_LDS) {                       // (K > 128: all eight words of the block before the prefetch)
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    a[s] = tb[64 * s];
                    a2[s] = tb[64 * (4 + s)];
                }
            }
            const bool last = b + 1 == nblk;
            if (!last) {
                nxt = frs_load_block(row, 16 * (b + 1) + 4 * q, K, src.visibility, src.taps);
            } else {
                asm volatile("" : "+v"(ngrp_l), "+v"(ngc_l));         // (not hoisted out of the block loop: see the forward)
                nvalid = valid[ngc_l];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (TAB_LDS) {
#pragma unroll
                for (int s = 0; s < 4; s++) a[s] = tb[64 * s];
            }
            f32x4 l[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int c = 0; c < 3; c++) l[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], bc[s][c], l[c], 0, 0, 0);
            const FrsGeom geo = frs_block_geometry(F, a[0], a[1], a[2]);
            if (last) frs_stage_group<true>(st_addr, ngrp_l, P, K, lane, src);
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int k = 16 * b + 4 * q + v;
                const bool ok = live_g && k < K;
                float vis;
                PackedTap t;
                frs_sample_of(cur, v, vis, t);
                float e[3], w4[4];
                int tex[4];
                env_fetch(t, s_env4, He, We, e, tex, w4);
                const float dn = geo.dn[v], lov = geo.lov[v];
                const float area_ndi = ok ? uniform_area * fmaxf(dn, 0.f) : 0.f;
                const float rawNoL = dn * G.nscale;
                const float uu = fmaxf(0.5f * lov + 0.5f, 1e-24f);
                const float uinv = __builtin_amdgcn_rsqf(uu);
                const float NoL = fminf(fmaxf(rawNoL, 1e-6f), 1.f);
                const float rawNoH = (rawNoL + G.rawNoV) * (0.5f * uinv), rawVoH = uu * uinv;
                const float NoH = fminf(fmaxf(rawNoH, 1e-6f), 1.f), VoH = fminf(fmaxf(rawVoH, 1e-6f), 1.f);
                const float p2 = exp2f((-5.55473f * VoH - 6.98316f) * VoH);
                const float frac0 = 0.04f + 0.96f * p2;
                const float frac = frac0 * G.a2;
                const float nom0 = NoH * NoH * (G.a2 - 1.f) + 1.f;
                const float nom2 = NoL * (1.f - G.kk) + G.kk;
                const float nomr = 4.f * kPi * nom0 * nom0 * nom1 * nom2;
                const float nom = fminf(fmaxf(nomr, 1e-6f), 4.f * kPi);
                const float spec = frac / nom;
                float gspec = 0.f, dlin[3], dl[3];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float lc = l[c][v];
                    const float lin = fmaxf(lc, 0.f) + e[c] * vis;
                    const float transport = lin * area_ndi;
                    const float dT = gp[c] * (fd[c] + spec) + gd[c];      // dL / d transport_c
                    gspec += gp[c] * transport;
                    accb[c] += gp[c] * transport / kPi;                   // albedo
                    dlin[c] = dT * area_ndi;                              // dL / d (incident light)_c
                    dl[c] = lc >= 0.f ? dlin[c] : 0.f;                    // clamp_min(0): gradient where the SH sum >= 0
                }
                // environment-texture gradient: 4 taps x 3 channels (an out-of-range tap carries weight 0)
                if (vis != 0.f && area_ndi != 0.f) {
                    const float ev[3] = {dlin[0] * vis, dlin[1] * vis, dlin[2] * vis};
                    if (fixed) {
                        const double scale_d = (double)fx_scale;
#pragma unroll
                        for (int tq = 0; tq < 4; tq++) {
#pragma unroll
                            for (int c = 0; c < 3; c++) {
                                const float cl = __builtin_amdgcn_fmed3f(ev[c] * w4[tq], -fx_clamp, fx_clamp);
                                const double dsum = __builtin_fma((double)cl, scale_d, 6755399441055744.0);
                                const unsigned long long bits =
                                    (unsigned long long)__double_as_longlong(dsum) - 0x4338000000000000ull;
                                atomicAdd(reinterpret_cast<unsigned long long*>(&s_denv[3 * tex[tq] + c]), bits);
                            }
                        }
                    } else {
#pragma unroll
                        for (int tq = 0; tq < 4; tq++)
                            if (w4[tq] != 0.f)
#pragma unroll
                                for (int c = 0; c < 3; c++) atomicAdd(&d_env[3 * (size_t)tex[tq] + c], ev[c] * w4[tq]);
                    }
                }
                // specular -> roughness, view direction
                const bool nom_free = nomr >= 1e-6f && nomr <= 4.f * kPi;
                const float dfrac = gspec / nom;
                const float dnom = nom_free ? -gspec * frac / (nom * nom) : 0.f;
                float da2 = dfrac * frac0;
                const float dfrac0 = dfrac * G.a2;
                const float dFMi = dfrac0 * 0.96f * 0.6931471805599453f * p2;
                float dVoH = dFMi * (-2.f * 5.55473f * VoH - 6.98316f);
                const float c4 = 4.f * kPi;
                const float dnom0 = dnom * c4 * 2.f * nom0 * nom1 * nom2;
                const float dnom1 = dnom * c4 * nom0 * nom0 * nom2;
                const float dnom2 = dnom * c4 * nom0 * nom0 * nom1;
                float dNoH = dnom0 * 2.f * NoH * (G.a2 - 1.f);
                da2 += dnom0 * NoH * NoH;
                float dNoV = dnom1 * (1.f - G.kk);
                const float dkk = dnom1 * (1.f - G.NoV) + dnom2 * (1.f - NoL);
                const float da = dkk / 8.f + da2 * 2.f * G.a;
                accb[3] += dkk * 2.f / 8.f + da * 2.f * G.r;                // roughness
                if (!(rawNoH >= 1e-6f && rawNoH <= 1.f)) dNoH = 0.f;
                if (!(rawVoH >= 1e-6f && rawVoH <= 1.f)) dVoH = 0.f;
                if (!(G.rawNoV >= 1e-6f && G.rawNoV <= 1.f)) dNoV = 0.f;
                const float q4 = 0.25f * uinv;
                const float gLoV = dVoH * q4 - dNoH * rawNoH * (q4 * uinv);
                accb[4] += dNoV + dNoH * (0.5f * uinv);
                // gradient products: dc'[i][c] += Yz[k][i] dl[c]; view: L_k = s_k R z_k, so the weight of z_k is gLoV_k s_k
                const float a2v = TAB_LDS ? tb[64 * (4 + v)] : a2[v];
#pragma unroll
                for (int c = 0; c < 3; c++) dcq[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2v, dl[c], dcq[c], 0, 0, 0);
                dvq = __builtin_amdgcn_mfma_f32_16x16x4f32(a2v, gLoV * geo.sk[v], dvq, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 5; i++) accb[i] = frs_sum4(accb[i]);
        if (live_g) {
            if (q == 0) {
                constexpr float C1 = 0.4886025119029199f;
                float R[9];
                frs_rotation(rn0, rn1, rn2, R);
                const float cx = -dvq[3] / C1, cy = -dvq[1] / C1, cz = dvq[2] / C1;            // sum_k gLoV_k s_k z_k (ray frame)
                float dV[3];
#pragma unroll
                for (int m = 0; m < 3; m++) dV[m] = R[3 * m] * cx + R[3 * m + 1] * cy + R[3 * m + 2] * cz + accb[4] * G.N[m];
                const float vd = G.V[0] * dV[0] + G.V[1] * dV[1] + G.V[2] * dV[2];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    d_base[3 * (size_t)g + c] = accb[c];
                    d_view[3 * (size_t)g + c] = (dV[c] - G.V[c] * vd) / G.vlen;
                }
                d_rough[g] = accb[3];
            }
            // lane (g, q) holds dL/dc'[4 q + v'][c] in dcq[c][v']: 12 consecutive floats of the Gaussian's row
            float4* o = reinterpret_cast<float4*>(dcp + (size_t)g * 48 + 12 * q);
            o[0] = make_float4(dcq[0][0], dcq[1][0], dcq[2][0], dcq[0][1]);
            o[1] = make_float4(dcq[1][1], dcq[2][1], dcq[0][2], dcq[1][2]);
            o[2] = make_float4(dcq[2][2], dcq[0][3], dcq[1][3], dcq[2][3]);
        }
    }
    if (fixed) {
        __syncthreads();
        const float inv = 1.0f / fx_scale;
        for (int i = threadIdx.x; i < 3 * ntexel; i += blockDim.x) {
            const long long v64 = s_denv[i];
            if (v64 != 0) atomicAdd(&d_env[i], (float)((double)v64 * (double)inv));
        }
    }
}

// =====================================================================================================================
// The Gaussians OFF the rotated path (R(n) not orthonormal to FRS_MAX_DEFECT: ~0.3 % of them): one wave per Gaussian, lane =
// sample, the general per-sample arithmetic of shading.hip (SH basis evaluated at the true direction) on directions that are
// REGENERATED from the ray normal and the z set exactly as the cache was generated -- nothing per sample is read but the
// visibility.  Rounds 3's launch of the general persistent kernels on a list cost 25 + 46..87 us per iteration for 772 Gaussians
// (prologues, DMA pipeline and flush of a grid sized for 300k) and 0.88 ms at 2M Gaussians; these are plain grid-stride
// kernels over a few hundred waves.
// =====================================================================================================================
constexpr int FRS_LISTED_WAVES = 4;

// record u[64] of one listed Gaussian -> LDS (lane l loads element l: 0..47 incidents, 48..50 albedo, 51 roughness, 52..54 normal,
// 55..57 view direction, 58..60 ray normal, BWD: 61..63 unused; the upstream gradients are loaded by the caller)
__device__ __forceinline__ float frs_listed_record(int lane, int g, const FrsSrc& p, const float* __restrict__ incidents)
{
    const float* q = incidents + (size_t)g * 48 + lane;
    if (lane >= 48) q = p.base_color + 3 * (size_t)g + (lane - 48);
    if (lane == 51) q = p.roughness + g;
    if (lane >= 52) q = p.normals + 3 * (size_t)g + (lane - 52);
    if (lane >= 55) q = p.viewdirs + 3 * (size_t)g + (lane - 55);
    if (lane >= 58) q = p.ray_normals + 3 * (size_t)g + (lane - 58);
    return lane < 61 ? *q : 0.f;
}

// sample k of the ray set of R: normalize(R z_k), as frs_build_taps_kernel
__device__ __forceinline__ void frs_listed_direction(const float (&R)[9], const float* __restrict__ zsamples, int k, float& dx,
                                                     float& dy, float& dz)
{
    const float zx = zsamples[3 * k], zy = zsamples[3 * k + 1], zz = zsamples[3 * k + 2];
    dx = R[0] * zx + R[1] * zy + R[2] * zz; dy = R[3] * zx + R[4] * zy + R[5] * zz; dz = R[6] * zx + R[7] * zy + R[8] * zz;
    const float len = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-12f);
    dx /= len; dy /= len; dz /= len;
}

__global__ void __launch_bounds__(64 * FRS_LISTED_WAVES)
shade_forward_frs_listed_kernel(int n_list, const int* __restrict__ list, int K, FrsSrc src,
                                const float* __restrict__ incidents, const float* __restrict__ env, int He, int We,
                                const float* __restrict__ zsamples, float uniform_area, float* __restrict__ out,
                                float* __restrict__ feat)
{
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    float* s_env = s_mem;                                                // [ntexel][3]
    const int ntex = 3 * He * We;
    for (int i = threadIdx.x; i < ntex; i += blockDim.x) s_env[i] = env[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* u = s_mem + ((ntex + 3) & ~3) + wave * 64;
    const float invK = 1.0f / (float)K;
    for (int i = blockIdx.x * FRS_LISTED_WAVES + wave; i < n_list; i += gridDim.x * FRS_LISTED_WAVES) {
        const int g = __builtin_amdgcn_readfirstlane(list[i]);
        u[lane] = frs_listed_record(lane, g, src, incidents);            // same wave, in-order LDS: no barrier needed
        GaussFwd G;
        gauss_setup(G, u);
        float R[9];
        frs_rotation(u[58], u[59], u[60], R);
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int kb = 0; kb < K; kb += 64) {
            const int k = kb + lane, kc = min(k, K - 1);
            float dx, dy, dz;
            frs_listed_direction(R, zsamples, kc, dx, dy, dz);
            const float vis = src.visibility[(size_t)g * K + kc];
            SampleFwd s;
            shade_sample<true>(s, G, u, 16, dx, dy, dz, vis, uniform_area, nullptr, s_env, nullptr, He, We);
            if (k < K) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    v[c] += (G.base[c] / kPi + s.spec) * s.transport[c];
                    v[3 + c] += s.transport[c];
                }
                v[6] += vis;
            }
        }
        const float r = transpose_reduce<8, true>(v);
        const int ch = transposed_channel<8>(lane);
        if (transposed_owner<8>(lane) && ch < 7) {
            out[(size_t)g * SHADE_NOUT + (ch < 6 ? ch : 18)] = r * invK;
            if (feat != nullptr) feat[(size_t)g * 16 + (ch < 3 ? 2 + ch : 9 + ch)] = r * invK;      // (3..5 -> 12..14, 6 -> 15)
        }
    }
}

// d_base / d_rough / d_view / d_inc rows of the listed Gaussians are written, d_env is accumulated (through 64-bit fixed-point
// LDS accumulators per workgroup, flushed once: the grid is small)
__global__ void __launch_bounds__(64 * FRS_LISTED_WAVES)
shade_backward_frs_listed_kernel(int n_list, const int* __restrict__ list, int K, FrsSrc src,
                                 const float* __restrict__ incidents, const float* __restrict__ env, int He, int We,
                                 const float* __restrict__ zsamples, float uniform_area, float* __restrict__ d_base,
                                 float* __restrict__ d_rough, float* __restrict__ d_view, float* __restrict__ d_inc,
                                 float* __restrict__ d_env, const unsigned int* __restrict__ gmax_bits, int gmax_n)
{
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    const int ntex = 3 * He * We, ntex4 = (ntex + 3) & ~3;
    float* s_env = s_mem;
    long long* s_denv = reinterpret_cast<long long*>(s_mem + ntex4);
    const unsigned int gmax_word = wave_gmax_bits(gmax_bits, gmax_n);
    const float gmax = __uint_as_float(gmax_word);
    const bool fixed = gmax_usable(gmax_word);
    const float fx_scale = fixed ? 34359738368.0f / gmax : 0.f;          // 2^35 / max|g|
    const float fx_clamp = gmax * 8192.0f;
    for (int i = threadIdx.x; i < ntex; i += blockDim.x) {
        s_env[i] = env[i];
        s_denv[i] = 0;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* u = s_mem + 3 * ntex4 + wave * 64;
    const float invK = 1.0f / (float)K;
    for (int i = blockIdx.x * FRS_LISTED_WAVES + wave; i < n_list; i += gridDim.x * FRS_LISTED_WAVES) {
        const int g = __builtin_amdgcn_readfirstlane(list[i]);
        u[lane] = frs_listed_record(lane, g, src, incidents);
        GaussFwd G;
        gauss_setup(G, u);
        float R[9];
        frs_rotation(u[58], u[59], u[60], R);
        float gp[3], gd[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            gp[c] = src.g_pbr[3 * (size_t)g + c] * invK;
            gd[c] = src.g_diff[3 * (size_t)g + c] * invK;
        }
        // 48 coefficient gradients (f = i * 3 + c), albedo 3, roughness, view 3 -- 55 of the 64 channels of one transposing reduction
        float acc[64];
#pragma unroll
        for (int f = 0; f < 64; f++) acc[f] = 0.f;
        for (int kb = 0; kb < K; kb += 64) {
            const int k = kb + lane, kc = min(k, K - 1);
            float dx, dy, dz;
            frs_listed_direction(R, zsamples, kc, dx, dy, dz);
            const float vis = src.visibility[(size_t)g * K + kc];
            SampleFwd s;
            shade_sample<true>(s, G, u, 16, dx, dy, dz, vis, k < K ? uniform_area : 0.f, nullptr, s_env, nullptr, He, We);
            float gspec = 0.f, dlin[3], dl[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float dT = gp[c] * (G.base[c] / kPi + s.spec) + gd[c];
                gspec += gp[c] * s.transport[c];
                acc[48 + c] += gp[c] * s.transport[c] / kPi;
                dlin[c] = dT * s.area_ndi;
                dl[c] = s.shsum[c] >= 0.f ? dlin[c] : 0.f;
            }
            if (s.vis != 0.f && s.area_ndi != 0.f) {
                const float ev[3] = {dlin[0] * s.vis, dlin[1] * s.vis, dlin[2] * s.vis};
                const double scale_d = (double)fx_scale;
#pragma unroll
                for (int tt = 0; tt < 4; tt++) {
                    if (s.taps.idx[tt] < 0) continue;
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const float val = ev[c] * s.taps.w[tt];
                        if (fixed) {
                            const float cl = __builtin_amdgcn_fmed3f(val, -fx_clamp, fx_clamp);
                            const double dsum = __builtin_fma((double)cl, scale_d, 6755399441055744.0);
                            const unsigned long long bits = (unsigned long long)__double_as_longlong(dsum) - 0x4338000000000000ull;
                            atomicAdd(reinterpret_cast<unsigned long long*>(&s_denv[3 * s.taps.idx[tt] + c]), bits);
                        } else {
                            atomicAdd(&d_env[3 * (size_t)s.taps.idx[tt] + c], val);
                        }
                    }
                }
            }
            const float frac = s.frac0 * G.a2;
            const bool nom_free = s.nomr >= 1e-6f && s.nomr <= 4.f * kPi;
            const float nom = fminf(fmaxf(s.nomr, 1e-6f), 4.f * kPi);
            const float dfrac = gspec / nom;
            const float dnom = nom_free ? -gspec * frac / (nom * nom) : 0.f;
            float da2 = dfrac * s.frac0;
            const float dFMi = dfrac * G.a2 * 0.96f * 0.6931471805599453f * s.p2;
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
            acc[51] += dkk * 2.f / 8.f + da * 2.f * G.r;
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
            for (int c = 0; c < 3; c++) acc[52 + c] += (dV[c] - G.V[c] * vd) / G.vlen;
#pragma unroll
            for (int f = 0; f < 48; f++) acc[f] += dl[f % 3] * s.Y[f / 3];
        }
        const float r = transpose_reduce<64, true>(acc);
        const int ch = transposed_channel<64>(lane);
        if (ch < 48) d_inc[(size_t)g * 48 + ch] = r;
        else if (ch < 51) d_base[3 * (size_t)g + (ch - 48)] = r;
        else if (ch == 51) d_rough[g] = r;
        else if (ch < 55) d_view[3 * (size_t)g + (ch - 52)] = r;
    }
    if (fixed) {
        __syncthreads();
        const float inv = 1.0f / fx_scale;
        for (int i = threadIdx.x; i < ntex; i += blockDim.x) {
            const long long v64 = s_denv[i];
            if (v64 != 0) atomicAdd(&d_env[i], (float)((double)v64 * (double)inv));
        }
    }
}

}  // namespace r3dg
