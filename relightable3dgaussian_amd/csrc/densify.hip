// Densification bookkeeping of the training loop, gfx950.
//
//   densify_accumulate_kernel   GaussianModel.add_densification_stats (scene/gaussian_model.py:931-937) + the max-radii
//                               update of train.py:164-165 in one pass over the Gaussians
//   densify_classify/scan/scatter   the decisions of densify_and_prune (:893-915) = densify_and_clone (:846-891) ->
//                               densify_and_split (:790-844) -> final prune, or of prune (:917-929), turned into ONE row map
//                               (destination row -> source row + kind) instead of two torch.cat passes and two
//                               boolean-mask passes over every parameter and both Adam moments
//   densify_gather_kernel       moves every surviving byte exactly once: all parameter groups and their exp_avg /
//                               exp_avg_sq in one launch (group table by value in the kernel arguments); split children
//                               get their sampled position and shrunken scale on the way
//   reset_opacity_kernel        GaussianModel.reset_opacity (:563-566) + replace_tensor_to_optimizer (:667-679)
//
// The reference's order of rows is kept: surviving originals, surviving clones, then the n_split blocks of surviving split
// children.  As-written behaviour that is kept on purpose (oracle/densify.py Q1-Q4): max_radii2D is already zero when a
// densify call evaluates the screen-size test; appended rows enter the final prune with weights_accum = 1; the split tests
// the signed mean, the clone its absolute value; 0/0 statistics count as 0.
// Decisions come from fp32 comparisons, so this file is compiled without FMA contraction (build.py).
#include "common.hpp"
#include "r3dg_hip.h"

namespace r3dg {

enum : uint8_t { DC_KEEP = 1, DC_CLONE = 2, DC_SPLIT = 4, DC_CHILD = 8 };

__global__ void __launch_bounds__(256)
densify_accumulate_kernel(int P, const float* __restrict__ viewspace_grad, const float* __restrict__ normal_grad,
                          const int* __restrict__ radii, const float* __restrict__ weights,
                          float* __restrict__ xyz_accum, float* __restrict__ normal_accum, float* __restrict__ denom,
                          float* __restrict__ weights_accum, float* __restrict__ max_radii2D,
                          const float* __restrict__ skip_flag)
{
    // a view the bounded forward dropped on the device (r3dg_rasterize_forward_begin_bounded) is no observation
    if (skip_flag != nullptr && *skip_flag != 0.0f) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    weights_accum[i] += weights[i];
    const int r = radii[i];
    if (r <= 0) return;                                   // visibility_filter = radii > 0
    const size_t i3 = 3 * (size_t)i;
    const float gx = viewspace_grad[i3], gy = viewspace_grad[i3 + 1];
    xyz_accum[i] += sqrtf(gx * gx + gy * gy);
    if (normal_grad) {
        const float nx = normal_grad[i3], ny = normal_grad[i3 + 1], nz = normal_grad[i3 + 2];
        normal_accum[i] += sqrtf(nx * nx + ny * ny + nz * nz);
    }
    denom[i] += 1.f;
    max_radii2D[i] = fmaxf(max_radii2D[i], (float)r);
}

__device__ __forceinline__ float sigmoid_as_torch(float x) { return 1.f / (1.f + expf(-x)); }

// One code byte per Gaussian + the four per-block totals (keep, clone, split, child) the scan needs.
__global__ void __launch_bounds__(256)
densify_classify_kernel(int P, r3dg_densify_config cfg, const float* __restrict__ scaling_raw,
                        const float* __restrict__ opacity_raw, const float* __restrict__ xyz_accum,
                        const float* __restrict__ normal_accum, const float* __restrict__ denom,
                        const float* __restrict__ weights_accum, const float* __restrict__ max_radii2D,
                        uint8_t* __restrict__ codes, uint32_t* __restrict__ block_counts)
{
    __shared__ uint32_t s_cnt[4];
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    uint8_t code = 0;
    if (i < P) {
        const size_t i3 = 3 * (size_t)i;
        const float s0 = expf(scaling_raw[i3]), s1 = expf(scaling_raw[i3 + 1]), s2 = expf(scaling_raw[i3 + 2]);
        const float smax = fmaxf(fmaxf(s0, s1), s2);
        const bool op_bad = sigmoid_as_torch(opacity_raw[i]) < cfg.min_opacity;
        const bool has_size = cfg.max_screen_size != 0.f;            // `if max_screen_size:` (None / 0 -> no size tests)
        if (cfg.mode == 1) {                                         // prune(): statistics as they stand
            bool bad = op_bad || weights_accum[i] < cfg.weights_threshold;
            if (has_size) bad = bad || max_radii2D[i] > cfg.max_screen_size || smax > cfg.world_size_limit;
            code = bad ? 0 : DC_KEEP;
        } else {
            float g = xyz_accum[i] / denom[i], gn = normal_accum[i] / denom[i];
            if (g != g) g = 0.f;
            if (gn != gn) gn = 0.f;
            const bool small = smax <= cfg.dense_size, large = smax > cfg.dense_size;
            const bool sel_clone = (fabsf(g) >= cfg.grad_threshold || fabsf(gn) >= cfg.grad_normal_threshold) && small;
            const bool sel_split = (g >= cfg.grad_threshold || gn >= cfg.grad_normal_threshold) && large;
            // final prune; the screen-space radius every row carries at that point is 0 (densification_postfix)
            const bool radius_bad = has_size && (0.f > cfg.max_screen_size);
            const bool new_weight_bad = 1.f < cfg.weights_threshold;
            const bool world_bad = has_size && smax > cfg.world_size_limit;
            if (!sel_split && !(op_bad || weights_accum[i] < cfg.weights_threshold || radius_bad || world_bad))
                code |= DC_KEEP;
            if (sel_clone && !(op_bad || new_weight_bad || radius_bad || world_bad)) code |= DC_CLONE;
            if (sel_split) {
                code |= DC_SPLIT;
                const float c0 = expf(logf(s0 / cfg.split_divisor)), c1 = expf(logf(s1 / cfg.split_divisor)),
                            c2 = expf(logf(s2 / cfg.split_divisor));
                const bool child_world_bad = has_size && fmaxf(fmaxf(c0, c1), c2) > cfg.world_size_limit;
                if (!(op_bad || new_weight_bad || radius_bad || child_world_bad)) code |= DC_CHILD;
            }
        }
        codes[i] = code;
    }
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const unsigned long long m = __ballot((code >> b) & 1);
        if (lane_id() == 0 && m) atomicAdd(&s_cnt[b], (uint32_t)__popcll(m));
    }
    __syncthreads();
    if (threadIdx.x < 4) block_counts[4 * (size_t)blockIdx.x + threadIdx.x] = s_cnt[threadIdx.x];
}

// Exclusive scan of the per-block totals (one workgroup; the table has P/256 rows) + the totals themselves.
__global__ void __launch_bounds__(1024)
densify_scan_kernel(int nblocks, int n_split, uint32_t* __restrict__ block_counts, int32_t* __restrict__ counts)
{
    __shared__ uint32_t s_wave[16][4];
    __shared__ uint32_t s_carry[4];
    if (threadIdx.x < 4) s_carry[threadIdx.x] = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    for (int base = 0; base < nblocks; base += 1024) {
        const int i = base + (int)threadIdx.x;
        uint32_t v[4] = {0, 0, 0, 0}, inc[4];
        if (i < nblocks) {
#pragma unroll
            for (int b = 0; b < 4; b++) v[b] = block_counts[4 * (size_t)i + b];
        }
#pragma unroll
        for (int b = 0; b < 4; b++) {
            inc[b] = wave_inclusive_scan_u32(v[b]);
            if (lane_id() == 63) s_wave[wave][b] = inc[b];
        }
        __syncthreads();
        uint32_t off[4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            uint32_t o = s_carry[b];
            for (int w = 0; w < wave; w++) o += s_wave[w][b];
            off[b] = o;
        }
        if (i < nblocks) {
#pragma unroll
            for (int b = 0; b < 4; b++) block_counts[4 * (size_t)i + b] = off[b] + inc[b] - v[b];
        }
        __syncthreads();
        if (threadIdx.x == 1023) {
#pragma unroll
            for (int b = 0; b < 4; b++) s_carry[b] = off[b] + inc[b];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const uint32_t keep = s_carry[0], clone = s_carry[1], split = s_carry[2], child = s_carry[3];
        counts[0] = (int32_t)(keep + clone + (uint32_t)n_split * child);
        counts[1] = (int32_t)keep;
        counts[2] = (int32_t)clone;
        counts[3] = (int32_t)split;
        counts[4] = (int32_t)child;
    }
}

// Destination row -> (source row, kind): kind -1 = surviving original (moments travel with it), -2 = clone (moments 0),
// k >= 0 = split child drawing row k of the caller's normal table (moments 0).
__global__ void __launch_bounds__(256)
densify_scatter_kernel(int P, int n_split, const uint8_t* __restrict__ codes, const uint32_t* __restrict__ block_offsets,
                       const int32_t* __restrict__ counts, int32_t* __restrict__ src_row, int32_t* __restrict__ kind)
{
    __shared__ uint32_t s_wave[4][4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const uint8_t code = i < P ? codes[i] : 0;
    const int wave = threadIdx.x >> 6;
    uint32_t rank[4];
    const unsigned long long below = (1ull << lane_id()) - 1ull;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const unsigned long long m = __ballot((code >> b) & 1);
        rank[b] = (uint32_t)__popcll(m & below);
        if (lane_id() == 0) s_wave[wave][b] = (uint32_t)__popcll(m);
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 4; b++) {
        uint32_t o = block_offsets[4 * (size_t)blockIdx.x + b];
        for (int w = 0; w < wave; w++) o += s_wave[w][b];
        rank[b] += o;
    }
    if (i >= P) return;
    const uint32_t n_keep = (uint32_t)counts[1], n_clone = (uint32_t)counts[2], n_split_all = (uint32_t)counts[3],
                   n_child = (uint32_t)counts[4];
    if (code & DC_KEEP) { src_row[rank[0]] = i; kind[rank[0]] = -1; }
    if (code & DC_CLONE) { src_row[n_keep + rank[1]] = i; kind[n_keep + rank[1]] = -2; }
    if (code & DC_CHILD) {
        for (int b = 0; b < n_split; b++) {
            const size_t d = (size_t)n_keep + n_clone + (size_t)b * n_child + rank[3];
            src_row[d] = i;
            kind[d] = (int32_t)((uint32_t)b * n_split_all + rank[2]);
        }
    }
}

struct GatherTable {
    r3dg_densify_group g[R3DG_DENSIFY_MAX_GROUPS];
    unsigned int first_block[R3DG_DENSIFY_MAX_GROUPS + 1];
    int n_groups;
};

__device__ __forceinline__ void quat_row(const float* __restrict__ q_raw, int c, float out[3])
{
    // build_rotation (utils/general_utils.py:82-103): normalise by the plain norm, (r, x, y, z) = q
    const float n = sqrtf(q_raw[0] * q_raw[0] + q_raw[1] * q_raw[1] + q_raw[2] * q_raw[2] + q_raw[3] * q_raw[3]);
    const float r = q_raw[0] / n, x = q_raw[1] / n, y = q_raw[2] / n, z = q_raw[3] / n;
    if (c == 0) { out[0] = 1.f - 2.f * (y * y + z * z); out[1] = 2.f * (x * y - r * z); out[2] = 2.f * (x * z + r * y); }
    else if (c == 1) { out[0] = 2.f * (x * y + r * z); out[1] = 1.f - 2.f * (x * x + z * z); out[2] = 2.f * (y * z - r * x); }
    else { out[0] = 2.f * (x * z - r * y); out[1] = 2.f * (y * z + r * x); out[2] = 1.f - 2.f * (x * x + y * y); }
}

// 1024 destination floats per block, 4 per thread (float4 stores; destination tensors are allocator-aligned).
__global__ void __launch_bounds__(256)
densify_gather_kernel(GatherTable t, unsigned int P_out, const int32_t* __restrict__ src_row,
                      const int32_t* __restrict__ kind, const float* __restrict__ xyz,
                      const float* __restrict__ scaling_raw, const float* __restrict__ rotation_raw,
                      const float* __restrict__ normal_table, float split_divisor)
{
    int gi = 0;
#pragma unroll 1
    while (gi + 1 < t.n_groups && blockIdx.x >= t.first_block[gi + 1]) gi++;
    const r3dg_densify_group grp = t.g[gi];
    const size_t n = (size_t)P_out * grp.row_floats;
    const size_t base = (size_t)(blockIdx.x - t.first_block[gi]) * 1024 + threadIdx.x * 4;
    if (base >= n) return;
    uint32_t row = (uint32_t)(base / grp.row_floats), col = (uint32_t)(base - (size_t)row * grp.row_floats);
    float pv[4], mv[4], vv[4];
    const bool moments = grp.src_exp_avg != nullptr;
    int32_t src = src_row[row], kd = kind[row];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (base + k < n) {
            const size_t s = (size_t)src * grp.row_floats + col;
            float p = grp.src_param[s];
            if (kd >= 0 && grp.role == R3DG_DENSIFY_ROLE_XYZ) {
                // new_xyz = R(q) (scale * z) + xyz   (densify_and_split, gaussian_model.py:808-811)
                const size_t s3 = 3 * (size_t)src, z3 = 3 * (size_t)kd;
                float Rrow[3];
                quat_row(rotation_raw + 4 * (size_t)src, (int)col, Rrow);
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < 3; j++) acc += Rrow[j] * (expf(scaling_raw[s3 + j]) * normal_table[z3 + j]);
                p = acc + xyz[s3 + col];
            } else if (kd >= 0 && grp.role == R3DG_DENSIFY_ROLE_SCALING) {
                p = logf(expf(p) / split_divisor);        // scaling_inverse_activation(get_scaling / (0.8 N)), :814
            }
            pv[k] = p;
            if (moments) {
                mv[k] = kd == -1 ? grp.src_exp_avg[s] : 0.f;
                vv[k] = kd == -1 ? grp.src_exp_avg_sq[s] : 0.f;
            }
        }
        if (++col == grp.row_floats) {
            col = 0;
            row++;
            if (row < P_out) { src = src_row[row]; kd = kind[row]; }
        }
    }
    if (base + 4 <= n) {
        *reinterpret_cast<float4*>(grp.dst_param + base) = *reinterpret_cast<float4*>(pv);
        if (moments) {
            *reinterpret_cast<float4*>(grp.dst_exp_avg + base) = *reinterpret_cast<float4*>(mv);
            *reinterpret_cast<float4*>(grp.dst_exp_avg_sq + base) = *reinterpret_cast<float4*>(vv);
        }
    } else {
        for (int k = 0; k < 4; k++)
            if (base + k < n) {
                grp.dst_param[base + k] = pv[k];
                if (moments) { grp.dst_exp_avg[base + k] = mv[k]; grp.dst_exp_avg_sq[base + k] = vv[k]; }
            }
    }
}

__global__ void __launch_bounds__(256)
reset_opacity_kernel(int P, float cap, float* __restrict__ opacity_raw, float* __restrict__ exp_avg,
                     float* __restrict__ exp_avg_sq)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float o = fminf(sigmoid_as_torch(opacity_raw[i]), cap);
    opacity_raw[i] = logf(o / (1.f - o));                 // inverse_sigmoid (utils/general_utils.py:17-18)
    if (exp_avg) exp_avg[i] = 0.f;
    if (exp_avg_sq) exp_avg_sq[i] = 0.f;
}

// ---- launchers ------------------------------------------------------------------------------------------------------
void launch_densify_accumulate(hipStream_t s, int P, const float* viewspace_grad, const float* normal_grad,
                               const int* radii, const float* weights, float* xyz_accum, float* normal_accum,
                               float* denom, float* weights_accum, float* max_radii2D, const float* skip_flag)
{
    densify_accumulate_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, viewspace_grad, normal_grad, radii, weights, xyz_accum,
                                                              normal_accum, denom, weights_accum, max_radii2D, skip_flag);
    check_launch(s, false, "densify_accumulate_kernel");
}

size_t densify_temp_bytes(size_t P)
{
    const size_t nblocks = (P + 255) / 256;
    return align_up(P, 256) + align_up(nblocks * 16, 256);
}

void launch_densify_plan(hipStream_t s, int P, const r3dg_densify_config& cfg, const float* scaling_raw,
                         const float* opacity_raw, const float* xyz_accum, const float* normal_accum,
                         const float* denom, const float* weights_accum, const float* max_radii2D, int32_t* src_row,
                         int32_t* kind, int32_t* counts, void* temp)
{
    const int nblocks = (P + 255) / 256;
    uint8_t* codes = static_cast<uint8_t*>(temp);
    uint32_t* block_counts = reinterpret_cast<uint32_t*>(codes + align_up((size_t)P, 256));
    densify_classify_kernel<<<nblocks, 256, 0, s>>>(P, cfg, scaling_raw, opacity_raw, xyz_accum, normal_accum, denom,
                                                    weights_accum, max_radii2D, codes, block_counts);
    check_launch(s, false, "densify_classify_kernel");
    densify_scan_kernel<<<1, 1024, 0, s>>>(nblocks, cfg.n_split, block_counts, counts);
    check_launch(s, false, "densify_scan_kernel");
    densify_scatter_kernel<<<nblocks, 256, 0, s>>>(P, cfg.n_split, codes, block_counts, counts, src_row, kind);
    check_launch(s, false, "densify_scatter_kernel");
}

void launch_densify_gather(hipStream_t s, int P_out, const int32_t* src_row, const int32_t* kind, int n_groups,
                           const r3dg_densify_group* groups, const float* xyz, const float* scaling_raw,
                           const float* rotation_raw, const float* normal_table, float split_divisor)
{
    GatherTable t;
    t.n_groups = n_groups;
    unsigned int blocks = 0;
    for (int i = 0; i < n_groups; i++) {
        t.g[i] = groups[i];
        t.first_block[i] = blocks;
        blocks += (unsigned int)(((size_t)P_out * groups[i].row_floats + 1023) / 1024);
    }
    t.first_block[n_groups] = blocks;
    if (blocks == 0) return;
    densify_gather_kernel<<<blocks, 256, 0, s>>>(t, (unsigned int)P_out, src_row, kind, xyz, scaling_raw, rotation_raw,
                                                 normal_table, split_divisor);
    check_launch(s, false, "densify_gather_kernel");
}

void launch_reset_opacity(hipStream_t s, int P, float cap, float* opacity_raw, float* exp_avg, float* exp_avg_sq)
{
    reset_opacity_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, cap, opacity_raw, exp_avg, exp_avg_sq);
    check_launch(s, false, "reset_opacity_kernel");
}

}  // namespace r3dg
