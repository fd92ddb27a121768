// Per-Gaussian stages of the rasterizer forward for gfx950: frustum mark (K1), preprocess (K2),
// tiles_touched scan (K3), key duplication (K5), tile ranges (K7).
//
// This translation unit is compiled with -ffp-contract=off: radii, tile rectangles and depth bits decide
// INTEGER outputs (tiles_touched, keys, sort order), which must be bit-identical to the CPU oracle, so every
// fp32 operation keeps the reference's order (forward.cu:74-258, auxiliary.h:41-164) with no fused
// multiply-add.  These kernels are HBM-bound (236 B read + ~75 B written per Gaussian); VALU cost is irrelevant.
#include <atomic>

#include "common.hpp"

namespace r3dg {

// --- 3x3 helper with glm's evaluation order (column-major, R[j][i] = A[0][i]*B[j][0]+A[1][i]*B[j][1]+A[2][i]*B[j][2]) ---
struct M3 {
    float c[3][3];
};
__device__ __forceinline__ M3 m3mul(const M3& A, const M3& B)
{
    M3 R;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++)
            R.c[j][i] = A.c[0][i] * B.c[j][0] + A.c[1][i] * B.c[j][1] + A.c[2][i] * B.c[j][2];
    return R;
}
__device__ __forceinline__ M3 m3t(const M3& A)
{
    M3 R;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++) R.c[j][i] = A.c[i][j];
    return R;
}

__device__ __forceinline__ float ndc_to_pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

// float -> int, round toward zero, saturating, NaN -> 0 (what CUDA's cvt.rzi.s32.f32 and gfx950's
// v_cvt_i32_f32 both do; spelled out because an out-of-range (int) cast is undefined in C++)
__device__ __forceinline__ int f2i_sat(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}

__device__ __forceinline__ void tile_rect(float px, float py, int max_radius, int gx, int gy, int& x0, int& y0,
                                          int& x1, int& y1)
{
    x0 = min(gx, max(0, f2i_sat((px - max_radius) / R3DG_TILE_X)));
    y0 = min(gy, max(0, f2i_sat((py - max_radius) / R3DG_TILE_Y)));
    x1 = min(gx, max(0, f2i_sat((px + max_radius + R3DG_TILE_X - 1) / R3DG_TILE_X)));
    y1 = min(gy, max(0, f2i_sat((py + max_radius + R3DG_TILE_Y - 1) / R3DG_TILE_Y)));
}

__constant__ float kSH_C0 = 0.28209479177387814f;
__constant__ float kSH_C1 = 0.4886025119029199f;
__constant__ float kSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                -1.0925484305920792f, 0.5462742152960396f};
__constant__ float kSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                -0.5900435899266435f};

// K1: reference checkFrustum / in_frustum (rasterizer_impl.cu:54-66, auxiliary.h:139-164)
__global__ void mark_visible_kernel(int P, const float* __restrict__ pts, const float* __restrict__ vm,
                                    uint8_t* __restrict__ present)
{
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    float x = pts[3 * idx], y = pts[3 * idx + 1], z = pts[3 * idx + 2];
    float vz = vm[2] * x + vm[6] * y + vm[10] * z + vm[14];
    present[idx] = vz > 0.2f;
}

// K2: reference preprocessCUDA (forward.cu:156-258).  One thread per Gaussian; each 256-thread block also
// reduces its tiles_touched into block_sums[blockIdx] so the scan (K3) never re-reads the per-Gaussian array.
// STAGED: the SH rows of the block's surviving Gaussians (192 B each, the bulk of this kernel's HBM reads) arrive through
// LDS with coalesced 16-byte loads (common.hpp stage_rows_in_256) after the cull decisions are known, instead of every
// thread walking its own row in HBM; rows of culled Gaussians are never fetched.  Same arithmetic, same results.
template <bool STAGED>
__global__ void __launch_bounds__(256)
preprocess_kernel(int P, int D, int M, const float* __restrict__ means3D, const float* __restrict__ scales,
                  float scale_modifier, const float* __restrict__ rotations, const float* __restrict__ opacities,
                  const float* __restrict__ shs, uint8_t* __restrict__ clamped, const float* __restrict__ cov3D_precomp,
                  const float* __restrict__ colors_precomp, const float* __restrict__ vm,
                  const float* __restrict__ pm, const float* __restrict__ cam_pos, int W, int H, float tan_fovx,
                  float tan_fovy, float focal_x, float focal_y, int* __restrict__ radii,
                  float2* __restrict__ means2D, float* __restrict__ depths, float* __restrict__ cov3Ds,
                  float* __restrict__ rgb, float4* __restrict__ conic_opacity, float4* __restrict__ splat, int gx, int gy,
                  uint32_t* __restrict__ tiles_touched, uint32_t* __restrict__ block_sums,
                  uint32_t* __restrict__ zero_words, int zero_n)
{
    extern __shared__ float s_rows[];                  // STAGED: 256 SH rows
    __shared__ uint8_t s_live[256];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    // (bounded forward: the tile counters of the direct binning, which runs right behind this kernel -- instead of a memset launch)
    if (idx < zero_n) zero_words[idx] = 0u;
    uint32_t my_tiles = 0;
    bool shade = false;                                // survived every cull and needs its colour from SH
    float px = 0.f, py = 0.f, pz = 0.f;
    float4 rec0 = make_float4(0.f, 0.f, 0.f, 0.f), rec1 = rec0;   // packed record of a surviving Gaussian (GeometryLayout::splat)
    if (idx < P) {
        int my_radius_i = 0;
        do {
            px = means3D[3 * idx]; py = means3D[3 * idx + 1]; pz = means3D[3 * idx + 2];
            // view-space point (auxiliary.h:58-66) and near cull (auxiliary.h:154)
            const float vx = vm[0] * px + vm[4] * py + vm[8] * pz + vm[12];
            const float vy = vm[1] * px + vm[5] * py + vm[9] * pz + vm[13];
            const float vz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
            if (vz <= 0.2f) break;
            const float hx = pm[0] * px + pm[4] * py + pm[8] * pz + pm[12];
            const float hy = pm[1] * px + pm[5] * py + pm[9] * pz + pm[13];
            const float hw = pm[3] * px + pm[7] * py + pm[11] * pz + pm[15];
            const float p_w = 1.0f / (hw + 0.0000001f);
            const float projx = hx * p_w, projy = hy * p_w;

            float c3[6];
            if (cov3D_precomp != nullptr) {
#pragma unroll
                for (int i = 0; i < 6; i++) c3[i] = cov3D_precomp[6 * idx + i];
            } else {
                // forward.cu:119-153; quaternion used as given
                const float sx = scale_modifier * scales[3 * idx], sy = scale_modifier * scales[3 * idx + 1],
                            sz = scale_modifier * scales[3 * idx + 2];
                const float r = rotations[4 * idx], x = rotations[4 * idx + 1], y = rotations[4 * idx + 2],
                            z = rotations[4 * idx + 3];
                M3 S = {{{sx, 0, 0}, {0, sy, 0}, {0, 0, sz}}};
                M3 R = {{{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                         {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                         {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}}};
                M3 Mm = m3mul(S, R);
                M3 Sg = m3mul(m3t(Mm), Mm);
                c3[0] = Sg.c[0][0]; c3[1] = Sg.c[0][1]; c3[2] = Sg.c[0][2];
                c3[3] = Sg.c[1][1]; c3[4] = Sg.c[1][2]; c3[5] = Sg.c[2][2];
#pragma unroll
                for (int i = 0; i < 6; i++) cov3Ds[6 * idx + i] = c3[i];
            }

            // EWA 2D covariance (forward.cu:74-113)
            float tx = vx, ty = vy;
            const float tz = vz;
            const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
            const float txtz = tx / tz, tytz = ty / tz;
            tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
            ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
            M3 J = {{{focal_x / tz, 0.0f, -(focal_x * tx) / (tz * tz)},
                     {0.0f, focal_y / tz, -(focal_y * ty) / (tz * tz)},
                     {0, 0, 0}}};
            M3 Wm = {{{vm[0], vm[4], vm[8]}, {vm[1], vm[5], vm[9]}, {vm[2], vm[6], vm[10]}}};
            M3 T = m3mul(Wm, J);
            M3 V = {{{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}}};
            M3 cov = m3mul(m3mul(m3t(T), m3t(V)), T);
            const float ca = cov.c[0][0] + 0.3f, cb = cov.c[0][1], cc = cov.c[1][1] + 0.3f;

            const float det = (ca * cc - cb * cb);
            if (det == 0.0f) break;
            const float det_inv = 1.f / det;
            const float mid = 0.5f * (ca + cc);
            const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
            const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
            const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
            const float pixx = ndc_to_pix(projx, W), pixy = ndc_to_pix(projy, H);
            int x0, y0, x1, y1;
            tile_rect(pixx, pixy, f2i_sat(my_radius), gx, gy, x0, y0, x1, y1);
            if ((x1 - x0) * (y1 - y0) == 0) break;

            shade = colors_precomp == nullptr;
            depths[idx] = vz;
            my_radius_i = f2i_sat(my_radius);
            means2D[idx] = make_float2(pixx, pixy);
            const float4 co = make_float4(cc * det_inv, -cb * det_inv, ca * det_inv, opacities[idx]);
            conic_opacity[idx] = co;
            rec0 = make_float4(pixx, pixy, co.x, co.y);
            rec1 = make_float4(co.z, co.w, vz, 0.f);
            my_tiles = (uint32_t)((y1 - y0) * (x1 - x0));
        } while (0);
        radii[idx] = my_radius_i;
        tiles_touched[idx] = my_tiles;
    }
    if (STAGED) {
        s_live[threadIdx.x] = shade;
        __syncthreads();
        stage_rows_in_256(shs, blockIdx.x * 256, P, 3 * M, s_live, s_rows);
        __syncthreads();
    }
    float col[3] = {0.f, 0.f, 0.f};
    if (shade) {
        // forward.cu:20-71
        float dx = px - cam_pos[0], dy = py - cam_pos[1], dz = pz - cam_pos[2];
        const float len = sqrtf(dx * dx + dy * dy + dz * dz);
        dx = dx / len; dy = dy / len; dz = dz / len;
        const float* sh = STAGED ? s_rows + threadIdx.x * staged_row_stride(3 * M) : shs + (size_t)idx * M * 3;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            float result = kSH_C0 * sh[ch];
            if (D > 0) {
                const float x = dx, y = dy, z = dz;
                result = result - kSH_C1 * y * sh[3 + ch] + kSH_C1 * z * sh[6 + ch] - kSH_C1 * x * sh[9 + ch];
                if (D > 1) {
                    const float xx = x * x, yy = y * y, zz = z * z;
                    const float xy = x * y, yz = y * z, xz = x * z;
                    result = result + kSH_C2[0] * xy * sh[12 + ch] + kSH_C2[1] * yz * sh[15 + ch] +
                             kSH_C2[2] * (2.0f * zz - xx - yy) * sh[18 + ch] + kSH_C2[3] * xz * sh[21 + ch] +
                             kSH_C2[4] * (xx - yy) * sh[24 + ch];
                    if (D > 2) {
                        result = result + kSH_C3[0] * y * (3.0f * xx - yy) * sh[27 + ch] +
                                 kSH_C3[1] * xy * z * sh[30 + ch] +
                                 kSH_C3[2] * y * (4.0f * zz - xx - yy) * sh[33 + ch] +
                                 kSH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[36 + ch] +
                                 kSH_C3[4] * x * (4.0f * zz - xx - yy) * sh[39 + ch] +
                                 kSH_C3[5] * z * (xx - yy) * sh[42 + ch] +
                                 kSH_C3[6] * x * (xx - 3.0f * yy) * sh[45 + ch];
                    }
                }
            }
            result += 0.5f;
            clamped[3 * idx + ch] = (result < 0);
            rgb[3 * idx + ch] = fmaxf(result, 0.0f);
            col[ch] = fmaxf(result, 0.0f);
        }
    }
    // the tile kernels stage one 48-byte record per instance from a 64-byte-aligned row instead of gathering xy, conic +
    // opacity, depth and colour from four arrays (four sub-line touches per instance; rocprofv3 showed render_forward
    // fetching 2.5x its algorithmic bytes)
    if (my_tiles != 0u) {
        if (colors_precomp != nullptr) {
            col[0] = colors_precomp[3 * idx]; col[1] = colors_precomp[3 * idx + 1]; col[2] = colors_precomp[3 * idx + 2];
        }
        float4* r = splat + 4 * (size_t)idx;
        r[0] = rec0;
        r[1] = rec1;
        r[2] = make_float4(col[0], col[1], col[2], 0.f);
    }
    // block reduction of tiles_touched -> block_sums
    __shared__ uint32_t s_wave[4];
    uint32_t ws = wave_sum_u32(my_tiles);
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = ws;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
}

// K3: exclusive scan of the per-block sums (one 1024-thread block; nb = ceil(P/256) is ~1.2k at P=300k).
// Writes block_sums[i] <- exclusive prefix and total[0] <- num_rendered (64-bit to detect u32 overflow).
__device__ __forceinline__ unsigned long long scan_block_sums_body(int nb, uint32_t* __restrict__ block_sums)
{
    __shared__ unsigned long long s_wave[16];
    __shared__ unsigned long long s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + tid;
        const uint32_t v = i < nb ? block_sums[i] : 0u;
        unsigned long long inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            unsigned long long n = __shfl_up(inc, o, 64);
            if (lane >= o) inc += n;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        unsigned long long wave_off = 0, chunk_total = 0;
        for (int w = 0; w < 16; w++) {
            if (w < wave) wave_off += s_wave[w];
            chunk_total += s_wave[w];
        }
        const unsigned long long carry = s_carry;
        if (i < nb) block_sums[i] = (uint32_t)(carry + wave_off + inc - v);
        __syncthreads();
        if (tid == 0) s_carry = carry + chunk_total;
        __syncthreads();
    }
    return s_carry;
}

__global__ void __launch_bounds__(1024) scan_block_sums_kernel(int nb, uint32_t* __restrict__ block_sums,
                                                               unsigned long long* __restrict__ total)
{
    const unsigned long long sum = scan_block_sums_body(nb, block_sums);
    if (threadIdx.x == 0) total[0] = sum;
}

// K5: reference duplicateWithKeys (rasterizer_impl.cu:70-111).  Also materialises the inclusive prefix sum
// point_offsets (K3's cub::DeviceScan::InclusiveSum output) from the block-local scan + scanned block sum.
// Emission order inside one Gaussian is row-major over its tile rectangle; key = tile_id<<32 | bits(depth).
// Rectangles larger than 32 tiles are expanded cooperatively by the whole wave so that one screen-filling
// Gaussian cannot serialise a wave (the reference loops per thread).
__global__ void __launch_bounds__(256)
duplicate_with_keys_kernel(int P, const float2* __restrict__ means2D, const float* __restrict__ depths,
                           const uint32_t* __restrict__ tiles_touched, const uint32_t* __restrict__ block_offsets,
                           uint32_t* __restrict__ point_offsets, uint64_t* __restrict__ keys,
                           uint32_t* __restrict__ values, const int* __restrict__ radii, int gx, int gy)
{
    __shared__ uint32_t s_wave[4];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t cnt = idx < P ? tiles_touched[idx] : 0u;
    const uint32_t inc = wave_inclusive_scan_u32(cnt);
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t off = block_offsets[blockIdx.x];
    for (int w = 0; w < wave; w++) off += s_wave[w];
    const uint32_t end = off + inc;      // inclusive prefix == reference point_offsets[idx]
    uint32_t begin = end - cnt;
    if (idx < P) point_offsets[idx] = end;

    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    uint32_t dbits = 0;
    const bool live = idx < P && radii[idx] > 0;
    if (live) {
        const float2 p = means2D[idx];
        tile_rect(p.x, p.y, radii[idx], gx, gy, x0, y0, x1, y1);
        dbits = __float_as_uint(depths[idx]);
    }
    const int w_rect = x1 - x0;
    const bool big = live && cnt > 32u;
    if (live && !big) {
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) {
                uint64_t key = (uint64_t)(uint32_t)(y * gx + x);
                key = (key << 32) | dbits;
                keys[begin] = key;
                values[begin] = (uint32_t)idx;
                begin++;
            }
    }
    // wave-cooperative expansion of the big rectangles
    unsigned long long todo = __ballot(big);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int bx0 = __shfl(x0, src, 64), by0 = __shfl(y0, src, 64), bw = __shfl(w_rect, src, 64);
        const uint32_t bcnt = (uint32_t)__shfl((int)cnt, src, 64);
        const uint32_t bbegin = (uint32_t)__shfl((int)(end - cnt), src, 64);
        const uint32_t bbits = (uint32_t)__shfl((int)dbits, src, 64);
        const uint32_t bid = (uint32_t)(blockIdx.x * 256 + wave * 64 + src);
        for (uint32_t k = lane; k < bcnt; k += 64) {
            const int y = by0 + (int)(k / (uint32_t)bw), x = bx0 + (int)(k % (uint32_t)bw);
            uint64_t key = (uint64_t)(uint32_t)(y * gx + x);
            key = (key << 32) | bbits;
            keys[bbegin + k] = key;
            values[bbegin + k] = bid;
        }
    }
}

// K7: reference identifyTileRanges (rasterizer_impl.cu:116-138); ranges pre-zeroed by hipMemsetAsync (:320).
__global__ void identify_tile_ranges_kernel(int L, const uint64_t* __restrict__ keys, uint2* __restrict__ ranges)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= L) return;
    const uint32_t currtile = (uint32_t)(keys[idx] >> 32);
    if (idx == 0)
        ranges[currtile].x = 0;
    else {
        const uint32_t prevtile = (uint32_t)(keys[idx - 1] >> 32);
        if (currtile != prevtile) {
            ranges[prevtile].y = idx;
            ranges[currtile].x = idx;
        }
    }
    if (idx == L - 1) ranges[currtile].y = L;
}

// Longest-tile-first block order for the two tile kernels: one block buckets the T tile lengths into 256 classes
// (descending) with LDS counters.  Order inside a class is arbitrary -- it only affects scheduling, never results.
__device__ __forceinline__ void tile_order_body(int T, const uint2* __restrict__ ranges, uint32_t* __restrict__ order,
                                                uint32_t small_cap, uint32_t* __restrict__ big_list,
                                                uint32_t* __restrict__ big_count)
{
    __shared__ uint32_t s_nbig;
    __shared__ uint32_t s_cnt[256];
    __shared__ uint32_t s_max;
    const int tid = threadIdx.x;
    if (tid < 256) s_cnt[tid] = 0;
    if (tid == 0) { s_max = 0; s_nbig = 0; }
    __syncthreads();
    uint32_t m = 0;
    for (int t = tid; t < T; t += 1024) {
        const uint32_t len = ranges[t].y - ranges[t].x;
        m = max(m, len);
        // tiles too long for the small in-LDS depth sort (tile-binned ordering, radix_sort.hip)
        if (big_list != nullptr && len > small_cap) big_list[atomicAdd(&s_nbig, 1u)] = (uint32_t)t;
    }
    atomicMax(&s_max, m);
    __syncthreads();
    const uint32_t scale = s_max + 1;
    for (int t = tid; t < T; t += 1024) {
        const uint32_t len = ranges[t].y - ranges[t].x;
        const uint32_t cls = 255u - (uint32_t)(((uint64_t)len * 256u) / scale);
        atomicAdd(&s_cnt[cls], 1u);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int c = 0; c < 256; c++) {
            const uint32_t n = s_cnt[c];
            s_cnt[c] = run;
            run += n;
        }
    }
    __syncthreads();
    for (int t = tid; t < T; t += 1024) {
        const uint32_t len = ranges[t].y - ranges[t].x;
        const uint32_t cls = 255u - (uint32_t)(((uint64_t)len * 256u) / scale);
        order[atomicAdd(&s_cnt[cls], 1u)] = (uint32_t)t;
    }
    if (tid == 0 && big_count != nullptr) {
        big_count[0] = s_nbig;
        big_count[1] = 0;               // the long-tile sort's work cursor (radix_sort.hip)
    }
}

__global__ void __launch_bounds__(1024)
tile_order_kernel(int T, const uint2* __restrict__ ranges, uint32_t* __restrict__ order, uint32_t small_cap,
                  uint32_t* __restrict__ big_list, uint32_t* __restrict__ big_count)
{
    tile_order_body(T, ranges, order, small_cap, big_list, big_count);
}

// ---- direct tile binning (R3DG_OPT_TILE_BINNING = 2, default) -----------------------------------------------------------------
// The reference emits (tile | depth, index) pairs in Gaussian order and sorts them globally; the round-1 formulation emitted
// them the same way, then histogrammed and partitioned them by tile id (duplicate -> hist -> scan -> scatter: the pairs are
// written, read, read, written again before any tile sort sees them -- 0.17 ms inside the iteration).  Here the instances
// go straight into their tile's segment:
//   tile_count_kernel   per block of 256 Gaussians an LDS histogram of the tiles their rectangles cover, flushed with one
//                       global atomic per (block, touched tile);
//   tile_scan_kernel    exclusive scan of the T tile counts = the tile RANGES (identifyTileRanges' result) + the cursors;
//   tile_emit_kernel    the same LDS histogram again, ONE global atomic per (block, touched tile) reserves a run inside the
//                       tile's segment, then every instance takes its slot with an LDS atomic and is written ONCE, as the
//                       sort entry (depth bits << 32 | Gaussian index) the per-tile sort wants.
// Order inside a tile is arbitrary here; the per-tile sort by the unique (depth, index) key makes the final lists
// bit-identical to the reference's stable global sort.  Falls back to the round-1 path when T exceeds the LDS histogram.
constexpr int BIN_MAX_TILES = 16384;           // 64 KB of LDS counters (+ the 8 KB list of large rectangles behind them)

constexpr int BIN_THREADS = 1024;
constexpr int BIN_MAX_ITERS = 4;            // Gaussians per block = R3DG_OPT_BINNING_BLOCK_K (1..4) x 1024

// The tile rectangle of Gaussian `idx`, loaded ONCE per kernel and kept in registers for every pass over it (round 5: each pass
// used to read radii[idx], wait, then means2D[idx] under `if (live)`, wait again -- two dependent round trips x Gaussians per
// thread x passes, in kernels whose VALU is busy 9 % of the time; tools/isa_waits.py).  Both loads are unconditional (clamped
// index) so that they leave together.
struct TileRect {
    int x0, y0, w, h;          // w * h == 0: culled / past the end
};

__device__ __forceinline__ TileRect load_tile_rect(int idx, int P, const float2* __restrict__ means2D,
                                                   const int* __restrict__ radii, int gx, int gy)
{
    const int i = idx < P ? idx : P - 1;
    const int r = radii[i];
    const float2 p = means2D[i];
    TileRect t;
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    if (idx < P && r > 0) tile_rect(p.x, p.y, r, gx, gy, x0, y0, x1, y1);
    t.x0 = x0; t.y0 = y0; t.w = x1 - x0; t.h = y1 - y0;
    return t;
}

// WHICH Gaussians a block takes (round 6).  Rounds 2-5: block b took the 2048 CONSECUTIVE Gaussians b * 2048 ... -- fine for the
// synthetic scene, whose index order is random in space and in size.  A scene that was DENSIFIED keeps its oldest Gaussians at the
// lowest indices, and the oldest are the largest: on a scene trained here from 4 000 points (trained_scene.py: 380k rows, 4.9 M
// instances) the first block held 1.0 M of the instances, one CU expanded them while 185 others had finished theirs, and the count
// + emit kernels took 0.90 ms (the i.i.d. scene: 0.07).  Now the 64-Gaussian wave chunks are DEALT round robin over the blocks
// (chunk q -> block q mod nb): loads stay coalesced per wave and any index-clustered tail is spread over all blocks (heaviest
// block of that scene: 62 k instances, mean 27 k).
__device__ __forceinline__ int dealt_index_of(int it, int thread, int nb)
{
    const int q = (it * (BIN_THREADS / 64) + (thread >> 6)) * nb + (int)blockIdx.x;
    return q * 64 + (thread & 63);
}
__device__ __forceinline__ int dealt_index(int it, int nb) { return dealt_index_of(it, (int)threadIdx.x, nb); }

// The tiles of the block's Gaussians: f(tile, Gaussian, that Gaussian's payload).
//   * rectangles of up to 32 tiles: one per lane, ONE flat loop (as a y / x loop nest the wave ran max-over-lanes(h) x
//     max-over-lanes(w) rounds -- up to 32 x 32 when one lane holds a 2 x 16 and another a 16 x 2 rectangle);
//   * larger rectangles go on the BLOCK's list (s_big: pass << 10 | thread, 16 bits) and, behind a barrier, are expanded by whole
//     waves -- every wave of the block takes every 16th entry, whichever wave loaded it: rounds 2-5 expanded them inside the wave
//     that owned them, one after the other (a wave of 64 screen-filling Gaussians: 39 k instances on one wave of the scene above).
//     The rectangle (and the payload) of a listed Gaussian is derived again from its index: two broadcast loads per >= 33 tiles.
// Every thread of the block must call both halves (they synchronise).
template <typename F>
__device__ __forceinline__ void small_rect_tiles(const TileRect& rc, int it, int idx, uint32_t payload, int gx, uint16_t* s_big,
                                                 uint32_t* s_nbig, F&& f)
{
    const uint32_t cnt = (uint32_t)(rc.w * rc.h);
    const bool big = cnt > 32u;
    if (cnt != 0u && !big) {
        int x = rc.x0, y = rc.y0;
        const int x1 = rc.x0 + rc.w;
        for (uint32_t k = 0; k < cnt; k++) {
            f((uint32_t)(y * gx + x), (uint32_t)idx, payload);
            if (++x == x1) {
                x = rc.x0;
                y++;
            }
        }
    }
    const unsigned long long m = __ballot(big);
    if (m) {
        const int lane = threadIdx.x & 63;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(s_nbig, (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, 0, 64);
        if (big) s_big[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)((it << 10) | (int)threadIdx.x);
    }
}

template <bool PAYLOAD_IS_DEPTH, typename F>
__device__ __forceinline__ void big_rect_tiles(const uint16_t* s_big, uint32_t n_big, int nb, const float2* __restrict__ means2D,
                                               const int* __restrict__ radii, const float* __restrict__ depths, int gx, int gy, F&& f)
{
    const int lane = threadIdx.x & 63;
    for (uint32_t r = threadIdx.x >> 6; r < n_big; r += BIN_THREADS / 64) {
        const int e = (int)s_big[r];
        const uint32_t g = (uint32_t)dealt_index_of(e >> 10, e & 1023, nb);
        const float2 p = means2D[g];
        int x0, y0, x1, y1;
        tile_rect(p.x, p.y, radii[g], gx, gy, x0, y0, x1, y1);
        const uint32_t pay = PAYLOAD_IS_DEPTH ? __float_as_uint(depths[g]) : 0u;
        const int w = x1 - x0;
        const uint32_t cnt = (uint32_t)(w * (y1 - y0));
        // lane k of round j takes tile 64 j + k of the rectangle: (x, y) advance by 64 tiles per round without a division
        int y = y0 + lane / w, x = x0 + lane % w;
        const int dy = 64 / w, dx = 64 % w;
        for (uint32_t k = lane; k < cnt; k += 64) {
            f((uint32_t)(y * gx + x), g, pay);
            x += dx;
            y += dy;
            if (x >= x1) {
                x -= w;
                y++;
            }
        }
    }
}

// each block takes `iters` x 1024 Gaussians (dealt_index): the more Gaussians share one LDS histogram, the fewer global
// atomics (one per block and touched tile; an unordered cloud touches nearly every tile from every block)
__global__ void __launch_bounds__(BIN_THREADS)
tile_count_kernel(int P, int T, int iters, const float2* __restrict__ means2D, const int* __restrict__ radii, int gx, int gy,
                  uint32_t* __restrict__ tile_counts)
{
    extern __shared__ uint32_t s_bins[];                       // T counters, then the list of large rectangles
    uint16_t* s_big = reinterpret_cast<uint16_t*>(s_bins + T);
    __shared__ uint32_t s_nbig;
    const int nb = gridDim.x;
    TileRect rc[BIN_MAX_ITERS];
#pragma unroll
    for (int it = 0; it < BIN_MAX_ITERS; it++)            // (the loads of every pass leave before the zero fill is waited for)
        rc[it] = load_tile_rect(it < iters ? dealt_index(it, nb) : P, P, means2D, radii, gx, gy);
    for (int t = threadIdx.x; t < T; t += BIN_THREADS) s_bins[t] = 0;
    if (threadIdx.x == 0) s_nbig = 0;
    __syncthreads();
    auto count = [&](uint32_t tile, uint32_t, uint32_t) { atomicAdd(&s_bins[tile], 1u); };
#pragma unroll
    for (int it = 0; it < BIN_MAX_ITERS; it++)
        if (it < iters) small_rect_tiles(rc[it], it, dealt_index(it, nb), 0u, gx, s_big, &s_nbig, count);
    __syncthreads();
    big_rect_tiles<false>(s_big, s_nbig, nb, means2D, radii, nullptr, gx, gy, count);
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += BIN_THREADS) {
        const uint32_t c = s_bins[t];
        if (c) atomicAdd(&tile_counts[t], c);
    }
}

// one 1024-thread block: ranges[t] = (start, end), cursor[t] = start.  Thread i owns a CONTIGUOUS run of the tile counts (and of
// the block sums): every load is issued before the one barrier and the second pass re-reads lines this block just pulled in,
// so the kernel costs two memory round trips instead of one per 1024 items and scan (17.6 -> ~6 us on the ordering chain).
__global__ void __launch_bounds__(1024)
tile_scan_kernel(int T, const uint32_t* __restrict__ tile_counts, uint2* __restrict__ ranges, uint32_t* __restrict__ cursor,
                 unsigned long long* __restrict__ total, long long capacity, float* __restrict__ overflow_flag,
                 unsigned int* __restrict__ overflow_count, int nb_block_sums, uint32_t* __restrict__ block_sums)
{
    __shared__ uint32_t s_wave[16];
    __shared__ unsigned long long s_wave_b[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per_t = (T + 1023) / 1024, t0 = tid * per_t, t1 = min(t0 + per_t, T);
    const int per_b = (nb_block_sums + 1023) / 1024, b0 = tid * per_b, b1 = min(b0 + per_b, nb_block_sums);
    uint32_t tsum = 0;
    for (int t = t0; t < t1; t++) tsum += tile_counts[t];
    // nb_block_sums > 0: the projection left its per-block instance counts unscanned (launch_preprocess with scan_now = false):
    // their exclusive scan and the total are produced here, one launch earlier than the kernels that read them (tile_emit_kernel)
    unsigned long long bsum = 0;
    for (int b = b0; b < b1; b++) bsum += block_sums[b];
    const uint32_t tinc = wave_inclusive_scan_u32(tsum);
    unsigned long long binc = bsum;
    if (nb_block_sums > 0) {
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long n = __shfl_up(binc, o, 64);
            if (lane >= o) binc += n;
        }
    }
    if (lane == 63) {
        s_wave[wave] = tinc;
        s_wave_b[wave] = binc;
    }
    __syncthreads();
    uint32_t toff = 0;
    unsigned long long boff = 0, count = 0;
    for (int w = 0; w < 16; w++) {
        if (w < wave) {
            toff += s_wave[w];
            boff += s_wave_b[w];
        }
        count += s_wave_b[w];
    }
    if (nb_block_sums > 0) {
        if (tid == 0) total[0] = count;
        unsigned long long run = boff + binc - bsum;
        for (int b = b0; b < b1; b++) {
            const uint32_t v = block_sums[b];
            block_sums[b] = (uint32_t)run;
            run += v;
        }
    } else {
        count = *total;
    }
    // bounded forward (capacity >= 0): the binning state holds `capacity` instance slots and the host did NOT look at the
    // count; a frame that needs more is dropped on the device -- every tile list empty, *overflow_flag = 1 for the caller
    const bool over = capacity >= 0 && count > (unsigned long long)capacity;
    if (tid == 0 && overflow_flag != nullptr) *overflow_flag = over ? 1.0f : 0.0f;
    if (tid == 0 && over && overflow_count != nullptr) *overflow_count += 1u;       // running count, never reset here
    uint32_t start = toff + tinc - tsum;
    for (int t = t0; t < t1; t++) {
        const uint32_t v = tile_counts[t];
        // an empty tile keeps (0,0) like the reference's zero-initialised ranges (rasterizer_impl.cu:320)
        ranges[t] = (v && !over) ? make_uint2(start, start + v) : make_uint2(0u, 0u);
        cursor[t] = over ? 0u : start;
        start += v;
    }
}

__global__ void __launch_bounds__(BIN_THREADS)
tile_emit_kernel(int P, int T, int iters, const float2* __restrict__ means2D, const float* __restrict__ depths,
                 const int* __restrict__ radii, const uint32_t* __restrict__ tiles_touched,
                 const uint32_t* __restrict__ block_offsets, int gx, int gy, uint32_t* __restrict__ cursor,
                 uint32_t* __restrict__ point_offsets, uint64_t* __restrict__ entries,
                 const unsigned long long* __restrict__ total, long long capacity, int emit_blocks,
                 const uint2* __restrict__ ranges, uint32_t* __restrict__ order, uint32_t small_cap,
                 uint32_t* __restrict__ big_list, uint32_t* __restrict__ big_count)
{
    extern __shared__ uint32_t s_bins[];                       // T counters, then the list of large rectangles
    uint16_t* s_big = reinterpret_cast<uint16_t*>(s_bins + T);
    __shared__ uint32_t s_wave[BIN_MAX_ITERS][BIN_THREADS / 64];
    __shared__ uint32_t s_nbig;
    // one block past the emitting ones (when the caller asked for it): the longest-tile-first order of the tile kernels, which
    // needs the ranges only -- beside the emission instead of a launch of its own behind it
    if ((int)blockIdx.x >= emit_blocks) {
        tile_order_body(T, ranges, order, small_cap, big_list, big_count);
        return;
    }
    const bool over = capacity >= 0 && *total > (unsigned long long)capacity;      // see tile_scan_kernel
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = threadIdx.x; t < T; t += BIN_THREADS) s_bins[t] = 0;
    if (threadIdx.x == 0) s_nbig = 0;
    // GeometryState::point_offsets (inclusive scan of tiles_touched, rasterizer_impl.cu:283-287): part of the state parity.
    // block_offsets are preprocess' exclusive sums per 256 Gaussians = 4 waves -- this part keeps the CONSECUTIVE mapping of
    // Gaussians to blocks (the four waves of a 256-chunk share s_wave); the rectangles below are those of the DEALT Gaussians.
    // All loads first, one barrier for all passes.
    const int lin = blockIdx.x * iters * BIN_THREADS + threadIdx.x;
    uint32_t inc[BIN_MAX_ITERS], boff[BIN_MAX_ITERS], dbits[BIN_MAX_ITERS];
    TileRect rc[BIN_MAX_ITERS];
#pragma unroll
    for (int it = 0; it < BIN_MAX_ITERS; it++) {
        const int idx = it < iters ? dealt_index(it, emit_blocks) : P;
        rc[it] = load_tile_rect(idx, P, means2D, radii, gx, gy);
        dbits[it] = __float_as_uint(depths[idx < P ? idx : P - 1]);
    }
#pragma unroll
    for (int it = 0; it < BIN_MAX_ITERS; it++) {
        const int idx = lin + it * BIN_THREADS;
        const bool in = it < iters && idx < P;
        const uint32_t cnt = in ? tiles_touched[idx] : 0u;
        boff[it] = in ? block_offsets[idx >> 8] : 0u;
        inc[it] = wave_inclusive_scan_u32(cnt);
        if (lane == 63) s_wave[it][wave] = inc[it];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < BIN_MAX_ITERS; it++) {
        const int idx = lin + it * BIN_THREADS;
        if (it < iters && idx < P) {
            uint32_t off = boff[it];
            for (int w = wave & ~3; w < wave; w++) off += s_wave[it][w];
            point_offsets[idx] = off + inc[it];
        }
    }
    if (over) return;
    auto count = [&](uint32_t tile, uint32_t, uint32_t) { atomicAdd(&s_bins[tile], 1u); };
#pragma unroll
    for (int it = 0; it < BIN_MAX_ITERS; it++)
        if (it < iters) small_rect_tiles(rc[it], it, dealt_index(it, emit_blocks), 0u, gx, s_big, &s_nbig, count);
    __syncthreads();
    const uint32_t n_big = s_nbig;
    big_rect_tiles<false>(s_big, n_big, emit_blocks, means2D, radii, depths, gx, gy, count);
    __syncthreads();
    // this block's run inside every touched tile's segment: ONE returning global atomic per (block, tile) -- four of a thread's in
    // flight at a time (as a plain loop each waited for its predecessor's return, a device-scope round trip of several us: half
    // of the block's lifetime at 2500 tiles)
    for (int t0 = threadIdx.x; t0 < T; t0 += 4 * BIN_THREADS) {
        uint32_t c[4], r[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int t = t0 + k * BIN_THREADS;
            c[k] = t < T ? s_bins[t] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) r[k] = c[k] ? atomicAdd(&cursor[t0 + k * BIN_THREADS], c[k]) : 0u;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (c[k]) s_bins[t0 + k * BIN_THREADS] = r[k];
    }
    __syncthreads();
    auto emit = [&](uint32_t tile, uint32_t g, uint32_t depth_bits) {
        const uint32_t pos = atomicAdd(&s_bins[tile], 1u);
        entries[pos] = ((uint64_t)depth_bits << 32) | (uint64_t)g;
    };
    // (the list is kept from the counting pass: the second pass over the small rectangles does not append again)
#pragma unroll
    for (int it = 0; it < BIN_MAX_ITERS; it++)
        if (it < iters) {
            const TileRect& q = rc[it];
            const uint32_t cnt = (uint32_t)(q.w * q.h);
            if (cnt != 0u && cnt <= 32u) {
                int x = q.x0, y = q.y0;
                const int x1 = q.x0 + q.w;
                const uint32_t g = (uint32_t)dealt_index(it, emit_blocks);
                for (uint32_t k = 0; k < cnt; k++) {
                    emit((uint32_t)(y * gx + x), g, dbits[it]);
                    if (++x == x1) {
                        x = q.x0;
                        y++;
                    }
                }
            }
        }
    big_rect_tiles<true>(s_big, n_big, emit_blocks, means2D, radii, depths, gx, gy, emit);
}

int g_bin_iters = 2;       // R3DG_OPT_BINNING_BLOCK_K (measured at 2M Gaussians too: 2 / 3 / 4 -> 163 / 156 / 161 it/s, no trend)

// `fused` (the bounded forward): tile_counts arrive zeroed (launch_preprocess zero_words), block_sums arrive UNSCANNED
// (launch_preprocess scan_now = false: the scan kernel scans them and writes *total), and the tile order (launch_tile_order's
// outputs) is produced by one extra block of the emit kernel -- three launches fewer on the ordering stream's chain.
void launch_tile_binning(hipStream_t s, int P, int T, const float* means2D, const float* depths, const int* radii,
                         const uint32_t* tiles_touched, uint32_t* block_offsets, int gx, int gy,
                         uint32_t* tile_counts /* T, scratch */, uint32_t* cursor /* T, scratch */, uint32_t* ranges,
                         uint32_t* point_offsets, uint64_t* entries, unsigned long long* total, long long capacity,
                         float* overflow_flag, unsigned int* overflow_count, bool fused, uint32_t* order, uint32_t small_cap,
                         uint32_t* big_list, uint32_t* big_count)
{
    if (!fused) R3DG_HIP(hipMemsetAsync(tile_counts, 0, (size_t)T * 4, s));
    const int iters = std::min(std::max(opt(R3DG_OPT_BINNING_BLOCK_K), 1), BIN_MAX_ITERS);
    const int per_block = iters * BIN_THREADS;
    const int nb = (P + per_block - 1) / per_block;
    const size_t smem = (size_t)T * 4 + (size_t)iters * BIN_THREADS * 2;       // tile counters + the list of large rectangles
    // above 64 KB (more than ~14 000 tiles, e.g. 2048 x 2048) the kernels need the per-device function attribute, asked for once
    if (smem + 1024 > 65536) {
        static std::atomic<unsigned long long> done{0};
        int dev = 0;
        R3DG_HIP(hipGetDevice(&dev));
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(done.load(std::memory_order_acquire) & bit)) {
            const int want = BIN_MAX_TILES * 4 + BIN_MAX_ITERS * BIN_THREADS * 2;
            R3DG_HIP(hipFuncSetAttribute((const void*)tile_count_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, want));
            R3DG_HIP(hipFuncSetAttribute((const void*)tile_emit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, want));
            done.fetch_or(bit, std::memory_order_release);
        }
    }
    tile_count_kernel<<<nb, BIN_THREADS, smem, s>>>(P, T, iters, (const float2*)means2D, radii, gx, gy, tile_counts);
    tile_scan_kernel<<<1, 1024, 0, s>>>(T, tile_counts, (uint2*)ranges, cursor, total, capacity, overflow_flag,
                                        overflow_count, fused ? (P + 255) / 256 : 0, block_offsets);
    tile_emit_kernel<<<nb + (fused ? 1 : 0), BIN_THREADS, smem, s>>>(
        P, T, iters, (const float2*)means2D, depths, radii, tiles_touched, block_offsets, gx, gy, cursor, point_offsets, entries,
        total, capacity, nb, (const uint2*)ranges, order, small_cap, big_list, big_count);
}

int tile_binning_max_tiles() { return BIN_MAX_TILES; }

// ---- host launchers -------------------------------------------------------------------------------------
void launch_tile_order(hipStream_t s, int T, const uint32_t* ranges, uint32_t* order, uint32_t small_cap,
                       uint32_t* big_list, uint32_t* big_count)
{
    tile_order_kernel<<<1, 1024, 0, s>>>(T, (const uint2*)ranges, order, small_cap, big_list, big_count);
}

void launch_mark_visible(hipStream_t s, int P, const float* means3D, const float* vm, uint8_t* present)
{
    if (P <= 0) return;
    mark_visible_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, means3D, vm, present);
}

extern int g_stage_sh_rows;     // rasterizer_preprocess_bwd.hip (R3DG_OPT_STAGE_SH_ROWS)

void launch_preprocess(hipStream_t s, int P, int D, int M, const float* means3D, const float* scales,
                       float scale_modifier, const float* rotations, const float* opacities, const float* shs,
                       uint8_t* clamped, const float* cov3D_precomp, const float* colors_precomp, const float* vm,
                       const float* pm, const float* cam_pos, int W, int H, float tan_fovx, float tan_fovy,
                       float focal_x, float focal_y, int* radii, float* means2D, float* depths, float* cov3Ds,
                       float* rgb, float* conic_opacity, float* splat, int gx, int gy, uint32_t* tiles_touched,
                       uint32_t* block_sums, unsigned long long* total, bool scan_now, uint32_t* zero_words, int zero_n)
{
    const int nb = (P + 255) / 256;
    if (zero_n > nb * 256) {                   // (fewer threads than words: a memset after all)
        R3DG_HIP(hipMemsetAsync(zero_words, 0, (size_t)zero_n * 4, s));
        zero_n = 0;
    }
    const bool staged = opt(R3DG_OPT_STAGE_SH_ROWS) && shs != nullptr && colors_precomp == nullptr && M >= 1 && M <= 16;
    if (staged)
        preprocess_kernel<true><<<nb, 256, 256 * ((3 * M) | 1) * sizeof(float), s>>>(
            P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs, clamped, cov3D_precomp, colors_precomp,
            vm, pm, cam_pos, W, H, tan_fovx, tan_fovy, focal_x, focal_y, radii, (float2*)means2D, depths, cov3Ds, rgb,
            (float4*)conic_opacity, (float4*)splat, gx, gy, tiles_touched, block_sums, zero_words, zero_n);
    else
        preprocess_kernel<false><<<nb, 256, 0, s>>>(
            P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs, clamped, cov3D_precomp, colors_precomp,
            vm, pm, cam_pos, W, H, tan_fovx, tan_fovy, focal_x, focal_y, radii, (float2*)means2D, depths, cov3Ds, rgb,
            (float4*)conic_opacity, (float4*)splat, gx, gy, tiles_touched, block_sums, zero_words, zero_n);
    if (scan_now) scan_block_sums_kernel<<<1, 1024, 0, s>>>(nb, block_sums, total);
}

void launch_duplicate_with_keys(hipStream_t s, int P, const float* means2D, const float* depths,
                                const uint32_t* tiles_touched, const uint32_t* block_offsets,
                                uint32_t* point_offsets, uint64_t* keys, uint32_t* values, const int* radii, int gx,
                                int gy)
{
    duplicate_with_keys_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, (const float2*)means2D, depths, tiles_touched,
                                                             block_offsets, point_offsets, keys, values, radii, gx, gy);
}

void launch_identify_tile_ranges(hipStream_t s, int L, const uint64_t* keys, uint32_t* ranges)
{
    if (L <= 0) return;
    identify_tile_ranges_kernel<<<(L + 255) / 256, 256, 0, s>>>(L, keys, (uint2*)ranges);
}

}  // namespace r3dg
