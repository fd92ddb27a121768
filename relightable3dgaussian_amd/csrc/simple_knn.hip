// distCUDA2 for gfx950: mean squared distance of every point to its 3 nearest neighbours.
// Reference semantics: SimpleKNN::knn, submodules/simple-knn/simple_knn.cu:185-221 (Morton sort, 1024-point boxes,
// box-distance pruning; bounds reduced with the reference's (0,0,0) initial value, :191-199).  The result is the exact
// 3-NN mean, so it does not depend on the traversal order; compiled with -ffp-contract=off so the fp32 distances equal
// the CPU oracle's bit for bit.  The Morton sort is this library's own stable radix sort (3 passes of 10 bits).
#include "common.hpp"
#include <cfloat>

namespace r3dg {

constexpr int KNN_BOX = 1024;

struct KnnBox {
    float lo[3], hi[3];
};

__global__ void __launch_bounds__(256) knn_bounds_partial_kernel(int P, const float* __restrict__ pts, float* __restrict__ partial)
{
    __shared__ float s[4][6];
    float b[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};          // the reference's reduction starts from (0,0,0)
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256)
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float v = pts[3 * (size_t)i + a];
            b[a] = fminf(b[a], v);
            b[3 + a] = fmaxf(b[3 + a], v);
        }
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float n = __shfl_xor(b[a], o, 64);
            b[a] = a < 3 ? fminf(b[a], n) : fmaxf(b[a], n);
        }
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int a = 0; a < 6; a++) s[threadIdx.x >> 6][a] = b[a];
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        float v = s[0][a];
        for (int w = 1; w < 4; w++) v = a < 3 ? fminf(v, s[w][a]) : fmaxf(v, s[w][a]);
        partial[blockIdx.x * 6 + a] = v;
    }
}
__global__ void knn_bounds_final_kernel(int nb, const float* __restrict__ partial, float* __restrict__ bounds)
{
    const int a = threadIdx.x;
    if (a >= 6) return;
    float v = 0.f;
    for (int i = 0; i < nb; i++) v = a < 3 ? fminf(v, partial[i * 6 + a]) : fmaxf(v, partial[i * 6 + a]);
    bounds[a] = v;
}

__device__ __forceinline__ uint32_t prep_morton(uint32_t x)
{
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}

__global__ void __launch_bounds__(256)
knn_morton_kernel(int P, const float* __restrict__ pts, const float* __restrict__ bounds, uint64_t* __restrict__ keys,
                  uint32_t* __restrict__ vals)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    uint32_t c[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float f = ((pts[3 * (size_t)i + a] - bounds[a]) / (bounds[3 + a] - bounds[a])) * (float)((1 << 10) - 1);
        c[a] = prep_morton((uint32_t)f);
    }
    keys[i] = (uint64_t)(c[0] | (c[1] << 1) | (c[2] << 2));
    vals[i] = (uint32_t)i;
}

__global__ void __launch_bounds__(KNN_BOX)
knn_box_minmax_kernel(uint32_t P, const float* __restrict__ pts, const uint32_t* __restrict__ indices, KnnBox* __restrict__ boxes)
{
    __shared__ float s[KNN_BOX / 64][6];
    const uint32_t idx = blockIdx.x * KNN_BOX + threadIdx.x;
    float b[6] = {FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (idx < P) {
        const uint32_t j = indices[idx];
#pragma unroll
        for (int a = 0; a < 3; a++) b[a] = b[3 + a] = pts[3 * (size_t)j + a];
    }
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float n = __shfl_xor(b[a], o, 64);
            b[a] = a < 3 ? fminf(b[a], n) : fmaxf(b[a], n);
        }
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int a = 0; a < 6; a++) s[threadIdx.x >> 6][a] = b[a];
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        float v = s[0][a];
        for (int w = 1; w < KNN_BOX / 64; w++) v = a < 3 ? fminf(v, s[w][a]) : fmaxf(v, s[w][a]);
        if (a < 3) boxes[blockIdx.x].lo[a] = v; else boxes[blockIdx.x].hi[a - 3] = v;
    }
}

__device__ __forceinline__ void update3(const float p[3], const float* __restrict__ q, float (&best)[3])
{
    const float dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
    float dist = dx * dx + dy * dy + dz * dz;
#pragma unroll
    for (int j = 0; j < 3; j++)
        if (best[j] > dist) { const float t = best[j]; best[j] = dist; dist = t; }
}

__global__ void __launch_bounds__(256)
knn_mean_dist_kernel(int P, const float* __restrict__ pts, const uint32_t* __restrict__ indices,
                     const KnnBox* __restrict__ boxes, float* __restrict__ dists)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const uint32_t me = indices[idx];
    const float p[3] = {pts[3 * (size_t)me], pts[3 * (size_t)me + 1], pts[3 * (size_t)me + 2]};
    float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    for (int i = max(0, idx - 3); i <= min(P - 1, idx + 3); i++) {
        if (i == idx) continue;
        update3(p, pts + 3 * (size_t)indices[i], best);
    }
    const float reject = best[2];
    best[0] = best[1] = best[2] = FLT_MAX;
    const int nb = (P + KNN_BOX - 1) / KNN_BOX;
    for (int b = 0; b < nb; b++) {
        const KnnBox box = boxes[b];
        float d2 = 0.f;
#pragma unroll
        for (int a = 0; a < 3; a++)
            if (p[a] < box.lo[a] || p[a] > box.hi[a]) {
                const float d = fminf(fabsf(p[a] - box.lo[a]), fabsf(p[a] - box.hi[a]));
                d2 += d * d;
            }
        if (d2 > reject || d2 > best[2]) continue;
        for (int i = b * KNN_BOX; i < min(P, (b + 1) * KNN_BOX); i++) {
            if (i == idx) continue;
            update3(p, pts + 3 * (size_t)indices[i], best);
        }
    }
    dists[me] = (best[0] + best[1] + best[2]) / 3.0f;
}

size_t knn_temp_bytes(size_t P)
{
    size_t o = 0;
    auto take = [&](size_t b) { o = align_up(o + b, 256); };
    take(P * 8); take(P * 8); take(P * 4); take(P * 4);
    take(((P + KNN_BOX - 1) / KNN_BOX + 1) * sizeof(KnnBox));
    take(1024 * 6 * 4); take(256);
    take(sort_temp_bytes(P));
    return o + 256;
}

void knn_dist2(hipStream_t s, int P, const float* pts, float* dists, void* temp)
{
    char* base = (char*)temp;
    size_t o = 0;
    auto take = [&](size_t b) { char* p = base + o; o = align_up(o + b, 256); return p; };
    uint64_t* k_in = (uint64_t*)take((size_t)P * 8);
    uint64_t* k_out = (uint64_t*)take((size_t)P * 8);
    uint32_t* v_in = (uint32_t*)take((size_t)P * 4);
    uint32_t* v_out = (uint32_t*)take((size_t)P * 4);
    KnnBox* boxes = (KnnBox*)take(((size_t)(P + KNN_BOX - 1) / KNN_BOX + 1) * sizeof(KnnBox));
    float* partial = (float*)take(1024 * 6 * 4);
    float* bounds = (float*)take(256);
    void* sort_temp = (void*)take(sort_temp_bytes((size_t)P));
    const int nb = min(1024, (P + 255) / 256);
    knn_bounds_partial_kernel<<<nb, 256, 0, s>>>(P, pts, partial);
    knn_bounds_final_kernel<<<1, 64, 0, s>>>(nb, partial, bounds);
    const int g = (P + 255) / 256;
    knn_morton_kernel<<<g, 256, 0, s>>>(P, pts, bounds, k_in, v_in);
    check_launch(s, false, "knn morton");
    sort_pairs(s, (size_t)P, k_in, v_in, k_out, v_out, 30, sort_temp, false);
    const int nboxes = (P + KNN_BOX - 1) / KNN_BOX;
    knn_box_minmax_kernel<<<nboxes, KNN_BOX, 0, s>>>((uint32_t)P, pts, v_out, boxes);
    knn_mean_dist_kernel<<<g, 256, 0, s>>>(P, pts, v_out, boxes, dists);
    check_launch(s, false, "knn mean dist");
}

}  // namespace r3dg
