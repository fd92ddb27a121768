// LBVH build (K17) and visibility trace (K18) for gfx950.
// Reference semantics: construct_bvh bvh/src/construct.cu:147-265 (Morton codes of leaf-box centroids, stable sort,
// Karras ranges/splits, bottom-up box merge + leaf counts) and trace_bvh_opacity_cuda bvh/src/trace.cu:196-286
// (stack traversal, per-leaf Gaussian attenuation, T < 0.9 -> 0).
//
// Compiled with -ffp-contract=off: Morton codes decide the (integer) tree topology and must equal the oracle's.
// Differences from the reference's thrust pipeline:
//   * the 30-bit Morton keys are sorted by this library's own stable LSD radix sort (3 passes of 10 bits);
//   * the bottom-up merge publishes each child box with an agent-scope release before the arrival flag and the
//     second arriver acquires before reading its sibling's box -- gfx950's per-XCD L2s are not coherent, so the
//     reference's fence-free atomicCAS hand-off (construct.cu:243-258) would read stale boxes here;
//   * the traversal stack holds 64 entries (reference: 32 with only a printf on overflow, trace.cuh:21-28); pushes
//     beyond that are dropped and counted in *overflow so callers can detect it.
#include "common.hpp"

namespace r3dg {

struct Box {
    float lo[3], hi[3];
};

__device__ __forceinline__ uint32_t expand_bits(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ int common_upper_bits(uint64_t a, uint64_t b) { return __clzll((long long)(a ^ b)); }

// ---- whole-scene box: per-block partials then one small block ----
__global__ void __launch_bounds__(256) whole_box_partial_kernel(int P, const float* __restrict__ leaf, float* __restrict__ partial)
{
    __shared__ float s[4][6];
    float b[6] = {100000.f, 100000.f, 100000.f, -100000.f, -100000.f, -100000.f};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            b[a] = fminf(b[a], leaf[6 * (size_t)i + a]);
            b[3 + a] = fmaxf(b[3 + a], leaf[6 * (size_t)i + 3 + a]);
        }
    }
#pragma unroll
    for (int a = 0; a < 6; a++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float n = __shfl_xor(b[a], o, 64);
            b[a] = a < 3 ? fminf(b[a], n) : fmaxf(b[a], n);
        }
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
__global__ void whole_box_final_kernel(int nb, const float* __restrict__ partial, float* __restrict__ whole)
{
    const int a = threadIdx.x;
    if (a >= 6) return;
    float v = a < 3 ? 100000.f : -100000.f;
    for (int i = 0; i < nb; i++) v = a < 3 ? fminf(v, partial[i * 6 + a]) : fmaxf(v, partial[i * 6 + a]);
    whole[a] = v;
}

// Morton code of the leaf-box centroid (construct.cu:23-51); also copies the unsorted boxes aside
__global__ void __launch_bounds__(256)
morton_kernel(int P, const float* __restrict__ leaf, const float* __restrict__ whole, uint64_t* __restrict__ keys,
              uint32_t* __restrict__ vals, float* __restrict__ leaf_copy)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float c[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float lo = leaf[6 * (size_t)i + a], hi = leaf[6 * (size_t)i + 3 + a];
        leaf_copy[6 * (size_t)i + a] = lo;
        leaf_copy[6 * (size_t)i + 3 + a] = hi;
        float p = (float)((hi + lo) * 0.5);
        p -= whole[a];
        p /= (whole[3 + a] - whole[a]);
        c[a] = fminf(fmaxf(p * 1024.0f, 0.0f), 1024.0f - 1.0f);
    }
    const uint32_t m = expand_bits((uint32_t)c[0]) * 4 + expand_bits((uint32_t)c[1]) * 2 + expand_bits((uint32_t)c[2]);
    keys[i] = (uint64_t)m;
    vals[i] = (uint32_t)i;
}

// sorted order -> leaf rows of aabbs, 64-bit codes (m << 31 | original index, sic), leaf object ids
__global__ void __launch_bounds__(256)
scatter_leaves_kernel(int P, const uint64_t* __restrict__ keys_sorted, const uint32_t* __restrict__ idx_sorted,
                      const float* __restrict__ leaf_copy, float* __restrict__ aabbs, int32_t* __restrict__ nodes,
                      uint64_t* __restrict__ morton)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= P) return;
    const uint32_t src = idx_sorted[j];
    const size_t row = (size_t)(P - 1 + j);
#pragma unroll
    for (int a = 0; a < 6; a++) aabbs[6 * row + a] = leaf_copy[6 * (size_t)src + a];
    morton[j] = (keys_sorted[j] << 31) | (uint64_t)src;
    nodes[5 * row + 3] = (int32_t)src;
}

// Karras internal nodes (construct.cu:54-145, 203-229)
__global__ void __launch_bounds__(256)
internal_nodes_kernel(int P, const uint64_t* __restrict__ code, int32_t* __restrict__ nodes, int2* __restrict__ ranges,
                      int* __restrict__ nonzero_count_seen)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P - 1) return;
    const int num_leaves = P;
    int first, last;
    if (idx == 0) {
        first = 0;
        last = num_leaves - 1;
    } else {
        const uint64_t self = code[idx];
        const int L_delta = common_upper_bits(self, code[idx - 1]);
        const int R_delta = common_upper_bits(self, code[idx + 1]);
        const int d = (R_delta > L_delta) ? 1 : -1;
        const int delta_min = min(L_delta, R_delta);
        int l_max = 2;
        int delta = -1;
        int i_tmp = idx + d * l_max;
        if (0 <= i_tmp && i_tmp < num_leaves) delta = common_upper_bits(self, code[i_tmp]);
        while (delta > delta_min) {
            l_max <<= 1;
            i_tmp = idx + d * l_max;
            delta = -1;
            if (0 <= i_tmp && i_tmp < num_leaves) delta = common_upper_bits(self, code[i_tmp]);
        }
        int l = 0;
        int t = l_max >> 1;
        while (t > 0) {
            i_tmp = idx + (l + t) * d;
            delta = -1;
            if (0 <= i_tmp && i_tmp < num_leaves) delta = common_upper_bits(self, code[i_tmp]);
            if (delta > delta_min) l += t;
            t >>= 1;
        }
        const int jdx = idx + l * d;
        first = min(idx, jdx);
        last = max(idx, jdx);
    }
    // find_split
    int split;
    {
        const uint64_t first_code = code[first], last_code = code[last];
        if (first_code == last_code) {
            split = (first + last) >> 1;
        } else {
            const int delta_node = common_upper_bits(first_code, last_code);
            split = first;
            int stride = last - first;
            do {
                stride = (stride + 1) >> 1;
                const int middle = split + stride;
                if (middle < last) {
                    const int delta = common_upper_bits(first_code, code[middle]);
                    if (delta > delta_node) split = middle;
                }
            } while (stride > 1);
        }
    }
    int left = split, right = split + 1;
    if (first == split) left += P - 1;
    if (last == split + 1) right += P - 1;
    int32_t* node = nodes + 5 * (size_t)idx;
    node[1] = left;
    node[2] = right;
    node[3] = -1;
    nodes[5 * (size_t)left] = idx;
    nodes[5 * (size_t)right] = idx;
    ranges[idx] = make_int2(first, last);                  // the leaves below this node (range_boxes_kernel)
    if (node[4] != 0) atomicOr(nonzero_count_seen, 1);     // caller-initialised leaf counter (bvh/__init__.py:29-57 puts 0)
}

// bottom-up merge (construct.cu:231-264) with explicit release/acquire around the arrival flag
__global__ void __launch_bounds__(256)
merge_boxes_kernel(int P, int32_t* __restrict__ nodes, float* __restrict__ aabbs, int* __restrict__ flags,
                   const int* __restrict__ nonzero_count_seen)
{
    // (only for a node table whose internal leaf counters were not zero on entry: the counters then depend on the walk)
    if (nonzero_count_seen != nullptr && *nonzero_count_seen == 0) return;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= P) return;
    int idx = P - 1 + j;
    int num = 1;
    int parent = nodes[5 * (size_t)idx];
    while (parent != -1) {
        __threadfence();                                        // release my (or the leaf's) box before arriving
        atomicAdd(&nodes[5 * (size_t)parent + 4], num);
        const int old = atomicCAS(&flags[parent], 0, 1);
        if (old == 0) return;                                   // first arrival: the sibling finishes this node
        __threadfence();                                        // acquire the sibling's box
        const int lidx = nodes[5 * (size_t)parent + 1], ridx = nodes[5 * (size_t)parent + 2];
        volatile const float* lb = aabbs + 6 * (size_t)lidx;
        volatile const float* rb = aabbs + 6 * (size_t)ridx;
        float m[6];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            m[a] = fminf(lb[a], rb[a]);
            m[3 + a] = fmaxf(lb[3 + a], rb[3 + a]);
        }
#pragma unroll
        for (int a = 0; a < 6; a++) aabbs[6 * (size_t)parent + a] = m[a];
        num = atomicAdd(&nodes[5 * (size_t)parent + 4], 0);
        idx = parent;
        parent = nodes[5 * (size_t)parent];
    }
}

// ---- boxes of the internal nodes without inter-thread synchronisation ----------------------------------------------------------
// The reference walks up from every leaf; the second thread to arrive at a node merges its children's boxes
// (construct.cu:231-264).  That needs a release/acquire pair per level and thread, and on this part a device-scope fence is
// an L2 write-back + invalidate across the 8 XCDs: merge_boxes_kernel below spends 2.06 ms on 300k leaves, 93 % of it
// waiting.  But the box of a node is just min / max over the leaves of its RANGE [first, last] (Karras ranges are
// contiguous in Morton order), and min / max give the same bits in any order.  So: boxes of aligned runs of 256 leaves and
// of 256 such runs (two tiny tables), then every internal node reduces its range from at most 2 x 255 leaves + 2 x 255 runs
// + the super-runs between them -- independent threads, no atomics, no fences.  The leaf counter of the node table
// (column 4: the reference adds the children's counters into whatever the caller put there) is last - first + 1 when the
// caller's internal rows hold 0 as the reference's own RayTracer prepares them; any other initial content is detected on
// the device and routed through the reference-shaped walk.
constexpr int BOX_RUN = 256;

__device__ __forceinline__ void box_include(float (&b)[6], const float* __restrict__ o)
{
#pragma unroll
    for (int a = 0; a < 3; a++) {
        b[a] = fminf(b[a], o[a]);
        b[3 + a] = fmaxf(b[3 + a], o[3 + a]);
    }
}

// level 0 -> 1: one thread per run of 256 consecutive boxes (n boxes in, ceil(n/256) out)
__global__ void __launch_bounds__(256)
run_boxes_kernel(int n, const float* __restrict__ in, float* __restrict__ out)
{
    __shared__ float s_b[4][6];
    const int run = blockIdx.x, i = run * BOX_RUN + threadIdx.x;
    float b[6] = {3.0e38f, 3.0e38f, 3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
    if (i < n) box_include(b, in + 6 * (size_t)i);
#pragma unroll
    for (int a = 0; a < 6; a++) {
        float v = b[a];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float w = __shfl_xor(v, o, 64);
            v = a < 3 ? fminf(v, w) : fmaxf(v, w);
        }
        if ((threadIdx.x & 63) == 0) s_b[threadIdx.x >> 6][a] = v;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        const float v0 = s_b[0][a], v1 = s_b[1][a], v2 = s_b[2][a], v3 = s_b[3][a];
        out[6 * (size_t)run + a] = a < 3 ? fminf(fminf(v0, v1), fminf(v2, v3)) : fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
    }
}

__global__ void __launch_bounds__(256)
range_boxes_kernel(int P, const int2* __restrict__ ranges, const float* __restrict__ leaf /* Morton order */,
                   const float* __restrict__ run1, const float* __restrict__ run2, const int* __restrict__ nonzero_count_seen,
                   int32_t* __restrict__ nodes, float* __restrict__ aabbs)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P - 1) return;
    const int2 r = ranges[idx];
    int i = r.x;
    const int last = r.y;
    float b[6] = {3.0e38f, 3.0e38f, 3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
    // leaves up to the next run boundary, whole runs up to the next super-run boundary, whole super-runs, and down again
    while (i <= last && (i % BOX_RUN) != 0) { box_include(b, leaf + 6 * (size_t)i); i++; }
    while (i + BOX_RUN - 1 <= last && ((i / BOX_RUN) % BOX_RUN) != 0) { box_include(b, run1 + 6 * (size_t)(i / BOX_RUN)); i += BOX_RUN; }
    while (i + BOX_RUN * BOX_RUN - 1 <= last) { box_include(b, run2 + 6 * (size_t)(i / (BOX_RUN * BOX_RUN))); i += BOX_RUN * BOX_RUN; }
    while (i + BOX_RUN - 1 <= last) { box_include(b, run1 + 6 * (size_t)(i / BOX_RUN)); i += BOX_RUN; }
    while (i <= last) { box_include(b, leaf + 6 * (size_t)i); i++; }
#pragma unroll
    for (int a = 0; a < 6; a++) aabbs[6 * (size_t)idx + a] = b[a];
    if (*nonzero_count_seen == 0) nodes[5 * (size_t)idx + 4] = last - r.x + 1;
}

// ---- traversal ----
__device__ __forceinline__ float slab_tmax(const float* __restrict__ box, float ox, float oy, float oz, float dx,
                                           float dy, float dz)
{
    float tmin = (box[0] - ox) / dx;
    float tmax = (box[3] - ox) / dx;
    if (tmin > tmax) { const float t = tmin; tmin = tmax; tmax = t; }
    float tymin = (box[1] - oy) / dy;
    float tymax = (box[4] - oy) / dy;
    if (tymin > tymax) { const float t = tymin; tymin = tymax; tymax = t; }
    if (tmin > tymax || tymin > tmax) return -1.0f;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (box[2] - oz) / dz;
    float tzmax = (box[5] - oz) / dz;
    if (tzmin > tzmax) { const float t = tzmin; tzmin = tzmax; tzmax = t; }
    if (tmin > tzmax || tzmin > tmax) return -1.0f;
    if (tzmax < tmax) tmax = tzmax;
    return tmax;
}

constexpr int TRACE_STACK = 64;

__global__ void __launch_bounds__(256)
trace_opacity_kernel(int num_rays, const int32_t* __restrict__ nodes, const float* __restrict__ aabbs,
                     const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                     const float* __restrict__ means, const float* __restrict__ covs, const float* __restrict__ opac,
                     const float* __restrict__ normals, int32_t* __restrict__ contributes, float* __restrict__ out,
                     int* __restrict__ overflow)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= num_rays) return;
    const float ox = rays_o[3 * (size_t)r], oy = rays_o[3 * (size_t)r + 1], oz = rays_o[3 * (size_t)r + 2];
    const float dx = rays_d[3 * (size_t)r], dy = rays_d[3 * (size_t)r + 1], dz = rays_d[3 * (size_t)r + 2];
    int stack[TRACE_STACK];
    int sp = 0;
    stack[sp++] = 0;
    int count = 0;
    float T = 1.0f;
    bool lost = false;
    while (sp > 0) {
        const int node_id = stack[--sp];
        const int32_t* node = nodes + 5 * (size_t)node_id;
        if (node[4] <= 1) {
            const int g = node[3];
            const float op = opac[g];
            if (op < 1.f / 255.f) continue;
            const float nx = normals[3 * (size_t)g], ny = normals[3 * (size_t)g + 1], nz = normals[3 * (size_t)g + 2];
            if (nx * dx + ny * dy + nz * dz > 0) continue;
            const float* ci = covs + 6 * (size_t)g;
            const float c0 = ci[0], c1 = ci[1], c2 = ci[2], c3 = ci[3], c4 = ci[4], c5 = ci[5];
            const float mx = means[3 * (size_t)g], my = means[3 * (size_t)g + 1], mz = means[3 * (size_t)g + 2];
            const float m0 = mx - ox, m1 = my - oy, m2 = mz - oz;
            const float t1 = c0 * m0 * dx + c1 * m0 * dy + c2 * m0 * dz + c1 * m1 * dx + c3 * m1 * dy + c4 * m1 * dz +
                             c2 * m2 * dx + c4 * m2 * dy + c5 * m2 * dz;
            const float t2 = c0 * dx * dx + c1 * dx * dy + c2 * dx * dz + c1 * dy * dx + c3 * dy * dy + c4 * dy * dz +
                             c2 * dz * dx + c4 * dz * dy + c5 * dz * dz;
            const float t = t1 / t2;
            if (t < 0.01) continue;
            const float px = ox + t * dx, py = oy + t * dy, pz = oz + t * dz;
            const float f0 = mx - px, f1 = my - py, f2 = mz - pz;
            const float s = f0 * f0 * c0 + f1 * f1 * c3 + f2 * f2 * c5 + 2 * f0 * f1 * c1 + 2 * f0 * f2 * c2 +
                            2 * f1 * f2 * c4;
            const float power = -0.5f * s;
            if (power > 0) continue;
            count += 1;
            const float alpha = op * __expf(power);
            T *= 1 - alpha;
            if (T < 0.9) {
                out[r] = 0.0f;        // contributes[r] keeps its initial 0 (trace.cu:251-254)
                return;
            }
        } else {
            const int lid = node[1], rid = node[2];
            const float tl = slab_tmax(aabbs + 6 * (size_t)lid, ox, oy, oz, dx, dy, dz);
            const float tr = slab_tmax(aabbs + 6 * (size_t)rid, ox, oy, oz, dx, dy, dz);
            const int first = tl > tr ? lid : rid, second = tl > tr ? rid : lid;
            const float tf = tl > tr ? tl : tr, ts = tl > tr ? tr : tl;
            if (tf > 0) { if (sp < TRACE_STACK) stack[sp++] = first; else lost = true; }
            if (ts > 0) { if (sp < TRACE_STACK) stack[sp++] = second; else lost = true; }
        }
    }
    contributes[r] = count;
    out[r] = T;
    if (lost) atomicAdd(overflow, 1);
}

// ---- wave-cooperative ("packet") traversal ---------------------------------------------------------------------------------
// One wave = 64 consecutive rays (the K rays of one Gaussian's visibility bundle share their origin, gaussian_model.py:312-342,
// but nothing below assumes it).  The wave keeps ONE traversal stack (LDS; node id + the 64-bit mask of the rays that hit
// that node's box) and visits the UNION of the nodes its rays reach, each once: node record, child boxes and leaf Gaussian
// are wave-uniform -> scalar loads, shared by the 64 slab tests / attenuation evaluations of one VALU pass.  The thread-per-
// ray kernel above gathers 17 words from up to 64 different nodes per step and idles the lanes whose ray finished early.
// Per-ray semantics are those of the reference (trace.cu:208-280): a ray meets a node iff tmax > 0 for that node's box
// (mask bit), every Gaussian it meets goes through the same accept chain, T < 0.9 retires the ray with 0.  The visit ORDER
// differs from a single ray's, which can only matter for a ray whose product crosses 0.9 within rounding.
constexpr int PACKET_STACK = 96;

__global__ void __launch_bounds__(256)
trace_opacity_packet_kernel(int num_rays, const int32_t* __restrict__ nodes, const float* __restrict__ aabbs,
                            const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                            const float* __restrict__ means, const float* __restrict__ covs,
                            const float* __restrict__ opac, const float* __restrict__ normals,
                            int32_t* __restrict__ contributes, float* __restrict__ out, int* __restrict__ overflow)
{
    __shared__ int s_node[4][PACKET_STACK];
    __shared__ unsigned long long s_mask[4][PACKET_STACK];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = blockIdx.x * 256 + threadIdx.x;
    const bool valid = r < num_rays;
    const int rr = valid ? r : num_rays - 1;
    const float ox = rays_o[3 * (size_t)rr], oy = rays_o[3 * (size_t)rr + 1], oz = rays_o[3 * (size_t)rr + 2];
    const float dx = rays_d[3 * (size_t)rr], dy = rays_d[3 * (size_t)rr + 1], dz = rays_d[3 * (size_t)rr + 2];
    int* st_node = s_node[wave];
    unsigned long long* st_mask = s_mask[wave];
    int sp = 0;
    const unsigned long long all = __ballot(valid);
    if (all == 0ull) return;
    if (lane == 0) { st_node[0] = 0; st_mask[0] = all; }
    sp = 1;
    int count = 0;
    float T = 1.0f;
    bool alive = valid;
    bool lost = false;
    while (sp > 0) {
        --sp;
        const int node_id = __builtin_amdgcn_readfirstlane(st_node[sp]);
        const unsigned long long m = st_mask[sp];
        const unsigned long long mask = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32) |
                                        (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)m);
        const unsigned long long live = mask & __ballot(alive);
        if (live == 0ull) continue;
        const bool me = ((live >> lane) & 1ull) != 0ull;
        const int32_t* node = nodes + 5 * (size_t)node_id;
        if (node[4] <= 1) {
            const int g = node[3];
            const float op = opac[g];
            if (op < 1.f / 255.f) continue;
            const float nx = normals[3 * (size_t)g], ny = normals[3 * (size_t)g + 1], nz = normals[3 * (size_t)g + 2];
            const float* ci = covs + 6 * (size_t)g;
            const float c0 = ci[0], c1 = ci[1], c2 = ci[2], c3 = ci[3], c4 = ci[4], c5 = ci[5];
            const float mx = means[3 * (size_t)g], my = means[3 * (size_t)g + 1], mz = means[3 * (size_t)g + 2];
            if (me && !(nx * dx + ny * dy + nz * dz > 0)) {
                const float m0 = mx - ox, m1 = my - oy, m2 = mz - oz;
                const float t1 = c0 * m0 * dx + c1 * m0 * dy + c2 * m0 * dz + c1 * m1 * dx + c3 * m1 * dy + c4 * m1 * dz +
                                 c2 * m2 * dx + c4 * m2 * dy + c5 * m2 * dz;
                const float t2 = c0 * dx * dx + c1 * dx * dy + c2 * dx * dz + c1 * dy * dx + c3 * dy * dy + c4 * dy * dz +
                                 c2 * dz * dx + c4 * dz * dy + c5 * dz * dz;
                const float t = t1 / t2;
                if (!(t < 0.01)) {
                    const float px = ox + t * dx, py = oy + t * dy, pz = oz + t * dz;
                    const float f0 = mx - px, f1 = my - py, f2 = mz - pz;
                    const float s = f0 * f0 * c0 + f1 * f1 * c3 + f2 * f2 * c5 + 2 * f0 * f1 * c1 + 2 * f0 * f2 * c2 +
                                    2 * f1 * f2 * c4;
                    const float power = -0.5f * s;
                    if (!(power > 0)) {
                        count += 1;
                        const float alpha = op * __expf(power);
                        T *= 1 - alpha;
                        if (T < 0.9) alive = false;          // retired: the result is 0 and the count stays 0 (trace.cu:251-254)
                    }
                }
            }
        } else {
            const int lid = node[1], rid = node[2];
            float tl = -1.0f, tr = -1.0f;
            if (me) {
                tl = slab_tmax(aabbs + 6 * (size_t)lid, ox, oy, oz, dx, dy, dz);
                tr = slab_tmax(aabbs + 6 * (size_t)rid, ox, oy, oz, dx, dy, dz);
            }
            const unsigned long long ml = __ballot(me && tl > 0), mr = __ballot(me && tr > 0);
            // the child more of the wave's rays want is popped first (pushed last)
            const bool left_first = __popcll(ml) >= __popcll(mr);
            const int n0 = left_first ? rid : lid, n1 = left_first ? lid : rid;
            const unsigned long long m0 = left_first ? mr : ml, m1 = left_first ? ml : mr;
            if (m0 != 0ull) {
                if (sp < PACKET_STACK) { if (lane == 0) { st_node[sp] = n0; st_mask[sp] = m0; } sp++; } else lost = true;
            }
            if (m1 != 0ull) {
                if (sp < PACKET_STACK) { if (lane == 0) { st_node[sp] = n1; st_mask[sp] = m1; } sp++; } else lost = true;
            }
        }
    }
    if (valid) {
        contributes[r] = alive ? count : 0;
        out[r] = alive ? T : 0.0f;
    }
    if (lost && lane == 0) atomicAdd(overflow, 1);
}

// ---- packed traversal records + XCD-aware thread-per-ray traversal -----------------------------------------------------------
// rocprofv3 on the thread-per-ray kernel above (P=300k, K=64; profiles/r02_pmc_trace.json): 12.7e9 L2 requests per 6.4 M rays,
// 45 % of them L2 misses (3.3 TB/s of 64-byte lines from the Infinity Cache), VALU busy 37 %, lanes 26 % utilised -- the walk
// is bound by memory transactions, not by issue.  Every step touches 3 lines (node record 20 B, two child boxes 24 B each at
// unrelated rows) or 5 (leaf: node + opacity + normal + covariance + mean from five arrays), the 42 MB working set is ten
// times one XCD's 4 MB L2, and consecutive blocks -- dispatched round-robin over the 8 XCDs -- carry unrelated origins.
//   * pack_traversal_kernel rewrites the tree once per trace call into one 64-byte record per internal node (both child ids
//     and both child boxes) and one per leaf IN MORTON ORDER (mean, inverse covariance, opacity, normal): one line per step;
//   * the host traces the ray bundles in Morton order of their origin Gaussian (train_step.update_visibility), and the kernel
//     hands each XCD a contiguous run of blocks, so an XCD's L2 serves one region of the scene (its subtree + the shared top
//     levels) instead of 1/8 of everything;
//   * per-ray semantics, arithmetic and visit order are unchanged (trace.cu:208-280): results are bit-identical.
// The wave-cooperative variant (shared stack, union of the 64 rays' nodes) and persistent waves with dynamic ray fetch were
// measured too: 37 and 49 vs 56 Mrays/s at K=64 -- fewer instructions (lanes 48 % utilised with refill) but MORE L2 misses.
// ---- exact quotients without the division expansion ---------------------------------------------------------------------------
// A slab test divides six differences by the ray direction; hipcc expands each IEEE fp32 division into 11 instructions
// (2 v_div_scale, v_rcp, 5 FMA/mul, v_div_fmas, v_div_fixup): 12 divisions = two thirds of a node step's VALU work, all by
// the SAME three divisors for the whole life of the ray.  The expansion is r = rcp(d); r += (1 - d r) r; q = a r;
// q += (a - d q) r; q += (a - d q) r, wrapped in a power-of-two pre/post scaling that only engages for extreme exponents
// and a fix-up for zeros / infinities / NaNs.  Outside those cases the scaling is the identity, so keeping the refined
// reciprocal per ray and running the five remaining operations gives the SAME bits (same operations on the same values).
// Sufficient for "no scaling, no fix-up" (ISA, V_DIV_SCALE_F32): 2^-63 <= |d| <= 2 and the numerator 0 or
// 2^-103 <= |a| < 2^32; the latter holds for every difference of two TAME coordinates (0, or 2^-60 <= |x| < 2^31).
// Rays or nodes that are not tame (never in practice) take the compiler's division.  The sign of a zero quotient may
// differ; quotients are only compared.
__device__ __forceinline__ bool tame_coordinate(float x)
{
    const uint32_t e = (__float_as_uint(x) >> 23) & 0xffu;
    return x == 0.0f || (e >= 127u - 60u && e < 127u + 31u);
}
__device__ __forceinline__ bool tame_direction(float d)
{
    const uint32_t e = (__float_as_uint(d) >> 23) & 0xffu;
    return e >= 127u - 63u && e <= 127u;            // 2^-63 <= |d| < 2
}
__device__ __forceinline__ float refined_reciprocal(float d)
{
    const float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float exact_quotient(float a, float neg_d, float r)
{
    const float q0 = a * r;
    const float e0 = __builtin_fmaf(neg_d, q0, a);
    const float q1 = __builtin_fmaf(e0, r, q0);
    const float e1 = __builtin_fmaf(neg_d, q1, a);
    return __builtin_fmaf(e1, r, q1);
}
// slab_tmax with the quotients above (same comparisons, same order)
__device__ __forceinline__ float slab_tmax_tame(const float* __restrict__ box, float ox, float oy, float oz, float ndx,
                                                float ndy, float ndz, float rx, float ry, float rz)
{
    float tmin = exact_quotient(box[0] - ox, ndx, rx);
    float tmax = exact_quotient(box[3] - ox, ndx, rx);
    if (tmin > tmax) { const float t = tmin; tmin = tmax; tmax = t; }
    float tymin = exact_quotient(box[1] - oy, ndy, ry);
    float tymax = exact_quotient(box[4] - oy, ndy, ry);
    if (tymin > tymax) { const float t = tymin; tymin = tymax; tymax = t; }
    if (tmin > tymax || tymin > tmax) return -1.0f;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = exact_quotient(box[2] - oz, ndz, rz);
    float tzmax = exact_quotient(box[5] - oz, ndz, rz);
    if (tzmin > tzmax) { const float t = tzmin; tzmin = tzmax; tzmax = t; }
    if (tmin > tzmax || tzmin > tmax) return -1.0f;
    if (tzmax < tmax) tmax = tzmax;
    return tmax;
}

struct __attribute__((aligned(16))) TNode {
    int left, right;
    float lb[6], rb[6];
    int pad[2];
};
struct __attribute__((aligned(16))) TLeaf {
    float mean[3], cov[6], op, n[3];
    float pad[3];
};
static_assert(sizeof(TNode) == 64 && sizeof(TLeaf) == 64, "one cache line per traversal record");

__global__ void __launch_bounds__(256)
pack_traversal_kernel(int P, const int32_t* __restrict__ nodes, const float* __restrict__ aabbs,
                      const float* __restrict__ means, const float* __restrict__ covs, const float* __restrict__ opac,
                      const float* __restrict__ normals, TNode* __restrict__ tn, TLeaf* __restrict__ tl)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < P - 1) {
        TNode o;
        o.left = nodes[5 * (size_t)i + 1];
        o.right = nodes[5 * (size_t)i + 2];
#pragma unroll
        for (int a = 0; a < 6; a++) {
            o.lb[a] = aabbs[6 * (size_t)o.left + a];
            o.rb[a] = aabbs[6 * (size_t)o.right + a];
        }
        // pad[0] = 1: every box coordinate is 0 or has 2^-60 <= |x| < 2^31 (see exact_quotient)
        bool tame = true;
#pragma unroll
        for (int a = 0; a < 6; a++) tame = tame && tame_coordinate(o.lb[a]) && tame_coordinate(o.rb[a]);
        o.pad[0] = tame ? 1 : 0;
        o.pad[1] = 0;
        tn[i] = o;
    }
    if (i < P) {
        const int g = nodes[5 * (size_t)(P - 1 + i) + 3];
        TLeaf o;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            o.mean[a] = means[3 * (size_t)g + a];
            o.n[a] = normals[3 * (size_t)g + a];
            o.pad[a] = 0.f;
        }
#pragma unroll
        for (int a = 0; a < 6; a++) o.cov[a] = covs[6 * (size_t)g + a];
        o.op = opac[g];
        tl[i] = o;
    }
}

__global__ void __launch_bounds__(256)
trace_opacity_packed_kernel(int num_rays, int P, int xcd_chunk, const TNode* __restrict__ tn, const TLeaf* __restrict__ tl,
                            const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                            int32_t* __restrict__ contributes, float* __restrict__ out, int* __restrict__ overflow)
{
    // hardware places block b on XCD b % 8: give every XCD a contiguous run of ray blocks (= a region of the scene when the
    // bundles arrive in Morton order)
    const int blk = (int)(blockIdx.x & 7u) * xcd_chunk + (int)(blockIdx.x >> 3);
    const long long r = (long long)blk * 256 + threadIdx.x;
    if (r >= num_rays) return;
    const float ox = rays_o[3 * (size_t)r], oy = rays_o[3 * (size_t)r + 1], oz = rays_o[3 * (size_t)r + 2];
    const float dx = rays_d[3 * (size_t)r], dy = rays_d[3 * (size_t)r + 1], dz = rays_d[3 * (size_t)r + 2];
    int stack[TRACE_STACK];
    int sp = 0;
    stack[sp++] = 0;
    int count = 0;
    float T = 1.0f;
    bool lost = false;
    const int first_leaf = P - 1;
    while (sp > 0) {
        const int node_id = stack[--sp];
        if (node_id >= first_leaf) {
            const float4* q = reinterpret_cast<const float4*>(tl + (node_id - first_leaf));
            const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            const float op = q2.y;
            if (op < 1.f / 255.f) continue;
            const float nx = q2.z, ny = q2.w, nz = q3.x;
            if (nx * dx + ny * dy + nz * dz > 0) continue;
            const float c0 = q0.w, c1 = q1.x, c2 = q1.y, c3 = q1.z, c4 = q1.w, c5 = q2.x;
            const float mx = q0.x, my = q0.y, mz = q0.z;
            const float m0 = mx - ox, m1 = my - oy, m2 = mz - oz;
            const float t1 = c0 * m0 * dx + c1 * m0 * dy + c2 * m0 * dz + c1 * m1 * dx + c3 * m1 * dy + c4 * m1 * dz +
                             c2 * m2 * dx + c4 * m2 * dy + c5 * m2 * dz;
            const float t2 = c0 * dx * dx + c1 * dx * dy + c2 * dx * dz + c1 * dy * dx + c3 * dy * dy + c4 * dy * dz +
                             c2 * dz * dx + c4 * dz * dy + c5 * dz * dz;
            const float t = t1 / t2;
            if (t < 0.01) continue;
            const float px = ox + t * dx, py = oy + t * dy, pz = oz + t * dz;
            const float f0 = mx - px, f1 = my - py, f2 = mz - pz;
            const float s = f0 * f0 * c0 + f1 * f1 * c3 + f2 * f2 * c5 + 2 * f0 * f1 * c1 + 2 * f0 * f2 * c2 +
                            2 * f1 * f2 * c4;
            const float power = -0.5f * s;
            if (power > 0) continue;
            count += 1;
            const float alpha = op * __expf(power);
            T *= 1 - alpha;
            if (T < 0.9) {
                out[r] = 0.0f;        // contributes[r] keeps its initial 0 (trace.cu:251-254)
                return;
            }
        } else {
            const float4* q = reinterpret_cast<const float4*>(tn + node_id);
            const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            const int lid = __float_as_int(q0.x), rid = __float_as_int(q0.y);
            const float lb[6] = {q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            const float rb[6] = {q2.x, q2.y, q2.z, q2.w, q3.x, q3.y};
            const float tl_ = slab_tmax(lb, ox, oy, oz, dx, dy, dz);
            const float tr_ = slab_tmax(rb, ox, oy, oz, dx, dy, dz);
            const int first = tl_ > tr_ ? lid : rid, second = tl_ > tr_ ? rid : lid;
            const float tf = tl_ > tr_ ? tl_ : tr_, ts = tl_ > tr_ ? tr_ : tl_;
            if (tf > 0) { if (sp < TRACE_STACK) stack[sp++] = first; else lost = true; }
            if (ts > 0) { if (sp < TRACE_STACK) stack[sp++] = second; else lost = true; }
        }
    }
    contributes[r] = count;
    out[r] = T;
    if (lost) atomicAdd(overflow, 1);
}

// The same walk with PERSISTENT waves: a lane whose ray has finished (83 % of the visibility rays are occluded after a handful of
// leaves while the rest walk thousands of nodes, so a fixed assignment leaves ~3/4 of the lanes idle) pulls the next ray from
// a per-XCD queue -- each XCD owns a contiguous eighth of the (Morton-ordered) ray set, so the locality argument above holds --
// with one wave-aggregated atomic per refill.  Per-ray arithmetic and visit order are unchanged.
constexpr int REFILL_MIN_IDLE = 16;      // refill when at least this many lanes are idle (or the whole wave is)

__global__ void __launch_bounds__(256)
trace_opacity_persistent_kernel(int num_rays, int P, const TNode* __restrict__ tn, const TLeaf* __restrict__ tl,
                                const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                int32_t* __restrict__ contributes, float* __restrict__ out, int* __restrict__ overflow,
                                int* __restrict__ queues /* 8 x 16 ints, zeroed */)
{
    const int lane = threadIdx.x & 63;
    const int xcd = (int)(blockIdx.x & 7u);
    const int per = (num_rays + 7) / 8;
    const int q_lo = xcd * per, q_hi = min(num_rays, q_lo + per);
    int* next_ray = queues + 16 * xcd;
    int stack[TRACE_STACK];
    int sp = 0, ray = -1, count = 0;
    float ox = 0.f, oy = 0.f, oz = 0.f, dx = 0.f, dy = 0.f, dz = 1.f, T = 1.0f;
    bool lost = false, exhausted = false;
    const int first_leaf = P - 1;
    while (true) {
        const unsigned long long idle = __ballot(ray < 0);
        if (idle != 0ull && !exhausted) {
            const int n_idle = __popcll(idle);
            if (n_idle >= REFILL_MIN_IDLE || idle == __ballot(true)) {
                int base = 0;
                const int leader = __builtin_ctzll(idle);
                if (lane == leader) base = atomicAdd(next_ray, n_idle);
                base = q_lo + __builtin_amdgcn_readlane(base, leader);
                if (base + n_idle >= q_hi) exhausted = true;
                if (ray < 0) {
                    const int idx = base + __popcll(idle & ((1ull << lane) - 1ull));
                    if (idx < q_hi) {
                        ray = idx;
                        ox = rays_o[3 * (size_t)idx]; oy = rays_o[3 * (size_t)idx + 1]; oz = rays_o[3 * (size_t)idx + 2];
                        dx = rays_d[3 * (size_t)idx]; dy = rays_d[3 * (size_t)idx + 1]; dz = rays_d[3 * (size_t)idx + 2];
                        stack[0] = 0;
                        sp = 1;
                        count = 0;
                        T = 1.0f;
                    }
                }
            }
        }
        if (__ballot(ray >= 0) == 0ull) {
            if (exhausted) break;
            continue;
        }
        if (ray >= 0) {
            bool finished = false;
            const int node_id = stack[--sp];
            if (node_id >= first_leaf) {
                const float4* q = reinterpret_cast<const float4*>(tl + (node_id - first_leaf));
                const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
                const float op = q2.y;
                const float nx = q2.z, ny = q2.w, nz = q3.x;
                if (!(op < 1.f / 255.f) && !(nx * dx + ny * dy + nz * dz > 0)) {
                    const float c0 = q0.w, c1 = q1.x, c2 = q1.y, c3 = q1.z, c4 = q1.w, c5 = q2.x;
                    const float mx = q0.x, my = q0.y, mz = q0.z;
                    const float m0 = mx - ox, m1 = my - oy, m2 = mz - oz;
                    const float t1 = c0 * m0 * dx + c1 * m0 * dy + c2 * m0 * dz + c1 * m1 * dx + c3 * m1 * dy + c4 * m1 * dz +
                                     c2 * m2 * dx + c4 * m2 * dy + c5 * m2 * dz;
                    const float t2 = c0 * dx * dx + c1 * dx * dy + c2 * dx * dz + c1 * dy * dx + c3 * dy * dy + c4 * dy * dz +
                                     c2 * dz * dx + c4 * dz * dy + c5 * dz * dz;
                    const float t = t1 / t2;
                    if (!(t < 0.01)) {
                        const float px = ox + t * dx, py = oy + t * dy, pz = oz + t * dz;
                        const float f0 = mx - px, f1 = my - py, f2 = mz - pz;
                        const float s = f0 * f0 * c0 + f1 * f1 * c3 + f2 * f2 * c5 + 2 * f0 * f1 * c1 + 2 * f0 * f2 * c2 +
                                        2 * f1 * f2 * c4;
                        const float power = -0.5f * s;
                        if (!(power > 0)) {
                            count += 1;
                            const float alpha = op * __expf(power);
                            T *= 1 - alpha;
                            if (T < 0.9) {          // retired with 0; contributes keeps 0 (trace.cu:251-254)
                                T = 0.0f;
                                count = 0;
                                finished = true;
                            }
                        }
                    }
                }
            } else {
                const float4* q = reinterpret_cast<const float4*>(tn + node_id);
                const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
                const int lid = __float_as_int(q0.x), rid = __float_as_int(q0.y);
                const float lb[6] = {q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                const float rb[6] = {q2.x, q2.y, q2.z, q2.w, q3.x, q3.y};
                const float tl_ = slab_tmax(lb, ox, oy, oz, dx, dy, dz);
                const float tr_ = slab_tmax(rb, ox, oy, oz, dx, dy, dz);
                const int first = tl_ > tr_ ? lid : rid, second = tl_ > tr_ ? rid : lid;
                const float tf = tl_ > tr_ ? tl_ : tr_, ts = tl_ > tr_ ? tr_ : tl_;
                if (tf > 0) { if (sp < TRACE_STACK) stack[sp++] = first; else lost = true; }
                if (ts > 0) { if (sp < TRACE_STACK) stack[sp++] = second; else lost = true; }
            }
            if (finished || sp == 0) {
                contributes[ray] = count;
                out[ray] = T;
                ray = -1;
                sp = 0;
            }
        }
    }
    if (lost) atomicAdd(overflow, 1);
}

// ---- phase-separated persistent traversal (R3DG_OPT_TRACE_FORMULATION = 4, default) -----------------------------------------------------
// In the kernel above every loop iteration executes BOTH the leaf body (Gaussian response, ~60 VALU) and the node body (two
// slab tests, ~70 VALU) whenever a wave holds lanes of either kind, which is nearly always: each lane advances one step for
// the price of two.  Here each lane keeps its current node in a register and the wave VOTES per iteration: the body with
// more ready lanes runs, the other lanes wait one turn.  The register also halves the stack traffic (a node whose children
// are both hit pushes one and continues with the other instead of push, push, pop).  Per-ray visit order, arithmetic and
// the overflow accounting are those of the kernels above, so results are identical.
// COUNT (R3DG_OPT_TRACE_COUNT_VISITS, measurement builds of the same kernel): every lane counts the node steps (one slab test of
// both children) and leaf steps (one Gaussian evaluated) of its rays; one 64-bit atomic pair per wave at the end, into words
// 8..11 of the wave's own queue line (zeroed with the queue heads before the launch; read by r3dg_bvh_trace_visits).
template <bool COUNT>
__global__ void __launch_bounds__(256)
trace_opacity_phased_kernel(int num_rays, int P, const TNode* __restrict__ tn, const TLeaf* __restrict__ tl,
                            const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                            int32_t* __restrict__ contributes, float* __restrict__ out, int* __restrict__ overflow,
                            int* __restrict__ queues /* 8 x 16 ints, zeroed */, int refill_min_idle, int node_weight,
                            int leaf_weight)
{
    const int lane = threadIdx.x & 63;
    const int xcd = (int)(blockIdx.x & 7u);
    const int per = (num_rays + 7) / 8;
    // own queue first (locality); once it is empty the wave helps the next XCD's queue, and so on round the ring, so
    // an XCD whose eighth of the scene is cheap does not idle while a dense eighth finishes
    int q_turn = 0;
    int q_lo = min(num_rays, xcd * per), q_hi = min(num_rays, q_lo + per);
    int* next_ray = queues + 16 * xcd;
    unsigned long long* visit_words = reinterpret_cast<unsigned long long*>(queues + 16 * xcd + 8);
    unsigned int n_node_steps = 0u, n_leaf_steps = 0u;
    int stack[TRACE_STACK];
    int sp = 0, ray = -1, count = 0, cur = -1;
    float ox = 0.f, oy = 0.f, oz = 0.f, dx = 0.f, dy = 0.f, dz = 1.f, T = 1.0f;
    float rx = 0.f, ry = 0.f, rz = 1.f;          // refined reciprocals of the direction (exact_quotient)
    bool lost = false, exhausted = false, tame_ray = false;
    const int first_leaf = P - 1;
    while (true) {
        const unsigned long long idle = __ballot(ray < 0);
        if (idle != 0ull && !exhausted) {
            const int n_idle = __popcll(idle);
            if (n_idle >= refill_min_idle || idle == __ballot(true)) {
                int base = 0;
                const int leader = __builtin_ctzll(idle);
                if (lane == leader) base = atomicAdd(next_ray, n_idle);
                base = q_lo + __builtin_amdgcn_readlane(base, leader);
                const bool drained = base + n_idle >= q_hi;
                if (ray < 0) {
                    const int idx = base + __popcll(idle & ((1ull << lane) - 1ull));
                    if (idx < q_hi) {
                        ray = idx;
                        ox = rays_o[3 * (size_t)idx]; oy = rays_o[3 * (size_t)idx + 1]; oz = rays_o[3 * (size_t)idx + 2];
                        dx = rays_d[3 * (size_t)idx]; dy = rays_d[3 * (size_t)idx + 1]; dz = rays_d[3 * (size_t)idx + 2];
                        rx = refined_reciprocal(dx); ry = refined_reciprocal(dy); rz = refined_reciprocal(dz);
                        tame_ray = tame_direction(dx) && tame_direction(dy) && tame_direction(dz) && tame_coordinate(ox) &&
                                   tame_coordinate(oy) && tame_coordinate(oz);
                        cur = 0;
                        sp = 0;
                        count = 0;
                        T = 1.0f;
                    }
                }
                if (drained) {
                    if (++q_turn == 8) exhausted = true;
                    else {
                        const int q = (xcd + q_turn) & 7;
                        q_lo = min(num_rays, q * per);
                        q_hi = min(num_rays, q_lo + per);
                        next_ray = queues + 16 * q;
                    }
                }
            }
        }
        const bool at_leaf = ray >= 0 && cur >= first_leaf;
        const bool at_node = ray >= 0 && cur < first_leaf;
        const unsigned long long leaf_m = __ballot(at_leaf), node_m = __ballot(at_node);
        if ((leaf_m | node_m) == 0ull) {
            if (exhausted) break;
            continue;
        }
        bool finished = false, stepped = false;
        if (__popcll(node_m) * node_weight >= __popcll(leaf_m) * leaf_weight) {
            float4 q0 = {}, q1 = {}, q2 = {}, q3 = {};
            if (at_node) {
                const float4* q = reinterpret_cast<const float4*>(tn + cur);
                q0 = q[0]; q1 = q[1]; q2 = q[2]; q3 = q[3];
            }
            const float lb[6] = {q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            const float rb[6] = {q2.x, q2.y, q2.z, q2.w, q3.x, q3.y};
            float tl_ = -1.0f, tr_ = -1.0f;
            if (__ballot(at_node && !(tame_ray && __float_as_int(q3.z) != 0)) == 0ull) {
                if (at_node) {
                    tl_ = slab_tmax_tame(lb, ox, oy, oz, -dx, -dy, -dz, rx, ry, rz);
                    tr_ = slab_tmax_tame(rb, ox, oy, oz, -dx, -dy, -dz, rx, ry, rz);
                }
            } else if (at_node) {
                tl_ = slab_tmax(lb, ox, oy, oz, dx, dy, dz);
                tr_ = slab_tmax(rb, ox, oy, oz, dx, dy, dz);
            }
            if (at_node) {
                stepped = true;
                if (COUNT) ++n_node_steps;
                const int lid = __float_as_int(q0.x), rid = __float_as_int(q0.y);
                const int first = tl_ > tr_ ? lid : rid, second = tl_ > tr_ ? rid : lid;
                const float tf = tl_ > tr_ ? tl_ : tr_, ts = tl_ > tr_ ? tr_ : tl_;
                // the kernels above push `first`, then `second`, and pop `second` next: it stays in the register instead
                cur = -1;
                if (tf > 0) {
                    if (ts > 0) {
                        if (sp < TRACE_STACK) stack[sp++] = first; else lost = true;
                        if (sp < TRACE_STACK) cur = second; else lost = true;
                    } else {
                        if (sp < TRACE_STACK) cur = first; else lost = true;
                    }
                }
            }
        } else if (at_leaf) {
            stepped = true;
            if (COUNT) ++n_leaf_steps;
            const float4* q = reinterpret_cast<const float4*>(tl + (cur - first_leaf));
            const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            const float op = q2.y;
            const float nx = q2.z, ny = q2.w, nz = q3.x;
            if (!(op < 1.f / 255.f) && !(nx * dx + ny * dy + nz * dz > 0)) {
                const float c0 = q0.w, c1 = q1.x, c2 = q1.y, c3 = q1.z, c4 = q1.w, c5 = q2.x;
                const float mx = q0.x, my = q0.y, mz = q0.z;
                const float m0 = mx - ox, m1 = my - oy, m2 = mz - oz;
                const float t1 = c0 * m0 * dx + c1 * m0 * dy + c2 * m0 * dz + c1 * m1 * dx + c3 * m1 * dy + c4 * m1 * dz +
                                 c2 * m2 * dx + c4 * m2 * dy + c5 * m2 * dz;
                const float t2 = c0 * dx * dx + c1 * dx * dy + c2 * dx * dz + c1 * dy * dx + c3 * dy * dy + c4 * dy * dz +
                                 c2 * dz * dx + c4 * dz * dy + c5 * dz * dz;
                const float t = t1 / t2;
                if (!(t < 0.01)) {
                    const float px = ox + t * dx, py = oy + t * dy, pz = oz + t * dz;
                    const float f0 = mx - px, f1 = my - py, f2 = mz - pz;
                    const float s = f0 * f0 * c0 + f1 * f1 * c3 + f2 * f2 * c5 + 2 * f0 * f1 * c1 + 2 * f0 * f2 * c2 +
                                    2 * f1 * f2 * c4;
                    const float power = -0.5f * s;
                    if (!(power > 0)) {
                        count += 1;
                        const float alpha = op * __expf(power);
                        T *= 1 - alpha;
                        if (T < 0.9) {          // retired with 0; contributes keeps 0 (trace.cu:251-254)
                            T = 0.0f;
                            count = 0;
                            finished = true;
                        }
                    }
                }
            }
            cur = -1;
        }
        if (stepped) {
            if (!finished && cur < 0 && sp > 0) cur = stack[--sp];
            if (finished || cur < 0) {
                contributes[ray] = count;
                out[ray] = T;
                ray = -1;
                sp = 0;
                cur = -1;
            }
        }
    }
    if (lost) atomicAdd(overflow, 1);
    if (COUNT) {
        unsigned long long a = n_node_steps, b = n_leaf_steps;
        for (int d = 32; d > 0; d >>= 1) {
            a += __shfl_xor(a, d);
            b += __shfl_xor(b, d);
        }
        if (lane == 0) {
            atomicAdd(visit_words, a);
            atomicAdd(visit_words + 1, b);
        }
    }
}

int g_trace_count_visits = 0;        // R3DG_OPT_TRACE_COUNT_VISITS
int g_trace_packet = 4;
int g_trace_refill = REFILL_MIN_IDLE, g_trace_node_weight = 1, g_trace_leaf_weight = 1;     // R3DG_OPT_TRACE_*: 4 = 3 + phase-separated bodies, 3 = packed records + persistent waves, 2 = packed records, 1 = wave-cooperative, 0 = round-1 kernel

// ---- trace_bvh: per-ray hit lists (K19; bvh/src/trace.cu:8-192, bound at bvh/src/bindings.cpp:11) ----------------------------
// Pass 1 counts, per ray, the leaves of every subtree of <= 4 leaves whose box the ray reaches (tmax > 0 on the way down);
// the caller scans the counts; pass 2 repeats the walk carrying each node's (tmin, tmax) and writes one entry per such leaf:
// t = (mean - o) . d (the POINT form of ray_intersects, utility.cuh:84-88), rejected (t = 1e6, id = -1) unless
// 0.01 <= t and tmin <= t <= tmax of the collapsed subtree's box; key = ray << 32 | bits(t), position = o + t d.
// The caller then sorts the entries of each ray by t (stable sort on the key).  No Python caller exists in the reference.
__device__ __forceinline__ float2 slab_interval(const float* __restrict__ box, float ox, float oy, float oz, float dx,
                                                float dy, float dz)
{
    float tmin = (box[0] - ox) / dx;
    float tmax = (box[3] - ox) / dx;
    if (tmin > tmax) { const float t = tmin; tmin = tmax; tmax = t; }
    float tymin = (box[1] - oy) / dy;
    float tymax = (box[4] - oy) / dy;
    if (tymin > tymax) { const float t = tymin; tymin = tymax; tymax = t; }
    if (tmin > tymax || tymin > tmax) return make_float2(-1.0f, -1.0f);
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (box[2] - oz) / dz;
    float tzmax = (box[5] - oz) / dz;
    if (tzmin > tzmax) { const float t = tzmin; tzmin = tzmax; tzmax = t; }
    if (tmin > tzmax || tzmin > tmax) return make_float2(-1.0f, -1.0f);
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    return make_float2(tmin, tmax);
}

__global__ void __launch_bounds__(256)
trace_count_kernel(int num_rays, const int32_t* __restrict__ nodes, const float* __restrict__ aabbs,
                   const float* __restrict__ rays_o, const float* __restrict__ rays_d, int32_t* __restrict__ counts,
                   int* __restrict__ overflow)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= num_rays) return;
    const float ox = rays_o[3 * (size_t)r], oy = rays_o[3 * (size_t)r + 1], oz = rays_o[3 * (size_t)r + 2];
    const float dx = rays_d[3 * (size_t)r], dy = rays_d[3 * (size_t)r + 1], dz = rays_d[3 * (size_t)r + 2];
    int stack[TRACE_STACK];
    int sp = 0, count = 0;
    bool lost = false;
    stack[sp++] = 0;
    while (sp > 0) {
        const int32_t* node = nodes + 5 * (size_t)stack[--sp];
        if (node[4] <= 4) {
            count += node[4];
        } else {
            const int lid = node[1], rid = node[2];
            const float tl = slab_interval(aabbs + 6 * (size_t)lid, ox, oy, oz, dx, dy, dz).y;
            const float tr = slab_interval(aabbs + 6 * (size_t)rid, ox, oy, oz, dx, dy, dz).y;
            const int first = tl > tr ? lid : rid, second = tl > tr ? rid : lid;
            const float tf = tl > tr ? tl : tr, ts = tl > tr ? tr : tl;
            if (tf > 0) { if (sp < TRACE_STACK) stack[sp++] = first; else lost = true; }
            if (ts > 0) { if (sp < TRACE_STACK) stack[sp++] = second; else lost = true; }
        }
    }
    counts[r] = count;
    if (lost) atomicAdd(overflow, 1);
}

__global__ void __launch_bounds__(256)
trace_fill_kernel(int num_rays, const int32_t* __restrict__ nodes, const float* __restrict__ aabbs,
                  const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ means,
                  const int32_t* __restrict__ counts, const int64_t* __restrict__ offsets_inclusive,
                  uint64_t* __restrict__ keys, int32_t* __restrict__ points, float* __restrict__ positions,
                  int32_t* __restrict__ ray_ids)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= num_rays) return;
    if (counts[r] == 0) return;
    const size_t offset = r == 0 ? 0 : (size_t)offsets_inclusive[r - 1];
    const float ox = rays_o[3 * (size_t)r], oy = rays_o[3 * (size_t)r + 1], oz = rays_o[3 * (size_t)r + 2];
    const float dx = rays_d[3 * (size_t)r], dy = rays_d[3 * (size_t)r + 1], dz = rays_d[3 * (size_t)r + 2];
    int stack[TRACE_STACK];
    float2 stack_t[TRACE_STACK];
    int sp = 0, count = 0;
    stack[0] = 0;
    stack_t[0] = make_float2(-1000.f, 1000.f);
    sp = 1;
    while (sp > 0) {
        --sp;
        const int node_id = stack[sp];
        const float2 iv = stack_t[sp];
        const int32_t* node = nodes + 5 * (size_t)node_id;
        if (node[4] <= 4) {
            int stack2[8];
            int sp2 = 0;
            stack2[sp2++] = node_id;
            while (sp2 > 0) {
                const int32_t* n2 = nodes + 5 * (size_t)stack2[--sp2];
                if (n2[3] >= 0) {
                    int object_id = n2[3];
                    float t = (means[3 * (size_t)object_id] - ox) * dx + (means[3 * (size_t)object_id + 1] - oy) * dy +
                              (means[3 * (size_t)object_id + 2] - oz) * dz;
                    if (t < 0.01 || t < iv.x || t > iv.y) {
                        t = 1000000.f;
                        object_id = -1;
                    }
                    const size_t w = offset + (size_t)count;
                    keys[w] = ((uint64_t)(uint32_t)r << 32) | (uint64_t)__float_as_uint(t);
                    points[w] = object_id;
                    ray_ids[w] = r;
                    positions[3 * w] = ox + t * dx;
                    positions[3 * w + 1] = oy + t * dy;
                    positions[3 * w + 2] = oz + t * dz;
                    ++count;
                } else if (sp2 + 2 <= 8) {
                    stack2[sp2++] = n2[1];
                    stack2[sp2++] = n2[2];
                }
            }
        } else {
            const int lid = node[1], rid = node[2];
            const float2 il = slab_interval(aabbs + 6 * (size_t)lid, ox, oy, oz, dx, dy, dz);
            const float2 ir = slab_interval(aabbs + 6 * (size_t)rid, ox, oy, oz, dx, dy, dz);
            const bool lf = il.y > ir.y;
            const int first = lf ? lid : rid, second = lf ? rid : lid;
            const float2 i1 = lf ? il : ir, i2 = lf ? ir : il;
            if (i1.y > 0 && sp < TRACE_STACK) { stack[sp] = first; stack_t[sp] = i1; sp++; }
            if (i2.y > 0 && sp < TRACE_STACK) { stack[sp] = second; stack_t[sp] = i2; sp++; }
        }
    }
}

void bvh_trace_count(hipStream_t s, int num_rays, const int32_t* nodes, const float* aabbs, const float* rays_o,
                     const float* rays_d, int32_t* counts, int* overflow)
{
    if (num_rays <= 0) return;
    trace_count_kernel<<<(num_rays + 255) / 256, 256, 0, s>>>(num_rays, nodes, aabbs, rays_o, rays_d, counts, overflow);
}

void bvh_trace_fill(hipStream_t s, int num_rays, const int32_t* nodes, const float* aabbs, const float* rays_o,
                    const float* rays_d, const float* means, const int32_t* counts, const int64_t* offsets_inclusive,
                    uint64_t* keys, int32_t* points, float* positions, int32_t* ray_ids)
{
    if (num_rays <= 0) return;
    trace_fill_kernel<<<(num_rays + 255) / 256, 256, 0, s>>>(num_rays, nodes, aabbs, rays_o, rays_d, means, counts,
                                                            offsets_inclusive, keys, points, positions, ray_ids);
}

// ---- host ----
size_t bvh_build_temp_bytes(size_t P)
{
    size_t o = 0;
    auto take = [&](size_t b) { o = align_up(o + b, 256); };
    take(P * 24);        // leaf box copy
    take(P * 8);         // keys in
    take(P * 8);         // keys out
    take(P * 4);         // vals in
    take(P * 4);         // vals out
    take(P * 4 + 4);     // flags
    take(1024 * 6 * 4);  // partial boxes
    take(256);           // whole box
    take(sort_temp_bytes(P));
    return o + 256;
}

void bvh_build(hipStream_t s, int P, int32_t* nodes, float* aabbs, uint64_t* morton, void* temp)
{
    char* base = (char*)temp;
    size_t o = 0;
    auto take = [&](size_t b) { char* p = base + o; o = align_up(o + b, 256); return p; };
    float* leaf_copy = (float*)take((size_t)P * 24);
    uint64_t* k_in = (uint64_t*)take((size_t)P * 8);
    uint64_t* k_out = (uint64_t*)take((size_t)P * 8);
    uint32_t* v_in = (uint32_t*)take((size_t)P * 4);
    uint32_t* v_out = (uint32_t*)take((size_t)P * 4);
    int* flags = (int*)take((size_t)P * 4 + 4);
    float* partial = (float*)take(1024 * 6 * 4);
    float* whole = (float*)take(256);
    void* sort_temp = (void*)take(sort_temp_bytes((size_t)P));

    float* leaf = aabbs + 6 * (size_t)(P - 1);
    const int nb = min(1024, (P + 255) / 256);
    whole_box_partial_kernel<<<nb, 256, 0, s>>>(P, leaf, partial);
    whole_box_final_kernel<<<1, 64, 0, s>>>(nb, partial, whole);
    const int g = (P + 255) / 256;
    morton_kernel<<<g, 256, 0, s>>>(P, leaf, whole, k_in, v_in, leaf_copy);
    check_launch(s, false, "bvh morton");
    sort_pairs(s, (size_t)P, k_in, v_in, k_out, v_out, 30, sort_temp, false);
    scatter_leaves_kernel<<<g, 256, 0, s>>>(P, k_out, v_out, leaf_copy, aabbs, nodes, morton);
    check_launch(s, false, "bvh scatter_leaves");
    if (P > 1) {
        R3DG_HIP(hipMemsetAsync(flags, 0, (size_t)P * 4 + 4, s));
        int* nonzero_count_seen = flags + P;
        int2* ranges = reinterpret_cast<int2*>(k_in);                    // (the unsorted keys are dead after the sort)
        internal_nodes_kernel<<<(P - 1 + 255) / 256, 256, 0, s>>>(P, morton, nodes, ranges, nonzero_count_seen);
        check_launch(s, false, "bvh internal_nodes");
        // run tables in the (equally dead) unsorted-value buffer: ceil(P/256) + ceil(P/65536) boxes of 24 bytes <= 4 P bytes
        float* run1 = reinterpret_cast<float*>(v_in);
        const int n1 = (P + BOX_RUN - 1) / BOX_RUN, n2 = (n1 + BOX_RUN - 1) / BOX_RUN;
        float* run2 = run1 + 6 * (size_t)n1;
        const float* leaf_sorted = aabbs + 6 * (size_t)(P - 1);
        run_boxes_kernel<<<n1, 256, 0, s>>>(P, leaf_sorted, run1);
        run_boxes_kernel<<<n2, 256, 0, s>>>(n1, run1, run2);
        range_boxes_kernel<<<(P - 1 + 255) / 256, 256, 0, s>>>(P, ranges, leaf_sorted, run1, run2, nonzero_count_seen, nodes,
                                                              aabbs);
        check_launch(s, false, "bvh range_boxes");
        merge_boxes_kernel<<<g, 256, 0, s>>>(P, nodes, aabbs, flags, nonzero_count_seen);
        check_launch(s, false, "bvh merge_boxes");
    }
}

// Packed traversal records: 64 bytes per internal node + 64 bytes per leaf + the 8 per-XCD ray-queue heads (8 x 64 bytes).
// The buffer belongs to the CALLER (r3dg_bvh_pack_traversal / r3dg_bvh_trace_opacity_packed: one per tracer, packed once per
// set of Gaussian arrays); the reference-shaped entry point r3dg_bvh_trace_opacity packs per call into a scratch buffer that
// is private to its (device, stream) pair, so tracers on different streams or threads never share records or queue heads.
size_t bvh_trace_records_bytes(size_t P) { return (P + 8) * 128 + 8 * 64; }

void bvh_pack_traversal(hipStream_t s, int P, const int32_t* nodes, const float* aabbs, const float* means, const float* covs,
                        const float* opac, const float* normals, void* records)
{
    if (P <= 0) return;
    char* rec = reinterpret_cast<char*>(records);
    TNode* tn = reinterpret_cast<TNode*>(rec);
    TLeaf* tl = reinterpret_cast<TLeaf*>(rec + (size_t)P * 64);
    pack_traversal_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, nodes, aabbs, means, covs, opac, normals, tn, tl);
}

// the packed formulations (g_trace_packet 2, 3, 4) over records written by bvh_pack_traversal
void bvh_trace_opacity_packed(hipStream_t s, int num_rays, int P, void* records, const float* rays_o, const float* rays_d,
                              int32_t* contributes, float* out, int* overflow)
{
    if (num_rays <= 0 || P <= 0) return;
    char* rec = reinterpret_cast<char*>(records);
    TNode* tn = reinterpret_cast<TNode*>(rec);
    TLeaf* tl = reinterpret_cast<TLeaf*>(rec + (size_t)P * 64);
    int* queues = reinterpret_cast<int*>(rec + (size_t)P * 128);               // 8 x 64 bytes behind the records
    const int nblk = (num_rays + 255) / 256, chunk = (nblk + 7) / 8;
    if (opt(R3DG_OPT_TRACE_FORMULATION) >= 3 || opt(R3DG_OPT_TRACE_FORMULATION) < 2) {
        int dev = 0, cus = 256;
        R3DG_HIP(hipGetDevice(&dev));
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        cus = cus > opt(R3DG_OPT_RESERVE_CUS) ? cus - opt(R3DG_OPT_RESERVE_CUS) : 1;                    // leave CUs to a concurrent collective
        R3DG_HIP(hipMemsetAsync(queues, 0, 8 * 64, s));
        const int cap = cus * 8;                                                // 8 waves per SIMD, all resident
        const int grid = chunk * 8 < cap ? chunk * 8 : cap;
        if (opt(R3DG_OPT_TRACE_FORMULATION) == 3)
            trace_opacity_persistent_kernel<<<grid, 256, 0, s>>>(num_rays, P, tn, tl, rays_o, rays_d, contributes, out,
                                                                overflow, queues);
        else if (opt(R3DG_OPT_TRACE_COUNT_VISITS))
            trace_opacity_phased_kernel<true><<<grid, 256, 0, s>>>(num_rays, P, tn, tl, rays_o, rays_d, contributes, out,
                                                                  overflow, queues, opt(R3DG_OPT_TRACE_REFILL),
                                                                  opt(R3DG_OPT_TRACE_NODE_WEIGHT), opt(R3DG_OPT_TRACE_LEAF_WEIGHT));
        else
            trace_opacity_phased_kernel<false><<<grid, 256, 0, s>>>(num_rays, P, tn, tl, rays_o, rays_d, contributes, out,
                                                                   overflow, queues, opt(R3DG_OPT_TRACE_REFILL),
                                                                   opt(R3DG_OPT_TRACE_NODE_WEIGHT), opt(R3DG_OPT_TRACE_LEAF_WEIGHT));
    } else {
        trace_opacity_packed_kernel<<<chunk * 8, 256, 0, s>>>(num_rays, P, chunk, tn, tl, rays_o, rays_d, contributes,
                                                             out, overflow);
    }
}

// node / leaf steps of the LAST counting trace over these records (sums over the eight queue lines; synchronises the stream)
void bvh_trace_visits(hipStream_t s, int P, const void* records, unsigned long long out[2])
{
    unsigned long long lines[8 * 8];
    R3DG_HIP(hipStreamSynchronize(s));
    R3DG_HIP(hipMemcpy(lines, reinterpret_cast<const char*>(records) + (size_t)P * 128, sizeof(lines), hipMemcpyDeviceToHost));
    out[0] = out[1] = 0ull;
    for (int q = 0; q < 8; ++q) {
        out[0] += lines[8 * q + 4];
        out[1] += lines[8 * q + 5];
    }
}

// scratch records of the reference-shaped entry point: one grow-only buffer per (device, stream) (common.hpp stream_scratch)
static void* trace_scratch(hipStream_t s, size_t bytes) { return stream_scratch(s, 2, bytes); }

// P = number of Gaussians (rows of means / leaves of the tree); P <= 0: unknown -> the round-1 kernel
void bvh_trace_opacity(hipStream_t s, int num_rays, int P, const int32_t* nodes, const float* aabbs, const float* rays_o,
                       const float* rays_d, const float* means, const float* covs, const float* opac,
                       const float* normals, int32_t* contributes, float* out, int* overflow)
{
    if (num_rays <= 0) return;
    if (opt(R3DG_OPT_TRACE_FORMULATION) >= 2 && P > 0) {
        void* rec = trace_scratch(s, bvh_trace_records_bytes((size_t)P));
        bvh_pack_traversal(s, P, nodes, aabbs, means, covs, opac, normals, rec);
        bvh_trace_opacity_packed(s, num_rays, P, rec, rays_o, rays_d, contributes, out, overflow);
    } else if (opt(R3DG_OPT_TRACE_FORMULATION) == 1)
        trace_opacity_packet_kernel<<<(num_rays + 255) / 256, 256, 0, s>>>(num_rays, nodes, aabbs, rays_o, rays_d, means,
                                                                          covs, opac, normals, contributes, out, overflow);
    else
        trace_opacity_kernel<<<(num_rays + 255) / 256, 256, 0, s>>>(num_rays, nodes, aabbs, rays_o, rays_d, means, covs,
                                                                   opac, normals, contributes, out, overflow);
}

}  // namespace r3dg
