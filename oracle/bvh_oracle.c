/*
 * oracle/bvh_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference LBVH build and visibility trace (NJU-3DV/Relightable3DGaussian bvh/):
 *   bvho_build              <- construct_bvh, bvh/src/construct.cu:147-265 (+ helpers :7-145)
 *   bvho_trace_opacity      <- trace_bvh_opacity_cuda, bvh/src/trace.cu:196-286 (+ utility.cuh:35-110)
 *   bvho_trace_bruteforce   <- SURVEY.md Appendix B: the same value without the tree (every Gaussian whose LEAF box
 *                              passes the slab test with tmax > 0 and the accept chain; product < 0.9 -> 0)
 * "parity unpinned by the reference": bvh/ ships no tests or golden vectors and is CUDA-only; the Python half of
 * RayTracer.__init__ (leaf boxes) IS pinned by tests/golden/bvh_leaf_reference.npz.  The tree-free brute force
 * cross-checks the traversal restatement.
 * Arithmetic: fp32, reference operation order, -ffp-contract=off (node tables, boxes and Morton codes are
 * compared bit-for-bit with the HIP build).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static uint32_t expand_bits(uint32_t v) /* construct.cu:7-15 */
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
static int clz64(uint64_t x) { return x == 0 ? 64 : __builtin_clzll(x); }
static int common_upper_bits(uint64_t a, uint64_t b) { return clz64(a ^ b); } /* construct.cu:17-21 */

static uint32_t morton_code(float x, float y, float z) /* construct.cu:23-33, resolution 1024 */
{
    const float res = 1024.0f;
    x = fminf(fmaxf(x * res, 0.0f), res - 1.0f);
    y = fminf(fmaxf(y * res, 0.0f), res - 1.0f);
    z = fminf(fmaxf(z * res, 0.0f), res - 1.0f);
    const uint32_t xx = expand_bits((uint32_t)x), yy = expand_bits((uint32_t)y), zz = expand_bits((uint32_t)z);
    return xx * 4 + yy * 2 + zz;
}

/* construct.cu:54-114 */
static void determine_range(const uint64_t* code, uint32_t num_leaves, uint32_t idx, uint32_t* o0, uint32_t* o1)
{
    if (idx == 0) { *o0 = 0; *o1 = num_leaves - 1; return; }
    const uint64_t self = code[idx];
    const int L_delta = common_upper_bits(self, code[idx - 1]);
    const int R_delta = common_upper_bits(self, code[idx + 1]);
    const int d = (R_delta > L_delta) ? 1 : -1;
    const int delta_min = L_delta < R_delta ? L_delta : R_delta;
    int l_max = 2;
    int delta = -1;
    int i_tmp = (int)idx + d * l_max;
    if (0 <= i_tmp && i_tmp < (int)num_leaves) delta = common_upper_bits(self, code[i_tmp]);
    while (delta > delta_min) {
        l_max <<= 1;
        i_tmp = (int)idx + d * l_max;
        delta = -1;
        if (0 <= i_tmp && i_tmp < (int)num_leaves) delta = common_upper_bits(self, code[i_tmp]);
    }
    int l = 0;
    int t = l_max >> 1;
    while (t > 0) {
        i_tmp = (int)idx + (l + t) * d;
        delta = -1;
        if (0 <= i_tmp && i_tmp < (int)num_leaves) delta = common_upper_bits(self, code[i_tmp]);
        if (delta > delta_min) l += t;
        t >>= 1;
    }
    uint32_t jdx = idx + l * d;
    if (d < 0) { uint32_t tmp = idx; idx = jdx; jdx = tmp; }
    *o0 = idx;
    *o1 = jdx;
}

/* construct.cu:116-145 */
static int32_t find_split(const uint64_t* code, int32_t first, int32_t last)
{
    const uint64_t first_code = code[first], last_code = code[last];
    if (first_code == last_code) return (first + last) >> 1;
    const int32_t delta_node = common_upper_bits(first_code, last_code);
    int32_t split = first;
    int32_t stride = last - first;
    do {
        stride = (stride + 1) >> 1;
        const int middle = split + stride;
        if (middle < last) {
            const int32_t delta = common_upper_bits(first_code, code[middle]);
            if (delta > delta_node) split = middle;
        }
    } while (stride > 1);
    return split;
}

typedef struct { uint32_t m; uint32_t idx; float box[6]; } leaf_t;

/* nodes int32[2P-1][5] = {parent,left,right,object_id,leaf_count} pre-initialised by the caller exactly like
 * bvh/__init__.py:31-33 (-1 everywhere, count 0 internal / 1 leaf); aabbs float[2P-1][6] = {lower xyz, upper xyz}
 * with the leaf rows P-1.. pre-filled. Both are updated in place; morton[P] receives the 64-bit codes. */
void bvho_build(int32_t P, int32_t* nodes, float* aabbs, uint64_t* morton)
{
    const int32_t ni = P - 1;
    float whole[6] = {100000.f, 100000.f, 100000.f, -100000.f, -100000.f, -100000.f};
    for (int i = 0; i < P; i++) {
        const float* b = aabbs + (size_t)(ni + i) * 6;
        for (int a = 0; a < 3; a++) {
            whole[a] = fminf(whole[a], b[a]);
            whole[3 + a] = fmaxf(whole[3 + a], b[3 + a]);
        }
    }
    leaf_t* lv = (leaf_t*)malloc(sizeof(leaf_t) * (size_t)P);
    for (int i = 0; i < P; i++) {
        const float* b = aabbs + (size_t)(ni + i) * 6;
        /* centroid (utility.cuh:12-20: (upper+lower)*0.5 with a double literal == exact fp32 halving) */
        float c[3];
        for (int a = 0; a < 3; a++) {
            c[a] = (float)((b[3 + a] + b[a]) * 0.5);
            c[a] -= whole[a];
            c[a] /= (whole[3 + a] - whole[a]);
        }
        lv[i].m = morton_code(c[0], c[1], c[2]);
        lv[i].idx = (uint32_t)i;
        memcpy(lv[i].box, b, sizeof(float) * 6);
    }
    /* stable sort by the 32-bit code (thrust::stable_sort_by_key, construct.cu:179-182): LSD counting sort */
    leaf_t* tmp = (leaf_t*)malloc(sizeof(leaf_t) * (size_t)P);
    for (int shift = 0; shift < 32; shift += 8) {
        size_t count[257] = {0};
        for (int i = 0; i < P; i++) count[((lv[i].m >> shift) & 255u) + 1]++;
        for (int d = 0; d < 256; d++) count[d + 1] += count[d];
        for (int i = 0; i < P; i++) tmp[count[(lv[i].m >> shift) & 255u]++] = lv[i];
        leaf_t* t = lv; lv = tmp; tmp = t;
    }
    for (int i = 0; i < P; i++) {
        memcpy(aabbs + (size_t)(ni + i) * 6, lv[i].box, sizeof(float) * 6);
        uint64_t m64 = lv[i].m;
        m64 <<= 31;                       /* sic: 31 (construct.cu:188) */
        m64 |= lv[i].idx;
        morton[i] = m64;
        nodes[(size_t)(ni + i) * 5 + 3] = (int32_t)lv[i].idx;   /* construct.cu:196-201 */
    }
    free(lv);
    free(tmp);
    for (int32_t idx = 0; idx < ni; idx++) {                     /* construct.cu:203-229 */
        int32_t* node = nodes + (size_t)idx * 5;
        node[3] = -1;
        uint32_t r0, r1;
        determine_range(morton, (uint32_t)P, (uint32_t)idx, &r0, &r1);
        const int32_t gamma = find_split(morton, (int32_t)r0, (int32_t)r1);
        node[1] = gamma;
        node[2] = gamma + 1;
        if ((r0 < r1 ? r0 : r1) == (uint32_t)gamma) node[1] += P - 1;
        if ((r0 > r1 ? r0 : r1) == (uint32_t)(gamma + 1)) node[2] += P - 1;
        nodes[(size_t)node[1] * 5] = idx;
        nodes[(size_t)node[2] * 5] = idx;
    }
    int* flags = (int*)calloc((size_t)(ni > 0 ? ni : 1), sizeof(int));       /* construct.cu:231-264 */
    for (int32_t idx = ni; idx < 2 * P - 1; idx++) {
        int32_t num = 1;
        int32_t parent = nodes[(size_t)idx * 5];
        while (parent != -1) {
            nodes[(size_t)parent * 5 + 4] += num;
            if (flags[parent] == 0) { flags[parent] = 1; break; }
            int32_t* pn = nodes + (size_t)parent * 5;
            const float* lb = aabbs + (size_t)pn[1] * 6;
            const float* rb = aabbs + (size_t)pn[2] * 6;
            float* pb = aabbs + (size_t)parent * 6;
            for (int a = 0; a < 3; a++) {
                pb[a] = fminf(lb[a], rb[a]);
                pb[3 + a] = fmaxf(lb[3 + a], rb[3 + a]);
            }
            num = pn[4];
            parent = pn[0];
        }
    }
    free(flags);
}

/* utility.cuh:35-82; returns tmax (and tmin through *tmin_out), (-1,-1) on a miss */
static float slab(const float* box, const float* o, const float* d, float* tmin_out)
{
    float tmin = (box[0] - o[0]) / d[0];
    float tmax = (box[3] - o[0]) / d[0];
    if (tmin > tmax) { float t = tmin; tmin = tmax; tmax = t; }
    float tymin = (box[1] - o[1]) / d[1];
    float tymax = (box[4] - o[1]) / d[1];
    if (tymin > tymax) { float t = tymin; tymin = tymax; tymax = t; }
    if (tmin > tymax || tymin > tmax) { *tmin_out = -1.0f; return -1.0f; }
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (box[2] - o[2]) / d[2];
    float tzmax = (box[5] - o[2]) / d[2];
    if (tzmin > tzmax) { float t = tzmin; tzmin = tzmax; tzmax = t; }
    if (tmin > tzmax || tzmin > tmax) { *tmin_out = -1.0f; return -1.0f; }
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    *tmin_out = tmin;
    return tmax;
}

/* leaf accept chain, trace.cu:223-249; returns 1 and *alpha when the Gaussian attenuates the ray */
static int leaf_alpha(int32_t g, const float* o, const float* d, const float* means, const float* covs,
                      const float* opac, const float* normals, float* alpha)
{
    if (opac[g] < 1.f / 255.f) return 0;
    const float* n = normals + 3 * (size_t)g;
    if (n[0] * d[0] + n[1] * d[1] + n[2] * d[2] > 0) return 0;
    const float* mu = means + 3 * (size_t)g;
    const float* ci = covs + 6 * (size_t)g;
    /* utility.cuh:90-100 */
    const float m0 = mu[0] - o[0], m1 = mu[1] - o[1], m2 = mu[2] - o[2];
    const float t1 = ci[0] * m0 * d[0] + ci[1] * m0 * d[1] + ci[2] * m0 * d[2] + ci[1] * m1 * d[0] + ci[3] * m1 * d[1] +
                     ci[4] * m1 * d[2] + ci[2] * m2 * d[0] + ci[4] * m2 * d[1] + ci[5] * m2 * d[2];
    const float t2 = ci[0] * d[0] * d[0] + ci[1] * d[0] * d[1] + ci[2] * d[0] * d[2] + ci[1] * d[1] * d[0] +
                     ci[3] * d[1] * d[1] + ci[4] * d[1] * d[2] + ci[2] * d[2] * d[0] + ci[4] * d[2] * d[1] +
                     ci[5] * d[2] * d[2];
    const float t = t1 / t2;
    if (t < 0.01) return 0;                     /* double literal in the reference: t promoted, same outcome */
    const float p0 = o[0] + t * d[0], p1 = o[1] + t * d[1], p2 = o[2] + t * d[2];
    /* utility.cuh:102-109 */
    const float f0 = mu[0] - p0, f1 = mu[1] - p1, f2 = mu[2] - p2;
    /* evaluation order of the expression as written: products in float ('2*x' is int*float -> float), sum in
       float, then times the double literal -0.5 and rounded back to float (an exact halving) */
    const float s = f0 * f0 * ci[0] + f1 * f1 * ci[3] + f2 * f2 * ci[5] + 2 * f0 * f1 * ci[1] + 2 * f0 * f2 * ci[2] +
                    2 * f1 * f2 * ci[4];
    const float power = (float)(-0.5 * (double)s);
    if (power > 0) return 0;
    *alpha = opac[g] * expf(power);
    return 1;
}

#define BVHO_STACK 64

void bvho_trace_opacity(int32_t num_rays, int32_t P, const int32_t* nodes, const float* aabbs, const float* rays_o,
                        const float* rays_d, const float* means, const float* covs, const float* opac,
                        const float* normals, int32_t* contributes, float* out_opacity)
{
    (void)P;
    for (int32_t r = 0; r < num_rays; r++) {
        contributes[r] = 0;          /* bvh.cu:101-102: outputs start at (0, 1) */
        out_opacity[r] = 1.0f;
        const float* o = rays_o + 3 * (size_t)r;
        const float* d = rays_d + 3 * (size_t)r;
        int32_t stack[BVHO_STACK];
        int sp = 0;
        stack[sp++] = 0;
        int32_t count = 0;
        float T = 1.0f;
        int killed = 0;
        while (sp > 0 && !killed) {
            const int32_t node_id = stack[--sp];
            const int32_t* node = nodes + (size_t)node_id * 5;
            if (node[4] <= 1) {
                float alpha;
                if (leaf_alpha(node[3], o, d, means, covs, opac, normals, &alpha)) {
                    count += 1;
                    T *= 1 - alpha;
                    if (T < 0.9) {              /* trace.cu:251-254: write 0 and return (count stays 0) */
                        out_opacity[r] = 0.0f;
                        killed = 1;
                    }
                }
            } else {
                const int32_t lid = node[1], rid = node[2];
                float tl0, tr0;
                const float tl = slab(aabbs + (size_t)lid * 6, o, d, &tl0);
                const float tr = slab(aabbs + (size_t)rid * 6, o, d, &tr0);
                if (tl > tr) {
                    if (tl > 0 && sp < BVHO_STACK) stack[sp++] = lid;
                    if (tr > 0 && sp < BVHO_STACK) stack[sp++] = rid;
                } else {
                    if (tr > 0 && sp < BVHO_STACK) stack[sp++] = rid;
                    if (tl > 0 && sp < BVHO_STACK) stack[sp++] = lid;
                }
            }
        }
        if (!killed) {
            contributes[r] = count;
            out_opacity[r] = T;
        }
    }
}

/* Tree-free value: product over every Gaussian whose LEAF box is hit (tmax>0) and that passes the accept chain,
 * in leaf (Morton) order; < 0.9 -> 0.  leaf_boxes = aabbs + (P-1)*6, leaf_ids = object ids of the sorted leaves. */
void bvho_trace_bruteforce(int32_t num_rays, int32_t P, const int32_t* nodes, const float* aabbs, const float* rays_o,
                           const float* rays_d, const float* means, const float* covs, const float* opac,
                           const float* normals, int32_t* contributes, double* product)
{
    for (int32_t r = 0; r < num_rays; r++) {
        const float* o = rays_o + 3 * (size_t)r;
        const float* d = rays_d + 3 * (size_t)r;
        double T = 1.0;
        int32_t count = 0;
        for (int32_t j = 0; j < P; j++) {
            const int32_t row = P - 1 + j;
            if (P > 1) {
                float t0;
                if (!(slab(aabbs + (size_t)row * 6, o, d, &t0) > 0)) continue;
            }
            float alpha;
            if (leaf_alpha(nodes[(size_t)row * 5 + 3], o, d, means, covs, opac, normals, &alpha)) {
                count++;
                T *= 1.0 - (double)alpha;
            }
        }
        contributes[r] = count;
        product[r] = T;
    }
}
