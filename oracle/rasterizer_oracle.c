/*
 * oracle/rasterizer_oracle.c -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
 *
 * Scalar CPU restatement of the reference tile rasterizer (NJU-3DV/Relightable3DGaussian,
 * r3dg-rasterization/cuda_rasterizer).  Each function cites the reference file:line it follows.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this file.
 *
 * PARITY STATUS: "parity unpinned by the reference" -- the reference ships no golden vectors or
 * tests for this path and its CUDA extension cannot be built in this image (no nvcc).  The
 * restatement is pinned instead against (a) the reference's own *Python* cross-implementations
 * imported from /root/reference (eval_sh, build_rotation/strip_symmetric covariance; see
 * tests/golden/make_golden.py) and (b) an independent pure-PyTorch restatement whose autograd
 * provides the backward (oracle/torch_rasterizer.py).
 *
 * Arithmetic: per-pair / per-Gaussian math is fp32 in the reference's operation order and is
 * compiled with -ffp-contract=off; sums that the reference forms with float atomics
 * (out_weights, every dL_d* accumulator) are formed here in double so that the oracle is the
 * order-independent "true" sum the GPU float atomics are compared against.
 *
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off -fno-fast-math -o liboracle.so *.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BLOCK_X 16 /* reference config.h:16 */
#define BLOCK_Y 16 /* reference config.h:17 */
#define BLOCK_SIZE (BLOCK_X * BLOCK_Y)

/* reference auxiliary.h:22-39 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

/* ---- tiny column-major 3x3 with glm's multiplication order (third_party/glm: mat3*mat3 is
 * Result[j][i] = A[0][i]*B[j][0] + A[1][i]*B[j][1] + A[2][i]*B[j][2], summed left to right) ---- */
typedef struct { float c[3][3]; } mat3; /* c[col][row] */

static mat3 m3_mul(const mat3* A, const mat3* B)
{
    mat3 R;
    for (int j = 0; j < 3; j++)
        for (int i = 0; i < 3; i++)
            R.c[j][i] = A->c[0][i] * B->c[j][0] + A->c[1][i] * B->c[j][1] + A->c[2][i] * B->c[j][2];
    return R;
}
static mat3 m3_T(const mat3* A)
{
    mat3 R;
    for (int j = 0; j < 3; j++)
        for (int i = 0; i < 3; i++) R.c[j][i] = A->c[i][j];
    return R;
}
/* glm::mat3(a..i) constructor fills columns */
static mat3 m3_cols(float a, float b, float c, float d, float e, float f, float g, float h, float i)
{
    mat3 R = {{{a, b, c}, {d, e, f}, {g, h, i}}};
    return R;
}

/* GPU float->int conversion saturates (CUDA cvt.rzi.s32.f32 and gfx950 v_cvt_i32_f32 alike); x86
 * cvttss2si does not, so restate the saturation explicitly. NaN -> 0. */
static int sat_f2i(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* reference auxiliary.h:41-44 : evaluated in double, rounded once to float */
static float ndc2Pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

/* reference auxiliary.h:46-56 */
static void getRect(float px, float py, int max_radius, int gx, int gy, int* rmin, int* rmax)
{
    rmin[0] = imin(gx, imax(0, sat_f2i((px - max_radius) / BLOCK_X)));
    rmin[1] = imin(gy, imax(0, sat_f2i((py - max_radius) / BLOCK_Y)));
    rmax[0] = imin(gx, imax(0, sat_f2i((px + max_radius + BLOCK_X - 1) / BLOCK_X)));
    rmax[1] = imin(gy, imax(0, sat_f2i((py + max_radius + BLOCK_Y - 1) / BLOCK_Y)));
}

/* reference auxiliary.h:58-77 */
static void transformPoint4x3(const float* p, const float* m, float* o)
{
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static void transformPoint4x4(const float* p, const float* m, float* o)
{
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* reference forward.cu:119-153 (quaternion used as given, NOT normalised: forward.cu:128) */
static void computeCov3D(const float* scale, float mod, const float* rot, float* cov3D)
{
    mat3 S = m3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
    S.c[0][0] = mod * scale[0];
    S.c[1][1] = mod * scale[1];
    S.c[2][2] = mod * scale[2];
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    mat3 R = m3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                     2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                     2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    mat3 M = m3_mul(&S, &R);
    mat3 Mt = m3_T(&M);
    mat3 Sigma = m3_mul(&Mt, &M);
    cov3D[0] = Sigma.c[0][0];
    cov3D[1] = Sigma.c[0][1];
    cov3D[2] = Sigma.c[0][2];
    cov3D[3] = Sigma.c[1][1];
    cov3D[4] = Sigma.c[1][2];
    cov3D[5] = Sigma.c[2][2];
}

/* shared by forward.cu:74-113 and backward.cu:160-198 */
typedef struct {
    mat3 T, W, Vrk;
    float t[3], txtz, tytz, limx, limy;
    float a, b, c; /* cov2D incl. +0.3 */
} cov2d_t;

static void cov2d_common(const float* mean, float fx, float fy, float tan_fovx, float tan_fovy,
                         const float* cov3D, const float* vm, cov2d_t* o)
{
    transformPoint4x3(mean, vm, o->t);
    o->limx = 1.3f * tan_fovx;
    o->limy = 1.3f * tan_fovy;
    o->txtz = o->t[0] / o->t[2];
    o->tytz = o->t[1] / o->t[2];
    o->t[0] = fminf(o->limx, fmaxf(-o->limx, o->txtz)) * o->t[2];
    o->t[1] = fminf(o->limy, fmaxf(-o->limy, o->tytz)) * o->t[2];
    const float* t = o->t;
    mat3 J = m3_cols(fx / t[2], 0.0f, -(fx * t[0]) / (t[2] * t[2]),
                     0.0f, fy / t[2], -(fy * t[1]) / (t[2] * t[2]),
                     0, 0, 0);
    o->W = m3_cols(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
    o->T = m3_mul(&o->W, &J);
    o->Vrk = m3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    mat3 Tt = m3_T(&o->T), Vt = m3_T(&o->Vrk);
    mat3 tmp = m3_mul(&Tt, &Vt);
    mat3 cov = m3_mul(&tmp, &o->T);
    cov.c[0][0] += 0.3f;
    cov.c[1][1] += 0.3f;
    o->a = cov.c[0][0];
    o->b = cov.c[0][1];
    o->c = cov.c[1][1];
}

/* reference forward.cu:20-71 */
static void computeColorFromSH(int idx, int deg, int max_coeffs, const float* means, const float* campos,
                               const float* shs, uint8_t* clamped, float* out)
{
    float dir[3] = {means[3 * idx] - campos[0], means[3 * idx + 1] - campos[1], means[3 * idx + 2] - campos[2]};
    float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    dir[0] = dir[0] / len;
    dir[1] = dir[1] / len;
    dir[2] = dir[2] / len;
    const float* sh = shs + (size_t)idx * max_coeffs * 3;
    for (int ch = 0; ch < 3; ch++) {
#define SH(k) sh[(k) * 3 + ch]
        float result = SH_C0 * SH(0);
        if (deg > 0) {
            float x = dir[0], y = dir[1], z = dir[2];
            result = result - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z;
                float xy = x * y, yz = y * z, xz = x * z;
                result = result + SH_C2[0] * xy * SH(4) + SH_C2[1] * yz * SH(5) +
                         SH_C2[2] * (2.0f * zz - xx - yy) * SH(6) + SH_C2[3] * xz * SH(7) +
                         SH_C2[4] * (xx - yy) * SH(8);
                if (deg > 2) {
                    result = result + SH_C3[0] * y * (3.0f * xx - yy) * SH(9) + SH_C3[1] * xy * z * SH(10) +
                             SH_C3[2] * y * (4.0f * zz - xx - yy) * SH(11) +
                             SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
                             SH_C3[4] * x * (4.0f * zz - xx - yy) * SH(13) + SH_C3[5] * z * (xx - yy) * SH(14) +
                             SH_C3[6] * x * (xx - 3.0f * yy) * SH(15);
                }
            }
        }
#undef SH
        result += 0.5f;
        clamped[3 * idx + ch] = (result < 0);
        out[ch] = fmaxf(result, 0.0f);
    }
}

/* reference forward.cu:156-258 (preprocessCUDA) + auxiliary.h:139-164 (in_frustum).
 * NULL pointers play the role of the reference's nullptr optionals. */
void r3dgo_preprocess(int P, int D, int M, const float* means3D, const float* scales, float scale_modifier,
                      const float* rotations, const float* opacities, const float* shs,
                      const float* cov3D_precomp, const float* colors_precomp, const float* viewmatrix,
                      const float* projmatrix, const float* cam_pos, int W, int H, float tan_fovx,
                      float tan_fovy, int32_t* radii, float* means2D, float* depths, float* cov3Ds, float* rgb,
                      float* conic_opacity, uint32_t* tiles_touched, uint8_t* clamped)
{
    const float focal_y = H / (2.0f * tan_fovy); /* rasterizer_impl.cu:232-233 */
    const float focal_x = W / (2.0f * tan_fovx);
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    for (int idx = 0; idx < P; idx++) {
        radii[idx] = 0;
        tiles_touched[idx] = 0;
        const float* p_orig = means3D + 3 * idx;
        float p_view[3];
        transformPoint4x3(p_orig, viewmatrix, p_view);
        if (p_view[2] <= 0.2f) continue; /* auxiliary.h:154 */
        float p_hom[4];
        transformPoint4x4(p_orig, projmatrix, p_hom);
        float p_w = 1.0f / (p_hom[3] + 0.0000001f);
        float p_proj[3] = {p_hom[0] * p_w, p_hom[1] * p_w, p_hom[2] * p_w};
        const float* cov3D;
        if (cov3D_precomp) {
            cov3D = cov3D_precomp + idx * 6;
        } else {
            computeCov3D(scales + 3 * idx, scale_modifier, rotations + 4 * idx, cov3Ds + idx * 6);
            cov3D = cov3Ds + idx * 6;
        }
        cov2d_t c2;
        cov2d_common(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, &c2);
        float cx = c2.a, cy = c2.b, cz = c2.c;
        float det = (cx * cz - cy * cy);
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float conic[3] = {cz * det_inv, -cy * det_inv, cx * det_inv};
        float mid = 0.5f * (cx + cz);
        float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        float pix[2] = {ndc2Pix(p_proj[0], W), ndc2Pix(p_proj[1], H)};
        int rmin[2], rmax[2];
        getRect(pix[0], pix[1], sat_f2i(my_radius), gx, gy, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;
        if (!colors_precomp) {
            float res[3];
            computeColorFromSH(idx, D, M, means3D, cam_pos, shs, clamped, res);
            rgb[idx * 3 + 0] = res[0];
            rgb[idx * 3 + 1] = res[1];
            rgb[idx * 3 + 2] = res[2];
        }
        depths[idx] = p_view[2];
        radii[idx] = sat_f2i(my_radius);
        means2D[2 * idx] = pix[0];
        means2D[2 * idx + 1] = pix[1];
        conic_opacity[4 * idx + 0] = conic[0];
        conic_opacity[4 * idx + 1] = conic[1];
        conic_opacity[4 * idx + 2] = conic[2];
        conic_opacity[4 * idx + 3] = opacities[idx];
        tiles_touched[idx] = (uint32_t)((rmax[1] - rmin[1]) * (rmax[0] - rmin[0]));
    }
}

/* reference rasterizer_impl.cu:54-66,141-153 (markVisible) */
void r3dgo_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present)
{
    for (int i = 0; i < P; i++) {
        float pv[3];
        transformPoint4x3(means3D + 3 * i, viewmatrix, pv);
        present[i] = pv[2] > 0.2f;
    }
}

/* reference rasterizer_impl.cu:35-50 */
uint32_t r3dgo_getHigherMsb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step;
        else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

/* Inclusive scan of tiles_touched (rasterizer_impl.cu:287) -> returns num_rendered */
int64_t r3dgo_scan(int P, const uint32_t* tiles_touched, uint32_t* point_offsets)
{
    uint32_t s = 0;
    for (int i = 0; i < P; i++) {
        s += tiles_touched[i];
        point_offsets[i] = s;
    }
    return P > 0 ? (int64_t)s : 0;
}

/* reference rasterizer_impl.cu:70-111 (duplicateWithKeys) */
void r3dgo_duplicate_with_keys(int P, const float* means2D, const float* depths, const uint32_t* offsets,
                               const int32_t* radii, int W, int H, uint64_t* keys, uint32_t* values)
{
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    for (int idx = 0; idx < P; idx++) {
        if (radii[idx] > 0) {
            uint32_t off = (idx == 0) ? 0 : offsets[idx - 1];
            int rmin[2], rmax[2];
            getRect(means2D[2 * idx], means2D[2 * idx + 1], radii[idx], gx, gy, rmin, rmax);
            uint32_t dbits;
            memcpy(&dbits, &depths[idx], 4);
            for (int y = rmin[1]; y < rmax[1]; y++)
                for (int x = rmin[0]; x < rmax[0]; x++) {
                    uint64_t key = (uint64_t)(y * gx + x);
                    key <<= 32;
                    key |= dbits;
                    keys[off] = key;
                    values[off] = (uint32_t)idx;
                    off++;
                }
        }
    }
}

/* cub::DeviceRadixSort::SortPairs semantics (rasterizer_impl.cu:313-318): STABLE ascending sort on
 * key bits [0, end_bit).  Restated as an LSD byte-wise counting sort. */
void r3dgo_sort_pairs(int64_t n, const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out,
                      uint32_t* vals_out, int end_bit)
{
    if (n <= 0) return;
    uint64_t* ka = (uint64_t*)malloc(sizeof(uint64_t) * n);
    uint64_t* kb = (uint64_t*)malloc(sizeof(uint64_t) * n);
    uint32_t* va = (uint32_t*)malloc(sizeof(uint32_t) * n);
    uint32_t* vb = (uint32_t*)malloc(sizeof(uint32_t) * n);
    memcpy(ka, keys_in, sizeof(uint64_t) * n);
    memcpy(va, vals_in, sizeof(uint32_t) * n);
    for (int shift = 0; shift < end_bit; shift += 8) {
        int bits = end_bit - shift < 8 ? end_bit - shift : 8;
        uint32_t mask = (1u << bits) - 1;
        int64_t count[257] = {0};
        for (int64_t i = 0; i < n; i++) count[((ka[i] >> shift) & mask) + 1]++;
        for (int d = 0; d < 256; d++) count[d + 1] += count[d];
        for (int64_t i = 0; i < n; i++) {
            int64_t pos = count[(ka[i] >> shift) & mask]++;
            kb[pos] = ka[i];
            vb[pos] = va[i];
        }
        uint64_t* tk = ka; ka = kb; kb = tk;
        uint32_t* tv = va; va = vb; vb = tv;
    }
    memcpy(keys_out, ka, sizeof(uint64_t) * n);
    memcpy(vals_out, va, sizeof(uint32_t) * n);
    free(ka); free(kb); free(va); free(vb);
}

/* reference rasterizer_impl.cu:116-138,320 (memset + identifyTileRanges); ranges is uint2[T] */
void r3dgo_identify_tile_ranges(int64_t L, const uint64_t* keys, int T, uint32_t* ranges)
{
    memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)T);
    for (int64_t idx = 0; idx < L; idx++) {
        uint32_t currtile = (uint32_t)(keys[idx] >> 32);
        if (idx == 0) ranges[2 * currtile] = 0;
        else {
            uint32_t prevtile = (uint32_t)(keys[idx - 1] >> 32);
            if (currtile != prevtile) {
                ranges[2 * prevtile + 1] = (uint32_t)idx;
                ranges[2 * currtile] = (uint32_t)idx;
            }
        }
        if (idx == L - 1) ranges[2 * currtile + 1] = (uint32_t)L;
    }
}

/* reference forward.cu:263-395 (renderCUDA forward), restated per pixel.
 * `margin` (optional, may be NULL) records per pixel the smallest relative distance of any
 * threshold decision (alpha vs 1/255, test_T vs 1e-4, power vs 0) from its threshold, so that a
 * parity test can exempt pixels whose discrete outcome (n_contrib) legitimately depends on the
 * last ulp of exp(). */
void r3dgo_render_forward(int W, int H, int S, const uint32_t* ranges, const uint32_t* point_list,
                          const float* means2D, const float* depths, const float* features, const float* colors,
                          const float* conic_opacity, const float* bg_color, float* final_T, uint32_t* n_contrib,
                          float* out_color, float* out_opacity, float* out_depth, float* out_feature,
                          double* out_weights /* [P] accumulated in double */, float* margin)
{
    const int gx = (W + BLOCK_X - 1) / BLOCK_X;
    const size_t HW = (size_t)H * W;
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            const int tile = (py / BLOCK_Y) * gx + (px / BLOCK_X);
            const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
            const size_t pix_id = (size_t)W * py + px;
            const float pixf[2] = {(float)px, (float)py};
            float T = 1.0f;
            uint32_t contributor = 0, last_contributor = 0;
            float C[3] = {0, 0, 0}, F[64], Depth = 0, Opacity = 0;
            float mrg = 1e30f;
            for (int ch = 0; ch < S; ch++) F[ch] = 0;
            for (uint32_t k = r0; k < r1; k++) {
                contributor++;
                const uint32_t g = point_list[k];
                float dx = means2D[2 * g] - pixf[0], dy = means2D[2 * g + 1] - pixf[1];
                const float* con_o = conic_opacity + 4 * g;
                float power = -0.5f * (con_o[0] * dx * dx + con_o[2] * dy * dy) - con_o[1] * dx * dy;
                if (power > 0.0f) continue;
                float alpha = fminf(0.99f, con_o[3] * expf(power));
                float m1 = fabsf(alpha - 1.0f / 255.0f) * 255.0f;
                if (m1 < mrg) mrg = m1;
                if (alpha < 1.0f / 255.0f) continue;
                float test_T = T * (1 - alpha);
                float m2 = fabsf(test_T - 0.0001f) * 10000.0f;
                if (m2 < mrg) mrg = m2;
                if (test_T < 0.0001f) break; /* done = true: nothing later can contribute */
                float weight = alpha * T;
                for (int ch = 0; ch < 3; ch++) C[ch] += colors[g * 3 + ch] * weight;
                for (int ch = 0; ch < S; ch++) F[ch] += features[(size_t)g * S + ch] * weight;
                Depth += depths[g] * weight;
                Opacity += weight;
                T = test_T;
                out_weights[g] += (double)weight;
                last_contributor = contributor;
            }
            final_T[pix_id] = T;
            n_contrib[pix_id] = last_contributor;
            for (int ch = 0; ch < 3; ch++) out_color[ch * HW + pix_id] = C[ch] + T * bg_color[ch];
            for (int ch = 0; ch < S; ch++) out_feature[ch * HW + pix_id] = F[ch];
            out_depth[pix_id] = Depth;
            out_opacity[pix_id] = Opacity;
            if (margin) margin[pix_id] = mrg;
        }
}

/* reference forward.cu:398-425 (renderSurfaceXYZCUDA) then forward.cu:427-491 (renderPseudoNormalCUDA) */
void r3dgo_pseudo_normal(int W, int H, const float* viewmatrix, float focal_x, float focal_y, float cx, float cy,
                         const float* opacities, const float* depths, float* normals, float* surface_xyz)
{
    const size_t HW = (size_t)H * W;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            size_t id = (size_t)W * y + x;
            float depth = depths[id] / fmaxf(opacities[id], 0.0000001f);
            surface_xyz[id] = (x - cx) / focal_x * depth;
            surface_xyz[HW + id] = (y - cy) / focal_y * depth;
            surface_xyz[2 * HW + id] = depth;
        }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int ym = y == 0 ? 0 : y - 1, yp = y == H - 1 ? H - 1 : y + 1;
            int xm = x == 0 ? 0 : x - 1, xp = x == W - 1 ? W - 1 : x + 1;
            size_t i00 = (size_t)W * ym + xm, i01 = (size_t)W * ym + x, i02 = (size_t)W * ym + xp;
            size_t i10 = (size_t)W * y + xm, i11 = (size_t)W * y + x, i12 = (size_t)W * y + xp;
            size_t i20 = (size_t)W * yp + xm, i21 = (size_t)W * yp + x, i22 = (size_t)W * yp + xp;
            float ga[3], gb[3];
            for (int i = 0; i < 3; i++) {
                const float* s = surface_xyz + i * HW;
                ga[i] = -0.125f * s[i00] + 0.125f * s[i02] - 0.25f * s[i10] + 0.25f * s[i12] - 0.125f * s[i20] +
                        0.125f * s[i22];
                gb[i] = -0.125f * s[i00] - 0.25f * s[i01] - 0.125f * s[i02] + 0.125f * s[i20] + 0.25f * s[i21] +
                        0.125f * s[i22];
            }
            float n[3] = {ga[1] * gb[2] - ga[2] * gb[1], -ga[0] * gb[2] + ga[2] * gb[0], ga[0] * gb[1] - ga[1] * gb[0]};
            float norm = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            if (norm <= 0.00000f) continue;
            n[0] = -n[0] / norm;
            n[1] = -n[1] / norm;
            n[2] = -n[2] / norm;
            const float* vm = viewmatrix;
            normals[i11] = vm[0] * n[0] + vm[1] * n[1] + vm[2] * n[2];
            normals[HW + i11] = vm[4] * n[0] + vm[5] * n[1] + vm[6] * n[2];
            normals[2 * HW + i11] = vm[8] * n[0] + vm[9] * n[1] + vm[10] * n[2];
        }
}

/* reference backward.cu:401-614 (renderCUDA backward), restated per pixel; accumulators in double.
 * dL_dmean2D is [P,3] (z = depth side channel, backward.cu:603), dL_dconic is [P,4] (x,y,_,w). */
void r3dgo_render_backward(int W, int H, int S, const uint32_t* ranges, const uint32_t* point_list,
                           const float* bg_color, const float* means2D, const float* depths,
                           const float* conic_opacity, const float* colors, const float* features,
                           const float* final_Ts, const uint32_t* n_contrib, const float* dL_dpixels,
                           const float* dL_dpixels_o, const float* dL_dpixels_d, const float* dL_dpixels_f,
                           int backward_geometry, double* dL_dmean2D, double* dL_dconic2D, double* dL_dopacity,
                           double* dL_dcolors, double* dL_dfeature)
{
    const int gx = (W + BLOCK_X - 1) / BLOCK_X;
    const size_t HW = (size_t)H * W;
    const float ddelx_dx = (float)(0.5 * W); /* backward.cu:476-477 (double product rounded to float) */
    const float ddely_dy = (float)(0.5 * H);
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            const int tile = (py / BLOCK_Y) * gx + (px / BLOCK_X);
            const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
            const size_t pix_id = (size_t)W * py + px;
            const float pixf[2] = {(float)px, (float)py};
            const float T_final = final_Ts[pix_id];
            float T = T_final;
            uint32_t contributor = r1 - r0;
            const uint32_t last_contributor = n_contrib[pix_id];
            float accum_rec[3] = {0, 0, 0}, accum_rec_d = 0, accum_rec_o = 0, accum_rec_f[64];
            float dL_dpixel[3], dL_dpixel_f[64];
            for (int i = 0; i < 3; i++) dL_dpixel[i] = dL_dpixels[i * HW + pix_id];
            float dL_dpixel_d = dL_dpixels_d[pix_id], dL_dpixel_o = dL_dpixels_o[pix_id];
            for (int i = 0; i < S; i++) {
                dL_dpixel_f[i] = dL_dpixels_f[i * HW + pix_id];
                accum_rec_f[i] = 0;
            }
            float last_alpha = 0, last_depth = 0, last_color[3] = {0, 0, 0}, last_feature[64];
            for (int i = 0; i < S; i++) last_feature[i] = 0;
            for (uint32_t kk = r1; kk > r0; kk--) {
                const uint32_t g = point_list[kk - 1];
                contributor--;
                if (contributor >= last_contributor) continue;
                float dx = means2D[2 * g] - pixf[0], dy = means2D[2 * g + 1] - pixf[1];
                const float* con_o = conic_opacity + 4 * g;
                const float power = -0.5f * (con_o[0] * dx * dx + con_o[2] * dy * dy) - con_o[1] * dx * dy;
                if (power > 0.0f) continue;
                const float G = expf(power);
                const float alpha = fminf(0.99f, con_o[3] * G);
                if (alpha < 1.0f / 255.0f) continue;
                T = T / (1.f - alpha);
                const float dchannel_dcolor = alpha * T;
                float dL_dalpha = 0.0f;
                for (int ch = 0; ch < 3; ch++) {
                    const float c = colors[g * 3 + ch];
                    accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                    last_color[ch] = c;
                    const float dL_dchannel = dL_dpixel[ch];
                    dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
                    dL_dcolors[(size_t)g * 3 + ch] += (double)(dchannel_dcolor * dL_dchannel);
                }
                for (int ch = 0; ch < S; ch++) {
                    const float f = features[(size_t)g * S + ch];
                    accum_rec_f[ch] = last_alpha * last_feature[ch] + (1.f - last_alpha) * accum_rec_f[ch];
                    last_feature[ch] = f;
                    const float dL_dchannel_f = dL_dpixel_f[ch];
                    if (backward_geometry) dL_dalpha += (f - accum_rec_f[ch]) * dL_dchannel_f;
                    dL_dfeature[(size_t)g * S + ch] += (double)(dchannel_dcolor * dL_dchannel_f);
                }
                const float depth = depths[g];
                accum_rec_d = last_alpha * last_depth + (1.f - last_alpha) * accum_rec_d;
                last_depth = depth;
                dL_dalpha += (depth - accum_rec_d) * dL_dpixel_d;
                accum_rec_o = last_alpha + (1.f - last_alpha) * accum_rec_o;
                dL_dalpha += (1.0f - accum_rec_o) * dL_dpixel_o;
                dL_dalpha *= T;
                last_alpha = alpha;
                float bg_dot_dpixel = 0;
                for (int i = 0; i < 3; i++) bg_dot_dpixel += bg_color[i] * dL_dpixel[i];
                dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
                const float dL_dG = con_o[3] * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * con_o[0] - gdy * con_o[1];
                const float dG_ddely = -gdy * con_o[2] - gdx * con_o[1];
                dL_dmean2D[(size_t)g * 3 + 0] += (double)(dL_dG * dG_ddelx * ddelx_dx);
                dL_dmean2D[(size_t)g * 3 + 1] += (double)(dL_dG * dG_ddely * ddely_dy);
                dL_dmean2D[(size_t)g * 3 + 2] += (double)(dL_dpixel_d * dchannel_dcolor);
                dL_dconic2D[(size_t)g * 4 + 0] += (double)(-0.5f * gdx * dx * dL_dG);
                dL_dconic2D[(size_t)g * 4 + 1] += (double)(-0.5f * gdx * dy * dL_dG);
                dL_dconic2D[(size_t)g * 4 + 3] += (double)(-0.5f * gdy * dy * dL_dG);
                dL_dopacity[g] += (double)(G * dL_dalpha);
            }
        }
}

/* reference backward.cu:144-276 (computeCov2DCUDA). Inputs/outputs are float like the reference's. */
void r3dgo_cov2d_backward(int P, const float* means, const int32_t* radii, const float* cov3Ds, float h_x,
                          float h_y, float tan_fovx, float tan_fovy, const float* view_matrix,
                          const float* dL_dconics /*[P,4]*/, const float* dL_dmean2D /*[P,3]*/,
                          float* dL_dmeans /*[P,3]*/, float* dL_dcov /*[P,6]*/)
{
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        const float* cov3D = cov3Ds + 6 * idx;
        float dL_dconic[3] = {dL_dconics[4 * idx], dL_dconics[4 * idx + 1], dL_dconics[4 * idx + 3]};
        cov2d_t c2;
        cov2d_common(means + 3 * idx, h_x, h_y, tan_fovx, tan_fovy, cov3D, view_matrix, &c2);
        const float x_grad_mul = (c2.txtz < -c2.limx || c2.txtz > c2.limx) ? 0 : 1;
        const float y_grad_mul = (c2.tytz < -c2.limy || c2.tytz > c2.limy) ? 0 : 1;
        float a = c2.a, b = c2.b, c = c2.c;
        float denom = a * c - b * b;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
#define T_(i, j) c2.T.c[i][j]
#define V_(i, j) c2.Vrk.c[i][j]
#define W_(i, j) c2.W.c[i][j]
        if (denom2inv != 0) {
            dL_da = denom2inv * (-c * c * dL_dconic[0] + 2 * b * c * dL_dconic[1] + (denom - a * c) * dL_dconic[2]);
            dL_dc = denom2inv * (-a * a * dL_dconic[2] + 2 * a * b * dL_dconic[1] + (denom - a * c) * dL_dconic[0]);
            dL_db = denom2inv * 2 * (b * c * dL_dconic[0] - (denom + 2 * b * b) * dL_dconic[1] + a * b * dL_dconic[2]);
            dL_dcov[6 * idx + 0] = (T_(0, 0) * T_(0, 0) * dL_da + T_(0, 0) * T_(1, 0) * dL_db + T_(1, 0) * T_(1, 0) * dL_dc);
            dL_dcov[6 * idx + 3] = (T_(0, 1) * T_(0, 1) * dL_da + T_(0, 1) * T_(1, 1) * dL_db + T_(1, 1) * T_(1, 1) * dL_dc);
            dL_dcov[6 * idx + 5] = (T_(0, 2) * T_(0, 2) * dL_da + T_(0, 2) * T_(1, 2) * dL_db + T_(1, 2) * T_(1, 2) * dL_dc);
            dL_dcov[6 * idx + 1] = 2 * T_(0, 0) * T_(0, 1) * dL_da + (T_(0, 0) * T_(1, 1) + T_(0, 1) * T_(1, 0)) * dL_db + 2 * T_(1, 0) * T_(1, 1) * dL_dc;
            dL_dcov[6 * idx + 2] = 2 * T_(0, 0) * T_(0, 2) * dL_da + (T_(0, 0) * T_(1, 2) + T_(0, 2) * T_(1, 0)) * dL_db + 2 * T_(1, 0) * T_(1, 2) * dL_dc;
            dL_dcov[6 * idx + 4] = 2 * T_(0, 2) * T_(0, 1) * dL_da + (T_(0, 1) * T_(1, 2) + T_(0, 2) * T_(1, 1)) * dL_db + 2 * T_(1, 1) * T_(1, 2) * dL_dc;
        } else {
            for (int i = 0; i < 6; i++) dL_dcov[6 * idx + i] = 0;
        }
        float dL_dT00 = 2 * (T_(0, 0) * V_(0, 0) + T_(0, 1) * V_(0, 1) + T_(0, 2) * V_(0, 2)) * dL_da +
                        (T_(1, 0) * V_(0, 0) + T_(1, 1) * V_(0, 1) + T_(1, 2) * V_(0, 2)) * dL_db;
        float dL_dT01 = 2 * (T_(0, 0) * V_(1, 0) + T_(0, 1) * V_(1, 1) + T_(0, 2) * V_(1, 2)) * dL_da +
                        (T_(1, 0) * V_(1, 0) + T_(1, 1) * V_(1, 1) + T_(1, 2) * V_(1, 2)) * dL_db;
        float dL_dT02 = 2 * (T_(0, 0) * V_(2, 0) + T_(0, 1) * V_(2, 1) + T_(0, 2) * V_(2, 2)) * dL_da +
                        (T_(1, 0) * V_(2, 0) + T_(1, 1) * V_(2, 1) + T_(1, 2) * V_(2, 2)) * dL_db;
        float dL_dT10 = 2 * (T_(1, 0) * V_(0, 0) + T_(1, 1) * V_(0, 1) + T_(1, 2) * V_(0, 2)) * dL_dc +
                        (T_(0, 0) * V_(0, 0) + T_(0, 1) * V_(0, 1) + T_(0, 2) * V_(0, 2)) * dL_db;
        float dL_dT11 = 2 * (T_(1, 0) * V_(1, 0) + T_(1, 1) * V_(1, 1) + T_(1, 2) * V_(1, 2)) * dL_dc +
                        (T_(0, 0) * V_(1, 0) + T_(0, 1) * V_(1, 1) + T_(0, 2) * V_(1, 2)) * dL_db;
        float dL_dT12 = 2 * (T_(1, 0) * V_(2, 0) + T_(1, 1) * V_(2, 1) + T_(1, 2) * V_(2, 2)) * dL_dc +
                        (T_(0, 0) * V_(2, 0) + T_(0, 1) * V_(2, 1) + T_(0, 2) * V_(2, 2)) * dL_db;
        float dL_dJ00 = W_(0, 0) * dL_dT00 + W_(0, 1) * dL_dT01 + W_(0, 2) * dL_dT02;
        float dL_dJ02 = W_(2, 0) * dL_dT00 + W_(2, 1) * dL_dT01 + W_(2, 2) * dL_dT02;
        float dL_dJ11 = W_(1, 0) * dL_dT10 + W_(1, 1) * dL_dT11 + W_(1, 2) * dL_dT12;
        float dL_dJ12 = W_(2, 0) * dL_dT10 + W_(2, 1) * dL_dT11 + W_(2, 2) * dL_dT12;
#undef T_
#undef V_
#undef W_
        const float* t = c2.t;
        float tz = 1.f / t[2];
        float tz2 = tz * tz;
        float tz3 = tz2 * tz;
        float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
        float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
        float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t[0]) * tz3 * dL_dJ02 +
                       (2 * h_y * t[1]) * tz3 * dL_dJ12;
        /* transformVec4x3Transpose of (dtx, dty, dtz + dL_dmean2D.z)  (backward.cu:269) */
        float v[3] = {dL_dtx, dL_dty, dL_dtz + dL_dmean2D[3 * idx + 2]};
        const float* m = view_matrix;
        dL_dmeans[3 * idx + 0] = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
        dL_dmeans[3 * idx + 1] = m[4] * v[0] + m[5] * v[1] + m[6] * v[2];
        dL_dmeans[3 * idx + 2] = m[8] * v[0] + m[9] * v[1] + m[10] * v[2];
    }
}

/* reference backward.cu:20-139 (SH backward), :280-343 (cov3D backward), :348-398 (preprocessCUDA bwd) */
void r3dgo_preprocess_backward(int P, int D, int M, const float* means, const int32_t* radii, const float* shs,
                               const uint8_t* clamped, const float* scales, const float* rotations,
                               float scale_modifier, const float* proj, const float* campos,
                               const float* dL_dmean2D /*[P,3]*/, float* dL_dmeans /*[P,3] in/out*/,
                               const float* dL_dcolor /*[P,3]*/, const float* dL_dcov3D /*[P,6]*/,
                               float* dL_dsh /*[P,M,3]*/, float* dL_dscale /*[P,3]*/, float* dL_drot /*[P,4]*/)
{
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        const float* m = means + 3 * idx;
        float m_hom[4];
        transformPoint4x4(m, proj, m_hom);
        float m_w = 1.0f / (m_hom[3] + 0.0000001f);
        float mul1 = (proj[0] * m[0] + proj[4] * m[1] + proj[8] * m[2] + proj[12]) * m_w * m_w;
        float mul2 = (proj[1] * m[0] + proj[5] * m[1] + proj[9] * m[2] + proj[13]) * m_w * m_w;
        const float g2x = dL_dmean2D[3 * idx], g2y = dL_dmean2D[3 * idx + 1];
        float dm[3];
        dm[0] = (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
        dm[1] = (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
        dm[2] = (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
        for (int i = 0; i < 3; i++) dL_dmeans[3 * idx + i] += dm[i];

        if (shs) { /* backward.cu:20-139 */
            float dir_orig[3] = {m[0] - campos[0], m[1] - campos[1], m[2] - campos[2]};
            float len = sqrtf(dir_orig[0] * dir_orig[0] + dir_orig[1] * dir_orig[1] + dir_orig[2] * dir_orig[2]);
            float x = dir_orig[0] / len, y = dir_orig[1] / len, z = dir_orig[2] / len;
            const float* sh = shs + (size_t)idx * M * 3;
            float* dsh = dL_dsh + (size_t)idx * M * 3;
            float dL_ddir[3] = {0, 0, 0};
            for (int ch = 0; ch < 3; ch++) {
#define SH(k) sh[(k) * 3 + ch]
#define DSH(k) dsh[(k) * 3 + ch]
                float dL_dRGB = dL_dcolor[3 * idx + ch] * (clamped[3 * idx + ch] ? 0.f : 1.f);
                float dRGBdx = 0, dRGBdy = 0, dRGBdz = 0;
                DSH(0) = SH_C0 * dL_dRGB;
                if (D > 0) {
                    DSH(1) = (-SH_C1 * y) * dL_dRGB;
                    DSH(2) = (SH_C1 * z) * dL_dRGB;
                    DSH(3) = (-SH_C1 * x) * dL_dRGB;
                    dRGBdx = -SH_C1 * SH(3);
                    dRGBdy = -SH_C1 * SH(1);
                    dRGBdz = SH_C1 * SH(2);
                    if (D > 1) {
                        float xx = x * x, yy = y * y, zz = z * z;
                        float xy = x * y, yz = y * z, xz = x * z;
                        DSH(4) = (SH_C2[0] * xy) * dL_dRGB;
                        DSH(5) = (SH_C2[1] * yz) * dL_dRGB;
                        DSH(6) = (SH_C2[2] * (2.f * zz - xx - yy)) * dL_dRGB;
                        DSH(7) = (SH_C2[3] * xz) * dL_dRGB;
                        DSH(8) = (SH_C2[4] * (xx - yy)) * dL_dRGB;
                        dRGBdx += SH_C2[0] * y * SH(4) + SH_C2[2] * 2.f * -x * SH(6) + SH_C2[3] * z * SH(7) + SH_C2[4] * 2.f * x * SH(8);
                        dRGBdy += SH_C2[0] * x * SH(4) + SH_C2[1] * z * SH(5) + SH_C2[2] * 2.f * -y * SH(6) + SH_C2[4] * 2.f * -y * SH(8);
                        dRGBdz += SH_C2[1] * y * SH(5) + SH_C2[2] * 2.f * 2.f * z * SH(6) + SH_C2[3] * x * SH(7);
                        if (D > 2) {
                            DSH(9) = (SH_C3[0] * y * (3.f * xx - yy)) * dL_dRGB;
                            DSH(10) = (SH_C3[1] * xy * z) * dL_dRGB;
                            DSH(11) = (SH_C3[2] * y * (4.f * zz - xx - yy)) * dL_dRGB;
                            DSH(12) = (SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)) * dL_dRGB;
                            DSH(13) = (SH_C3[4] * x * (4.f * zz - xx - yy)) * dL_dRGB;
                            DSH(14) = (SH_C3[5] * z * (xx - yy)) * dL_dRGB;
                            DSH(15) = (SH_C3[6] * x * (xx - 3.f * yy)) * dL_dRGB;
                            dRGBdx += (SH_C3[0] * SH(9) * 3.f * 2.f * xy + SH_C3[1] * SH(10) * yz +
                                       SH_C3[2] * SH(11) * -2.f * xy + SH_C3[3] * SH(12) * -3.f * 2.f * xz +
                                       SH_C3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) + SH_C3[5] * SH(14) * 2.f * xz +
                                       SH_C3[6] * SH(15) * 3.f * (xx - yy));
                            dRGBdy += (SH_C3[0] * SH(9) * 3.f * (xx - yy) + SH_C3[1] * SH(10) * xz +
                                       SH_C3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) +
                                       SH_C3[3] * SH(12) * -3.f * 2.f * yz + SH_C3[4] * SH(13) * -2.f * xy +
                                       SH_C3[5] * SH(14) * -2.f * yz + SH_C3[6] * SH(15) * -3.f * 2.f * xy);
                            dRGBdz += (SH_C3[1] * SH(10) * xy + SH_C3[2] * SH(11) * 4.f * 2.f * yz +
                                       SH_C3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) +
                                       SH_C3[4] * SH(13) * 4.f * 2.f * xz + SH_C3[5] * SH(14) * (xx - yy));
                        }
                    }
                }
#undef SH
#undef DSH
                /* glm::dot(dRGBdx, dL_dRGB) sums channels x,y,z left to right */
                dL_ddir[0] += dRGBdx * dL_dRGB;
                dL_ddir[1] += dRGBdy * dL_dRGB;
                dL_ddir[2] += dRGBdz * dL_dRGB;
            }
            /* auxiliary.h:107-117 dnormvdv(float3) */
            const float* v = dir_orig;
            float sum2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
            float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            float o0 = ((+sum2 - v[0] * v[0]) * dL_ddir[0] - v[1] * v[0] * dL_ddir[1] - v[2] * v[0] * dL_ddir[2]) * invsum32;
            float o1 = (-v[0] * v[1] * dL_ddir[0] + (sum2 - v[1] * v[1]) * dL_ddir[1] - v[2] * v[1] * dL_ddir[2]) * invsum32;
            float o2 = (-v[0] * v[2] * dL_ddir[0] - v[1] * v[2] * dL_ddir[1] + (sum2 - v[2] * v[2]) * dL_ddir[2]) * invsum32;
            dL_dmeans[3 * idx + 0] += o0;
            dL_dmeans[3 * idx + 1] += o1;
            dL_dmeans[3 * idx + 2] += o2;
        }

        if (scales) { /* backward.cu:280-343 */
            const float* rot = rotations + 4 * idx;
            float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
            mat3 R = m3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                             2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                             2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
            mat3 S = m3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
            float s[3] = {scale_modifier * scales[3 * idx], scale_modifier * scales[3 * idx + 1],
                          scale_modifier * scales[3 * idx + 2]};
            S.c[0][0] = s[0];
            S.c[1][1] = s[1];
            S.c[2][2] = s[2];
            mat3 Mm = m3_mul(&S, &R);
            const float* d = dL_dcov3D + 6 * idx;
            mat3 dL_dSigma = m3_cols(d[0], 0.5f * d[1], 0.5f * d[2], 0.5f * d[1], d[3], 0.5f * d[4], 0.5f * d[2],
                                     0.5f * d[4], d[5]);
            /* dL_dM = 2.0f * M * dL_dSigma : glm evaluates (2.0f*M) then * dL_dSigma */
            mat3 M2 = Mm;
            for (int j = 0; j < 3; j++)
                for (int i = 0; i < 3; i++) M2.c[j][i] = 2.0f * Mm.c[j][i];
            mat3 dL_dM = m3_mul(&M2, &dL_dSigma);
            mat3 Rt = m3_T(&R);
            mat3 dL_dMt = m3_T(&dL_dM);
            for (int k = 0; k < 3; k++)
                dL_dscale[3 * idx + k] = Rt.c[k][0] * dL_dMt.c[k][0] + Rt.c[k][1] * dL_dMt.c[k][1] + Rt.c[k][2] * dL_dMt.c[k][2];
            for (int k = 0; k < 3; k++)
                for (int i = 0; i < 3; i++) dL_dMt.c[k][i] *= s[k];
#define D_(i, j) dL_dMt.c[i][j]
            float q0 = 2 * z * (D_(0, 1) - D_(1, 0)) + 2 * y * (D_(2, 0) - D_(0, 2)) + 2 * x * (D_(1, 2) - D_(2, 1));
            float q1 = 2 * y * (D_(1, 0) + D_(0, 1)) + 2 * z * (D_(2, 0) + D_(0, 2)) + 2 * r * (D_(1, 2) - D_(2, 1)) - 4 * x * (D_(2, 2) + D_(1, 1));
            float q2 = 2 * x * (D_(1, 0) + D_(0, 1)) + 2 * r * (D_(2, 0) - D_(0, 2)) + 2 * z * (D_(1, 2) + D_(2, 1)) - 4 * y * (D_(2, 2) + D_(0, 0));
            float q3 = 2 * r * (D_(0, 1) - D_(1, 0)) + 2 * x * (D_(2, 0) + D_(0, 2)) + 2 * y * (D_(1, 2) + D_(2, 1)) - 4 * z * (D_(1, 1) + D_(0, 0));
#undef D_
            dL_drot[4 * idx + 0] = q0;
            dL_drot[4 * idx + 1] = q1;
            dL_drot[4 * idx + 2] = q2;
            dL_drot[4 * idx + 3] = q3;
        }
    }
}

/* Thin entry points so the two per-Gaussian helpers can be pinned against the reference's Python twins
 * (eval_sh, utils/sh_utils.py:71-128; build_scaling_rotation+strip_symmetric, utils/general_utils.py). */
void r3dgo_sh_to_rgb(int P, int D, int M, const float* means, const float* campos, const float* shs,
                     uint8_t* clamped, float* rgb)
{
    for (int i = 0; i < P; i++) computeColorFromSH(i, D, M, means, campos, shs, clamped, rgb + 3 * i);
}
void r3dgo_cov3d(int P, const float* scales, float mod, const float* rots, float* cov3D)
{
    for (int i = 0; i < P; i++) computeCov3D(scales + 3 * i, mod, rots + 4 * i, cov3D + 6 * i);
}
