/*
 * oracle/shading_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement, AS WRITTEN, of the reference's render_equation.cu (the "contract model" of SURVEY.md Appendix C2;
 * dead code in the reference snapshot -- never compiled, bound or called -- but named by the north star):
 *   reo_forward          <- render_equation_forward_kernel          render_equation.cu:555-666
 *   reo_forward_complex  <- render_equation_forward_complex_kernel  render_equation.cu:55-190
 *   reo_backward         <- render_equation_backward_kernel         render_equation.cu:280-463
 * "parity unpinned by the reference": there are no tests, no golden vectors and no caller; the Python twin
 * (neilf_composite.py:201-294) uses a different ray set (10-degree z floor) and np.pi, so it cannot pin this bit-for-bit.
 *
 * Quirks of the backward kept exactly (a hypothetical CUDA run would produce them):
 *   Q1 dL_dn_d_i is OVERWRITTEN by the V-term gradient (:406), dropping the transport/cosine path;
 *   Q2 the incidents gradient loop is bounded by S_direct, not S_incident (:453) -- here additionally clamped to
 *      S_incident so it cannot write out of bounds;
 *   Q3 the masks for max(.,0) on the global/local light test the CLAMPED values (<0), i.e. never fire (:444-452);
 *   Q4 the half-vector normalisation is not differentiated ("TODO: consider norm", :431);
 *   Q5 dL_ddirect_shs[i] += ... is a non-atomic cross-thread race in the reference (:447); the only well-defined
 *      reading -- the sum over all Gaussians -- is what is computed here (in double).
 * fp32 per-sample math in the reference's order, pi = 3.14159f as in the source.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

static const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
static const float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                            0.5462742152960396f};
static const float C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                            -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

static void sh_coef3(const float* d, float* coef) /* render_equation.cu:20-53 with deg = 3 */
{
    const float x = d[0], y = d[1], z = d[2];
    coef[0] = C0;
    coef[1] = -C1 * y; coef[2] = C1 * z; coef[3] = -C1 * x;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    coef[4] = C2[0] * xy; coef[5] = C2[1] * yz; coef[6] = C2[2] * (2.0f * zz - xx - yy); coef[7] = C2[3] * xz;
    coef[8] = C2[4] * (xx - yy);
    coef[9] = C3[0] * y * (3.0f * xx - yy); coef[10] = C3[1] * xy * z; coef[11] = C3[2] * y * (4.0f * zz - xx - yy);
    coef[12] = C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy); coef[13] = C3[4] * x * (4.0f * zz - xx - yy);
    coef[14] = C3[5] * z * (xx - yy); coef[15] = C3[6] * x * (xx - 3.0f * yy);
}

/* Fibonacci direction rotated z -> normal (render_equation.cu:583-610) */
static void sample_dir(const float* normal, int ray_id, int sample_num, const float* rand_float, float* dir)
{
    const float delta = 3.14159f * (3.0f - sqrtf(5.0f));
    const float z = 1 - 2 * (float)ray_id / (2 * (float)sample_num - 1);
    const float rad = sqrtf(1 - z * z);
    float theta = delta * ray_id;
    if (rand_float) theta = rand_float[0] * 2 * 3.14159f + theta;
    const float y = cosf(theta) * rad;
    const float x = sinf(theta) * rad;
    const float v1 = -normal[1], v2 = normal[0], v3 = 0.f;
    const float v11 = v1 * v1, v22 = v2 * v2, v33 = v3 * v3, v12 = v1 * v2, v13 = v1 * v3, v23 = v2 * v3;
    const float cos_p_1 = fmaxf(normal[2] + 1, 0.0000001f);
    const float zs[3] = {
        (1 + (-v33 - v22) / cos_p_1) * x + (-v3 + v12 / cos_p_1) * y + (v2 + v13 / cos_p_1) * z,
        (v3 + v12 / cos_p_1) * x + (1 + (-v33 - v11) / cos_p_1) * y + (-v1 + v23 / cos_p_1) * z,
        (-v2 + v13 / cos_p_1) * x + (v1 + v23 / cos_p_1) * y + (1 + (-v22 - v11) / cos_p_1) * z};
    const float norm = sqrtf(fmaxf(0.0000001f, zs[0] * zs[0] + zs[1] * zs[1] + zs[2] * zs[2]));
    dir[0] = zs[0] / norm; dir[1] = zs[1] / norm; dir[2] = zs[2] / norm;
}

typedef struct {
    float coef[16], local[3], globl_raw[3], globl[3], vis, light[3];
    float half_n[3], half_norm, h_d_n, h_d_o, n_d_i, n_d_o;
    float f_d[3], r2, amp, sharp, expf_amp, D, F0[3], F[3], r2v, denom1, denom2, g1, g2, V, f_s[3];
} smp_t;

static float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

static void eval_sample(int idx, int Si, int Sd, int Sv, const float* base, float rough, float metal, const float* normal,
                        const float* viewdir, const float* inc, const float* direct, const float* vis, const float* dir,
                        smp_t* s)
{
    sh_coef3(dir, s->coef);
    for (int c = 0; c < 3; c++) {
        float l = 0.f, g = 0.5f;
        for (int i = 0; i < Si; i++) l += inc[((size_t)idx * Si + i) * 3 + c] * s->coef[i];
        for (int i = 0; i < Sd; i++) g += direct[i * 3 + c] * s->coef[i];
        s->local[c] = fmaxf(l, 0.0f);
        s->globl_raw[c] = fmaxf(g, 0.0f);
    }
    float v = 0.5f;
    for (int i = 0; i < Sv; i++) v += vis[(size_t)idx * Sv + i] * s->coef[i];
    s->vis = fmaxf(0.0f, fminf(v, 1.0f));
    float hd[3];
    for (int c = 0; c < 3; c++) {
        s->globl[c] = s->vis * s->globl_raw[c];
        s->light[c] = s->globl[c] + s->local[c];
        hd[c] = dir[c] + viewdir[c];
    }
    s->half_norm = fmaxf(sqrtf(dot3(hd, hd)), 0.0000001f);
    for (int c = 0; c < 3; c++) s->half_n[c] = hd[c] / s->half_norm;
    s->h_d_n = fmaxf(dot3(s->half_n, normal), 0.0f);
    s->h_d_o = fmaxf(dot3(s->half_n, viewdir), 0.0f);
    s->n_d_i = fmaxf(dot3(normal, dir), 0.0f);
    s->n_d_o = fmaxf(dot3(normal, viewdir), 0.0f);
    for (int c = 0; c < 3; c++) s->f_d[c] = (1 - metal) * base[c] / 3.14159f;
    s->r2 = fmaxf(rough * rough, 0.0000001f);
    s->amp = 1.0f / (s->r2 * 3.14159f);
    s->sharp = 2.0f / s->r2;
    s->expf_amp = expf(s->sharp * (s->h_d_n - 1.0f));
    s->D = s->amp * s->expf_amp;
    const float p5 = powf(1.0f - s->h_d_o, 5.0f);
    for (int c = 0; c < 3; c++) {
        s->F0[c] = 0.04f * (1.0f - metal) + base[c] * metal;
        s->F[c] = s->F0[c] + (1.0f - s->F0[c]) * p5;
    }
    s->r2v = powf(1.0f + rough, 2.0f) / 8.0f;
    s->denom1 = fmaxf(s->n_d_i * (1 - s->r2v) + s->r2v, 0.0000001f);
    s->denom2 = fmaxf(s->n_d_o * (1 - s->r2v) + s->r2v, 0.0000001f);
    s->g1 = 0.5f / s->denom1;
    s->g2 = 0.5f / s->denom2;
    s->V = s->g1 * s->g2;
    for (int c = 0; c < 3; c++) s->f_s[c] = s->D * s->F[c] * s->V;
}

void reo_forward(int P, int Si, int Sd, int Sv, const float* base_color, const float* roughness, const float* metallic,
                 const float* normals, const float* viewdirs, const float* inc, const float* direct, const float* vis,
                 int sample_num, const float* rand_float /* NULL unless is_training */, float* incident_dirs,
                 float* out_pbr, float* out_diffuse)
{
    for (int idx = 0; idx < P; idx++) {
        float pbr[3] = {0, 0, 0}, dl[3] = {0, 0, 0};
        for (int k = 0; k < sample_num; k++) {
            const size_t w = (size_t)idx * sample_num + k;
            float dir[3];
            sample_dir(normals + 3 * idx, k, sample_num, rand_float ? rand_float + w : 0, dir);
            smp_t s;
            eval_sample(idx, Si, Sd, Sv, base_color + 3 * idx, roughness[idx], metallic[idx], normals + 3 * idx,
                        viewdirs + 3 * idx, inc, direct, vis, dir, &s);
            const float tw = 2.0f * 3.14159f * s.n_d_i / (float)sample_num;
            for (int c = 0; c < 3; c++) {
                const float tr = s.light[c] * tw;
                pbr[c] += (s.f_d[c] + s.f_s[c]) * tr;
                dl[c] += tr;
                incident_dirs[3 * w + c] = dir[c];
            }
        }
        for (int c = 0; c < 3; c++) { out_pbr[3 * idx + c] = pbr[c]; out_diffuse[3 * idx + c] = dl[c]; }
    }
}

void reo_forward_complex(int P, int Si, int Sd, int Sv, const float* base_color, const float* roughness,
                         const float* metallic, const float* normals, const float* viewdirs, const float* inc,
                         const float* direct, const float* vis, int sample_num, float* incident_dirs, float* out_pbr,
                         float* out_lights, float* out_local, float* out_global, float* out_vis, float* out_diffuse,
                         float* out_local_diffuse, float* out_accum, float* out_rgb_d, float* out_rgb_s)
{
    for (int idx = 0; idx < P; idx++) {
        float rd[3] = {0, 0, 0}, rs[3] = {0, 0, 0}, dl[3] = {0, 0, 0}, ldl[3] = {0, 0, 0};
        for (int k = 0; k < sample_num; k++) {
            const size_t w = (size_t)idx * sample_num + k;
            float dir[3];
            sample_dir(normals + 3 * idx, k, sample_num, 0, dir);
            smp_t s;
            eval_sample(idx, Si, Sd, Sv, base_color + 3 * idx, roughness[idx], metallic[idx], normals + 3 * idx,
                        viewdirs + 3 * idx, inc, direct, vis, dir, &s);
            const float tmp = 2.0f * 3.14159f * s.n_d_i / (float)sample_num;
            for (int c = 0; c < 3; c++) {
                const float tr = s.light[c] * tmp, ltr = s.local[c] * tmp;
                dl[c] += tr; ldl[c] += ltr; rd[c] += s.f_d[c] * tr; rs[c] += s.f_s[c] * tr;
                incident_dirs[3 * w + c] = dir[c];
                out_lights[3 * w + c] = s.light[c];
                out_local[3 * w + c] = s.local[c];
                out_global[3 * w + c] = s.globl[c];
            }
            out_vis[w] = s.vis;
        }
        float av[3];
        for (int c = 0; c < 3; c++) {
            av[c] = dl[c] / 3.14159f + rs[c];
            out_pbr[3 * idx + c] = rd[c] + rs[c];
            out_rgb_d[3 * idx + c] = rd[c]; out_rgb_s[3 * idx + c] = rs[c];
            out_diffuse[3 * idx + c] = dl[c]; out_local_diffuse[3 * idx + c] = ldl[c];
        }
        out_accum[idx] = (av[0] + av[1] + av[2]) / 3;
    }
}

void reo_backward(int P, int Si, int Sd, int Sv, const float* base_color, const float* roughness, const float* metallic,
                  const float* normals, const float* viewdirs, const float* inc, const float* direct, const float* vis,
                  int sample_num, const float* incident_dirs, const float* dL_dpbrs, const float* dL_ddiffuse_lights,
                  float* dL_dbase_color, float* dL_droughness, float* dL_dmetallic, float* dL_dnormals,
                  float* dL_dviewdirs, float* dL_dincidents, double* dL_ddirect /* [Sd,3] */, float* dL_dvis)
{
    const int Sinc_loop = Sd < Si ? Sd : Si; /* Q2 */
    for (int idx = 0; idx < P; idx++) {
        const float* normal = normals + 3 * idx;
        const float* viewdir = viewdirs + 3 * idx;
        const float* base = base_color + 3 * idx;
        const float* dL_dpbr = dL_dpbrs + 3 * idx;
        const float* dL_ddl = dL_ddiffuse_lights + 3 * idx;
        const float metal = metallic[idx], rough = roughness[idx];
        for (int k = 0; k < sample_num; k++) {
            const size_t w = (size_t)idx * sample_num + k;
            const float* dir = incident_dirs + 3 * w;
            smp_t s;
            eval_sample(idx, Si, Sd, Sv, base, rough, metal, normal, viewdir, inc, direct, vis, dir, &s);
            const float tw = 2.0f * 3.14159f * s.n_d_i / (float)sample_num;
            float dL_dfd[3], dL_dfs[3], dL_dlight[3];
            for (int c = 0; c < 3; c++) {
                dL_dfd[c] = dL_dpbr[c] * s.light[c] * tw;
                dL_dfs[c] = dL_dpbr[c] * s.light[c] * tw;
                dL_dlight[c] = dL_dpbr[c] * (s.f_d[c] + s.f_s[c]) * tw;
                dL_dlight[c] += dL_ddl[c] * tw;
            }
            /* from dL_dfd */
            float dL_dbase[3];
            for (int c = 0; c < 3; c++) dL_dbase[c] = dL_dfd[c] * (1 - metal) / 3.14159f;
            float dL_dmetal = -dot3(dL_dfd, base) / 3.14159f;
            /* from dL_dfs */
            float t3[3];
            for (int c = 0; c < 3; c++) t3[c] = dL_dfs[c] * s.V;
            const float dL_dD = dot3(t3, s.F);
            float dL_dF[3];
            for (int c = 0; c < 3; c++) { dL_dF[c] = dL_dfs[c] * s.D * s.V; t3[c] = dL_dfs[c] * s.D; }
            const float dL_dV = dot3(t3, s.F);
            /* from dL_dD */
            const float dL_damp = dL_dD * s.expf_amp;
            const float dL_dexpf_amp = dL_dD * s.amp;
            const float dL_dsharp = (s.h_d_n - 1.0f) * s.expf_amp * dL_dexpf_amp;
            const float dL_dh_d_n = s.sharp * s.expf_amp * dL_dexpf_amp;
            const float dL_dr2 = -2.0f / (s.r2 * s.r2) * dL_dsharp - 1.0f / (s.r2 * s.r2 * 3.14159f) * dL_damp;
            float dL_drough = dL_dr2 * 2.0f * rough;
            /* from dL_dF */
            const float p5 = powf(1.0f - s.h_d_o, 5.0f), p4 = powf(1.0f - s.h_d_o, 4.0f);
            float dL_dF0[3], omF0[3], bm[3];
            for (int c = 0; c < 3; c++) { dL_dF0[c] = (1.0f - p5) * dL_dF[c]; omF0[c] = 1.0f - s.F0[c]; bm[c] = base[c] - 0.04f; }
            const float dL_dh_d_o = dot3(omF0, dL_dF) * -5.0f * p4;
            for (int c = 0; c < 3; c++) dL_dbase[c] += metal * dL_dF0[c];
            dL_dmetal += dot3(bm, dL_dF0);
            /* from dL_dV */
            const float dL_dg1 = dL_dV * s.g2, dL_dg2 = dL_dV * s.g1;
            const float dL_dden1 = -0.5f / (s.denom1 * s.denom1) * dL_dg1;
            const float dL_dden2 = -0.5f / (s.denom2 * s.denom2) * dL_dg2;
            const float dL_dn_d_i = dL_dden1 * (1 - s.r2v); /* Q1: overwrite */
            const float dL_dn_d_o = dL_dden2 * (1 - s.r2v);
            const float dL_dr2v = (1.0f - s.n_d_i) * dL_dden1 + (1.0f - s.n_d_o) * dL_dden2;
            dL_drough += (1.0f + rough) / 4.0f * dL_dr2v;
            float dh[3] = {0, 0, 0}, dn[3] = {0, 0, 0}, dv[3] = {0, 0, 0};
            if (s.h_d_n > 0.0f) for (int c = 0; c < 3; c++) { dh[c] += normal[c] * dL_dh_d_n; dn[c] += s.half_n[c] * dL_dh_d_n; }
            if (s.h_d_o > 0.0f) for (int c = 0; c < 3; c++) { dh[c] += viewdir[c] * dL_dh_d_o; dv[c] += s.half_n[c] * dL_dh_d_o; }
            if (s.n_d_i > 0.0f) for (int c = 0; c < 3; c++) dn[c] += dir[c] * dL_dn_d_i;
            if (s.n_d_o > 0.0f) for (int c = 0; c < 3; c++) { dn[c] += viewdir[c] * dL_dn_d_o; dv[c] += normal[c] * dL_dn_d_o; }
            for (int c = 0; c < 3; c++) dv[c] += dh[c] / s.half_norm; /* Q4 */
            /* shs */
            float dL_dglob[3];
            for (int c = 0; c < 3; c++) dL_dglob[c] = dL_dlight[c] * s.vis;
            const float dL_dvisib = dot3(dL_dlight, s.globl_raw);
            if (s.vis <= 1.0f && s.vis >= 0.0f)
                for (int i = 0; i < Sv; i++) dL_dvis[(size_t)idx * Sv + i] += dL_dvisib * s.coef[i];
            /* Q3: masks on the clamped values never fire */
            for (int i = 0; i < Sd; i++)
                for (int c = 0; c < 3; c++) dL_ddirect[i * 3 + c] += (double)(dL_dglob[c] * s.coef[i]);
            for (int i = 0; i < Sinc_loop; i++)
                for (int c = 0; c < 3; c++) dL_dincidents[((size_t)idx * Si + i) * 3 + c] += dL_dlight[c] * s.coef[i];
            for (int c = 0; c < 3; c++) {
                dL_dviewdirs[3 * idx + c] += dv[c];
                dL_dnormals[3 * idx + c] += dn[c];
                dL_dbase_color[3 * idx + c] += dL_dbase[c];
            }
            dL_dmetallic[idx] += dL_dmetal;
            dL_droughness[idx] += dL_drough;
        }
    }
}
