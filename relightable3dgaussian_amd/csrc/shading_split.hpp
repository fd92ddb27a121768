// The relight frame under a light that turns with every frame: "split transport" cache + per-frame kernel with the lat-long
// lookup inside (see the banner below).  Included by shading.hip only (it uses the lookup helpers defined there).
#pragma once

namespace r3dg {

// =====================================================================================================================
// Relighting under a light that TURNS WITH EVERY FRAME (relighting.py:160-161 with configs/nerf_syn_light/light_transform.json,
// configs/tnt/light_transform.json): the lookup direction changes per frame, the rest of the transport does not.  "split
// transport" cache, light-independent:
//     per sample   lt_k = (max(SH_incident(d_k), 0) * a_k,  a_k)   with a_k = area_k * max(n . d_k, 0)         16 bytes
//                  vis_k                                                                                           4 bytes
//     per Gaussian mean local light (3), mean visibility
// and per frame   radiance_k = env(T d_k)   (lat-long lookup IN the kernel: acos / atan2 / 4 taps of the float4 texture in L2),
//     transport_k = lt_k.xyz + radiance_k * vis_k * a_k,   the GGX lobe, the sums.
// Layout and work split are different from the fixed-light kernels above, for the texture's sake: the map is 2 MB (256 x 512
// float4 texels) and a Gaussian's K directions cover its whole hemisphere, so with lane = sample a wave's 64 lookups touch 256
// unrelated cache lines.  Here lane = GAUSSIAN, the Gaussians are visited in an order sorted by their normal (the caller's
// permutation: a Morton code of the octahedral image of the normal), and all 64 lanes look up the SAME sample index k: their
// directions differ by the few degrees their normals differ, the taps fall into a handful of cache lines.  The caches are
// stored sample-major, [K][P] in that order, so the lanes' loads are contiguous; z_k and T are wave-uniform (scalar loads);
// every lane owns its Gaussian's sums -- no cross-lane reduction at all.
// =====================================================================================================================
constexpr int SPLIT_CONSTS = 4;        // per Gaussian: mean local light 3 | mean visibility

__device__ __forceinline__ void tr_rotation(const float n0, const float n1, const float n2, float (&R)[9])
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

// thread = Gaussian perm[i] (sorted order), loop over its K samples; dirs == nullptr: directions regenerated from the normal
// (normalize(R(n) z_k), as update_visibility generated them); visibility [P,K] in the caller's layout
__global__ void __launch_bounds__(256)
shade_build_split_kernel(int P, int K, const int* __restrict__ perm, const float* __restrict__ normals,
                         const float* __restrict__ incidents /*[P,16,3]*/, const float* __restrict__ visibility,
                         const float* __restrict__ dirs, const float* __restrict__ zsamples, float uniform_area,
                         float4* __restrict__ lt /*[K][P]*/, float* __restrict__ vis_t /*[K/4][P][4]*/, float* __restrict__ consts)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int g = perm[i];
    float c[48];
#pragma unroll
    for (int q = 0; q < 12; q++) {
        const float4 v = reinterpret_cast<const float4*>(incidents + (size_t)g * 48)[q];
        c[4 * q] = v.x; c[4 * q + 1] = v.y; c[4 * q + 2] = v.z; c[4 * q + 3] = v.w;
    }
    const float nx = normals[3 * (size_t)g], ny = normals[3 * (size_t)g + 1], nz = normals[3 * (size_t)g + 2];
    float R[9];
    tr_rotation(nx, ny, nz, R);
    float loc_sum[3] = {0.f, 0.f, 0.f}, vis_sum = 0.f;
    for (int k = 0; k < K; k++) {
        float dx, dy, dz;
        if (dirs != nullptr) {
            const float* d = dirs + 3 * ((size_t)g * K + k);
            dx = d[0]; dy = d[1]; dz = d[2];
        } else {
            const float zx = zsamples[3 * k], zy = zsamples[3 * k + 1], zz = zsamples[3 * k + 2];
            dx = R[0] * zx + R[1] * zy + R[2] * zz; dy = R[3] * zx + R[4] * zy + R[5] * zz; dz = R[6] * zx + R[7] * zy + R[8] * zz;
            const float len = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-12f);
            dx /= len; dy /= len; dz /= len;
        }
        float Y[16];
        sh_basis16(dx, dy, dz, 16, Y);
        float l[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < 48; f++) l[f % 3] += Y[f / 3] * c[f];
        const float a = uniform_area * fmaxf(nx * dx + ny * dy + nz * dz, 0.f);
        const float v = visibility[(size_t)g * K + k];
        const float l0 = fmaxf(l[0], 0.f), l1 = fmaxf(l[1], 0.f), l2 = fmaxf(l[2], 0.f);
        lt[(size_t)k * P + i] = make_float4(l0 * a, l1 * a, l2 * a, a);
        vis_t[((size_t)(k >> 2) * P + i) * 4 + (k & 3)] = v;      // [K / 4][P][4]: the four visibilities of a chunk are one load
        loc_sum[0] += l0; loc_sum[1] += l1; loc_sum[2] += l2;
        vis_sum += v;
    }
    const float invK = 1.0f / (float)K;
    reinterpret_cast<float4*>(consts)[i] = make_float4(loc_sum[0] * invK, loc_sum[1] * invK, loc_sum[2] * invK, vis_sum * invK);
}

// The map as bilinear FOOTPRINTS: record (y0 + 1) * (We + 1) + (x0 + 1), x0 in [-1, We - 1], y0 in [-1, He - 1], holds the four
// texels (x0, y0), (x0 + 1, y0), (x0, y0 + 1), (x0 + 1, y0 + 1) as 12 floats (48 bytes), zeros outside the map
// (grid_sample's zero padding).  A lookup is then THREE 16-byte loads of one record instead of four gathers of four texels: the
// per-frame kernel below is bound by the texture-address unit (a wave's 64 divergent 16-byte loads take it 64 cycles:
// 4 gathers x 384 samples x 18 waves per CU = 0.78 ms, which is what the kernel took), and needs no clamping arithmetic.
__global__ void __launch_bounds__(256)
shade_env_footprints_kernel(int He, int We, const float* __restrict__ env /*[He][We][3]*/, float4* __restrict__ fp /*[(He+1)(We+1)][3]*/)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= (He + 1) * (We + 1)) return;
    const int y0 = i / (We + 1) - 1, x0 = i % (We + 1) - 1;
    float v[12];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int x = x0 + b, y = y0 + a;
            const bool ok = x >= 0 && x < We && y >= 0 && y < He;
#pragma unroll
            for (int c = 0; c < 3; c++) v[(2 * a + b) * 3 + c] = ok ? env[3 * ((size_t)y * We + x) + c] : 0.f;
        }
#pragma unroll
    for (int q = 0; q < 3; q++) fp[3 * (size_t)i + q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

// env_fp: the HDR map as footprints (above); tr: row-major 3x3 light rotation applied to the lookup direction
// (envmap.py:39-42) or nullptr
// blockIdx.y = part: the kernel covers samples [part * Kp, min(K, part * Kp + Kp)) and leaves its nine sums in
// partial[part][i][12]; shade_split_combine_kernel adds the parts in a fixed order (deterministic) and writes the 19 columns.
// Why parts: with lane = Gaussian the launch is only P / 64 waves long -- 4.6 per SIMD at 300k Gaussians, one long dependent
// chain per sample each (acos / atan2 -> record address -> three loads -> blend): the SIMDs idled 30 % of the time
// (profiles/r04_pmc_valu_relight.json: VALU busy 0.70, 2.7 waves resident on average).  Three parts make it 13.7 shorter waves.
__global__ void __launch_bounds__(256)
shade_forward_split_kernel(int P, int K, int Kp, const int* __restrict__ perm, const float* __restrict__ base_color,
                           const float* __restrict__ roughness, const float* __restrict__ normals,
                           const float* __restrict__ viewdirs, const float4* __restrict__ lt, const float* __restrict__ vis_t,
                           const float* __restrict__ zsamples, const float* __restrict__ tr,
                           const float4* __restrict__ env_fp, int He, int We, float4* __restrict__ partial)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int g = perm[i];
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
    // Everything per sample happens in the LIGHT's frame: the lookup direction is T d_k = (T R) z_k, and the GGX lobe only needs
    // dot products, which T (a rotation) preserves -- N' = T N, V' = T V once per Gaussian, ONE 3x3 product per sample.
    float M[9], Np[3], Vp[3];
    {
        float R[9];
        tr_rotation(G.n[0], G.n[1], G.n[2], R);
        if (tr != nullptr) {
#pragma unroll
            for (int r = 0; r < 3; r++) {
#pragma unroll
                for (int c = 0; c < 3; c++) M[3 * r + c] = tr[3 * r] * R[c] + tr[3 * r + 1] * R[3 + c] + tr[3 * r + 2] * R[6 + c];
                Np[r] = tr[3 * r] * G.N[0] + tr[3 * r + 1] * G.N[1] + tr[3 * r + 2] * G.N[2];
                Vp[r] = tr[3 * r] * G.V[0] + tr[3 * r + 1] * G.V[1] + tr[3 * r + 2] * G.V[2];
            }
        } else {
#pragma unroll
            for (int q = 0; q < 9; q++) M[q] = R[q];
#pragma unroll
            for (int r = 0; r < 3; r++) { Np[r] = G.N[r]; Vp[r] = G.V[r]; }
        }
    }
    const float a2 = G.a2, kk = G.kk;
    const float nom1 = G.NoV * (1.f - kk) + kk;
    float S[3] = {0.f, 0.f, 0.f}, D[3] = {0.f, 0.f, 0.f}, Gl[3] = {0.f, 0.f, 0.f};
    // (requesting the next FOUR samples at once instead of one: no faster -- 125 registers; two: slower)
    const int kbeg = (int)blockIdx.y * Kp, kend = min(K, kbeg + Kp);
    float4 t_next = lt[(size_t)kbeg * P + i];
    float v_next = vis_t[((size_t)(kbeg >> 2) * P + i) * 4 + (kbeg & 3)];
    for (int k = kbeg; k < kend; k++) {
      {
        const float4 t = t_next;
        const float vis = v_next;
        if (k + 1 < kend) {                               // next sample's 20 bytes under this sample's arithmetic
            t_next = lt[(size_t)(k + 1) * P + i];
            v_next = vis_t[((size_t)((k + 1) >> 2) * P + i) * 4 + ((k + 1) & 3)];
        }
        const float zx = zsamples[3 * k], zy = zsamples[3 * k + 1], zz = zsamples[3 * k + 2];      // (wave-uniform: scalar loads)
        const float rx = M[0] * zx + M[1] * zy + M[2] * zz, ry = M[3] * zx + M[4] * zy + M[5] * zz,
                    rz = M[6] * zx + M[7] * zy + M[8] * zz;
        const float dinv = __builtin_amdgcn_rsqf(fmaxf(rx * rx + ry * ry + rz * rz, 1e-24f));
        const float Lx = rx * dinv, Ly = ry * dinv, Lz = rz * dinv;
        // radiance of the rotated direction
        const PackedTap tap = make_tap(Lx, Ly, Lz, nullptr, He, We);
        float e[3];
        {
            const float4* rec = env_fp + 3 * (size_t)(__mul24((int)(tap.xy >> 16), We + 1) + (int)(tap.xy & 0xffffu));
            const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
            const float wx0 = 1.f - tap.wx1, wy0 = 1.f - tap.wy1;
            const float w00 = wy0 * wx0, w01 = wy0 * tap.wx1, w10 = tap.wy1 * wx0, w11 = tap.wy1 * tap.wx1;
            e[0] = r0.x * w00 + r0.w * w01 + r1.z * w10 + r2.y * w11;
            e[1] = r0.y * w00 + r1.x * w01 + r1.w * w10 + r2.z * w11;
            e[2] = r0.z * w00 + r1.y * w01 + r2.x * w10 + r2.w * w11;
        }
        // GGX lobe (neilf.py:374-407) from N.L, N.V, L.V alone: |(L + V) / 2|^2 = (1 + L.V) / 2, N.H = (N.L + N.V) / (2 |u|), V.H = |u|
        const float rawNoL = Np[0] * Lx + Np[1] * Ly + Np[2] * Lz;
        const float lov = Vp[0] * Lx + Vp[1] * Ly + Vp[2] * Lz;
        const float uu = fmaxf(0.5f * lov + 0.5f, 1e-24f);
        const float uinv = __builtin_amdgcn_rsqf(uu);
        const float NoL = fminf(fmaxf(rawNoL, 1e-6f), 1.f);
        const float NoH = fminf(fmaxf((rawNoL + G.rawNoV) * (0.5f * uinv), 1e-6f), 1.f);
        const float VoH = fminf(fmaxf(uu * uinv, 1e-6f), 1.f);
        const float p2 = exp2f((-5.55473f * VoH - 6.98316f) * VoH);
        const float frac = (0.04f + 0.96f * p2) * a2;
        const float nom0 = NoH * NoH * (a2 - 1.f) + 1.f;
        const float nom2 = NoL * (1.f - kk) + kk;
        const float nom = fminf(fmaxf(4.f * kPi * nom0 * nom0 * nom1 * nom2, 1e-6f), 4.f * kPi);
        const float spec = frac / nom;
        const float wv = vis * t.w;
        const float tl[3] = {t.x, t.y, t.z};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float glob = e[c] * vis;
            const float tc = tl[c] + e[c] * wv;
            S[c] += spec * tc;
            D[c] += tc;
            Gl[c] += glob;
        }
      }
    }
    float4* o = partial + ((size_t)blockIdx.y * P + i) * 3;
    o[0] = make_float4(S[0], S[1], S[2], D[0]);
    o[1] = make_float4(D[1], D[2], Gl[0], Gl[1]);
    o[2] = make_float4(Gl[2], 0.f, 0.f, 0.f);
}

__global__ void __launch_bounds__(256)
shade_split_combine_kernel(int P, int K, int parts, const int* __restrict__ perm, const float* __restrict__ base_color,
                           const float4* __restrict__ partial, const float* __restrict__ consts, float* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int g = perm[i];
    float s9[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < parts; p++) {
        const float4* q = partial + ((size_t)p * P + i) * 3;
        const float4 a = q[0], b = q[1], c = q[2];
        s9[0] += a.x; s9[1] += a.y; s9[2] += a.z; s9[3] += a.w; s9[4] += b.x; s9[5] += b.y; s9[6] += b.z; s9[7] += b.w; s9[8] += c.x;
    }
    const float invK = 1.0f / (float)K;
    const float4 cst = reinterpret_cast<const float4*>(consts)[i];
    const float loc[3] = {cst.x, cst.y, cst.z};
    float* o = out + (size_t)g * SHADE_NOUT;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float d = s9[3 + c] * invK, sp = s9[c] * invK, gl = s9[6 + c] * invK;
        o[c] = base_color[3 * (size_t)g + c] / kPi * d + sp;           // pbr
        o[3 + c] = d;                               // diffuse_light
        o[6 + c] = sp;                              // specular
        o[9 + c] = loc[c] + gl;                     // mean incident light
        o[12 + c] = loc[c];                         // local
        o[15 + c] = gl;                             // global
    }
    o[18] = cst.w;                                  // mean visibility
}

}  // namespace r3dg
