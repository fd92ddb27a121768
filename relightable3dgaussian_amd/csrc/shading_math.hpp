// Per-sample / per-Gaussian arithmetic of the shading integral shared by the kernels of shading.hip: SH basis, the GGX
// set-up of one Gaussian, the local-light sum.  Plain C++ apart from the `__device__ __forceinline__` markers and float4, so
// that tests/emu (a lock-step CPU emulation of simple kernels, test infrastructure) can compile the same source.
#pragma once

namespace r3dg {

constexpr float kPi = 3.14159265358979323846f;
constexpr int SHADE_NOUT = 19;               // pbr3 diffuse3 specular3 lights3 local3 global3 vis1

// ---- real SH basis, degree 3, reference sign convention (sh_utils.py:92-127) ----
__device__ __forceinline__ void sh_basis16(float x, float y, float z, int M, float (&Y)[16])
{
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    Y[0] = C0;
#pragma unroll
    for (int i = 1; i < 16; i++) Y[i] = 0.f;
    if (M > 1) {
        Y[1] = -C1 * y; Y[2] = C1 * z; Y[3] = -C1 * x;
        if (M > 4) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            Y[4] = 1.0925484305920792f * xy;
            Y[5] = -1.0925484305920792f * yz;
            Y[6] = 0.31539156525252005f * (2.0f * zz - xx - yy);
            Y[7] = -1.0925484305920792f * xz;
            Y[8] = 0.5462742152960396f * (xx - yy);
            if (M > 9) {
                Y[9] = -0.5900435899266435f * y * (3.f * xx - yy);
                Y[10] = 2.890611442640554f * xy * z;
                Y[11] = -0.4570457994644658f * y * (4.f * zz - xx - yy);
                Y[12] = 0.3731763325901154f * z * (2.f * zz - 3.f * xx - 3.f * yy);
                Y[13] = -0.4570457994644658f * x * (4.f * zz - xx - yy);
                Y[14] = 1.445305721320277f * z * (xx - yy);
                Y[15] = -0.5900435899266435f * x * (xx - 3.f * yy);
            }
        }
    }
}

struct GaussFwd {            // wave-uniform per-Gaussian quantities
    float base[3], r, n[3], V[3], vlen, N[3], NoV, rawNoV, a, a2, kk;
    float nscale;            // N = n * nscale (sign(N0 . V) / |n|)
    float v_raw[3];
};

// record u[64]: 0..47 SH coefficients (i*3+c), 48..50 albedo, 51 roughness, 52..54 normal, 55..57 view direction
__device__ __forceinline__ void gauss_setup(GaussFwd& G, const float* u)
{
#pragma unroll
    for (int c = 0; c < 3; c++) {
        G.base[c] = u[48 + c];
        G.n[c] = u[52 + c];
        G.v_raw[c] = u[55 + c];
    }
    G.r = u[51];
    G.vlen = fmaxf(sqrtf(G.v_raw[0] * G.v_raw[0] + G.v_raw[1] * G.v_raw[1] + G.v_raw[2] * G.v_raw[2]), 1e-12f);
    const float nlen = fmaxf(sqrtf(G.n[0] * G.n[0] + G.n[1] * G.n[1] + G.n[2] * G.n[2]), 1e-12f);
    float N0[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        G.V[c] = G.v_raw[c] / G.vlen;
        N0[c] = G.n[c] / nlen;
    }
    const float d0 = G.V[0] * N0[0] + G.V[1] * N0[1] + G.V[2] * N0[2];
    const float sgn = d0 > 0.f ? 1.f : (d0 < 0.f ? -1.f : 0.f);
#pragma unroll
    for (int c = 0; c < 3; c++) G.N[c] = N0[c] * sgn;
    G.nscale = sgn / nlen;
    G.rawNoV = G.N[0] * G.V[0] + G.N[1] * G.V[1] + G.N[2] * G.V[2];
    G.NoV = fminf(fmaxf(G.rawNoV, 1e-6f), 1.f);
    G.a = G.r * G.r;
    G.a2 = G.a * G.a;
    G.kk = (G.a + 2.f * G.r + 1.0f) / 8.0f;
}

// local incident light before the clamp: sum_i Y_i(d) * sh[i][c]  (48 coefficients as 12 broadcast ds_read_b128)
__device__ __forceinline__ void sh_local_sum(const float* sh /*[48] in LDS, zero padded*/, const float (&Y)[16],
                                             float (&acc)[3])
{
    acc[0] = acc[1] = acc[2] = 0.f;
#pragma unroll
    for (int q = 0; q < 12; q++) {
        const float4 c4 = reinterpret_cast<const float4*>(sh)[q];
        const float cf[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int f = 4 * q + e;
            acc[f % 3] += Y[f / 3] * cf[e];
        }
    }
}

}  // namespace r3dg
