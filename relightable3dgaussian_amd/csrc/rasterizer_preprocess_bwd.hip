// Per-Gaussian backward (K12 + K13 fused into one launch) for gfx950.
// Reference semantics: computeCov2DCUDA backward.cu:144-276, preprocessCUDA backward.cu:348-398,
// SH backward :20-139, cov3D backward :280-343.  The reference launches two kernels because of code length;
// both are one-thread-per-Gaussian and touch the same rows, so one pass halves the HBM traffic on
// means/radii/dL_dmean2D (the pass is HBM-bound: ~104 + 531 B per Gaussian, dominated by the 192 B SH row read and the
// 192 B dL_dsh row written).  Those two rows travel through LDS (STAGED): the block moves its 256 rows with coalesced
// 16-byte accesses (common.hpp stage_rows_*), each thread walks its own row in LDS; a thread-per-Gaussian walk straight
// in HBM makes every wave load/store touch 64 different cache lines.
#include "common.hpp"

namespace r3dg {

struct B3 {
    float c[3][3];   // c[col][row], glm order
};
__device__ __forceinline__ B3 b3mul(const B3& A, const B3& B)
{
    B3 R;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++)
            R.c[j][i] = A.c[0][i] * B.c[j][0] + A.c[1][i] * B.c[j][1] + A.c[2][i] * B.c[j][2];
    return R;
}
__device__ __forceinline__ B3 b3t(const B3& A)
{
    B3 R;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++) R.c[j][i] = A.c[i][j];
    return R;
}

__device__ const float bSH_C0 = 0.28209479177387814f;
__device__ const float bSH_C1 = 0.4886025119029199f;
__device__ const float bSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                    -1.0925484305920792f, 0.5462742152960396f};
__device__ const float bSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                    0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                    -0.5900435899266435f};

template <bool STAGED>
__global__ void __launch_bounds__(256)
preprocess_backward_kernel(int P, int D, int M, const float* __restrict__ means, const int* __restrict__ radii,
                           const float* __restrict__ shs, const uint8_t* __restrict__ clamped,
                           const float* __restrict__ scales, const float* __restrict__ rotations, float scale_modifier,
                           const float* __restrict__ cov3Ds, const float* __restrict__ vm, const float* __restrict__ proj,
                           float h_x, float h_y, float tan_fovx, float tan_fovy, const float* __restrict__ campos,
                           float* __restrict__ dL_dmean2D, const float* __restrict__ dL_dconics,
                           const float4* __restrict__ conic_opacity, float half_w, float half_h,
                           float* __restrict__ dL_dmeans, const float* __restrict__ dL_dcolor,
                           float* __restrict__ dL_dcov3D, float* __restrict__ dL_dsh, float* __restrict__ dL_dscale,
                           float* __restrict__ dL_drot)
{
    extern __shared__ float s_rows[];                  // STAGED: 256 SH rows in, the same 256 dL_dsh rows out
    __shared__ uint8_t s_live[256];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const bool in_range = idx < P;
    const bool visible = in_range && radii[idx] > 0;
    float* row = nullptr;
    if (STAGED) {
        s_live[threadIdx.x] = visible;
        __syncthreads();
        stage_rows_in_256(shs, blockIdx.x * 256, P, 3 * M, s_live, s_rows);
        __syncthreads();
        row = s_rows + threadIdx.x * staged_row_stride(3 * M);
    }
    do {
    if (!in_range) break;
    if (!visible) {
        // invisible Gaussian: the reference leaves its torch::zeros rows untouched; here the rows are written so the
        // caller can hand in uninitialised memory (saves five zero-fill launches per backward)
#pragma unroll
        for (int i = 0; i < 3; i++) dL_dmeans[3 * idx + i] = 0.f;
#pragma unroll
        for (int i = 0; i < 6; i++) dL_dcov3D[6 * idx + i] = 0.f;
        if (shs != nullptr) {
            float* z = STAGED ? row : dL_dsh + (size_t)idx * M * 3;
            for (int i = 0; i < 3 * M; i++) z[i] = 0.f;
        }
        if (scales != nullptr) {
#pragma unroll
            for (int i = 0; i < 3; i++) dL_dscale[3 * idx + i] = 0.f;
#pragma unroll
            for (int i = 0; i < 4; i++) dL_drot[4 * idx + i] = 0.f;
        }
        break;
    }

    const float mx = means[3 * idx], my = means[3 * idx + 1], mz = means[3 * idx + 2];
    // moments_to_gradients (round 5).  The tile kernel (render_backward_wave_kernel) leaves the RAW moments of m = G dL_dG in the
    // mean / conic slots -- S_x, S_y in dL_dmean2D.xy, S_xx, S_xy, S_yy in dL_dconic.{x,y,w} -- because the conic they are to be
    // multiplied with is a constant of the Gaussian (backward.cu:579-601 multiplies per pixel):
    //     dL_dmean2D.x = -W/2 (A S_x + B S_y)   dL_dmean2D.y = -H/2 (C S_y + B S_x)   dL_dconic = -1/2 (S_xx, S_xy, S_yy)
    // with (A, B, C) the conic the forward stored (GeometryState::conic_opacity, the values the tile kernels blended with).  The
    // final dL_dmean2D goes back to its slot: it is an OUTPUT of the op (the viewspace gradient the densification reads).
    const float4 co = conic_opacity[idx];
    const float s_x = dL_dmean2D[3 * idx], s_y = dL_dmean2D[3 * idx + 1], g2z = dL_dmean2D[3 * idx + 2];
    const float g2x = -half_w * (co.x * s_x + co.y * s_y), g2y = -half_h * (co.z * s_y + co.y * s_x);
    dL_dmean2D[3 * idx] = g2x;
    dL_dmean2D[3 * idx + 1] = g2y;

    // ---------------- K12: conic -> cov2D -> cov3D / mean (backward.cu:144-276) ----------------
    const float* c3 = cov3Ds + 6 * idx;
    const float dcx = -0.5f * dL_dconics[4 * idx], dcy = -0.5f * dL_dconics[4 * idx + 1], dcw = -0.5f * dL_dconics[4 * idx + 3];
    float tx = vm[0] * mx + vm[4] * my + vm[8] * mz + vm[12];
    float ty = vm[1] * mx + vm[5] * my + vm[9] * mz + vm[13];
    const float tz = vm[2] * mx + vm[6] * my + vm[10] * mz + vm[14];
    const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
    const float txtz = tx / tz, tytz = ty / tz;
    tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
    ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
    const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;

    B3 J = {{{h_x / tz, 0.0f, -(h_x * tx) / (tz * tz)}, {0.0f, h_y / tz, -(h_y * ty) / (tz * tz)}, {0, 0, 0}}};
    B3 Wm = {{{vm[0], vm[4], vm[8]}, {vm[1], vm[5], vm[9]}, {vm[2], vm[6], vm[10]}}};
    B3 V = {{{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}}};
    B3 T = b3mul(Wm, J);
    B3 cov2D = b3mul(b3mul(b3t(T), b3t(V)), T);
    const float a = cov2D.c[0][0] + 0.3f, b = cov2D.c[0][1], c = cov2D.c[1][1] + 0.3f;
    const float denom = a * c - b * b;
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    float dcov[6];
#define T_(i, j) T.c[i][j]
#define V_(i, j) V.c[i][j]
#define W_(i, j) Wm.c[i][j]
    if (denom2inv != 0) {
        dL_da = denom2inv * (-c * c * dcx + 2 * b * c * dcy + (denom - a * c) * dcw);
        dL_dc = denom2inv * (-a * a * dcw + 2 * a * b * dcy + (denom - a * c) * dcx);
        dL_db = denom2inv * 2 * (b * c * dcx - (denom + 2 * b * b) * dcy + a * b * dcw);
        dcov[0] = (T_(0, 0) * T_(0, 0) * dL_da + T_(0, 0) * T_(1, 0) * dL_db + T_(1, 0) * T_(1, 0) * dL_dc);
        dcov[3] = (T_(0, 1) * T_(0, 1) * dL_da + T_(0, 1) * T_(1, 1) * dL_db + T_(1, 1) * T_(1, 1) * dL_dc);
        dcov[5] = (T_(0, 2) * T_(0, 2) * dL_da + T_(0, 2) * T_(1, 2) * dL_db + T_(1, 2) * T_(1, 2) * dL_dc);
        dcov[1] = 2 * T_(0, 0) * T_(0, 1) * dL_da + (T_(0, 0) * T_(1, 1) + T_(0, 1) * T_(1, 0)) * dL_db + 2 * T_(1, 0) * T_(1, 1) * dL_dc;
        dcov[2] = 2 * T_(0, 0) * T_(0, 2) * dL_da + (T_(0, 0) * T_(1, 2) + T_(0, 2) * T_(1, 0)) * dL_db + 2 * T_(1, 0) * T_(1, 2) * dL_dc;
        dcov[4] = 2 * T_(0, 2) * T_(0, 1) * dL_da + (T_(0, 1) * T_(1, 2) + T_(0, 2) * T_(1, 1)) * dL_db + 2 * T_(1, 1) * T_(1, 2) * dL_dc;
    } else {
#pragma unroll
        for (int i = 0; i < 6; i++) dcov[i] = 0;
    }
#pragma unroll
    for (int i = 0; i < 6; i++) dL_dcov3D[6 * idx + i] = dcov[i];

    const float r0 = T_(0, 0) * V_(0, 0) + T_(0, 1) * V_(0, 1) + T_(0, 2) * V_(0, 2);
    const float r1 = T_(0, 0) * V_(1, 0) + T_(0, 1) * V_(1, 1) + T_(0, 2) * V_(1, 2);
    const float r2 = T_(0, 0) * V_(2, 0) + T_(0, 1) * V_(2, 1) + T_(0, 2) * V_(2, 2);
    const float s0 = T_(1, 0) * V_(0, 0) + T_(1, 1) * V_(0, 1) + T_(1, 2) * V_(0, 2);
    const float s1 = T_(1, 0) * V_(1, 0) + T_(1, 1) * V_(1, 1) + T_(1, 2) * V_(1, 2);
    const float s2 = T_(1, 0) * V_(2, 0) + T_(1, 1) * V_(2, 1) + T_(1, 2) * V_(2, 2);
    const float dL_dT00 = 2 * r0 * dL_da + s0 * dL_db;
    const float dL_dT01 = 2 * r1 * dL_da + s1 * dL_db;
    const float dL_dT02 = 2 * r2 * dL_da + s2 * dL_db;
    const float dL_dT10 = 2 * s0 * dL_dc + r0 * dL_db;
    const float dL_dT11 = 2 * s1 * dL_dc + r1 * dL_db;
    const float dL_dT12 = 2 * s2 * dL_dc + r2 * dL_db;
    const float dL_dJ00 = W_(0, 0) * dL_dT00 + W_(0, 1) * dL_dT01 + W_(0, 2) * dL_dT02;
    const float dL_dJ02 = W_(2, 0) * dL_dT00 + W_(2, 1) * dL_dT01 + W_(2, 2) * dL_dT02;
    const float dL_dJ11 = W_(1, 0) * dL_dT10 + W_(1, 1) * dL_dT11 + W_(1, 2) * dL_dT12;
    const float dL_dJ12 = W_(2, 0) * dL_dT10 + W_(2, 1) * dL_dT11 + W_(2, 2) * dL_dT12;
#undef T_
#undef V_
#undef W_
    const float itz = 1.f / tz, itz2 = itz * itz, itz3 = itz2 * itz;
    const float dL_dtx = x_grad_mul * -h_x * itz2 * dL_dJ02;
    const float dL_dty = y_grad_mul * -h_y * itz2 * dL_dJ12;
    const float dL_dtz = -h_x * itz2 * dL_dJ00 - h_y * itz2 * dL_dJ11 + (2 * h_x * tx) * itz3 * dL_dJ02 +
                         (2 * h_y * ty) * itz3 * dL_dJ12;
    // depth gradient arrives through the z slot of dL_dmean2D (backward.cu:269,603)
    const float vz = dL_dtz + g2z;
    float dmx = vm[0] * dL_dtx + vm[1] * dL_dty + vm[2] * vz;
    float dmy = vm[4] * dL_dtx + vm[5] * dL_dty + vm[6] * vz;
    float dmz = vm[8] * dL_dtx + vm[9] * dL_dty + vm[10] * vz;

    // ---------------- K13: projection path (backward.cu:369-385) ----------------
    {
        const float m_hom_w = proj[3] * mx + proj[7] * my + proj[11] * mz + proj[15];
        const float m_w = 1.0f / (m_hom_w + 0.0000001f);
        const float mul1 = (proj[0] * mx + proj[4] * my + proj[8] * mz + proj[12]) * m_w * m_w;
        const float mul2 = (proj[1] * mx + proj[5] * my + proj[9] * mz + proj[13]) * m_w * m_w;
        dmx += (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
        dmy += (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
        dmz += (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
    }

    // ---------------- SH backward (backward.cu:20-139) ----------------
    if (shs != nullptr) {
        const float ox = mx - campos[0], oy = my - campos[1], oz = mz - campos[2];
        const float len = sqrtf(ox * ox + oy * oy + oz * oz);
        const float x = ox / len, y = oy / len, z = oz / len;
        const float* sh = shs + (size_t)idx * M * 3;
        float* dsh = STAGED ? row : dL_dsh + (size_t)idx * M * 3;
        // STAGED: dL_dsh overwrites the SH row in place, so the coefficients are lifted into registers first
        float shv[48];
        if (STAGED) {
#pragma unroll
            for (int i = 0; i < 48; i++) shv[i] = i < 3 * M ? row[i] : 0.f;
        }
        float ddx = 0, ddy = 0, ddz = 0;
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
#define SH(k) (STAGED ? shv[(k) * 3 + ch] : sh[(k) * 3 + ch])
#define DSH(k) dsh[(k) * 3 + ch]
            const float g = dL_dcolor[3 * idx + ch] * (clamped[3 * idx + ch] ? 0.f : 1.f);
            float dRx = 0, dRy = 0, dRz = 0;
            for (int k = (D + 1) * (D + 1); k < M; k++) DSH(k) = 0.f;      // coefficients above the active degree
            DSH(0) = bSH_C0 * g;
            if (D > 0) {
                DSH(1) = (-bSH_C1 * y) * g;
                DSH(2) = (bSH_C1 * z) * g;
                DSH(3) = (-bSH_C1 * x) * g;
                dRx = -bSH_C1 * SH(3);
                dRy = -bSH_C1 * SH(1);
                dRz = bSH_C1 * SH(2);
                if (D > 1) {
                    DSH(4) = (bSH_C2[0] * xy) * g;
                    DSH(5) = (bSH_C2[1] * yz) * g;
                    DSH(6) = (bSH_C2[2] * (2.f * zz - xx - yy)) * g;
                    DSH(7) = (bSH_C2[3] * xz) * g;
                    DSH(8) = (bSH_C2[4] * (xx - yy)) * g;
                    dRx += bSH_C2[0] * y * SH(4) + bSH_C2[2] * 2.f * -x * SH(6) + bSH_C2[3] * z * SH(7) + bSH_C2[4] * 2.f * x * SH(8);
                    dRy += bSH_C2[0] * x * SH(4) + bSH_C2[1] * z * SH(5) + bSH_C2[2] * 2.f * -y * SH(6) + bSH_C2[4] * 2.f * -y * SH(8);
                    dRz += bSH_C2[1] * y * SH(5) + bSH_C2[2] * 2.f * 2.f * z * SH(6) + bSH_C2[3] * x * SH(7);
                    if (D > 2) {
                        DSH(9) = (bSH_C3[0] * y * (3.f * xx - yy)) * g;
                        DSH(10) = (bSH_C3[1] * xy * z) * g;
                        DSH(11) = (bSH_C3[2] * y * (4.f * zz - xx - yy)) * g;
                        DSH(12) = (bSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)) * g;
                        DSH(13) = (bSH_C3[4] * x * (4.f * zz - xx - yy)) * g;
                        DSH(14) = (bSH_C3[5] * z * (xx - yy)) * g;
                        DSH(15) = (bSH_C3[6] * x * (xx - 3.f * yy)) * g;
                        dRx += (bSH_C3[0] * SH(9) * 3.f * 2.f * xy + bSH_C3[1] * SH(10) * yz + bSH_C3[2] * SH(11) * -2.f * xy +
                                bSH_C3[3] * SH(12) * -3.f * 2.f * xz + bSH_C3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) +
                                bSH_C3[5] * SH(14) * 2.f * xz + bSH_C3[6] * SH(15) * 3.f * (xx - yy));
                        dRy += (bSH_C3[0] * SH(9) * 3.f * (xx - yy) + bSH_C3[1] * SH(10) * xz +
                                bSH_C3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) + bSH_C3[3] * SH(12) * -3.f * 2.f * yz +
                                bSH_C3[4] * SH(13) * -2.f * xy + bSH_C3[5] * SH(14) * -2.f * yz +
                                bSH_C3[6] * SH(15) * -3.f * 2.f * xy);
                        dRz += (bSH_C3[1] * SH(10) * xy + bSH_C3[2] * SH(11) * 4.f * 2.f * yz +
                                bSH_C3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) + bSH_C3[4] * SH(13) * 4.f * 2.f * xz +
                                bSH_C3[5] * SH(14) * (xx - yy));
                    }
                }
            }
#undef SH
#undef DSH
            ddx += dRx * g;
            ddy += dRy * g;
            ddz += dRz * g;
        }
        // d normalize(v)/dv applied to (ddx,ddy,ddz)  (auxiliary.h:107-117)
        const float sum2 = ox * ox + oy * oy + oz * oz;
        const float inv32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
        dmx += ((sum2 - ox * ox) * ddx - oy * ox * ddy - oz * ox * ddz) * inv32;
        dmy += (-ox * oy * ddx + (sum2 - oy * oy) * ddy - oz * oy * ddz) * inv32;
        dmz += (-ox * oz * ddx - oy * oz * ddy + (sum2 - oz * oz) * ddz) * inv32;
    }
    dL_dmeans[3 * idx] = dmx;
    dL_dmeans[3 * idx + 1] = dmy;
    dL_dmeans[3 * idx + 2] = dmz;

    // ---------------- cov3D -> scale / rotation (backward.cu:280-343) ----------------
    if (scales != nullptr) {
        const float r = rotations[4 * idx], x = rotations[4 * idx + 1], y = rotations[4 * idx + 2], z = rotations[4 * idx + 3];
        B3 R = {{{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                 {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                 {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}}};
        const float s[3] = {scale_modifier * scales[3 * idx], scale_modifier * scales[3 * idx + 1],
                            scale_modifier * scales[3 * idx + 2]};
        B3 Sm = {{{s[0], 0, 0}, {0, s[1], 0}, {0, 0, s[2]}}};
        B3 Mm = b3mul(Sm, R);
        B3 dSig = {{{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]},
                    {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
                    {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}}};
        B3 M2;
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int i = 0; i < 3; i++) M2.c[j][i] = 2.0f * Mm.c[j][i];
        B3 dM = b3mul(M2, dSig);
        B3 Rt = b3t(R);
        B3 dMt = b3t(dM);
#pragma unroll
        for (int k = 0; k < 3; k++)
            dL_dscale[3 * idx + k] = Rt.c[k][0] * dMt.c[k][0] + Rt.c[k][1] * dMt.c[k][1] + Rt.c[k][2] * dMt.c[k][2];
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int i = 0; i < 3; i++) dMt.c[k][i] *= s[k];
#define D_(i, j) dMt.c[i][j]
        dL_drot[4 * idx + 0] = 2 * z * (D_(0, 1) - D_(1, 0)) + 2 * y * (D_(2, 0) - D_(0, 2)) + 2 * x * (D_(1, 2) - D_(2, 1));
        dL_drot[4 * idx + 1] = 2 * y * (D_(1, 0) + D_(0, 1)) + 2 * z * (D_(2, 0) + D_(0, 2)) + 2 * r * (D_(1, 2) - D_(2, 1)) - 4 * x * (D_(2, 2) + D_(1, 1));
        dL_drot[4 * idx + 2] = 2 * x * (D_(1, 0) + D_(0, 1)) + 2 * r * (D_(2, 0) - D_(0, 2)) + 2 * z * (D_(1, 2) + D_(2, 1)) - 4 * y * (D_(2, 2) + D_(0, 0));
        dL_drot[4 * idx + 3] = 2 * r * (D_(0, 1) - D_(1, 0)) + 2 * x * (D_(2, 0) + D_(0, 2)) + 2 * y * (D_(1, 2) + D_(2, 1)) - 4 * z * (D_(1, 1) + D_(0, 0));
#undef D_
    }
    } while (0);
    if (STAGED) {
        __syncthreads();
        stage_rows_out_256(dL_dsh, blockIdx.x * 256, P, 3 * M, s_rows);
    }
}

int g_stage_sh_rows = 1;        // R3DG_OPT_STAGE_SH_ROWS: 1 = SH / dL_dsh rows through LDS (default), 0 = direct per-thread walks
static inline int staged_row_stride_host(int row_floats) { return row_floats | 1; }

void launch_preprocess_backward(hipStream_t s, int P, int D, int M, const float* means, const int* radii,
                                const float* shs, const uint8_t* clamped, const float* scales, const float* rotations,
                                float scale_modifier, const float* cov3Ds, const float* vm, const float* proj, float h_x,
                                float h_y, float tan_fovx, float tan_fovy, const float* campos, float* dL_dmean2D,
                                const float* dL_dconic, const float* conic_opacity, int W, int H, float* dL_dmeans,
                                const float* dL_dcolor, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot)
{
    if (P <= 0) return;
    // SH rows through LDS when there are SH coefficients of at most degree 3 (256 x 49 words = 49 KB per block)
    const bool staged = opt(R3DG_OPT_STAGE_SH_ROWS) && shs != nullptr && M >= 1 && M <= 16;
    if (staged)
        preprocess_backward_kernel<true><<<(P + 255) / 256, 256, 256 * staged_row_stride_host(3 * M) * sizeof(float), s>>>(
            P, D, M, means, radii, shs, clamped, scales, rotations, scale_modifier, cov3Ds, vm, proj, h_x, h_y, tan_fovx,
            tan_fovy, campos, dL_dmean2D, dL_dconic, (const float4*)conic_opacity, 0.5f * W, 0.5f * H, dL_dmeans, dL_dcolor, dL_dcov3D,
            dL_dsh, dL_dscale, dL_drot);
    else
        preprocess_backward_kernel<false><<<(P + 255) / 256, 256, 0, s>>>(
            P, D, M, means, radii, shs, clamped, scales, rotations, scale_modifier, cov3Ds, vm, proj, h_x, h_y, tan_fovx,
            tan_fovy, campos, dL_dmean2D, dL_dconic, (const float4*)conic_opacity, 0.5f * W, 0.5f * H, dL_dmeans, dL_dcolor, dL_dcov3D,
            dL_dsh, dL_dscale, dL_drot);
}

}  // namespace r3dg
