// SSIM term of the training loss (utils/loss_utils.py:20-63: 11x11 Gaussian window, sigma 1.5, zero padding, C1 = 0.01^2,
// C2 = 0.03^2, mean over all channels and pixels), forward and backward w.r.t. the rendered image, for gfx950.
// The reference runs five grouped conv2d calls plus a dozen elementwise kernels and their autograd twins; here:
//   ssim_forward_kernel   one 16x16 tile (+5 pixel halo) per workgroup: both images staged in LDS, separable window
//                         (11 horizontal taps into LDS, 11 vertical taps from it) for mu1, mu2, E[x^2], E[y^2], E[xy];
//                         writes the three per-pixel partial derivatives dS/dmu1, dS/dE[x^2], dS/dE[xy] and adds the tile's
//                         SSIM sum to *sum (one atomic per workgroup);
//   ssim_backward_kernel  the window is symmetric, so dL/dx = scale * (W*dS/dmu1 + 2x W*dS/dE[x^2] + y W*dS/dE[xy])
//                         with the same tiling over the three partial maps.
// HBM-bound streaming kernels: 5 map reads + 3 writes forward, 5 reads + 1 write backward per channel.
#include "common.hpp"

namespace r3dg {

constexpr int SSIM_R = 5;                    // window radius (window_size 11)
constexpr int SSIM_T = 16;                   // tile edge
constexpr int SSIM_E = SSIM_T + 2 * SSIM_R;  // 26: tile + halo

// gaussian(11, 1.5) / sum, the fp32 values the reference's create_window produces (loss_utils.py:20-29)
__constant__ float kSsimWin[11] = {1.028380124e-03f, 7.598758209e-03f, 3.600077331e-02f, 1.093606874e-01f,
                                   2.130055279e-01f, 2.660117149e-01f, 2.130055279e-01f, 1.093606874e-01f,
                                   3.600077331e-02f, 7.598758209e-03f, 1.028380124e-03f};

__device__ __forceinline__ float block_sum_256s(float v, float* s_part)
{
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    return s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

__global__ void __launch_bounds__(256)
ssim_forward_kernel(int W, int H, const float* __restrict__ x, const float* __restrict__ y,
                    float* __restrict__ partials, float* __restrict__ sum)
{
    __shared__ float s_x[SSIM_E][SSIM_E + 1], s_y[SSIM_E][SSIM_E + 1];
    __shared__ float s_h[5][SSIM_E][SSIM_T + 1];
    __shared__ float s_part[4];
    const size_t HW = (size_t)H * W;
    const int c = blockIdx.z;
    const float* xc = x + c * HW;
    const float* yc = y + c * HW;
    const int bx = blockIdx.x * SSIM_T, by = blockIdx.y * SSIM_T;
    for (int i = threadIdx.x; i < SSIM_E * SSIM_E; i += 256) {
        const int r = i / SSIM_E, q = i % SSIM_E;
        const int gy = by + r - SSIM_R, gx = bx + q - SSIM_R;
        const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;                 // zero padding
        s_x[r][q] = in ? xc[(size_t)gy * W + gx] : 0.f;
        s_y[r][q] = in ? yc[(size_t)gy * W + gx] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SSIM_E * SSIM_T; i += 256) {               // horizontal taps
        const int r = i / SSIM_T, q = i % SSIM_T;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float w = kSsimWin[k], xv = s_x[r][q + k], yv = s_y[r][q + k];
            a0 += w * xv; a1 += w * yv; a2 += w * xv * xv; a3 += w * yv * yv; a4 += w * xv * yv;
        }
        s_h[0][r][q] = a0; s_h[1][r][q] = a1; s_h[2][r][q] = a2; s_h[3][r][q] = a3; s_h[4][r][q] = a4;
    }
    __syncthreads();
    const int tx = threadIdx.x % SSIM_T, ty = threadIdx.x / SSIM_T;
    const int px = bx + tx, py = by + ty;
    float ssim = 0.f;
    if (px < W && py < H) {
        float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {                                           // vertical taps
            const float w = kSsimWin[k];
            mu1 += w * s_h[0][ty + k][tx]; mu2 += w * s_h[1][ty + k][tx];
            e11 += w * s_h[2][ty + k][tx]; e22 += w * s_h[3][ty + k][tx]; e12 += w * s_h[4][ty + k][tx];
        }
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        const float A = 2.f * mu1 * mu2 + C1;
        const float B = 2.f * (e12 - mu1 * mu2) + C2;
        const float C = mu1 * mu1 + mu2 * mu2 + C1;
        const float D = (e11 - mu1 * mu1) + (e22 - mu2 * mu2) + C2;
        const float inv = 1.f / (C * D);
        ssim = A * B * inv;
        // partial derivatives of ssim w.r.t. mu1, E[x^2], E[xy] (treated as independent window averages)
        const float d_mu1 = 2.f * mu2 * (B - A) * inv - ssim * 2.f * mu1 * (D - C) * inv;
        const float d_e11 = -ssim / D;
        const float d_e12 = 2.f * A * inv;
        const size_t o = (size_t)py * W + px;
        float* pc = partials + (size_t)c * 3 * HW;
        pc[o] = d_mu1;
        pc[HW + o] = d_e11;
        pc[2 * HW + o] = d_e12;
    }
    const float tot = block_sum_256s(ssim, s_part);
    if (threadIdx.x == 0 && sum != nullptr) atomicAdd(sum, tot);
}

__global__ void __launch_bounds__(256)
ssim_backward_kernel(int W, int H, const float* __restrict__ x, const float* __restrict__ y,
                     const float* __restrict__ partials, float scale, float* __restrict__ grad_x)
{
    __shared__ float s_p[3][SSIM_E][SSIM_E + 1];
    __shared__ float s_h[3][SSIM_E][SSIM_T + 1];
    const size_t HW = (size_t)H * W;
    const int c = blockIdx.z;
    const float* pc = partials + (size_t)c * 3 * HW;
    const int bx = blockIdx.x * SSIM_T, by = blockIdx.y * SSIM_T;
    for (int i = threadIdx.x; i < SSIM_E * SSIM_E; i += 256) {
        const int r = i / SSIM_E, q = i % SSIM_E;
        const int gy = by + r - SSIM_R, gx = bx + q - SSIM_R;
        const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;      // windows centred outside the image do not exist
        const size_t o = (size_t)gy * W + gx;
        s_p[0][r][q] = in ? pc[o] : 0.f;
        s_p[1][r][q] = in ? pc[HW + o] : 0.f;
        s_p[2][r][q] = in ? pc[2 * HW + o] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SSIM_E * SSIM_T; i += 256) {
        const int r = i / SSIM_T, q = i % SSIM_T;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float w = kSsimWin[k];
            a0 += w * s_p[0][r][q + k]; a1 += w * s_p[1][r][q + k]; a2 += w * s_p[2][r][q + k];
        }
        s_h[0][r][q] = a0; s_h[1][r][q] = a1; s_h[2][r][q] = a2;
    }
    __syncthreads();
    const int tx = threadIdx.x % SSIM_T, ty = threadIdx.x / SSIM_T;
    const int px = bx + tx, py = by + ty;
    if (px < W && py < H) {
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float w = kSsimWin[k];
            c0 += w * s_h[0][ty + k][tx]; c1 += w * s_h[1][ty + k][tx]; c2 += w * s_h[2][ty + k][tx];
        }
        const size_t o = (size_t)py * W + px;
        const float xv = x[c * HW + o], yv = y[c * HW + o];
        grad_x[c * HW + o] = scale * (c0 + 2.f * xv * c1 + yv * c2);
    }
}

void launch_ssim_forward(hipStream_t s, int W, int H, int C, const float* x, const float* y, float* partials, float* sum)
{
    dim3 grid((W + SSIM_T - 1) / SSIM_T, (H + SSIM_T - 1) / SSIM_T, C);
    ssim_forward_kernel<<<grid, 256, 0, s>>>(W, H, x, y, partials, sum);
    check_launch(s, false, "ssim_forward_kernel");
}

void launch_ssim_backward(hipStream_t s, int W, int H, int C, const float* x, const float* y, const float* partials,
                          float scale, float* grad_x)
{
    dim3 grid((W + SSIM_T - 1) / SSIM_T, (H + SSIM_T - 1) / SSIM_T, C);
    ssim_backward_kernel<<<grid, 256, 0, s>>>(W, H, x, y, partials, scale, grad_x);
    check_launch(s, false, "ssim_backward_kernel");
}

}  // namespace r3dg
