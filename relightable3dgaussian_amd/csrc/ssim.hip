// SSIM term of the training loss (utils/loss_utils.py:20-63: 11x11 Gaussian window, sigma 1.5, zero padding, C1 = 0.01^2,
// C2 = 0.03^2, mean over all channels and pixels), forward and backward w.r.t. the rendered image, for gfx950.
// The reference runs five grouped conv2d calls plus a dozen elementwise kernels and their autograd twins; here:
//   ssim_forward_kernel   one 32x32 tile (+5 pixel halo) per workgroup, 4 outputs per thread per pass: both images staged in LDS, separable window
//                         (11 horizontal taps into LDS, 11 vertical taps from it) for mu1, mu2, E[x^2], E[y^2], E[xy];
//                         writes the three per-pixel partial derivatives dS/dmu1, dS/dE[x^2], dS/dE[xy] and adds the tile's
//                         SSIM sum to *sum (one atomic per workgroup);
//   ssim_backward_kernel  the window is symmetric, so dL/dx = scale * (W*dS/dmu1 + 2x W*dS/dE[x^2] + y W*dS/dE[xy])
//                         with the same tiling over the three partial maps.
// HBM-bound streaming kernels: 5 map reads + 3 writes forward, 5 reads + 1 write backward per channel.
#include "common.hpp"

namespace r3dg {

constexpr int SSIM_R = 5;                    // window radius (window_size 11)
constexpr int SSIM_T = 32;                   // tile edge (outputs)
constexpr int SSIM_E = SSIM_T + 2 * SSIM_R;  // 42: tile + halo
constexpr int SSIM_B = 4;                    // outputs per thread along the filtered axis (sliding window in registers)
constexpr int SSIM_NLOAD = (SSIM_E * SSIM_E + 255) / 256;     // halo-region elements per thread (7)

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

// Both kernels: one 32x32 output tile per 256-thread workgroup.  Horizontal pass: a work item = (row of the 42-row
// halo region, group of 4 adjacent columns): 14 LDS reads per map feed 4 outputs.  Vertical pass: a work item = (column,
// group of 4 adjacent rows), exactly one per thread.
// One launch serves one or two images against the same target shape (the SH image and the sRGB PBR image of a stage-2
// iteration): blockIdx.z = image * C + channel.  Twice the workgroups per launch halve the share of the partially filled
// last wave of workgroups (a 800x800x3 image is 1875 of them, 768 resident) and the number of launches.
struct SsimImage {
    const float* x;
    const float* y;
    float* partials;       // forward: written, backward: read
    float* sum;            // forward only
    float* grad_x;         // backward only
    float scale;           // backward only
};
struct SsimBatch {
    SsimImage im[2];
};

__global__ void __launch_bounds__(256)
ssim_forward_kernel(int W, int H, int C, SsimBatch batch)
{
    const SsimImage I = batch.im[blockIdx.z / C];
    const float* __restrict__ x = I.x;
    const float* __restrict__ y = I.y;
    float* __restrict__ partials = I.partials;
    float* __restrict__ sum = I.sum;
    // the row-filtered maps take the place of the staged inputs (results wait in registers across a barrier): 27 KB
    // instead of 42 KB per workgroup = 5 resident workgroups per CU instead of 3 for a kernel that lives on latency hiding
    constexpr int IN_FLOATS = 2 * SSIM_E * (SSIM_E + 1), H_FLOATS = 5 * SSIM_E * (SSIM_T + 1);
    __shared__ float s_raw[IN_FLOATS > H_FLOATS ? IN_FLOATS : H_FLOATS];
    float (*s_x)[SSIM_E + 1] = reinterpret_cast<float (*)[SSIM_E + 1]>(s_raw);
    float (*s_y)[SSIM_E + 1] = reinterpret_cast<float (*)[SSIM_E + 1]>(s_raw + SSIM_E * (SSIM_E + 1));
    float (*s_h)[SSIM_E][SSIM_T + 1] = reinterpret_cast<float (*)[SSIM_E][SSIM_T + 1]>(s_raw);
    __shared__ float s_part[4];
    const size_t HW = (size_t)H * W;
    const int c = blockIdx.z % C;
    const float* xc = x + c * HW;
    const float* yc = y + c * HW;
    const int bx = blockIdx.x * SSIM_T, by = blockIdx.y * SSIM_T;
    // (every load of the halo region is issued before the first one is used: written as a plain loop the compiler waits for
    // each load before it issues the next -- 14 memory latencies in a row per workgroup, which was most of this kernel's time)
    {
        float xv[SSIM_NLOAD], yv[SSIM_NLOAD];
#pragma unroll
        for (int j = 0; j < SSIM_NLOAD; j++) {
            const int i = threadIdx.x + j * 256;
            const int r = i / SSIM_E, q = i % SSIM_E;
            const int gy = by + r - SSIM_R, gx = bx + q - SSIM_R;
            const bool in = i < SSIM_E * SSIM_E && gx >= 0 && gx < W && gy >= 0 && gy < H;     // zero padding
            const size_t o = (size_t)(in ? gy : 0) * W + (in ? gx : 0);
            const float vx = xc[o], vy = yc[o];
            xv[j] = in ? vx : 0.f;
            yv[j] = in ? vy : 0.f;
        }
#pragma unroll
        for (int j = 0; j < SSIM_NLOAD; j++) {
            const int i = threadIdx.x + j * 256;
            if (i < SSIM_E * SSIM_E) {
                s_x[i / SSIM_E][i % SSIM_E] = xv[j];
                s_y[i / SSIM_E][i % SSIM_E] = yv[j];
            }
        }
    }
    __syncthreads();
    float w[11];
#pragma unroll
    for (int k = 0; k < 11; k++) w[k] = kSsimWin[k];
    constexpr int ITEMS = SSIM_E * (SSIM_T / SSIM_B), ROUNDS = (ITEMS + 255) / 256;     // horizontal taps
    float hres[ROUNDS][5][SSIM_B];
#pragma unroll
    for (int rd = 0; rd < ROUNDS; rd++) {
        const int i = threadIdx.x + rd * 256;
        if (i < ITEMS) {
            const int r = i / (SSIM_T / SSIM_B), q0 = (i % (SSIM_T / SSIM_B)) * SSIM_B;
            float xv[SSIM_B + 10], yv[SSIM_B + 10];
#pragma unroll
            for (int k = 0; k < SSIM_B + 10; k++) { xv[k] = s_x[r][q0 + k]; yv[k] = s_y[r][q0 + k]; }
#pragma unroll
            for (int o = 0; o < SSIM_B; o++) {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
                for (int k = 0; k < 11; k++) {
                    const float wx = w[k] * xv[o + k], wy = w[k] * yv[o + k];
                    a0 += wx; a1 += wy; a2 += wx * xv[o + k]; a3 += wy * yv[o + k]; a4 += wx * yv[o + k];
                }
                hres[rd][0][o] = a0; hres[rd][1][o] = a1; hres[rd][2][o] = a2; hres[rd][3][o] = a3; hres[rd][4][o] = a4;
            }
        }
    }
    __syncthreads();                                   // every read of the staged inputs is done: reuse their storage
#pragma unroll
    for (int rd = 0; rd < ROUNDS; rd++) {
        const int i = threadIdx.x + rd * 256;
        if (i < ITEMS) {
            const int r = i / (SSIM_T / SSIM_B), q0 = (i % (SSIM_T / SSIM_B)) * SSIM_B;
#pragma unroll
            for (int m = 0; m < 5; m++)
#pragma unroll
                for (int o = 0; o < SSIM_B; o++) s_h[m][r][q0 + o] = hres[rd][m][o];
        }
    }
    __syncthreads();
    const int tx = threadIdx.x % SSIM_T, ty0 = (threadIdx.x / SSIM_T) * SSIM_B;
    const int px = bx + tx;
    float acc[5][SSIM_B];
#pragma unroll
    for (int m = 0; m < 5; m++) {
        float col[SSIM_B + 10];
#pragma unroll
        for (int k = 0; k < SSIM_B + 10; k++) col[k] = s_h[m][ty0 + k][tx];
#pragma unroll
        for (int o = 0; o < SSIM_B; o++) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) a += w[k] * col[o + k];
            acc[m][o] = a;
        }
    }
    float ssim_sum = 0.f;
    float* pc = partials + (size_t)c * 3 * HW;
#pragma unroll
    for (int o = 0; o < SSIM_B; o++) {
        const int py = by + ty0 + o;
        if (px < W && py < H) {
            const float mu1 = acc[0][o], mu2 = acc[1][o], e11 = acc[2][o], e22 = acc[3][o], e12 = acc[4][o];
            const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
            const float A = 2.f * mu1 * mu2 + C1;
            const float B = 2.f * (e12 - mu1 * mu2) + C2;
            const float C = mu1 * mu1 + mu2 * mu2 + C1;
            const float D = (e11 - mu1 * mu1) + (e22 - mu2 * mu2) + C2;
            const float inv = 1.f / (C * D);
            const float ssim = A * B * inv;
            ssim_sum += ssim;
            // partial derivatives of ssim w.r.t. mu1, E[x^2], E[xy] (treated as independent window averages)
            const size_t off = (size_t)py * W + px;
            pc[off] = 2.f * mu2 * (B - A) * inv - ssim * 2.f * mu1 * (D - C) * inv;
            pc[HW + off] = -ssim / D;
            pc[2 * HW + off] = 2.f * A * inv;
        }
    }
    const float tot = block_sum_256s(ssim_sum, s_part);
    if (threadIdx.x == 0 && sum != nullptr) atomicAdd(sum_slot(sum), tot);
}

__global__ void __launch_bounds__(256)
ssim_backward_kernel(int W, int H, int C, SsimBatch batch)
{
    const SsimImage I = batch.im[blockIdx.z / C];
    const float* __restrict__ x = I.x;
    const float* __restrict__ y = I.y;
    const float* __restrict__ partials = I.partials;
    const float scale = I.scale;
    float* __restrict__ grad_x = I.grad_x;
    constexpr int IN_FLOATS = 3 * SSIM_E * (SSIM_E + 1);                       // (row-filtered maps alias the inputs)
    __shared__ float s_raw[IN_FLOATS];
    float (*s_p)[SSIM_E][SSIM_E + 1] = reinterpret_cast<float (*)[SSIM_E][SSIM_E + 1]>(s_raw);
    float (*s_h)[SSIM_E][SSIM_T + 1] = reinterpret_cast<float (*)[SSIM_E][SSIM_T + 1]>(s_raw);
    const size_t HW = (size_t)H * W;
    const int c = blockIdx.z % C;
    const float* pc = partials + (size_t)c * 3 * HW;
    const int bx = blockIdx.x * SSIM_T, by = blockIdx.y * SSIM_T;
    // the thread's own four output pixels of x and y (read by the last step): requested first, they arrive under everything else
    const int tx = threadIdx.x % SSIM_T, ty0 = (threadIdx.x / SSIM_T) * SSIM_B;
    const int px = bx + tx;
    float xo[SSIM_B], yo[SSIM_B];
#pragma unroll
    for (int o = 0; o < SSIM_B; o++) {
        const int py = by + ty0 + o;
        const bool in = px < W && py < H;
        const size_t off = in ? (size_t)py * W + px : 0;
        xo[o] = x[c * HW + off];
        yo[o] = y[c * HW + off];
    }
    {
        float pv[3][SSIM_NLOAD];               // (all loads in flight at once, see ssim_forward_kernel)
#pragma unroll
        for (int j = 0; j < SSIM_NLOAD; j++) {
            const int i = threadIdx.x + j * 256;
            const int r = i / SSIM_E, q = i % SSIM_E;
            const int gy = by + r - SSIM_R, gx = bx + q - SSIM_R;
            // windows centred outside the image do not exist
            const bool in = i < SSIM_E * SSIM_E && gx >= 0 && gx < W && gy >= 0 && gy < H;
            const size_t o = (size_t)(in ? gy : 0) * W + (in ? gx : 0);
            const float v0 = pc[o], v1 = pc[HW + o], v2 = pc[2 * HW + o];
            pv[0][j] = in ? v0 : 0.f;
            pv[1][j] = in ? v1 : 0.f;
            pv[2][j] = in ? v2 : 0.f;
        }
#pragma unroll
        for (int j = 0; j < SSIM_NLOAD; j++) {
            const int i = threadIdx.x + j * 256;
            if (i < SSIM_E * SSIM_E) {
                s_p[0][i / SSIM_E][i % SSIM_E] = pv[0][j];
                s_p[1][i / SSIM_E][i % SSIM_E] = pv[1][j];
                s_p[2][i / SSIM_E][i % SSIM_E] = pv[2][j];
            }
        }
    }
    __syncthreads();
    float w[11];
#pragma unroll
    for (int k = 0; k < 11; k++) w[k] = kSsimWin[k];
    constexpr int ITEMS = SSIM_E * (SSIM_T / SSIM_B), ROUNDS = (ITEMS + 255) / 256;
    float hres[ROUNDS][3][SSIM_B];
#pragma unroll
    for (int rd = 0; rd < ROUNDS; rd++) {
        const int i = threadIdx.x + rd * 256;
        if (i < ITEMS) {
            const int r = i / (SSIM_T / SSIM_B), q0 = (i % (SSIM_T / SSIM_B)) * SSIM_B;
#pragma unroll
            for (int m = 0; m < 3; m++) {
                float v[SSIM_B + 10];
#pragma unroll
                for (int k = 0; k < SSIM_B + 10; k++) v[k] = s_p[m][r][q0 + k];
#pragma unroll
                for (int o = 0; o < SSIM_B; o++) {
                    float a = 0.f;
#pragma unroll
                    for (int k = 0; k < 11; k++) a += w[k] * v[o + k];
                    hres[rd][m][o] = a;
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int rd = 0; rd < ROUNDS; rd++) {
        const int i = threadIdx.x + rd * 256;
        if (i < ITEMS) {
            const int r = i / (SSIM_T / SSIM_B), q0 = (i % (SSIM_T / SSIM_B)) * SSIM_B;
#pragma unroll
            for (int m = 0; m < 3; m++)
#pragma unroll
                for (int o = 0; o < SSIM_B; o++) s_h[m][r][q0 + o] = hres[rd][m][o];
        }
    }
    __syncthreads();
    float acc[3][SSIM_B];
#pragma unroll
    for (int m = 0; m < 3; m++) {
        float col[SSIM_B + 10];
#pragma unroll
        for (int k = 0; k < SSIM_B + 10; k++) col[k] = s_h[m][ty0 + k][tx];
#pragma unroll
        for (int o = 0; o < SSIM_B; o++) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) a += w[k] * col[o + k];
            acc[m][o] = a;
        }
    }
#pragma unroll
    for (int o = 0; o < SSIM_B; o++) {
        const int py = by + ty0 + o;
        if (px < W && py < H) {
            const size_t off = (size_t)py * W + px;
            grad_x[c * HW + off] = scale * (acc[0][o] + 2.f * xo[o] * acc[1][o] + yo[o] * acc[2][o]);
        }
    }
}

void launch_ssim_forward(hipStream_t s, int W, int H, int C, int n_images, const float* const* x, const float* y,
                         float* const* partials, float* const* sum)
{
    SsimBatch b = {};
    for (int i = 0; i < n_images; i++) b.im[i] = SsimImage{x[i], y, partials[i], sum[i], nullptr, 0.f};
    dim3 grid((W + SSIM_T - 1) / SSIM_T, (H + SSIM_T - 1) / SSIM_T, C * n_images);
    ssim_forward_kernel<<<grid, 256, 0, s>>>(W, H, C, b);
    check_launch(s, false, "ssim_forward_kernel");
}

void launch_ssim_backward(hipStream_t s, int W, int H, int C, int n_images, const float* const* x, const float* y,
                          float* const* partials, const float* scale, float* const* grad_x)
{
    SsimBatch b = {};
    for (int i = 0; i < n_images; i++) b.im[i] = SsimImage{x[i], y, partials[i], nullptr, grad_x[i], scale[i]};
    dim3 grid((W + SSIM_T - 1) / SSIM_T, (H + SSIM_T - 1) / SSIM_T, C * n_images);
    ssim_backward_kernel<<<grid, 256, 0, s>>>(W, H, C, b);
    check_launch(s, false, "ssim_backward_kernel");
}

}  // namespace r3dg
