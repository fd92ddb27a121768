// K9 + K10 of the rasterizer forward as one per-pixel device function (shared by pseudo_normal_kernel, rasterizer_render_fwd.hip, and
// the fused tail kernel s2_normals_srgb_kernel, stage2_glue.hip).
// K9: surface point in camera space from the premultiplied depth / opacity buffers (forward.cu:398-425);
// K10: pseudo normal from a 3x3 edge-clamped stencil on those points (forward.cu:427-491).  The reference runs K10 behind a
// grid-wide barrier on K9's output; here every thread forms the nine points of its stencil itself, with K9's expression (the
// same values bit for bit: nine divisions instead of one per pixel, on a launch that waits for memory), and stores its own.
#pragma once
#include "common.hpp"

namespace r3dg {

__device__ __forceinline__ void surface_point(int x, int y, int W, float focal_x, float focal_y, float cx, float cy,
                                              const float* __restrict__ opacities, const float* __restrict__ depths, float (&p)[3])
{
    const size_t id = (size_t)y * W + x;
    const float depth = depths[id] / fmaxf(opacities[id], 0.0000001f);
    p[0] = (x - cx) / focal_x * depth;
    p[1] = (y - cy) / focal_y * depth;
    p[2] = depth;
}

// pixel (x, y), inside the image
__device__ __forceinline__ void pseudo_normal_pixel(int x, int y, int W, int H, float focal_x, float focal_y, float cx, float cy,
                                                    const float* __restrict__ vm, const float* __restrict__ opacities,
                                                    const float* __restrict__ depths, float* __restrict__ normals,
                                                    float* __restrict__ surface_xyz)
{
    const size_t HW = (size_t)H * W;
    const int ys[3] = {y == 0 ? 0 : y - 1, y, y == H - 1 ? H - 1 : y + 1};
    const int xs[3] = {x == 0 ? 0 : x - 1, x, x == W - 1 ? W - 1 : x + 1};
    float s[3][3][3];                                        // [row][column][component]
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) surface_point(xs[b], ys[a], W, focal_x, focal_y, cx, cy, opacities, depths, s[a][b]);
    const size_t i11 = (size_t)W * y + x;
    float ga[3], gb[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        surface_xyz[i * HW + i11] = s[1][1][i];
        ga[i] = -0.125f * s[0][0][i] + 0.125f * s[0][2][i] - 0.25f * s[1][0][i] + 0.25f * s[1][2][i] - 0.125f * s[2][0][i] +
                0.125f * s[2][2][i];
        gb[i] = -0.125f * s[0][0][i] - 0.25f * s[0][1][i] - 0.125f * s[0][2][i] + 0.125f * s[2][0][i] + 0.25f * s[2][1][i] +
                0.125f * s[2][2][i];
    }
    float nx = ga[1] * gb[2] - ga[2] * gb[1];
    float ny = -ga[0] * gb[2] + ga[2] * gb[0];
    float nz = ga[0] * gb[1] - ga[1] * gb[0];
    const float norm = sqrtf(nx * nx + ny * ny + nz * nz);
    if (norm <= 0.0f) {            // the reference leaves its zero-initialised output untouched here
        normals[i11] = 0.f;
        normals[HW + i11] = 0.f;
        normals[2 * HW + i11] = 0.f;
        return;
    }
    nx = -nx / norm; ny = -ny / norm; nz = -nz / norm;
    normals[i11] = vm[0] * nx + vm[1] * ny + vm[2] * nz;
    normals[HW + i11] = vm[4] * nx + vm[5] * ny + vm[6] * nz;
    normals[2 * HW + i11] = vm[8] * nx + vm[9] * ny + vm[10] * nz;
}

}  // namespace r3dg
