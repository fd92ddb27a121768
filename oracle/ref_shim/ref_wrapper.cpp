// TEST INFRASTRUCTURE ONLY -- C wrapper around the REAL reference kernels (compiled unmodified from /root/reference
// for gfx950 by oracle/build_ref.py) so the GPU tests can compare this repo's HIP kernels with the reference's own
// CUDA kernels on identical inputs.  Calls CudaRasterizer::Rasterizer::forward/backward (cuda_rasterizer/rasterizer.h),
// construct_bvh (bvh/include/construct.cuh), trace_bvh_opacity_cuda / trace_bvh_cuda (bvh/include/trace.cuh) and SimpleKNN::knn
// (submodules/simple-knn/simple_knn.h).  Never linked into
// libr3dg_hip.so; the reference launches on the null stream.
#include <hip/hip_runtime.h>
#include <functional>
#include <cstdint>
#include "rasterizer.h"
#include "rasterizer_impl.h"
#include "construct.cuh"
#include "trace.cuh"
#include "simple_knn.h"

typedef void* (*alloc_fn)(void* user, size_t bytes);

extern "C" {

int ref_rasterize_forward(alloc_fn geom, alloc_fn binning, alloc_fn img, void* user, int P, int S, int D, int M,
                          const float* background, int W, int H, const float* means3D, const float* shs,
                          const float* colors_precomp, const float* features, const float* opacities,
                          const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                          const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                          float tan_fovy, float cx, float cy, int pseudo_normal, float* out_color, float* out_opacity,
                          float* out_depth, float* out_feature, float* out_normal, float* out_xyz, float* out_weights,
                          int* radii)
{
    auto mk = [user](alloc_fn f) { return std::function<char*(size_t)>([f, user](size_t n) { return (char*)f(user, n); }); };
    int r = CudaRasterizer::Rasterizer::forward(mk(geom), mk(binning), mk(img), P, S, D, M, background, W, H, means3D, shs,
                                                colors_precomp, features, opacities, scales, scale_modifier, rotations,
                                                cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, cx,
                                                cy, false, pseudo_normal != 0, out_color, out_opacity, out_depth,
                                                out_feature, out_normal, out_xyz, out_weights, radii, false);
    (void)hipDeviceSynchronize();
    return r;
}

// copies n_contrib / final_T / ranges and the sorted point list out of the reference's private state buffers
void ref_decode_state(char* img_buffer, char* binning_buffer, int W, int H, int R, uint32_t* n_contrib, float* final_T,
                      uint32_t* ranges, uint32_t* point_list, uint64_t* keys)
{
    char* p = img_buffer;
    CudaRasterizer::ImageState is = CudaRasterizer::ImageState::fromChunk(p, (size_t)W * H);
    const int T = ((W + 15) / 16) * ((H + 15) / 16);
    (void)hipMemcpy(n_contrib, is.n_contrib, sizeof(uint32_t) * W * H, hipMemcpyDeviceToDevice);
    (void)hipMemcpy(final_T, is.accum_alpha, sizeof(float) * W * H, hipMemcpyDeviceToDevice);
    (void)hipMemcpy(ranges, is.ranges, sizeof(uint32_t) * 2 * T, hipMemcpyDeviceToDevice);
    if (R > 0) {
        char* b = binning_buffer;
        CudaRasterizer::BinningState bs = CudaRasterizer::BinningState::fromChunk(b, (size_t)R);
        (void)hipMemcpy(point_list, bs.point_list, sizeof(uint32_t) * R, hipMemcpyDeviceToDevice);
        (void)hipMemcpy(keys, bs.point_list_keys, sizeof(uint64_t) * R, hipMemcpyDeviceToDevice);
    }
    (void)hipDeviceSynchronize();
}

void ref_rasterize_backward(int P, int S, int D, int M, int R, const float* background, int W, int H,
                            const float* means3D, const float* shs, const float* features, const float* colors_precomp,
                            const float* scales, float scale_modifier, const float* rotations,
                            const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                            const float* campos, float tan_fovx, float tan_fovy, const int* radii, char* geom_buffer,
                            char* binning_buffer, char* img_buffer, const float* dL_dpix, const float* dL_dpix_o,
                            const float* dL_dpix_d, const float* dL_dpix_f, float* dL_dmean2D, float* dL_dconic,
                            float* dL_dopacity, float* dL_dcolor, float* dL_dfeature, float* dL_dmean3D,
                            float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, int backward_geometry)
{
    CudaRasterizer::Rasterizer::backward(P, S, D, M, R, background, W, H, means3D, shs, features, colors_precomp, scales,
                                         scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, campos,
                                         tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer, img_buffer, dL_dpix,
                                         dL_dpix_o, dL_dpix_d, dL_dpix_f, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor,
                                         dL_dfeature, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot,
                                         backward_geometry != 0, false);
    (void)hipDeviceSynchronize();
}

void ref_bvh_build(int P, const float* means3D, const float* scales, const float* rotations, int32_t* nodes, float* aabbs,
                   uint64_t* morton)
{
    construct_bvh(P, means3D, scales, rotations, nodes, aabbs, morton);
    (void)hipDeviceSynchronize();
}

void ref_bvh_trace_opacity(int num_rays, int32_t* nodes, float* aabbs, float* rays_o, float* rays_d, float* means3D,
                           float* covs3D, float* opacities, float* normals, int32_t* contributes, float* opacity)
{
    trace_bvh_opacity_cuda(num_rays, nodes, aabbs, (float3*)rays_o, (float3*)rays_d, (float3*)means3D, covs3D, opacities,
                           (float3*)normals, contributes, opacity);
    (void)hipDeviceSynchronize();
}

// trace_bvh_cuda (bvh/src/trace.cu:8-192): returns num_rendered; copies the lists when they fit into `capacity` entries
int ref_bvh_trace(int num_rays, int32_t* nodes, float* aabbs, float* rays_o, float* rays_d, float* means3D, float* covs3D,
                  float* opacities, int32_t* num_contributes, int capacity, int32_t* point_list, float* position_list,
                  int32_t* ray_id_list)
{
    auto res = trace_bvh_cuda(num_rays, nodes, aabbs, (float3*)rays_o, (float3*)rays_d, (float3*)means3D, covs3D, opacities,
                              num_contributes);
    (void)hipDeviceSynchronize();
    const int n = std::get<0>(res);
    if (n > 0 && n <= capacity) {
        (void)hipMemcpy(point_list, thrust::raw_pointer_cast(std::get<1>(res).data()), sizeof(int32_t) * n, hipMemcpyDeviceToDevice);
        (void)hipMemcpy(position_list, thrust::raw_pointer_cast(std::get<2>(res).data()), sizeof(float) * 3 * n,
                        hipMemcpyDeviceToDevice);
        (void)hipMemcpy(ray_id_list, thrust::raw_pointer_cast(std::get<3>(res).data()), sizeof(int32_t) * n, hipMemcpyDeviceToDevice);
        (void)hipDeviceSynchronize();
    }
    return n;
}

// SimpleKNN::knn (submodules/simple-knn/simple_knn.cu:185)
void ref_knn_dist2(int P, float* points, float* mean_dists)
{
    SimpleKNN::knn(P, (float3*)points, mean_dists);
    (void)hipDeviceSynchronize();
}

}  // extern "C"
