// C ABI of libr3dg_hip.so (declared in include/r3dg_hip.h): orchestration of the rasterizer forward/backward,
// state-buffer layouts, error plumbing.  Mirrors CudaRasterizer::Rasterizer::forward/backward
// (rasterizer_impl.cu:199-380, :384-491) in the order of work, not in code.
#include "common.hpp"
#include <cstdlib>
#include <new>
#include "../../include/r3dg_hip.h"

#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace r3dg {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }

// launchers implemented in the kernel translation units
void launch_mark_visible(hipStream_t s, int P, const float* means3D, const float* vm, uint8_t* present);
void launch_preprocess(hipStream_t s, int P, int D, int M, const float* means3D, const float* scales,
                       float scale_modifier, const float* rotations, const float* opacities, const float* shs,
                       uint8_t* clamped, const float* cov3D_precomp, const float* colors_precomp, const float* vm,
                       const float* pm, const float* cam_pos, int W, int H, float tan_fovx, float tan_fovy,
                       float focal_x, float focal_y, int* radii, float* means2D, float* depths, float* cov3Ds,
                       float* rgb, float* conic_opacity, float* splat, int gx, int gy, uint32_t* tiles_touched,
                       uint32_t* block_sums, unsigned long long* total, bool scan_now, uint32_t* zero_words, int zero_n);
void launch_duplicate_with_keys(hipStream_t s, int P, const float* means2D, const float* depths,
                                const uint32_t* tiles_touched, const uint32_t* block_offsets,
                                uint32_t* point_offsets, uint64_t* keys, uint32_t* values, const int* radii, int gx,
                                int gy);
void launch_identify_tile_ranges(hipStream_t s, int L, const uint64_t* keys, uint32_t* ranges);
void launch_tile_order(hipStream_t s, int T, const uint32_t* ranges, uint32_t* order, uint32_t small_cap,
                       uint32_t* big_list, uint32_t* big_count);
void launch_render_forward(hipStream_t s, int W, int H, int S, const uint32_t* tile_order, const uint32_t* ranges,
                           const uint32_t* point_list, const float* splat, const float* features, float* final_T,
                           uint32_t* n_contrib, const float* bg, float* out_color, float* out_opacity, float* out_depth,
                           float* out_feature, float* out_weights);
void launch_pseudo_normal(hipStream_t s, int W, int H, const float* vm, float focal_x, float focal_y, float cx,
                          float cy, const float* opacities, const float* depths, float* normals, float* surface_xyz,
                          bool debug);
void launch_render_backward(hipStream_t s, int P, int W, int H, int S, int n_active, const int* active,
                            const uint32_t* tile_order, const uint32_t* ranges,
                            const uint32_t* point_list,
                            const float* bg, const float* splat, const float* features, const float* final_Ts,
                            const uint32_t* n_contrib, const float* dL_dpix, const float* dL_dpix_o,
                            const float* dL_dpix_d, const float* dL_dpix_f, float* dL_dmean2D, float* dL_dconic,
                            float* dL_dopacity, float* dL_dcolor, float* dL_dfeature, int bg_geom);
void launch_render_backward_features(hipStream_t s, int W, int H, int S, int n_active, const int* active,
                                     const uint32_t* tile_order, const uint32_t* ranges, const uint32_t* point_list,
                                     const float* splat, const float* final_Ts, const uint32_t* n_contrib,
                                     const float* dL_dpix_f, float* dL_dfeature);
void launch_preprocess_backward(hipStream_t s, int P, int D, int M, const float* means, const int* radii,
                                const float* shs, const uint8_t* clamped, const float* scales, const float* rotations,
                                float scale_modifier, const float* cov3Ds, const float* vm, const float* proj, float h_x,
                                float h_y, float tan_fovx, float tan_fovy, const float* campos, float* dL_dmean2D,
                                const float* dL_dconic, const float* conic_opacity, int W, int H, float* dL_dmeans,
                                const float* dL_dcolor, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot);
void launch_shade_forward(hipStream_t s, int P, int K, int M, const float* base_color, const float* roughness,
                          const float* normals, const float* viewdirs, const float* incidents, const float* env, int He,
                          int We, const float* tr, const float* visibility, const float* dirs, const float* areas,
                          float* out, const uint32_t* taps, bool train_outputs, float uniform_area, bool taps_are_radiance,
                          bool leave_room);
size_t shade_frs_table_floats(int K);
bool shade_frs_supported(int K, int M, int He, int We);
void launch_shade_frs_build_tables(hipStream_t s, int K, const float* zsamples, float* tables);
void launch_shade_frs_classify(hipStream_t s, int P, const float* ray_normals, uint8_t* valid);
void launch_shade_frs_build_taps(hipStream_t s, int P, int K, const float* ray_normals, const float* zsamples, int He, int We,
                                 uint32_t* taps);
void launch_shade_frs_forward_aux(hipStream_t s, int P, const float* incidents, const float* ray_normals, float* cprime);
void launch_shade_frs_forward_main(hipStream_t s, int P, int K, const float* base_color, const float* roughness,
                                   const float* normals, const float* viewdirs, const float* env, int He, int We,
                                   const float* visibility, float uniform_area, const uint32_t* taps, const float* ray_normals,
                                   const float* tables, const uint8_t* valid, const float* cprime, bool leave_room, float* out,
                                   float* feat);
void launch_shade_frs_forward_listed(hipStream_t s, int K, const float* base_color, const float* roughness,
                                     const float* normals, const float* viewdirs, const float* incidents, const float* env, int He,
                                     int We, const float* visibility, const float* ray_normals, const float* zsamples,
                                     float uniform_area, const int* invalid_list, int n_invalid, float* out, float* feat);
void launch_shade_frs_incident_chain(hipStream_t s, int P, const float* ray_normals, const uint8_t* valid, const float* dcp,
                                     float* d_inc, float* incidents, float* exp_avg, float* exp_avg_sq, float* cprime, float lr,
                                     float lr_tail, float beta1, float beta2, float eps, int step, float grad_scale,
                                     const float* skip_flag, int listed_in_dcprime);
const unsigned int* launch_shade_frs_backward_aux(hipStream_t s, int P, const float* g_pbr, const float* g_diff,
                                                  const float* block_absmax, int n_block_absmax, int* gmax_n);
void launch_shade_frs_backward_main(hipStream_t s, int P, int K, const float* base_color, const float* roughness,
                                    const float* normals, const float* viewdirs, const float* env, int He, int We,
                                    const float* visibility, float uniform_area, const uint32_t* taps, const float* ray_normals,
                                    const float* tables, const uint8_t* valid, const float* cprime, float* dcp, const float* g_pbr,
                                    const float* g_diff, float* d_base, float* d_rough, float* d_view, float* d_env,
                                    const unsigned int* gmax, int gmax_n);
void launch_shade_frs_backward_rotate(hipStream_t s, int P, const float* ray_normals, const float* dcp, float* d_inc,
                                      const uint8_t* valid);
void launch_shade_frs_backward_listed(hipStream_t s, int K, const float* base_color, const float* roughness,
                                      const float* normals, const float* viewdirs, const float* incidents, const float* env, int He,
                                      int We, const float* visibility, const float* ray_normals, const float* zsamples,
                                      float uniform_area, const int* invalid_list, int n_invalid, const float* g_pbr,
                                      const float* g_diff, float* d_base, float* d_rough, float* d_view, float* d_inc, float* d_env,
                                      const unsigned int* gmax, int gmax_n);
void launch_shade_build_taps(hipStream_t s, size_t n, const float* dirs, const float* tr, int He, int We, const float* env,
                             uint32_t* taps);
void launch_shade_build_split(hipStream_t s, int P, int K, const int* perm, const float* normals, const float* incidents,
                              const float* visibility, const float* dirs, const float* zsamples, float uniform_area, float* lt,
                              float* vis_t, float* consts);
void launch_shade_forward_split(hipStream_t s, int P, int K, const int* perm, const float* base_color, const float* roughness,
                                const float* normals, const float* viewdirs, const float* lt, const float* vis_t,
                                const float* consts, const float* zsamples, const float* tr, const float* env4, int He, int We,
                                float* out);
void launch_shade_env_footprints(hipStream_t s, int He, int We, const float* env, float* fp);
void launch_shade_build_transport(hipStream_t s, int P, int K, int M, const float* normals, const float* incidents,
                                  const float* visibility, const float* dirs, const float* areas, float uniform_area,
                                  float* radiance_to_transport, float* consts);
void launch_shade_forward_transport(hipStream_t s, int P, int K, const float* base_color, const float* roughness,
                                    const float* normals, const float* viewdirs, const float* transport, const float* consts,
                                    const float* zsamples, const float* dirs, float* out);
extern int g_trace_packet, g_trace_refill, g_trace_node_weight, g_trace_leaf_weight, g_trace_count_visits;
extern int g_bwd_lean;
extern int g_shade_row_blocks_per_cu;
void launch_shade_backward(hipStream_t s, int P, int K, int M, const float* base_color, const float* roughness,
                           const float* normals, const float* viewdirs, const float* incidents, const float* env,
                           int He, int We, const float* tr, const float* visibility, const float* dirs,
                           const float* areas, const float* g_pbr, const float* g_diff, float* d_base, float* d_rough,
                           float* d_view, float* d_inc, float* d_env, const uint32_t* taps, const float* block_absmax, int n_block_absmax);
void launch_re_forward(hipStream_t s, bool complex_, int P, int Si, int Sd, int Sv, const float* base_color,
                       const float* roughness, const float* metallic, const float* normals, const float* viewdirs,
                       const float* inc, const float* direct, const float* vis, int K, const float* rand_float,
                       float* incident_dirs, float* out_pbr, float* out_lights, float* out_local, float* out_global,
                       float* out_vis, float* out_diffuse, float* out_local_diffuse, float* out_accum, float* out_rgb_d,
                       float* out_rgb_s);
void launch_re_backward(hipStream_t s, int P, int Si, int Sd, int Sv, const float* base_color, const float* roughness,
                        const float* metallic, const float* normals, const float* viewdirs, const float* inc,
                        const float* direct, const float* vis, int K, const float* incident_dirs, const float* dL_dpbr,
                        const float* dL_ddl, float* dL_dbase, float* dL_drough, float* dL_dmetal, float* dL_dnormals,
                        float* dL_dviewdirs, float* dL_dinc, float* dL_ddirect, float* dL_dvis);
void launch_s2_activate(hipStream_t s, int P, const float* xyz, const float* scaling_raw, const float* rotation_raw,
                        const float* opacity_raw, const float* normal_raw, const float* base_raw,
                        const float* rough_raw, const float* campos, float* scales, float* rot, float* opacity,
                        float* normal, float* base_color, float* roughness, float* viewdirs, const float* viewmatrix,
                        float* features, int n_env, const float* env_raw, float* env, float* zero, int n_zero);
void launch_s2_pack(hipStream_t s, int P, const float* xyz, const float* viewmatrix, const float* normal,
                    const float* base_color, const float* roughness, const float* shade_out, float* features,
                    float* light_l1_sum);
void launch_s2_unpack(hipStream_t s, int P, const float* dL_dfeatures, const float* shade_out, float light_weight,
                      float* dL_dpbr, float* dL_ddiffuse, float* block_absmax, float* light_l1_sum);
void launch_s2_activate_backward(hipStream_t s, int P, const float* xyz, const float* scaling_raw,
                                 const float* rotation_raw, const float* opacity_raw, const float* normal_raw,
                                 const float* base_raw, const float* rough_raw, const float* viewmatrix,
                                 const float* campos, const float* dL_dfeatures, const float* dL_dbase_shade,
                                 const float* dL_drough_shade, const float* dL_dviewdirs, const float* dL_dscales,
                                 const float* dL_drot, const float* dL_dopacity, const float* dL_dmeans3D, float* g_xyz,
                                 float* g_scaling, float* g_rotation, float* g_opacity, float* g_normal, float* g_base,
                                 float* g_rough, int He, int We, const float* env_raw, const float* env, float* dL_denv,
                                 float w_tv, float* g_env_raw, float* tv_sum, int consume);
void launch_s2_loss(hipStream_t s, int HW, const float* image, const float* opacity, const float* feature,
                    const float* pseudo_normal, const int* n_contrib, const float* gt, const float* bg,
                    const float* image_mask, float w_l1, float w_pbr, float w_normal, const float* extra_dimage,
                    const float* extra_dsrgb, float* dL_dimage, float* dL_dopacity, float* dL_dfeature, float* sums,
                    int sparse);
void launch_s2_smooth_forward(hipStream_t s, int W, int H, const float* opacity, const float* feature, const int* n_contrib,
                              const float* gt, const float* image_mask, float w_base, float w_rough, float w_light,
                              float* scratch, float* sums3);
void launch_s2_smooth_fused(hipStream_t s, int W, int H, const float* opacity, const float* feature, const int* n_contrib,
                            const float* gt, const float* image_mask, float w_base, float w_rough, float w_light,
                            int accumulate_normal, float* dL_dopacity, float* dL_dfeature, float* sums3);
void launch_s2_smooth_backward(hipStream_t s, int W, int H, const float* opacity, const float* feature, const int* n_contrib,
                               const float* image_mask, const float* scratch, int has_base, int has_rough, int has_light,
                               int accumulate_normal, float* dL_dopacity, float* dL_dfeature);
void launch_s2_pbr_srgb(hipStream_t s, int HW, const float* opacity, const float* feature, const int* n_contrib,
                        const float* bg, float* srgb);
void launch_s2_normals_srgb(hipStream_t s, int W, int H, const float* vm, float focal_x, float focal_y, float cx, float cy,
                            const float* opacity, const float* depths, float* normals, float* surface_xyz, const float* feature,
                            const int* n_contrib, const float* bg, float* srgb);
void launch_ssim_forward(hipStream_t s, int W, int H, int C, int n_images, const float* const* x, const float* y,
                         float* const* partials, float* const* sum);
void launch_ssim_backward(hipStream_t s, int W, int H, int C, int n_images, const float* const* x, const float* y,
                          float* const* partials, const float* scale, float* const* grad_x);
void launch_adam(hipStream_t s, int n_groups, const r3dg_adam_group* groups, float beta1, float beta2, float eps,
                 int step, float grad_scale, const float* skip_flag);
void launch_s1_pack(hipStream_t s, int P, const float* xyz, const float* viewmatrix, const float* normal, float* features);
void launch_s1_edge(hipStream_t s, int W, int H, const float* feature, const float* opacity, const int* n_contrib,
                    const float* gt, float* edge_g, float* sum_out);
void launch_s1_loss(hipStream_t s, int W, int H, const float* image, const float* opacity, const float* feature,
                    const float* pseudo_normal, const int* n_contrib, const float* gt, const float* image_mask, float w_l1,
                    float w_entropy, float w_normal, float w_smooth, float w_var, const float* extra_dimage,
                    const float* edge_g, float* dL_dimage, float* dL_dopacity, float* dL_dfeature, float* sums);
void launch_s1_activate_backward(hipStream_t s, int P, const float* xyz, const float* scaling_raw,
                                 const float* rotation_raw, const float* opacity_raw, const float* normal_raw,
                                 const float* viewmatrix, const float* dL_dfeatures, const float* dL_dscales,
                                 const float* dL_drot, const float* dL_dopacity, const float* dL_dmeans3D, float* g_xyz,
                                 float* g_scaling, float* g_rotation, float* g_opacity, float* g_normal);
void launch_s2_env_backward(hipStream_t s, int He, int We, const float* raw, const float* env, float* dL_denv,
                            float w_tv, float* g_raw, float* tv_sum, int consume);
uint32_t tile_sort_small_cap();
void launch_tile_sort(hipStream_t s, int T, const uint32_t* tile_order, const uint32_t* ranges, const uint32_t* big_list,
                      uint32_t* big_count, uint64_t* keys, uint32_t* vals, uint64_t* scratch, bool entries);
void launch_tile_binning(hipStream_t s, int P, int T, const float* means2D, const float* depths, const int* radii,
                         const uint32_t* tiles_touched, uint32_t* block_offsets, int gx, int gy,
                         uint32_t* tile_counts, uint32_t* cursor, uint32_t* ranges, uint32_t* point_offsets,
                         uint64_t* entries, unsigned long long* total, long long capacity, float* overflow_flag,
                         unsigned int* overflow_count, bool fused, uint32_t* order, uint32_t small_cap, uint32_t* big_list,
                         uint32_t* big_count);
int tile_binning_max_tiles();
extern int g_bin_iters;
void launch_densify_accumulate(hipStream_t s, int P, const float* viewspace_grad, const float* normal_grad,
                               const int* radii, const float* weights, float* xyz_accum, float* normal_accum,
                               float* denom, float* weights_accum, float* max_radii2D, const float* skip_flag);
size_t densify_temp_bytes(size_t P);
void launch_densify_plan(hipStream_t s, int P, const r3dg_densify_config& cfg, const float* scaling_raw,
                         const float* opacity_raw, const float* xyz_accum, const float* normal_accum,
                         const float* denom, const float* weights_accum, const float* max_radii2D, int32_t* src_row,
                         int32_t* kind, int32_t* counts, void* temp);
void launch_densify_gather(hipStream_t s, int P_out, const int32_t* src_row, const int32_t* kind, int n_groups,
                           const r3dg_densify_group* groups, const float* xyz, const float* scaling_raw,
                           const float* rotation_raw, const float* normal_table, float split_divisor);
void launch_reset_opacity(hipStream_t s, int P, float cap, float* opacity_raw, float* exp_avg, float* exp_avg_sq);
void launch_relight_pack(hipStream_t s, int P, const float* xyz, const float* viewmatrix, const float* normal,
                         const float* base_color, const float* roughness, const float* shade_out, float* features);
void launch_relight_compose(hipStream_t s, int W, int H, float fx, float fy, float cx, float cy,
                            const float* viewmatrix, const float* tr, const float* env, int He, int We,
                            const float* image, const float* opacity, const float* feature, const int* n_contrib,
                            float* pbr_env, float* render_env, float* env_only);
size_t knn_temp_bytes(size_t P);
void knn_dist2(hipStream_t s, int P, const float* pts, float* dists, void* temp);
size_t bvh_build_temp_bytes(size_t P);
void bvh_build(hipStream_t s, int P, int32_t* nodes, float* aabbs, uint64_t* morton, void* temp);
void bvh_trace_count(hipStream_t s, int num_rays, const int32_t* nodes, const float* aabbs, const float* rays_o,
                     const float* rays_d, int32_t* counts, int* overflow);
void bvh_trace_fill(hipStream_t s, int num_rays, const int32_t* nodes, const float* aabbs, const float* rays_o,
                    const float* rays_d, const float* means, const int32_t* counts, const int64_t* offsets_inclusive,
                    uint64_t* keys, int32_t* points, float* positions, int32_t* ray_ids);
void bvh_trace_opacity(hipStream_t s, int num_rays, int P, const int32_t* nodes, const float* aabbs, const float* rays_o,
                       const float* rays_d, const float* means, const float* covs, const float* opac,
                       const float* normals, int32_t* contributes, float* out, int* overflow);
size_t bvh_trace_records_bytes(size_t P);
void bvh_pack_traversal(hipStream_t s, int P, const int32_t* nodes, const float* aabbs, const float* means, const float* covs,
                        const float* opac, const float* normals, void* records);
void bvh_trace_visits(hipStream_t s, int P, const void* records, unsigned long long out[2]);
void bvh_trace_opacity_packed(hipStream_t s, int num_rays, int P, void* records, const float* rays_o, const float* rays_d,
                              int32_t* contributes, float* out, int* overflow);
int g_reserve_cus = 0;
extern int g_cull;
extern int g_stage_sh_rows;
int g_tile_binning = 2;   // 2: instances emitted straight into their tile's segment + per-tile LDS sort; 1: emitted in Gaussian
                          // order, radix-partitioned by tile, per-tile LDS sort; 0: the reference's global (tile|depth) radix sort
int g_tile_order = 1;   // 1: longest-tile-first block order, 0: XCD-contiguous natural order
void launch_transpose_selftest(hipStream_t s, int N, int dpp, const float* in, float* out, int* chan, int* owner);

// ---- optional per-stage timing with HIP events on the launch stream (bench.py's roofline numbers) ----
enum Stage { ST_PREPROCESS = 0, ST_DUPKEYS, ST_SORT, ST_RANGES, ST_RENDER_FWD, ST_NORMAL, ST_RENDER_BWD, ST_PREPROCESS_BWD,
             ST_SHADE_FWD, ST_SHADE_BWD, ST_SHADE_AUX, ST_SHADE_LISTED, ST_BVH_BUILD, ST_BVH_TRACE, ST_S2_ACTIVATE, ST_S2_PACK, ST_S2_LOSS,
             ST_S2_UNPACK, ST_S2_ACTIVATE_BWD, ST_ADAM, ST_KNN, ST_SSIM, ST_DENSIFY, ST_RELIGHT_PACK, ST_RELIGHT_COMPOSE,
             ST_COUNT };
static const char* kStageNames[ST_COUNT] = {"preprocess", "duplicate_with_keys", "sort_pairs", "identify_tile_ranges",
                                            "render_forward", "pseudo_normal", "render_backward", "preprocess_backward",
                                            "shade_forward", "shade_backward", "shade_frs_aux", "shade_frs_listed", "bvh_build", "bvh_trace",
                                            "stage2_activate", "stage2_pack_features", "stage2_loss",
                                            "stage2_unpack_gradients", "stage2_activate_backward", "adam_step",
                                            "knn_dist2", "ssim", "densify", "relight_pack_features", "relight_compose"};
static int g_profiling = 0;
struct EventPair { hipEvent_t a, b; };
static std::vector<EventPair> g_events[ST_COUNT];
static std::mutex g_prof_mutex;

struct StageTimer {
    hipStream_t s;
    int stage;
    EventPair ev;
    bool on;
    StageTimer(hipStream_t s_, int stage_) : s(s_), stage(stage_), on(g_profiling != 0)
    {
        if (on) {
            R3DG_HIP(hipEventCreate(&ev.a));
            R3DG_HIP(hipEventCreate(&ev.b));
            R3DG_HIP(hipEventRecord(ev.a, s));
        }
    }
    ~StageTimer()
    {
        try { stop(); } catch (...) {}
    }
    void stop()
    {
        if (on) {
            R3DG_HIP(hipEventRecord(ev.b, s));
            std::lock_guard<std::mutex> lk(g_prof_mutex);
            g_events[stage].push_back(ev);
            on = false;
        }
    }
};

// ---- state layouts (opaque to callers; 256-byte aligned sub-arrays) ----
GeometryLayout GeometryLayout::make(size_t P)
{
    GeometryLayout L;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
    L.depths = take(P * 4);
    L.clamped = take(P * 3);
    L.radii = take(P * 4);
    L.means2D = take(P * 8);
    L.cov3D = take(P * 24);
    L.conic_opacity = take(P * 16);
    L.rgb = take(P * 12);
    L.tiles_touched = take(P * 4);
    L.point_offsets = take(P * 4);
    L.block_sums = take(((P + 255) / 256 + 1) * 4);
    L.total = take(8);
    // packed per-Gaussian record read by the tile kernels (64-byte stride = one line per staged instance):
    // [mean.x mean.y conic.x conic.y | conic.z opacity depth 0 | r g b 0 | unused]
    L.splat = take(P * 64);
    L.bytes = o;
    return L;
}
ImageLayout ImageLayout::make(size_t N, size_t T)
{
    ImageLayout L;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
    L.final_T = take(N * 4);
    L.n_contrib = take(N * 4);
    L.ranges = take(T * 8);
    L.tile_order = take(T * 4);
    L.big_list = take(T * 4);        // tile-binned ordering: tiles too long for the small in-LDS sort
    L.big_count = take(256);
    L.bytes = o;
    return L;
}
BinningLayout BinningLayout::make(size_t R)
{
    BinningLayout L;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
    L.keys_unsorted = take(R * 8);
    L.keys = take(R * 8);
    L.vals_unsorted = take(R * 4);
    L.vals = take(R * 4);
    L.sort_temp = take(sort_temp_bytes(R));
    L.bytes = o;
    return L;
}

// reference getHigherMsb (rasterizer_impl.cu:35-50): position of the bit above the MSB of n
static uint32_t higher_msb(uint32_t n)
{
    uint32_t b = 0;
    while (b < 32 && (n >> b)) b++;
    return b;
}

template <typename F>
static int guarded(F&& f)
{
    try {
        return f();
    } catch (const HipError& e) {
        (void)hipGetLastError();
        return e.code == -1 ? R3DG_EINVAL : R3DG_EHIP;
    } catch (const std::exception& e) {
        set_error(e.what());
        return R3DG_EHIP;
    }
}

static int invalid(const std::string& msg)
{
    set_error(msg);
    return R3DG_EINVAL;
}

namespace {
struct ScratchBuf { void* p = nullptr; size_t cap = 0; };
std::mutex g_scratch_mu;
std::map<std::tuple<int, hipStream_t, int>, ScratchBuf> g_scratch;
}

void* stream_scratch(hipStream_t stream, int slot, size_t bytes)
{
    int dev = 0;
    R3DG_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    ScratchBuf& b = g_scratch[std::make_tuple(dev, stream, slot)];
    if (b.cap < bytes) {
        // geometric growth: a scene that densifies outgrows its buffer O(log) times, not at every 12 % (each growth is a stream
        // synchronise + hipFree, a device-wide wait)
        const size_t want = b.p == nullptr ? bytes + bytes / 8 + 4096 : std::max(bytes + 4096, 2 * b.cap);
        if (b.p != nullptr) {
            R3DG_HIP(hipStreamSynchronize(stream));        // only this stream ever used the old buffer
            R3DG_HIP(hipFree(b.p));
            b.p = nullptr;
            b.cap = 0;
        }
        R3DG_HIP(hipMalloc(&b.p, want));
        b.cap = want;
    }
    return b.p;
}

void release_gradient_records();        // rasterizer_render_bwd.hip

static void release_all_scratch()
{
    R3DG_HIP(hipDeviceSynchronize());
    {
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        for (auto& kv : g_scratch)
            if (kv.second.p != nullptr) (void)hipFree(kv.second.p);
        g_scratch.clear();
    }
    release_gradient_records();
}

}  // namespace r3dg

using namespace r3dg;

extern "C" {

const char* r3dg_last_error(void) { return g_last_error.c_str(); }
int r3dg_release_scratch(void)
{
    return guarded([&]() {
        release_all_scratch();
        return R3DG_OK;
    });
}
int r3dg_version(void) { return 100; }
int r3dg_max_features_forward(void) { return R3DG_MAX_S_FWD; }
int r3dg_max_features_backward(void) { return R3DG_MAX_S_BWD; }
int r3dg_bounded_forward_supported(int width, int height)
{
    if (width <= 0 || height <= 0) return 0;
    const long long gx = (width + R3DG_TILE_X - 1) / R3DG_TILE_X, gy = (height + R3DG_TILE_Y - 1) / R3DG_TILE_Y;
    return opt(R3DG_OPT_TILE_BINNING) == 2 && gx * gy <= (long long)tile_binning_max_tiles() ? 1 : 0;
}









// tuning / experiment knobs; not part of the drop-in surface (include/r3dg_hip.h "r3dg_option")
static int* option_slot(int option)
{
    switch (option) {
        case R3DG_OPT_TILE_ORDER: return &g_tile_order;
        case R3DG_OPT_CULL: return &g_cull;
        case R3DG_OPT_TILE_BINNING: return &g_tile_binning;
        case R3DG_OPT_BINNING_BLOCK_K: return &g_bin_iters;
        case R3DG_OPT_STAGE_SH_ROWS: return &g_stage_sh_rows;
        case R3DG_OPT_SHADE_FWD_BLOCKS_PER_CU: return &g_shade_row_blocks_per_cu;
        case R3DG_OPT_TRACE_FORMULATION: return &g_trace_packet;
        case R3DG_OPT_TRACE_REFILL: return &g_trace_refill;
        case R3DG_OPT_TRACE_NODE_WEIGHT: return &g_trace_node_weight;
        case R3DG_OPT_TRACE_LEAF_WEIGHT: return &g_trace_leaf_weight;
        case R3DG_OPT_RESERVE_CUS: return &g_reserve_cus;
        case R3DG_OPT_TRACE_COUNT_VISITS: return &g_trace_count_visits;
        case R3DG_OPT_BWD_LEAN: return &g_bwd_lean;
        default: return nullptr;
    }
}

static bool option_in_range(int option, int value)
{
    static const int lo[R3DG_OPT_COUNT] = {0, 0, 0, 1, 0, 0, 0, 1, 1, 1, 0, 0, 0};
    static const int hi[R3DG_OPT_COUNT] = {1, 1, 2, 4, 1, 8, 4, 64, 15, 15, 128, 1, 1};
    return value >= lo[option] && value <= hi[option];
}

int r3dg_set_option(int option, int value)
{
    int* slot = option_slot(option);
    if (slot == nullptr) return invalid("set_option: unknown option");
    if (!option_in_range(option, value)) return invalid("set_option: value out of range");
    *slot = value;
    return R3DG_OK;
}

// ---- option contexts: per-object settings instead of process-global ones ------------------------------------------------------
struct r3dg_context_ {
    int value[R3DG_OPT_COUNT];
    unsigned int set_mask;
};
static thread_local const r3dg_context_* tl_context = nullptr;

}  // extern "C"
namespace r3dg {
int opt(int option)
{
    const r3dg_context_* c = tl_context;
    if (c != nullptr && ((c->set_mask >> option) & 1u)) return c->value[option];
    return *option_slot(option);
}
}  // namespace r3dg
extern "C" {

void* r3dg_context_create(void)
{
    r3dg_context_* c = new (std::nothrow) r3dg_context_();
    if (c != nullptr) c->set_mask = 0u;
    return c;
}

void r3dg_context_destroy(void* ctx)
{
    if (tl_context == ctx) tl_context = nullptr;
    delete static_cast<r3dg_context_*>(ctx);
}

int r3dg_context_set_option(void* ctx, int option, int value)
{
    if (ctx == nullptr || option_slot(option) == nullptr) return invalid("context_set_option: null context or unknown option");
    if (!option_in_range(option, value)) return invalid("context_set_option: value out of range");
    r3dg_context_* c = static_cast<r3dg_context_*>(ctx);
    c->value[option] = value;
    c->set_mask |= 1u << option;
    return R3DG_OK;
}

int r3dg_context_make_current(void* ctx, void** previous)
{
    if (previous != nullptr) *previous = const_cast<r3dg_context_*>(tl_context);
    tl_context = static_cast<const r3dg_context_*>(ctx);
    return R3DG_OK;
}

int r3dg_get_option(int option, int* value)
{
    int* slot = option_slot(option);
    if (slot == nullptr || value == nullptr) return invalid("get_option: unknown option or null pointer");
    *value = r3dg::opt(option);              // (what a launch on this thread would see right now)
    return R3DG_OK;
}

int r3dg_selftest_transpose_reduce(void* stream_, int N, int dpp, const float* d_in, float* d_out, int* d_chan,
                                   int* d_owner)
{
    if (N != 12 && N != 16 && N != 32 && N != 64) return invalid("selftest_transpose_reduce: N must be 12, 16, 32 or 64");
    return guarded([&]() -> int {
        launch_transpose_selftest((hipStream_t)stream_, N, dpp, d_in, d_out, d_chan, d_owner);
        check_launch((hipStream_t)stream_, true, "transpose_selftest");
        return R3DG_OK;
    });
}

// Per-stage HIP-event timing. r3dg_profile_enable(1) starts recording (and clears), r3dg_profile_read waits for
// the recorded events and returns, per stage, the summed milliseconds and the number of timed launches.
int r3dg_profile_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_prof_mutex);
    for (int s = 0; s < ST_COUNT; s++) {
        for (auto& e : g_events[s]) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
        g_events[s].clear();
    }
    g_profiling = on;
    return R3DG_OK;
}
// suspend / resume the recording without discarding what was recorded (sampled profiling: the event pairs cost ~2 us of
// host time each, so a caller may time every n-th iteration only)
int r3dg_profile_pause(int paused)
{
    std::lock_guard<std::mutex> lk(g_prof_mutex);
    g_profiling = paused ? 0 : 1;
    return R3DG_OK;
}
int r3dg_profile_num_stages(void) { return ST_COUNT; }
const char* r3dg_profile_stage_name(int stage) { return (stage >= 0 && stage < ST_COUNT) ? kStageNames[stage] : ""; }
int r3dg_profile_read(double* ms_out, int* count_out)
{
    return guarded([&]() -> int {
        std::lock_guard<std::mutex> lk(g_prof_mutex);
        for (int s = 0; s < ST_COUNT; s++) {
            double total = 0;
            for (auto& e : g_events[s]) {
                R3DG_HIP(hipEventSynchronize(e.b));
                float ms = 0;
                R3DG_HIP(hipEventElapsedTime(&ms, e.a, e.b));
                total += ms;
            }
            ms_out[s] = total;
            count_out[s] = (int)g_events[s].size();
        }
        return R3DG_OK;
    });
}

size_t r3dg_geometry_state_bytes(int P) { return GeometryLayout::make((size_t)(P > 0 ? P : 0)).bytes; }
size_t r3dg_image_state_bytes(int width, int height)
{
    const size_t T = (size_t)((width + 15) / 16) * ((height + 15) / 16);
    return ImageLayout::make((size_t)width * height, T).bytes;
}
size_t r3dg_binning_state_bytes(int64_t R) { return BinningLayout::make((size_t)(R > 0 ? R : 0)).bytes; }

int r3dg_geometry_state_offsets(int P, size_t* o)
{
    GeometryLayout L = GeometryLayout::make((size_t)P);
    o[0] = L.depths; o[1] = L.clamped; o[2] = L.radii; o[3] = L.means2D; o[4] = L.cov3D; o[5] = L.conic_opacity;
    o[6] = L.rgb; o[7] = L.tiles_touched; o[8] = L.point_offsets;
    return R3DG_OK;
}
size_t r3dg_geometry_state_total_offset(int P)
{
    return GeometryLayout::make((size_t)(P < 0 ? 0 : P)).total;
}
int r3dg_image_state_offsets(int width, int height, size_t* o)
{
    const size_t T = (size_t)((width + 15) / 16) * ((height + 15) / 16);
    ImageLayout L = ImageLayout::make((size_t)width * height, T);
    o[0] = L.final_T; o[1] = L.n_contrib; o[2] = L.ranges;
    return R3DG_OK;
}
int r3dg_binning_state_offsets(int64_t R, size_t* o)
{
    BinningLayout L = BinningLayout::make((size_t)R);
    o[0] = L.keys_unsorted; o[1] = L.keys; o[2] = L.vals_unsorted; o[3] = L.vals;
    return R3DG_OK;
}

// ---- forward, in two halves -----------------------------------------------------------------------------------------
// begin : validation, state allocation, preprocess (K2/K3), asynchronous read-back of num_rendered (event recorded)
// finish: waits for THAT event only (not for the stream), sizes the binning state, orders the instances, renders.
// Work the caller enqueues on the stream between the two halves (e.g. the shading kernels that produce the feature
// rows) keeps the GPU busy while the host waits for the count and enqueues the second half.
// `waiter` waits for everything queued on `signaller` so far.  Events come from a ring created once: creating and destroying one
// per call costs the host several microseconds, 8 such calls per iteration (651-656 it/s against 646 with per-call events).
// (hipEventReleaseToDevice on these events measured the same as the default system-scope release: 652-655 vs 656.)
static int event_flags() { return hipEventDisableTiming; }
static void stream_wait_stream(hipStream_t waiter, hipStream_t signaller)
{
    if (waiter == signaller) return;
    constexpr int RING = 64;
    static std::mutex mu;
    static std::map<int, std::vector<hipEvent_t>> rings;
    static std::map<int, int> next;
    int dev = 0;
    R3DG_HIP(hipGetDevice(&dev));
    hipEvent_t ev;
    {
        std::lock_guard<std::mutex> lk(mu);
        std::vector<hipEvent_t>& r = rings[dev];
        if (r.empty()) {
            r.resize(RING);
            for (auto& e : r) R3DG_HIP(hipEventCreateWithFlags(&e, event_flags()));
        }
        int& i = next[dev];
        ev = r[i];
        i = (i + 1) % RING;
    }
    R3DG_HIP(hipEventRecord(ev, signaller));
    R3DG_HIP(hipStreamWaitEvent(waiter, ev, 0));
}

struct ForwardTicket {
    hipStream_t stream;
    r3dg_alloc_fn binning_alloc;
    void* user;
    int P, S, D, M, width, height, compute_pseudo_normal, debug;
    const float *background, *means3D, *features, *colors_precomp, *viewmatrix;
    float tan_fovx, tan_fovy, cx, cy, focal_x, focal_y;
    float *out_color, *out_opacity, *out_depth, *out_feature, *out_normal, *out_surface_xyz, *out_weights;
    int32_t* radii_p;
    char *gbuf, *ibuf;
    hipEvent_t ready, ordered;           // two-phase forward: the count has arrived | bounded forward: the ordering is queued
    unsigned long long* host_total;      // pinned
    // bounded forward (r3dg_rasterize_forward_begin_bounded): the binning state is laid out for `capacity` instances, the
    // ordering was enqueued by _begin_ and `ready` marks its end; capacity < 0: the exact two-phase forward
    long long capacity;
    float* overflow_flag;
    unsigned int* overflow_count;
    char* bbuf;
    bool fused_front;                    // _begin_ allocated bbuf and launched the folded front end (forward_begin_impl)
};

static std::mutex g_ticket_mutex;
static std::vector<ForwardTicket*> g_ticket_pool;

static ForwardTicket* ticket_acquire()
{
    {
        std::lock_guard<std::mutex> lk(g_ticket_mutex);
        if (!g_ticket_pool.empty()) {
            ForwardTicket* t = g_ticket_pool.back();
            g_ticket_pool.pop_back();
            return t;
        }
    }
    ForwardTicket* t = new ForwardTicket();
    R3DG_HIP(hipEventCreateWithFlags(&t->ready, hipEventDisableTiming));     // (the host synchronises on this one: system scope)
    R3DG_HIP(hipEventCreateWithFlags(&t->ordered, event_flags()));
    R3DG_HIP(hipHostMalloc((void**)&t->host_total, sizeof(unsigned long long), hipHostMallocDefault));
    return t;
}
static void ticket_release(ForwardTicket* t)
{
    std::lock_guard<std::mutex> lk(g_ticket_mutex);
    g_ticket_pool.push_back(t);
}

static int enqueue_ordering(ForwardTicket* t, hipStream_t stream, int R);

static int forward_begin_impl(void* stream_, r3dg_alloc_fn geometry_alloc, r3dg_alloc_fn binning_alloc,
                           r3dg_alloc_fn image_alloc, void* user, int P, int S, int D, int M,
                           const float* background, int width, int height, const float* means3D, const float* shs,
                           const float* colors_precomp, const float* features, const float* opacities,
                           const float* scales, float scale_modifier, const float* rotations,
                           const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                           const float* cam_pos, float tan_fovx, float tan_fovy, float cx, float cy, int prefiltered,
                           int compute_pseudo_normal, float* out_color, float* out_opacity, float* out_depth,
                           float* out_feature, float* out_normal, float* out_surface_xyz, float* out_weights,
                           int32_t* radii, int debug_, void** ticket_out, long long capacity,
                           float* overflow_flag, unsigned int* overflow_count, void* ordering_stream_)
{
    (void)prefiltered;
    if (!ticket_out) return invalid("rasterize_forward_begin: null ticket pointer");
    *ticket_out = nullptr;
    if (P < 0 || width <= 0 || height <= 0) return invalid("rasterize_forward: bad P/width/height");
    if (S < 0 || S > R3DG_MAX_S_FWD) return invalid("rasterize_forward: feature channels S must be in [0,36]");
    if (!geometry_alloc || !binning_alloc || !image_alloc) return invalid("rasterize_forward: null resize callback");
    if (shs == nullptr && colors_precomp == nullptr)
        return invalid("rasterize_forward: provide SHs or precomputed colours");
    if (shs != nullptr && colors_precomp == nullptr && (M < (D + 1) * (D + 1) || D > 3 || D < 0))
        return invalid("rasterize_forward: SH degree/coefficients mismatch");
    if (cov3D_precomp == nullptr && (scales == nullptr || rotations == nullptr))
        return invalid("rasterize_forward: provide scales+rotations or a precomputed 3D covariance");
    if (P == 0) return R3DG_OK;                  // no ticket: nothing was launched (finish accepts NULL)

    return guarded([&]() -> int {
        hipStream_t stream = (hipStream_t)stream_;
        const bool debug = debug_ != 0;
        const float focal_y = height / (2.0f * tan_fovy);
        const float focal_x = width / (2.0f * tan_fovx);
        const int gx = (width + R3DG_TILE_X - 1) / R3DG_TILE_X, gy = (height + R3DG_TILE_Y - 1) / R3DG_TILE_Y;
        const size_t T = (size_t)gx * gy, N = (size_t)width * height;

        GeometryLayout G = GeometryLayout::make((size_t)P);
        char* gbuf = (char*)geometry_alloc(user, G.bytes);
        ImageLayout I = ImageLayout::make(N, T);
        char* ibuf = (char*)image_alloc(user, I.bytes);
        if (!gbuf || !ibuf) { set_error("rasterize_forward: resize callback returned NULL"); return R3DG_EALLOC; }

        int* radii_p = radii ? radii : (int*)(gbuf + G.radii);
        float* g_depths = (float*)(gbuf + G.depths);
        float* g_means2D = (float*)(gbuf + G.means2D);
        float* g_conic = (float*)(gbuf + G.conic_opacity);
        float* g_rgb = (float*)(gbuf + G.rgb);
        uint32_t* g_tiles = (uint32_t*)(gbuf + G.tiles_touched);
        uint32_t* g_block = (uint32_t*)(gbuf + G.block_sums);
        unsigned long long* g_total = (unsigned long long*)(gbuf + G.total);

        ForwardTicket* t = ticket_acquire();
        struct Release { ForwardTicket* t; ~Release() { if (t) ticket_release(t); } } on_error{t};      // (a launch check may throw)
        // bounded with an ordering stream: the projection goes there too, behind everything the caller has queued on `stream`
        // so far -- the whole front end of the rasterizer then runs beside what the caller queues on `stream` next, and
        // r3dg_rasterize_forward_finish_bounded joins it
        const hipStream_t order_stream = capacity >= 0 && ordering_stream_ ? (hipStream_t)ordering_stream_ : stream;
        if (order_stream != stream) {
            stream_wait_stream(order_stream, stream);
        }
        // bounded + direct binning: the front end is one chain whose launches do not depend on the count, so three of them fold
        // into their neighbours (launch_tile_binning `fused`): the projection zeroes the tile counters, the tile scan also scans
        // the projection's block sums, an extra block of the emit kernel orders the tiles
        t->bbuf = nullptr;
        t->fused_front = false;
        uint32_t* zero_words = nullptr;
        int zero_n = 0;
        if (capacity >= 0 && opt(R3DG_OPT_TILE_BINNING) == 2 && (int)T <= tile_binning_max_tiles()) {
            BinningLayout B = BinningLayout::make((size_t)capacity);
            t->bbuf = (char*)binning_alloc(user, B.bytes);
            if (!t->bbuf) { set_error("rasterize_forward: binning resize callback returned NULL"); return R3DG_EALLOC; }
            t->fused_front = true;
            zero_words = (uint32_t*)(t->bbuf + B.sort_temp);
            zero_n = (int)T;
        }
        StageTimer t_pre(order_stream, ST_PREPROCESS);
        launch_preprocess(order_stream, P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs,
                          (uint8_t*)(gbuf + G.clamped), cov3D_precomp, colors_precomp, viewmatrix, projmatrix, cam_pos,
                          width, height, tan_fovx, tan_fovy, focal_x, focal_y, radii_p, g_means2D, g_depths,
                          (float*)(gbuf + G.cov3D), g_rgb, g_conic, (float*)(gbuf + G.splat), gx, gy, g_tiles, g_block, g_total,
                          !t->fused_front, zero_words, zero_n);
        check_launch(order_stream, debug, "preprocess");
        t_pre.stop();

        t->stream = stream; t->binning_alloc = binning_alloc; t->user = user;
        t->P = P; t->S = S; t->D = D; t->M = M; t->width = width; t->height = height;
        t->compute_pseudo_normal = compute_pseudo_normal; t->debug = debug_;
        t->background = background; t->means3D = means3D; t->features = features; t->colors_precomp = colors_precomp;
        t->viewmatrix = viewmatrix;
        t->tan_fovx = tan_fovx; t->tan_fovy = tan_fovy; t->cx = cx; t->cy = cy; t->focal_x = focal_x; t->focal_y = focal_y;
        t->out_color = out_color; t->out_opacity = out_opacity; t->out_depth = out_depth; t->out_feature = out_feature;
        t->out_normal = out_normal; t->out_surface_xyz = out_surface_xyz; t->out_weights = out_weights;
        t->radii_p = radii_p; t->gbuf = gbuf; t->ibuf = ibuf;
        t->capacity = capacity; t->overflow_flag = overflow_flag; t->overflow_count = overflow_count;
        if (capacity >= 0) {
            // bounded: nobody reads the count; the ordering follows the projection right away
            const int st_order = enqueue_ordering(t, order_stream, (int)capacity);
            if (st_order != R3DG_OK) return st_order;
            R3DG_HIP(hipEventRecord(t->ordered, order_stream));
            on_error.t = nullptr;
            *ticket_out = t;
            return R3DG_OK;
        }
        // the one device->host read-back of the forward (reference rasterizer_impl.cu:291), asynchronous here
        R3DG_HIP(hipMemcpyAsync(t->host_total, g_total, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
        R3DG_HIP(hipEventRecord(t->ready, stream));
        on_error.t = nullptr;
        *ticket_out = t;
        return R3DG_OK;
    });
}

#define R3DG_FORWARD_ARGS                                                                                              \
    stream_, geometry_alloc, binning_alloc, image_alloc, user, P, S, D, M, background, width, height, means3D, shs,    \
        colors_precomp, features, opacities, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, \
        cam_pos, tan_fovx, tan_fovy, cx, cy, prefiltered, compute_pseudo_normal, out_color, out_opacity, out_depth,    \
        out_feature, out_normal, out_surface_xyz, out_weights, radii, debug_, ticket_out

int r3dg_rasterize_forward_begin(void* stream_, r3dg_alloc_fn geometry_alloc, r3dg_alloc_fn binning_alloc,
                           r3dg_alloc_fn image_alloc, void* user, int P, int S, int D, int M,
                           const float* background, int width, int height, const float* means3D, const float* shs,
                           const float* colors_precomp, const float* features, const float* opacities,
                           const float* scales, float scale_modifier, const float* rotations,
                           const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                           const float* cam_pos, float tan_fovx, float tan_fovy, float cx, float cy, int prefiltered,
                           int compute_pseudo_normal, float* out_color, float* out_opacity, float* out_depth,
                           float* out_feature, float* out_normal, float* out_surface_xyz, float* out_weights,
                           int32_t* radii, int debug_, void** ticket_out)
{
    return forward_begin_impl(R3DG_FORWARD_ARGS, -1, nullptr, nullptr, nullptr);
}

int r3dg_rasterize_forward_begin_bounded(void* stream_, r3dg_alloc_fn geometry_alloc, r3dg_alloc_fn binning_alloc,
                           r3dg_alloc_fn image_alloc, void* user, int P, int S, int D, int M,
                           const float* background, int width, int height, const float* means3D, const float* shs,
                           const float* colors_precomp, const float* features, const float* opacities,
                           const float* scales, float scale_modifier, const float* rotations,
                           const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                           const float* cam_pos, float tan_fovx, float tan_fovy, float cx, float cy, int prefiltered,
                           int compute_pseudo_normal, float* out_color, float* out_opacity, float* out_depth,
                           float* out_feature, float* out_normal, float* out_surface_xyz, float* out_weights,
                           int32_t* radii, int debug_, void* ordering_stream, long long capacity,
                           float* overflow_flag, unsigned int* overflow_count, void** ticket_out)
{
    if (capacity < 0 || capacity > 0x7fffffffll) return invalid("rasterize_forward (bounded): capacity must be in [0, 2^31)");
    return forward_begin_impl(R3DG_FORWARD_ARGS, capacity, overflow_flag, overflow_count, ordering_stream);
}
#undef R3DG_FORWARD_ARGS

int r3dg_rasterize_forward_finish_bounded(void* ticket_, void* main_stream_)
{
    if (!ticket_) return R3DG_OK;                // P == 0
    ForwardTicket* t = (ForwardTicket*)ticket_;
    if (t->capacity < 0) return invalid("rasterize_forward_finish_bounded: not a bounded ticket");
    const int st = guarded([&]() -> int {
        const hipStream_t stream = (hipStream_t)main_stream_;
        const bool debug = t->debug != 0;
        const int P = t->P, S = t->S, width = t->width, height = t->height;
        const int gx = (width + R3DG_TILE_X - 1) / R3DG_TILE_X, gy = (height + R3DG_TILE_Y - 1) / R3DG_TILE_Y;
        const size_t T = (size_t)gx * gy, N = (size_t)width * height;
        GeometryLayout G = GeometryLayout::make((size_t)P);
        ImageLayout I = ImageLayout::make(N, T);
        BinningLayout B = BinningLayout::make((size_t)t->capacity);
        // join: the tile kernel needs the ordering (begin's stream) AND whatever the caller queued on `stream` (feature rows)
        R3DG_HIP(hipStreamWaitEvent(stream, t->ordered, 0));
        StageTimer t_rf(stream, ST_RENDER_FWD);
        launch_render_forward(stream, width, height, S, opt(R3DG_OPT_TILE_ORDER) ? (uint32_t*)(t->ibuf + I.tile_order) : nullptr,
                              (uint32_t*)(t->ibuf + I.ranges), (uint32_t*)(t->bbuf + B.vals),
                              (const float*)(t->gbuf + G.splat), t->features, (float*)(t->ibuf + I.final_T),
                              (uint32_t*)(t->ibuf + I.n_contrib), t->background, t->out_color, t->out_opacity,
                              t->out_depth, t->out_feature, t->out_weights);
        check_launch(stream, debug, "render_forward");
        t_rf.stop();
        if (t->compute_pseudo_normal) {
            StageTimer t_n(stream, ST_NORMAL);
            launch_pseudo_normal(stream, width, height, t->viewmatrix, t->focal_x, t->focal_y, t->cx, t->cy, t->out_opacity,
                                 t->out_depth, t->out_normal, t->out_surface_xyz, debug);
            t_n.stop();
        }
        return R3DG_OK;
    });
    ticket_release(t);
    return st;
}

int r3dg_rasterize_forward_finish(void* ticket_, int* num_rendered_out)
{
    return r3dg_rasterize_forward_finish_on(ticket_, nullptr, num_rendered_out);
}

// Instance ordering (K5-K7) of a forward whose projection has run: binning state for R instance slots, tile ranges, the
// depth-sorted per-tile lists.  Bounded tickets (capacity >= 0) need the direct tile binning -- the only formulation whose
// launches do not depend on the count.
static int enqueue_ordering(ForwardTicket* t, hipStream_t stream, int R)
{
    const bool debug = t->debug != 0;
    const int P = t->P, width = t->width, height = t->height;
    const int gx = (width + R3DG_TILE_X - 1) / R3DG_TILE_X, gy = (height + R3DG_TILE_Y - 1) / R3DG_TILE_Y;
    const size_t T = (size_t)gx * gy, N = (size_t)width * height;
    GeometryLayout G = GeometryLayout::make((size_t)P);
    ImageLayout I = ImageLayout::make(N, T);
    char* gbuf = t->gbuf;
    char* ibuf = t->ibuf;
    int* radii_p = t->radii_p;
    float* g_depths = (float*)(gbuf + G.depths);
    float* g_means2D = (float*)(gbuf + G.means2D);
    uint32_t* g_tiles = (uint32_t*)(gbuf + G.tiles_touched);
    uint32_t* g_block = (uint32_t*)(gbuf + G.block_sums);
    r3dg_alloc_fn binning_alloc = t->binning_alloc;
    void* user = t->user;
    if (t->capacity >= 0 && !(opt(R3DG_OPT_TILE_BINNING) == 2 && (int)T <= tile_binning_max_tiles())) {
        set_error("rasterize_forward (bounded): needs the direct tile binning and at most 16384 tiles: ask r3dg_bounded_forward_supported(width, height) first");
        return R3DG_EINVAL;
    }
        BinningLayout B = BinningLayout::make((size_t)R);
        char* bbuf = t->fused_front ? t->bbuf : (char*)binning_alloc(user, B.bytes);     // (fused: allocated by _begin)
        t->bbuf = bbuf;
        if (!bbuf) { set_error("rasterize_forward: binning resize callback returned NULL"); return R3DG_EALLOC; }
        uint64_t* keys_u = (uint64_t*)(bbuf + B.keys_unsorted);
        uint64_t* keys = (uint64_t*)(bbuf + B.keys);
        uint32_t* vals_u = (uint32_t*)(bbuf + B.vals_unsorted);
        uint32_t* vals = (uint32_t*)(bbuf + B.vals);

        uint32_t* ranges = (uint32_t*)(ibuf + I.ranges);
        uint32_t* tile_order = opt(R3DG_OPT_TILE_ORDER) ? (uint32_t*)(ibuf + I.tile_order) : nullptr;
        if (opt(R3DG_OPT_TILE_BINNING) == 2 && (int)T <= tile_binning_max_tiles()) {
            // direct binning (rasterizer_preprocess.hip): count -> scan (= the tile ranges) -> emit into the segments, then
            // the per-tile sort by (depth, index): same final lists as the global stable sort
            uint32_t* big_list = (uint32_t*)(ibuf + I.big_list);
            uint32_t* big_count = (uint32_t*)(ibuf + I.big_count);
            uint32_t* tile_counts = (uint32_t*)(bbuf + B.sort_temp);
            StageTimer t_dup(stream, ST_DUPKEYS);
            uint32_t* order = tile_order ? tile_order : (uint32_t*)(ibuf + I.tile_order);
            launch_tile_binning(stream, P, (int)T, g_means2D, g_depths, radii_p, g_tiles, g_block, gx, gy, tile_counts,
                                tile_counts + T, ranges, (uint32_t*)(gbuf + G.point_offsets), keys_u,
                                (unsigned long long*)(gbuf + G.total), t->capacity, t->overflow_flag,
                                t->overflow_count, t->fused_front, order, tile_sort_small_cap(), big_list, big_count);
            check_launch(stream, debug, "tile_binning");
            t_dup.stop();
            StageTimer t_sort(stream, ST_SORT);
            if (!t->fused_front) {
                launch_tile_order(stream, (int)T, ranges, order, tile_sort_small_cap(), big_list, big_count);
                check_launch(stream, debug, "tile_order");
            }
            launch_tile_sort(stream, (int)T, order, ranges, big_list, big_count, keys, vals, keys_u, true);
            check_launch(stream, debug, "tile_sort");
            t_sort.stop();
        } else if (opt(R3DG_OPT_TILE_BINNING)) {
            // stable partition by tile id (one radix pass over the tile bits), then a per-tile depth sort in LDS: same
            // final order as the global 44-bit sort (radix_sort.hip)
            uint32_t* big_list = (uint32_t*)(ibuf + I.big_list);
            uint32_t* big_count = (uint32_t*)(ibuf + I.big_count);
            StageTimer t_dup(stream, ST_DUPKEYS);
            launch_duplicate_with_keys(stream, P, g_means2D, g_depths, g_tiles, g_block,
                                       (uint32_t*)(gbuf + G.point_offsets), keys_u, vals_u, radii_p, gx, gy);
            check_launch(stream, debug, "duplicate_with_keys");
            t_dup.stop();
            const int bit = (int)higher_msb((uint32_t)T);
            StageTimer t_sort(stream, ST_SORT);
            sort_pairs_range(stream, (size_t)R, keys_u, vals_u, keys, vals, 32, 32 + bit, bbuf + B.sort_temp, debug,
                             /*stable=*/false);
            R3DG_HIP(hipMemsetAsync(ranges, 0, T * 8, stream));
            launch_identify_tile_ranges(stream, R, keys, ranges);
            check_launch(stream, debug, "identify_tile_ranges");
            uint32_t* order = tile_order ? tile_order : (uint32_t*)(ibuf + I.tile_order);
            launch_tile_order(stream, (int)T, ranges, order, tile_sort_small_cap(), big_list, big_count);
            check_launch(stream, debug, "tile_order");
            launch_tile_sort(stream, (int)T, order, ranges, big_list, big_count, keys, vals, keys_u, false);
            check_launch(stream, debug, "tile_sort");
            t_sort.stop();
        } else {
            StageTimer t_dup(stream, ST_DUPKEYS);
            launch_duplicate_with_keys(stream, P, g_means2D, g_depths, g_tiles, g_block,
                                       (uint32_t*)(gbuf + G.point_offsets), keys_u, vals_u, radii_p, gx, gy);
            check_launch(stream, debug, "duplicate_with_keys");
            t_dup.stop();

            const int bit = (int)higher_msb((uint32_t)T);
            StageTimer t_sort(stream, ST_SORT);
            sort_pairs(stream, (size_t)R, keys_u, vals_u, keys, vals, 32 + bit, bbuf + B.sort_temp, debug);
            t_sort.stop();

            StageTimer t_rng(stream, ST_RANGES);
            R3DG_HIP(hipMemsetAsync(ranges, 0, T * 8, stream));
            launch_identify_tile_ranges(stream, R, keys, ranges);
            check_launch(stream, debug, "identify_tile_ranges");
            t_rng.stop();
            if (tile_order) {
                launch_tile_order(stream, (int)T, ranges, tile_order, 0u, nullptr, nullptr);
                check_launch(stream, debug, "tile_order");
            }
        }
    return R3DG_OK;
}

int r3dg_rasterize_forward_finish_on(void* ticket_, void* ordering_stream_, int* num_rendered_out)
{
    if (num_rendered_out) *num_rendered_out = 0;
    if (!ticket_) return R3DG_OK;                // P == 0
    ForwardTicket* t = (ForwardTicket*)ticket_;
    if (t->capacity >= 0) return invalid("rasterize_forward_finish: bounded ticket (use r3dg_rasterize_forward_finish_bounded)");
    const int st = guarded([&]() -> int {
        const hipStream_t main_stream = t->stream;
        // instance ordering (K5-K7) may run on its own stream: it depends on the projection only, so it can overlap the
        // kernels the caller queued on the main stream after _begin (the shading that produces the feature rows)
        const hipStream_t order_stream = ordering_stream_ ? (hipStream_t)ordering_stream_ : main_stream;
        hipStream_t stream = order_stream;
        const bool debug = t->debug != 0;
        const int P = t->P, S = t->S, width = t->width, height = t->height;
        const int gx = (width + R3DG_TILE_X - 1) / R3DG_TILE_X, gy = (height + R3DG_TILE_Y - 1) / R3DG_TILE_Y;
        const size_t T = (size_t)gx * gy, N = (size_t)width * height;
        GeometryLayout G = GeometryLayout::make((size_t)P);
        ImageLayout I = ImageLayout::make(N, T);
        char* gbuf = t->gbuf;
        char* ibuf = t->ibuf;
        const float* g_splat = (const float*)(gbuf + G.splat);
        const float* features = t->features;
        const float* background = t->background;
        const float* viewmatrix = t->viewmatrix;
        const float focal_x = t->focal_x, focal_y = t->focal_y, cx = t->cx, cy = t->cy;
        float *out_color = t->out_color, *out_opacity = t->out_opacity, *out_depth = t->out_depth,
              *out_feature = t->out_feature, *out_normal = t->out_normal, *out_surface_xyz = t->out_surface_xyz,
              *out_weights = t->out_weights;
        const int compute_pseudo_normal = t->compute_pseudo_normal;

        R3DG_HIP(hipEventSynchronize(t->ready));
        const unsigned long long total = *t->host_total;
        if (order_stream != main_stream) R3DG_HIP(hipStreamWaitEvent(order_stream, t->ready, 0));
        if (total > 0x7fffffffull) { set_error("rasterize_forward: num_rendered exceeds 2^31-1"); return R3DG_EINVAL; }
        const int R = (int)total;

        {
            const int st_order = enqueue_ordering(t, stream, R);
            if (st_order != R3DG_OK) return st_order;
        }
        BinningLayout B = BinningLayout::make((size_t)R);
        uint32_t* vals = (uint32_t*)(t->bbuf + B.vals);
        uint32_t* ranges = (uint32_t*)(ibuf + I.ranges);
        uint32_t* tile_order = opt(R3DG_OPT_TILE_ORDER) ? (uint32_t*)(ibuf + I.tile_order) : nullptr;
        if (order_stream != main_stream) {          // join: the tile kernel needs the ordering AND the feature rows
            stream_wait_stream(main_stream, order_stream);
        }
        stream = main_stream;
        StageTimer t_rf(stream, ST_RENDER_FWD);
        launch_render_forward(stream, width, height, S, tile_order, ranges, vals, g_splat, features,
                              (float*)(ibuf + I.final_T), (uint32_t*)(ibuf + I.n_contrib), background,
                              out_color, out_opacity, out_depth, out_feature, out_weights);
        check_launch(stream, debug, "render_forward");
        t_rf.stop();

        if (compute_pseudo_normal) {
            StageTimer t_n(stream, ST_NORMAL);
            launch_pseudo_normal(stream, width, height, viewmatrix, focal_x, focal_y, cx, cy, out_opacity, out_depth,
                                 out_normal, out_surface_xyz, debug);
            t_n.stop();
        }
        if (num_rendered_out) *num_rendered_out = R;
        return R3DG_OK;
    });
    ticket_release(t);
    return st;
}

int r3dg_rasterize_forward(void* stream_, r3dg_alloc_fn geometry_alloc, r3dg_alloc_fn binning_alloc,
                           r3dg_alloc_fn image_alloc, void* user, int P, int S, int D, int M,
                           const float* background, int width, int height, const float* means3D, const float* shs,
                           const float* colors_precomp, const float* features, const float* opacities,
                           const float* scales, float scale_modifier, const float* rotations,
                           const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                           const float* cam_pos, float tan_fovx, float tan_fovy, float cx, float cy, int prefiltered,
                           int compute_pseudo_normal, float* out_color, float* out_opacity, float* out_depth,
                           float* out_feature, float* out_normal, float* out_surface_xyz, float* out_weights,
                           int32_t* radii, int debug_, int* num_rendered_out)
{
    if (num_rendered_out) *num_rendered_out = 0;
    void* ticket = nullptr;
    const int st = r3dg_rasterize_forward_begin(stream_, geometry_alloc, binning_alloc, image_alloc, user, P, S, D, M,
                                                background, width, height, means3D, shs, colors_precomp, features,
                                                opacities, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix,
                                                projmatrix, cam_pos, tan_fovx, tan_fovy, cx, cy, prefiltered,
                                                compute_pseudo_normal, out_color, out_opacity, out_depth, out_feature,
                                                out_normal, out_surface_xyz, out_weights, radii, debug_, &ticket);
    if (st != R3DG_OK) return st;
    return r3dg_rasterize_forward_finish(ticket, num_rendered_out);
}

// (defined below)
int r3dg_rasterize_backward(void* stream_, int P, int S, int D, int M, int R, const float* background, int width,
                            int height, const float* means3D, const float* shs, const float* features,
                            const float* colors_precomp, const float* scales, float scale_modifier,
                            const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                            const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy,
                            const int32_t* radii, const void* geom_buffer, const void* binning_buffer,
                            const void* img_buffer, const float* dL_dpix, const float* dL_dpix_o,
                            const float* dL_dpix_d, const float* dL_dpix_f, float* dL_dmean2D, float* dL_dconic,
                            float* dL_dopacity, float* dL_dcolor, float* dL_dfeature, float* dL_dmean3D,
                            float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                            int backward_geometry, int debug_)
{
    return r3dg_rasterize_backward_split(stream_, stream_, P, S, D, M, R, background, width, height, means3D, shs,
                                         features, colors_precomp, scales, scale_modifier, rotations, cov3D_precomp,
                                         viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, geom_buffer,
                                         binning_buffer, img_buffer, dL_dpix, dL_dpix_o, dL_dpix_d, dL_dpix_f, dL_dmean2D,
                                         dL_dconic, dL_dopacity, dL_dcolor, dL_dfeature, dL_dmean3D, dL_dcov3D, dL_dsh,
                                         dL_dscale, dL_drot, backward_geometry, debug_, -1, nullptr);
}

int r3dg_rasterize_backward_split(void* stream_, void* geometry_stream_, int P, int S, int D, int M, int R,
                                  const float* background, int width, int height, const float* means3D,
                                  const float* shs, const float* features, const float* colors_precomp,
                                  const float* scales, float scale_modifier, const float* rotations,
                                  const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                                  const float* campos, float tan_fovx, float tan_fovy, const int32_t* radii,
                                  const void* geom_buffer, const void* binning_buffer, const void* img_buffer,
                                  const float* dL_dpix, const float* dL_dpix_o, const float* dL_dpix_d,
                                  const float* dL_dpix_f, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                                  float* dL_dcolor, float* dL_dfeature, float* dL_dmean3D, float* dL_dcov3D,
                                  float* dL_dsh, float* dL_dscale, float* dL_drot, int backward_geometry, int debug_,
                                  int n_active_features, const int* active_features)
{
    if (P < 0 || width <= 0 || height <= 0 || R < 0) return invalid("rasterize_backward: bad P/R/width/height");
    if (S < 0 || S > R3DG_MAX_S_BWD) return invalid("rasterize_backward: feature channels S must be in [0,36]");
    if (P == 0) return R3DG_OK;
    if (!geom_buffer || !img_buffer || (R > 0 && !binning_buffer)) return invalid("rasterize_backward: null state buffer");
    if (!dL_dpix || !dL_dpix_o || (S > 0 && !dL_dpix_f)) return invalid("rasterize_backward: null upstream gradient");
    if (n_active_features >= 0) {
        if (n_active_features > S || !active_features) return invalid("rasterize_backward: bad active feature list");
        for (int i = 0; i < n_active_features; i++)
            if (active_features[i] < 0 || active_features[i] >= S)
                return invalid("rasterize_backward: active feature index out of range");
    }

    return guarded([&]() -> int {
        hipStream_t stream = (hipStream_t)stream_;
        const bool debug = debug_ != 0;
        const float focal_y = height / (2.0f * tan_fovy);
        const float focal_x = width / (2.0f * tan_fovx);
        const int gx = (width + R3DG_TILE_X - 1) / R3DG_TILE_X, gy = (height + R3DG_TILE_Y - 1) / R3DG_TILE_Y;
        const size_t T = (size_t)gx * gy, N = (size_t)width * height;
        GeometryLayout G = GeometryLayout::make((size_t)P);
        ImageLayout I = ImageLayout::make(N, T);
        BinningLayout B = BinningLayout::make((size_t)R);
        const char* gbuf = (const char*)geom_buffer;
        const char* ibuf = (const char*)img_buffer;
        const char* bbuf = (const char*)binning_buffer;
        const int* radii_p = radii ? radii : (const int*)(gbuf + G.radii);

        if (R > 0) {
            StageTimer t_rb(stream, ST_RENDER_BWD);
            launch_render_backward(stream, P, width, height, S, n_active_features, active_features,
                                   opt(R3DG_OPT_TILE_ORDER) ? (const uint32_t*)(ibuf + I.tile_order) : nullptr,
                                   (const uint32_t*)(ibuf + I.ranges),
                                   (const uint32_t*)(bbuf + B.vals), background, (const float*)(gbuf + G.splat),
                                   features, (const float*)(ibuf + I.final_T), (const uint32_t*)(ibuf + I.n_contrib),
                                   dL_dpix, dL_dpix_o, dL_dpix_d, dL_dpix_f, dL_dmean2D, dL_dconic, dL_dopacity,
                                   dL_dcolor, dL_dfeature, backward_geometry);
            check_launch(stream, debug, "render_backward");
            t_rb.stop();
        } else {
            // nothing was rendered: the five per-Gaussian outputs the tile pass writes are zero
            R3DG_HIP(hipMemsetAsync(dL_dmean2D, 0, (size_t)P * 3 * sizeof(float), stream));
            R3DG_HIP(hipMemsetAsync(dL_dconic, 0, (size_t)P * 4 * sizeof(float), stream));
            R3DG_HIP(hipMemsetAsync(dL_dopacity, 0, (size_t)P * sizeof(float), stream));
            R3DG_HIP(hipMemsetAsync(dL_dcolor, 0, (size_t)P * 3 * sizeof(float), stream));
            if (S > 0 && dL_dfeature != nullptr) R3DG_HIP(hipMemsetAsync(dL_dfeature, 0, (size_t)P * S * sizeof(float), stream));
        }
        const float* cov3D_ptr = cov3D_precomp != nullptr ? cov3D_precomp : (const float*)(gbuf + G.cov3D);
        // the per-Gaussian geometry backward may run on a second stream, ordered after the tile kernel by an event
        hipStream_t gstream = (hipStream_t)geometry_stream_;
        if (gstream != stream) {
            stream_wait_stream(gstream, stream);
        }
        stream = gstream;
        StageTimer t_pb(stream, ST_PREPROCESS_BWD);
        launch_preprocess_backward(stream, P, D, M, means3D, radii_p, colors_precomp == nullptr ? shs : nullptr,
                                   (const uint8_t*)(gbuf + G.clamped), cov3D_precomp == nullptr ? scales : nullptr,
                                   rotations, scale_modifier, cov3D_ptr, viewmatrix, projmatrix, focal_x, focal_y,
                                   tan_fovx, tan_fovy, campos, dL_dmean2D, dL_dconic, (const float*)(gbuf + G.conic_opacity),
                                   width, height, dL_dmean3D, dL_dcolor, dL_dcov3D, dL_dsh, dL_dscale, dL_drot);
        check_launch(stream, debug, "preprocess_backward");
        t_pb.stop();
        return R3DG_OK;
    });
}

int r3dg_rasterize_backward_features(void* stream_, int P, int S, int R, int width, int height, const void* geom_buffer,
                                     const void* binning_buffer, const void* img_buffer, const float* dL_dpix_f,
                                     float* dL_dfeature, int n_active_features, const int* active_features, int debug_)
{
    if (P < 0 || width <= 0 || height <= 0 || R < 0) return invalid("rasterize_backward_features: bad P/R/width/height");
    if (S <= 0 || S > R3DG_MAX_S_BWD) return invalid("rasterize_backward_features: feature channels S must be in [1,36]");
    if (P == 0 || R == 0) return R3DG_OK;
    if (!geom_buffer || !img_buffer || !binning_buffer || !dL_dpix_f || !dL_dfeature)
        return invalid("rasterize_backward_features: null buffer");
    if (n_active_features >= 0) {
        if (n_active_features > S || !active_features) return invalid("rasterize_backward_features: bad active feature list");
        for (int i = 0; i < n_active_features; i++)
            if (active_features[i] < 0 || active_features[i] >= S)
                return invalid("rasterize_backward_features: active feature index out of range");
    }
    return guarded([&]() -> int {
        hipStream_t stream = (hipStream_t)stream_;
        const int gx = (width + R3DG_TILE_X - 1) / R3DG_TILE_X, gy = (height + R3DG_TILE_Y - 1) / R3DG_TILE_Y;
        const size_t T = (size_t)gx * gy, N = (size_t)width * height;
        GeometryLayout G = GeometryLayout::make((size_t)P);
        ImageLayout I = ImageLayout::make(N, T);
        BinningLayout B = BinningLayout::make((size_t)R);
        const char* gbuf = (const char*)geom_buffer;
        const char* ibuf = (const char*)img_buffer;
        const char* bbuf = (const char*)binning_buffer;
        StageTimer t_rb(stream, ST_RENDER_BWD);
        launch_render_backward_features(stream, width, height, S, n_active_features, active_features,
                                        opt(R3DG_OPT_TILE_ORDER) ? (const uint32_t*)(ibuf + I.tile_order) : nullptr,
                                        (const uint32_t*)(ibuf + I.ranges), (const uint32_t*)(bbuf + B.vals),
                                        (const float*)(gbuf + G.splat), (const float*)(ibuf + I.final_T),
                                        (const uint32_t*)(ibuf + I.n_contrib), dL_dpix_f, dL_dfeature);
        check_launch(stream, debug_ != 0, "render_backward_features");
        t_rb.stop();
        return R3DG_OK;
    });
}

int r3dg_mark_visible(void* stream_, int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                      uint8_t* present)
{
    (void)projmatrix;
    if (P < 0) return invalid("mark_visible: bad P");
    if (P == 0) return R3DG_OK;
    return guarded([&]() -> int {
        launch_mark_visible((hipStream_t)stream_, P, means3D, viewmatrix, present);
        check_launch((hipStream_t)stream_, false, "mark_visible");
        return R3DG_OK;
    });
}

int r3dg_shade_forward_cached(void* stream_, int P, int K, int M, const float* base_color, const float* roughness,
                              const float* normals, const float* viewdirs, const float* incidents, const float* env,
                              int He, int We, const float* env_transform, const float* visibility,
                              const float* incident_dirs, const float* incident_areas, float uniform_area,
                              const uint32_t* taps, int flags, float* out)
{
    if (P < 0 || K <= 0 || He <= 0 || We <= 0) return invalid("shade_forward: bad P/K/env size");
    if (He > 32767 || We > 32767) return invalid("shade_forward: environment map larger than 32767 texels per side");
    if (M != 1 && M != 4 && M != 9 && M != 16) return invalid("shade_forward: incidents must hold 1, 4, 9 or 16 SH coefficients");
    if (P == 0) return R3DG_OK;
    if (!base_color || !roughness || !normals || !viewdirs || !incidents || !env || !visibility || !incident_dirs || !out)
        return invalid("shade_forward: null buffer");
    return guarded([&]() -> int {
        hipStream_t stream = (hipStream_t)stream_;
        StageTimer t(stream, ST_SHADE_FWD);
        launch_shade_forward(stream, P, K, M, base_color, roughness, normals, viewdirs, incidents, env, He, We,
                             env_transform, visibility, incident_dirs, incident_areas, out, taps,
                             (flags & R3DG_SHADE_TRAIN_OUTPUTS) != 0, uniform_area, (flags & R3DG_SHADE_TAPS_ARE_RADIANCE) != 0,
                             (flags & R3DG_SHADE_LEAVE_ROOM) != 0);
        check_launch(stream, false, "shade_forward");
        t.stop();
        return R3DG_OK;
    });
}

int r3dg_shade_forward(void* stream_, int P, int K, int M, const float* base_color, const float* roughness,
                       const float* normals, const float* viewdirs, const float* incidents, const float* env, int He,
                       int We, const float* env_transform, const float* visibility, const float* incident_dirs,
                       const float* incident_areas, float* out)
{
    return r3dg_shade_forward_cached(stream_, P, K, M, base_color, roughness, normals, viewdirs, incidents, env, He, We,
                                     env_transform, visibility, incident_dirs, incident_areas, 0.f, nullptr, 0, out);
}

int r3dg_shade_build_taps(void* stream_, int64_t num_samples, const float* incident_dirs, const float* env_transform,
                          int He, int We, const float* env_radiance, uint32_t* taps)
{
    if (num_samples < 0 || He <= 0 || We <= 0 || He > 32767 || We > 32767) return invalid("shade_build_taps: bad sizes");
    if (num_samples == 0) return R3DG_OK;
    if (!incident_dirs || !taps) return invalid("shade_build_taps: null buffer");
    return guarded([&]() -> int {
        launch_shade_build_taps((hipStream_t)stream_, (size_t)num_samples, incident_dirs, env_transform, He, We,
                                env_radiance, taps);
        return R3DG_OK;
    });
}

int r3dg_shade_build_transport(void* stream_, int P, int K, int M, const float* normals, const float* incidents,
                               const float* visibility, const float* incident_dirs, const float* incident_areas,
                               float uniform_area, float* radiance_inout, float* consts)
{
    if (P < 0 || K <= 0) return invalid("shade_build_transport: bad P/K");
    if (M != 1 && M != 4 && M != 9 && M != 16) return invalid("shade_build_transport: incidents must hold 1, 4, 9 or 16 SH coefficients");
    if (P == 0) return R3DG_OK;
    if (!normals || !incidents || !visibility || !incident_dirs || !radiance_inout || !consts)
        return invalid("shade_build_transport: null buffer");
    return guarded([&]() -> int {
        launch_shade_build_transport((hipStream_t)stream_, P, K, M, normals, incidents, visibility, incident_dirs,
                                     incident_areas, uniform_area, radiance_inout, consts);
        return R3DG_OK;
    });
}

int r3dg_shade_build_split(void* stream_, int P, int K, const int32_t* perm, const float* normals, const float* incidents,
                           const float* visibility, const float* incident_dirs, const float* zsamples, float uniform_area,
                           float* lt, float* vis_t, float* consts)
{
    if (P < 0 || K <= 0 || (K % 4) != 0) return invalid("shade_build_split: bad P/K (K must be a multiple of 4)");
    if (P == 0) return R3DG_OK;
    if (!perm || !normals || !incidents || !visibility || (!incident_dirs && !zsamples) || !lt || !vis_t || !consts)
        return invalid("shade_build_split: null buffer");
    return guarded([&]() -> int {
        launch_shade_build_split((hipStream_t)stream_, P, K, perm, normals, incidents, visibility, incident_dirs, zsamples,
                                 uniform_area, lt, vis_t, consts);
        return R3DG_OK;
    });
}

size_t r3dg_shade_env_footprints_bytes(int He, int We)
{
    return He > 0 && We > 0 ? (size_t)(He + 1) * (size_t)(We + 1) * 48 : 0;
}

int r3dg_shade_env_footprints(void* stream_, int He, int We, const float* env, float* footprints)
{
    if (He <= 0 || We <= 0 || He > 4095 || We > 4095 || !env || !footprints)
        return invalid("shade_env_footprints: bad size or null buffer");
    return guarded([&]() -> int {
        launch_shade_env_footprints((hipStream_t)stream_, He, We, env, footprints);
        return R3DG_OK;
    });
}

int r3dg_shade_forward_split(void* stream_, int P, int K, const int32_t* perm, const float* base_color, const float* roughness,
                             const float* normals, const float* viewdirs, const float* lt, const float* vis_t,
                             const float* consts, const float* zsamples, const float* env_transform, const float* env4, int He,
                             int We, float* out)
{
    if (P < 0 || K <= 0 || (K % 4) != 0 || He <= 0 || We <= 0 || He > 4095 || We > 4095)
        return invalid("shade_forward_split: bad sizes (K must be a multiple of 4)");
    if (P == 0) return R3DG_OK;
    if (!perm || !base_color || !roughness || !normals || !viewdirs || !lt || !vis_t || !consts || !zsamples || !env4 || !out)
        return invalid("shade_forward_split: null buffer");
    return guarded([&]() -> int {
        hipStream_t stream = (hipStream_t)stream_;
        StageTimer t(stream, ST_SHADE_FWD);
        launch_shade_forward_split(stream, P, K, perm, base_color, roughness, normals, viewdirs, lt, vis_t, consts, zsamples,
                                   env_transform, env4, He, We, out);
        t.stop();
        return R3DG_OK;
    });
}

int r3dg_shade_forward_transport(void* stream_, int P, int K, const float* base_color, const float* roughness,
                                 const float* normals, const float* viewdirs, const float* transport, const float* consts,
                                 const float* zsamples, const float* incident_dirs, float* out)
{
    if (P < 0 || K <= 0) return invalid("shade_forward_transport: bad P/K");
    if (P == 0) return R3DG_OK;
    if (!base_color || !roughness || !normals || !viewdirs || !transport || !consts || !out)
        return invalid("shade_forward_transport: null buffer");
    if (!zsamples && !incident_dirs) return invalid("shade_forward_transport: needs d_zsamples or d_incident_dirs");
    return guarded([&]() -> int {
        hipStream_t stream = (hipStream_t)stream_;
        StageTimer t(stream, ST_SHADE_FWD);
        launch_shade_forward_transport(stream, P, K, base_color, roughness, normals, viewdirs, transport, consts, zsamples,
                                       incident_dirs, out);
        t.stop();
        return R3DG_OK;
    });
}


int r3dg_shade_backward_cached(void* stream_, int P, int K, int M, const float* base_color, const float* roughness,
                               const float* normals, const float* viewdirs, const float* incidents, const float* env,
                               int He, int We, const float* env_transform, const float* visibility,
                               const float* incident_dirs, const float* incident_areas, const uint32_t* taps,
                               const float* dL_dpbr, const float* dL_ddiffuse_light, float* dL_dbase_color,
                               float* dL_droughness, float* dL_dviewdirs, float* dL_dincidents, float* dL_denv,
                               const float* block_absmax, int n_block_absmax)
{
    if (P < 0 || K <= 0 || He <= 0 || We <= 0) return invalid("shade_backward: bad P/K/env size");
    if (n_block_absmax < 0) return invalid("shade_backward: bad block_absmax count");
    if (He > 32767 || We > 32767) return invalid("shade_backward: environment map larger than 32767 texels per side");
    if (M != 1 && M != 4 && M != 9 && M != 16) return invalid("shade_backward: incidents must hold 1, 4, 9 or 16 SH coefficients");
    if (P == 0) return R3DG_OK;
    return guarded([&]() -> int {
        hipStream_t stream = (hipStream_t)stream_;
        StageTimer t(stream, ST_SHADE_BWD);
        launch_shade_backward(stream, P, K, M, base_color, roughness, normals, viewdirs, incidents, env, He, We,
                              env_transform, visibility, incident_dirs, incident_areas, dL_dpbr, dL_ddiffuse_light,
                              dL_dbase_color, dL_droughness, dL_dviewdirs, dL_dincidents, dL_denv, taps, block_absmax,
                              n_block_absmax);
        check_launch(stream, false, "shade_backward");
        t.stop();
        return R3DG_OK;
    });
}

size_t r3dg_shade_frs_tables_bytes(int K) { return K > 0 ? shade_frs_table_floats(K) * sizeof(float) : 0; }

int r3dg_shade_frs_supported(int K, int M, int He, int We) { return shade_frs_supported(K, M, He, We) ? 1 : 0; }

int r3dg_shade_frs_build_tables(void* stream_, int K, const float* zsamples, float* tables)
{
    if (K <= 0 || !zsamples || !tables) return invalid("shade_frs_build_tables: bad K or null buffer");
    return guarded([&]() -> int {
        launch_shade_frs_build_tables((hipStream_t)stream_, K, zsamples, tables);
        return R3DG_OK;
    });
}

int r3dg_shade_frs_classify(void* stream_, int P, const float* ray_normals, uint8_t* valid)
{
    if (P < 0) return invalid("shade_frs_classify: bad P");
    if (P == 0) return R3DG_OK;
    if (!ray_normals || !valid) return invalid("shade_frs_classify: null buffer");
    return guarded([&]() -> int {
        launch_shade_frs_classify((hipStream_t)stream_, P, ray_normals, valid);
        return R3DG_OK;
    });
}

int r3dg_stream_wait_stream(void* waiter, void* signaller)
{
    return guarded([&]() -> int {
        stream_wait_stream((hipStream_t)waiter, (hipStream_t)signaller);
        return R3DG_OK;
    });
}

// one wave that does nothing for `us` microseconds of the device's constant-rate wall clock (s_memrealtime)
__global__ void __launch_bounds__(64) spin_kernel(unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// one thread: *dst = *src (dst: device address of pinned host memory)
__global__ void store_u64_kernel(const unsigned long long* __restrict__ src, volatile unsigned long long* dst) { *dst = *src; }

int r3dg_store_u64_to_host(void* stream_, const void* d_src, void* h_pinned_dst)
{
    if (d_src == nullptr || h_pinned_dst == nullptr) return invalid("store_u64_to_host: null pointer");
    return guarded([&]() -> int {
        void* mapped = nullptr;
        R3DG_HIP(hipHostGetDevicePointer(&mapped, h_pinned_dst, 0));
        store_u64_kernel<<<1, 1, 0, (hipStream_t)stream_>>>((const unsigned long long*)d_src, (volatile unsigned long long*)mapped);
        check_launch((hipStream_t)stream_, false, "store_u64_kernel");
        return R3DG_OK;
    });
}

int r3dg_spin(void* stream_, float microseconds)
{
    if (!(microseconds >= 0.f) || microseconds > 1e6f) return invalid("spin: 0 .. 1e6 microseconds");
    return guarded([&]() -> int {
        int dev = 0, khz = 100000;
        R3DG_HIP(hipGetDevice(&dev));
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
        const unsigned long long ticks = (unsigned long long)((double)microseconds * 1e-3 * (double)khz);
        spin_kernel<<<1, 64, 0, (hipStream_t)stream_>>>(ticks);
        check_launch((hipStream_t)stream_, false, "spin_kernel");
        return R3DG_OK;
    });
}

// every wave of a device-filling grid: `iters` x 16 independent fp32 FMAs per lane (VALU issue is the only thing it does), the
// shader-clock counter (s_memtime) and the constant-rate wall clock (s_memrealtime) read on both sides.  out[0] += shader cycles,
// out[1] += wall ticks, out[2] += 1 per wave: sum(cycles) / sum(ticks) x wall-clock rate = the shader clock UNDER VALU LOAD.
__global__ void __launch_bounds__(256) clock_probe_kernel(int iters, unsigned long long* __restrict__ out, float* __restrict__ sink)
{
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = (float)(threadIdx.x + i) * 1e-3f;
    const float m = 0.999f, c = 1e-4f;
    const unsigned long long w0 = wall_clock64();
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) a[i] = __builtin_fmaf(a[i], m, c);
    }
    const long long t1 = clock64();
    const unsigned long long w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i];
    if (s == 123456.789f) sink[0] = s;                       // (keeps the loop)
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], (unsigned long long)(t1 - t0));
        atomicAdd(&out[1], w1 - w0);
        atomicAdd(&out[2], 1ull);
    }
}

int r3dg_clock_probe(void* stream_, int iters, unsigned long long* d_out3, float* d_sink, int* wall_clock_khz)
{
    if (iters <= 0 || !d_out3 || !d_sink) return invalid("clock_probe: bad arguments");
    return guarded([&]() -> int {
        int dev = 0, khz = 100000, cus = 256;
        R3DG_HIP(hipGetDevice(&dev));
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        if (wall_clock_khz) *wall_clock_khz = khz;
        R3DG_HIP(hipMemsetAsync(d_out3, 0, 3 * sizeof(unsigned long long), (hipStream_t)stream_));
        clock_probe_kernel<<<cus * 8, 256, 0, (hipStream_t)stream_>>>(iters, d_out3, d_sink);      // 8 waves per SIMD
        check_launch((hipStream_t)stream_, false, "clock_probe_kernel");
        return R3DG_OK;
    });
}

int r3dg_shade_frs_rotate(void* stream_, int P, const float* incidents, const float* ray_normals, float* cprime)
{
    if (P < 0) return invalid("shade_frs_rotate: bad sizes");
    if (P == 0) return R3DG_OK;
    if (!incidents || !ray_normals || !cprime) return invalid("shade_frs_rotate: null buffer");
    return guarded([&]() -> int {
        hipStream_t stream = (hipStream_t)stream_;
        StageTimer t(stream, ST_SHADE_AUX);
        launch_shade_frs_forward_aux(stream, P, incidents, ray_normals, cprime);
        return R3DG_OK;
    });
}

int r3dg_shade_frs_build_taps(void* stream_, int P, int K, const float* ray_normals, const float* zsamples, int He, int We,
                              uint32_t* taps)
{
    if (P < 0 || K <= 0 || He <= 0 || We <= 0 || He > 511 || We > 511) return invalid("shade_frs_build_taps: bad sizes");
    if (P == 0) return R3DG_OK;
    if (!ray_normals || !zsamples || !taps) return invalid("shade_frs_build_taps: null buffer");
    return guarded([&]() -> int {
        launch_shade_frs_build_taps((hipStream_t)stream_, P, K, ray_normals, zsamples, He, We, taps);
        return R3DG_OK;
    });
}

int r3dg_shade_frs_forward(void* stream_, int P, int K, const float* base_color, const float* roughness,
                           const float* normals, const float* viewdirs, const float* incidents, const float* env, int He,
                           int We, const float* visibility, float uniform_area, const uint32_t* taps,
                           const float* ray_normals, const float* zsamples, const float* tables, const uint8_t* valid,
                           const int32_t* invalid_list, int n_invalid, float* cprime, int flags, float* out,
                           void* listed_stream_, float* feature_rows)
{
    if (P < 0 || K <= 0 || He <= 0 || We <= 0 || n_invalid < 0 || n_invalid > P) return invalid("shade_frs_forward: bad sizes");
    if (!shade_frs_supported(K, 16, He, We))
        return invalid("shade_frs_forward: needs K % 4 == 0 and an environment texture that fits LDS (r3dg_shade_frs_supported)");
    if (P == 0) return R3DG_OK;
    if (!base_color || !roughness || !normals || !viewdirs || !incidents || !env || !visibility || !taps || !ray_normals ||
        !zsamples || !tables || !valid || !cprime || !out || (n_invalid > 0 && !invalid_list))
        return invalid("shade_frs_forward: null buffer");
    return guarded([&]() -> int {
        hipStream_t stream = (hipStream_t)stream_;
        const bool leave_room = (flags & R3DG_SHADE_LEAVE_ROOM) != 0;
        // the kernel on the listed Gaussians (disjoint rows of `out`) may run on a second stream, ordered after everything
        // queued on `stream` so far: it then runs beside the rotation and the main kernel instead of after them (the CALLER
        // joins that stream before anything reads `out`)
        hipStream_t lstream = listed_stream_ != nullptr ? (hipStream_t)listed_stream_ : stream;
        if (n_invalid > 0) {
            if (lstream != stream) {
                stream_wait_stream(lstream, stream);
            }
            StageTimer t(lstream, ST_SHADE_LISTED);
            launch_shade_frs_forward_listed(lstream, K, base_color, roughness, normals, viewdirs, incidents, env, He, We,
                                            visibility, ray_normals, zsamples, uniform_area, invalid_list, n_invalid, out,
                                            feature_rows);
        }
        if ((flags & R3DG_SHADE_ROTATED) == 0) {
            StageTimer t(stream, ST_SHADE_AUX);
            launch_shade_frs_forward_aux(stream, P, incidents, ray_normals, cprime);
        }
        {
            StageTimer t(stream, ST_SHADE_FWD);
            launch_shade_frs_forward_main(stream, P, K, base_color, roughness, normals, viewdirs, env, He, We, visibility,
                                          uniform_area, taps, ray_normals, tables, valid, cprime, leave_room, out, feature_rows);
        }
        return R3DG_OK;
    });
}

int r3dg_shade_frs_backward(void* stream_, int P, int K, const float* base_color, const float* roughness,
                            const float* normals, const float* viewdirs, const float* incidents, const float* env, int He,
                            int We, const float* visibility, float uniform_area, const uint32_t* taps,
                            const float* ray_normals, const float* zsamples, const float* tables, const uint8_t* valid,
                            const int32_t* invalid_list, int n_invalid, const float* cprime, float* dcprime,
                            const float* dL_dpbr, const float* dL_ddiffuse_light, float* dL_dbase_color,
                            float* dL_droughness, float* dL_dviewdirs, float* dL_dincidents, float* dL_denv,
                            const float* block_absmax, int n_block_absmax, void* rotate_stream_)
{
    // rotate_stream_ == R3DG_SHADE_NO_ROTATION_BACK: the caller finishes dL_dincidents itself (r3dg_shade_frs_incident_chain)
    const bool no_rotation_back = rotate_stream_ == R3DG_SHADE_NO_ROTATION_BACK;
    if (no_rotation_back) rotate_stream_ = nullptr;
    if (P < 0 || K <= 0 || He <= 0 || We <= 0 || n_invalid < 0 || n_invalid > P || n_block_absmax < 0)
        return invalid("shade_frs_backward: bad sizes");
    if (!shade_frs_supported(K, 16, He, We))
        return invalid("shade_frs_backward: needs K % 4 == 0 and an environment texture that fits LDS (r3dg_shade_frs_supported)");
    if (P == 0) return R3DG_OK;
    if (!base_color || !roughness || !normals || !viewdirs || !incidents || !env || !visibility || !taps || !ray_normals ||
        !zsamples || !tables || !valid || !cprime || !dcprime || !dL_dpbr || !dL_ddiffuse_light || !dL_dbase_color ||
        !dL_droughness || !dL_dviewdirs || !dL_dincidents || !dL_denv || (n_invalid > 0 && !invalid_list))
        return invalid("shade_frs_backward: null buffer");
    return guarded([&]() -> int {
        hipStream_t stream = (hipStream_t)stream_;
        int gmax_n = 1;
        const unsigned int* gmax;
        {
            StageTimer t(stream, ST_SHADE_AUX);
            gmax = launch_shade_frs_backward_aux(stream, P, dL_dpbr, dL_ddiffuse_light, block_absmax, n_block_absmax, &gmax_n);
        }
        // the kernel on the listed Gaussians goes FIRST (its rows of the per-Gaussian outputs are disjoint from the main
        // kernel's, the texture gradient is accumulated by both): a small launch that a caller can put beside whatever it has
        // running on another stream at this point (fused_step: the rasterizer's per-Gaussian geometry backward)
        // (measured the other way round, round 4: listed AFTER the main kernel lets the geometry backward and the SH group's Adam run
        // into the main kernel's start instead -- shading backward 0.243 -> 0.355 ms, 687 -> 642 it/s)
        auto listed = [&]() {
            if (n_invalid <= 0) return;
            StageTimer t(stream, ST_SHADE_LISTED);
            launch_shade_frs_backward_listed(stream, K, base_color, roughness, normals, viewdirs, incidents, env, He, We,
                                             visibility, ray_normals, zsamples, uniform_area, invalid_list, n_invalid, dL_dpbr,
                                             dL_ddiffuse_light, dL_dbase_color, dL_droughness, dL_dviewdirs, dL_dincidents,
                                             dL_denv, gmax, gmax_n);
        };
        listed();
        {
            StageTimer t(stream, ST_SHADE_BWD);
            launch_shade_frs_backward_main(stream, P, K, base_color, roughness, normals, viewdirs, env, He, We, visibility,
                                           uniform_area, taps, ray_normals, tables, valid, cprime, dcprime, dL_dpbr,
                                           dL_ddiffuse_light, dL_dbase_color, dL_droughness, dL_dviewdirs, dL_denv, gmax, gmax_n);
        }
        // the rotation back may run on a second stream (ordered after the main kernel by an event; the CALLER joins that stream
        // before anything reads dL_dincidents): it then overlaps whatever the caller queues next on `stream`.  It leaves the
        // listed Gaussians' rows (written above) alone.
        if (no_rotation_back) return R3DG_OK;
        hipStream_t rstream = rotate_stream_ != nullptr ? (hipStream_t)rotate_stream_ : stream;
        if (rstream != stream) {
            stream_wait_stream(rstream, stream);
        }
        {
            StageTimer t(rstream, ST_SHADE_AUX);
            launch_shade_frs_backward_rotate(rstream, P, ray_normals, dcprime, dL_dincidents, n_invalid > 0 ? valid : nullptr);
        }
        return R3DG_OK;
    });
}

int r3dg_shade_frs_incident_chain(void* stream_, int P, const float* ray_normals, const uint8_t* valid, const float* dcprime,
                                  float* dL_dincidents, float* incidents, float* exp_avg, float* exp_avg_sq, float* cprime,
                                  float lr, float lr_tail, float beta1, float beta2, float eps, int step, float grad_scale,
                                  const float* skip_flag, int listed_rows_in_dcprime)
{
    if (P < 0) return invalid("shade_frs_incident_chain: bad sizes");
    if (step < 1) return invalid("shade_frs_incident_chain: step counts from 1");
    if (P == 0) return R3DG_OK;
    if (!ray_normals || !dcprime || !dL_dincidents || !incidents || !exp_avg || !exp_avg_sq || !cprime)
        return invalid("shade_frs_incident_chain: null buffer");
    return guarded([&]() -> int {
        hipStream_t stream = (hipStream_t)stream_;
        StageTimer t(stream, ST_SHADE_AUX);
        launch_shade_frs_incident_chain(stream, P, ray_normals, valid, dcprime, dL_dincidents, incidents, exp_avg, exp_avg_sq,
                                        cprime, lr, lr_tail, beta1, beta2, eps, step, grad_scale, skip_flag,
                                        listed_rows_in_dcprime != 0 ? 1 : 0);
        return R3DG_OK;
    });
}

int r3dg_shade_backward(void* stream_, int P, int K, int M, const float* base_color, const float* roughness,
                        const float* normals, const float* viewdirs, const float* incidents, const float* env, int He,
                        int We, const float* env_transform, const float* visibility, const float* incident_dirs,
                        const float* incident_areas, const float* dL_dpbr, const float* dL_ddiffuse_light,
                        float* dL_dbase_color, float* dL_droughness, float* dL_dviewdirs, float* dL_dincidents,
                        float* dL_denv)
{
    return r3dg_shade_backward_cached(stream_, P, K, M, base_color, roughness, normals, viewdirs, incidents, env, He, We,
                                      env_transform, visibility, incident_dirs, incident_areas, nullptr, dL_dpbr,
                                      dL_ddiffuse_light, dL_dbase_color, dL_droughness, dL_dviewdirs, dL_dincidents, dL_denv,
                                      nullptr, 0);
}

static int re_check(int P, int Si, int Sd, int Sv, int K)
{
    if (P < 0 || K <= 0) return invalid("render_equation: bad P/sample_num");
    if (Si < 0 || Si > 16 || Sd < 0 || Sd > 16 || Sv < 0 || Sv > 16)
        return invalid("render_equation: SH coefficient counts must be in [0,16]");
    return R3DG_OK;
}

int r3dg_render_equation_forward(void* stream_, int P, int Si, int Sd, int Sv, const float* base_color,
                                 const float* roughness, const float* metallic, const float* normals,
                                 const float* viewdirs, const float* incidents_shs, const float* direct_shs,
                                 const float* visibility_shs, int sample_num, const float* rand_float,
                                 float* incident_dirs, float* pbr, float* diffuse_light)
{
    if (int e = re_check(P, Si, Sd, Sv, sample_num)) return e;
    if (P == 0) return R3DG_OK;
    return guarded([&]() -> int {
        hipStream_t stream = (hipStream_t)stream_;
        launch_re_forward(stream, false, P, Si, Sd, Sv, base_color, roughness, metallic, normals, viewdirs, incidents_shs,
                          direct_shs, visibility_shs, sample_num, rand_float, incident_dirs, pbr, nullptr, nullptr,
                          nullptr, nullptr, diffuse_light, nullptr, nullptr, nullptr, nullptr);
        check_launch(stream, false, "render_equation_forward");
        return R3DG_OK;
    });
}

int r3dg_render_equation_forward_complex(void* stream_, int P, int Si, int Sd, int Sv, const float* base_color,
                                         const float* roughness, const float* metallic, const float* normals,
                                         const float* viewdirs, const float* incidents_shs, const float* direct_shs,
                                         const float* visibility_shs, int sample_num, float* incident_dirs, float* pbr,
                                         float* incident_lights, float* local_incident_lights,
                                         float* global_incident_lights, float* incident_visibility, float* diffuse_light,
                                         float* local_diffuse_light, float* accum, float* rgb_d, float* rgb_s)
{
    if (int e = re_check(P, Si, Sd, Sv, sample_num)) return e;
    if (P == 0) return R3DG_OK;
    return guarded([&]() -> int {
        hipStream_t stream = (hipStream_t)stream_;
        launch_re_forward(stream, true, P, Si, Sd, Sv, base_color, roughness, metallic, normals, viewdirs, incidents_shs,
                          direct_shs, visibility_shs, sample_num, nullptr, incident_dirs, pbr, incident_lights,
                          local_incident_lights, global_incident_lights, incident_visibility, diffuse_light,
                          local_diffuse_light, accum, rgb_d, rgb_s);
        check_launch(stream, false, "render_equation_forward_complex");
        return R3DG_OK;
    });
}

int r3dg_render_equation_backward(void* stream_, int P, int Si, int Sd, int Sv, const float* base_color,
                                  const float* roughness, const float* metallic, const float* normals,
                                  const float* viewdirs, const float* incidents_shs, const float* direct_shs,
                                  const float* visibility_shs, int sample_num, const float* incident_dirs,
                                  const float* dL_dpbr, const float* dL_ddiffuse_light, float* dL_dbase_color,
                                  float* dL_droughness, float* dL_dmetallic, float* dL_dnormals, float* dL_dviewdirs,
                                  float* dL_dincidents_shs, float* dL_ddirect_shs, float* dL_dvisibility_shs)
{
    if (int e = re_check(P, Si, Sd, Sv, sample_num)) return e;
    if (P == 0) return R3DG_OK;
    return guarded([&]() -> int {
        hipStream_t stream = (hipStream_t)stream_;
        launch_re_backward(stream, P, Si, Sd, Sv, base_color, roughness, metallic, normals, viewdirs, incidents_shs,
                           direct_shs, visibility_shs, sample_num, incident_dirs, dL_dpbr, dL_ddiffuse_light,
                           dL_dbase_color, dL_droughness, dL_dmetallic, dL_dnormals, dL_dviewdirs, dL_dincidents_shs,
                           dL_ddirect_shs, dL_dvisibility_shs);
        check_launch(stream, false, "render_equation_backward");
        return R3DG_OK;
    });
}

int r3dg_stage2_activate(void* stream_, int P, const float* xyz, const float* scaling_raw, const float* rotation_raw,
                         const float* opacity_raw, const float* normal_raw, const float* base_raw,
                         const float* rough_raw, const float* campos, float* scales, float* rot, float* opacity,
                         float* normal, float* base_color, float* roughness, float* viewdirs, const float* viewmatrix,
                         float* features)
{
    return r3dg_stage2_activate_with(stream_, P, xyz, scaling_raw, rotation_raw, opacity_raw, normal_raw, base_raw, rough_raw,
                                     campos, scales, rot, opacity, normal, base_color, roughness, viewdirs, viewmatrix, features,
                                     0, nullptr, nullptr, nullptr, 0);
}

int r3dg_stage2_activate_with(void* stream_, int P, const float* xyz, const float* scaling_raw, const float* rotation_raw,
                              const float* opacity_raw, const float* normal_raw, const float* base_raw,
                              const float* rough_raw, const float* campos, float* scales, float* rot, float* opacity,
                              float* normal, float* base_color, float* roughness, float* viewdirs, const float* viewmatrix,
                              float* features, int n_env, const float* env_raw, float* env, float* zero, int n_zero)
{
    if (P < 0) return invalid("stage2_activate: bad P");
    if (n_env < 0 || n_zero < 0) return invalid("stage2_activate: bad side-job size");
    if (n_env > 0 && (!env_raw || !env)) return invalid("stage2_activate: null texture buffer");
    if (n_zero > 0 && !zero) return invalid("stage2_activate: null buffer to zero");
    if (P == 0 && n_env == 0 && n_zero == 0) return R3DG_OK;
    if (P == 0) {                      // (only side jobs: run them behind zero Gaussian workgroups)
        return guarded([&]() -> int {
            launch_s2_activate((hipStream_t)stream_, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                               nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, n_env, env_raw,
                               env, zero, n_zero);
            return R3DG_OK;
        });
    }
    if (!xyz || !scaling_raw || !rotation_raw || !opacity_raw || !normal_raw || !scales || !rot || !opacity || !normal)
        return invalid("stage2_activate: null buffer");
    if (base_raw && (!rough_raw || !campos || !base_color || !roughness || !viewdirs))
        return invalid("stage2_activate: stage-2 inputs/outputs incomplete");
    if (features && (!base_raw || !viewmatrix)) return invalid("stage2_activate: feature rows need the stage-2 inputs and the view matrix");
    return guarded([&]() -> int {
        StageTimer t((hipStream_t)stream_, ST_S2_ACTIVATE);
        launch_s2_activate((hipStream_t)stream_, P, xyz, scaling_raw, rotation_raw, opacity_raw, normal_raw, base_raw,
                           rough_raw, campos, scales, rot, opacity, normal, base_color, roughness, viewdirs, viewmatrix, features,
                           n_env, env_raw, env, zero, n_zero);
        return R3DG_OK;
    });
}

int r3dg_stage2_pack_features(void* stream_, int P, const float* xyz, const float* viewmatrix, const float* normal,
                              const float* base_color, const float* roughness, const float* shade_out, float* features,
                              float* light_l1_sum)
{
    if (P < 0) return invalid("stage2_pack_features: bad P");
    if (P == 0) return R3DG_OK;
    if (!xyz || !viewmatrix || !normal || !base_color || !roughness || !shade_out || !features)
        return invalid("stage2_pack_features: null buffer");
    return guarded([&]() -> int {
        StageTimer t((hipStream_t)stream_, ST_S2_PACK);
        launch_s2_pack((hipStream_t)stream_, P, xyz, viewmatrix, normal, base_color, roughness, shade_out, features,
                       light_l1_sum);
        return R3DG_OK;
    });
}

int r3dg_stage2_unpack_gradients(void* stream_, int P, const float* dL_dfeatures, const float* shade_out,
                                 float light_weight, float* dL_dpbr, float* dL_ddiffuse, float* block_absmax,
                                 float* light_l1_sum)
{
    if (P < 0) return invalid("stage2_unpack_gradients: bad P");
    if (P == 0) return R3DG_OK;
    if (!dL_dfeatures || !shade_out || !dL_dpbr || !dL_ddiffuse) return invalid("stage2_unpack_gradients: null buffer");
    return guarded([&]() -> int {
        StageTimer t((hipStream_t)stream_, ST_S2_UNPACK);
        launch_s2_unpack((hipStream_t)stream_, P, dL_dfeatures, shade_out, light_weight, dL_dpbr, dL_ddiffuse, block_absmax,
                         light_l1_sum);
        return R3DG_OK;
    });
}

int r3dg_stage2_activate_backward(void* stream_, int P, const float* xyz, const float* scaling_raw,
                                  const float* rotation_raw, const float* opacity_raw, const float* normal_raw,
                                  const float* base_raw, const float* rough_raw, const float* viewmatrix,
                                  const float* campos, const float* dL_dfeatures, const float* dL_dbase_shade,
                                  const float* dL_drough_shade, const float* dL_dviewdirs, const float* dL_dscales,
                                  const float* dL_drot, const float* dL_dopacity, const float* dL_dmeans3D, float* g_xyz,
                                  float* g_scaling, float* g_rotation, float* g_opacity, float* g_normal, float* g_base,
                                  float* g_rough)
{
    return r3dg_stage2_activate_backward_with(stream_, P, xyz, scaling_raw, rotation_raw, opacity_raw, normal_raw, base_raw,
                                              rough_raw, viewmatrix, campos, dL_dfeatures, dL_dbase_shade, dL_drough_shade,
                                              dL_dviewdirs, dL_dscales, dL_drot, dL_dopacity, dL_dmeans3D, g_xyz, g_scaling,
                                              g_rotation, g_opacity, g_normal, g_base, g_rough, 0, 0, nullptr, nullptr, nullptr,
                                              0.f, nullptr, nullptr, 0);
}

int r3dg_stage2_activate_backward_with(void* stream_, int P, const float* xyz, const float* scaling_raw,
                                  const float* rotation_raw, const float* opacity_raw, const float* normal_raw,
                                  const float* base_raw, const float* rough_raw, const float* viewmatrix,
                                  const float* campos, const float* dL_dfeatures, const float* dL_dbase_shade,
                                  const float* dL_drough_shade, const float* dL_dviewdirs, const float* dL_dscales,
                                  const float* dL_drot, const float* dL_dopacity, const float* dL_dmeans3D, float* g_xyz,
                                  float* g_scaling, float* g_rotation, float* g_opacity, float* g_normal, float* g_base,
                                  float* g_rough, int He, int We,
                                       const float* env_raw, const float* env, float* dL_denv, float w_tv,
                                       float* g_env_raw, float* tv_sum, int consume)
{
    if (P < 0) return invalid("stage2_activate_backward: bad P");
    if (He < 0 || We < 0) return invalid("stage2_activate_backward: bad texture size");
    const bool env_job = He * We != 0;
    if (env_job && (!env_raw || !env || !dL_denv || !g_env_raw)) return invalid("stage2_activate_backward: null texture buffer");
    if (P == 0) return env_job ? r3dg_stage2_env_backward(stream_, He, We, env_raw, env, dL_denv, w_tv, g_env_raw, tv_sum, consume) : R3DG_OK;
    if (!base_raw || !rough_raw || !dL_dfeatures || !dL_dbase_shade || !dL_drough_shade || !g_base || !g_rough)
        return invalid("stage2_activate_backward: null buffer");
    // g_xyz == NULL: frozen geometry -- only g_base / g_rough are produced and the geometry inputs are not read
    if (g_xyz != nullptr &&
        (!xyz || !scaling_raw || !rotation_raw || !opacity_raw || !normal_raw || !viewmatrix || !campos || !dL_dviewdirs ||
         !dL_dscales || !dL_drot || !dL_dopacity || !dL_dmeans3D || !g_scaling || !g_rotation || !g_opacity || !g_normal))
        return invalid("stage2_activate_backward: null buffer");
    return guarded([&]() -> int {
        StageTimer t((hipStream_t)stream_, ST_S2_ACTIVATE_BWD);
        launch_s2_activate_backward((hipStream_t)stream_, P, xyz, scaling_raw, rotation_raw, opacity_raw, normal_raw,
                                    base_raw, rough_raw, viewmatrix, campos, dL_dfeatures, dL_dbase_shade,
                                    dL_drough_shade, dL_dviewdirs, dL_dscales, dL_drot, dL_dopacity, dL_dmeans3D, g_xyz,
                                    g_scaling, g_rotation, g_opacity, g_normal, g_base, g_rough, env_job ? He : 0, env_job ? We : 0,
                                    env_job ? env_raw : nullptr, env, dL_denv, w_tv, g_env_raw, tv_sum, consume);
        return R3DG_OK;
    });
}

int r3dg_stage2_loss(void* stream_, int width, int height, const float* image, const float* opacity,
                     const float* feature, const float* pseudo_normal, const int32_t* n_contrib, const float* gt,
                     const float* bg, const float* image_mask, float w_l1, float w_pbr, float w_normal,
                     const float* extra_dimage, const float* extra_dsrgb, float* dL_dimage, float* dL_dopacity,
                     float* dL_dfeature, float* sums, int sparse_feature_gradients)
{
    if (width < 0 || height < 0) return invalid("stage2_loss: bad image size");
    if ((long long)width * height == 0) return R3DG_OK;
    if (!image || !opacity || !feature || !pseudo_normal || !n_contrib || !gt || !bg || !dL_dimage || !dL_dopacity ||
        !dL_dfeature || !sums)
        return invalid("stage2_loss: null buffer");
    return guarded([&]() -> int {
        StageTimer t((hipStream_t)stream_, ST_S2_LOSS);
        launch_s2_loss((hipStream_t)stream_, width * height, image, opacity, feature, pseudo_normal, n_contrib, gt, bg,
                       image_mask, w_l1, w_pbr, w_normal, extra_dimage, extra_dsrgb, dL_dimage, dL_dopacity, dL_dfeature, sums,
                       sparse_feature_gradients);
        return R3DG_OK;
    });
}

int r3dg_stage2_smooth_forward(void* stream_, int width, int height, const float* opacity, const float* feature,
                               const int32_t* n_contrib, const float* gt, const float* image_mask, float w_base_color,
                               float w_roughness, float w_light, float* scratch, float* sums3)
{
    if (width < 0 || height < 0) return invalid("stage2_smooth_forward: bad image size");
    if ((long long)width * height == 0) return R3DG_OK;
    if (!opacity || !feature || !n_contrib || !gt || !scratch || !sums3) return invalid("stage2_smooth_forward: null buffer");
    return guarded([&]() -> int {
        StageTimer t((hipStream_t)stream_, ST_S2_LOSS);
        launch_s2_smooth_forward((hipStream_t)stream_, width, height, opacity, feature, n_contrib, gt, image_mask,
                                 w_base_color, w_roughness, w_light, scratch, sums3);
        return R3DG_OK;
    });
}

int r3dg_stage2_smooth_backward(void* stream_, int width, int height, const float* opacity, const float* feature,
                                const int32_t* n_contrib, const float* image_mask, const float* scratch, float w_base_color,
                                float w_roughness, float w_light, int accumulate_normal, float* dL_dopacity,
                                float* dL_dfeature)
{
    if (width < 0 || height < 0) return invalid("stage2_smooth_backward: bad image size");
    if ((long long)width * height == 0) return R3DG_OK;
    if (!opacity || !feature || !n_contrib || !scratch || !dL_dopacity || !dL_dfeature)
        return invalid("stage2_smooth_backward: null buffer");
    return guarded([&]() -> int {
        StageTimer t((hipStream_t)stream_, ST_S2_LOSS);
        launch_s2_smooth_backward((hipStream_t)stream_, width, height, opacity, feature, n_contrib, image_mask, scratch,
                                  w_base_color != 0.f, w_roughness != 0.f, w_light != 0.f, accumulate_normal, dL_dopacity,
                                  dL_dfeature);
        return R3DG_OK;
    });
}

int r3dg_stage2_smooth_fused(void* stream_, int width, int height, const float* opacity, const float* feature,
                             const int32_t* n_contrib, const float* gt, const float* image_mask, float w_base_color,
                             float w_roughness, float w_light, int accumulate_normal, float* dL_dopacity, float* dL_dfeature,
                             float* sums3)
{
    if (width < 0 || height < 0) return invalid("stage2_smooth_fused: bad image size");
    if ((long long)width * height == 0) return R3DG_OK;
    if (!opacity || !feature || !n_contrib || !gt || !dL_dopacity || !dL_dfeature || !sums3)
        return invalid("stage2_smooth_fused: null buffer");
    return guarded([&]() -> int {
        StageTimer t((hipStream_t)stream_, ST_S2_LOSS);
        launch_s2_smooth_fused((hipStream_t)stream_, width, height, opacity, feature, n_contrib, gt, image_mask, w_base_color,
                               w_roughness, w_light, accumulate_normal, dL_dopacity, dL_dfeature, sums3);
        return R3DG_OK;
    });
}

int r3dg_stage1_pack_features(void* stream_, int P, const float* xyz, const float* viewmatrix, const float* normal,
                              float* features)
{
    if (P < 0) return invalid("stage1_pack_features: bad P");
    if (P == 0) return R3DG_OK;
    if (!xyz || !viewmatrix || !normal || !features) return invalid("stage1_pack_features: null buffer");
    return guarded([&]() -> int {
        StageTimer t((hipStream_t)stream_, ST_S2_PACK);
        launch_s1_pack((hipStream_t)stream_, P, xyz, viewmatrix, normal, features);
        return R3DG_OK;
    });
}

int r3dg_stage1_loss(void* stream_, int width, int height, const float* image, const float* opacity,
                     const float* feature, const float* pseudo_normal, const int32_t* n_contrib, const float* gt,
                     const float* image_mask, float w_l1, float w_mask_entropy, float w_normal, float w_normal_smooth,
                     float w_depth_var, const float* extra_dimage, float* edge_scratch, float* dL_dimage,
                     float* dL_dopacity, float* dL_dfeature, float* sums)
{
    if (width < 0 || height < 0) return invalid("stage1_loss: bad image size");
    if ((long long)width * height == 0) return R3DG_OK;
    if (!image || !opacity || !feature || !pseudo_normal || !n_contrib || !gt || !dL_dimage || !dL_dopacity ||
        !dL_dfeature || !sums)
        return invalid("stage1_loss: null buffer");
    if (w_normal_smooth != 0.f && !edge_scratch) return invalid("stage1_loss: the normal-smoothness term needs edge_scratch");
    return guarded([&]() -> int {
        StageTimer t((hipStream_t)stream_, ST_S2_LOSS);
        if (w_normal_smooth != 0.f)
            launch_s1_edge((hipStream_t)stream_, width, height, feature, opacity, n_contrib, gt, edge_scratch, sums + 4 * R3DG_SUM_SLOTS);
        launch_s1_loss((hipStream_t)stream_, width, height, image, opacity, feature, pseudo_normal, n_contrib, gt, image_mask,
                       w_l1, w_mask_entropy, w_normal, w_normal_smooth, w_depth_var, extra_dimage,
                       w_normal_smooth != 0.f ? edge_scratch : nullptr, dL_dimage, dL_dopacity, dL_dfeature, sums);
        return R3DG_OK;
    });
}

int r3dg_stage1_activate_backward(void* stream_, int P, const float* xyz, const float* scaling_raw,
                                  const float* rotation_raw, const float* opacity_raw, const float* normal_raw,
                                  const float* viewmatrix, const float* dL_dfeatures, const float* dL_dscales,
                                  const float* dL_drot, const float* dL_dopacity, const float* dL_dmeans3D, float* g_xyz,
                                  float* g_scaling, float* g_rotation, float* g_opacity, float* g_normal)
{
    if (P < 0) return invalid("stage1_activate_backward: bad P");
    if (P == 0) return R3DG_OK;
    if (!xyz || !scaling_raw || !rotation_raw || !opacity_raw || !normal_raw || !viewmatrix || !dL_dfeatures ||
        !dL_dscales || !dL_drot || !dL_dopacity || !dL_dmeans3D || !g_xyz || !g_scaling || !g_rotation || !g_opacity ||
        !g_normal)
        return invalid("stage1_activate_backward: null buffer");
    return guarded([&]() -> int {
        StageTimer t((hipStream_t)stream_, ST_S2_ACTIVATE_BWD);
        launch_s1_activate_backward((hipStream_t)stream_, P, xyz, scaling_raw, rotation_raw, opacity_raw, normal_raw,
                                    viewmatrix, dL_dfeatures, dL_dscales, dL_drot, dL_dopacity, dL_dmeans3D, g_xyz,
                                    g_scaling, g_rotation, g_opacity, g_normal);
        return R3DG_OK;
    });
}

int r3dg_stage2_pbr_srgb(void* stream_, int width, int height, const float* opacity, const float* feature,
                         const int32_t* n_contrib, const float* bg, float* srgb)
{
    if (width < 0 || height < 0) return invalid("stage2_pbr_srgb: bad image size");
    if ((long long)width * height == 0) return R3DG_OK;
    if (!opacity || !feature || !n_contrib || !bg || !srgb) return invalid("stage2_pbr_srgb: null buffer");
    return guarded([&]() -> int {
        launch_s2_pbr_srgb((hipStream_t)stream_, width * height, opacity, feature, n_contrib, bg, srgb);
        return R3DG_OK;
    });
}

int r3dg_stage2_normals_srgb(void* stream_, int width, int height, const float* viewmatrix, float tan_fovx, float tan_fovy,
                             float cx, float cy, const float* opacity, const float* depth, float* pseudo_normal,
                             float* surface_xyz, const float* feature, const int32_t* n_contrib, const float* bg, float* srgb)
{
    if (width < 0 || height < 0) return invalid("stage2_normals_srgb: bad image size");
    if ((long long)width * height == 0) return R3DG_OK;
    if ((long long)width * height > 0x7fffffffLL) return invalid("stage2_normals_srgb: image too large");
    if (!viewmatrix || !opacity || !depth || !pseudo_normal || !surface_xyz || !feature || !n_contrib || !bg || !srgb)
        return invalid("stage2_normals_srgb: null buffer");
    return guarded([&]() -> int {
        StageTimer t((hipStream_t)stream_, ST_NORMAL);
        // focal lengths exactly as the rasterizer forward derives them (rasterizer_impl.cu:239-240)
        const float focal_y = height / (2.0f * tan_fovy), focal_x = width / (2.0f * tan_fovx);
        launch_s2_normals_srgb((hipStream_t)stream_, width, height, viewmatrix, focal_x, focal_y, cx, cy, opacity, depth,
                               pseudo_normal, surface_xyz, feature, n_contrib, bg, srgb);
        return R3DG_OK;
    });
}

int r3dg_ssim_forward_pair(void* stream_, int width, int height, int channels, const float* x0, const float* x1,
                           const float* y, float* partials0, float* partials1, float* sum0, float* sum1)
{
    if (width < 0 || height < 0 || channels < 0) return invalid("ssim_forward: bad shape");
    if ((long long)width * height * channels == 0) return R3DG_OK;
    if (!x0 || !y || !partials0 || (x1 && !partials1)) return invalid("ssim_forward: null buffer");
    if ((long long)channels * 2 > 65535) return invalid("ssim_forward: too many channels");
    return guarded([&]() -> int {
        StageTimer t((hipStream_t)stream_, ST_SSIM);
        const float* x[2] = {x0, x1};
        float* partials[2] = {partials0, partials1};
        float* sum[2] = {sum0, sum1};
        launch_ssim_forward((hipStream_t)stream_, width, height, channels, x1 ? 2 : 1, x, y, partials, sum);
        return R3DG_OK;
    });
}

int r3dg_ssim_backward_pair(void* stream_, int width, int height, int channels, const float* x0, const float* x1,
                            const float* y, const float* partials0, const float* partials1, float scale0, float scale1,
                            float* grad_x0, float* grad_x1)
{
    if (width < 0 || height < 0 || channels < 0) return invalid("ssim_backward: bad shape");
    if ((long long)width * height * channels == 0) return R3DG_OK;
    if (!x0 || !y || !partials0 || !grad_x0 || (x1 && (!partials1 || !grad_x1))) return invalid("ssim_backward: null buffer");
    if ((long long)channels * 2 > 65535) return invalid("ssim_backward: too many channels");
    return guarded([&]() -> int {
        StageTimer t((hipStream_t)stream_, ST_SSIM);
        const float* x[2] = {x0, x1};
        float* partials[2] = {const_cast<float*>(partials0), const_cast<float*>(partials1)};
        const float scale[2] = {scale0, scale1};
        float* grad[2] = {grad_x0, grad_x1};
        launch_ssim_backward((hipStream_t)stream_, width, height, channels, x1 ? 2 : 1, x, y, partials, scale, grad);
        return R3DG_OK;
    });
}

int r3dg_ssim_forward(void* stream_, int width, int height, int channels, const float* x, const float* y,
                      float* partials, float* sum)
{
    return r3dg_ssim_forward_pair(stream_, width, height, channels, x, nullptr, y, partials, nullptr, sum, nullptr);
}

int r3dg_ssim_backward(void* stream_, int width, int height, int channels, const float* x, const float* y,
                       const float* partials, float scale, float* grad_x)
{
    return r3dg_ssim_backward_pair(stream_, width, height, channels, x, nullptr, y, partials, nullptr, scale, 0.f, grad_x,
                                   nullptr);
}

int r3dg_stage2_env_backward(void* stream_, int He, int We, const float* raw, const float* env, float* dL_denv,
                             float w_tv, float* g_raw, float* tv_sum, int consume)
{
    if (He < 0 || We < 0) return invalid("stage2_env_backward: bad texture size");
    if (He * We == 0) return R3DG_OK;
    if (!raw || !env || !dL_denv || !g_raw) return invalid("stage2_env_backward: null buffer");
    return guarded([&]() -> int {
        launch_s2_env_backward((hipStream_t)stream_, He, We, raw, env, dL_denv, w_tv, g_raw, tv_sum, consume);
        return R3DG_OK;
    });
}

int r3dg_adam_step(void* stream_, int n_groups, const r3dg_adam_group* groups, float beta1, float beta2, float eps,
                   int step, float grad_scale, const float* skip_flag)
{
    if (n_groups < 0 || n_groups > R3DG_ADAM_MAX_GROUPS) return invalid("adam_step: bad group count");
    if (step < 1) return invalid("adam_step: step counts from 1");
    if (n_groups == 0) return R3DG_OK;
    if (!groups) return invalid("adam_step: null group table");
    for (int i = 0; i < n_groups; i++) {
        if (groups[i].n >= (1ull << 32)) return invalid("adam_step: group larger than 2^32 elements");
        if (groups[i].n && (!groups[i].param || !groups[i].grad || !groups[i].exp_avg || !groups[i].exp_avg_sq))
            return invalid("adam_step: null buffer in group");
    }
    return guarded([&]() -> int {
        StageTimer t((hipStream_t)stream_, ST_ADAM);
        launch_adam((hipStream_t)stream_, n_groups, groups, beta1, beta2, eps, step, grad_scale, skip_flag);
        return R3DG_OK;
    });
}

int r3dg_relight_pack_features(void* stream_, int P, const float* xyz, const float* viewmatrix, const float* normal,
                               const float* base_color, const float* roughness, const float* shade_out, float* features)
{
    if (P < 0) return invalid("relight_pack_features: bad P");
    if (P == 0) return R3DG_OK;
    if (!xyz || !viewmatrix || !normal || !base_color || !roughness || !shade_out || !features)
        return invalid("relight_pack_features: null buffer");
    if ((size_t)features & 15) return invalid("relight_pack_features: features must be 16-byte aligned");
    return guarded([&]() -> int {
        StageTimer t((hipStream_t)stream_, ST_RELIGHT_PACK);
        launch_relight_pack((hipStream_t)stream_, P, xyz, viewmatrix, normal, base_color, roughness, shade_out, features);
        return R3DG_OK;
    });
}

int r3dg_relight_compose(void* stream_, int width, int height, float focal_x, float focal_y, float cx, float cy,
                         const float* viewmatrix, const float* light_transform, const float* envmap, int He, int We,
                         const float* image, const float* opacity, const float* feature, const int32_t* n_contrib,
                         float* pbr_env, float* render_env, float* env_only)
{
    if (width < 0 || height < 0 || He < 1 || We < 1) return invalid("relight_compose: bad shape");
    if ((long long)width * height == 0) return R3DG_OK;
    if ((long long)width * height >= (1ll << 31)) return invalid("relight_compose: image too large");
    if (!viewmatrix || !envmap || !opacity || !feature || !n_contrib) return invalid("relight_compose: null buffer");
    if (render_env && !image) return invalid("relight_compose: render_env needs the rendered image");
    if (!(focal_x > 0.f) || !(focal_y > 0.f)) return invalid("relight_compose: focal lengths must be positive");
    return guarded([&]() -> int {
        StageTimer t((hipStream_t)stream_, ST_RELIGHT_COMPOSE);
        launch_relight_compose((hipStream_t)stream_, width, height, focal_x, focal_y, cx, cy, viewmatrix, light_transform,
                               envmap, He, We, image, opacity, feature, n_contrib, pbr_env, render_env, env_only);
        return R3DG_OK;
    });
}

int r3dg_densify_accumulate(void* stream_, int P, const float* viewspace_grad, const float* normal_grad,
                            const int32_t* radii, const float* weights, float* xyz_accum, float* normal_accum,
                            float* denom, float* weights_accum, float* max_radii2D, const float* skip_flag)
{
    if (P < 0) return invalid("densify_accumulate: bad P");
    if (P == 0) return R3DG_OK;
    if (!viewspace_grad || !radii || !weights || !xyz_accum || !normal_accum || !denom || !weights_accum || !max_radii2D)
        return invalid("densify_accumulate: null buffer");
    return guarded([&]() -> int {
        StageTimer t((hipStream_t)stream_, ST_DENSIFY);
        launch_densify_accumulate((hipStream_t)stream_, P, viewspace_grad, normal_grad, radii, weights, xyz_accum,
                                  normal_accum, denom, weights_accum, max_radii2D, skip_flag);
        return R3DG_OK;
    });
}

size_t r3dg_densify_temp_bytes(int P) { return densify_temp_bytes((size_t)(P > 0 ? P : 0)); }

int r3dg_densify_plan(void* stream_, int P, const r3dg_densify_config* cfg, const float* scaling_raw,
                      const float* opacity_raw, const float* xyz_accum, const float* normal_accum, const float* denom,
                      const float* weights_accum, const float* max_radii2D, int32_t* src_row, int32_t* kind,
                      int32_t* counts, void* temp)
{
    if (P < 0) return invalid("densify_plan: bad P");
    if (!cfg || !counts) return invalid("densify_plan: null config / counts");
    if (cfg->mode != 0 && cfg->mode != 1) return invalid("densify_plan: mode is 0 (densify_and_prune) or 1 (prune)");
    if (cfg->n_split < 1 || cfg->n_split > 8) return invalid("densify_plan: n_split out of range");
    if (cfg->mode == 0 && !(cfg->split_divisor > 0.f)) return invalid("densify_plan: split_divisor must be positive");
    if ((int64_t)P * (cfg->n_split > 2 ? cfg->n_split : 2) >= (1ll << 31)) return invalid("densify_plan: row map too large");
    if (P > 0 && (!scaling_raw || !opacity_raw || !xyz_accum || !normal_accum || !denom || !weights_accum ||
                  !max_radii2D || !src_row || !kind || !temp))
        return invalid("densify_plan: null buffer");
    return guarded([&]() -> int {
        hipStream_t s = (hipStream_t)stream_;
        if (P == 0) {
            R3DG_HIP(hipMemsetAsync(counts, 0, 8 * sizeof(int32_t), s));
            return R3DG_OK;
        }
        StageTimer t(s, ST_DENSIFY);
        launch_densify_plan(s, P, *cfg, scaling_raw, opacity_raw, xyz_accum, normal_accum, denom, weights_accum,
                            max_radii2D, src_row, kind, counts, temp);
        return R3DG_OK;
    });
}

int r3dg_densify_gather(void* stream_, int rows_out, const int32_t* src_row, const int32_t* kind, int n_groups,
                        const r3dg_densify_group* groups, const float* xyz, const float* scaling_raw,
                        const float* rotation_raw, const float* normal_table, float split_divisor)
{
    if (rows_out < 0) return invalid("densify_gather: bad row count");
    if (n_groups < 0 || n_groups > R3DG_DENSIFY_MAX_GROUPS) return invalid("densify_gather: bad group count");
    if (rows_out == 0 || n_groups == 0) return R3DG_OK;
    if (!groups || !src_row || !kind) return invalid("densify_gather: null table / row map");
    for (int i = 0; i < n_groups; i++) {
        const r3dg_densify_group& g = groups[i];
        if (g.row_floats == 0 || !g.src_param || !g.dst_param) return invalid("densify_gather: empty group");
        if ((g.src_exp_avg != nullptr) != (g.src_exp_avg_sq != nullptr) ||
            (g.src_exp_avg && (!g.dst_exp_avg || !g.dst_exp_avg_sq)))
            return invalid("densify_gather: moments must be given as complete source/destination pairs");
        if (g.role == R3DG_DENSIFY_ROLE_XYZ && g.row_floats != 3) return invalid("densify_gather: xyz rows are 3 floats");
        if (g.role == R3DG_DENSIFY_ROLE_SCALING && g.row_floats != 3)
            return invalid("densify_gather: scaling rows are 3 floats");
        if (g.role > R3DG_DENSIFY_ROLE_SCALING) return invalid("densify_gather: unknown role");
        if (g.role != R3DG_DENSIFY_ROLE_COPY && (!xyz || !scaling_raw || !rotation_raw || !(split_divisor > 0.f)))
            return invalid("densify_gather: split sources missing");
        if ((uint64_t)rows_out * g.row_floats >= (1ull << 41)) return invalid("densify_gather: group too large");
    }
    return guarded([&]() -> int {
        StageTimer t((hipStream_t)stream_, ST_DENSIFY);
        launch_densify_gather((hipStream_t)stream_, rows_out, src_row, kind, n_groups, groups, xyz, scaling_raw,
                              rotation_raw, normal_table, split_divisor);
        return R3DG_OK;
    });
}

int r3dg_reset_opacity(void* stream_, int P, float* opacity_raw, float* exp_avg, float* exp_avg_sq)
{
    if (P < 0) return invalid("reset_opacity: bad P");
    if (P == 0) return R3DG_OK;
    if (!opacity_raw) return invalid("reset_opacity: null buffer");
    return guarded([&]() -> int {
        launch_reset_opacity((hipStream_t)stream_, P, 0.01f, opacity_raw, exp_avg, exp_avg_sq);
        return R3DG_OK;
    });
}

size_t r3dg_knn_temp_bytes(int P) { return knn_temp_bytes((size_t)(P > 0 ? P : 0)); }

int r3dg_knn_dist2(void* stream_, int P, const float* points, float* mean_dist2, void* temp)
{
    if (P < 0) return invalid("knn_dist2: bad P");
    if (P == 0) return R3DG_OK;
    if (!points || !mean_dist2 || !temp) return invalid("knn_dist2: null buffer");
    return guarded([&]() -> int {
        StageTimer t((hipStream_t)stream_, ST_KNN);
        knn_dist2((hipStream_t)stream_, P, points, mean_dist2, temp);
        return R3DG_OK;
    });
}

size_t r3dg_bvh_build_temp_bytes(int P) { return bvh_build_temp_bytes((size_t)(P > 0 ? P : 0)); }

int r3dg_bvh_build(void* stream_, int P, int32_t* nodes, float* aabbs, int64_t* morton, void* temp)
{
    if (P < 0) return invalid("bvh_build: bad P");
    if (P == 0) return R3DG_OK;
    if (!nodes || !aabbs || !morton || !temp) return invalid("bvh_build: null buffer");
    return guarded([&]() -> int {
        hipStream_t stream = (hipStream_t)stream_;
        StageTimer t(stream, ST_BVH_BUILD);
        bvh_build(stream, P, nodes, aabbs, (uint64_t*)morton, temp);
        t.stop();
        return R3DG_OK;
    });
}

int r3dg_bvh_trace_opacity(void* stream_, int64_t num_rays, int num_gaussians, const int32_t* nodes, const float* aabbs,
                           const float* rays_o, const float* rays_d, const float* means3D, const float* covs3D,
                           const float* opacities, const float* normals, int32_t* num_contributes,
                           float* rendered_opacity, int32_t* stack_overflow)
{
    if (num_rays < 0 || num_rays > 0x7fffffffll) return invalid("bvh_trace_opacity: bad ray count");
    if (num_rays == 0) return R3DG_OK;
    if (!stack_overflow) return invalid("bvh_trace_opacity: null overflow counter");
    return guarded([&]() -> int {
        hipStream_t stream = (hipStream_t)stream_;
        StageTimer t(stream, ST_BVH_TRACE);
        bvh_trace_opacity(stream, (int)num_rays, num_gaussians, nodes, aabbs, rays_o, rays_d, means3D, covs3D, opacities,
                          normals, num_contributes, rendered_opacity, stack_overflow);
        check_launch(stream, false, "bvh_trace_opacity");
        t.stop();
        return R3DG_OK;
    });
}

size_t r3dg_bvh_trace_records_bytes(int num_gaussians)
{
    return bvh_trace_records_bytes((size_t)(num_gaussians > 0 ? num_gaussians : 0));
}

int r3dg_bvh_pack_traversal(void* stream_, int num_gaussians, const int32_t* nodes, const float* aabbs, const float* means3D,
                            const float* covs3D, const float* opacities, const float* normals, void* records)
{
    if (num_gaussians < 0) return invalid("bvh_pack_traversal: bad Gaussian count");
    if (num_gaussians == 0) return R3DG_OK;
    if (!nodes || !aabbs || !means3D || !covs3D || !opacities || !normals || !records)
        return invalid("bvh_pack_traversal: null buffer");
    return guarded([&]() -> int {
        bvh_pack_traversal((hipStream_t)stream_, num_gaussians, nodes, aabbs, means3D, covs3D, opacities, normals, records);
        check_launch((hipStream_t)stream_, false, "bvh_pack_traversal");
        return R3DG_OK;
    });
}

int r3dg_bvh_trace_opacity_packed(void* stream_, int64_t num_rays, int num_gaussians, void* records, const float* rays_o,
                                  const float* rays_d, int32_t* num_contributes, float* rendered_opacity,
                                  int32_t* stack_overflow)
{
    if (num_rays < 0 || num_rays > 0x7fffffffll) return invalid("bvh_trace_opacity_packed: bad ray count");
    if (num_gaussians <= 0) return invalid("bvh_trace_opacity_packed: bad Gaussian count");
    if (num_rays == 0) return R3DG_OK;
    if (!records || !rays_o || !rays_d || !num_contributes || !rendered_opacity || !stack_overflow)
        return invalid("bvh_trace_opacity_packed: null buffer");
    return guarded([&]() -> int {
        hipStream_t stream = (hipStream_t)stream_;
        StageTimer t(stream, ST_BVH_TRACE);
        bvh_trace_opacity_packed(stream, (int)num_rays, num_gaussians, records, rays_o, rays_d, num_contributes,
                                 rendered_opacity, stack_overflow);
        check_launch(stream, false, "bvh_trace_opacity_packed");
        t.stop();
        return R3DG_OK;
    });
}

int r3dg_bvh_trace_visits(void* stream_, int num_gaussians, const void* records, uint64_t* node_and_leaf_steps)
{
    if (num_gaussians <= 0 || !records || !node_and_leaf_steps) return invalid("bvh_trace_visits: bad arguments");
    return guarded([&]() -> int {
        unsigned long long v[2];
        bvh_trace_visits((hipStream_t)stream_, num_gaussians, records, v);
        node_and_leaf_steps[0] = v[0];
        node_and_leaf_steps[1] = v[1];
        return R3DG_OK;
    });
}

size_t r3dg_sort_temp_bytes(int64_t n) { return sort_temp_bytes((size_t)(n > 0 ? n : 0)); }

int r3dg_bvh_trace_count(void* stream_, int64_t num_rays, const int32_t* nodes, const float* aabbs, const float* rays_o,
                         const float* rays_d, int32_t* num_contributes, int32_t* stack_overflow)
{
    if (num_rays < 0 || num_rays > 0x7fffffffll) return invalid("bvh_trace_count: bad ray count");
    if (num_rays == 0) return R3DG_OK;
    if (!nodes || !aabbs || !rays_o || !rays_d || !num_contributes || !stack_overflow)
        return invalid("bvh_trace_count: null buffer");
    return guarded([&]() -> int {
        bvh_trace_count((hipStream_t)stream_, (int)num_rays, nodes, aabbs, rays_o, rays_d, num_contributes, stack_overflow);
        check_launch((hipStream_t)stream_, false, "bvh_trace_count");
        return R3DG_OK;
    });
}

int r3dg_bvh_trace_fill(void* stream_, int64_t num_rays, const int32_t* nodes, const float* aabbs, const float* rays_o,
                        const float* rays_d, const float* means3D, const int32_t* num_contributes,
                        const int64_t* offsets_inclusive, uint64_t* keys, int32_t* point_list, float* position_list,
                        int32_t* ray_id_list)
{
    if (num_rays < 0 || num_rays > 0x7fffffffll) return invalid("bvh_trace_fill: bad ray count");
    if (num_rays == 0) return R3DG_OK;
    if (!nodes || !aabbs || !rays_o || !rays_d || !means3D || !num_contributes || !offsets_inclusive || !keys ||
        !point_list || !position_list || !ray_id_list)
        return invalid("bvh_trace_fill: null buffer");
    return guarded([&]() -> int {
        bvh_trace_fill((hipStream_t)stream_, (int)num_rays, nodes, aabbs, rays_o, rays_d, means3D, num_contributes,
                       offsets_inclusive, keys, point_list, position_list, ray_id_list);
        check_launch((hipStream_t)stream_, false, "bvh_trace_fill");
        return R3DG_OK;
    });
}

int r3dg_sort_pairs(void* stream_, int64_t n, uint64_t* keys_in, uint32_t* vals_in, uint64_t* keys_out,
                    uint32_t* vals_out, int end_bit, void* temp)
{
    if (n < 0 || end_bit < 1 || end_bit > 64) return invalid("sort_pairs: bad n/end_bit");
    if (n == 0) return R3DG_OK;
    return guarded([&]() -> int {
        sort_pairs((hipStream_t)stream_, (size_t)n, keys_in, vals_in, keys_out, vals_out, end_bit, temp, false);
        return R3DG_OK;
    });
}

}  // extern "C"
