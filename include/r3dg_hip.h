/*
 * r3dg_hip.h -- C ABI of libr3dg_hip.so, the MI355X (gfx950) implementation of the relightable-3DGS
 * hot path.  Plain pointers and sizes only; every pointer named d_* is DEVICE memory (HBM), everything
 * is fp32 / int32 unless stated.  All entry points enqueue work on `stream` (a hipStream_t passed as
 * void*; NULL = the null stream) and return 0 on success or a negative R3DG_E* code; the failing call's
 * message is available from r3dg_last_error().
 *
 * Each entry point replaces one function of the reference's pybind boundary (file:line under
 * NJU-3DV/Relightable3DGaussian):
 *   r3dg_rasterize_forward   <- CudaRasterizer::Rasterizer::forward   (cuda_rasterizer/rasterizer.h:31-66,
 *                               called from RasterizeGaussiansCUDA, rasterize_points.cu:36-141)
 *   r3dg_rasterize_backward  <- CudaRasterizer::Rasterizer::backward  (cuda_rasterizer/rasterizer.h:68-98,
 *                               called from RasterizeGaussiansBackwardCUDA, rasterize_points.cu:143-235)
 *   r3dg_mark_visible        <- CudaRasterizer::Rasterizer::markVisible (rasterizer.h:24-29, rasterize_points.cu:237-255)
 *   r3dg_shade_*             <- neilf.py:339-407 rendering_equation / GGX_specular (the LIVE stage-2 model)
 *   r3dg_render_equation_*   <- render_equation.h:7-46 (RenderEquationForwardCUDA / _complex / BackwardCUDA)
 *   r3dg_bvh_build           <- construct_bvh (bvh/include/construct.cuh, bvh/src/construct.cu:147-265)
 *   r3dg_bvh_trace_opacity   <- trace_bvh_opacity_cuda (bvh/include/trace.cuh:50-55, bvh/src/trace.cu:196-286)
 *   r3dg_knn_dist2           <- SimpleKNN::knn (submodules/simple-knn/simple_knn.cu:185-221)
 *   r3dg_densify_*           <- GaussianModel.add_densification_stats / densify_and_prune / prune / reset_opacity
 *                               (scene/gaussian_model.py:931-937, :893-915, :917-929, :563-566; train.py:158-175)
 */
#ifndef R3DG_HIP_H
#define R3DG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define R3DG_OK 0
#define R3DG_EINVAL (-1)   /* bad argument (shape / S limit / null pointer)  -> RuntimeError in the host mirror */
#define R3DG_EHIP (-2)     /* a HIP runtime call or kernel launch failed                                      */
#define R3DG_EALLOC (-3)   /* a resize callback returned NULL                                                 */

/* Resize callback, the C form of the reference's std::function<char*(size_t)> (rasterizer.h:32-34):
 * must return a DEVICE pointer to at least `bytes` bytes that stays valid until the matching backward. */
typedef void* (*r3dg_alloc_fn)(void* user, size_t bytes);

const char* r3dg_last_error(void);
/* Frees every scratch buffer the library holds (grow-only buffers per (device, stream): the tile backward's gradient records,
 * the shading kernels' records, the trace kernel's ray records; raw hipMalloc, outside any caching allocator).  Synchronises the
 * device.  Call it after a stream the library ran on has been destroyed, or to hand the memory back between phases; the next call
 * that needs a buffer allocates it again.  Not for use during stream capture (hipMalloc / hipFree). */
int r3dg_release_scratch(void);
int r3dg_version(void);

/* Limits (reference: forward F[33] forward.cu:312, backward 24 backward.cu:449). Ours: both 36. */
int r3dg_max_features_forward(void);
int r3dg_max_features_backward(void);

/* Scratch sizes (bytes) of the three opaque state buffers (rasterizer_impl.h:213-265 `required<T>`). */
size_t r3dg_geometry_state_bytes(int P);
size_t r3dg_image_state_bytes(int width, int height);
size_t r3dg_binning_state_bytes(int64_t num_rendered);
/* Byte offsets of named sub-arrays inside the state buffers, for tests/debugging only.
 * geometry: 0 depths f32[P], 1 clamped u8[3P], 2 radii i32[P], 3 means2D f32[2P], 4 cov3D f32[6P],
 *           5 conic_opacity f32[4P], 6 rgb f32[3P], 7 tiles_touched u32[P], 8 point_offsets u32[P]
 * image:    0 final_T f32[N], 1 n_contrib u32[N], 2 ranges u32[2T]
 * binning:  0 keys_unsorted u64[R], 1 keys u64[R], 2 vals_unsorted u32[R], 3 point_list u32[R] */
int r3dg_geometry_state_offsets(int P, size_t* offsets9);
/* byte offset of the uint64 instance count (num_rendered) of the last forward inside the geometry state */
size_t r3dg_geometry_state_total_offset(int P);
int r3dg_image_state_offsets(int width, int height, size_t* offsets3);
int r3dg_binning_state_offsets(int64_t num_rendered, size_t* offsets4);

/* Forward. Optional inputs (d_shs, d_colors_precomp, d_scales, d_rotations, d_cov3D_precomp) are NULL when
 * absent, exactly like the reference's nullptr convention (forward.cu:206,242).  Every image output and d_radii is
 * fully written (uninitialised memory is fine); only d_out_weights is accumulated and must be zero-filled by the
 * caller (the reference zero-fills everything, rasterize_points.cu:72-79).  `d_radii` may be NULL; so may d_out_weights
 * (the per-Gaussian blend weights, read by the densification statistics only: then they are not computed).
 * Returns num_rendered in *num_rendered_out.  One device->host sync (rasterizer_impl.cu:291). */
int r3dg_rasterize_forward(void* stream, r3dg_alloc_fn geometry_alloc, r3dg_alloc_fn binning_alloc,
                           r3dg_alloc_fn image_alloc, void* alloc_user, int P, int S, int D, int M,
                           const float* d_background, int width, int height, const float* d_means3D,
                           const float* d_shs, const float* d_colors_precomp, const float* d_features,
                           const float* d_opacities, const float* d_scales, float scale_modifier,
                           const float* d_rotations, const float* d_cov3D_precomp, const float* d_viewmatrix,
                           const float* d_projmatrix, const float* d_cam_pos, float tan_fovx, float tan_fovy,
                           float cx, float cy, int prefiltered, int compute_pseudo_normal, float* d_out_color,
                           float* d_out_opacity, float* d_out_depth, float* d_out_feature, float* d_out_normal,
                           float* d_out_surface_xyz, float* d_out_weights, int32_t* d_radii, int debug,
                           int* num_rendered_out);

/* The forward in two halves.  _begin takes the arguments of r3dg_rasterize_forward, runs the projection and starts the
 * asynchronous read-back of num_rendered (an event is recorded on `stream`); _finish waits for that event only, sizes the
 * binning state, orders the instances and renders.  Kernels the caller enqueues on `stream` between the two calls run
 * while the host waits for the count -- `d_features` (and every other pointer) is only dereferenced by the device, so the
 * feature rows may be produced in between.  *ticket is NULL when P == 0; _finish(NULL) is a no-op.  Every ticket must be
 * finished exactly once.  r3dg_rasterize_forward == _begin immediately followed by _finish. */
int r3dg_rasterize_forward_begin(void* stream, r3dg_alloc_fn geometry_alloc, r3dg_alloc_fn binning_alloc,
                                 r3dg_alloc_fn image_alloc, void* alloc_user, int P, int S, int D, int M,
                                 const float* d_background, int width, int height, const float* d_means3D,
                                 const float* d_shs, const float* d_colors_precomp, const float* d_features,
                                 const float* d_opacities, const float* d_scales, float scale_modifier,
                                 const float* d_rotations, const float* d_cov3D_precomp, const float* d_viewmatrix,
                                 const float* d_projmatrix, const float* d_cam_pos, float tan_fovx, float tan_fovy,
                                 float cx, float cy, int prefiltered, int compute_pseudo_normal, float* d_out_color,
                                 float* d_out_opacity, float* d_out_depth, float* d_out_feature, float* d_out_normal,
                                 float* d_out_surface_xyz, float* d_out_weights, int32_t* d_radii, int debug,
                                 void** ticket);
int r3dg_rasterize_forward_finish(void* ticket, int* num_rendered_out);
/* _finish with the instance ordering (key duplication, sort, tile ranges) on `ordering_stream` (NULL = the forward's
 * stream): it waits for the projection only and is joined before the tile kernel, so it can run concurrently with
 * whatever the caller queued on the forward's stream after _begin.  The binning state buffer is first touched on
 * `ordering_stream`. */
int r3dg_rasterize_forward_finish_on(void* ticket, void* ordering_stream, int* num_rendered_out);

/* The forward WITHOUT the host read-back of num_rendered (the reference's one synchronisation per forward,
 * rasterizer_impl.cu:291).  _begin_bounded queues the projection AND the instance ordering behind it at once on
 * `ordering_stream` (NULL = `stream`; otherwise ordered after everything queued on `stream` before the call), with the binning
 * state laid out for `capacity` instances (the value to pass as num_rendered to the backward, whose state layout it selects);
 * _finish_bounded makes `main_stream` wait for the ordering and renders there -- kernels the caller queues on `stream` after
 * _begin_ (the feature rows) run beside the whole front end when the two streams differ; the per-Gaussian outputs of the
 * projection (d_radii, the geometry state) are valid on `main_stream` after _finish_bounded.
 * A frame that needs MORE than `capacity` instances is dropped on the device: every tile list comes out empty (the
 * images are background only, n_contrib 0, all gradients of that frame zero) and *d_overflow_flag = 1.0f; otherwise the
 * flag is set to 0.0f (may be NULL); *d_overflow_count (may be NULL) is incremented for every dropped frame and never
 * reset by the library, so a caller that polls rarely still hears about it.  The true count stays readable in the geometry state
 * (r3dg_geometry_state_total_offset).  Needs the direct tile binning (the default) and at most 16384 tiles (a 2048x2048
 * image): its launches do not depend on the count.  r3dg_bounded_forward_supported(width, height) answers 1 when both hold
 * -- callers take the two-phase forward otherwise (_begin_bounded returns R3DG_EINVAL from _finish_bounded if asked anyway). */
int r3dg_bounded_forward_supported(int width, int height);
int r3dg_rasterize_forward_begin_bounded(void* stream, r3dg_alloc_fn geometry_alloc, r3dg_alloc_fn binning_alloc,
                                         r3dg_alloc_fn image_alloc, void* alloc_user, int P, int S, int D, int M,
                                         const float* d_background, int width, int height, const float* d_means3D,
                                         const float* d_shs, const float* d_colors_precomp, const float* d_features,
                                         const float* d_opacities, const float* d_scales, float scale_modifier,
                                         const float* d_rotations, const float* d_cov3D_precomp,
                                         const float* d_viewmatrix, const float* d_projmatrix, const float* d_cam_pos,
                                         float tan_fovx, float tan_fovy, float cx, float cy, int prefiltered,
                                         int compute_pseudo_normal, float* d_out_color, float* d_out_opacity,
                                         float* d_out_depth, float* d_out_feature, float* d_out_normal,
                                         float* d_out_surface_xyz, float* d_out_weights, int32_t* d_radii, int debug,
                                         void* ordering_stream, long long capacity, float* d_overflow_flag,
                                         unsigned int* d_overflow_count, void** ticket);
int r3dg_rasterize_forward_finish_bounded(void* ticket, void* main_stream);

/* Backward.  d_dL_dpix_d == NULL: the depth image carries no gradient (the caller's promise; the tile kernel then takes its
 * instances without a depth slot).  Every output is FULLY WRITTEN (zeros for invisible Gaussians, for feature channels outside
 * `active_features` and when nothing was rendered) and may be uninitialised: d_dL_dmean2D [P,3] (z = depth side channel),
 * d_dL_dopacity, d_dL_dcolor, d_dL_dfeature [P,S], d_dL_dmean3D, d_dL_dcov3D and -- when SHs / scales+rotations are the active
 * inputs -- d_dL_dsh, d_dL_dscale, d_dL_drot (otherwise those three are left untouched; the reference zero-fills all ten,
 * rasterize_points.cu:179-188).  d_dL_dconic [P,4] is SCRATCH (it receives the second moments the tile pass accumulates, not the
 * reference's intermediate).  Rounds 1-4 accumulated the first five with atomics into caller-zeroed memory; since round 5 the tile
 * kernel adds into one 64-byte record per Gaussian (library scratch) and a scatter pass writes the arrays. */
int r3dg_rasterize_backward(void* stream, int P, int S, int D, int M, int R, const float* d_background, int width,
                            int height, const float* d_means3D, const float* d_shs, const float* d_features,
                            const float* d_colors_precomp, const float* d_scales, float scale_modifier,
                            const float* d_rotations, const float* d_cov3D_precomp, const float* d_viewmatrix,
                            const float* d_projmatrix, const float* d_campos, float tan_fovx, float tan_fovy,
                            const int32_t* d_radii, const void* d_geom_buffer, const void* d_binning_buffer,
                            const void* d_img_buffer, const float* d_dL_dpix, const float* d_dL_dpix_o,
                            const float* d_dL_dpix_d, const float* d_dL_dpix_f, float* d_dL_dmean2D,
                            float* d_dL_dconic, float* d_dL_dopacity, float* d_dL_dcolor, float* d_dL_dfeature,
                            float* d_dL_dmean3D, float* d_dL_dcov3D, float* d_dL_dsh, float* d_dL_dscale,
                            float* d_dL_drot, int backward_geometry, int debug);

/* Same as r3dg_rasterize_backward, but the per-Gaussian geometry backward (K12+K13) is launched on `geometry_stream`, ordered
 * after the tile kernel by an event, so work that only needs the tile pass's FINAL outputs can follow on `stream` concurrently.
 * geometry_stream == stream is exactly r3dg_rasterize_backward.
 *   final on `stream` (behind the tile kernel + its scatter pass):  d_dL_dfeature, d_dL_dopacity, d_dL_dcolor;
 *   valid ONLY AFTER JOINING geometry_stream:  d_dL_dmean3D, d_dL_dcov3D, d_dL_dsh, d_dL_dscale, d_dL_drot -- AND d_dL_dmean2D
 *     and d_dL_dconic: since round 5 the tile pass leaves RAW MOMENTS in those two (S_x, S_y / the second moments) and the
 *     per-Gaussian geometry kernel turns them into the viewspace gradient IN PLACE (rasterizer_preprocess_bwd.hip).  A consumer on
 *     `stream` that has not joined (e.g. densification statistics reading dL_dmeans2D) would read raw moments or race with that
 *     rewrite.  r3dg_stream_wait_stream(stream, geometry_stream) is the join.
 * active_features (HOST array of n_active_features distinct channel indices, or n_active_features < 0 for "all"): the
 * caller's promise that every OTHER channel of d_dL_dpix_f is zero everywhere; those channels are then not carried
 * through the tile kernel at all and the scatter pass WRITES ZEROS into their d_dL_dfeature columns (every output is fully
 * written: no zero fill by the caller is needed or relied upon).  Same results, less work: a loss normally reads a few of
 * the S feature maps.
 * Scratch: the tile pass adds into one gradient record per Gaussian, P x (16 or, with all S = 16 channels live, 32) floats of
 * LIBRARY scratch per (device, stream) that ever ran a backward -- 64 / 128 bytes per Gaussian (2M Gaussians: 128 / 256 MB),
 * grown geometrically and kept until r3dg_release_scratch(). */
int r3dg_rasterize_backward_split(void* stream, void* geometry_stream, int P, int S, int D, int M, int R,
                                  const float* d_background, int width, int height, const float* d_means3D,
                                  const float* d_shs, const float* d_features, const float* d_colors_precomp,
                                  const float* d_scales, float scale_modifier, const float* d_rotations,
                                  const float* d_cov3D_precomp, const float* d_viewmatrix, const float* d_projmatrix,
                                  const float* d_campos, float tan_fovx, float tan_fovy, const int32_t* d_radii,
                                  const void* d_geom_buffer, const void* d_binning_buffer, const void* d_img_buffer,
                                  const float* d_dL_dpix, const float* d_dL_dpix_o, const float* d_dL_dpix_d,
                                  const float* d_dL_dpix_f, float* d_dL_dmean2D, float* d_dL_dconic,
                                  float* d_dL_dopacity, float* d_dL_dcolor, float* d_dL_dfeature, float* d_dL_dmean3D,
                                  float* d_dL_dcov3D, float* d_dL_dsh, float* d_dL_dscale, float* d_dL_drot,
                                  int backward_geometry, int debug, int n_active_features,
                                  const int* active_features);

/* The backward for FROZEN GEOMETRY: only d_dL_dfeature [P,S] (ACCUMULATED with atomics: zero it first) from d_dL_dpix_f
 * [S,H,W].  The stage-2 schedules of script/run_syn4.sh:27-33 and run_dtu.sh set the learning rates of positions, normals, SH
 * colour, opacity, scaling and rotation to 0: every gradient r3dg_rasterize_backward produces except dL_dfeature
 * (= sum over pixels of alpha * T * dL_dpixel_f, backward.cu:566) is then multiplied by 0 in the optimizer, so neither the
 * alpha-gradient recursion of the tile kernel nor the per-Gaussian geometry backward (K12 + K13) needs to run.  Same state
 * buffers, same active_features contract as r3dg_rasterize_backward_split; values equal that call's d_dL_dfeature up to the
 * order of the float atomics. */
int r3dg_rasterize_backward_features(void* stream, int P, int S, int R, int width, int height, const void* d_geom_buffer,
                                     const void* d_binning_buffer, const void* d_img_buffer, const float* d_dL_dpix_f,
                                     float* d_dL_dfeature, int n_active_features, const int* active_features, int debug);

int r3dg_mark_visible(void* stream, int P, const float* d_means3D, const float* d_viewmatrix,
                      const float* d_projmatrix, uint8_t* d_present);

/* Fused per-Gaussian shading integral, LIVE model (gaussian_renderer/neilf.py:339-407): for each Gaussian the mean
 * over K cached samples of (albedo/pi + GGX) * (max(SH_local(d),0) + env(d)*vis) * area * max(n.d,0).
 *   base_color[P,3] roughness[P] normals[P,3] viewdirs[P,3] incidents[P,M,3] (M = 1,4,9,16 SH coefficients)
 *   env[He,We,3]: ACTIVATED lat-long environment texture (softplus(DirectLightMap.env) or EnvLight.envmap);
 *   env_transform: optional row-major 3x3 light rotation applied to the lookup direction (envmap.py:39-42) or NULL
 *   visibility[P,K] incident_dirs[P,K,3] incident_areas[P,K]  (GaussianModel.update_visibility caches)
 * out[P,19] = pbr(3) diffuse_light(3) specular(3) mean_k incident_lights(3) mean_k local(3) mean_k global(3) mean_k vis(1)
 * Backward: gradients of <pbr,dL_dpbr> + <diffuse_light,dL_ddiffuse_light>; dL_dbase_color[P,3], dL_droughness[P],
 * dL_dviewdirs[P,3], dL_dincidents[P,M,3] are overwritten, dL_denv[He,We,3] is ACCUMULATED (zero it first). */
int r3dg_shade_forward(void* stream, int P, int K, int M, const float* d_base_color, const float* d_roughness,
                       const float* d_normals, const float* d_viewdirs, const float* d_incidents, const float* d_env,
                       int He, int We, const float* d_env_transform, const float* d_visibility,
                       const float* d_incident_dirs, const float* d_incident_areas, float* d_out);
/* The same forward with two optional savings for callers that own the sample caches (fused iteration, relight renderer):
 *   d_taps [P*K*3] uint32: the lat-long lookup of every cached direction (texel corner + two bilinear weights, 12 bytes per
 *     sample), built ONCE per visibility update by r3dg_shade_build_taps for the SAME incident_dirs, env size and
 *     env_transform -- the directions are frozen between updates (gaussian_model.py:312-342), so acos/atan2 need not be
 *     re-evaluated per iteration; NULL = evaluate the lookup in the kernel;
 *   flags & R3DG_SHADE_TRAIN_OUTPUTS: write only pbr (out[:,0:3]), diffuse_light (out[:,3:6]) and the mean visibility (out[:,18]),
 *     the columns the training feature row reads (neilf.py:120-122); the other 12 columns are left untouched;
 *   d_incident_areas == NULL: every sample has the area `uniform_area` (fibonacci_sphere_sampling assigns 2*pi to all of
 *     them, graphics_utils.py:36) -- saves the 4 bytes per sample of the area cache. */
#define R3DG_SHADE_TRAIN_OUTPUTS 1
#define R3DG_SHADE_TAPS_ARE_RADIANCE 2
#define R3DG_SHADE_LEAVE_ROOM 4      /* the caller runs other kernels beside this one: occupy half of each CU */
#define R3DG_SHADE_ROTATED 8         /* r3dg_shade_frs_forward only: d_cprime already holds r3dg_shade_frs_rotate's result */
int r3dg_shade_forward_cached(void* stream, int P, int K, int M, const float* d_base_color, const float* d_roughness,
                              const float* d_normals, const float* d_viewdirs, const float* d_incidents,
                              const float* d_env, int He, int We, const float* d_env_transform,
                              const float* d_visibility, const float* d_incident_dirs, const float* d_incident_areas,
                              float uniform_area, const uint32_t* d_taps, int flags, float* d_out);
/* d_env_radiance == NULL: lookup records.  d_env_radiance = an env[He,We,3] that is NOT being trained (relighting under a fixed
 * HDR map, envmap.py:35-53): the records hold the bilinearly sampled RADIANCE of every cached direction instead -- pass them
 * with R3DG_SHADE_TAPS_ARE_RADIANCE and the shading kernel touches no texture at all. */
int r3dg_shade_build_taps(void* stream, int64_t num_samples, const float* d_incident_dirs, const float* d_env_transform,
                          int He, int We, const float* d_env_radiance, uint32_t* d_taps);
/* Relighting under a FIXED light with static Gaussians (relighting.py:114-170 with configs/teaser, configs/nerf_syn: only the
 * camera moves): the view-independent part of rendering_equation (neilf.py:339-371) as a cache.
 * r3dg_shade_build_transport: d_radiance_inout [P*K*3] holds the sampled radiance of every cached direction
 *   (r3dg_shade_build_taps with d_env_radiance) and is REWRITTEN IN PLACE with the sample's transport
 *   (max(SH_incident(d), 0) + radiance * visibility) * area * max(n . d, 0); d_consts [P*16] receives per Gaussian
 *   diffuse_light 3 | mean incident light 3 | mean local light 3 | mean global light 3 | mean visibility 1 | 3 unused.
 *   Rebuild whenever the light, its rotation, the incident-light coefficients, the normals or the visibility cache change.
 * r3dg_shade_forward_transport: the per-frame remainder -- the GGX lobe of every sample against the cached transport --
 *   writes the same 19 columns per Gaussian as r3dg_shade_forward.  Directions: d_incident_dirs [P,K,3] as cached, or NULL
 *   to regenerate them from the normal and d_zsamples [K,3], the Fibonacci set around +z (graphics_utils.py:9-37 before
 *   the rotation; sh_utils.py:36-68): 12 instead of 24 bytes per sample. */
int r3dg_shade_build_transport(void* stream, int P, int K, int M, const float* d_normals, const float* d_incidents,
                               const float* d_visibility, const float* d_incident_dirs, const float* d_incident_areas,
                               float uniform_area, float* d_radiance_inout, float* d_consts);
int r3dg_shade_forward_transport(void* stream, int P, int K, const float* d_base_color, const float* d_roughness,
                                 const float* d_normals, const float* d_viewdirs, const float* d_transport,
                                 const float* d_consts, const float* d_zsamples, const float* d_incident_dirs,
                                 float* d_out);
/* Relighting under a light that TURNS WITH EVERY FRAME (relighting.py:160-161 with configs/nerf_syn_light or configs/tnt
 * light_transform.json) and static Gaussians: the "split transport" cache holds what does NOT depend on the light --
 *   d_lt [K,P,4]: per sample (max(SH_incident(d), 0) * a, a) with a = area * max(n . d, 0);  d_vis_t [K/4,P,4]: the visibility
 *   (K % 4 == 0);
 *   d_consts [P,4]: mean local light, mean visibility
 * sample-major and in the order d_perm [P] lists the Gaussians (the caller sorts them by normal: all lanes of a wave then look up
 * neighbouring texels) -- and r3dg_shade_forward_split evaluates, per frame, the lat-long lookup of the rotated direction
 * (d_env_footprints: the map as 48-byte bilinear footprints, r3dg_shade_env_footprints into r3dg_shade_env_footprints_bytes(He, We)
 * bytes: a lookup is three 16-byte loads of one record instead of four gathers), the transport and the GGX lobe: the 19 columns of r3dg_shade_forward
 * for 16 incident-light coefficients, a uniform sample area and Fibonacci directions (d_incident_dirs, or NULL to regenerate them
 * from the normals and d_zsamples as the cache was generated). */
int r3dg_shade_build_split(void* stream, int P, int K, const int32_t* d_perm, const float* d_normals, const float* d_incidents,
                           const float* d_visibility, const float* d_incident_dirs, const float* d_zsamples,
                           float uniform_area, float* d_lt, float* d_vis_t, float* d_consts);
size_t r3dg_shade_env_footprints_bytes(int He, int We);
int r3dg_shade_env_footprints(void* stream, int He, int We, const float* d_env, float* d_env_footprints);
int r3dg_shade_forward_split(void* stream, int P, int K, const int32_t* d_perm, const float* d_base_color,
                             const float* d_roughness, const float* d_normals, const float* d_viewdirs, const float* d_lt,
                             const float* d_vis_t, const float* d_consts, const float* d_zsamples,
                             const float* d_env_transform, const float* d_env_footprints, int He, int We, float* d_out);
/* ---- the same integral over a FIXED RAY SET (csrc/shading_frs.hpp) -----------------------------------------------------------
 * For callers whose cached directions are the Fibonacci set rotated to each Gaussian's normal, d_k = normalize(R(n) z_k) --
 * what GaussianModel.update_visibility produces (scene/gaussian_model.py:312-342 -> utils/graphics_utils.py:9-37,
 * rotation_between_z utils/sh_utils.py:36-68).  Then NOTHING per sample depends on a stored direction (SURVEY section 8f n2:
 * "store the frozen-normal snapshot, 12 B per Gaussian, instead of the [P,K,3] cache"):
 *   - the local incident light sum_i c_i Y_i(R z_k) equals sum_i c'_i Y_i(z_k) for the ROTATED coefficients c', and Y_i(z_k) is
 *     one K x 16 table for all Gaussians: both SH contractions run on the matrix cores against that constant table;
 *   - every dot product of d_k with a per-Gaussian vector w (the shading normal, the view vector) is z_k . (R^T w): one more
 *     matrix product per vector against the direction rows of the same table; the GGX half-vector terms follow from N.L, N.V
 *     and L.V alone;
 *   - the lat-long lookup of d_k is a constant between visibility updates: 8 bytes per sample (r3dg_shade_frs_build_taps).
 * What the kernels stream per sample: visibility (4 B) + the lookup record (8 B) -- round 3 read 28 B (direction 12, a 12-byte
 * record, visibility 4).  Same outputs and values (fp32 rounding) as r3dg_shade_forward_cached / r3dg_shade_backward_cached
 * with R3DG_SHADE_TRAIN_OUTPUTS and no light rotation; 16 incident-light coefficients, K % 4 == 0, a texture that fits LDS:
 * r3dg_shade_frs_supported.
 *   d_ray_normals [P,3]: the normals the ray set was GENERATED from (the snapshot update_visibility used, not the trained
 *     normal); d_zsamples [K,3]: the z set; d_tables: r3dg_shade_frs_build_tables(K, d_zsamples, r3dg_shade_frs_tables_bytes(K)
 *     bytes); d_taps [P,K,2] uint32: r3dg_shade_frs_build_taps for THIS texture size (rebuilt per visibility update);
 *   d_valid [P] bytes from r3dg_shade_frs_classify: 1 where R(n) is orthonormal to 2e-5 in fp32 -- normals within ~2.5 degrees of
 *     -z lose that to cancellation, their directions are not a rigid copy of the z set, and those Gaussians (d_invalid_list
 *     [n_invalid] int32 row indices, built by the caller from d_valid) are shaded inside the same call by wave-per-Gaussian
 *     kernels that regenerate their directions from the ray normal and evaluate the SH basis there;
 *   d_cprime [P,48]: scratch written by _forward (the rotated coefficients) and read by _backward of the same parameters;
 *   d_dcprime [P,48]: scratch of _backward.
 *   Sample areas: the fixed ray set has ONE area for every sample -- uniform_area, or 2*pi (what fibonacci_sphere_sampling assigns,
 *     utils/graphics_utils.py:36) when uniform_area == 0. */
int r3dg_shade_frs_supported(int K, int M, int He, int We);
size_t r3dg_shade_frs_tables_bytes(int K);
int r3dg_shade_frs_build_tables(void* stream, int K, const float* d_zsamples, float* d_tables);
int r3dg_shade_frs_classify(void* stream, int P, const float* d_ray_normals, uint8_t* d_valid);
/* d_taps [P,K,2]: per sample (x0 + 1) << 23 | mantissa(1 + wx1), (y0 + 1) << 23 | mantissa(1 + wy1) -- the texel corner and the
 * two bilinear weights (23-bit fixed point) of the lat-long lookup (direct_light_map.py:70-83) of d_k = normalize(R(n) z_k),
 * regenerated from the ray normal: no [P,K,3] direction array is read.  He, We <= 511. */
int r3dg_shade_frs_build_taps(void* stream, int P, int K, const float* d_ray_normals, const float* d_zsamples, int He, int We,
                              uint32_t* d_taps);
/* Stream plumbing for callers that spread one iteration over several streams of one device: `waiter` waits for everything queued
 * on `signaller` so far (hipEventRecord + hipStreamWaitEvent on a pooled event -- what the library's own entry points use between
 * the streams they are handed). */
int r3dg_stream_wait_stream(void* waiter, void* signaller);
/* Occupies `stream` (ONE wave, no memory traffic) for `microseconds` of the device's constant-rate wall clock.  Rehearsal tool:
 * fused_step prices an all-reduce it cannot run on a one-GPU box by queueing this behind each bucket's one-rank (identity)
 * collective, sized to the ring time of an assumed bus bandwidth (R3DG_DP_FAKE_COMM_GBS). */
int r3dg_spin(void* stream, float microseconds);
/* *h_pinned_dst = *d_src (8 bytes) behind everything queued on `stream`, written by a one-thread kernel straight into PINNED,
 * device-mapped host memory (hipHostMalloc / torch's pin_memory()) -- how a caller gets a count off the device without waiting
 * for it and without the runtime's blit path (an 8-byte hipMemcpyAsync to pinned memory runs as `__amd_rocclr_copyBuffer`, which
 * showed as 128 us on a hardware queue of the training iteration).  The host reads the slot after synchronising with the stream. */
int r3dg_store_u64_to_host(void* stream, const void* d_src, void* h_pinned_dst);
/* Measurement aid: a device-filling grid of VALU-only waves (iters x 16 fp32 FMAs per lane) that read the shader-clock counter and
 * the constant-rate wall clock on both sides.  d_out3[0] += shader cycles, [1] += wall ticks, [2] += waves (zeroed here);
 * shader clock under VALU load = d_out3[0] / d_out3[1] x *wall_clock_khz.  bench.py reports it beside the headline, so that a box
 * that ran at a lower clock says so (two boxes of the pool differed by 8 % on identical code in round 4).  d_sink: one float. */
int r3dg_clock_probe(void* stream, int iters, unsigned long long* d_out3, float* d_sink, int* wall_clock_khz);

/* The first step of r3dg_shade_frs_forward on its own: d_cprime [P,48] = the incident-light coefficients rotated into each
 * Gaussian's ray frame.  It depends on d_incidents and d_ray_normals only, so a caller can queue it (on another stream) as soon
 * as the coefficients are final and pass R3DG_SHADE_ROTATED to the forward. */
int r3dg_shade_frs_rotate(void* stream, int P, const float* d_incidents, const float* d_ray_normals, float* d_cprime);
int r3dg_shade_frs_forward(void* stream, int P, int K, const float* d_base_color, const float* d_roughness,
                           const float* d_normals, const float* d_viewdirs, const float* d_incidents, const float* d_env,
                           int He, int We, const float* d_visibility, float uniform_area, const uint32_t* d_taps,
                           const float* d_ray_normals, const float* d_zsamples, const float* d_tables,
                           const uint8_t* d_valid, const int32_t* d_invalid_list, int n_invalid, float* d_cprime, int flags,
                           float* d_out, void* listed_stream, float* d_feature_rows);
/*   listed_stream: NULL, or a second stream for the kernel on the listed Gaussians (ordered after everything queued on
 *   `stream` before the call; it then runs beside the rotation and the main kernel).  The caller joins listed_stream before anything
 *   reads d_out.
 *   d_feature_rows: NULL, or the rasterizer's [P,16] feature rows (neilf.py:115-122): the kernels then ALSO write pbr, diffuse
 *   light and mean visibility into columns 2..4, 12..14, 15 of each row -- with r3dg_stage2_activate(..., d_features) writing the
 *   other nine columns this replaces r3dg_stage2_pack_features (one launch and one pass over d_out less between the shading
 *   integral and the rasterizer; the light-smoothness sum then comes from r3dg_stage2_unpack_gradients). */
int r3dg_shade_frs_backward(void* stream, int P, int K, const float* d_base_color, const float* d_roughness,
                            const float* d_normals, const float* d_viewdirs, const float* d_incidents, const float* d_env,
                            int He, int We, const float* d_visibility, float uniform_area, const uint32_t* d_taps,
                            const float* d_ray_normals, const float* d_zsamples, const float* d_tables,
                            const uint8_t* d_valid, const int32_t* d_invalid_list, int n_invalid, const float* d_cprime,
                            float* d_dcprime, const float* d_dL_dpbr, const float* d_dL_ddiffuse_light,
                            float* d_dL_dbase_color, float* d_dL_droughness, float* d_dL_dviewdirs, float* d_dL_dincidents,
                            float* d_dL_denv, const float* d_block_absmax, int n_block_absmax, void* rotate_stream);
/* rotate_stream == R3DG_SHADE_NO_ROTATION_BACK: r3dg_shade_frs_backward leaves the coefficient gradient in the rotated frame
 * (d_dcprime) and does NOT write d_dL_dincidents for the Gaussians on the rotated path; the caller follows with
 * r3dg_shade_frs_incident_chain -- ONE pass per Gaussian that rotates the gradient back (writing d_dL_dincidents), applies the
 * Adam update of the incident-light group (the arithmetic of r3dg_adam_step for ONE group of [P,16,3] rows: columns 0..2 with
 * `lr`, the rest with `lr_tail`; `step` counts from 1; d_skip_flag as there) and rotates the NEW coefficients into the ray frames
 * (d_cprime, what r3dg_shade_frs_rotate would produce): 1536 instead of 2112 bytes per Gaussian and one launch instead of
 * three at the end of a whole training iteration (fused_step.py, "incident-light chain").
 * listed_rows_in_dcprime != 0: the world-frame gradient rows of the Gaussians OFF the rotated path (d_valid[g] == 0) are read from
 * their rows of d_dcprime instead of d_dL_dincidents -- for a caller that handed d_dcprime to r3dg_shade_frs_backward as
 * d_dL_dincidents too, so that ONE [P,48] buffer holds the whole coefficient gradient (rotated frame on the rotated path, world
 * frame off it) and can be summed over ranks as it is: the rotation is linear and the same on every rank (data-parallel
 * iterations, round 6: grad_scale = 1 / world). */
#define R3DG_SHADE_NO_ROTATION_BACK ((void*)(intptr_t)-1)
int r3dg_shade_frs_incident_chain(void* stream, int P, const float* d_ray_normals, const uint8_t* d_valid,
                                  const float* d_dcprime, float* d_dL_dincidents, float* d_incidents, float* d_exp_avg,
                                  float* d_exp_avg_sq, float* d_cprime, float lr, float lr_tail, float beta1, float beta2,
                                  float eps, int step, float grad_scale, const float* d_skip_flag, int listed_rows_in_dcprime);
/*   Launch order on `stream`: the kernel on the listed Gaussians, then the main kernel (a caller that has other work
 *   running on another stream when it calls this gets the small launch beside that work).
 *   rotate_stream: NULL, or a second stream for the rotation of the coefficient gradient back to d_dL_dincidents (ordered after the
 *   main kernel by an event; it overlaps the caller's next launches on `stream`).  The caller joins rotate_stream before anything
 *   reads d_dL_dincidents. */

int r3dg_shade_backward(void* stream, int P, int K, int M, const float* d_base_color, const float* d_roughness,
                        const float* d_normals, const float* d_viewdirs, const float* d_incidents, const float* d_env,
                        int He, int We, const float* d_env_transform, const float* d_visibility,
                        const float* d_incident_dirs, const float* d_incident_areas, const float* d_dL_dpbr,
                        const float* d_dL_ddiffuse_light, float* d_dL_dbase_color, float* d_dL_droughness,
                        float* d_dL_dviewdirs, float* d_dL_dincidents, float* d_dL_denv);

/* r3dg_shade_backward with the cached lookup records of r3dg_shade_build_taps (lookup mode, NOT radiance: the texture's
 * gradient needs the texel indices); d_taps == NULL = r3dg_shade_backward.
 * d_block_absmax (may be NULL) / n_block_absmax: non-negative floats whose maximum is max(|dL_dpbr|, |dL_ddiffuse_light|)
 * over all elements, +inf if any element is not finite -- the scale of the texture gradient's fixed-point accumulation.
 * r3dg_stage2_unpack_gradients writes them as it produces the two gradients; NULL: reduced here by an extra pass. */
int r3dg_shade_backward_cached(void* stream, int P, int K, int M, const float* d_base_color, const float* d_roughness,
                               const float* d_normals, const float* d_viewdirs, const float* d_incidents,
                               const float* d_env, int He, int We, const float* d_env_transform,
                               const float* d_visibility, const float* d_incident_dirs, const float* d_incident_areas,
                               const uint32_t* d_taps, const float* d_dL_dpbr, const float* d_dL_ddiffuse_light,
                               float* d_dL_dbase_color, float* d_dL_droughness, float* d_dL_dviewdirs,
                               float* d_dL_dincidents, float* d_dL_denv, const float* d_block_absmax,
                               int n_block_absmax);

/* The reference's render_equation.{cu,h} contract model (render_equation.h:7-46): metallic BRDF with a spherical-
 * Gaussian D, SH environment light direct_shs[Sd,3] (+0.5), SH visibility visibility_shs[P,Sv] (+0.5, clamped), SH local
 * light incidents_shs[P,Si,3]; rays are the Fibonacci set rotated to the normal (no 10-degree floor), optionally with
 * a per-sample random angle rand_float[P,K] in [0,1) (is_training; NULL otherwise).  Backward: dL_dbase_color[P,3],
 * dL_droughness[P], dL_dmetallic[P], dL_dnormals[P,3], dL_dviewdirs[P,3], dL_dincidents_shs[P,Si,3],
 * dL_dvisibility_shs[P,Sv] are overwritten, dL_ddirect_shs[Sd,3] is ACCUMULATED (zero it first).  The backward keeps
 * the reference's quirks Q1-Q4 (see oracle/shading_oracle.c) and replaces its race on dL_ddirect_shs by the sum. */
int r3dg_render_equation_forward(void* stream, int P, int Si, int Sd, int Sv, const float* d_base_color,
                                 const float* d_roughness, const float* d_metallic, const float* d_normals,
                                 const float* d_viewdirs, const float* d_incidents_shs, const float* d_direct_shs,
                                 const float* d_visibility_shs, int sample_num, const float* d_rand_float,
                                 float* d_incident_dirs, float* d_pbr, float* d_diffuse_light);
int r3dg_render_equation_forward_complex(void* stream, int P, int Si, int Sd, int Sv, const float* d_base_color,
                                         const float* d_roughness, const float* d_metallic, const float* d_normals,
                                         const float* d_viewdirs, const float* d_incidents_shs,
                                         const float* d_direct_shs, const float* d_visibility_shs, int sample_num,
                                         float* d_incident_dirs, float* d_pbr, float* d_incident_lights,
                                         float* d_local_incident_lights, float* d_global_incident_lights,
                                         float* d_incident_visibility, float* d_diffuse_light,
                                         float* d_local_diffuse_light, float* d_accum, float* d_rgb_d, float* d_rgb_s);
int r3dg_render_equation_backward(void* stream, int P, int Si, int Sd, int Sv, const float* d_base_color,
                                  const float* d_roughness, const float* d_metallic, const float* d_normals,
                                  const float* d_viewdirs, const float* d_incidents_shs, const float* d_direct_shs,
                                  const float* d_visibility_shs, int sample_num, const float* d_incident_dirs,
                                  const float* d_dL_dpbr, const float* d_dL_ddiffuse_light, float* d_dL_dbase_color,
                                  float* d_dL_droughness, float* d_dL_dmetallic, float* d_dL_dnormals,
                                  float* d_dL_dviewdirs, float* d_dL_dincidents_shs, float* d_dL_ddirect_shs,
                                  float* d_dL_dvisibility_shs);

/* Scalar accumulators ("sum" arguments of r3dg_stage2_pack_features, r3dg_stage2_loss, r3dg_stage1_loss, r3dg_ssim_forward,
 * r3dg_stage2_env_backward): every quantity is R3DG_SUM_SLOTS consecutive floats and its value is the SUM of them -- each
 * workgroup adds its total to one slot (same-address float atomics serialise at ~35 ns each on MI355X; thousands of
 * workgroups on one word take longer than the kernels themselves).  `d_sums` arrays hold quantity q at
 * d_sums + q * R3DG_SUM_SLOTS.  Callers zero them before the first kernel of an iteration. */
#define R3DG_SUM_SLOTS 32

/* ---- Stage-2 iteration glue: the elementwise code around the hot ops, fused -------------------
 * All tensors fp32, device, contiguous.  Shapes: xyz/scaling/normal/base_color raw [P,3], rotation raw [P,4],
 * opacity/roughness raw [P]; viewmatrix = world_view_transform (16 floats, row-vector convention, scene/cameras.py:62),
 * campos[3]; features [P,16]; shade_out [P,19] (r3dg_shade_forward).
 * r3dg_stage2_activate: GaussianModel.get_scaling/get_rotation/get_opacity/get_normal/get_base_color/get_roughness
 *   (scene/gaussian_model.py:183-232; exp, F.normalize, sigmoid, 0.03+0.77s, 0.09+0.9s) and
 *   viewdirs = normalize(campos - xyz) (gaussian_renderer/neilf.py:74-76).  base_raw == NULL skips the stage-2 outputs.
 *   d_features (may be NULL; needs d_viewmatrix and the stage-2 inputs): the columns of the feature row that do not depend on
 *   the shading integral are written here -- depth, depth^2 (0, 1), normal (5..7), base colour (8..10), roughness (11).
 * r3dg_stage2_pack_features: the S=16 feature row of neilf.py:115-122 = depth, depth^2, pbr, normal, base_color,
 *   roughness, diffuse_light, mean visibility; *light_l1_sum (may be NULL) += sum_p sum_c |diffuse_c - mean_c diffuse|
 *   (light-smoothness term, neilf.py:286-292).
 * r3dg_stage2_unpack_gradients: dL_dfeatures[P,16] -> the shading op's upstream gradients dL_dpbr[P,3] and
 *   dL_ddiffuse_light[P,3], the latter including light_weight * d(sum_c |diffuse_c - mean|)/d diffuse;
 *   d_block_absmax (may be NULL): [ceil(P/256)] floats, max |.| of the rows written by each block (+inf: not finite),
 *   for r3dg_shade_backward_cached.  d_light_l1_sum (may be NULL): += sum_p sum_c |diffuse_c - mean_c diffuse|, the term's
 *   value, for callers that did not run r3dg_stage2_pack_features (which adds the same sum).
 * r3dg_stage2_activate_backward: chain rule of every activation above; combines the rasterizer's dL_dscales, dL_drot,
 *   dL_dopacity, dL_dmeans3D, dL_dfeatures and the shading op's dL_dbase_color, dL_droughness, dL_dviewdirs into the
 *   raw-parameter gradients (all seven outputs fully written).  d_g_xyz == NULL = frozen geometry (run_syn4.sh / run_dtu.sh:
 *   learning rate 0 on everything but the PBR groups): only d_g_base / d_g_rough are written, from d_dL_dfeatures and the
 *   shading op's two gradients; the geometry inputs and outputs are not touched and may be NULL.
 * r3dg_stage2_loss: image-space terms of calculate_loss (neilf.py:212-318) and their gradients in one pass:
 *   sums[0] += sum |image - gt|, sums[1] += sum |srgb(pbr_img) - gt|, sums[2] += sum m^2 (normal_render - pseudo_normal)^2
 *   (m = d_image_mask [HW], the view's object mask of neilf.py:258-264; NULL = all ones)
 *   with feat = feature / max(opacity,1e-5) * (n_contrib > 0), pbr_img = feat[2:5]*opacity + (1-opacity)*bg;
 *   dL_dimage[3,HW], dL_dopacity[HW], dL_dfeature[16,HW] are fully written for weights w_* per element;
 *   d_extra_dL_dimage / d_extra_dL_dsrgb (may be NULL): gradients of further terms w.r.t. the image and the sRGB PBR
 *   image (the SSIM terms), added before the chain rule.  sparse_feature_gradients != 0: only the maps that carry a loss
 *   term are written (channels 2-4, and 5-7 when w_normal != 0; the others are left untouched and, with w_normal == 0,
 *   the normal maps are not read and sums[2] is not accumulated) -- for a rasterizer backward restricted to those
 *   channels (active_features). */
int r3dg_stage2_activate(void* stream, int P, const float* d_xyz, const float* d_scaling_raw,
                         const float* d_rotation_raw, const float* d_opacity_raw, const float* d_normal_raw,
                         const float* d_base_raw, const float* d_rough_raw, const float* d_campos, float* d_scales,
                         float* d_rotations, float* d_opacity, float* d_normal, float* d_base_color,
                         float* d_roughness, float* d_viewdirs, const float* d_viewmatrix, float* d_features);
int r3dg_stage2_pack_features(void* stream, int P, const float* d_xyz, const float* d_viewmatrix, const float* d_normal,
                              const float* d_base_color, const float* d_roughness, const float* d_shade_out,
                              float* d_features, float* d_light_l1_sum);
int r3dg_stage2_unpack_gradients(void* stream, int P, const float* d_dL_dfeatures, const float* d_shade_out,
                                 float light_weight, float* d_dL_dpbr, float* d_dL_ddiffuse_light,
                                 float* d_block_absmax, float* d_light_l1_sum);
int r3dg_stage2_activate_backward(void* stream, int P, const float* d_xyz, const float* d_scaling_raw,
                                  const float* d_rotation_raw, const float* d_opacity_raw, const float* d_normal_raw,
                                  const float* d_base_raw, const float* d_rough_raw, const float* d_viewmatrix,
                                  const float* d_campos, const float* d_dL_dfeatures, const float* d_dL_dbase_color,
                                  const float* d_dL_droughness, const float* d_dL_dviewdirs, const float* d_dL_dscales,
                                  const float* d_dL_drotations, const float* d_dL_dopacity, const float* d_dL_dmeans3D,
                                  float* d_g_xyz, float* d_g_scaling, float* d_g_rotation, float* d_g_opacity,
                                  float* d_g_normal, float* d_g_base, float* d_g_rough);
int r3dg_stage2_loss(void* stream, int width, int height, const float* d_image, const float* d_opacity,
                     const float* d_feature, const float* d_pseudo_normal, const int32_t* d_n_contrib,
                     const float* d_gt, const float* d_background, const float* d_image_mask, float w_l1, float w_pbr,
                     float w_normal, const float* d_extra_dL_dimage, const float* d_extra_dL_dsrgb, float* d_dL_dimage,
                     float* d_dL_dopacity, float* d_dL_dfeature, float* d_sums,
                     int sparse_feature_gradients);
/* The three edge-aware smoothness terms of the Synthetic4Relight / DTU objective (neilf.py:275-292 with
 * script/run_syn4.sh:34-36 / run_dtu.sh:36-38; first_order_edge_aware_loss, utils/loss_utils.py:104-105 = kornia 0.6.12's
 * Sobel / 8 with replicate padding):
 *   w_base_color sum_{c,d} |G_d (base_color_c m)| exp(-|G_d gt_c|) + w_roughness sum_{c,d} |G_d (roughness m)| exp(-|G_d gt_c|)
 *   + w_light sum_{c,d} |G_d (diffuse_c m)| exp(-|G_d normal_c|)          (the guide of the light term is the RENDERED normal
 *   and carries a gradient), with X = feature_X / max(opacity, 1e-5) * (n_contrib > 0) on the S=16 training feature image --
 *   base_color and diffuse additionally through rgb_to_srgb with its clip to [0,1] (they are results["base_color"] /
 *   results["diffuse"], neilf.py:153-155; utils/graphics_utils.py:207-213), roughness and normal as rendered -- and
 *   m = d_image_mask [HW] (NULL = all ones).  The weights carry the 1 / (3 H W) of the reference's means; a zero weight
 *   switches its term off.
 * _forward: sums3[0..2] (R3DG_SUM_SLOTS floats each) += the three UNWEIGHTED sums; d_scratch [30 * H * W] floats receives
 *   what _backward needs.  _backward (after r3dg_stage2_loss, same images): d_dL_dopacity is ADDED to; of d_dL_dfeature
 *   [16,HW] the maps 8..10 / 11 / 12..14 of the active terms are written, and with the light term the normal maps 5..7 are
 *   written (accumulate_normal == 0) or added to (!= 0: r3dg_stage2_loss wrote them, w_normal != 0). */
int r3dg_stage2_smooth_forward(void* stream, int width, int height, const float* d_opacity, const float* d_feature,
                               const int32_t* d_n_contrib, const float* d_gt, const float* d_image_mask,
                               float w_base_color, float w_roughness, float w_light, float* d_scratch, float* d_sums3);
int r3dg_stage2_smooth_backward(void* stream, int width, int height, const float* d_opacity, const float* d_feature,
                                const int32_t* d_n_contrib, const float* d_image_mask, const float* d_scratch,
                                float w_base_color, float w_roughness, float w_light, int accumulate_normal,
                                float* d_dL_dopacity, float* d_dL_dfeature);
/* _forward + _backward as ONE kernel that streams the image through registers (lane = column, the wave walks down 60-column
 * strips; neighbours by DPP lane shifts; the divided maps, the stencil outputs and the adjoint inputs never exist in HBM or
 * LDS: no d_scratch): same arithmetic bit for bit, same outputs (d_sums3 += the three unweighted sums,
 * d_dL_dopacity added to, d_dL_dfeature maps written / added to as described above). */
int r3dg_stage2_smooth_fused(void* stream, int width, int height, const float* d_opacity, const float* d_feature,
                             const int32_t* d_n_contrib, const float* d_gt, const float* d_image_mask, float w_base_color,
                             float w_roughness, float w_light, int accumulate_normal, float* d_dL_dopacity,
                             float* d_dL_dfeature, float* d_sums3);
/* sRGB-mapped PBR image [3,HW] exactly as r3dg_stage2_loss forms it (input of the SSIM term on the PBR image). */
int r3dg_stage2_pbr_srgb(void* stream, int width, int height, const float* d_opacity, const float* d_feature,
                         const int32_t* d_n_contrib, const float* d_background, float* d_srgb);

/* Launches folded into their neighbours (round 5; each was 5-20 us alone on the iteration's critical stream):
 * r3dg_stage2_activate_with: r3dg_stage2_activate + two SIDE JOBS that ride as extra workgroups of the same launch:
 *   d_env[0..n_env) = softplus(d_env_raw[..]) (DirectLightMap.get_env, scene/direct_light_map.py:18-23; beta 1, threshold 20 as
 *   torch.nn.functional.softplus) and d_zero[0..n_zero) = 0 (the iteration's loss sums).  n_env == 0 / n_zero == 0: that job is
 *   skipped.
 * r3dg_stage2_activate_backward_with: r3dg_stage2_activate_backward + r3dg_stage2_env_backward (same arguments, same results)
 *   in one launch; He * We == 0: no texture job.
 * r3dg_stage2_normals_srgb: the rasterizer forward's surface-point + pseudo-normal pass (forward.cu:398-491; what
 *   r3dg_rasterize_forward computes when compute_pseudo_normal != 0 -- call the forward with 0 and this afterwards) and
 *   r3dg_stage2_pbr_srgb as one per-pixel kernel.  d_pseudo_normal, d_surface_xyz [3,HW] and d_srgb [3,HW] are fully written;
 *   tan_fovx/tan_fovy/cx/cy as passed to the forward. */
int r3dg_stage2_activate_with(void* stream, int P, const float* d_xyz, const float* d_scaling_raw,
                              const float* d_rotation_raw, const float* d_opacity_raw, const float* d_normal_raw,
                              const float* d_base_raw, const float* d_rough_raw, const float* d_campos, float* d_scales,
                              float* d_rotations, float* d_opacity, float* d_normal, float* d_base_color,
                              float* d_roughness, float* d_viewdirs, const float* d_viewmatrix, float* d_features,
                              int n_env, const float* d_env_raw, float* d_env, float* d_zero, int n_zero);
int r3dg_stage2_activate_backward_with(void* stream, int P, const float* d_xyz, const float* d_scaling_raw,
                                       const float* d_rotation_raw, const float* d_opacity_raw, const float* d_normal_raw,
                                       const float* d_base_raw, const float* d_rough_raw, const float* d_viewmatrix,
                                       const float* d_campos, const float* d_dL_dfeatures, const float* d_dL_dbase_color,
                                       const float* d_dL_droughness, const float* d_dL_dviewdirs, const float* d_dL_dscales,
                                       const float* d_dL_drot, const float* d_dL_dopacity, const float* d_dL_dmeans3D,
                                       float* d_g_xyz, float* d_g_scaling, float* d_g_rotation, float* d_g_opacity,
                                       float* d_g_normal, float* d_g_base, float* d_g_rough, int He, int We,
                                       const float* d_env_raw, const float* d_env, float* d_dL_denv, float w_tv,
                                       float* d_g_env_raw, float* d_tv_sum, int consume);
int r3dg_stage2_normals_srgb(void* stream, int width, int height, const float* d_viewmatrix, float tan_fovx, float tan_fovy,
                             float cx, float cy, const float* d_opacity, const float* d_depth, float* d_pseudo_normal,
                             float* d_surface_xyz, const float* d_feature, const int32_t* d_n_contrib,
                             const float* d_background, float* d_srgb);

/* SSIM (utils/loss_utils.py:20-63: 11x11 Gaussian window, sigma 1.5, zero padding) of x against y, both [C,H,W].
 * _forward: *d_sum += sum of the SSIM map (divide by C*H*W for the reference's mean); d_partials [C,3,H,W] receives the
 * per-pixel partial derivatives the backward needs.  _backward: d_grad_x [C,H,W] = scale * d(sum SSIM)/dx (overwritten);
 * scale = -lambda_dssim / (C*H*W) for the loss term lambda_dssim * (1 - ssim). */
int r3dg_ssim_forward(void* stream, int width, int height, int channels, const float* d_x, const float* d_y,
                      float* d_partials, float* d_sum);
int r3dg_ssim_backward(void* stream, int width, int height, int channels, const float* d_x, const float* d_y,
                       const float* d_partials, float scale, float* d_grad_x);
/* The same for TWO images against one target in one launch each (the SH image and the sRGB PBR image of a stage-2
 * iteration, neilf.py:225-239): image 1 is optional (d_x1 NULL = the single-image call). */
int r3dg_ssim_forward_pair(void* stream, int width, int height, int channels, const float* d_x0, const float* d_x1,
                           const float* d_y, float* d_partials0, float* d_partials1, float* d_sum0, float* d_sum1);
int r3dg_ssim_backward_pair(void* stream, int width, int height, int channels, const float* d_x0, const float* d_x1,
                            const float* d_y, const float* d_partials0, const float* d_partials1, float scale0,
                            float scale1, float* d_grad_x0, float* d_grad_x1);

/* Stage-1 counterparts (plain 3DGS + normals, gaussian_renderer/render.py:15-130; r3dg_stage2_activate with
 * d_base_raw == NULL provides the activations): features [P,5] = normal(3), depth, depth^2.
 * r3dg_stage1_loss = the image-space terms of calculate_loss (gaussian_renderer/render.py:137-223) that script/run_nerf.sh:7-14
 * enables, values AND gradients in one call, with [normal, depth, depth2] = feature / max(opacity,1e-5) * (n_contrib > 0):
 *   w_l1 sum|image-gt| + w_mask_entropy sum -(m log o + (1-m) log(1-o)), o = clamp(opacity,1e-6,1-1e-6)      (:156-160)
 *   + w_normal sum (m normal - m pseudo_normal)^2                                                              (:162-167)
 *   + w_normal_smooth sum_{c,d} |G_d normal_c| exp(-|G_d gt_c|), G = 3x3 Sobel / 8, replicate padding (kornia 0.6.12
 *     spatial_gradient, utils/loss_utils.py:104-105)                                                           (:169-173)
 *   + w_depth_var sum sqrt(max(depth2 - depth^2, 1e-6))                                                        (:199-205)
 * d_image_mask [HW] may be NULL (all ones); d_edge_scratch [6*HW] floats is required when w_normal_smooth != 0;
 * d_extra_dL_dimage (may be NULL) is added to dL_dimage (the SSIM gradient).  The weights carry the 1/count of the
 * reference's means.  sums[0] += L1 sum, [1] += normal sum, [2] += entropy sum, [4] += edge-aware sum, [5] += sqrt-variance
 * sum (slot 3 is left to r3dg_ssim_forward); dL_dfeature is [5,HW]. */
int r3dg_stage1_pack_features(void* stream, int P, const float* d_xyz, const float* d_viewmatrix, const float* d_normal,
                              float* d_features);
int r3dg_stage1_loss(void* stream, int width, int height, const float* d_image, const float* d_opacity,
                     const float* d_feature, const float* d_pseudo_normal, const int32_t* d_n_contrib, const float* d_gt,
                     const float* d_image_mask, float w_l1, float w_mask_entropy, float w_normal, float w_normal_smooth,
                     float w_depth_var, const float* d_extra_dL_dimage, float* d_edge_scratch, float* d_dL_dimage,
                     float* d_dL_dopacity, float* d_dL_dfeature, float* d_sums);
int r3dg_stage1_activate_backward(void* stream, int P, const float* d_xyz, const float* d_scaling_raw,
                                  const float* d_rotation_raw, const float* d_opacity_raw, const float* d_normal_raw,
                                  const float* d_viewmatrix, const float* d_dL_dfeatures, const float* d_dL_dscales,
                                  const float* d_dL_drotations, const float* d_dL_dopacity, const float* d_dL_dmeans3D,
                                  float* d_g_xyz, float* d_g_scaling, float* d_g_rotation, float* d_g_opacity,
                                  float* d_g_normal);

/* Learnable environment texture (DirectLightMap, scene/direct_light_map.py:18-27): env = softplus(raw), [He,We,3].
 * g_raw = (dL_denv + w_tv * dTV(env)/denv) * softplus'(raw) with TV = mean (d/dh)^2 + mean (d/dw)^2 (tv_loss, utils/loss_utils.py:113-117: the env-smoothness term,
 * neilf.py:294-300); *tv_sum (may be NULL) += TV(env).  consume != 0: dL_denv is zeroed after it was read, so the
 * buffer can be handed to the next r3dg_shade_backward (which accumulates into it) without a fill. */
int r3dg_stage2_env_backward(void* stream, int He, int We, const float* d_raw, const float* d_env, float* d_dL_denv,
                             float w_tv, float* d_g_raw, float* d_tv_sum, int consume);

/* Adam over up to R3DG_ADAM_MAX_GROUPS parameter groups in ONE launch (torch.optim.Adam semantics, no weight decay /
 * amsgrad; GaussianModel.training_setup + step, scene/gaussian_model.py:465-497).  Elements whose index modulo `period`
 * is >= `split` use lr_tail (period 0: one rate) -- e.g. a [P,16,3] SH tensor with period 48, split 3 carries the
 * features_dc / features_rest rates.  `step` is the 1-based step count for the bias corrections; every gradient is
 * multiplied by `grad_scale` first (1/world_size after a sum all-reduce, 1 otherwise).  d_skip_flag (may be NULL): when the
 * float it points to is non-zero on the device the launch updates nothing (the overflow flag of a bounded forward). */
#define R3DG_ADAM_MAX_GROUPS 16
typedef struct r3dg_adam_group {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    uint64_t n;
    float lr, lr_tail;
    uint32_t period, split;
} r3dg_adam_group;
int r3dg_adam_step(void* stream, int n_groups, const r3dg_adam_group* groups, float beta1, float beta2, float eps,
                   int step, float grad_scale, const float* d_skip_flag);

/* ---- relight / eval frame glue (relighting.py:114-170 -> gaussian_renderer/neilf.py:74-209 with is_training=False) -----
 * r3dg_relight_pack_features: the S=28 eval feature row (neilf.py:124-130) from the activated parameters and the 19
 *   outputs of r3dg_shade_forward: depth, depth^2, pbr3, normal3, base_color3, roughness, diffuse3, specular3, incident /
 *   local / global light means (3 each), mean visibility.  d_features [P,28] must be 16-byte aligned.
 * r3dg_relight_compose: per pixel the camera ray in world space (Camera.get_world_directions, scene/cameras.py:79-91;
 *   d_viewmatrix = world_view_transform, whose upper-left 3x3 is the camera-to-world rotation), the lat-long lookup of the
 *   HDR environment map d_envmap [He,We,3] (EnvLight.direct_light, scene/envmap.py:35-53; d_light_transform = row-major
 *   3x3 rotation applied as dirs @ T^T, or NULL) and the composites of neilf.py:203-207 -- any output may be NULL:
 *     d_pbr_env = srgb(pbr * opacity + (1 - opacity) * env), d_render_env = image + (1 - opacity) * srgb(env),
 *     d_env_only = srgb(env), all [3,H,W].  d_feature is the rasterizer's [S>=5,H,W] feature image (pbr in channels 2..4,
 *   un-normalised: it is divided by max(opacity, 1e-5) and masked by n_contrib > 0 here, neilf.py:146-147). */
int r3dg_relight_pack_features(void* stream, int P, const float* d_xyz, const float* d_viewmatrix, const float* d_normal,
                               const float* d_base_color, const float* d_roughness, const float* d_shade_out,
                               float* d_features);
int r3dg_relight_compose(void* stream, int width, int height, float focal_x, float focal_y, float cx, float cy,
                         const float* d_viewmatrix, const float* d_light_transform, const float* d_envmap, int He, int We,
                         const float* d_image, const float* d_opacity, const float* d_feature,
                         const int32_t* d_n_contrib, float* d_pbr_env, float* d_render_env, float* d_env_only);

/* ---- densification bookkeeping -------------------------------------------------------------------
 * r3dg_densify_accumulate: GaussianModel.add_densification_stats (scene/gaussian_model.py:931-937) + the max-radii
 *   update of train.py:164-165, one pass.  The visibility filter is radii > 0 (render.py / neilf.py `visibility_filter`).
 *   d_viewspace_grad [P,3] is the gradient slot of the screen-space dummy (dL_dmeans2D); d_normal_grad [P,3] is the
 *   gradient of the raw normal parameter (NULL: the normal statistic is left alone); d_weights [P] are the rasterizer's
 *   per-Gaussian blend weights.  All five statistics are [P] floats updated in place.  d_skip_flag (may be NULL): a
 *   non-zero float on the device turns the call into a no-op (the overflow flag of a bounded forward). */
int r3dg_densify_accumulate(void* stream, int P, const float* d_viewspace_grad, const float* d_normal_grad,
                            const int32_t* d_radii, const float* d_weights, float* d_xyz_gradient_accum,
                            float* d_normal_gradient_accum, float* d_denom, float* d_weights_accum,
                            float* d_max_radii2D,
                            const float* d_skip_flag);

/* Thresholds of one densify_and_prune (mode 0, gaussian_model.py:893-915) or prune (mode 1, :917-929) call.  The
 * products the reference forms in Python doubles are passed already rounded to fp32:
 *   dense_size = percent_dense * scene_extent (:855, :804), world_size_limit = 0.1 * extent (:909),
 *   split_divisor = 0.8 * n_split (:814).  max_screen_size 0 = None (no screen / world size tests). */
typedef struct r3dg_densify_config {
    int32_t mode;
    int32_t n_split;
    float grad_threshold, grad_normal_threshold, min_opacity, weights_threshold;
    float dense_size, world_size_limit, split_divisor, max_screen_size;
} r3dg_densify_config;

/* r3dg_densify_plan: every decision of the call as ONE row map.  d_src_row / d_kind need max(2, n_split) * P entries;
 *   on return entries [0, d_counts[0]) hold, in the reference's output order (surviving originals, surviving clones,
 *   n_split blocks of surviving split children), the source row and its kind: -1 original (Adam moments travel with
 *   it), -2 clone (moments 0), k >= 0 split child that uses row k of the caller's standard-normal table
 *   [n_split * d_counts[3], 3] (row b * d_counts[3] + rank of the parent among ALL split Gaussians -- the row
 *   torch.normal would have filled, so the random stream is consumed exactly as in the reference).
 *   d_counts int32[8]: [0] rows out, [1] surviving originals, [2] surviving clones, [3] split Gaussians,
 *   [4] split Gaussians whose children survive.  d_temp: r3dg_densify_temp_bytes(P).  Statistics are [P] floats. */
size_t r3dg_densify_temp_bytes(int P);
int r3dg_densify_plan(void* stream, int P, const r3dg_densify_config* config, const float* d_scaling_raw,
                      const float* d_opacity_raw, const float* d_xyz_gradient_accum,
                      const float* d_normal_gradient_accum, const float* d_denom, const float* d_weights_accum,
                      const float* d_max_radii2D, int32_t* d_src_row, int32_t* d_kind, int32_t* d_counts, void* d_temp);

/* r3dg_densify_gather: builds every output tensor from the row map in ONE launch (the reference runs cat + cat +
 *   mask + mask over each parameter and both Adam moments: cat_tensors_to_optimizer :721-744, _prune_optimizer :681-698).
 *   Groups are [rows, row_floats] fp32; exp_avg pointers may be NULL (no optimizer state).  Split children: the
 *   ROLE_XYZ group receives R(q) (exp(scaling) * z) + xyz, the ROLE_SCALING group log(exp(scaling) / split_divisor)
 *   (densify_and_split :806-814); d_xyz / d_scaling_raw / d_rotation_raw are the SOURCE tensors of those formulas. */
#define R3DG_DENSIFY_MAX_GROUPS 24
#define R3DG_DENSIFY_ROLE_COPY 0
#define R3DG_DENSIFY_ROLE_XYZ 1
#define R3DG_DENSIFY_ROLE_SCALING 2
typedef struct r3dg_densify_group {
    const float* src_param;
    const float* src_exp_avg;
    const float* src_exp_avg_sq;
    float* dst_param;
    float* dst_exp_avg;
    float* dst_exp_avg_sq;
    uint32_t row_floats, role;
} r3dg_densify_group;
int r3dg_densify_gather(void* stream, int rows_out, const int32_t* d_src_row, const int32_t* d_kind, int n_groups,
                        const r3dg_densify_group* groups, const float* d_xyz, const float* d_scaling_raw,
                        const float* d_rotation_raw, const float* d_normal_table, float split_divisor);

/* GaussianModel.reset_opacity (:563-566): raw opacity <- inverse_sigmoid(min(sigmoid(raw), 0.01)) and zeroed Adam
 * moments (replace_tensor_to_optimizer :667-679; moment pointers may be NULL). */
int r3dg_reset_opacity(void* stream, int P, float* d_opacity_raw, float* d_exp_avg, float* d_exp_avg_sq);

/* distCUDA2 (submodules/simple-knn/spatial.cu:14-26 -> SimpleKNN::knn, simple_knn.cu:185-221): d_mean_dist2[i] = mean of
 * the squared distances from point i to its 3 nearest neighbours (FLT_MAX terms when P < 4, like the reference). */
size_t r3dg_knn_temp_bytes(int P);
int r3dg_knn_dist2(void* stream, int P, const float* d_points, float* d_mean_dist2, void* d_temp);

/* LBVH over per-Gaussian leaf boxes + visibility trace (reference bvh/include/bvh.h:5-18).
 * r3dg_bvh_build: d_nodes int32[2P-1,5] = (parent,left,right,object_id,leaf_count) and d_aabbs float[2P-1,6] =
 *   (lower xyz, upper xyz) arrive initialised as bvh/__init__.py:31-57 prepares them (nodes -1, counts 0 internal /
 *   1 leaf, leaf boxes in rows P-1..) and are completed IN PLACE; d_morton int64[P] receives the 64-bit codes.
 * r3dg_bvh_trace_opacity: one ray per (rays_o, rays_d) row; covs3D is the 6-vector INVERSE covariance
 *   (GaussianModel.get_inverse_covariance); outputs must be pre-set by the caller to 0 / 1 (bvh.cu:101-102);
 *   *d_stack_overflow (zeroed by the caller) counts rays that needed more than the 64-entry traversal stack;
 *   num_gaussians = P of the tree (rows of d_means3D): lets the library repack the tree into one 64-byte record per node for
 *   the walk (<= 0: unknown, the arrays are walked as they are).  Rays are traced in blocks of 256 consecutive rows; callers
 *   that can should order them so that consecutive rays start near each other (train_step.update_visibility does). */
size_t r3dg_bvh_build_temp_bytes(int P);
int r3dg_bvh_build(void* stream, int P, int32_t* d_nodes, float* d_aabbs, int64_t* d_morton, void* d_temp);
int r3dg_bvh_trace_opacity(void* stream, int64_t num_rays, int num_gaussians, const int32_t* d_nodes, const float* d_aabbs,
                           const float* d_rays_o, const float* d_rays_d, const float* d_means3D,
                           const float* d_covs3D, const float* d_opacities, const float* d_normals,
                           int32_t* d_num_contributes, float* d_rendered_opacity, int32_t* d_stack_overflow);

/* The same trace for callers that keep a tracer around (bvh.RayTracer; update_visibility traces the ray bundles in chunks
 * against ONE tree): the 64-byte traversal records r3dg_bvh_trace_opacity rebuilds on every call are packed ONCE into a buffer
 * the caller owns -- r3dg_bvh_trace_records_bytes(P) bytes, valid while the tree and the four per-Gaussian arrays are
 * unchanged -- and traced any number of times.  The buffer also holds the per-XCD ray-queue heads of a trace in flight: one
 * trace at a time per buffer (tracers on different streams use their own buffers; r3dg_bvh_trace_opacity keeps one private
 * scratch per device and stream).  Results are bit-identical to r3dg_bvh_trace_opacity. */
size_t r3dg_bvh_trace_records_bytes(int num_gaussians);
int r3dg_bvh_pack_traversal(void* stream, int num_gaussians, const int32_t* d_nodes, const float* d_aabbs,
                            const float* d_means3D, const float* d_covs3D, const float* d_opacities,
                            const float* d_normals, void* d_records);
int r3dg_bvh_trace_opacity_packed(void* stream, int64_t num_rays, int num_gaussians, void* d_records, const float* d_rays_o,
                                  const float* d_rays_d, int32_t* d_num_contributes, float* d_rendered_opacity,
                                  int32_t* d_stack_overflow);
/* Measurement ("report rays/s and node-visits/s"): with R3DG_OPT_TRACE_COUNT_VISITS = 1 the phased trace counts
 * its node steps (one slab test of both children of a node, trace.cu:228-246) and leaf steps (one Gaussian evaluated,
 * :247-276); r3dg_bvh_trace_visits synchronises `stream` and returns the two sums of the LAST trace over `d_records` in
 * node_and_leaf_steps[0..1] (host memory).  Results of the trace itself are unchanged. */
int r3dg_bvh_trace_visits(void* stream, int num_gaussians, const void* d_records, uint64_t* node_and_leaf_steps);

/* trace_bvh (bvh/include/bvh.h:8-12, bvh/src/trace.cu:8-192; no caller in the reference's Python): per-ray hit lists.
 *   r3dg_bvh_trace_count: num_contributes[r] = number of leaves in the <=4-leaf subtrees ray r reaches (trace.cu:21-58);
 *   (caller: inclusive scan of the counts into int64 offsets, allocation of the n = last offset entries)
 *   r3dg_bvh_trace_fill: the entries of every ray in traversal order: key = ray << 32 | bits(t), point id (-1 = rejected,
 *     t = 1e6), position = o + t d, ray id (trace.cu:88-166);
 *   (caller: stable sort by key -- r3dg_sort_pairs -- and the gather of point_list / position_list, trace.cu:171-175). */
int r3dg_bvh_trace_count(void* stream, int64_t num_rays, const int32_t* d_nodes, const float* d_aabbs,
                         const float* d_rays_o, const float* d_rays_d, int32_t* d_num_contributes,
                         int32_t* d_stack_overflow);
int r3dg_bvh_trace_fill(void* stream, int64_t num_rays, const int32_t* d_nodes, const float* d_aabbs,
                        const float* d_rays_o, const float* d_rays_d, const float* d_means3D,
                        const int32_t* d_num_contributes, const int64_t* d_offsets_inclusive, uint64_t* d_keys,
                        int32_t* d_point_list, float* d_position_list, int32_t* d_ray_id_list);

/* Stable ascending radix sort of (u64 key, u32 value) pairs on key bits [0,end_bit) -- the semantics of
 * cub::DeviceRadixSort::SortPairs as used at rasterizer_impl.cu:313-318. Exposed for tests/benchmarks.
 * d_temp must hold r3dg_sort_temp_bytes(n) bytes. Inputs are clobbered (used as ping-pong space). */
size_t r3dg_sort_temp_bytes(int64_t n);
int r3dg_sort_pairs(void* stream, int64_t n, uint64_t* d_keys_in, uint32_t* d_vals_in, uint64_t* d_keys_out,
                    uint32_t* d_vals_out, int end_bit, void* d_temp);

/* Tuning / experiment knobs (NOT part of the drop-in surface; defaults are the measured best, results never depend on them
 * beyond the order of float atomics).  r3dg_set_option returns R3DG_EINVAL for an unknown option or a value out of range. */
enum r3dg_option {
    R3DG_OPT_TILE_ORDER = 0,            /* block -> tile map: 1 longest tile list first (default), 0 natural order */
    R3DG_OPT_CULL,                      /* 1 = conservative per-block cull of the staged entries in the tile kernels (default), 0 = every
                                         * entry is evaluated; identical results */
    R3DG_OPT_TILE_BINNING,              /* instance ordering: 2 = direct tile binning + per-tile sort (default), 1 = radix partition by
                                         * tile + per-tile sort, 0 = the reference's one global radix sort; identical lists either way */
    R3DG_OPT_BINNING_BLOCK_K,           /* direct binning: Gaussians per workgroup / 1024 (1..4, default 2) */
    R3DG_OPT_STAGE_SH_ROWS,             /* per-Gaussian kernels move SH / dL_dsh rows through LDS (1, default) or walk them in HBM (0) */
    R3DG_OPT_SHADE_FWD_BLOCKS_PER_CU,   /* persistent workgroups per CU of the general shading forward; 0 = as many as fit (default) */
    R3DG_OPT_TRACE_FORMULATION,         /* visibility trace: 4 = phase-separated persistent waves (default), 3 = persistent waves,
                                         * 2 = packed records, 1 = wave-cooperative packets, 0 = one fixed ray per thread */
    R3DG_OPT_TRACE_REFILL,              /* idle lanes at which a persistent trace wave refills (default 16) */
    R3DG_OPT_TRACE_NODE_WEIGHT,         /* vote weights of the phased trace */
    R3DG_OPT_TRACE_LEAF_WEIGHT,
    R3DG_OPT_RESERVE_CUS,               /* CUs the persistent kernels (shading, trace, long-tile sort) leave free for a collective running
                                         * beside them (default 0; the data-parallel iteration sets it) */
    R3DG_OPT_TRACE_COUNT_VISITS,        /* 1 = the phased trace counts node and leaf steps (measurement: r3dg_bvh_trace_visits); default 0 */
    R3DG_OPT_BWD_LEAN,                  /* 1 (default) = the tile backward takes its lean instances (no depth slot) when the caller passed no
                                         * depth gradient; 0 = never (A/B and parity tests: same gradients up to rounding order) */
    R3DG_OPT_COUNT
};
int r3dg_set_option(int option, int value);       /* the PROCESS default */
int r3dg_get_option(int option, int* value);      /* what a launch issued by the calling thread would see right now */
/* Option contexts -- settings that belong to an OBJECT (a training step, a renderer, a tracer), not to the process.  A context
 * overrides the process defaults for the options set on it; r3dg_context_make_current installs it for the CALLING THREAD (every
 * entry point of this library launches on its caller's thread and reads its knobs at launch time), NULL removes it; *previous
 * (may be NULL) receives the context that was current, so calls nest.  Two objects with different settings in one process, or two
 * threads, never see each other's values -- r3dg_set_option alone is process-global mutable state. */
void* r3dg_context_create(void);
void r3dg_context_destroy(void* ctx);
int r3dg_context_set_option(void* ctx, int option, int value);
int r3dg_context_make_current(void* ctx, void** previous);
/* r3dg_selftest_transpose_reduce: one wave reduces d_in[64][N] -> d_out[64] (+ channel / owner maps) with the transposing
 * DPP / permlane reduction (dpp != 0) or the __shfl_xor one. */
int r3dg_selftest_transpose_reduce(void* stream, int N, int dpp, const float* d_in, float* d_out, int* d_chan,
                                   int* d_owner);

/* Per-stage kernel timing with HIP events recorded on the launch stream (used by bench.py for the roofline
 * numbers).  Stages: see r3dg_profile_stage_name(0..r3dg_profile_num_stages()-1). */
int r3dg_profile_enable(int on);
int r3dg_profile_pause(int paused);   /* suspend (1) / resume (0) recording without clearing */
int r3dg_profile_num_stages(void);
const char* r3dg_profile_stage_name(int stage);
int r3dg_profile_read(double* ms_out, int* count_out);

#ifdef __cplusplus
}
#endif
#endif /* R3DG_HIP_H */
