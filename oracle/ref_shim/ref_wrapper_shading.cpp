// TEST INFRASTRUCTURE ONLY -- C wrapper around the REAL reference r3dg-rasterization/render_equation.cu (compiled
// unmodified from /root/reference for gfx950, with the torch headers of this image, by oracle/build_ref.py into
// oracle/_ref/libr3dg_reference_shading.so).  The file's raw-pointer launchers are called directly:
//   render_equation_forward_cuda          (render_equation.cu:668-688)  -> kernel :555-666
//   render_equation_forward_complex_cuda  (render_equation.cu:192-220)  -> kernel :55-190
//   render_equation_backward_cuda         (render_equation.cu:465-495)  -> kernel :280-463
// i.e. the same code path RenderEquation{Forward,Forward_complex,Backward}CUDA (:223-278, :497-553, :691-730) take after
// their tensor allocation; the random-angle table of the training variant is handed in by the test (the torch face
// draws it with torch::rand, :711) so both sides see the same numbers.  Never linked into libr3dg_hip.so.
#include <hip/hip_runtime.h>
#include <glm/glm.hpp>

void render_equation_forward_cuda(const int P, const int S_incident, const int S_direct, const int S_vis,
                                  const glm::vec3* base_color, const float* roughness, const float* metallic,
                                  const glm::vec3* normals, const glm::vec3* viewdirs, const glm::vec3* incidents_shs,
                                  const glm::vec3* direct_shs, const float* visibility_shs, const int sample_num,
                                  const bool is_training, const float* rand_float, glm::vec3* incident_dirs,
                                  glm::vec3* out_pbr, glm::vec3* out_diffuse_light);

void render_equation_forward_complex_cuda(const int P, const int S_incident, const int S_direct, const int S_vis,
                                          const glm::vec3* base_color, const float* roughness, const float* metallic,
                                          const glm::vec3* normals, const glm::vec3* viewdirs,
                                          const glm::vec3* incidents_shs, const glm::vec3* direct_shs,
                                          const float* visibility_shs, const int sample_num, glm::vec3* incident_dirs,
                                          glm::vec3* out_pbr, glm::vec3* incident_lights,
                                          glm::vec3* local_incident_lights, glm::vec3* global_incident_lights,
                                          float* incident_visibility, glm::vec3* diffuse_light,
                                          glm::vec3* local_diffuse_light, float* accum, glm::vec3* rgb_d,
                                          glm::vec3* rgb_s);

void render_equation_backward_cuda(const int P, const int S_incident, const int S_direct, const int S_vis,
                                   const glm::vec3* base_color, const float* roughness, const float* metallic,
                                   const glm::vec3* normals, const glm::vec3* viewdirs, const glm::vec3* incidents_shs,
                                   const glm::vec3* direct_shs, const float* visibility_shs, const int sample_num,
                                   const glm::vec3* incident_dirs, const glm::vec3* dL_dpbrs,
                                   const glm::vec3* dL_ddiffuse_light, glm::vec3* dL_dbase_color, float* dL_droughness,
                                   float* dL_dmetallic, glm::vec3* dL_dnormals, glm::vec3* dL_dviewdirs,
                                   glm::vec3* dL_dincidents_shs, glm::vec3* dL_ddirect_shs, float* dL_dvisibility_shs);

#define V3(p) ((glm::vec3*)(p))
#define CV3(p) ((const glm::vec3*)(p))

extern "C" {

void ref_render_equation_forward(int P, int Si, int Sd, int Sv, const float* base_color, const float* roughness,
                                 const float* metallic, const float* normals, const float* viewdirs,
                                 const float* incidents_shs, const float* direct_shs, const float* visibility_shs,
                                 int sample_num, int is_training, const float* rand_float, float* incident_dirs,
                                 float* out_pbr, float* out_diffuse_light)
{
    render_equation_forward_cuda(P, Si, Sd, Sv, CV3(base_color), roughness, metallic, CV3(normals), CV3(viewdirs),
                                 CV3(incidents_shs), CV3(direct_shs), visibility_shs, sample_num, is_training != 0,
                                 rand_float, V3(incident_dirs), V3(out_pbr), V3(out_diffuse_light));
    (void)hipDeviceSynchronize();
}

void ref_render_equation_forward_complex(int P, int Si, int Sd, int Sv, const float* base_color, const float* roughness,
                                         const float* metallic, const float* normals, const float* viewdirs,
                                         const float* incidents_shs, const float* direct_shs,
                                         const float* visibility_shs, int sample_num, float* incident_dirs,
                                         float* out_pbr, float* incident_lights, float* local_incident_lights,
                                         float* global_incident_lights, float* incident_visibility,
                                         float* diffuse_light, float* local_diffuse_light, float* accum, float* rgb_d,
                                         float* rgb_s)
{
    render_equation_forward_complex_cuda(P, Si, Sd, Sv, CV3(base_color), roughness, metallic, CV3(normals),
                                         CV3(viewdirs), CV3(incidents_shs), CV3(direct_shs), visibility_shs, sample_num,
                                         V3(incident_dirs), V3(out_pbr), V3(incident_lights),
                                         V3(local_incident_lights), V3(global_incident_lights), incident_visibility,
                                         V3(diffuse_light), V3(local_diffuse_light), accum, V3(rgb_d), V3(rgb_s));
    (void)hipDeviceSynchronize();
}

void ref_render_equation_backward(int P, int Si, int Sd, int Sv, const float* base_color, const float* roughness,
                                  const float* metallic, const float* normals, const float* viewdirs,
                                  const float* incidents_shs, const float* direct_shs, const float* visibility_shs,
                                  int sample_num, const float* incident_dirs, const float* dL_dpbrs,
                                  const float* dL_ddiffuse_light, float* dL_dbase_color, float* dL_droughness,
                                  float* dL_dmetallic, float* dL_dnormals, float* dL_dviewdirs,
                                  float* dL_dincidents_shs, float* dL_ddirect_shs, float* dL_dvisibility_shs)
{
    render_equation_backward_cuda(P, Si, Sd, Sv, CV3(base_color), roughness, metallic, CV3(normals), CV3(viewdirs),
                                  CV3(incidents_shs), CV3(direct_shs), visibility_shs, sample_num, CV3(incident_dirs),
                                  CV3(dL_dpbrs), CV3(dL_ddiffuse_light), V3(dL_dbase_color), dL_droughness,
                                  dL_dmetallic, V3(dL_dnormals), V3(dL_dviewdirs), V3(dL_dincidents_shs),
                                  V3(dL_ddirect_shs), dL_dvisibility_shs);
    (void)hipDeviceSynchronize();
}

}  // extern "C"
