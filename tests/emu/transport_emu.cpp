// TEST INFRASTRUCTURE ONLY: the relight transport-cache kernels (csrc/shading_math.hpp, csrc/shading_transport.hpp -- the
// very files shading.hip compiles for gfx950) run through the lock-step CPU emulation of hip_emu.hpp.
//   g++ -std=c++20 -O1 -pthread -shared -fPIC -I relightable3dgaussian_amd/csrc tests/emu/transport_emu.cpp -o libtransport_emu.so
#include "hip_emu.hpp"
#include "shading_math.hpp"
#include "shading_transport.hpp"

extern "C" {

void emu_shade_build_transport(int P, int K, int M, const float* normals, const float* incidents, const float* visibility,
                               const float* dirs, const float* areas, float uniform_area, float* radiance_inout,
                               float* consts)
{
    emu_launch(r3dg::shade_build_transport_kernel, (unsigned)((P + r3dg::TR_WAVES - 1) / r3dg::TR_WAVES),
               64u * r3dg::TR_WAVES, P, K, M, normals, incidents, visibility, dirs, areas, uniform_area, radiance_inout,
               consts);
}

void emu_shade_forward_transport(int P, int K, const float* base_color, const float* roughness, const float* normals,
                                 const float* viewdirs, const float* transport, const float* consts,
                                 const float* zsamples, const float* dirs, float* out)
{
    emu_launch(r3dg::shade_forward_transport_kernel, (unsigned)((P + r3dg::TR_WAVES - 1) / r3dg::TR_WAVES),
               64u * r3dg::TR_WAVES, P, K, base_color, roughness, normals, viewdirs, transport, consts, zsamples, dirs, out);
}

}
