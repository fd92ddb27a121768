// Per-tile alpha compositing forward (K8) + surface-xyz / pseudo-normal (K9, K10) for gfx950.
// Reference semantics: renderCUDA forward.cu:263-395, renderSurfaceXYZCUDA :398-425, renderPseudoNormalCUDA :427-491.
//
// CDNA4 formulation (not the reference's 256-thread / 1-pixel-per-thread CUDA block): see render_forward_wave_kernel below --
//   * one wave per 8x8 pixel block, walking its tile's sorted list on its own (no workgroup barrier);
//   * per round of 64 entries ONE gather per lane of the 64-byte splat record; a conservative per-block cull (exact edge
//     minima of the conic quadratic over the 8x8 box) decides in registers which entries are staged at all -- entries that
//     provably stay below alpha = 1/255 on all 64 pixels cost a record read and nothing else, results are bit-identical;
//   * survivors are compacted into LDS with their blend payload (rgb + S features, zero-padded to SPAD); inner-loop LDS
//     reads are wave-uniform (broadcast) ds_read_b128; the rounds are software-pipelined (index two rounds ahead, records
//     one, the next round's cull before the current blend so that its survivors' rows load under it);
//   * S is a template parameter rounded up to a multiple of 4 (SPAD): accumulators live in VGPRs, loops unroll;
//   * per-Gaussian `weights` are summed over the wave before ONE atomic per wave
//     (the reference issues one atomic per contributing pixel, forward.cu:374);
//   * early-out: a wave stops walking when all its pixels are done (64-bit ballot).
#include "common.hpp"
#include "pseudo_normal.hpp"

namespace r3dg {

__device__ __forceinline__ float fast_exp(float x)
{
    // v_exp_f32 is 2^x: exp(x) = 2^(x*log2(e)); |x| < ~6 wherever the result matters (alpha >= 1/255)
    return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
}

// One wave = one workgroup = one 8x8 pixel block of a 16x16 tile (round 3).  Rounds 1-2 ran four waves per tile over a shared
// staging buffer: 256 entries staged by all, two workgroup barriers per round, every wave waiting for the one whose block has the
// most candidates -- PMC: the VALU the busiest unit at 54 %, the waves half their life in s_waitcnt, 2.8 resident per SIMD
// (DESIGN.md section 6).  Here every wave walks the tile's sorted list 64 entries at a time by itself, culls every entry
// against its own box while the records are still in registers and stages only the survivors (compacted, with their colour /
// feature rows) in its private LDS (7 KB at S = 16) -- no barrier anywhere, the blocks of a tile drift apart freely, 4x as many
// workgroups for the dispatcher to balance.  Price: every block reads the 32 geometry bytes of every entry of its tile (4x; the
// four blocks of a tile are placed on ONE XCD so the repeats are L2 hits) and the cull arithmetic is not shared.  Measured
// (300k Gaussians, 800x800, S = 16): 0.252 -> 0.193 ms inside the iteration, 0.290 -> 0.254 ms alone; identical outputs
// (same arithmetic per (pixel, entry) in the same order).
// One Gaussian's feature row, padded to SPAD = 4 * ceil(S / 4) floats.  ROW4 (a template parameter, not a test of S: any branch
// around the loads, uniform or not, ends in a merge where the compiler copies the loaded registers and therefore WAITS for them --
// under `if (4 * q < S)` that was SPAD / 4 memory round trips in a row, where one was meant to run under the previous round's
// arithmetic): S == SPAD, every float4 exists and is loaded unconditionally.
// (written out at both sites: returned by value from a helper, a row of more than 16 floats stays in scratch memory)
template <int SPAD, int U, bool ROW4>
__global__ void __launch_bounds__(64)
render_forward_wave_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int S, int W, int H,
                           int tiles_x, int num_tiles, int cull, const uint32_t* __restrict__ tile_order,
                           const float4* __restrict__ splat, const float* __restrict__ features, float* __restrict__ final_T,
                           uint32_t* __restrict__ n_contrib, const float* __restrict__ bg_color, float* __restrict__ out_color,
                           float* __restrict__ out_opacity, float* __restrict__ out_depth, float* __restrict__ out_feature,
                           float* __restrict__ out_weights)
{
    constexpr int PAY = 4 + SPAD;       // r, g, b, (position of the entry in its round), features[SPAD]
    // workgroup b runs on XCD b % 8: the four blocks of a tile get workgroups 8 apart (same XCD, dispatched together)
    const int xcd = (int)(blockIdx.x & 7u), j = (int)(blockIdx.x >> 3), sub = j & 3, rank = (j >> 2) * 8 + xcd;
    if (rank >= num_tiles) return;
    const int tile = tile_order != nullptr ? (int)tile_order[rank] : rank;
    const int tile_x = tile % tiles_x, tile_y = tile / tiles_x;

    __shared__ float4 s_geo0[64];                 // mean.x, mean.y, conic.x, conic.y
    __shared__ float4 s_geo1[64];                 // conic.z, opacity, depth, id bits
    __shared__ __attribute__((aligned(16))) float s_pay[64 * PAY];

    const int lane = threadIdx.x;
    const int bx = 8 * (sub & 1), by = 8 * (sub >> 1);
    const int px = tile_x * R3DG_TILE_X + bx + (lane & 7);
    const int py = tile_y * R3DG_TILE_Y + by + (lane >> 3);
    const float pxf = (float)px, pyf = (float)py;
    const float x0 = (float)(tile_x * R3DG_TILE_X + bx), y0 = (float)(tile_y * R3DG_TILE_Y + by);

    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    const bool inside = px < W && py < H;
    bool done = !inside;
    float T = 1.0f, C[3] = {0.f, 0.f, 0.f}, F[SPAD > 0 ? SPAD : 1], Dp = 0.f, Op = 0.f;
    uint32_t last = 0;
#pragma unroll
    for (int ch = 0; ch < SPAD; ch++) F[ch] = 0.f;

    // Software pipeline over the rounds of 64 entries (all global latency sits behind the blend of the round before):
    //   index of round r+2 and records of round r+1 are in flight / in registers while round r is blended;
    //   the cull of round r+1 runs BEFORE the blend of round r, so that the colour / feature rows of its survivors load under it;
    //   they are written to LDS (compacted) after the blend of round r.
    auto load_index = [&](int base_) -> uint32_t {
        return base_ + lane < n ? point_list[range.x + base_ + lane] : 0u;
    };
    uint32_t g_cur = load_index(0);                      // Gaussian of this lane's entry, round being staged
    float4 r0 = splat[4 * (size_t)g_cur], r1 = splat[4 * (size_t)g_cur + 1];
    uint32_t g_nxt = load_index(64);
    // stage round 0
    int ncand;
    {
        const bool cand = lane < n && (cull == 0 || splat_may_touch(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, x0, x0 + 7.f, y0, y0 + 7.f));
        const unsigned long long m = __ballot(cand);
        ncand = __popcll(m);
        if (cand) {
            const int slot = __popcll(m & ((1ull << lane) - 1ull));
            const float4 r2 = splat[4 * (size_t)g_cur + 2];
            s_geo0[slot] = r0;
            s_geo1[slot] = make_float4(r1.x, r1.y, r1.z, __uint_as_float(g_cur));
            float* pay = s_pay + slot * PAY;
            *reinterpret_cast<float4*>(pay) = make_float4(r2.x, r2.y, r2.z, __uint_as_float((uint32_t)lane));
            if constexpr (SPAD > 0) {
                const float* f = features + (size_t)g_cur * S;
                float4 fv[SPAD / 4];
#pragma unroll
                for (int q = 0; q < SPAD / 4; q++) {
                    if constexpr (ROW4) {
                        const float4 t = *reinterpret_cast<const float4*>(f + 4 * q);
                        fv[q] = make_float4(t.x, t.y, t.z, t.w);      // (component-wise: the array stays in registers)
                    }
                    else          // clamped addresses, no selects: the padding channels are blended but never written
                        fv[q] = make_float4(f[min(4 * q, S - 1)], f[min(4 * q + 1, S - 1)], f[min(4 * q + 2, S - 1)],
                                            f[min(4 * q + 3, S - 1)]);
                }
#pragma unroll
                for (int q = 0; q < SPAD / 4; q++) *reinterpret_cast<float4*>(pay + 4 + 4 * q) = fv[q];
            }
        }
    }
    // records of round 1 (its index has been requested above)
    g_cur = g_nxt;
    r0 = splat[4 * (size_t)g_cur];
    r1 = splat[4 * (size_t)g_cur + 1];
    g_nxt = load_index(128);
    __builtin_amdgcn_wave_barrier();

    for (int base = 0; base < n; base += 64) {
        if (__ballot(!done) == 0ull) break;                     // this block's pixels are all finished
        // ---- cull of the NEXT round (records in registers), its survivors' rows requested now ----
        const bool more = base + 64 < n;
        const bool cand1 = more && base + 64 + lane < n &&
                           (cull == 0 || splat_may_touch(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, x0, x0 + 7.f, y0, y0 + 7.f));
        const unsigned long long m1 = __ballot(cand1);
        const uint32_t g1n = g_cur;
        const float4 n0 = r0, n1 = r1;
        // ---- records of the round after next, index of the one after that: BEFORE the survivors' rows are requested.  The index
        // (g_nxt) is still in flight here; waited for behind the branch-skippable loads below, that wait would be vmcnt(0) -- the
        // two paths into it carry different numbers of loads -- and drain the very prefetch that is meant to run under the round ----
        g_cur = g_nxt;
        if (base + 128 < n) {
            r0 = splat[4 * (size_t)g_cur];
            r1 = splat[4 * (size_t)g_cur + 1];
            g_nxt = load_index(base + 192);
        }
        float4 n2 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 nf[SPAD > 0 ? SPAD / 4 : 1];
        if (cand1) {
            n2 = splat[4 * (size_t)g1n + 2];
            if constexpr (SPAD > 0) {
                const float* f = features + (size_t)g1n * S;
#pragma unroll
                for (int q = 0; q < SPAD / 4; q++) {
                    if constexpr (ROW4) {
                        const float4 t = *reinterpret_cast<const float4*>(f + 4 * q);
                        nf[q] = make_float4(t.x, t.y, t.z, t.w);      // (component-wise: the array stays in registers)
                    }
                    else          // clamped addresses, no selects: the padding channels are blended but never written
                        nf[q] = make_float4(f[min(4 * q, S - 1)], f[min(4 * q + 1, S - 1)], f[min(4 * q + 2, S - 1)],
                                            f[min(4 * q + 3, S - 1)]);
                }
            }
        }

        for (int k0 = 0; k0 < ncand; k0 += U) {
            if (__ballot(!done) == 0ull) break;
            float4 g0[U], g1[U];
            float alpha[U];
            bool valid[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                valid[u] = k0 + u < ncand;
                const int jj = valid[u] ? k0 + u : k0;      // tail: re-read, ignored below
                g0[u] = s_geo0[jj];
                g1[u] = s_geo1[jj];
            }
            bool any_alpha = false;
#pragma unroll
            for (int u = 0; u < U; u++) {
                const float dx = g0[u].x - pxf, dy = g0[u].y - pyf;
                const float power = -0.5f * (g0[u].z * dx * dx + g1[u].x * dy * dy) - g0[u].w * dx * dy;
                float a = fminf(0.99f, g1[u].y * fast_exp(power));
                if (power > 0.0f || a < 1.0f / 255.0f || !valid[u]) a = 0.f;
                alpha[u] = a;
                any_alpha = any_alpha || (a != 0.f && !done);
            }
            if (__ballot(any_alpha) == 0ull) continue;
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (!valid[u]) break;
                const float test_T = T * (1.f - alpha[u]);
                const bool hit = !done && alpha[u] != 0.f;
                const bool blend = hit && !(test_T < 0.0001f);
                done = done || (hit && !blend);
                const float w = blend ? alpha[u] * T : 0.f;
                T = blend ? test_T : T;
                if (__ballot(blend) == 0ull) continue;      // nobody in this block blends this Gaussian
                const float* pay = s_pay + (k0 + u) * PAY;
                const float4 c4 = *reinterpret_cast<const float4*>(pay);
                last = blend ? (uint32_t)base + __float_as_uint(c4.w) + 1u : last;
                C[0] += c4.x * w;
                C[1] += c4.y * w;
                C[2] += c4.z * w;
                Dp += g1[u].z * w;
                Op += w;
#pragma unroll
                for (int q = 0; q < SPAD / 4; q++) {
                    const float4 f4 = *reinterpret_cast<const float4*>(pay + 4 + 4 * q);
                    F[4 * q + 0] += f4.x * w;
                    F[4 * q + 1] += f4.y * w;
                    F[4 * q + 2] += f4.z * w;
                    F[4 * q + 3] += f4.w * w;
                }
                if (out_weights != nullptr) {
                    const float wtot = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_sum_to_lane63(w)), 63));
                    if (lane == 0) atomicAdd(&out_weights[__float_as_uint(g1[u].w)], wtot);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();          // (reads of this round before the next round's staging writes)
        // ---- stage the next round: its survivors, compacted ----
        ncand = __popcll(m1);
        if (cand1) {
            const int slot = __popcll(m1 & ((1ull << lane) - 1ull));
            s_geo0[slot] = n0;
            s_geo1[slot] = make_float4(n1.x, n1.y, n1.z, __uint_as_float(g1n));
            float* pay = s_pay + slot * PAY;
            *reinterpret_cast<float4*>(pay) = make_float4(n2.x, n2.y, n2.z, __uint_as_float((uint32_t)lane));
            if constexpr (SPAD > 0) {
#pragma unroll
                for (int q = 0; q < SPAD / 4; q++) *reinterpret_cast<float4*>(pay + 4 + 4 * q) = nf[q];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    if (inside) {
        const size_t HW = (size_t)H * W, pix = (size_t)py * W + px;
        final_T[pix] = T;
        n_contrib[pix] = last;
        out_color[pix] = C[0] + T * bg_color[0];
        out_color[HW + pix] = C[1] + T * bg_color[1];
        out_color[2 * HW + pix] = C[2] + T * bg_color[2];
#pragma unroll
        for (int ch = 0; ch < SPAD; ch++)
            if (ch < S) out_feature[(size_t)ch * HW + pix] = F[ch];
        out_depth[pix] = Dp;
        out_opacity[pix] = Op;
    }
}

// K9 + K10 in one launch (pseudo_normal.hpp).
__global__ void __launch_bounds__(256)
pseudo_normal_kernel(int W, int H, float focal_x, float focal_y, float cx, float cy, const float* __restrict__ vm,
                     const float* __restrict__ opacities, const float* __restrict__ depths, float* __restrict__ normals,
                     float* __restrict__ surface_xyz)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    pseudo_normal_pixel(x, y, W, H, focal_x, focal_y, cx, cy, vm, opacities, depths, normals, surface_xyz);
}

// ---- launchers ------------------------------------------------------------------------------------------
int g_cull = 1;         // conservative per-block cull of the staged entries (results do not depend on it): R3DG_OPT_CULL

// `splat`: the packed per-Gaussian records of preprocess_kernel (GeometryLayout::splat, 64-byte stride)
void launch_render_forward(hipStream_t s, int W, int H, int S, const uint32_t* tile_order, const uint32_t* ranges,
                           const uint32_t* point_list, const float* splat, const float* features, float* final_T,
                           uint32_t* n_contrib, const float* bg, float* out_color, float* out_opacity, float* out_depth,
                           float* out_feature, float* out_weights)
{
    const int tiles_x = (W + R3DG_TILE_X - 1) / R3DG_TILE_X, tiles_y = (H + R3DG_TILE_Y - 1) / R3DG_TILE_Y;
    const int T = tiles_x * tiles_y;
    const int grid = ((T + 7) / 8) * 8 * 4;       // 4 single-wave workgroups per tile, tile ranks padded to a multiple of 8
#define R3DG_FWD(SP)                                                                                                  \
    do {                                                                                                               \
        if ((S & 3) == 0)                                                                                              \
            render_forward_wave_kernel<SP, 4, true><<<grid, 64, 0, s>>>(                                               \
                (const uint2*)ranges, point_list, S, W, H, tiles_x, T, opt(R3DG_OPT_CULL), tile_order, (const float4*)splat,     \
                features, final_T, n_contrib, bg, out_color, out_opacity, out_depth, out_feature, out_weights);        \
        else                                                                                                           \
            render_forward_wave_kernel<SP, 4, false><<<grid, 64, 0, s>>>(                                              \
                (const uint2*)ranges, point_list, S, W, H, tiles_x, T, opt(R3DG_OPT_CULL), tile_order, (const float4*)splat,     \
                features, final_T, n_contrib, bg, out_color, out_opacity, out_depth, out_feature, out_weights);        \
    } while (0)
    switch ((S + 3) / 4) {
        case 0: R3DG_FWD(0); break;
        case 1: R3DG_FWD(4); break;
        case 2: R3DG_FWD(8); break;
        case 3: R3DG_FWD(12); break;
        case 4: R3DG_FWD(16); break;
        case 5: R3DG_FWD(20); break;
        case 6: R3DG_FWD(24); break;
        case 7: R3DG_FWD(28); break;
        case 8: R3DG_FWD(32); break;
        default: R3DG_FWD(36); break;
    }
#undef R3DG_FWD
}

void launch_pseudo_normal(hipStream_t s, int W, int H, const float* vm, float focal_x, float focal_y, float cx,
                          float cy, const float* opacities, const float* depths, float* normals, float* surface_xyz,
                          bool debug)
{
    dim3 grid((W + 63) / 64, (H + 3) / 4);
    pseudo_normal_kernel<<<grid, 256, 0, s>>>(W, H, focal_x, focal_y, cx, cy, vm, opacities, depths, normals, surface_xyz);
    check_launch(s, debug, "pseudo_normal_kernel");
}

}  // namespace r3dg
