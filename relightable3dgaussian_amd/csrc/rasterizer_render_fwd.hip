// Per-tile alpha compositing forward (K8) + surface-xyz / pseudo-normal (K9, K10) for gfx950.
// Reference semantics: renderCUDA forward.cu:263-395, renderSurfaceXYZCUDA :398-425, renderPseudoNormalCUDA :427-491.
//
// CDNA4 formulation (not the reference's 256-thread / 1-pixel-per-thread CUDA block):
//   * one 16x16 tile = 256/PPL threads; each lane owns PPL pixels ("slots") in the same column, 4 rows apart
//     inside its wave's 4*PPL-row band, so per-Gaussian LDS reads and the dx terms are shared by PPL pixels and
//     a wave-uniform branch skips whole 4x16 sub-bands the Gaussian does not touch;
//   * each round the block stages 256/PPL sorted Gaussians in LDS -- geometry (xy, conic, opacity, depth, id)
//     AND the blend payload (rgb + S features, zero-padded to SPAD) -- with one gather per thread, so the inner
//     loop never touches global memory (the reference re-reads colours/features/depths per pixel, forward.cu:364-370);
//     inner-loop LDS reads are wave-uniform (broadcast) ds_read_b128;
//   * S is a template parameter rounded up to a multiple of 4 (SPAD): accumulators live in VGPRs, loops unroll;
//   * per-Gaussian `weights` are summed over the wave (64 lanes x PPL pixels) before ONE atomic per wave
//     (the reference issues one atomic per contributing pixel, forward.cu:374);
//   * early-out: a wave stops walking the batch when all its pixels are done (64-bit ballot); the block stops
//     fetching when all waves are done (__syncthreads_and).
#include "common.hpp"

namespace r3dg {

__device__ __forceinline__ float fast_exp(float x)
{
    // v_exp_f32 is 2^x: exp(x) = 2^(x*log2(e)); |x| < ~6 wherever the result matters (alpha >= 1/255)
    return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
}

template <int SPAD, int PPL, int U>
__global__ void __launch_bounds__(256 / PPL)
render_forward_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int S, int W, int H,
                      int tiles_x, int num_tiles, int xcd_chunk, int wave8, int cull, const uint32_t* __restrict__ tile_order,
                      const float4* __restrict__ splat, const float* __restrict__ features,
                      float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, const float* __restrict__ bg_color,
                      float* __restrict__ out_color, float* __restrict__ out_opacity, float* __restrict__ out_depth,
                      float* __restrict__ out_feature, float* __restrict__ out_weights)
{
    constexpr int NT = 256 / PPL;       // threads per block == Gaussians staged per round
    constexpr int PAY = 4 + SPAD;       // payload floats per Gaussian: r,g,b,(pad), features[SPAD]

    // XCD-aware tile order: hardware places block b on XCD b%8, so give each XCD a contiguous run of tiles
    // (neighbouring tiles share Gaussians -> shared lines stay in one XCD's L2).
    // ... or, when a tile_order is given, longest-tile-first (the hardware dispatches blocks in index order, so the
    // long tiles start first and the short ones fill the tail).
    int tile;
    if (tile_order != nullptr) {
        if ((int)blockIdx.x >= num_tiles) return;
        tile = (int)tile_order[blockIdx.x];
    } else {
        tile = (int)(blockIdx.x & 7u) * xcd_chunk + (int)(blockIdx.x >> 3);
        if (tile >= num_tiles) return;
    }
    const int tile_x = tile % tiles_x, tile_y = tile / tiles_x;

    __shared__ float4 s_geo0[NT];                 // mean.x, mean.y, conic.x, conic.y
    __shared__ float4 s_geo1[NT];                 // conic.z, opacity, depth, id bits
    __shared__ __attribute__((aligned(16))) float s_pay[NT * PAY];
    constexpr int NW = NT / 64;
    __shared__ unsigned long long s_cand[NW][NW];  // [pixel wave][64-entry group]: entries that may touch that wave's box

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // lane -> pixel: PPL == 1 and wave8: each wave owns a compact 8x8 block (fewer waves touched per Gaussian than
    // with 16x4 strips); otherwise 16-wide rows, PPL pixels per lane 4 rows apart
    int lx = lane & 15, ly = wave * (4 * PPL) + (lane >> 4);
    if (PPL == 1 && wave8) {
        lx = (lane & 7) + 8 * (wave & 1);
        ly = (lane >> 3) + 8 * (wave >> 1);
    }
    const int px = tile_x * R3DG_TILE_X + lx;
    const int py0 = tile_y * R3DG_TILE_Y + ly;
    const float pxf = (float)px;

    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);

    float T[PPL], C[PPL][3], F[PPL][SPAD > 0 ? SPAD : 1], Dp[PPL], Op[PPL], pyf[PPL];
    uint32_t last[PPL];
    bool done[PPL], inside[PPL];
#pragma unroll
    for (int i = 0; i < PPL; i++) {
        const int py = py0 + 4 * i;
        inside[i] = px < W && py < H;
        done[i] = !inside[i];
        pyf[i] = (float)py;
        T[i] = 1.0f; Dp[i] = 0.f; Op[i] = 0.f; last[i] = 0;
        C[i][0] = C[i][1] = C[i][2] = 0.f;
#pragma unroll
        for (int ch = 0; ch < SPAD; ch++) F[i][ch] = 0.f;
    }

    for (int base = 0; base < n; base += NT) {
        bool all_done = true;
#pragma unroll
        for (int i = 0; i < PPL; i++) all_done = all_done && done[i];
        // barrier (protects the staging buffers of the previous round) + block-wide vote
        if (__syncthreads_and(all_done)) break;

        // ---- stage one Gaussian per thread ----
        float4 my_geo = make_float4(0.f, 0.f, 0.f, 0.f);   // mean.xy, conic.x, conic.y
        float2 my_co = make_float2(0.f, 0.f);               // conic.z, opacity
        if (base + tid < n) {
            const uint32_t g = point_list[range.x + base + tid];
            // ONE 64-byte-aligned record per instance (preprocess_kernel packs xy, conic, opacity, depth and colour)
            const float4* rec = splat + 4 * (size_t)g;
            const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
            s_geo0[tid] = my_geo = r0;
            s_geo1[tid] = make_float4(r1.x, r1.y, r1.z, __uint_as_float(g));
            my_co = make_float2(r1.x, r1.y);
            float* pay = s_pay + tid * PAY;
            *reinterpret_cast<float4*>(pay) = make_float4(r2.x, r2.y, r2.z, 0.f);
            if constexpr (SPAD > 0) {
                const float* f = features + (size_t)g * S;
                if ((S & 3) == 0) {
#pragma unroll
                    for (int q = 0; q < SPAD / 4; q++) {
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (4 * q < S) v = *reinterpret_cast<const float4*>(f + 4 * q);
                        *reinterpret_cast<float4*>(pay + 4 + 4 * q) = v;
                    }
                } else {
#pragma unroll
                    for (int ch = 0; ch < SPAD; ch++) pay[4 + ch] = ch < S ? f[ch] : 0.f;
                }
            }
        }
        // per pixel-wave candidate masks: staging thread t tests its entry against the pixel box of every wave
#pragma unroll
        for (int w = 0; w < NW; w++) {
            int bx = 0, by = w * (4 * PPL), bw = 15, bh = 4 * PPL - 1;
            if (PPL == 1 && wave8) { bx = 8 * (w & 1); by = 8 * (w >> 1); bw = 7; bh = 7; }
            const float x0 = (float)(tile_x * R3DG_TILE_X + bx), y0 = (float)(tile_y * R3DG_TILE_Y + by);
            const bool c = cull == 0 || splat_may_touch(my_geo.x, my_geo.y, my_geo.z, my_geo.w, my_co.x, my_co.y, x0,
                                                          x0 + (float)bw, y0, y0 + (float)bh);
            const unsigned long long m = __ballot(c && base + tid < n);
            if (lane == 0) s_cand[w][wave] = m;
        }
        __syncthreads();

        // Walk this wave's candidate entries U at a time: the U geometry records are fetched with back-to-back LDS reads
        // and their U x PPL alphas are evaluated as independent work (ILP hides the LDS / exp latency); only the
        // short transmittance update stays serial per entry.
        bool wave_done = false;
        for (int grp = 0; grp < NW && !wave_done; grp++) {
          const unsigned long long mv = s_cand[wave][grp];
          // the mask is wave-uniform: move it to SGPRs so the bit walk below runs on the scalar unit
          unsigned long long m = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(mv >> 32)) << 32) |
                                 (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)mv);
          while (m != 0ull) {
            bool active = false;
#pragma unroll
            for (int i = 0; i < PPL; i++) active = active || !done[i];
            if (__ballot(active) == 0ull) { wave_done = true; break; }   // this wave's pixels are all finished

            float4 g0[U], g1[U];
            float alpha[U][PPL];
            int jj[U];
            bool valid[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                valid[u] = m != 0ull;
                jj[u] = valid[u] ? grp * 64 + __builtin_ctzll(m) : (u > 0 ? jj[u - 1] : 0);   // tail: re-read, ignored below
                if (valid[u]) m &= m - 1ull;
                g0[u] = s_geo0[jj[u]];
                g1[u] = s_geo1[jj[u]];
            }
            bool any_alpha = false;
#pragma unroll
            for (int u = 0; u < U; u++) {
                const float dx = g0[u].x - pxf;
#pragma unroll
                for (int i = 0; i < PPL; i++) {
                    const float dy = g0[u].y - pyf[i];
                    const float power = -0.5f * (g0[u].z * dx * dx + g1[u].x * dy * dy) - g0[u].w * dx * dy;
                    float a = fminf(0.99f, g1[u].y * fast_exp(power));
                    if (power > 0.0f || a < 1.0f / 255.0f || !valid[u]) a = 0.f;      // a == 0 marks "skip" (a real alpha is >= 1/255)
                    alpha[u][i] = a;
                    any_alpha = any_alpha || (a != 0.f && !done[i]);
                }
            }
            if (__ballot(any_alpha) == 0ull) continue;      // none of the U entries touches this wave's live pixels

#pragma unroll
            for (int u = 0; u < U; u++) {
                if (!valid[u]) break;
                float w[PPL];
                bool any_lane = false;
#pragma unroll
                for (int i = 0; i < PPL; i++) {
                    // select form (no EXEC-masked regions): a live pixel hit by this Gaussian either saturates
                    // (T would drop below 1e-4 -> done, forward.cu:349-354) or blends it with weight alpha * T
                    const float test_T = T[i] * (1.f - alpha[u][i]);
                    const bool cand = !done[i] && alpha[u][i] != 0.f;
                    const bool blend = cand && !(test_T < 0.0001f);
                    done[i] = done[i] || (cand && !blend);
                    w[i] = blend ? alpha[u][i] * T[i] : 0.f;
                    T[i] = blend ? test_T : T[i];
                    last[i] = blend ? (uint32_t)(base + jj[u] + 1) : last[i];
                    any_lane = any_lane || blend;
                }
                if (__ballot(any_lane) == 0ull) continue;   // nobody in this wave blends this Gaussian

                const float* pay = s_pay + jj[u] * PAY;
                const float4 c4 = *reinterpret_cast<const float4*>(pay);
                float wsum = 0.f;
#pragma unroll
                for (int i = 0; i < PPL; i++) {
                    if (PPL > 1 && __ballot(w[i] != 0.f) == 0ull) continue;   // this 4x16 sub-band is untouched
                    C[i][0] += c4.x * w[i];
                    C[i][1] += c4.y * w[i];
                    C[i][2] += c4.z * w[i];
                    Dp[i] += g1[u].z * w[i];
                    Op[i] += w[i];
                    wsum += w[i];
#pragma unroll
                    for (int q = 0; q < SPAD / 4; q++) {
                        const float4 f4 = *reinterpret_cast<const float4*>(pay + 4 + 4 * q);
                        F[i][4 * q + 0] += f4.x * w[i];
                        F[i][4 * q + 1] += f4.y * w[i];
                        F[i][4 * q + 2] += f4.z * w[i];
                        F[i][4 * q + 3] += f4.w * w[i];
                    }
                }
                // wave total -> SGPR -> one lane issues the atomic (a wave-uniform value keeps the compiler's
                // uniform-address atomic rewrite down to a couple of scalar instructions).  out_weights == NULL (a caller
                // that does not read the per-Gaussian blend weights -- only densification does): nothing to reduce
                if (out_weights != nullptr) {
                    const float wtot = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_sum_to_lane63(wsum)), 63));
                    if (lane == 0) atomicAdd(&out_weights[__float_as_uint(g1[u].w)], wtot);
                }
            }
          }
        }
    }

    const size_t HW = (size_t)H * W;
#pragma unroll
    for (int i = 0; i < PPL; i++) {
        if (inside[i]) {
            const size_t pix = (size_t)(py0 + 4 * i) * W + px;
            final_T[pix] = T[i];
            n_contrib[pix] = last[i];
            out_color[pix] = C[i][0] + T[i] * bg_color[0];
            out_color[HW + pix] = C[i][1] + T[i] * bg_color[1];
            out_color[2 * HW + pix] = C[i][2] + T[i] * bg_color[2];
#pragma unroll
            for (int ch = 0; ch < SPAD; ch++)
                if (ch < S) out_feature[(size_t)ch * HW + pix] = F[i][ch];
            out_depth[pix] = Dp[i];
            out_opacity[pix] = Op[i];
        }
    }
}

// ---- the same blend with DECOUPLED waves (R3DG_OPT_FWD_DECOUPLED) ----------------------------------------------------------------
// render_forward_kernel advances the four waves of a tile round by round: 256 entries staged by all, two workgroup barriers per
// round, every wave waiting for the one whose 8x8 block has the most candidates.  PMC: the VALU is the
// busiest unit at 54 %, the waves spend half their life in s_waitcnt, 2.8 resident per SIMD.  Here ONE WAVE IS ONE WORKGROUP:
// it owns an 8x8 pixel block, walks the tile's sorted list 64 entries at a time by itself, culls every entry against its own
// box while the records are still in registers and stages only the survivors (compacted, with their colour / feature rows) in
// its private 7 KB of LDS -- no barrier anywhere, rounds of different blocks of a tile drift apart freely, 4x as many
// workgroups for the dispatcher to balance.  Price: every block reads the 32 geometry bytes of every entry of its tile (4x; the
// four blocks of a tile are placed on ONE XCD so the repeats are L2 hits) and the cull arithmetic is not shared.  Same
// arithmetic per (pixel, entry) in the same order: identical outputs.
template <int SPAD, int U>
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
#pragma unroll
                for (int q = 0; q < SPAD / 4; q++) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if ((S & 3) == 0) { if (4 * q < S) v = *reinterpret_cast<const float4*>(f + 4 * q); }
                    else {
                        v.x = 4 * q < S ? f[4 * q] : 0.f; v.y = 4 * q + 1 < S ? f[4 * q + 1] : 0.f;
                        v.z = 4 * q + 2 < S ? f[4 * q + 2] : 0.f; v.w = 4 * q + 3 < S ? f[4 * q + 3] : 0.f;
                    }
                    *reinterpret_cast<float4*>(pay + 4 + 4 * q) = v;
                }
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
        float4 n2 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 nf[SPAD > 0 ? SPAD / 4 : 1];
        if (cand1) {
            n2 = splat[4 * (size_t)g1n + 2];
            if constexpr (SPAD > 0) {
                const float* f = features + (size_t)g1n * S;
#pragma unroll
                for (int q = 0; q < SPAD / 4; q++) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if ((S & 3) == 0) { if (4 * q < S) v = *reinterpret_cast<const float4*>(f + 4 * q); }
                    else {
                        v.x = 4 * q < S ? f[4 * q] : 0.f; v.y = 4 * q + 1 < S ? f[4 * q + 1] : 0.f;
                        v.z = 4 * q + 2 < S ? f[4 * q + 2] : 0.f; v.w = 4 * q + 3 < S ? f[4 * q + 3] : 0.f;
                    }
                    nf[q] = v;
                }
            }
        }
        // ---- records of the round after next, index of the one after that ----
        g_cur = g_nxt;
        if (base + 128 < n) {
            r0 = splat[4 * (size_t)g_cur];
            r1 = splat[4 * (size_t)g_cur + 1];
            g_nxt = load_index(base + 192);
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

// K9: surface point in camera space from the premultiplied depth / opacity buffers (forward.cu:398-425)
__global__ void __launch_bounds__(256)
surface_xyz_kernel(int W, int H, float focal_x, float focal_y, float cx, float cy, const float* __restrict__ opacities,
                   const float* __restrict__ depths, float* __restrict__ surface_xyz)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t HW = (size_t)H * W, id = (size_t)y * W + x;
    const float depth = depths[id] / fmaxf(opacities[id], 0.0000001f);
    surface_xyz[id] = (x - cx) / focal_x * depth;
    surface_xyz[HW + id] = (y - cy) / focal_y * depth;
    surface_xyz[2 * HW + id] = depth;
}

// K10: pseudo normal from a 3x3 edge-clamped stencil on surface_xyz (forward.cu:427-491); needs K9 complete.
__global__ void __launch_bounds__(256)
pseudo_normal_kernel(int W, int H, const float* __restrict__ vm, float* __restrict__ normals,
                     const float* __restrict__ surface_xyz)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t HW = (size_t)H * W;
    const int ym = y == 0 ? 0 : y - 1, yp = y == H - 1 ? H - 1 : y + 1;
    const int xm = x == 0 ? 0 : x - 1, xp = x == W - 1 ? W - 1 : x + 1;
    const size_t i00 = (size_t)W * ym + xm, i01 = (size_t)W * ym + x, i02 = (size_t)W * ym + xp;
    const size_t i10 = (size_t)W * y + xm, i11 = (size_t)W * y + x, i12 = (size_t)W * y + xp;
    const size_t i20 = (size_t)W * yp + xm, i21 = (size_t)W * yp + x, i22 = (size_t)W * yp + xp;
    float ga[3], gb[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float* s = surface_xyz + i * HW;
        ga[i] = -0.125f * s[i00] + 0.125f * s[i02] - 0.25f * s[i10] + 0.25f * s[i12] - 0.125f * s[i20] + 0.125f * s[i22];
        gb[i] = -0.125f * s[i00] - 0.25f * s[i01] - 0.125f * s[i02] + 0.125f * s[i20] + 0.25f * s[i21] + 0.125f * s[i22];
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

// ---- launchers ------------------------------------------------------------------------------------------
int g_fwd_wave8x8 = 1;  // 1-pixel-per-lane kernel: wave = 8x8 pixel block (1) or 16x4 strip (0); R3DG_OPT_FWD_WAVE8X8
int g_cull = 1;         // per-wave conservative sub-tile cull of staged entries (results do not depend on it)
int g_fwd_ppl = 1;   // pixels per lane; R3DG_OPT_FWD_PIXELS_PER_LANE
int g_fwd_unroll = 4;   // staged entries evaluated per inner-loop step (1 = entry-at-a-time)
int g_fwd_decoupled = 0;   // R3DG_OPT_FWD_DECOUPLED: 1 = one wave per 8x8 block walking the tile's list on its own (render_forward_wave_kernel)

template <int SPAD, int PPL>
static void launch_fwd_inst(hipStream_t s, int T, int tiles_x, const uint32_t* tile_order, const uint32_t* ranges,
                            const uint32_t* point_list, int S, int W, int H, const float* splat, const float* features,
                            float* final_T, uint32_t* n_contrib, const float* bg, float* out_color, float* out_opacity,
                            float* out_depth, float* out_feature, float* out_weights)
{
    const int chunk = (T + 7) / 8;
    if (PPL == 1 && g_fwd_decoupled) {
        // 4 single-wave workgroups per tile, tile ranks padded to a multiple of 8 (see the kernel's index mapping)
        render_forward_wave_kernel<SPAD, 4><<<chunk * 8 * 4, 64, 0, s>>>(
            (const uint2*)ranges, point_list, S, W, H, tiles_x, T, g_cull, tile_order, (const float4*)splat, features, final_T,
            n_contrib, bg, out_color, out_opacity, out_depth, out_feature, out_weights);
        return;
    }
#define R3DG_FWD_LAUNCH(U)                                                                                            \
    render_forward_kernel<SPAD, PPL, U><<<chunk * 8, 256 / PPL, 0, s>>>(                                              \
        (const uint2*)ranges, point_list, S, W, H, tiles_x, T, chunk, g_fwd_wave8x8, g_cull, tile_order,              \
        (const float4*)splat, features, final_T, n_contrib, bg, out_color, out_opacity, out_depth, out_feature,       \
        out_weights)
    if (g_fwd_unroll >= 4) R3DG_FWD_LAUNCH(4);
    else if (g_fwd_unroll >= 2) R3DG_FWD_LAUNCH(2);
    else R3DG_FWD_LAUNCH(1);
#undef R3DG_FWD_LAUNCH
}

template <int SPAD>
static void launch_fwd_ppl(int ppl, hipStream_t s, int T, int tiles_x, const uint32_t* tile_order,
                           const uint32_t* ranges, const uint32_t* point_list, int S, int W, int H, const float* splat,
                           const float* features, float* final_T, uint32_t* n_contrib, const float* bg, float* out_color,
                           float* out_opacity, float* out_depth, float* out_feature, float* out_weights)
{
#define R3DG_FWD_ARGS s, T, tiles_x, tile_order, ranges, point_list, S, W, H, splat, features, final_T, n_contrib, bg, \
                      out_color, out_opacity, out_depth, out_feature, out_weights
    if (ppl >= 4 && SPAD <= 20) launch_fwd_inst<SPAD, 4>(R3DG_FWD_ARGS);
    else if (ppl >= 2) launch_fwd_inst<SPAD, 2>(R3DG_FWD_ARGS);
    else launch_fwd_inst<SPAD, 1>(R3DG_FWD_ARGS);
}

// `splat`: the packed per-Gaussian records of preprocess_kernel (GeometryLayout::splat, 64-byte stride)
void launch_render_forward(hipStream_t s, int W, int H, int S, const uint32_t* tile_order, const uint32_t* ranges,
                           const uint32_t* point_list, const float* splat, const float* features, float* final_T,
                           uint32_t* n_contrib, const float* bg, float* out_color, float* out_opacity, float* out_depth,
                           float* out_feature, float* out_weights)
{
    const int tiles_x = (W + R3DG_TILE_X - 1) / R3DG_TILE_X, tiles_y = (H + R3DG_TILE_Y - 1) / R3DG_TILE_Y;
    const int T = tiles_x * tiles_y;
    const int ppl = g_fwd_ppl;
    switch ((S + 3) / 4) {
        case 0: launch_fwd_ppl<0>(ppl, R3DG_FWD_ARGS); break;
        case 1: launch_fwd_ppl<4>(ppl, R3DG_FWD_ARGS); break;
        case 2: launch_fwd_ppl<8>(ppl, R3DG_FWD_ARGS); break;
        case 3: launch_fwd_ppl<12>(ppl, R3DG_FWD_ARGS); break;
        case 4: launch_fwd_ppl<16>(ppl, R3DG_FWD_ARGS); break;
        case 5: launch_fwd_ppl<20>(ppl, R3DG_FWD_ARGS); break;
        case 6: launch_fwd_ppl<24>(ppl, R3DG_FWD_ARGS); break;
        case 7: launch_fwd_ppl<28>(ppl, R3DG_FWD_ARGS); break;
        case 8: launch_fwd_ppl<32>(ppl, R3DG_FWD_ARGS); break;
        default: launch_fwd_ppl<36>(ppl, R3DG_FWD_ARGS); break;
    }
#undef R3DG_FWD_ARGS
}

void launch_pseudo_normal(hipStream_t s, int W, int H, const float* vm, float focal_x, float focal_y, float cx,
                          float cy, const float* opacities, const float* depths, float* normals, float* surface_xyz,
                          bool debug)
{
    dim3 grid((W + 63) / 64, (H + 3) / 4);
    surface_xyz_kernel<<<grid, 256, 0, s>>>(W, H, focal_x, focal_y, cx, cy, opacities, depths, surface_xyz);
    check_launch(s, debug, "surface_xyz_kernel");
    pseudo_normal_kernel<<<grid, 256, 0, s>>>(W, H, vm, normals, surface_xyz);
    check_launch(s, debug, "pseudo_normal_kernel");
}

}  // namespace r3dg
