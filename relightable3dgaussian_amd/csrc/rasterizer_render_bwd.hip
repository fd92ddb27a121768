// Per-tile alpha compositing backward (K11) for gfx950.  Reference semantics: renderCUDA backward.cu:401-614.
//
// The reference issues 10+S float atomics per contributing (pixel, Gaussian) pair.  Here:
//   * same tile decomposition as the forward (256/PPL threads, PPL pixels per lane, LDS-staged batches that
//     include colour/feature/depth payload, wave-uniform broadcast reads);
//   * the back-to-front walk starts at the tile's deepest contributor (block max of n_contrib) instead of the
//     end of the tile list -- entries behind every pixel's last contributor are never fetched;
//   * the recursive "accum_rec" blend (backward.cu:538-577) is kept bit-for-bit but without the reference's
//     last_color/last_feature/last_depth copies: accum is advanced with the current (alpha, value) right after use,
//     which evaluates the same expression one iteration earlier;
//   * per-Gaussian gradients (3 colour + 3 mean2D + 3 conic + 1 opacity + S feature = 10+S values per lane, already
//     summed over the lane's PPL pixels) are reduced across the 64 lanes with a TRANSPOSING butterfly: after
//     log2(NV) exchange levels lane l holds the wave total of channel chan(l), so the whole 10+S-vector costs
//     ~NV shuffles (not 6*NV) and leaves as ONE atomic instruction with 10+S active lanes per (wave, Gaussian).
#include "common.hpp"
#include "wave_reduce.hpp"

#include <map>
#include <mutex>
#include <utility>

#ifndef R3DG_BWD_WAVES_SMALL
#define R3DG_BWD_WAVES_SMALL 4     // waves per SIMD the <= 4-channel instances are compiled for (measured: 4 -> 0.333 ms, 5 -> 0.367 with 5 spilled registers, 6 -> 0.467)
#endif

namespace r3dg {

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float fast_exp_b(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// self-test kernel: out[lane] = transpose_reduce of in[lane*N + k]; host compares against a plain sum
template <int N, bool DPP>
__global__ void transpose_reduce_selftest_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                 int* __restrict__ chan_out, int* __restrict__ owner_out)
{
    float v[N];
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = in[threadIdx.x * N + k];
    out[threadIdx.x] = transpose_reduce<N, DPP>(v);
    chan_out[threadIdx.x] = transposed_channel<N>(threadIdx.x);
    owner_out[threadIdx.x] = transposed_owner<N>(threadIdx.x) ? 1 : 0;
}

__global__ void transpose_reduce12_selftest_kernel(const float* __restrict__ in, float* __restrict__ out, int* __restrict__ chan_out,
                                                   int* __restrict__ owner_out)
{
    float v[12];
#pragma unroll
    for (int k = 0; k < 12; k++) v[k] = in[threadIdx.x * 12 + k];
    out[threadIdx.x] = transpose_reduce12(v);
    chan_out[threadIdx.x] = transposed_channel12(threadIdx.x);
    owner_out[threadIdx.x] = transposed_owner12(threadIdx.x) ? 1 : 0;
}

void launch_transpose_selftest(hipStream_t s, int N, int dpp, const float* in, float* out, int* chan, int* owner)
{
    if (N == 12) {
        transpose_reduce12_selftest_kernel<<<1, 64, 0, s>>>(in, out, chan, owner);
        return;
    }
#define R3DG_ST(NN)                                                                              \
    if (dpp) transpose_reduce_selftest_kernel<NN, true><<<1, 64, 0, s>>>(in, out, chan, owner);  \
    else transpose_reduce_selftest_kernel<NN, false><<<1, 64, 0, s>>>(in, out, chan, owner);
    if (N == 16) { R3DG_ST(16) } else if (N == 32) { R3DG_ST(32) } else { R3DG_ST(64) }
#undef R3DG_ST
}

constexpr int next_pow2(int v) { return v <= 16 ? 16 : (v <= 32 ? 32 : 64); }

// Feature channels the kernel processes: by default all S; a caller that knows which channels of dL_dout_feature are
// non-zero (a loss usually reads a few of them) passes that subset -- channels with a zero upstream gradient contribute
// neither to dL_dalpha nor to dL_dfeature, so they need no recursion, no payload and no slot in the reduction.
struct ChannelList {
    int n;               // processed channels (<= SPAD)
    int identity;        // 1: channel j is feature j and n == S
    int c[R3DG_MAX_S_BWD];
};

// One wave = one workgroup = one 8x8 pixel block (round 3; rounds 1-2 ran four waves per tile over a shared staging buffer with
// two workgroup barriers per round -- see render_forward_wave_kernel).  The wave walks the tile's list back to front from ITS OWN
// deepest contributor, culls every entry against its own box, stages only the survivors (compacted) in its private LDS, and
// software-pipelines the rounds like the forward: no workgroup barrier.
//   * only the feature channels with a non-zero upstream gradient are carried (ChannelList);
//   * per-pixel state as float2 pairs so the per-channel recursions compile to packed fp32 ops (v_pk_fma_f32 / v_pk_mul_f32 /
//     v_pk_add_f32: two channels per instruction);
//   * branch-free over the lanes: a lane that does not blend this Gaussian has alpha == 0, which leaves its recursions unchanged
//     (T *= 1, acc += 0 * d) and zeroes every gradient term;
//   * the per-Gaussian sums leave through the transposing reduction + ONE atomic instruction per (wave, Gaussian).
// Round 5 (the kernel is VALU-bound, cut instructions):
//   * GEOMETRY MOMENTS.  With m = G * dL_dG per lane, the five mean / conic gradients of backward.cu:579-601 are linear in the
//     five moments  S_x = sum m dx, S_y = sum m dy, S_xx = sum m dx^2, S_xy = sum m dx dy, S_yy = sum m dy^2:
//         dL_dmean2D.x = -W/2 (A S_x + B S_y)    dL_dmean2D.y = -H/2 (C S_y + B S_x)    dL_dconic = -1/2 (S_xx, S_xy, S_yy)
//     with the conic (A, B, C) a per-GAUSSIAN constant.  The lanes accumulate the raw moments (6 multiplications instead of the
//     ~16 of the expanded terms) into the dL_dmean2D / dL_dconic slots, and preprocess_backward_kernel -- which reads those slots
//     once per Gaussian anyway -- applies the conic (`moments_to_gradients`, rasterizer_preprocess_bwd.hip) and writes the final
//     dL_dmean2D back.  Same sums in another order of rounding.
//   * ONE LDS RECORD per staged entry (geometry 32 B + payload), one address register, immediate offsets; the NEXT entry's
//     geometry is requested before the current entry's arithmetic (the loop used to wait for three separate LDS reads per entry,
//     one of them -- the Gaussian's index for the atomic -- at the very end).
//   * LEAN instances (no depth gradient -- dL_dpixels_d == nullptr, what the stage-2 objectives pass -- and at least one padding
//     feature slot): the channel vector of the packed recursions is [r g b f0 ..] without the depth slot, one packed pair and one
//     reduction channel fewer (3 live features: 3 pairs instead of 4; up to 7 live features still reduce 16-wide).
//   * ONE GRADIENT RECORD PER GAUSSIAN.  Rounds 1-4 sent the (wave, Gaussian) sums straight to the op's five output arrays: one
//     atomic instruction, but its 10 + S lanes hit FIVE different cache lines (colours, mean2D, conic, opacity, feature row).
//     Ablation builds (tools/variants_bwd.py, profiles/r05_bwd_ablation.txt): without the atomics the kernel takes 0.253 instead
//     of 0.417 ms -- and with all lanes of an entry aimed at ONE 64-byte row per Gaussian 0.258 ms: the device-scope atomics, not
//     the VALU work, were 40 % of the launch, and nearly all of that is the number of LINES touched.  The sums now go to channel
//     `chan` of a record of NVP floats per Gaussian (library scratch, zero between calls), and render_backward_scatter_kernel
//     -- one coalesced pass per Gaussian behind the tile kernel -- writes the five output arrays from it (and zeroes the record).
// Channel vector:  !LEAN  [r g b depth f0 .. f(SPAD-1)]      LEAN  [r g b f0 .. f(SPAD-2)]
// Record channels (= reduction channels) -> array, applied by the scatter kernel:  0..2 dL_dcolors[g*3+c] | 3,4 moments S_x, S_y ->
// dL_dmean2D[g*3+{0,1}] | (!LEAN) 5 dL_dmean2D[g*3+2] | next 3: S_xx, S_xy, S_yy -> dL_dconic2D[g*4+{0,1,3}] | next: dL_dopacity[g] |
// rest: dL_dfeature[g*S+c].
template <int SPAD, bool SMALLV, bool ROW4, bool LEAN>
__global__ void __launch_bounds__(64, (SPAD <= 4 ? R3DG_BWD_WAVES_SMALL : 1))
render_backward_wave_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int S,
                            ChannelList chan_list, int W, int H, int tiles_x, int num_tiles, int cull,
                            const uint32_t* __restrict__ tile_order, const float* __restrict__ bg_color,
                            const float4* __restrict__ splat, const float* __restrict__ features,
                            const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib,
                            const float* __restrict__ dL_dpixels, const float* __restrict__ dL_dpixels_o,
                            const float* __restrict__ dL_dpixels_d, const float* __restrict__ dL_dpixels_f,
                            float* __restrict__ grad_records, int backward_geometry)
{
    static_assert(!LEAN || SPAD >= 4, "a lean instance needs a padding feature slot");
    constexpr int NC = LEAN ? 2 + SPAD : 4 + SPAD;        // channels of the packed recursions (even)
    constexpr int NF = LEAN ? SPAD - 1 : SPAD;            // feature slots among them
    constexpr int F0 = LEAN ? 3 : 4;                      // first feature channel
    constexpr int NCOL = F0;                              // colour (+ depth) channels: their part of dL_dalpha is unconditional
    constexpr int PAYF = (NC + 3) / 4 * 4;                // payload floats of a record (whole float4 rows)
    constexpr int REC = 8 + PAYF;                         // floats per LDS record: geometry 8 | payload
    constexpr int V_CONIC = LEAN ? 5 : 6;
    constexpr int V_OPAC = V_CONIC + 3;
    constexpr int V_FEAT = V_OPAC + 1;
    constexpr int NV = V_FEAT + NF;
    constexpr int NVP = SMALLV ? 16 : next_pow2(NV);
    // (SMALLV with NV > 16: the channels past 15 are padding feature slots -- the launcher guarantees it -- and are dropped)
    const int SA = chan_list.n;
    const int xcd = (int)(blockIdx.x & 7u), j = (int)(blockIdx.x >> 3), sub = j & 3, rank = (j >> 2) * 8 + xcd;
    if (rank >= num_tiles) return;
    const int tile = tile_order != nullptr ? (int)tile_order[rank] : rank;
    const int tile_x = tile % tiles_x, tile_y = tile / tiles_x;

    // record k: [mean.x mean.y conic.x conic.y | conic.z opacity front-index id | payload]; one spare record behind the 64
    // (the loop requests entry k + 1 unconditionally)
    __shared__ __attribute__((aligned(16))) float s_rec[65 * REC];

    const int lane = threadIdx.x;
    const int bx = 8 * (sub & 1), by = 8 * (sub >> 1);
    const int px = tile_x * R3DG_TILE_X + bx + (lane & 7);
    const int py = tile_y * R3DG_TILE_Y + by + (lane >> 3);
    const float pxf = (float)px, pyf = (float)py;
    const float x0 = (float)(tile_x * R3DG_TILE_X + bx), y0 = (float)(tile_y * R3DG_TILE_Y + by);
    const size_t HW = (size_t)H * W;
    const uint2 range = ranges[tile];

    // the lane that owns channel `chan` after the transposing reduction adds it to word `chan` of the Gaussian's record
    constexpr bool R12 = NV == 12;                         // exactly twelve channels: the 25-instruction reduction (wave_reduce.hpp)
    const int chan = R12 ? transposed_channel12(lane) : transposed_channel<NVP>(lane);
    const bool owner = R12 ? transposed_owner12(lane) : transposed_owner<NVP>(lane);
    float* dst_base = nullptr;
    if (owner && (chan < V_FEAT || (chan - V_FEAT < SA && chan - V_FEAT < NF))) dst_base = grad_records + chan;

    const bool inside = px < W && py < H;
    const size_t pix = (size_t)py * W + px;
    float T = inside ? final_Ts[pix] : 0.f;
    const uint32_t lastc = inside ? n_contrib[pix] : 0u;
    f2 acc2[NC / 2], dL2[NC / 2];
    float acc_o = 0.f, bgT, dLo;
    {
        float dl[NC];
        float bg_dot = 0.f;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            dl[ch] = inside ? dL_dpixels[ch * HW + pix] : 0.f;
            bg_dot += bg_color[ch] * dl[ch];
        }
        bgT = -T * bg_dot;
        if constexpr (!LEAN) dl[3] = (inside && dL_dpixels_d != nullptr) ? dL_dpixels_d[pix] : 0.f;
        dLo = inside ? dL_dpixels_o[pix] : 0.f;
#pragma unroll
        for (int ch = 0; ch < NF; ch++)
            dl[F0 + ch] = (inside && ch < SA) ? dL_dpixels_f[(size_t)chan_list.c[ch] * HW + pix] : 0.f;
#pragma unroll
        for (int q = 0; q < NC / 2; q++) {
            dL2[q] = f2{dl[2 * q], dl[2 * q + 1]};
            acc2[q] = f2{0.f, 0.f};
        }
    }
    // this block's deepest last contributor: the walk covers front indices [0, n) back to front
    uint32_t mx = lastc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
    const int n = (int)mx;

    int cidx[SPAD > 0 ? SPAD : 1];                        // (uniform: scalar registers)
#pragma unroll
    for (int jc = 0; jc < (SPAD > 0 ? SPAD : 1); jc++) cidx[jc] = chan_list.c[jc < R3DG_MAX_S_BWD ? jc : 0];
    // entry `e` of the walk (0 = deepest) is list position n - 1 - e
    auto load_index = [&](int e0) -> uint32_t {
        return e0 + lane < n ? point_list[range.x + (uint32_t)(n - 1 - (e0 + lane))] : 0u;
    };
    auto cull_ok = [&](const float4& a0, const float4& a1) -> bool {
        return cull == 0 || splat_may_touch(a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, x0, x0 + 7.f, y0, y0 + 7.f);
    };
    // No branch and no select around these loads: a branch (uniform or not) ends in a merge, a select consumes the value -- either
    // way the compiler waits for the load right there, and the survivors' rows were requested one round early precisely so that
    // they arrive under this round's arithmetic.  Channels >= SA of the padded vector read channel c[j] = 0 of the list (the
    // launcher fills the tail with 0): a finite value that meets a zero upstream gradient, i.e. contributes nothing.
    auto load_payload = [&](uint32_t g, float4& c4, float4 (&f4)[SPAD > 0 ? SPAD / 4 : 1]) {
        const float4 r2 = splat[4 * (size_t)g + 2];
        c4 = make_float4(r2.x, r2.y, r2.z, 0.f);
        if constexpr (SPAD > 0) {
            const float* f = features + (size_t)g * S;
            if constexpr (ROW4) {                     // identity list, S == SPAD: the row as float4s
#pragma unroll
                for (int q = 0; q < SPAD / 4; q++) {
                    const float4 t = *reinterpret_cast<const float4*>(f + 4 * q);
                    f4[q] = make_float4(t.x, t.y, t.z, t.w);          // (component-wise: the array stays in registers)
                }
            } else {
#pragma unroll
                for (int q = 0; q < SPAD / 4; q++)
                    f4[q] = make_float4(f[cidx[4 * q]], f[cidx[4 * q + 1]], f[cidx[4 * q + 2]], f[cidx[4 * q + 3]]);
            }
        }
    };
    auto stage = [&](bool cand, unsigned long long m, const float4& a0, const float4& a1, uint32_t g, const float4& c4,
                     const float4 (&f4)[SPAD > 0 ? SPAD / 4 : 1], int e0) {
        if (cand) {
            const int slot = __popcll(m & ((1ull << lane) - 1ull));
            float* rec = s_rec + slot * REC;
            *reinterpret_cast<float4*>(rec) = a0;
            *reinterpret_cast<float4*>(rec + 4) =
                make_float4(a1.x, a1.y, __uint_as_float((uint32_t)(n - 1 - (e0 + lane))), __uint_as_float(g));
            float pv[PAYF];
#pragma unroll
            for (int q = 0; q < PAYF; q++) pv[q] = 0.f;
            pv[0] = c4.x; pv[1] = c4.y; pv[2] = c4.z;
            if constexpr (!LEAN) pv[3] = a1.z;                                   // depth
            if constexpr (SPAD > 0) {
#pragma unroll
                for (int q = 0; q < SPAD / 4; q++) {
                    const float fv[4] = {f4[q].x, f4[q].y, f4[q].z, f4[q].w};
#pragma unroll
                    for (int c = 0; c < 4; c++)
                        if (4 * q + c < NF) pv[F0 + 4 * q + c] = fv[c];
                }
            }
#pragma unroll
            for (int q = 0; q < PAYF / 4; q++)
                *reinterpret_cast<float4*>(rec + 8 + 4 * q) = make_float4(pv[4 * q], pv[4 * q + 1], pv[4 * q + 2], pv[4 * q + 3]);
        }
    };

    // prologue: round 0 staged, records of round 1 in registers, index of round 2 requested
    uint32_t g_cur = load_index(0);
    float4 r0 = splat[4 * (size_t)g_cur], r1 = splat[4 * (size_t)g_cur + 1];
    uint32_t g_nxt = load_index(64);
    int ncand;
    {
        const bool cand = lane < n && cull_ok(r0, r1);
        const unsigned long long m = __ballot(cand);
        ncand = __popcll(m);
        float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 f4[SPAD > 0 ? SPAD / 4 : 1];
        if (cand) load_payload(g_cur, c4, f4);
        stage(cand, m, r0, r1, g_cur, c4, f4, 0);
    }
    g_cur = g_nxt;
    r0 = splat[4 * (size_t)g_cur];
    r1 = splat[4 * (size_t)g_cur + 1];
    g_nxt = load_index(128);
    __builtin_amdgcn_wave_barrier();

    for (int base = 0; base < n; base += 64) {
        // cull of the next round, its survivors' rows requested now (they load under this round's arithmetic)
        const bool cand1 = base + 64 + lane < n && cull_ok(r0, r1);
        const unsigned long long m1 = __ballot(cand1);
        const uint32_t g1n = g_cur;
        const float4 n0 = r0, n1 = r1;
        // the records of the round after next FIRST: their index (g_nxt) is still in flight, and a wait placed behind the
        // branch-skippable loads of load_payload would have to be vmcnt(0) (the two paths into it carry different numbers of
        // loads) -- it drained the survivors' rows in front of every round
        g_cur = g_nxt;
        if (base + 128 < n) {
            r0 = splat[4 * (size_t)g_cur];
            r1 = splat[4 * (size_t)g_cur + 1];
            g_nxt = load_index(base + 192);
        }
        float4 nc4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 nf4[SPAD > 0 ? SPAD / 4 : 1];
        if (cand1) load_payload(g1n, nc4, nf4);

        float4 g0 = *reinterpret_cast<const float4*>(s_rec), g1 = *reinterpret_cast<const float4*>(s_rec + 4);
        for (int k = 0; k < ncand; k++) {
            // the next entry's geometry is requested now (slot ncand <= 64 exists; what it holds is never used)
            const float* rec = s_rec + k * REC;
            const float4 g0n = *reinterpret_cast<const float4*>(rec + REC), g1n = *reinterpret_cast<const float4*>(rec + REC + 4);
            const uint32_t front = __float_as_uint(g1.z);
            const float dx = g0.x - pxf, dy = g0.y - pyf;
            const float power = -0.5f * (g0.z * dx * dx + g1.x * dy * dy) - g0.w * dx * dy;
            const float Gv0 = fast_exp_b(power);
            float al = fminf(0.99f, g1.y * Gv0);
            // reference: skip while contributor >= last_contributor, power > 0, alpha < 1/255
            if (!(front < lastc) || power > 0.0f || al < 1.0f / 255.0f) al = 0.f;
            if (__ballot(al != 0.f) != 0ull) {
                const float* pay = rec + 8;
                constexpr int NVA = NV > NVP ? NV : NVP;
                float v[NVA];
#pragma unroll
                for (int q = NV; q < NVA; q++) v[q] = 0.f;
                const bool hit = al != 0.f;
                const float rcp = __builtin_amdgcn_rcpf(1.f - al);
                T = T * rcp;
                const float wgt = al * T;
                const f2 al2 = f2{al, al}, w2 = f2{wgt, wgt};
                f2 sc = f2{0.f, 0.f}, sf = f2{0.f, 0.f};       // dL_dalpha: colour (+depth) part | feature part
#pragma unroll
                for (int q = 0; q < PAYF / 4; q++) {
                    const float4 p4 = *reinterpret_cast<const float4*>(pay + 4 * q);
                    const f2 pv2[2] = {f2{p4.x, p4.y}, f2{p4.z, p4.w}};
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const int pr = 2 * q + h;                       // pair index: channels 2 pr, 2 pr + 1
                        if (2 * pr >= NC) continue;
                        const f2 d = pv2[h] - acc2[pr];
                        const f2 t = d * dL2[pr];
                        // which part of dL_dalpha a channel belongs to is a compile-time property of its index
                        if (2 * pr + 1 < NCOL) sc += t;
                        else if (2 * pr >= NCOL) sf += t;
                        else { sc.x += t.x; sf.y += t.y; }
                        acc2[pr] += al2 * d;
                        const f2 vv = w2 * dL2[pr];
                        const float vc[2] = {vv.x, vv.y};
#pragma unroll
                        for (int c = 0; c < 2; c++) {
                            const int ch = 2 * pr + c;
                            if (ch < 3) v[ch] = vc[c];
                            else if (!LEAN && ch == 3) v[5] = vc[c];               // depth -> dL_dmean2D.z
                            else v[V_FEAT + (ch - F0)] = vc[c];
                        }
                    }
                }
                float dL_dalpha = sc.x + sc.y;
                if (backward_geometry) dL_dalpha += sf.x + sf.y;
                const float d_o = 1.0f - acc_o;
                dL_dalpha += d_o * dLo;
                acc_o += al * d_o;
                dL_dalpha *= T;
                dL_dalpha += bgT * rcp;
                // moments of m = G dL_dG (dL_dG = opacity dL_dalpha); a lane that does not blend contributes nothing (its
                // exp may have overflowed: selected away, not multiplied by zero)
                const float t9 = hit ? Gv0 * dL_dalpha : 0.f;          // dL_dopacity term (backward.cu:597)
                const float mm = g1.y * t9;
                const float mdx = mm * dx, mdy = mm * dy;
                v[3] = mdx;
                v[4] = mdy;
                v[V_CONIC] = mdx * dx;
                v[V_CONIC + 1] = mdx * dy;
                v[V_CONIC + 2] = mdy * dy;
                v[V_OPAC] = t9;
                float total;
                if constexpr (R12) {
                    float vr[12];
#pragma unroll
                    for (int q = 0; q < 12; q++) vr[q] = v[q];
                    total = transpose_reduce12(vr);
                } else {
                    float vr[NVP];
#pragma unroll
                    for (int q = 0; q < NVP; q++) vr[q] = v[q];
                    total = transpose_reduce<NVP, true>(vr);
                }
                if (dst_base != nullptr) atomicAdd(dst_base + (size_t)__float_as_uint(g1.w) * NVP, total);
            }
            g0 = g0n;
            g1 = g1n;
        }
        __builtin_amdgcn_wave_barrier();          // (reads of this round before the next round's staging writes)
        ncand = __popcll(m1);
        stage(cand1, m1, n0, n1, g1n, nc4, nf4, base + 64);
        __builtin_amdgcn_wave_barrier();
    }
}

// Record -> output arrays (see render_backward_wave_kernel).  A workgroup takes 256 consecutive Gaussians: their records come in
// as one contiguous run of float4s (and zeros go back: the scratch is zero between calls) into LDS, rows padded to NVP + 1 words;
// then each output array's slice of the workgroup -- contiguous in memory -- is written by consecutive threads (the first version,
// one thread per Gaussian with 64-byte strides between the lanes of every load and store, took 28 us for 56 MB).  Every element of
// the five arrays is written: the caller's zero fill of the formerly atomically accumulated outputs is no longer needed.
// NVP / LEAN / NF as the tile kernel that filled the records was instantiated.
template <int NVP, bool LEAN>
__global__ void __launch_bounds__(256)
render_backward_scatter_kernel(int P, int S, int NF, ChannelList chan_list, float4* __restrict__ grad_records,
                               float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic2D, float* __restrict__ dL_dopacity,
                               float* __restrict__ dL_dcolors, float* __restrict__ dL_dfeature)
{
    constexpr int V_CONIC = LEAN ? 5 : 6, V_OPAC = V_CONIC + 3, V_FEAT = V_OPAC + 1;
    constexpr int LD = NVP + 1;                                   // odd row stride: conflict-free column walks
    __shared__ float s_v[256 * LD];
    __shared__ int s_col[R3DG_MAX_S_BWD];                         // feature column -> record channel, or -1 (written as zero)
    const int tid = threadIdx.x;
    const int g0 = blockIdx.x * 256;
    const int ng = min(256, P - g0);
    if (tid < R3DG_MAX_S_BWD) s_col[tid] = -1;
    __syncthreads();
    {
        const int n = chan_list.n < NF ? chan_list.n : NF;
        if (tid < n && tid < NVP - V_FEAT) s_col[chan_list.identity ? tid : chan_list.c[tid]] = V_FEAT + tid;
    }
    float4* rec = grad_records + (size_t)g0 * (NVP / 4);
    constexpr int Q = NVP / 4;                                    // float4s per record
#pragma unroll
    for (int i = 0; i < Q; i++) {
        const int e = tid + 256 * i;                              // float4 index inside the workgroup's run
        if (e < ng * Q) {
            const float4 t = rec[e];
            rec[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            float* d = s_v + (e / Q) * LD + 4 * (e % Q);
            d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
        }
    }
    __syncthreads();
    // dL_dcolors [ng,3]
    for (int e = tid; e < 3 * ng; e += 256) dL_dcolors[3 * (size_t)g0 + e] = s_v[(e / 3) * LD + e % 3];
    // dL_dmean2D [ng,3]: S_x, S_y, depth side channel
    for (int e = tid; e < 3 * ng; e += 256) {
        const int c = e % 3;
        dL_dmean2D[3 * (size_t)g0 + e] = (c == 2 && LEAN) ? 0.f : s_v[(e / 3) * LD + 3 + c];
    }
    // dL_dconic [ng,4]: S_xx, S_xy, -, S_yy
    for (int e = tid; e < 4 * ng; e += 256) {
        const int c = e & 3;
        dL_dconic2D[4 * (size_t)g0 + e] = c == 2 ? 0.f : s_v[(e >> 2) * LD + V_CONIC + (c == 3 ? 2 : c)];
    }
    if (tid < ng) dL_dopacity[g0 + tid] = s_v[tid * LD + V_OPAC];
    // dL_dfeature [ng,S]
    for (int e = tid; e < S * ng; e += 256) {
        const int col = s_col[e % S];
        dL_dfeature[(size_t)g0 * S + e] = col < 0 ? 0.f : s_v[(e / S) * LD + col];
    }
}

// The records: library scratch per (device, stream), zero whenever no backward is in flight on that stream -- the scatter kernel
// zeroes what the tile kernel may have written.  `dirty` guards the invariant on the host: set before the tile kernel is
// enqueued, cleared once the scatter kernel is; a call that finds it set (an enqueue in between failed) clears the buffer itself.
namespace {
struct RecordState { float* p = nullptr; size_t cap = 0; bool dirty = false; };
std::mutex g_records_mu;
std::map<std::pair<int, hipStream_t>, RecordState> g_records;
}

static float* gradient_records(hipStream_t s, size_t floats, bool** dirty_out)
{
    int dev = 0;
    R3DG_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_records_mu);
    RecordState& st = g_records[std::make_pair(dev, s)];
    if (st.cap < floats) {
        // geometric growth (first allocation: + 12 %): each growth costs a stream synchronise, a hipFree (device-wide wait),
        // a hipMalloc and a full memset -- a densifying scene must not pay that at every step of its growth
        const size_t want = st.p == nullptr ? floats + floats / 8 + 1024 : std::max(floats + 1024, 2 * st.cap);
        if (st.p != nullptr) {
            R3DG_HIP(hipStreamSynchronize(s));
            R3DG_HIP(hipFree(st.p));
            st.p = nullptr;
            st.cap = 0;
        }
        R3DG_HIP(hipMalloc((void**)&st.p, want * sizeof(float)));
        st.cap = want;
        st.dirty = true;
    }
    if (st.dirty) {
        R3DG_HIP(hipMemsetAsync(st.p, 0, st.cap * sizeof(float), s));
        st.dirty = false;
    }
    *dirty_out = &st.dirty;
    return st.p;
}

// r3dg_release_scratch (capi.hip): the device is idle when this runs
void release_gradient_records()
{
    std::lock_guard<std::mutex> lk(g_records_mu);
    for (auto& kv : g_records)
        if (kv.second.p != nullptr) (void)hipFree(kv.second.p);
    g_records.clear();
}

extern int g_cull;
int g_bwd_lean = 1;          // R3DG_OPT_BWD_LEAN

// ---- feature gradients only (frozen geometry) -------------------------------------------------------------------------
// The Synthetic4Relight / DTU stage-2 schedule (script/run_syn4.sh:27-33, run_dtu.sh) freezes positions, covariances, opacities
// and SH colour (learning rate 0) and trains only what reaches the image through the FEATURE maps (base colour, roughness,
// incident light).  Of the reference's backward (backward.cu:401-614) only
//     dL_dfeature[g, c] += alpha * T * dL_dpixel_f[c]                                                   (backward.cu:566)
// is then ever used: no accum_rec recursion, no dL_dalpha, no mean / conic / opacity / colour atomics, no payload staging
// (the feature VALUES do not enter) and no per-Gaussian geometry backward behind it.  One wave per 8x8 block like
// render_backward_wave_kernel (back-to-front walk from the block's own deepest contributor, per-block cull, compacted geometry in
// 2 KB of private LDS, records one round ahead), the transposing wave reduction; alpha and T are evaluated exactly as there.
template <int SPAD>
__global__ void __launch_bounds__(64)
render_backward_features_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int S,
                                ChannelList chan_list, int W, int H, int tiles_x, int num_tiles, int cull,
                                const uint32_t* __restrict__ tile_order, const float4* __restrict__ splat,
                                const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib,
                                const float* __restrict__ dL_dpixels_f, float* __restrict__ dL_dfeature)
{
    constexpr int NVP = next_pow2(SPAD);
    const int SA = chan_list.n;
    const int xcd = (int)(blockIdx.x & 7u), j = (int)(blockIdx.x >> 3), sub = j & 3, rank = (j >> 2) * 8 + xcd;
    if (rank >= num_tiles) return;
    const int tile = tile_order != nullptr ? (int)tile_order[rank] : rank;
    const int tile_x = tile % tiles_x, tile_y = tile / tiles_x;
    __shared__ float4 s_geo0[64];                 // mean.x, mean.y, conic.x, conic.y
    __shared__ float4 s_geo1[64];                 // conic.z, opacity, front index bits, id bits
    const int lane = threadIdx.x;
    const int bx = 8 * (sub & 1), by = 8 * (sub >> 1);
    const int px = tile_x * R3DG_TILE_X + bx + (lane & 7);
    const int py = tile_y * R3DG_TILE_Y + by + (lane >> 3);
    const float pxf = (float)px, pyf = (float)py;
    const float x0 = (float)(tile_x * R3DG_TILE_X + bx), y0 = (float)(tile_y * R3DG_TILE_Y + by);
    const size_t HW = (size_t)H * W;
    const uint2 range = ranges[tile];
    const int chan = transposed_channel<NVP>(lane);
    float* dst_base = nullptr;
    if (transposed_owner<NVP>(lane) && chan < SA) dst_base = dL_dfeature + chan_list.c[chan];
    const bool inside = px < W && py < H;
    const size_t pix = (size_t)py * W + px;
    float T = inside ? final_Ts[pix] : 0.f;
    const uint32_t lastc = inside ? n_contrib[pix] : 0u;
    float dl[SPAD];
#pragma unroll
    for (int ch = 0; ch < SPAD; ch++) dl[ch] = (inside && ch < SA) ? dL_dpixels_f[(size_t)chan_list.c[ch] * HW + pix] : 0.f;
    uint32_t mx = lastc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
    const int n = (int)mx;                        // this block's deepest contributor: the walk covers front indices [0, n)

    auto load_index = [&](int e0) -> uint32_t {
        return e0 + lane < n ? point_list[range.x + (uint32_t)(n - 1 - (e0 + lane))] : 0u;
    };
    auto stage = [&](const float4& a0, const float4& a1, uint32_t g, int e0) -> int {
        const bool cand = e0 + lane < n &&
                          (cull == 0 || splat_may_touch(a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, x0, x0 + 7.f, y0, y0 + 7.f));
        const unsigned long long m = __ballot(cand);
        if (cand) {
            const int slot = __popcll(m & ((1ull << lane) - 1ull));
            s_geo0[slot] = a0;
            s_geo1[slot] = make_float4(a1.x, a1.y, __uint_as_float((uint32_t)(n - 1 - (e0 + lane))), __uint_as_float(g));
        }
        return __popcll(m);
    };
    // records one round ahead, index two (render_backward_wave_kernel's pipeline without a payload)
    uint32_t g_cur = load_index(0);
    float4 r0 = splat[4 * (size_t)g_cur], r1 = splat[4 * (size_t)g_cur + 1];
    uint32_t g_nxt = load_index(64);
    for (int base = 0; base < n; base += 64) {
        const int ncand = stage(r0, r1, g_cur, base);
        g_cur = g_nxt;
        if (base + 64 < n) {
            r0 = splat[4 * (size_t)g_cur];
            r1 = splat[4 * (size_t)g_cur + 1];
            g_nxt = load_index(base + 128);
        }
        __builtin_amdgcn_wave_barrier();
        for (int k = 0; k < ncand; k++) {
            const float4 g0 = s_geo0[k], g1 = s_geo1[k];
            const uint32_t front = __float_as_uint(g1.z);
            const float dx = g0.x - pxf, dy = g0.y - pyf;
            const float power = -0.5f * (g0.z * dx * dx + g1.x * dy * dy) - g0.w * dx * dy;
            float a = fminf(0.99f, g1.y * fast_exp_b(power));
            if (!(front < lastc) || power > 0.0f || a < 1.0f / 255.0f) a = 0.f;
            if (__ballot(a != 0.f) == 0ull) continue;
            // back to front: T before this Gaussian = T after it / (1 - alpha)   (backward.cu:533)
            T = T * __builtin_amdgcn_rcpf(1.f - a);
            const float wgt = a * T;
            float vr[NVP];
#pragma unroll
            for (int q = 0; q < NVP; q++) vr[q] = q < SPAD ? wgt * dl[q] : 0.f;
            const float total = transpose_reduce<NVP, true>(vr);
            if (dst_base != nullptr) atomicAdd(dst_base + (size_t)__float_as_uint(g1.w) * (size_t)S, total);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

void launch_render_backward_features(hipStream_t s, int W, int H, int S, int n_active, const int* active,
                                     const uint32_t* tile_order, const uint32_t* ranges, const uint32_t* point_list,
                                     const float* splat, const float* final_Ts, const uint32_t* n_contrib,
                                     const float* dL_dpix_f, float* dL_dfeature)
{
    const int tiles_x = (W + R3DG_TILE_X - 1) / R3DG_TILE_X, tiles_y = (H + R3DG_TILE_Y - 1) / R3DG_TILE_Y;
    const int T = tiles_x * tiles_y, chunk = (T + 7) / 8;
    ChannelList cl;
    if (n_active < 0 || active == nullptr) {
        cl.n = S;
        cl.identity = 1;
        for (int i = 0; i < R3DG_MAX_S_BWD; i++) cl.c[i] = i < S ? i : 0;
    } else {
        cl.n = n_active;
        cl.identity = 0;
        for (int i = 0; i < R3DG_MAX_S_BWD; i++) cl.c[i] = i < n_active ? active[i] : 0;
    }
    if (cl.n == 0) return;
#define R3DG_BF(SP_)                                                                                                   \
    render_backward_features_kernel<SP_><<<chunk * 8 * 4, 64, 0, s>>>(                                                 \
        (const uint2*)ranges, point_list, S, cl, W, H, tiles_x, T, opt(R3DG_OPT_CULL), tile_order, (const float4*)splat, final_Ts,   \
        n_contrib, dL_dpix_f, dL_dfeature)
    switch ((cl.n + 3) / 4) {
        case 1: R3DG_BF(4); break;
        case 2: R3DG_BF(8); break;
        case 3: R3DG_BF(12); break;
        case 4: R3DG_BF(16); break;
        case 5: R3DG_BF(20); break;
        case 6: R3DG_BF(24); break;
        case 7: R3DG_BF(28); break;
        case 8: R3DG_BF(32); break;
        default: R3DG_BF(36); break;
    }
#undef R3DG_BF
}


void launch_render_backward(hipStream_t s, int P, int W, int H, int S, int n_active, const int* active,
                            const uint32_t* tile_order, const uint32_t* ranges, const uint32_t* point_list,
                            const float* bg, const float* splat, const float* features, const float* final_Ts,
                            const uint32_t* n_contrib, const float* dL_dpix, const float* dL_dpix_o,
                            const float* dL_dpix_d, const float* dL_dpix_f, float* dL_dmean2D, float* dL_dconic,
                            float* dL_dopacity, float* dL_dcolor, float* dL_dfeature, int bg_geom)
{
    const int tiles_x = (W + R3DG_TILE_X - 1) / R3DG_TILE_X, tiles_y = (H + R3DG_TILE_Y - 1) / R3DG_TILE_Y;
    const int T = tiles_x * tiles_y;
    ChannelList cl;
    if (n_active < 0 || active == nullptr) {
        cl.n = S;
        cl.identity = 1;
        for (int i = 0; i < R3DG_MAX_S_BWD; i++) cl.c[i] = i < S ? i : 0;
    } else {
        cl.n = n_active;
        cl.identity = 0;
        for (int i = 0; i < R3DG_MAX_S_BWD; i++) cl.c[i] = i < n_active ? active[i] : 0;
    }
    const int SP = cl.n;                 // channels the kernel carries
    const int grid = ((T + 7) / 8) * 8 * 4;       // 4 single-wave workgroups per tile, tile ranks padded to a multiple of 8
    // ROW4: every feature channel is carried and the rows are float4-aligned (S == SPAD): the payload is read as float4s
    const bool row4 = cl.identity != 0 && (S & 3) == 0;
    const int spad = (SP + 3) / 4 * 4;
    // LEAN: no depth gradient (the caller passed no dL_dout_depth), at least one padding slot among the SPAD feature slots, the
    // geometry part of dL_dalpha on (what every caller passes); only instantiated where it pays: up to 8 feature slots
    const bool lean = dL_dpix_d == nullptr && bg_geom != 0 && SP >= 1 && SP < spad && spad <= 8 && !row4;
    // reduction / record width of the instance that runs (the channels past it, if any, are padding feature slots)
    const bool use_lean = lean && opt(R3DG_OPT_BWD_LEAN) != 0;
    const int nv = use_lean ? 8 + spad : 10 + spad;
    const bool smallv = use_lean || spad == 4 || (spad == 8 && cl.n <= 6);
    const int nvp = smallv ? 16 : next_pow2(nv);
    bool* dirty = nullptr;
    float* rec = gradient_records(s, (size_t)P * nvp, &dirty);
    *dirty = true;
#define R3DG_BWD_ARGS                                                                                                  \
    (const uint2*)ranges, point_list, S, cl, W, H, tiles_x, T, opt(R3DG_OPT_CULL), tile_order, bg, (const float4*)splat, features,       \
        final_Ts, n_contrib, dL_dpix, dL_dpix_o, dL_dpix_d, dL_dpix_f, rec, bg_geom
#define R3DG_BWD(SP_, SV)                                                                                             \
    do {                                                                                                               \
        if (row4) render_backward_wave_kernel<SP_, SV, true, false><<<grid, 64, 0, s>>>(R3DG_BWD_ARGS);                \
        else render_backward_wave_kernel<SP_, SV, false, false><<<grid, 64, 0, s>>>(R3DG_BWD_ARGS);                    \
    } while (0)
    if (use_lean) {
        // 9 + (SPAD - 1) reduction channels: 12 (SPAD 4) and 16 (SPAD 8) -- both reduce 16-wide
        if (spad == 4) render_backward_wave_kernel<4, true, false, true><<<grid, 64, 0, s>>>(R3DG_BWD_ARGS);
        else render_backward_wave_kernel<8, true, false, true><<<grid, 64, 0, s>>>(R3DG_BWD_ARGS);
    } else {
        switch (spad / 4) {
            case 0: render_backward_wave_kernel<0, false, false, false><<<grid, 64, 0, s>>>(R3DG_BWD_ARGS); break;
            // 10 + n <= 16 gradient channels: half-size reduction
            case 1: R3DG_BWD(4, true); break;
            case 2: if (cl.n <= 6) R3DG_BWD(8, true); else R3DG_BWD(8, false); break;
            case 3: R3DG_BWD(12, false); break;
            case 4: R3DG_BWD(16, false); break;
            case 5: R3DG_BWD(20, false); break;
            case 6: R3DG_BWD(24, false); break;
            case 7: R3DG_BWD(28, false); break;
            case 8: R3DG_BWD(32, false); break;
            default: R3DG_BWD(36, false); break;
        }
    }
    // records -> the op's output arrays
    const int nf = use_lean ? spad - 1 : spad;
    const int sgrid = (P + 255) / 256;
#define R3DG_SCATTER(NVP_, LEAN_)                                                                                      \
    render_backward_scatter_kernel<NVP_, LEAN_><<<sgrid, 256, 0, s>>>(P, S, nf, cl, (float4*)rec, dL_dmean2D, dL_dconic, dL_dopacity, \
                                                                       dL_dcolor, dL_dfeature)
    if (use_lean) R3DG_SCATTER(16, true);
    else if (nvp == 16) R3DG_SCATTER(16, false);
    else if (nvp == 32) R3DG_SCATTER(32, false);
    else R3DG_SCATTER(64, false);
#undef R3DG_SCATTER
    *dirty = false;
#undef R3DG_BWD
#undef R3DG_BWD_ARGS
}

}  // namespace r3dg
