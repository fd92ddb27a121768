// Transposing wave reduction for gfx950 (wave = 64 lanes), shared by the rasterizer backward, the shading kernels and
// the render-equation kernels.
#pragma once
#include "common.hpp"

namespace r3dg {

template <int N>
struct Log2 {
    static constexpr int value = 1 + Log2<N / 2>::value;
};
template <>
struct Log2<1> {
    static constexpr int value = 0;
};

// ---- transposing wave reduction ---------------------------------------------------------------------------
// Input: N (power of two, 16..64) partial values per lane.  Output: every lane holds the 64-lane total of ONE
// channel, chan(lane) = sum_t bit_{5-t}(lane) * (N >> (t+1)), t < log2 N.  Exchange levels run at lane distance
// 32,16,8,4,2,1: at each transposing level a pair of lanes (l, l^d) splits the remaining channels -- the lane with
// bit d clear keeps the lower half, the other the upper half -- so the live value count halves every level
// (N/2 + N/4 + ... exchanges instead of 6*N).  gfx950 specifics: distance 32/16 use v_permlane32_swap /
// v_permlane16_swap (swap half-waves / odd-even rows of two registers: exchange + select in ONE instruction),
// distance 8/4 use two bank-masked row_shl/row_shr DPP moves, distance 2/1 a quad_perm DPP move.  Everything
// stays in the VALU; no LDS-crossbar (ds_bpermute) traffic and no long-latency results to keep live.
template <int N>
__device__ __forceinline__ int transposed_channel(int lane)
{
    int idx = 0;
#pragma unroll
    for (int t = 0; t < Log2<N>::value; t++)
        if (lane & (32 >> t)) idx += N >> (t + 1);
    return idx;
}
// true for the one lane per channel that owns the result (low, non-transposed lane bits are zero)
template <int N>
__device__ __forceinline__ bool transposed_owner(int lane)
{
    return (lane & ((64 / N) - 1)) == 0;
}

template <int D>
__device__ __forceinline__ float lane_xor_dpp(float x)
{
    const int xi = __float_as_int(x);
    int r;
    if constexpr (D == 1) r = __builtin_amdgcn_update_dpp(0, xi, 0xB1, 0xF, 0xF, false);        // quad_perm [1,0,3,2]
    else if constexpr (D == 2) r = __builtin_amdgcn_update_dpp(0, xi, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    else if constexpr (D == 4) {
        r = __builtin_amdgcn_update_dpp(0, xi, 0x104, 0xF, 0x5, false);   // banks 0,2 read lane+4 (row_shl:4)
        r = __builtin_amdgcn_update_dpp(r, xi, 0x114, 0xF, 0xA, false);   // banks 1,3 read lane-4 (row_shr:4)
    } else {
        static_assert(D == 8, "DPP xor distance");
        r = __builtin_amdgcn_update_dpp(0, xi, 0x108, 0xF, 0x3, false);   // banks 0,1 read lane+8
        r = __builtin_amdgcn_update_dpp(r, xi, 0x118, 0xF, 0xC, false);   // banks 2,3 read lane-8
    }
    return __int_as_float(r);
}

template <int D, bool DPP>
__device__ __forceinline__ float lane_xor(float x)
{
    if constexpr (DPP && D <= 8) return lane_xor_dpp<D>(x);
    else return __shfl_xor(x, D, 64);
}

// Distance 8 / 4 transposing exchange in TWO instructions: lanes with bit D clear need a[l] + a[l+D], lanes with it set
// b[l] + b[l-D]; inside a 16-lane row those two lane sets are whole DPP banks (4 lanes each), so a bank-masked
// `v_add_f32_dpp r, a, a row_shl:D` fills the first set and `v_add_f32_dpp r, b, b row_shr:D` the second -- no selects.
// (Inline asm: the compiler will not fold a bank-masked DPP move into the add; s_nop covers the VALU-write -> DPP-read
// hazard that it cannot see through the asm.)
template <int D>
__device__ __forceinline__ float transpose_step_banked(float a, float b)
{
    float r;
    if constexpr (D == 8)
        asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
                     "v_add_f32_dpp %0, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xc"
                     : "=&v"(r) : "v"(a), "v"(b));
    else
        asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                     "v_add_f32_dpp %0, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa"
                     : "=&v"(r) : "v"(a), "v"(b));
    return r;
}

// one transposing exchange of the pair (lo-half value a, hi-half value b) at lane distance D
template <int D, bool DPP>
__device__ __forceinline__ float transpose_step(float a, float b, bool hi)
{
    if constexpr (DPP && (D == 8 || D == 4)) {
        return transpose_step_banked<D>(a, b);
    } else if constexpr (DPP && D == 32) {
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else if constexpr (DPP && D == 16) {
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else {
        const float send = hi ? a : b;
        const float keep = hi ? b : a;
        return keep + lane_xor<D, DPP>(send);
    }
}

template <int N, int LVL, bool DPP>
__device__ __forceinline__ void transpose_level(float (&v)[N], int lane)
{
    constexpr int D = 32 >> LVL;
    constexpr int half = N >> (LVL + 1);
    const bool hi = (lane & D) != 0;
#pragma unroll
    for (int k = 0; k < half; k++) v[k] = transpose_step<D, DPP>(v[k], v[k + half], hi);
}

template <int N, bool DPP>
__device__ __forceinline__ float transpose_reduce(float (&v)[N])
{
    const int lane = lane_id();
    constexpr int L = Log2<N>::value;
    if constexpr (L > 0) transpose_level<N, 0, DPP>(v, lane);
    if constexpr (L > 1) transpose_level<N, 1, DPP>(v, lane);
    if constexpr (L > 2) transpose_level<N, 2, DPP>(v, lane);
    if constexpr (L > 3) transpose_level<N, 3, DPP>(v, lane);
    if constexpr (L > 4) transpose_level<N, 4, DPP>(v, lane);
    if constexpr (L > 5) transpose_level<N, 5, DPP>(v, lane);
    float r = v[0];
    // remaining (non-transposing) distances: plain butterfly adds
    if constexpr (L <= 2) r += lane_xor<8, DPP>(r);
    if constexpr (L <= 3) r += lane_xor<4, DPP>(r);
    if constexpr (L <= 4) r += lane_xor<2, DPP>(r);
    if constexpr (L <= 5) r += lane_xor<1, DPP>(r);
    return r;
}

// ---- twelve channels ------------------------------------------------------------------------------------------------------
// The tile backward's lean instances carry 12 gradient channels (3 colour, 2 + 3 geometry moments, 1 opacity, 3 features).  Padded
// to 16 the generic reduction spends 15 pair steps (two instructions each) + 2 butterflies; twelve need only 11 pair steps when the
// levels split 12 -> 6 -> 3 -> (2 + 1) -> 1:
//   D = 32  permlane32_swap: pairs (c, c + 6); lanes with bit 5 clear keep channels 0..5, the others 6..11      6 x (swap, add)
//   D = 16  permlane16_swap: pairs (j, j + 3) of the six; bit 4 picks the first or the last three               3 x (swap, add)
//   D = 8   the first two of the three as one banked-DPP transposing step (bit 3 picks), the third one as a plain butterfly
//           (row_ror:8 = lane ^ 8 inside a 16-lane row, folded into the add)                                     2 + 1
//   D = 4   (that pair's value, the third one) as one banked-DPP transposing step: bit 2 picks                  2
//   D = 2, 1  plain butterflies                                                                                  1 + 1
// 25 VALU instructions instead of ~37.  chan(lane) = 6 b5 + 3 b4 + (b2 ? 2 : b3); the total of a channel sits in four lanes
// (bits 1, 0 free) -- and for the "third" channels in eight (bit 3 free too): the OWNER is the one with those bits clear.
__device__ __forceinline__ int transposed_channel12(int lane)
{
    return 6 * ((lane >> 5) & 1) + 3 * ((lane >> 4) & 1) + ((lane & 4) ? 2 : ((lane >> 3) & 1));
}
__device__ __forceinline__ bool transposed_owner12(int lane) { return (lane & 3) == 0 && !((lane & 4) && (lane & 8)); }

__device__ __forceinline__ float transpose_reduce12(float (&v)[12])
{
    float u[6], w[3];
#pragma unroll
    for (int c = 0; c < 6; c++) {
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[c]), __float_as_uint(v[c + 6]), false, false);
        u[c] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(u[c]), __float_as_uint(u[c + 3]), false, false);
        w[c] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    const float x = transpose_step_banked<8>(w[0], w[1]);
    float y;
    asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(y) : "v"(w[2]));
    float z = transpose_step_banked<4>(x, y);
    z += lane_xor_dpp<2>(z);
    z += lane_xor_dpp<1>(z);
    return z;
}

// ---- the same within each HALF-wave (32 lanes): lanes 0..31 and 32..63 reduce independent value sets --------------------
// N (power of two, 2..32) values per lane; every lane ends with the 32-lane total of channel
// chan(lane) = sum_t bit_{4-t}(lane) * (N >> (t+1)), t < log2 N; exchange distances 16, 8, 4, 2, 1 (no distance-32 level).
template <int N>
__device__ __forceinline__ int transposed_channel_half(int lane)
{
    int idx = 0;
#pragma unroll
    for (int t = 0; t < Log2<N>::value; t++)
        if (lane & (16 >> t)) idx += N >> (t + 1);
    return idx;
}
template <int N>
__device__ __forceinline__ bool transposed_owner_half(int lane)
{
    return (lane & ((32 / N) - 1)) == 0;
}
template <int N, int LVL>
__device__ __forceinline__ void transpose_level_half(float (&v)[N], int lane)
{
    constexpr int D = 16 >> LVL;
    constexpr int half = N >> (LVL + 1);
    const bool hi = (lane & D) != 0;
#pragma unroll
    for (int k = 0; k < half; k++) v[k] = transpose_step<D, true>(v[k], v[k + half], hi);
}
template <int N>
__device__ __forceinline__ float transpose_reduce_half(float (&v)[N])
{
    static_assert(N >= 2 && N <= 32, "half-wave transposing reduction: 2..32 values");
    const int lane = lane_id();
    constexpr int L = Log2<N>::value;
    if constexpr (L > 0) transpose_level_half<N, 0>(v, lane);
    if constexpr (L > 1) transpose_level_half<N, 1>(v, lane);
    if constexpr (L > 2) transpose_level_half<N, 2>(v, lane);
    if constexpr (L > 3) transpose_level_half<N, 3>(v, lane);
    if constexpr (L > 4) transpose_level_half<N, 4>(v, lane);
    float r = v[0];
    if constexpr (L <= 1) r += lane_xor<8, true>(r);
    if constexpr (L <= 2) r += lane_xor<4, true>(r);
    if constexpr (L <= 3) r += lane_xor<2, true>(r);
    if constexpr (L <= 4) r += lane_xor<1, true>(r);
    return r;
}

}  // namespace r3dg
