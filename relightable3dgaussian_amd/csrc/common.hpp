// Shared host/device helpers for libr3dg_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include "r3dg_hip.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <string>

#define R3DG_TILE_X 16
#define R3DG_TILE_Y 16
#define R3DG_TILE_PIX 256
#define R3DG_MAX_S_FWD 36
#define R3DG_MAX_S_BWD 36

namespace r3dg {

void set_error(const std::string& msg);

struct HipError {
    int code;
};

// Host-side error plumbing: every HIP call goes through R3DG_HIP; the C-ABI wrapper catches and maps to a
// negative return code + r3dg_last_error().
#define R3DG_HIP(expr)                                                                                   \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess) {                                                                          \
            ::r3dg::set_error(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " (" + __FILE__ + \
                              ":" + std::to_string(__LINE__) + ")");                                     \
            throw ::r3dg::HipError{(int)_e};                                                             \
        }                                                                                                \
    } while (0)

// After a kernel launch: always check the launch; with debug also synchronise (reference CHECK_CUDA,
// auxiliary.h:166-173: sync + throw only when debug).
inline void check_launch(hipStream_t s, bool debug, const char* what)
{
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && debug) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        set_error(std::string(what) + ": " + hipGetErrorString(e));
        throw HipError{(int)e};
    }
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// CUs the persistent kernels (shading forward / backward, visibility trace) leave unoccupied so that a collective running
// beside them on another stream (RCCL's workgroups need LDS and registers on SOME CU) is not serialised behind them:
// their grids are sized for (CUs - g_reserve_cus).  0 on a single GPU; set by the data-parallel iteration
// (r3dg_set_option(R3DG_OPT_RESERVE_CUS)).
extern int g_reserve_cus;

// The value of a tuning option (enum r3dg_option) for THIS call: the calling thread's current option context
// (r3dg_context_make_current) where it sets the option, the process default (r3dg_set_option) otherwise.  Every launcher reads
// its knobs through this, at launch time, on the caller's thread -- two objects with different settings in one process (or two
// threads) therefore never see each other's values (capi.hip).
int opt(int option);

// Library-internal device scratch: one grow-only buffer per (device, stream, slot); the pointer stays valid until the next
// call with the same key asks for more (growth synchronises THAT stream only, so nothing else can still be using the old
// buffer).  Kernels of different streams or host threads never share scratch.  (capi.hip)
void* stream_scratch(hipStream_t stream, int slot, size_t bytes);

struct GeometryLayout {  // byte offsets into the opaque geometry buffer
    size_t depths, clamped, radii, means2D, cov3D, conic_opacity, rgb, tiles_touched, point_offsets, block_sums,
        total, splat, bytes;
    static GeometryLayout make(size_t P);
};
struct ImageLayout {
    size_t final_T, n_contrib, ranges, tile_order, big_list, big_count, bytes;
    static ImageLayout make(size_t N, size_t T);
};
struct BinningLayout {
    size_t keys_unsorted, keys, vals_unsorted, vals, sort_temp, bytes;
    static BinningLayout make(size_t R);
};

size_t sort_temp_bytes(size_t n);
void sort_pairs(hipStream_t stream, size_t n, uint64_t* keys_in, uint32_t* vals_in, uint64_t* keys_out,
                uint32_t* vals_out, int end_bit, void* temp, bool debug);
void sort_pairs_range(hipStream_t stream, size_t n, uint64_t* keys_in, uint32_t* vals_in, uint64_t* keys_out,
                      uint32_t* vals_out, int begin_bit, int end_bit, void* temp, bool debug, bool stable);

// ---- device helpers -------------------------------------------------------------------------------------
#ifdef __HIPCC__
// Scalar accumulators of the glue kernels (loss sums): R3DG_SUM_SLOTS consecutive floats per quantity, the value is the sum
// of the slots.  A float atomic on ONE address retires every ~35 ns on this part (measured: the stage-2 loss kernel got
// 27 us slower for every 768 additional workgroups); a few thousand workgroups adding to one word serialise for longer
// than the kernel runs.  Each workgroup adds to the slot of its linear index instead.
__device__ __forceinline__ float* sum_slot(float* sum)
{
    const unsigned b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    return sum + (b & (unsigned)(R3DG_SUM_SLOTS - 1));
}
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// Full-wave (64-lane) sum; result valid in every lane.
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Full-wave sum that stays in the VALU (DPP row shifts + row broadcasts, no LDS-crossbar shuffles): the total is
// valid in lane 63 ONLY.
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
#define R3DG_DPP_ADD(ctrl, rmask)                                                                                 \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xF, true))
    R3DG_DPP_ADD(0x111, 0xF);   // row_shr:1
    R3DG_DPP_ADD(0x112, 0xF);   // row_shr:2
    R3DG_DPP_ADD(0x114, 0xF);   // row_shr:4
    R3DG_DPP_ADD(0x118, 0xF);   // row_shr:8  -> lane 15 of each row holds the row total
    R3DG_DPP_ADD(0x142, 0xA);   // row_bcast:15 into rows 1 and 3
    R3DG_DPP_ADD(0x143, 0xC);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave total
#undef R3DG_DPP_ADD
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor((int)v, o, 64);
    return v;
}
// Conservative sub-tile cull used by both tile kernels: can the Gaussian (mean m, conic (a, b, c), opacity op) reach
// alpha >= 1/255 at ANY pixel centre of the box [x0,x1] x [y0,y1]?  alpha = op * exp(-q/2) with the convex quadratic
// q(d) = a dx^2 + 2 b dx dy + c dy^2, so it is enough to bound q from below over the box: 0 if the mean is inside,
// otherwise the smallest of the four edge minima (1-D parabolas, minimiser clamped to the edge).  A false answer
// means every pixel of the box would take the reference's `alpha < 1/255 -> continue` branch (forward.cu:343-345,
// backward.cu:529-531), so skipping the entry for the whole wave changes no result; slack terms keep borderline
// entries (fp32 rounding of q, approximate exp) on the evaluated side.  Non positive-definite conics are never culled.
__device__ __forceinline__ bool splat_may_touch(float mx, float my, float a, float b, float c, float op, float x0,
                                                float x1, float y0, float y1)
{
    const float X0 = x0 - mx, X1 = x1 - mx, Y0 = y0 - my, Y1 = y1 - my;
    if (!(a > 0.f && c > 0.f && a * c - b * b > 0.f)) return true;
    if (X0 <= 0.f && X1 >= 0.f && Y0 <= 0.f && Y1 >= 0.f) return true;
    const float ty = -b * __builtin_amdgcn_rcpf(c), tx = -b * __builtin_amdgcn_rcpf(a);
    float qmin;
    {
        const float d0 = fminf(fmaxf(ty * X0, Y0), Y1), d1 = fminf(fmaxf(ty * X1, Y0), Y1);
        const float q0 = a * X0 * X0 + 2.f * b * X0 * d0 + c * d0 * d0;
        const float q1 = a * X1 * X1 + 2.f * b * X1 * d1 + c * d1 * d1;
        qmin = fminf(q0, q1);
    }
    {
        const float d0 = fminf(fmaxf(tx * Y0, X0), X1), d1 = fminf(fmaxf(tx * Y1, X0), X1);
        const float q0 = a * d0 * d0 + 2.f * b * d0 * Y0 + c * Y0 * Y0;
        const float q1 = a * d1 * d1 + 2.f * b * d1 * Y1 + c * Y1 * Y1;
        qmin = fminf(qmin, fminf(q0, q1));
    }
    const float Xm = fmaxf(fabsf(X0), fabsf(X1)), Ym = fmaxf(fabsf(Y0), fabsf(Y1));
    const float mag = a * Xm * Xm + c * Ym * Ym + 2.f * fabsf(b) * Xm * Ym;    // size of the cancelling terms
    const float q = qmin - 1e-5f * mag;
    const float amax = op * __builtin_amdgcn_exp2f(-0.5f * 1.4426950408889634f * q);
    return !(amax < 0.98f / 255.0f);
}

// ---- block-cooperative row staging (256-thread blocks, one item per thread) ---------------------------------------------
// A thread-per-Gaussian kernel that walks its own [row_floats] row (e.g. the 48 SH coefficients, 192 B) makes every
// wave-level load touch 64 different cache lines.  These helpers move the block's 256 rows between HBM and LDS with
// fully coalesced 16-byte accesses instead; rows sit in LDS with an odd stride (row_floats + 1 words), so the per-thread
// walk afterwards is bank-conflict free.  Arithmetic is untouched -- results stay bit-identical.
__device__ __forceinline__ int staged_row_stride(int row_floats) { return row_floats | 1; }

// rows [first, first + 256) ∩ [0, P) of g -> s_rows; rows whose s_live byte is 0 are skipped (never read from HBM)
__device__ __forceinline__ void stage_rows_in_256(const float* __restrict__ g, int first, int P, int row_floats,
                                                  const uint8_t* s_live, float* s_rows)
{
    const int stride = staged_row_stride(row_floats);
    const int nrows = min(256, P - first);
    const float* __restrict__ src = g + (size_t)first * row_floats;
    if ((row_floats & 3) == 0 && ((size_t)src & 15) == 0) {
        const int q_per_row = row_floats >> 2, nq = nrows * q_per_row;
        // six loads in flight per thread before the first one is stored, and NO branch around a load: written as one loop of
        // `if (live) { load; store }` every load sits in its own branch and is waited for there -- 12 memory round trips in a row
        // for the 192-byte SH rows, most of the projection kernel's time.  A dead row's slot reads the block's first float4
        // instead (one cached line for all of them: still nothing of a culled Gaussian's row comes from HBM).
        constexpr int B = 6;
        for (int q0 = threadIdx.x; q0 < nq; q0 += 256 * B) {
            float4 v[B];
            int dst[B];
#pragma unroll
            for (int j = 0; j < B; j++) {
                const int q = min(q0 + j * 256, nq - 1);
                const int row = q / q_per_row, c = (q - row * q_per_row) * 4;
                const bool live = q0 + j * 256 < nq && s_live[row] != 0;
                dst[j] = live ? row * stride + c : -1;
                v[j] = *reinterpret_cast<const float4*>(src + (live ? 4 * (size_t)q : (size_t)0));
            }
#pragma unroll
            for (int j = 0; j < B; j++) {
                if (dst[j] >= 0) {
                    float* d = s_rows + dst[j];
                    d[0] = v[j].x; d[1] = v[j].y; d[2] = v[j].z; d[3] = v[j].w;
                }
            }
        }
    } else {
        const int n = nrows * row_floats;
        for (int f = threadIdx.x; f < n; f += 256) {
            const int row = f / row_floats, c = f - row * row_floats;
            if (s_live[row]) s_rows[row * stride + c] = src[f];
        }
    }
}

// s_rows -> rows [first, first + 256) ∩ [0, P) of g (every row is written)
__device__ __forceinline__ void stage_rows_out_256(float* __restrict__ g, int first, int P, int row_floats,
                                                   const float* s_rows)
{
    const int stride = staged_row_stride(row_floats);
    const int nrows = min(256, P - first);
    float* __restrict__ dst = g + (size_t)first * row_floats;
    if ((row_floats & 3) == 0 && ((size_t)dst & 15) == 0) {
        const int q_per_row = row_floats >> 2, nq = nrows * q_per_row;
#pragma unroll 4
        for (int q = threadIdx.x; q < nq; q += 256) {
            const int row = q / q_per_row, c = (q - row * q_per_row) * 4;
            const float* d = s_rows + row * stride + c;
            *reinterpret_cast<float4*>(dst + 4 * (size_t)q) = make_float4(d[0], d[1], d[2], d[3]);
        }
    } else {
        const int n = nrows * row_floats;
        for (int f = threadIdx.x; f < n; f += 256) {
            const int row = f / row_floats, c = f - row * row_floats;
            dst[f] = s_rows[row * stride + c];
        }
    }
}

// Inclusive scan across the wave.
__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v)
{
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t n = (uint32_t)__shfl_up((int)v, o, 64);
        if (l >= o) v += n;
    }
    return v;
}
#endif

}  // namespace r3dg
