// On-device STABLE LSD radix sort of (u64 key, u32 value) pairs for gfx950 -- replaces the reference's
// cub::DeviceRadixSort::SortPairs call (rasterizer_impl.cu:313-318): ascending on key bits [0,end_bit),
// equal keys keep their input order (the rasterizer relies on that: ties in (tile, depth) resolve to
// ascending Gaussian index).
//
// Pure integer work; wave64 formulation:
//   * digit width is chosen per call (<= 11 bits) so 44-45 key bits (800x800 .. 1600x1200) need 4 passes;
//   * per pass: (1) per-block digit histogram, (2) one block per digit scans that digit's per-block counts
//     (plus the digit's global base), (3) scatter with a wave-synchronous stable rank: each wave matches
//     equal digits with one 64-bit __ballot per digit bit, ranks by popcount below the lane, and keeps
//     per-wave digit counters in LDS -- no atomics and no block barrier inside the ranking loop.
#include "common.hpp"

namespace r3dg {

constexpr int SORT_THREADS = 256;
constexpr int SORT_WAVES = SORT_THREADS / 64;
constexpr int SORT_ITEMS = 16;                                   // keys per thread
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;             // 4096 keys per block
constexpr int SORT_MAX_BITS = 11;
constexpr int SORT_MAX_BINS = 1 << SORT_MAX_BITS;

__device__ __forceinline__ uint32_t digit_of(uint64_t key, int shift, uint32_t mask)
{
    return (uint32_t)(key >> shift) & mask;
}

// (1) hist[d * nblocks + b] = number of keys of block b with digit d; digit_total[d] += same (global atomics).
__global__ void __launch_bounds__(SORT_THREADS)
sort_hist_kernel(size_t n, const uint64_t* __restrict__ keys, int shift, int bits, uint32_t nblocks,
                 uint32_t* __restrict__ hist, uint32_t* __restrict__ digit_total)
{
    __shared__ uint32_t s_hist[SORT_MAX_BINS];
    const int bins = 1 << bits;
    const uint32_t mask = (uint32_t)bins - 1u;
    for (int i = threadIdx.x; i < bins; i += SORT_THREADS) s_hist[i] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * SORT_TILE;
#pragma unroll 4
    for (int r = 0; r < SORT_ITEMS; r++) {
        const size_t i = base + (size_t)r * SORT_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&s_hist[digit_of(keys[i], shift, mask)], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < bins; d += SORT_THREADS) {
        const uint32_t c = s_hist[d];
        hist[(size_t)d * nblocks + blockIdx.x] = c;
        if (c) atomicAdd(&digit_total[d], c);
    }
}

// (2) block d: base = sum(digit_total[0..d)) ; hist[d*nblocks + b] <- base + exclusive prefix over b.
__global__ void __launch_bounds__(256)
sort_scan_kernel(int bins, uint32_t nblocks, uint32_t* __restrict__ hist, const uint32_t* __restrict__ digit_total)
{
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_carry;
    const int d = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // digit base: reduce digit_total[0..d)
    uint32_t part = 0;
    for (int i = tid; i < d; i += 256) part += digit_total[i];
    part = wave_sum_u32(part);
    if (lane == 0) s_wave[wave] = part;
    __syncthreads();
    if (tid == 0) s_carry = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
    uint32_t* row = hist + (size_t)d * nblocks;
    for (uint32_t b0 = 0; b0 < nblocks; b0 += 256) {
        const uint32_t b = b0 + tid;
        const uint32_t v = b < nblocks ? row[b] : 0u;
        const uint32_t inc = wave_inclusive_scan_u32(v);
        __syncthreads();            // previous iteration's readers of s_wave are done
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        uint32_t off = s_carry;
        for (int w = 0; w < wave; w++) off += s_wave[w];
        if (b < nblocks) row[b] = off + inc - v;
        __syncthreads();
        if (tid == 0) s_carry += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
}

// (3) stable scatter. Wave w of block b owns keys [b*TILE + w*ITEMS*64, +ITEMS*64), visited 64 at a time in
// index order, so (block, wave, round, lane) order == input order.
__global__ void __launch_bounds__(SORT_THREADS)
sort_scatter_kernel(size_t n, const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                    uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, int shift, int bits,
                    uint32_t nblocks, const uint32_t* __restrict__ hist)
{
    __shared__ uint32_t s_cnt[SORT_WAVES * SORT_MAX_BINS];
    const int bins = 1 << bits;
    const uint32_t mask = (uint32_t)bins - 1u;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < SORT_WAVES * bins; i += SORT_THREADS) s_cnt[i] = 0;
    __syncthreads();

    volatile uint32_t* cnt = s_cnt + wave * bins;
    const size_t wave_base = (size_t)blockIdx.x * SORT_TILE + (size_t)wave * (SORT_ITEMS * 64);
    uint64_t key[SORT_ITEMS];
    uint32_t rank[SORT_ITEMS];
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; r++) {
        const size_t i = wave_base + (size_t)r * 64 + lane;
        key[r] = i < n ? keys_in[i] : ~0ull;
    }
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; r++) {
        const size_t i = wave_base + (size_t)r * 64 + lane;
        const bool valid = i < n;
        const uint32_t d = digit_of(key[r], shift, mask);
        // peers = lanes of this wave holding the same digit (and valid)
        unsigned long long peers = __ballot(valid);
        for (int b = 0; b < bits; b++) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const uint32_t below = __popcll(peers & ((1ull << lane) - 1ull));
        const uint32_t count = __popcll(peers);
        uint32_t old = 0;
        if (valid) old = cnt[d];                       // every peer reads the pre-round counter ...
        __builtin_amdgcn_wave_barrier();
        if (valid && below == 0) cnt[d] = old + count; // ... then the lowest peer bumps it (LDS ops are in order per wave)
        __builtin_amdgcn_wave_barrier();
        rank[r] = old + below;
    }
    __syncthreads();
    // turn per-wave counts into global positions: hist base + exclusive prefix over the block's waves
    for (int d = tid; d < bins; d += SORT_THREADS) {
        uint32_t run = hist[(size_t)d * nblocks + blockIdx.x];
#pragma unroll
        for (int w = 0; w < SORT_WAVES; w++) {
            const uint32_t c = s_cnt[w * bins + d];
            s_cnt[w * bins + d] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; r++) {
        const size_t i = wave_base + (size_t)r * 64 + lane;
        if (i < n) {
            const uint32_t d = digit_of(key[r], shift, mask);
            const uint32_t pos = s_cnt[wave * bins + d] + rank[r];
            keys_out[pos] = key[r];
            vals_out[pos] = vals_in[i];
        }
    }
}

static inline uint32_t sort_nblocks(size_t n) { return (uint32_t)((n + SORT_TILE - 1) / SORT_TILE); }

static void sort_plan(int end_bit, int& passes, int& bits)
{
    if (end_bit < 1) end_bit = 1;
    passes = (end_bit + SORT_MAX_BITS - 1) / SORT_MAX_BITS;
    bits = (end_bit + passes - 1) / passes;
}

size_t sort_temp_bytes(size_t n)
{
    // per-pass digit totals (cleared once) + the (digit, block) table
    const size_t nb = sort_nblocks(n ? n : 1);
    return align_up((size_t)8 * SORT_MAX_BINS * sizeof(uint32_t), 256) + align_up(nb * SORT_MAX_BINS * sizeof(uint32_t), 256);
}

void sort_pairs(hipStream_t stream, size_t n, uint64_t* keys_in, uint32_t* vals_in, uint64_t* keys_out,
                uint32_t* vals_out, int end_bit, void* temp, bool debug)
{
    if (n == 0) return;
    int passes, bits;
    sort_plan(end_bit, passes, bits);
    if (passes > 8) {
        set_error("sort_pairs: end_bit too large");
        throw HipError{-1};
    }
    const uint32_t nb = sort_nblocks(n);
    uint32_t* digit_total = (uint32_t*)temp;                                  // [passes][SORT_MAX_BINS]
    uint32_t* hist = (uint32_t*)((char*)temp + align_up((size_t)8 * SORT_MAX_BINS * sizeof(uint32_t), 256));
    R3DG_HIP(hipMemsetAsync(digit_total, 0, (size_t)passes * SORT_MAX_BINS * sizeof(uint32_t), stream));

    uint64_t* ksrc = keys_in;
    uint32_t* vsrc = vals_in;
    uint64_t* kdst = keys_out;
    uint32_t* vdst = vals_out;
    if ((passes & 1) == 0) {
        // even number of ping-pongs would end in the input buffers: start from a copy in the output buffers
        R3DG_HIP(hipMemcpyAsync(keys_out, keys_in, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream));
        R3DG_HIP(hipMemcpyAsync(vals_out, vals_in, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
        ksrc = keys_out; vsrc = vals_out; kdst = keys_in; vdst = vals_in;
    }
    for (int p = 0; p < passes; p++) {
        const int shift = p * bits;
        const int pbits = (end_bit - shift) < bits ? (end_bit - shift) : bits;
        uint32_t* tot = digit_total + (size_t)p * SORT_MAX_BINS;
        sort_hist_kernel<<<nb, SORT_THREADS, 0, stream>>>(n, ksrc, shift, pbits, nb, hist, tot);
        check_launch(stream, debug, "sort_hist_kernel");
        sort_scan_kernel<<<1 << pbits, 256, 0, stream>>>(1 << pbits, nb, hist, tot);
        check_launch(stream, debug, "sort_scan_kernel");
        sort_scatter_kernel<<<nb, SORT_THREADS, 0, stream>>>(n, ksrc, vsrc, kdst, vdst, shift, pbits, nb, hist);
        check_launch(stream, debug, "sort_scatter_kernel");
        uint64_t* tk = ksrc; ksrc = kdst; kdst = tk;
        uint32_t* tv = vsrc; vsrc = vdst; vdst = tv;
    }
}

}  // namespace r3dg
