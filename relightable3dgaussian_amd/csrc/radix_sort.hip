// On-device STABLE LSD radix sort of (u64 key, u32 value) pairs for gfx950 -- replaces the reference's
// cub::DeviceRadixSort::SortPairs call (rasterizer_impl.cu:313-318): ascending on key bits [0,end_bit),
// equal keys keep their input order (the rasterizer relies on that: ties in (tile, depth) resolve to
// ascending Gaussian index).
//
// Pure integer work; wave64 formulation:
//   * digit width is chosen per call (<= 12 bits) so 44-45 key bits (800x800 .. 1600x1200) need 4 passes, and the
//     12 tile-id bits of an 800x800 frame are ONE pass (tile-binned ordering, below);
//   * per pass: (1) per-block digit histogram, (2) one block per digit scans that digit's per-block counts
//     (plus the digit's global base), (3) scatter with a wave-synchronous stable rank: each wave matches
//     equal digits with one 64-bit __ballot per digit bit, ranks by popcount below the lane, and keeps
//     per-wave digit counters in LDS -- no atomics and no block barrier inside the ranking loop.
#include "common.hpp"
#include <vector>
#include <mutex>
#include <map>

namespace r3dg {

constexpr int SORT_THREADS = 256;
constexpr int SORT_WAVES = SORT_THREADS / 64;
constexpr int SORT_ITEMS = 16;                                   // keys per thread
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;             // 4096 keys per block
constexpr int SORT_MAX_BITS = 12;
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
    extern __shared__ uint32_t s_cnt[];           // [SORT_WAVES][bins]
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

// (3') UNSTABLE scatter for partitions whose order inside a bucket does not matter (the tile-binned ordering sorts
// every bucket by a unique key afterwards): the rank inside the block is one integer LDS atomic per key -- no ballots,
// one counter array per block instead of one per wave.
__global__ void __launch_bounds__(SORT_THREADS)
partition_scatter_kernel(size_t n, const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                         uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, int shift, int bits,
                         uint32_t nblocks, const uint32_t* __restrict__ hist)
{
    extern __shared__ uint32_t s_cnt[];           // [bins]: running position of each digit for this block
    const int bins = 1 << bits;
    const uint32_t mask = (uint32_t)bins - 1u;
    for (int d = threadIdx.x; d < bins; d += SORT_THREADS) s_cnt[d] = hist[(size_t)d * nblocks + blockIdx.x];
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * SORT_TILE;
#pragma unroll 4
    for (int r = 0; r < SORT_ITEMS; r++) {
        const size_t i = base + (size_t)r * SORT_THREADS + threadIdx.x;
        if (i < n) {
            const uint64_t k = keys_in[i];
            const uint32_t pos = atomicAdd(&s_cnt[digit_of(k, shift, mask)], 1u);
            keys_out[pos] = k;
            vals_out[pos] = vals_in[i];
        }
    }
}

static inline uint32_t sort_nblocks(size_t n) { return (uint32_t)((n + SORT_TILE - 1) / SORT_TILE); }

static void sort_plan(int nbits, int& passes, int& bits)
{
    if (nbits < 1) nbits = 1;
    passes = (nbits + SORT_MAX_BITS - 1) / SORT_MAX_BITS;
    bits = (nbits + passes - 1) / passes;
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
    sort_pairs_range(stream, n, keys_in, vals_in, keys_out, vals_out, 0, end_bit, temp, debug, true);
}

// sort on key bits [begin_bit, end_bit) only; stable == false: equal keys end up in arbitrary order (single-pass
// partitions only -- a multi-pass LSD sort needs stable passes)
void sort_pairs_range(hipStream_t stream, size_t n, uint64_t* keys_in, uint32_t* vals_in, uint64_t* keys_out,
                      uint32_t* vals_out, int begin_bit, int end_bit, void* temp, bool debug, bool stable)
{
    if (n == 0) return;
    int passes, bits;
    sort_plan(end_bit - begin_bit, passes, bits);
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
        const int shift = begin_bit + p * bits;
        const int pbits = (end_bit - shift) < bits ? (end_bit - shift) : bits;
        uint32_t* tot = digit_total + (size_t)p * SORT_MAX_BINS;
        sort_hist_kernel<<<nb, SORT_THREADS, 0, stream>>>(n, ksrc, shift, pbits, nb, hist, tot);
        check_launch(stream, debug, "sort_hist_kernel");
        sort_scan_kernel<<<1 << pbits, 256, 0, stream>>>(1 << pbits, nb, hist, tot);
        check_launch(stream, debug, "sort_scan_kernel");
        if (!stable && passes == 1)
            partition_scatter_kernel<<<nb, SORT_THREADS, sizeof(uint32_t) << pbits, stream>>>(n, ksrc, vsrc, kdst, vdst,
                                                                                              shift, pbits, nb, hist);
        else
            sort_scatter_kernel<<<nb, SORT_THREADS, (size_t)SORT_WAVES * sizeof(uint32_t) << pbits, stream>>>(
                n, ksrc, vsrc, kdst, vdst, shift, pbits, nb, hist);
        check_launch(stream, debug, "sort_scatter_kernel");
        uint64_t* tk = ksrc; ksrc = kdst; kdst = tk;
        uint32_t* tv = vsrc; vsrc = vdst; vdst = tv;
    }
}

// ---- tile-binned ordering: per-tile depth sort ---------------------------------------------------------------------
// The reference sorts all R (tile | depth) keys globally (rasterizer_impl.cu:313-318: 4 radix passes over 44 bits here).
// Same final order with less memory traffic: ONE stable radix pass on the tile-id bits only (sort_pairs_range), which
// leaves every tile's instances contiguous and in Gaussian-index order, then one workgroup per tile sorts its list by
// (depth, index) -- a unique key, so any comparison sort reproduces the reference's stable order bit for bit.
// One workgroup sorts one tile's (depth bits << 32 | Gaussian index) entries ascending and writes the reference's sorted key
// (tile << 32 | depth) and point-list arrays: lists of up to 4096 entries with a bitonic network in LDS ("flip" first step, so
// every compare-exchange is ascending and +inf padding never moves), longer ones with the segmented radix sort further down
// (whose rare fall-back is the same network in place in global memory, a fence per stage).
template <typename Mem>
__device__ __forceinline__ void bitonic_ascending(Mem& m, uint32_t n, uint32_t n_pad, int nt)
{
    for (uint32_t k = 2; k <= n_pad; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < n_pad / 2; t += nt) {
                uint32_t lo, hi;
                if (j == (k >> 1)) {                       // flip: i <-> mirror inside its k-block
                    const uint32_t blk = t / j, off = t % j;
                    lo = blk * k + off;
                    hi = blk * k + (k - 1 - off);
                } else {
                    lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    hi = lo | j;
                }
                if (hi < n) {                              // virtual +inf padding: pairs reaching past n never swap
                    const uint64_t a = m.load(lo), b = m.load(hi);
                    if (a > b) { m.store(lo, b); m.store(hi, a); }
                }
            }
            m.stage_sync();
        }
    }
}

// LDS bitonic sort of n_pad (power of two, padded with +inf) keys by NT/64 waves with few workgroup barriers: every
// wave owns a contiguous chunk of C = n_pad / waves keys; all compare-exchange stages at distance < C stay inside one
// wave's chunk and need no barrier (LDS operations of one wave execute in order), only the distance >= C stages of the
// merges across chunks are separated by __syncthreads -- 6 barriers instead of 55 for 1024 keys on 4 waves.
template <int NT>
__device__ __forceinline__ void bitonic_lds_waves(uint64_t* s, uint32_t n_pad)
{
    constexpr uint32_t NW = NT / 64;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // chunk per wave; tiny lists are sorted by wave 0 alone
    const uint32_t C = n_pad >= 128u * NW ? n_pad / NW : n_pad;
    const bool mine = n_pad >= 128u * NW || wave == 0;
    uint64_t* c = s + (n_pad >= 128u * NW ? wave * C : 0u);
    auto local_stage = [&](uint32_t k, uint32_t j, bool flip) {
        for (uint32_t t = lane; t < C / 2; t += 64) {
            uint32_t lo, hi;
            if (flip) {
                const uint32_t blk = t / j, off = t % j;
                lo = blk * k + off;
                hi = blk * k + (k - 1 - off);
            } else {
                lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                hi = lo | j;
            }
            const uint64_t a = c[lo], b = c[hi];
            if (a > b) { c[lo] = b; c[hi] = a; }
        }
    };
    if (mine)
        for (uint32_t k = 2; k <= C; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) local_stage(k, j, j == (k >> 1));
    for (uint32_t k = 2 * C; k <= n_pad; k <<= 1) {
        for (uint32_t j = k >> 1; j >= C; j >>= 1) {           // cross-chunk stages
            __syncthreads();
            for (uint32_t t = threadIdx.x; t < n_pad / 2; t += NT) {
                uint32_t lo, hi;
                if (j == (k >> 1)) {
                    const uint32_t blk = t / j, off = t % j;
                    lo = blk * k + off;
                    hi = blk * k + (k - 1 - off);
                } else {
                    lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    hi = lo | j;
                }
                const uint64_t a = s[lo], b = s[hi];
                if (a > b) { s[lo] = b; s[hi] = a; }
            }
        }
        __syncthreads();
        for (uint32_t j = C >> 1; j > 0; j >>= 1) local_stage(2 * j, j, false);   // plain i <-> i^j inside the chunk
    }
    __syncthreads();
}

struct LdsMem {
    uint64_t* p;
    __device__ __forceinline__ uint64_t load(uint32_t i) const { return p[i]; }
    __device__ __forceinline__ void store(uint32_t i, uint64_t v) const { p[i] = v; }
    __device__ __forceinline__ void stage_sync() const { __syncthreads(); }
};
struct GlobalMem {
    uint64_t* p;
    __device__ __forceinline__ uint64_t load(uint32_t i) const { return p[i]; }
    __device__ __forceinline__ void store(uint32_t i, uint64_t v) const { p[i] = v; }
    __device__ __forceinline__ void stage_sync() const { __threadfence(); __syncthreads(); }
};

__device__ __forceinline__ uint32_t next_pow2_u32(uint32_t n)
{
    return n <= 1u ? 1u : 1u << (32 - __builtin_clz(n - 1u));
}

// keys/vals hold the tile's instances (tile << 32 | depth, Gaussian index) in index order (stable partition by tile);
// they are replaced in place by the same instances in ascending (depth, index) order.
// ENTRIES: `scratch` already holds the tile's (depth << 32 | index) sort entries (direct tile binning,
// rasterizer_preprocess.hip) and `tile` names the tile; otherwise keys/vals hold the partitioned (tile | depth, index) pairs.
template <int NT, bool ENTRIES = false>
__device__ __forceinline__ void sort_one_tile(uint2 rg, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                              uint64_t* __restrict__ scratch, uint64_t* s_buf, uint32_t cap,
                                              uint32_t tile = 0)
{
    const uint32_t n = rg.y - rg.x;
    const uint32_t n_pad = next_pow2_u32(n);
    const uint64_t tile_hi = ENTRIES ? ((uint64_t)tile << 32) : (keys[rg.x] & 0xffffffff00000000ull);
    for (uint32_t i = threadIdx.x; i < n_pad; i += NT)
        s_buf[i] = i < n ? (ENTRIES ? scratch[rg.x + i] : (keys[rg.x + i] << 32) | vals[rg.x + i])
                         : ~0ull;      // real +inf padding in LDS
    __syncthreads();
    bitonic_lds_waves<NT>(s_buf, n_pad);
    for (uint32_t i = threadIdx.x; i < n; i += NT) {
        const uint64_t e = s_buf[i];
        keys[rg.x + i] = tile_hi | (e >> 32);
        vals[rg.x + i] = (uint32_t)e;
    }
    __syncthreads();
}

// small tiles: one 256-thread workgroup per tile (longest first), lists up to `cap` entries in dynamic LDS
template <bool ENTRIES>
__global__ void __launch_bounds__(256)
tile_sort_small_kernel(int T, const uint32_t* __restrict__ tile_order, const uint2* __restrict__ ranges, uint32_t cap,
                       uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, uint64_t* __restrict__ scratch)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t s_sort[];
    const uint32_t tile = tile_order ? tile_order[blockIdx.x] : blockIdx.x;
    const uint2 rg = ranges[tile];
    const uint32_t n = rg.y - rg.x;
    if (n > cap || n == 0) return;
    if (n < 2 && !ENTRIES) return;                     // (a single partitioned pair is already in place)
    sort_one_tile<256, ENTRIES>(rg, keys, vals, scratch, s_sort, cap, tile);
}

// ---- long tiles: segmented LSD radix sort -------------------------------------------------------------------------------------
// A bitonic network is O(n log^2 n): at 2 M Gaussians 2577 of a 1800x700 frame's 4972 tiles hold more than 4096 instances
// (15.4 M of 19.0 M; longest 13410) and the network -- in 128 KB of LDS up to 16384 entries, one workgroup per CU -- took 884 us
// of the forward (rounds 1-2; 2.7 ms before it had a workgroup on every CU).  This kernel: 378 us for the same lists
// (profiles/r03_sort_long_tiles_kernel_stats.md), same bits.  One 1024-thread workgroup per long tile runs a STABLE least-significant-digit radix sort over the
// depth bits that actually differ inside the tile (the common prefix -- sign, most of the exponent -- is found first:
// typically 22-26 bits, 3-4 passes of <= 8 bits), ping-ponging the tile's (depth << 32 | index) entries between its scratch
// segment and its slice of the output key array (both L2-resident).  Per pass: every wave owns a contiguous 1/16 of the
// list; (1) per-wave digit histogram (LDS atomics), (2) exclusive scan over (digit, wave), (3) every wave walks its range in
// order, 64 keys at a time, ranks equal digits with one ballot per digit bit (sort_scatter_kernel's scheme) and scatters --
// no workgroup barrier inside (1) or (3).  The direct tile binning hands the entries over in ARBITRARY order, so equal depths
// are ordered by Gaussian index afterwards: runs of equal depth are short in any real frame and each member counts the
// smaller indices of its run; a run longer than LONG_TIE_RUN sends the tile to the bitonic network on the full
// (depth, index) key, which needs no such assumption.
constexpr int LONG_THREADS = 1024;
constexpr int LONG_WAVES = LONG_THREADS / 64;
constexpr int LONG_BATCH = 8;                 // keys per lane in registers at a time
constexpr int LONG_BITS = 8;
constexpr int LONG_BINS = 1 << LONG_BITS;
constexpr uint32_t LONG_TIE_RUN = 64;

// loads of entries that other waves of this workgroup wrote earlier in the kernel, a workgroup barrier in between: the waves of
// a workgroup share their CU's write-through vector L1, so workgroup scope needs no cache maintenance (an agent-scope
// __threadfence() here writes back and invalidates the XCD's L2 -- measured: 7.7 ms instead of 1 ms for the 2577 long tiles of
// the 2 M-Gaussian frame).  Relaxed atomic loads at workgroup scope are ordinary global loads that the compiler may neither
// hoist above the barrier nor keep in registers.
__device__ __forceinline__ uint64_t load_wg(const uint64_t* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t load_wg(const uint32_t* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

struct LongSortLds {
    uint32_t cnt[LONG_WAVES * LONG_BINS];
    uint32_t base[LONG_BINS];
    uint32_t wave_sum[LONG_BINS / 64];
    uint32_t diff, ties, next;
};

template <bool ENTRIES>
__device__ __forceinline__ void radix_sort_one_tile(uint2 rg, uint64_t* keys, uint32_t* vals, uint64_t* scratch, uint32_t tile,
                                                    LongSortLds& L)
{
    const uint32_t n = rg.y - rg.x;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint64_t* seg = scratch + rg.x;                 // the tile's entries (buffer A)
    uint64_t* alt = keys + rg.x;                    // buffer B: the tile's slice of the output keys
    const uint64_t tile_hi = ENTRIES ? ((uint64_t)tile << 32) : (keys[rg.x] & 0xffffffff00000000ull);
    const uint32_t first = ENTRIES ? (uint32_t)(seg[0] >> 32) : (uint32_t)keys[rg.x];
    if (tid == 0) { L.diff = 0; L.ties = 0; }
    __syncthreads();
    uint32_t diff = 0;
    for (uint32_t i = tid; i < n; i += LONG_THREADS) {
        uint64_t e;
        if (ENTRIES) {
            e = seg[i];
        } else {
            e = (keys[rg.x + i] << 32) | vals[rg.x + i];
            seg[i] = e;
        }
        diff |= (uint32_t)(e >> 32) ^ first;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) diff |= (uint32_t)__shfl_xor((int)diff, o, 64);
    if (lane == 0 && diff) atomicOr(&L.diff, diff);
    __syncthreads();
    diff = L.diff;
    const int nbits = diff ? 32 - __builtin_clz(diff) : 0;
    const int passes = (nbits + LONG_BITS - 1) / LONG_BITS;
    const int bits = passes ? (nbits + passes - 1) / passes : 0;
    uint64_t* src = seg;
    uint64_t* dst = alt;
    if (passes & 1) {                               // the passes must END in seg: the output loop reads seg and writes the slice
        for (uint32_t i = tid; i < n; i += LONG_THREADS) alt[i] = load_wg(seg + i);
        __syncthreads();
        src = alt;
        dst = seg;
    }
    const uint32_t chunk = ((n + LONG_THREADS - 1) / LONG_THREADS) * 64;       // keys per wave (a multiple of 64)
    const uint32_t w_lo = min(n, wave * chunk), w_hi = min(n, w_lo + chunk);
    const bool single = chunk <= 64u * LONG_BATCH;   // the wave's keys stay in registers between the histogram and the scatter
    volatile uint32_t* cnt = L.cnt + wave * LONG_BINS;
    for (int p = 0; p < passes; p++) {
        const int shift = 32 + p * bits;
        const int pb = (nbits - p * bits) < bits ? (nbits - p * bits) : bits;
        const uint32_t bins = 1u << pb, mask = bins - 1u;
        for (uint32_t i = tid; i < LONG_WAVES * LONG_BINS; i += LONG_THREADS) L.cnt[i] = 0;
        __syncthreads();
        uint64_t key[LONG_BATCH];
        for (uint32_t b0 = w_lo; b0 < w_hi; b0 += 64u * LONG_BATCH) {
#pragma unroll
            for (int r = 0; r < LONG_BATCH; r++) {
                const uint32_t i = b0 + r * 64 + lane;
                key[r] = i < w_hi ? load_wg(src + i) : ~0ull;
            }
#pragma unroll
            for (int r = 0; r < LONG_BATCH; r++)
                if (b0 + r * 64 + lane < w_hi) atomicAdd(&L.cnt[wave * LONG_BINS + digit_of(key[r], shift, mask)], 1u);
        }
        __syncthreads();
        // cnt[w][d] <- number of digit-d keys in waves before w; base[d] <- number of keys with a smaller digit
        uint32_t total = 0, inc = 0;
        if (tid < LONG_BINS) {
            if (tid < bins) {
                uint32_t run = 0;
#pragma unroll
                for (int w = 0; w < LONG_WAVES; w++) {
                    const uint32_t c = L.cnt[w * LONG_BINS + tid];
                    L.cnt[w * LONG_BINS + tid] = run;
                    run += c;
                }
                total = run;
            }
            inc = wave_inclusive_scan_u32(total);
            if (lane == 63) L.wave_sum[wave] = inc;
        }
        __syncthreads();
        if (tid < LONG_BINS) {
            uint32_t off = 0;
            for (uint32_t w = 0; w < wave; w++) off += L.wave_sum[w];
            L.base[tid] = off + inc - total;
        }
        __syncthreads();
        for (uint32_t b0 = w_lo; b0 < w_hi; b0 += 64u * LONG_BATCH) {
            if (!single) {
#pragma unroll
                for (int r = 0; r < LONG_BATCH; r++) {
                    const uint32_t i = b0 + r * 64 + lane;
                    key[r] = i < w_hi ? load_wg(src + i) : ~0ull;
                }
            }
#pragma unroll
            for (int r = 0; r < LONG_BATCH; r++) {
                if (b0 + r * 64 >= w_hi) break;                          // (wave-uniform)
                const bool valid = b0 + r * 64 + lane < w_hi;
                const uint32_t d = digit_of(key[r], shift, mask);
                unsigned long long peers = __ballot(valid);
                for (int b = 0; b < pb; b++) {
                    const bool bit = (d >> b) & 1u;
                    const unsigned long long bal = __ballot(bit);
                    peers &= bit ? bal : ~bal;
                }
                const uint32_t below = __popcll(peers & ((1ull << lane) - 1ull));
                const uint32_t count = __popcll(peers);
                uint32_t old = 0, db = 0;
                if (valid) { old = cnt[d]; db = L.base[d]; }
                __builtin_amdgcn_wave_barrier();
                if (valid && below == 0) cnt[d] = old + count;
                __builtin_amdgcn_wave_barrier();
                if (valid) dst[db + old + below] = key[r];
            }
        }
        __syncthreads();
        uint64_t* t = src; src = dst; dst = t;
    }
    // seg holds the entries in ascending depth; equal depths in arbitrary order: each member of such a run takes the
    // place of its index rank inside the run
    const uint32_t* seg32 = reinterpret_cast<const uint32_t*>(seg);          // [2 i] = index, [2 i + 1] = depth bits
    bool long_run = false;
    for (uint32_t i = tid; i < n; i += LONG_THREADS) {
        const uint64_t e = load_wg(seg + i);
        const uint32_t d = (uint32_t)(e >> 32), idx = (uint32_t)e;
        uint32_t pos = i;
        const bool tie = (i > 0 && load_wg(seg32 + 2 * (i - 1) + 1) == d) || (i + 1 < n && load_wg(seg32 + 2 * (i + 1) + 1) == d);
        if (tie) {
            uint32_t lo = i, len = 1, rank = 0;
            while (lo > 0 && len <= LONG_TIE_RUN && load_wg(seg32 + 2 * (lo - 1) + 1) == d) {
                lo--; len++;
                rank += load_wg(seg32 + 2 * lo) < idx;
            }
            for (uint32_t j = i + 1; j < n && len <= LONG_TIE_RUN && load_wg(seg32 + 2 * j + 1) == d; j++) {
                len++;
                rank += load_wg(seg32 + 2 * j) < idx;
            }
            long_run = long_run || len > LONG_TIE_RUN;
            pos = lo + rank;
        }
        keys[rg.x + pos] = tile_hi | d;
        vals[rg.x + pos] = idx;
    }
    if (long_run) L.ties = 1;
    __syncthreads();
    if (L.ties) {                                   // (rare) a long run of equal depths: the network on the unique 64-bit key
        GlobalMem m{seg};
        bitonic_ascending(m, n, next_pow2_u32(n), LONG_THREADS);
        for (uint32_t i = tid; i < n; i += LONG_THREADS) {
            const uint64_t e = load_wg(seg + i);
            keys[rg.x + i] = tile_hi | (e >> 32);
            vals[rg.x + i] = (uint32_t)e;
        }
    }
    __syncthreads();
}

// persistent workgroups take the long tiles off big_list through a cursor (big_count[1], zeroed by tile_order_kernel)
template <bool ENTRIES>
__global__ void __launch_bounds__(LONG_THREADS)
tile_sort_long_kernel(const uint32_t* __restrict__ big_list, uint32_t* __restrict__ big_count,
                      const uint2* __restrict__ ranges, uint64_t* keys, uint32_t* vals, uint64_t* scratch)
{
    __shared__ LongSortLds L;
    const uint32_t nbig = big_count[0];
    for (;;) {
        if (threadIdx.x == 0) L.next = atomicAdd(&big_count[1], 1u);
        __syncthreads();
        const uint32_t k = L.next;
        __syncthreads();
        if (k >= nbig) break;
        const uint32_t tile = big_list[k];
        radix_sort_one_tile<ENTRIES>(ranges[tile], keys, vals, scratch, tile, L);
    }
}

// Tiles up to this many entries take the bitonic network in LDS (x 8 bytes per workgroup), longer ones the segmented radix sort.
// 4096 (rounds 2-3): a 2049..4096-entry tile pads to 4096 -- 78 dependent stages of 8 pairs per thread at 5 workgroups per CU --
// and those tiles, dispatched first, set the kernel's length: 104.6 us at 300k Gaussians / 800x800.  2048: 80.9 us (16 KB of LDS,
// 10 workgroups per CU; the ~10 % of tiles above it cost the radix kernel less than they cost the network); 1024: 87.0 us.
#ifndef R3DG_TILE_SORT_SMALL_CAP
#define R3DG_TILE_SORT_SMALL_CAP 2048
#endif
constexpr uint32_t TILE_SORT_SMALL_CAP = R3DG_TILE_SORT_SMALL_CAP;

uint32_t tile_sort_small_cap() { return TILE_SORT_SMALL_CAP; }

// entries == true: `scratch` holds the (depth << 32 | index) entries of the direct tile binning, already in their tiles' segments
void launch_tile_sort(hipStream_t s, int T, const uint32_t* tile_order, const uint32_t* ranges, const uint32_t* big_list,
                      uint32_t* big_count, uint64_t* keys, uint32_t* vals, uint64_t* scratch, bool entries)
{
    const uint2* rg = (const uint2*)ranges;
    // persistent grid: 2 x 1024 threads fill a CU (most workgroups exit at once when few tiles are long); the device's CU count
    // minus the CUs left to a collective
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int reserve = opt(R3DG_OPT_RESERVE_CUS);
    const int grid = 2 * (cus > reserve ? cus - reserve : 1);
    // (Round 5, measured and not kept: the long tiles' kernel on a side stream of the library beside the small tiles' kernel -- the
    // two sort disjoint tiles.  Beside each other the long kernel takes 82 us instead of 30, the pair ends where the sequence did:
    // 772-774 against 776-777 it/s.  One stream.)
    if (entries) {
        tile_sort_small_kernel<true><<<T, 256, TILE_SORT_SMALL_CAP * 8, s>>>(T, tile_order, rg, TILE_SORT_SMALL_CAP, keys, vals, scratch);
        tile_sort_long_kernel<true><<<grid, LONG_THREADS, 0, s>>>(big_list, big_count, rg, keys, vals, scratch);
    } else {
        tile_sort_small_kernel<false><<<T, 256, TILE_SORT_SMALL_CAP * 8, s>>>(T, tile_order, rg, TILE_SORT_SMALL_CAP, keys, vals, scratch);
        tile_sort_long_kernel<false><<<grid, LONG_THREADS, 0, s>>>(big_list, big_count, rg, keys, vals, scratch);
    }
}

}  // namespace r3dg
