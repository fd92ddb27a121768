// TEST INFRASTRUCTURE ONLY -- a lock-step CPU emulation of the HIP execution model for SIMPLE kernels (no inline asm, no
// DPP / LDS-DMA builtins), so that kernel SOURCE written without GPU access can be run by `pytest -m "not gpu"`:
// every lane of a workgroup is a std::thread; threadIdx / blockIdx are thread-local; `__shared__` variables are function
// statics (workgroups run one after the other); __syncthreads is a barrier over the workgroup, __shfl_xor a barrier-fenced
// exchange inside the 64-lane wave.  Nothing of the product links against this; the parity tests proper run the same
// source on the MI355X (tests/test_relight_gpu.py).
#pragma once
#include <barrier>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <thread>
#include <vector>

struct emu_uint3 { unsigned x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct float3 { float x, y, z; };
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

inline thread_local emu_uint3 threadIdx, blockIdx, blockDim, gridDim;

struct EmuBlock {
    std::unique_ptr<std::barrier<>> block_barrier;
    std::vector<std::unique_ptr<std::barrier<>>> wave_barrier;
    std::vector<float> exchange;          // one slot per lane
};
inline EmuBlock* g_emu_block = nullptr;

inline void __syncthreads() { g_emu_block->block_barrier->arrive_and_wait(); }

inline float __shfl_xor(float v, int mask, int width = 64)
{
    (void)width;
    const unsigned t = threadIdx.x, wave = t >> 6;
    g_emu_block->exchange[t] = v;
    g_emu_block->wave_barrier[wave]->arrive_and_wait();
    const float r = g_emu_block->exchange[(t & ~63u) | ((t ^ (unsigned)mask) & 63u)];
    g_emu_block->wave_barrier[wave]->arrive_and_wait();
    return r;
}

inline float __builtin_amdgcn_rsqf(float x) { return 1.0f / std::sqrt(x); }

// On the hardware the lanes of a wave execute one instruction stream, so "store to LDS, then read what other lanes of the
// SAME wave stored" needs no synchronisation; the kernels mark such points with this scheduling barrier (no instruction
// is emitted), and here, where lanes are free-running threads, it is a real barrier over the wave.
inline void __builtin_amdgcn_wave_barrier() { g_emu_block->wave_barrier[threadIdx.x >> 6]->arrive_and_wait(); }

#define __global__
#define __device__
#define __forceinline__ inline
#define __restrict__
#define __launch_bounds__(...)
#define __shared__ static

// Runs `kernel(args...)` for a 1-D grid of `grid` workgroups of `block` threads (a multiple of 64).  A thread that
// returns early must do so together with its whole wave (and never before a __syncthreads other waves still reach) --
// the same rule the hardware imposes.
template <typename K, typename... A>
void emu_launch(K kernel, unsigned grid, unsigned block, A... args)
{
    for (unsigned b = 0; b < grid; b++) {
        EmuBlock blk;
        blk.block_barrier = std::make_unique<std::barrier<>>(block);
        for (unsigned w = 0; w < block / 64; w++) blk.wave_barrier.push_back(std::make_unique<std::barrier<>>(64));
        blk.exchange.assign(block, 0.f);
        g_emu_block = &blk;
        std::vector<std::thread> lanes;
        lanes.reserve(block);
        for (unsigned t = 0; t < block; t++)
            lanes.emplace_back([=]() {
                threadIdx = {t, 0, 0};
                blockIdx = {b, 0, 0};
                blockDim = {block, 1, 1};
                gridDim = {grid, 1, 1};
                kernel(args...);
            });
        for (auto& th : lanes) th.join();
        g_emu_block = nullptr;
    }
}
