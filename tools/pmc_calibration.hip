// Calibration micro-kernels for rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950: every kernel
// moves a KNOWN number of bytes in ONE access pattern over a 1 GiB array (4x the 256 MiB Infinity Cache), so the ratio
// counter/bytes can be read off per pattern and applied to the product kernels' counters.
//   hipcc --offload-arch=gfx950 -O3 tools/pmc_calibration.hip -o tools/pmc_calibration
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d ... -- tools/pmc_calibration      (WRITE_SIZE in a second pass)
// tools/pmc_calibration.py merges the passes into profiles/<round>_pmc_calibration.json.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

namespace calib {

__device__ inline uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// coalesced 16 B/lane streaming read, n float4
__global__ void stream_read16(const float4* __restrict__ a, size_t n, float* sink)
{
    float s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = a[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456f) *sink = s;
}
// coalesced 4 B/lane streaming read, n floats
__global__ void stream_read4(const float* __restrict__ a, size_t n, float* sink)
{
    float s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
    if (s == 123.456f) *sink = s;
}
// coalesced 16 B/lane copy
__global__ void stream_copy16(const float4* __restrict__ a, float4* __restrict__ b, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// coalesced 4 B/lane / 16 B/lane streaming writes
__global__ void stream_write4(float* __restrict__ b, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = 1.0f;
}
__global__ void stream_write16(float4* __restrict__ b, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        b[i] = make_float4(1, 2, 3, 4);
}
// random gather of RECORD-byte records (RECORD = 4, 8, 16: one lane per record; 64: four lanes x 16 B per record --
// the shape of the tile kernels' staging gathers): m records out of nrec, each record index drawn by a hash
template <int RECORD>
__global__ void gather_records(const char* __restrict__ a, uint32_t nrec, size_t m, float* sink)
{
    float s = 0;
    constexpr int LANES = RECORD >= 16 ? RECORD / 16 : 1;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < m * LANES; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t rec = hash32((uint32_t)(i / LANES)) % nrec;
        const char* p = a + (size_t)rec * RECORD + (i % LANES) * 16;
        if (RECORD == 4) s += *(const float*)p;
        else if (RECORD == 8) { float2 v = *(const float2*)p; s += v.x + v.y; }
        else { float4 v = *(const float4*)p; s += v.x + v.y + v.z + v.w; }
    }
    if (s == 123.456f) *sink = s;
}
// float atomics scattered over a small table (the per-Gaussian gradient accumulation pattern): m atomics into `slots` floats
__global__ void scatter_atomic4(float* __restrict__ b, uint32_t slots, size_t m)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < m; i += (size_t)gridDim.x * blockDim.x)
        atomicAdd(&b[hash32((uint32_t)i) % slots], 1.0f);
}

}  // namespace calib

int main()
{
    const size_t BYTES = (size_t)1 << 30;
    char *a, *b;
    float* sink;
    CK(hipMalloc(&a, BYTES));
    CK(hipMalloc(&b, BYTES));
    CK(hipMalloc(&sink, 256));
    CK(hipMemset(a, 0, BYTES));
    CK(hipMemset(b, 0, BYTES));
    const int G = 256 * 16, T = 256;
    const size_t M = (size_t)16 << 20;           // gathered records per launch
    for (int rep = 0; rep < 3; rep++) {
        calib::stream_read16<<<G, T>>>((const float4*)a, BYTES / 16, sink);
        calib::stream_read4<<<G, T>>>((const float*)a, BYTES / 4, sink);
        calib::stream_copy16<<<G, T>>>((const float4*)a, (float4*)b, BYTES / 16);
        calib::stream_write4<<<G, T>>>((float*)b, BYTES / 4);
        calib::stream_write16<<<G, T>>>((float4*)b, BYTES / 16);
        calib::gather_records<4><<<G, T>>>(a, (uint32_t)(BYTES / 4), M, sink);
        calib::gather_records<8><<<G, T>>>(a, (uint32_t)(BYTES / 8), M, sink);
        calib::gather_records<16><<<G, T>>>(a, (uint32_t)(BYTES / 16), M, sink);
        calib::gather_records<64><<<G, T>>>(a, (uint32_t)(BYTES / 64), M, sink);
        calib::scatter_atomic4<<<G, T>>>((float*)b, 300000 * 26, M);
        CK(hipDeviceSynchronize());
    }
    printf("{\"bytes\": %zu, \"gather_records\": %zu, \"atomic_slots\": %d}\n", BYTES, M, 300000 * 26);
    return 0;
}
