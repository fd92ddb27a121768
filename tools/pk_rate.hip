// Issue rate of v_pk_fma_f32 against v_fma_f32 on the device at hand: 8 independent accumulator chains per lane,
// 4096 iterations, 8 waves per SIMD.  Prints lane-FMAs per clock per CU for both forms.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <bool PK>
__global__ void __launch_bounds__(256) rate_kernel(float* out, float a, float b, int iters)
{
    f2 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = (f2){(float)threadIdx.x + i, (float)i};
    const f2 va = {a, a * 1.0001f}, vb = {b, b};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (PK) {
                asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(acc[i]) : "v"(va), "v"(vb));
            } else {
                asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(acc[i].x) : "v"(va.x), "v"(vb.x));
                asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(acc[i].y) : "v"(va.y), "v"(vb.y));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; i++) s += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, blocks = cus * 8, iters = 4096;
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pk = 0; pk < 2; pk++) {
        float ms = 0.f;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (pk) rate_kernel<true><<<blocks, 256>>>(out, 0.999f, 0.001f, iters);
            else rate_kernel<false><<<blocks, 256>>>(out, 0.999f, 0.001f, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double lane_fmas = (double)blocks * 256 * iters * 16;
        const double clk = prop.clockRate * 1e3;      // Hz
        printf("%s: %.3f ms, %.1f lane-FMAs per clock per CU (at %.2f GHz nominal), %.1f TFLOP/s\n",
               pk ? "v_pk_fma_f32" : "v_fma_f32   ", ms, lane_fmas / (ms * 1e-3) / clk / cus, clk / 1e9,
               2.0 * lane_fmas / (ms * 1e-3) / 1e12);
    }
    return 0;
}
