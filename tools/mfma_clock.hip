// Sustained shader clock and f32 MFMA rate of the GPU under the load shape of the MLP backward kernels:
// every SIMD of every CU issues v_mfma_f32_16x16x4_f32 back to back (4 independent accumulators) for ~20 ms.
// Prints the core clock (s_memtime ticks per 100 MHz s_memrealtime tick) and the achieved TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_clock.hip -o tools/_mfma_clock && tools/_mfma_clock [waves_per_simd]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

using f32x4 = __attribute__((ext_vector_type(4))) float;

__global__ __launch_bounds__(256) void mfma_loop(int iters, unsigned long long* out, float* sink)
{
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const float x = (float)threadIdx.x * 1e-3f, y = 1.0f + (float)blockIdx.x * 1e-6f;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        out[2 * w] = c1 - c0;
        out[2 * w + 1] = w1 - w0;
    }
    const f32x4 s = a0 + a1 + a2 + a3;
    if (s[0] == 12345.678f) sink[0] = s[1] + s[2] + s[3];
}

int main(int argc, char** argv)
{
    const int per_simd = argc > 1 ? atoi(argv[1]) : 1;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, blocks = cus * per_simd, iters = 40000 / per_simd;
    unsigned long long* out;
    float* sink;
    hipMalloc(&out, sizeof(unsigned long long) * 2 * blocks * 4);
    hipMalloc(&sink, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, iters, out, sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2];
        hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
        const double flops = (double)blocks * 4 * iters * 32.0 * 2048.0;
        printf("CUs %d, %d wave(s)/SIMD, %.2f ms: core clock %.0f MHz (wave 0), %.1f TFLOP/s f32 MFMA, %.1f cycles per MFMA per SIMD\n",
               cus, per_simd, ms, 100.0 * (double)h[0] / (double)h[1], flops / (ms * 1e-3) / 1e12,
               (double)h[0] / ((double)iters * 32.0 * per_simd));
    }
    return 0;
}
