// Calibration of rocprofv3's FETCH_SIZE on gfx950 for narrow loads (MI355X_MICROARCH.md, HBM: "other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern"): streams a 2 GiB buffer (eight times the
// 256 MiB Infinity Cache) once per kernel with 4-, 8- and 16-byte-per-lane coalesced loads, and with the scaled environment
// kernel's pattern (a wavefront reads a 2,624-byte record: 8-byte-per-lane and 4-byte-per-lane pieces, records 2,624 bytes
// apart, only the first 832 bytes of each touched).
//   hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o tools/_calib_fetch
//   rocprofv3 --pmc FETCH_SIZE -d gpurun_out/calib -o calib -- tools/_calib_fetch      (then tools/pmc_summary.py)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

template <typename T>
__global__ __launch_bounds__(256) void read_kernel(const T* __restrict__ p, size_t n, float* __restrict__ out)
{
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T v = p[i];
        const float* f = reinterpret_cast<const float*>(&v);
        for (unsigned k = 0; k < sizeof(T) / 4; ++k) acc += f[k];
    }
    if (acc == 12345.678f) out[0] = acc;      // keeps the loads alive
}

// one wavefront per record: 64 doubles (load/queue-like, 8 B per lane) + 64 ints (4 B per lane) + 16 doubles = 896 B of 2,624
__global__ __launch_bounds__(256) void record_kernel(const unsigned char* __restrict__ p, size_t n_rec, float* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    double acc = 0.0;
    for (size_t r = wave; r < n_rec; r += n_waves) {
        const unsigned char* rec = p + r * 2624;
        acc += reinterpret_cast<const double*>(rec)[lane];
        acc += (double)reinterpret_cast<const int*>(rec + 512)[lane];
        if (lane < 8) acc += reinterpret_cast<const double*>(rec + 768)[lane];
    }
    if (acc == 12345.678) out[0] = (float)acc;
}

int main()
{
    const size_t bytes = (size_t)2 << 30;
    unsigned char* buf = nullptr;
    float* out = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    const int grid = 256 * 8;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read_kernel<float>, dim3(grid), dim3(256), 0, 0, (const float*)buf, bytes / 4, out);
        hipLaunchKernelGGL(read_kernel<float2>, dim3(grid), dim3(256), 0, 0, (const float2*)buf, bytes / 8, out);
        hipLaunchKernelGGL(read_kernel<float4>, dim3(grid), dim3(256), 0, 0, (const float4*)buf, bytes / 16, out);
        hipLaunchKernelGGL(record_kernel, dim3(grid), dim3(256), 0, 0, buf, bytes / 2624, out);
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("run failed\n"); return 1; }
    printf("bytes per launch: stream %zu, records %zu x (touched 832 of 2624)\n", bytes, bytes / 2624);
    return 0;
}
