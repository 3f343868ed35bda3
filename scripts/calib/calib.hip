// Calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on gfx950 for the access shapes the
// regrid kernels use (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access pattern").
// Every mode reads each byte of buf exactly once from HBM (buf >> Infinity Cache) and writes n/64 floats.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ void __launch_bounds__(256) read_kernel(const float* __restrict__ buf, size_t n, int mode, float* __restrict__ sink)
{
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t nthreads = (size_t)gridDim.x * 256;
    float acc = 0.f;
    if (mode == 0) {            // one dword per lane, consecutive lanes consecutive dwords
        for (size_t i = tid; i < n; i += nthreads) acc += buf[i];
    } else if (mode == 1) {     // bilinear-like: lane i reads dwords 2i and 2i+1 with two dword loads
        for (size_t i = tid; 2 * i + 1 < n; i += nthreads) { acc += buf[2 * i]; acc += buf[2 * i + 1]; }
    } else if (mode == 2) {     // 16 bytes per lane
        const float4* b4 = reinterpret_cast<const float4*>(buf);
        for (size_t i = tid; i < n / 4; i += nthreads) { float4 v = b4[i]; acc += v.x + v.y + v.z + v.w; }
    } else {                    // 8 rows in flight per lane, stride-2 dword pairs (8 independent 'slices')
        const size_t slice = n / 8;
        for (size_t i = tid; 2 * i + 1 < slice; i += nthreads) {
            float v[16];
#pragma unroll
            for (int k = 0; k < 8; ++k) { v[2 * k] = buf[k * slice + 2 * i]; v[2 * k + 1] = buf[k * slice + 2 * i + 1]; }
#pragma unroll
            for (int k = 0; k < 16; ++k) acc += v[k];
        }
    }
    if ((tid & 63) == 0 || acc == 123.456f) sink[tid / 64] = acc;
}

int main()
{
    const size_t n = (size_t)1 << 30;  // 4 GiB of floats
    float *buf, *sink;
    if (hipMalloc(&buf, n * 4) != hipSuccess) return 1;
    const int blocks = 256 * 8;
    if (hipMalloc(&sink, (size_t)blocks * 4 * 4) != hipSuccess) return 1;
    (void)hipMemset(buf, 0, n * 4);
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 4; ++mode) {
            hipEvent_t a, b;
            (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            (void)hipEventRecord(a);
            read_kernel<<<blocks, 256>>>(buf, n, mode, sink);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, a, b);
            printf("mode %d: %.3f ms, %.1f GB/s for %zu bytes\n", mode, ms, n * 4 / ms / 1e6, n * 4);
        }
    (void)hipFree(buf); (void)hipFree(sink);
    return 0;
}
