// Four consecutive floats of an LDS image at a 4-byte (not 16-byte) aligned address: what does the 4 x 4 stencil of the bicubic
// kernel pay for them?  The compiler splits such a read into two ds_read2_b32 (it assumes aligned DS access on gfx9); the hardware
// runs compute queues in unaligned mode (KFD sets SH_MEM_CONFIG.ALIGNMENT_MODE to unaligned), so ds_read_b128 / ds_read_b64 at
// any dword address may simply work.  Measured here, one workgroup per CU: (1) are the values right for every alignment,
// (2) cycles per wave instruction group for lanes whose windows start `stride` dwords apart (2.16 on the benchmark plan),
// for 2 x ds_read2_b32, 1 x ds_read_b128, 2 x ds_read_b64 and ds_read_b96 + b32.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__device__ __forceinline__ f4 read4(const float* p)
{
    f4 v;
#if defined(__HIP_DEVICE_COMPILE__)
    // (the kernel has no static LDS: the dynamic array starts at LDS address 0, an LDS address is the byte offset into it)
    const uint32_t a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
    if (MODE == 0) {
        f2 lo, hi;
        asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read2_b32 %1, %2 offset0:2 offset1:3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(lo), "=&v"(hi) : "v"(a) : "memory");
        v.x = lo.x; v.y = lo.y; v.z = hi.x; v.w = hi.y;
    } else if (MODE == 1) {
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a) : "memory");
    } else if (MODE == 2) {
        f2 lo, hi;
        asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)" : "=&v"(lo), "=&v"(hi) : "v"(a) : "memory");
        v.x = lo.x; v.y = lo.y; v.z = hi.x; v.w = hi.y;
    } else {
        float x0, x1, x2, x3;
        asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:4\n\tds_read_b32 %2, %4 offset:8\n\tds_read_b32 %3, %4 offset:12\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(a) : "memory");
        v.x = x0; v.y = x1; v.z = x2; v.w = x3;
    }
#else
    (void)p; v = f4{0, 0, 0, 0};
#endif
    return v;
}

// lanes read windows of 4 floats starting at (shift + lane * strideQ / 16) dwords; rows: 4 rows `pitch` dwords apart
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cycles, int shift, int strideQ, int pitch, int iters)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 12288; i += 256) lds[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int start = shift + (lane * strideQ) / 16 + wave * 1024;
    f4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f4 v = read4<MODE>(lds + start + r * pitch + (it & 3));
            acc += v;
        }
    }
    const unsigned long long t1 = clock64();
    if (blockIdx.x == 0) {
        out[threadIdx.x * 4 + 0] = acc.x; out[threadIdx.x * 4 + 1] = acc.y; out[threadIdx.x * 4 + 2] = acc.z; out[threadIdx.x * 4 + 3] = acc.w;
        if (threadIdx.x == 0) *cycles = t1 - t0;
    }
}

int main()
{
    float* out; unsigned long long* cyc;
    CK(hipMalloc(&out, 4096)); CK(hipMalloc(&cyc, 8));
    std::vector<float> r(1024);
    const char* names[4] = {"2 x ds_read2_b32", "ds_read_b128", "2 x ds_read_b64", "4 x ds_read_b32"};
    for (int mode = 0; mode < 4; ++mode)
        for (int shift = 0; shift < 4; ++shift)
            for (int strideQ : {16, 35, 64}) {  // 1, 2.19 and 4 dwords between neighbouring lanes' windows
                const int iters = 4096, pitch = 257;
                auto launch = [&](int grid) {
                    if (mode == 0) k<0><<<grid, 256, 49152>>>(out, cyc, shift, strideQ, pitch, iters);
                    else if (mode == 1) k<1><<<grid, 256, 49152>>>(out, cyc, shift, strideQ, pitch, iters);
                    else if (mode == 2) k<2><<<grid, 256, 49152>>>(out, cyc, shift, strideQ, pitch, iters);
                    else k<3><<<grid, 256, 49152>>>(out, cyc, shift, strideQ, pitch, iters);
                };
                launch(256);
                CK(hipDeviceSynchronize());
                unsigned long long c = 0;
                CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(r.data(), out, 4096, hipMemcpyDeviceToHost));
                // expected: sum over iterations and rows of the window values
                int bad = 0;
                for (int t = 0; t < 256; ++t) {
                    const int lane = t & 63, wave = t >> 6;
                    const int start = shift + (lane * strideQ) / 16 + wave * 1024;
                    double e[4] = {0, 0, 0, 0};
                    for (int it = 0; it < iters; ++it)
                        for (int rr = 0; rr < 4; ++rr)
                            for (int j = 0; j < 4; ++j) e[j] += (double)(start + rr * pitch + (it & 3) + j);
                    for (int j = 0; j < 4; ++j) if ((float)e[j] != r[t * 4 + j] && fabs(e[j] - r[t * 4 + j]) > 1e-3 * e[j]) ++bad;
                }
                printf("{\"read\": \"%s\", \"first_dword_mod4\": %d, \"lane_stride_dwords\": %.2f, \"wrong_sums\": %d, \"cycles_per_4x4_window_per_wave\": %.1f}\n",
                       names[mode], shift, strideQ / 16.0, bad, (double)c / iters);
            }
    return 0;
}
