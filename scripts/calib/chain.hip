// Latency of the fill2d step's dependent chain on one wave (no memory): cycles per step for variants of the arithmetic.
// build on the box: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/chain scripts/calib/chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float dpp_up(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138, 0xf, 0xf, false)); }
template <int MODE>
__global__ void chain(float* out, long long* cyc, int iters, float r, float d, float w)
{
    float l = threadIdx.x * 0.001f, c = 1.0f, up = 0.5f;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        float e;
        if (MODE == 0) e = (float)((double)(((r + l) + d) + up) * 0.25 - (double)c);          // as the reference: via double
        else if (MODE == 1) e = (((r + l) + d) + up) * 0.25f - c;                              // all float (NOT bit-exact; for comparison)
        else e = (float)((double)(((r + l) + d) + up) * 0.25 - (double)c);
        const float res = c + e * w;
        l = res;
        if (MODE != 2) up = dpp_up(res);
        c = c + 1e-7f;
    }
    long long t1 = clock64();
    out[threadIdx.x] = l + up;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}
int main()
{
    float* d; long long* c; hipMalloc(&d, 256); hipMalloc(&c, 8);
    const int iters = 200000;
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            if (mode == 0) chain<0><<<1, 64>>>(d, c, iters, 1.f, 2.f, 0.7f);
            else if (mode == 1) chain<1><<<1, 64>>>(d, c, iters, 1.f, 2.f, 0.7f);
            else chain<2><<<1, 64>>>(d, c, iters, 1.f, 2.f, 0.7f);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
            if (rep) printf("{\"mode\": %d, \"ns_per_step\": %.1f, \"clock64_ticks_per_step\": %.1f}\n", mode, ms * 1e6 / iters, (double)cy / iters);
        }
    }
    return 0;
}
