// What do the out-of-range lanes of an LDS-DMA (buffer_load_dwordx4 ... lds) do to LDS on gfx950?  One wave presets 1 KiB of
// LDS to 7.0, then moves 64 x 16 bytes of which the odd lanes carry the offset ~0u (beyond num_records), and the LDS image is
// returned: 7.0 in the odd lanes' places = nothing written, 0.0 = zeros written.  Second case: the same lanes switched off
// in EXEC instead.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
using rsrc_t = __amdgpu_buffer_rsrc_t;
__global__ void k(const float* src, float* out, int mode)
{
    __shared__ __attribute__((aligned(16))) float lds[256];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = 7.0f;
    __syncthreads();
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 8192, 0x00020000);
    const bool odd = threadIdx.x & 1;
#if defined(__HIP_DEVICE_COMPILE__)
    using lds_ptr = __attribute__((address_space(3))) void*;
    if (mode == 0) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)lds, 16, odd ? 0xFFFFFFFFu : threadIdx.x * 16, 0, 0, 0);
    } else {
        if (!odd) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)lds, 16, threadIdx.x * 16, 0, 0, 0);
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}
int main()
{
    float *src, *out;
    CK(hipMalloc(&src, 16384)); CK(hipMalloc(&out, 1024));
    std::vector<float> h(4096), r(256);
    for (int i = 0; i < 4096; ++i) h[i] = (float)(i + 100);
    CK(hipMemcpy(src, h.data(), 16384, hipMemcpyHostToDevice));
    for (int mode = 0; mode < 2; ++mode) {
        CK(hipMemset(out, 0, 1024));
        k<<<1, 64>>>(src, out, mode);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(r.data(), out, 1024, hipMemcpyDeviceToHost));
        printf("{\"mode\": \"%s\", \"lane0\": [%g, %g, %g, %g], \"lane1\": [%g, %g, %g, %g], \"lane2\": [%g, %g, %g, %g], \"lane3\": [%g, %g, %g, %g]}\n",
               mode == 0 ? "offset out of range" : "EXEC off", r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[10], r[11], r[12], r[13], r[14], r[15]);
    }
    return 0;
}
