// Does buffer_load_dwordx4 ... lds (LDS-DMA) need 16-byte aligned global addresses on gfx950?  One wave moves 64 x 16 bytes
// from src + shift (shift = 0, 4, 8, 12 bytes, through the base pointer and through the per-lane offset) and the result is
// compared on the host.  Also dword (4-byte) DMA at odd dword offsets.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
using rsrc_t = __amdgpu_buffer_rsrc_t;
__global__ void k(const float* src, float* out, int shiftBase, int shiftOff)
{
    __shared__ __attribute__((aligned(16))) float lds[256];
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(src)) + shiftBase, 0, 8192, 0x00020000);
#if defined(__HIP_DEVICE_COMPILE__)
    using lds_ptr = __attribute__((address_space(3))) void*;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)lds, 16, threadIdx.x * 16 + shiftOff, 0, 0, 0);
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
    for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    CK(hipMemcpy(src, h.data(), 16384, hipMemcpyHostToDevice));
    for (int sb = 0; sb < 16; sb += 4)
        for (int so = 0; so < 16; so += 4) {
            CK(hipMemset(out, 0, 1024));
            k<<<1, 64>>>(src, out, sb, so);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(r.data(), out, 1024, hipMemcpyDeviceToHost));
            int bad = 0;
            for (int i = 0; i < 256; ++i) if (r[i] != (float)(i + (sb + so) / 4)) ++bad;
            printf("{\"base_shift\": %d, \"offset_shift\": %d, \"wrong_values\": %d, \"first\": [%g, %g, %g, %g, %g]}\n", sb, so, bad, r[0], r[1], r[2], r[3], r[4]);
        }
    return 0;
}
