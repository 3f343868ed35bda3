// Device memory through HIP's virtual memory management calls with a chosen alignment of the VIRTUAL address (hipMalloc aligns to
// 2 MiB): the page table fragment the driver can use for a physically contiguous block is limited by the alignment of the virtual
// range, so a batch whose virtual address is aligned to 1 GiB can be translated in few, large fragments.  Experiment helper
// (scripts/bench_placement7.py), not product code.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
extern "C" {
struct VmmBlock { void* ptr; size_t size; hipMemGenericAllocationHandle_t handle; };
int vmm_alloc(size_t bytes, size_t alignment, VmmBlock* out)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) return 2;
    const size_t size = (bytes + gran - 1) / gran * gran;
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, size, &prop, 0) != hipSuccess) return 3;
    void* ptr = nullptr;
    if (hipMemAddressReserve(&ptr, size, alignment, nullptr, 0) != hipSuccess) { hipMemRelease(h); return 4; }
    if (hipMemMap(ptr, size, 0, h, 0) != hipSuccess) { hipMemAddressFree(ptr, size); hipMemRelease(h); return 5; }
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = dev;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(ptr, size, &acc, 1) != hipSuccess) { hipMemUnmap(ptr, size); hipMemAddressFree(ptr, size); hipMemRelease(h); return 6; }
    out->ptr = ptr; out->size = size; out->handle = h;
    return 0;
}
int vmm_free(VmmBlock* b)
{
    if (!b || !b->ptr) return 0;
    hipMemUnmap(b->ptr, b->size);
    hipMemAddressFree(b->ptr, b->size);
    hipMemRelease(b->handle);
    b->ptr = nullptr;
    return 0;
}
size_t vmm_granularity()
{
    int dev = 0;
    hipGetDevice(&dev);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
    return gran;
}
}
