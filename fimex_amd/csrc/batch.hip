// Output batches placed by the library (include/fimex_amd.h: fimex_amd_regrid_batch_alloc_device).
//
// The reference allocates the result of every interpolateValues call itself (src/CachedInterpolation.cc:123, `new float[newSize]`).
// A device-resident caller allocates the batch once and reuses it, and on this memory system WHERE that batch lies moves the
// apply launch by several per cent with identical code and traffic (the read and the write stream of the launch meet
// in the memory channels differently).  So the allocation is a service of the library: the batch is mapped at a few windows of
// one reserved address range, each window backed by its own physical chunks (HIP virtual memory management), the plan's own apply
// launch is timed on the caller's source batch with the output in each window, the fastest window stays and the physical memory
// of all others goes back to the driver.  What the probing cost (seconds, bytes mapped meanwhile) is reported with the batch.
#include "plan.hpp"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <memory>
#include <vector>

struct fimex_amd_batch {
    int device = 0;
    bool vmm = false;
    char* base = nullptr;      // reserved range (vmm) or hipMalloc'ed arena
    size_t reserved = 0;       // bytes of the range
    size_t chunkBytes = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;  // one per chunk of the range
    std::vector<char> mapped;                              // chunk still mapped?
    fimex_amd_batch_info info{};
};

namespace fimex_amd {

namespace {

constexpr size_t kChunkBytes = size_t(32) << 20;  // physical handle size: large enough for large page-table fragments
constexpr size_t kStepChunks = 22;                // windows 704 MiB apart

void release_batch(fimex_amd_batch& b) noexcept
{
    if (!b.base) return;
    if (b.vmm) {
        for (size_t c = 0; c < b.handles.size(); ++c) {
            if (b.mapped[c]) (void)hipMemUnmap(b.base + c * b.chunkBytes, b.chunkBytes);
            if (b.mapped[c]) (void)hipMemRelease(b.handles[c]);
        }
        (void)hipMemAddressFree(b.base, b.reserved);
    } else {
        (void)hipFree(b.base);
    }
    b.base = nullptr;
}

// maps `nChunks` fresh physical chunks behind one reserved range; false: virtual memory management is not usable here
bool map_range(fimex_amd_batch& b, size_t nChunks)
{
    int supported = 0;
    if (hipDeviceGetAttribute(&supported, hipDeviceAttributeVirtualMemoryManagementSupported, b.device) != hipSuccess || !supported) {
        (void)hipGetLastError();
        return false;
    }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = b.device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0 ||
        kChunkBytes % gran != 0) {
        (void)hipGetLastError();
        return false;
    }
    void* ptr = nullptr;
    const size_t bytes = nChunks * kChunkBytes;
    if (hipMemAddressReserve(&ptr, bytes, 0, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
    b.vmm = true;
    b.base = static_cast<char*>(ptr);
    b.reserved = bytes;
    b.chunkBytes = kChunkBytes;
    b.handles.assign(nChunks, hipMemGenericAllocationHandle_t{});
    b.mapped.assign(nChunks, 0);
    for (size_t c = 0; c < nChunks; ++c) {
        if (hipMemCreate(&b.handles[c], kChunkBytes, &prop, 0) != hipSuccess) { (void)hipGetLastError(); release_batch(b); return false; }
        if (hipMemMap(b.base + c * kChunkBytes, kChunkBytes, 0, b.handles[c], 0) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipMemRelease(b.handles[c]);
            release_batch(b);
            return false;
        }
        b.mapped[c] = 1;
    }
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = b.device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(b.base, bytes, &acc, 1) != hipSuccess) { (void)hipGetLastError(); release_batch(b); return false; }
    return true;
}

}  // namespace

fimex_amd_batch* batch_alloc(const fimex_amd_regrid_plan& plan, const float* d_in, size_t nz, int positions, hipStream_t stream)
{
    FA_REQUIRE(positions >= 1 && positions <= FIMEX_AMD_BATCH_MAX_POSITIONS, "positions must be 1 .. 16");
    const size_t bytes = nz * plan.outX * plan.outY * sizeof(float);
    FA_REQUIRE(bytes > 0, "empty batch");
    const auto t0 = std::chrono::steady_clock::now();
    auto b = std::make_unique<fimex_amd_batch>();
    b->device = plan.device;
    const size_t window = ceil_div(bytes, kChunkBytes);
    size_t stepBytes = kStepChunks * kChunkBytes;
    // the whole range has to fit beside what the caller holds: fewer windows rather than a failure
    size_t freeB = 0, totalB = 0;
    FA_HIP(hipMemGetInfo(&freeB, &totalB));
    while (positions > 1 && (window + (size_t)(positions - 1) * kStepChunks) * kChunkBytes + (size_t(1) << 30) > freeB) --positions;
    const size_t nChunks = window + (size_t)(positions - 1) * kStepChunks;
    if (!map_range(*b, nChunks)) {  // no virtual memory management: one plain allocation that stays whole
        b->vmm = false;
        void* p = nullptr;
        FA_HIP(hipMalloc(&p, nChunks * kChunkBytes));
        b->base = static_cast<char*>(p);
        b->reserved = nChunks * kChunkBytes;
        b->chunkBytes = kChunkBytes;
    }
    fimex_amd_batch_info& info = b->info;
    info.bytes = bytes;
    info.bytesProbed = nChunks * kChunkBytes;
    info.positions = positions;
    info.stepBytes = stepBytes;
    info.chosen = 0;
    try {
        if (positions > 1) {
            FA_REQUIRE(d_in != nullptr, "the probing regrids the caller's source batch: d_in is NULL");
            hipEvent_t e0, e1;
            FA_HIP(hipEventCreate(&e0));
            FA_HIP(hipEventCreate(&e1));
            try {
                for (int k = 0; k < positions; ++k) {
                    float* w = reinterpret_cast<float*>(b->base + (size_t)k * stepBytes);
                    float ms[3];
                    for (int rep = 0; rep < 4; ++rep) {  // one launch to settle, three timed: the median counts
                        FA_HIP(hipEventRecord(e0, stream));
                        apply_plan_device(plan, d_in, nz, w, stream);
                        FA_HIP(hipEventRecord(e1, stream));
                        FA_HIP(hipEventSynchronize(e1));
                        if (rep > 0) FA_HIP(hipEventElapsedTime(&ms[rep - 1], e0, e1));
                    }
                    std::sort(ms, ms + 3);
                    info.msAtPosition[k] = ms[1];
                    if (ms[1] < info.msAtPosition[info.chosen]) info.chosen = k;
                }
            } catch (...) {
                (void)hipEventDestroy(e0);
                (void)hipEventDestroy(e1);
                throw;
            }
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
        }
        const size_t first = (size_t)info.chosen * kStepChunks;
        info.d_data = b->base + first * kChunkBytes;
        info.bytesHeld = b->reserved;
        if (b->vmm) {  // the physical memory of every other window goes back to the driver
            FA_HIP(hipStreamSynchronize(stream));
            for (size_t c = 0; c < nChunks; ++c) {
                if (c >= first && c < first + window) continue;
                FA_HIP(hipMemUnmap(b->base + c * kChunkBytes, kChunkBytes));
                FA_HIP(hipMemRelease(b->handles[c]));
                b->mapped[c] = 0;
            }
            info.bytesHeld = window * kChunkBytes;
        }
    } catch (...) {
        release_batch(*b);
        throw;
    }
    info.trimmed = b->vmm ? 1 : 0;
    info.probeSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return b.release();
}

// The SOURCE batch [nz][inY][inX] of a resident pipeline.  Which allocation the source slices lie in moves the apply launch more than
// the output's does (profiles/calib/r03_placement_matrix.jsonl: five source batches x five output batches, each its own allocation:
// 2.32-2.34 ms with three of the sources, 2.42-2.47 ms with the other two, whatever the output), and windows a few hundred MiB apart
// share most of their memory for a batch of this size: the candidates are whole allocations here (hipMalloc: one region each), all
// held at once so that they are different memory, filled with zeros, timed with the plan's launch into a scratch output; the
// fastest stays, the others are freed before the call returns.
fimex_amd_batch* batch_alloc_source(const fimex_amd_regrid_plan& plan, size_t nz, int candidates, hipStream_t stream)
{
    FA_REQUIRE(candidates >= 1 && candidates <= FIMEX_AMD_BATCH_MAX_POSITIONS, "candidates must be 1 .. 16");
    const size_t bytes = nz * plan.inX * plan.inY * sizeof(float), outBytes = nz * plan.outX * plan.outY * sizeof(float);
    FA_REQUIRE(bytes > 0, "empty batch");
    const auto t0 = std::chrono::steady_clock::now();
    size_t freeB = 0, totalB = 0;
    FA_HIP(hipMemGetInfo(&freeB, &totalB));
    while (candidates > 1 && (size_t)candidates * bytes + outBytes + (size_t(2) << 30) > freeB) --candidates;
    std::vector<void*> cand((size_t)candidates, nullptr);
    void* scratch = nullptr;
    auto b = std::make_unique<fimex_amd_batch>();
    b->device = plan.device;
    fimex_amd_batch_info& info = b->info;
    info.bytes = bytes;
    info.positions = candidates;
    info.chosen = 0;
    auto release = [&](int keep) {
        for (int c = 0; c < (int)cand.size(); ++c)
            if (c != keep && cand[c]) { (void)hipFree(cand[c]); cand[c] = nullptr; }
        if (scratch) { (void)hipFree(scratch); scratch = nullptr; }
    };
    try {
        for (int c = 0; c < candidates; ++c) {
            FA_HIP(hipMalloc(&cand[c], bytes));
            FA_HIP(hipMemsetAsync(cand[c], 0, bytes, stream));
        }
        info.bytesProbed = (size_t)candidates * bytes;
        if (candidates > 1) {
            FA_HIP(hipMalloc(&scratch, outBytes));
            info.bytesProbed += outBytes;
            hipEvent_t e0, e1;
            FA_HIP(hipEventCreate(&e0));
            FA_HIP(hipEventCreate(&e1));
            try {
                for (int c = 0; c < candidates; ++c) {
                    float ms[3];
                    for (int rep = 0; rep < 4; ++rep) {
                        FA_HIP(hipEventRecord(e0, stream));
                        apply_plan_device(plan, static_cast<const float*>(cand[c]), nz, static_cast<float*>(scratch), stream);
                        FA_HIP(hipEventRecord(e1, stream));
                        FA_HIP(hipEventSynchronize(e1));
                        if (rep > 0) FA_HIP(hipEventElapsedTime(&ms[rep - 1], e0, e1));
                    }
                    std::sort(ms, ms + 3);
                    info.msAtPosition[c] = ms[1];
                    if (ms[1] < info.msAtPosition[info.chosen]) info.chosen = c;
                }
            } catch (...) {
                (void)hipEventDestroy(e0);
                (void)hipEventDestroy(e1);
                throw;
            }
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
        }
        FA_HIP(hipStreamSynchronize(stream));
    } catch (...) {
        release(-1);
        throw;
    }
    release(info.chosen);
    b->vmm = false;
    b->base = static_cast<char*>(cand[(size_t)info.chosen]);
    b->reserved = bytes;
    info.d_data = b->base;
    info.bytesHeld = bytes;
    info.stepBytes = 0;
    info.trimmed = 1;
    info.probeSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return b.release();
}

const fimex_amd_batch_info& batch_info(const fimex_amd_batch& b) { return b.info; }

void batch_free(fimex_amd_batch* b)
{
    if (!b) return;
    ScopedDevice dev(b->device);
    release_batch(*b);
    delete b;
}

}  // namespace fimex_amd
