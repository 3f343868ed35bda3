// Source and output batches placed by the library (include/fimex_amd.h: fimex_amd_regrid_source_batch_alloc_device,
// fimex_amd_regrid_batch_alloc_device).
//
// The reference allocates the result of every interpolateValues call itself (src/CachedInterpolation.cc:123, `new float[newSize]`)
// and receives its input from the reader.  A device-resident caller allocates both batches once and reuses them, and on this
// memory system WHICH allocation they lie in moves the apply launch by several per cent with identical code and traffic
// (4-5 % by the source's allocation, 1-3 % by the output's; per-channel request counts are equal, the read stream
// waits longer for DRAM credits in the slow placements).  So the allocation is a service of the library: `candidates` whole
// allocations are made and held at once (so that they are different memory), the plan's own apply launch is timed with each of
// them in its role -- the caller's source batch into each output candidate, each zero-filled source candidate into a scratch
// output -- the fastest stays, the others are freed before the call returns.  What the probing cost is reported with the batch.
//
// (Round 3 first mapped the output batch at windows of one reserved address range through HIP's virtual memory management, 32 MiB
// physical chunks per window, and returned the chunks of the windows it did not keep.  Memory mapped that way ran the launch 2 %
// slower than a plain hipMalloc allocation in the same process -- 2.31 against 2.26 ms, profiles/r03_bench_runs_vmm_windows.jsonl --
// and is awkward as a send buffer for RCCL; whole allocations are simpler and faster.)
#include "plan.hpp"

#include <algorithm>
#include <chrono>
#include <functional>
#include <memory>
#include <vector>

struct fimex_amd_batch {
    int device = 0;
    void* base = nullptr;  // hipMalloc'ed
    fimex_amd_batch_info info{};
};

namespace fimex_amd {

namespace {

// `candidates` allocations of `bytes` (fewer when device memory is short), probe(candidate) timed three times each after one
// launch to settle (and three untimed launches at the very start: the clocks of a device that has just been idle, allocating,
// are not yet the ones the launch runs at, and the first candidate must not pay for that); the fastest is kept.
fimex_amd_batch* alloc_best(const fimex_amd_regrid_plan& plan, size_t bytes, size_t extraBytes, int candidates, bool zeroFill,
                            const std::function<void(void*)>& probe, hipStream_t stream)
{
    FA_REQUIRE(candidates >= 1 && candidates <= FIMEX_AMD_BATCH_MAX_POSITIONS, "candidates must be 1 .. 16");
    FA_REQUIRE(bytes > 0, "empty batch");
    const auto t0 = std::chrono::steady_clock::now();
    size_t freeB = 0, totalB = 0;
    FA_HIP(hipMemGetInfo(&freeB, &totalB));
    while (candidates > 1 && (size_t)candidates * bytes + extraBytes + (size_t(2) << 30) > freeB) --candidates;
    std::vector<void*> cand((size_t)candidates, nullptr);
    auto b = std::make_unique<fimex_amd_batch>();
    b->device = plan.device;
    fimex_amd_batch_info& info = b->info;
    info.bytes = bytes;
    info.positions = candidates;
    info.chosen = 0;
    auto release = [&](int keep) {
        for (int c = 0; c < (int)cand.size(); ++c)
            if (c != keep && cand[(size_t)c]) { (void)hipFree(cand[(size_t)c]); cand[(size_t)c] = nullptr; }
    };
    try {
        // tuning build: BATCH_CONTIGUOUS 1 = every candidate, 2 = every other one physically contiguous (hipDeviceMallocContiguous).
        // Measured (profiles/calib/r03_bench_contiguous_candidates.jsonl): contiguous OUTPUT batches are the slow kind throughout
        // (2.33 against 2.26-2.28 ms for their plain neighbours), contiguous source batches are fast or slow like plain ones.
        const int contiguous = tuning("BATCH_CONTIGUOUS", 0);
        for (int c = 0; c < candidates; ++c) {
            if (contiguous == 1 || (contiguous == 2 && (c & 1))) {
                if (hipExtMallocWithFlags(&cand[(size_t)c], bytes, hipDeviceMallocContiguous) != hipSuccess) {
                    (void)hipGetLastError();
                    cand[(size_t)c] = nullptr;
                }
            }
            if (!cand[(size_t)c]) FA_HIP(hipMalloc(&cand[(size_t)c], bytes));
            if (zeroFill) FA_HIP(hipMemsetAsync(cand[(size_t)c], 0, bytes, stream));
        }
        info.bytesProbed = (size_t)candidates * bytes + (candidates > 1 ? extraBytes : 0);
        if (candidates > 1) {
            hipEvent_t e0, e1;
            FA_HIP(hipEventCreate(&e0));
            FA_HIP(hipEventCreate(&e1));
            try {
                for (int rep = 0; rep < 3; ++rep) probe(cand[0]);
                for (int c = 0; c < candidates; ++c) {
                    float ms[3];
                    for (int rep = 0; rep < 4; ++rep) {
                        FA_HIP(hipEventRecord(e0, stream));
                        probe(cand[(size_t)c]);
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
    b->base = cand[(size_t)info.chosen];
    info.d_data = b->base;
    info.bytesHeld = bytes;
    info.stepBytes = 0;
    info.trimmed = 1;
    info.probeSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return b.release();
}

}  // namespace

// the output batch [nz][outY][outX]: the caller's source batch regridded into every candidate
fimex_amd_batch* batch_alloc(const fimex_amd_regrid_plan& plan, const float* d_in, size_t nz, int positions, hipStream_t stream)
{
    FA_REQUIRE(positions == 1 || d_in != nullptr, "the probing regrids the caller's source batch: d_in is NULL");
    const size_t bytes = nz * plan.outX * plan.outY * sizeof(float);
    return alloc_best(plan, bytes, 0, positions, false,
                      [&](void* out) { apply_plan_device(plan, d_in, nz, static_cast<float*>(out), stream); }, stream);
}

// the source batch [nz][inY][inX]: every zero-filled candidate regridded into a scratch output
fimex_amd_batch* batch_alloc_source(const fimex_amd_regrid_plan& plan, size_t nz, int candidates, hipStream_t stream)
{
    const size_t bytes = nz * plan.inX * plan.inY * sizeof(float), outBytes = nz * plan.outX * plan.outY * sizeof(float);
    DeviceArray<float> scratch;
    if (candidates > 1) scratch.allocate(outBytes / sizeof(float));
    return alloc_best(plan, bytes, outBytes, candidates, true,
                      [&](void* in) { apply_plan_device(plan, static_cast<const float*>(in), nz, scratch.get(), stream); }, stream);
}

const fimex_amd_batch_info& batch_info(const fimex_amd_batch& b) { return b.info; }

void batch_free(fimex_amd_batch* b)
{
    if (!b) return;
    ScopedDevice dev(b->device);
    if (b->base) (void)hipFree(b->base);
    delete b;
}

}  // namespace fimex_amd
