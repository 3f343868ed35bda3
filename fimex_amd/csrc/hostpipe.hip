// The *_host entry points move caller-owned, pageable buffers over PCIe: that, not the kernels, is what a drop-in caller
// waits for (measured on this pool, scripts/calib/pcie.hip: pageable hipMemcpy 24 GB/s up / 32 GB/s down, pinned 57 GB/s
// each way and 75 GB/s duplex, hipHostRegister 50 ms per GB -- not worth it per call, host memcpy 30 GB/s on one thread,
// 127 GB/s on eight).  So slices are streamed: a few worker threads copy the next chunk into pinned staging while the DMA
// engines upload the previous one, the kernels run, and results flow back the same way in the other direction.
#include "plan.hpp"

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace fimex_amd {

namespace {

constexpr size_t kPinBytes = (size_t)64 << 20;   // pinned staging per slot and direction
constexpr size_t kDevFloatBytes = (size_t)128 << 20;  // float scratch per slot and direction (typed slices)
constexpr int kSlots = 3;
constexpr size_t kPiece = (size_t)8 << 20;            // a slot's transfer goes in pieces: host memcpy of one overlaps the DMA of the other
constexpr int kPieces = (int)(kPinBytes / kPiece);
constexpr int kMaxCached = 4;  // idle pipes kept for the next call
constexpr int kMaxLive = 8;    // pipes in use at once: further callers wait for one (each pins up to 384 MB of host memory)

// memcpy on several threads: the workers live as long as the pipe
class ParallelCopier {
public:
    explicit ParallelCopier(int workers)
    {
        for (int i = 0; i < workers; ++i) threads_.emplace_back([this, i] { run(i); });
    }
    ~ParallelCopier()
    {
        { std::lock_guard<std::mutex> l(m_); stop_ = true; ++generation_; }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    void copy(void* dst, const void* src, size_t bytes)
    {
        const size_t parts = threads_.size() + 1;
        if (bytes < ((size_t)1 << 20) || threads_.empty()) { std::memcpy(dst, src, bytes); return; }
        const size_t per = ((bytes + parts - 1) / parts + 4095) & ~(size_t)4095;
        {
            std::lock_guard<std::mutex> l(m_);
            dst_ = static_cast<char*>(dst);
            src_ = static_cast<const char*>(src);
            bytes_ = bytes;
            per_ = per;
            pending_ = (int)threads_.size();
            ++generation_;
        }
        cv_.notify_all();
        part(threads_.size());  // the caller takes the last part
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [this] { return pending_ == 0; });
    }

private:
    void part(size_t k)
    {
        const size_t off = k * per_;
        if (off < bytes_) std::memcpy(dst_ + off, src_ + off, std::min(per_, bytes_ - off));
    }
    void run(int k)
    {
        unsigned long long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return generation_ != seen; });
                seen = generation_;
                if (stop_) return;
            }
            part((size_t)k);
            std::lock_guard<std::mutex> l(m_);
            if (--pending_ == 0) done_.notify_one();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    unsigned long long generation_ = 0;
    bool stop_ = false;
    char* dst_ = nullptr;
    const char* src_ = nullptr;
    size_t bytes_ = 0, per_ = 0;
    int pending_ = 0;
};

struct Slot {
    char *pinIn = nullptr, *pinOut = nullptr, *dRawIn = nullptr, *dRawOut = nullptr;
    float *dFIn = nullptr, *dFOut = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    hipEvent_t piece[kPieces] = {};   // piece p of the result has arrived in pinOut
};

// Buffers are made on first use and only the kind a call needs: host_to_device touches pinIn alone, a float regrid the four
// raw buffers, only a conversion of stored types the float scratch (3 x 256 MB of HBM otherwise held for nothing).
class HostPipe {
public:
    explicit HostPipe(int device) : device(device), copier(worker_count())
    {
        for (Slot& s : slots) {
            FA_HIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
            FA_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
            for (hipEvent_t& e : s.piece) FA_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
    }
    ~HostPipe()
    {
        for (Slot& s : slots) {
            if (s.stream) (void)hipStreamSynchronize(s.stream);
            if (s.pinIn) (void)hipHostFree(s.pinIn);
            if (s.pinOut) (void)hipHostFree(s.pinOut);
            if (s.dRawIn) (void)hipFree(s.dRawIn);
            if (s.dRawOut) (void)hipFree(s.dRawOut);
            if (s.dFIn) (void)hipFree(s.dFIn);
            if (s.dFOut) (void)hipFree(s.dFOut);
            if (s.done) (void)hipEventDestroy(s.done);
            for (hipEvent_t e : s.piece) if (e) (void)hipEventDestroy(e);
            if (s.stream) (void)hipStreamDestroy(s.stream);
        }
    }
    enum Need { PinIn = 1, PinOut = 2, RawIn = 4, RawOut = 8, FloatIn = 16, FloatOut = 32 };
    void ensure(Slot& s, int need)
    {
        if ((need & PinIn) && !s.pinIn) FA_HIP(hipHostMalloc(reinterpret_cast<void**>(&s.pinIn), kPinBytes));
        if ((need & PinOut) && !s.pinOut) FA_HIP(hipHostMalloc(reinterpret_cast<void**>(&s.pinOut), kPinBytes));
        if ((need & RawIn) && !s.dRawIn) FA_HIP(hipMalloc(reinterpret_cast<void**>(&s.dRawIn), kPinBytes));
        if ((need & RawOut) && !s.dRawOut) FA_HIP(hipMalloc(reinterpret_cast<void**>(&s.dRawOut), kPinBytes));
        if ((need & FloatIn) && !s.dFIn) FA_HIP(hipMalloc(reinterpret_cast<void**>(&s.dFIn), kDevFloatBytes));
        if ((need & FloatOut) && !s.dFOut) FA_HIP(hipMalloc(reinterpret_cast<void**>(&s.dFOut), kDevFloatBytes));
    }
    static int worker_count()
    {
        const int forced = tuning("HOST_COPY_THREADS", 0);
        if (forced > 0) return forced - 1;
        const unsigned hw = std::thread::hardware_concurrency();
        return (int)std::min(7u, std::max(1u, hw / 4));
    }
    int device;
    Slot slots[kSlots];
    ParallelCopier copier;
};

std::mutex g_poolMutex;
std::condition_variable g_poolFree;
std::vector<HostPipe*> g_pool;
int g_live = 0;  // pipes handed out

HostPipe* acquire_pipe(int device)
{
    {
        std::unique_lock<std::mutex> l(g_poolMutex);
        // concurrent callers (interpolateValues is re-entrant) get a pipe each, up to kMaxLive; the next one waits
        g_poolFree.wait(l, [] { return g_live < kMaxLive; });
        ++g_live;
        for (size_t i = 0; i < g_pool.size(); ++i)
            if (g_pool[i]->device == device) {
                HostPipe* p = g_pool[i];
                g_pool.erase(g_pool.begin() + (long)i);
                return p;
            }
    }
    try {
        return new HostPipe(device);
    } catch (...) {
        { std::lock_guard<std::mutex> l(g_poolMutex); --g_live; }
        g_poolFree.notify_one();
        throw;
    }
}

void release_pipe(HostPipe* p)
{
    bool keep = false;
    {
        std::lock_guard<std::mutex> l(g_poolMutex);
        --g_live;
        if ((int)g_pool.size() < kMaxCached) { g_pool.push_back(p); keep = true; }
    }
    g_poolFree.notify_one();
    if (!keep) delete p;
}

struct PipeLease {
    explicit PipeLease(int device) : pipe(acquire_pipe(device)) {}
    ~PipeLease() { release_pipe(pipe); }
    HostPipe* pipe;
};

}  // namespace

// in / out: nz slices of inSliceBytes / outSliceBytes in caller memory.  fn(dRawIn, dRawOut, dFIn, dFOut, nzc, stream) turns
// nzc uploaded slices into nzc result slices in dRawOut (dFIn / dFOut: float scratch of inSliceFloats / outSliceFloats per
// slice).  Returns false when a single slice does not fit the staging buffers (the caller takes the plain path).
bool pipelined_slices(int device, const void* in, size_t inSliceBytes, void* out, size_t outSliceBytes, size_t inSliceFloats,
                      size_t outSliceFloats, size_t nz, const SliceChunkFn& fn)
{
    if (tuning("HOST_PIPE", 1) == 0 || nz == 0) return false;
    size_t nzc = std::min({kPinBytes / std::max<size_t>(inSliceBytes, 1), kPinBytes / std::max<size_t>(outSliceBytes, 1),
                           kDevFloatBytes / std::max<size_t>(inSliceFloats * 4, 1), kDevFloatBytes / std::max<size_t>(outSliceFloats * 4, 1)});
    if (nzc == 0) return false;
    nzc = std::min(nzc, nz);
    PipeLease lease(device);
    HostPipe& p = *lease.pipe;
    const int need = HostPipe::PinIn | HostPipe::PinOut | HostPipe::RawIn | HostPipe::RawOut | (inSliceFloats ? HostPipe::FloatIn : 0) |
                     (outSliceFloats ? HostPipe::FloatOut : 0);
    const size_t nChunks = (nz + nzc - 1) / nzc;
    for (size_t c = 0; c < std::min<size_t>(nChunks, kSlots); ++c) p.ensure(p.slots[c], need);
    const char* src = static_cast<const char*>(in);
    char* dst = static_cast<char*>(out);
    auto count = [&](size_t c) { return std::min(nzc, nz - c * nzc); };
    auto finish = [&](size_t c) {  // piece by piece: the host copy of one piece overlaps the DMA of the next
        Slot& s = p.slots[c % kSlots];
        const size_t bytes = count(c) * outSliceBytes;
        char* to = dst + c * nzc * outSliceBytes;
        for (size_t off = 0, q = 0; off < bytes; off += kPiece, ++q) {
            FA_HIP(hipEventSynchronize(s.piece[q]));
            p.copier.copy(to + off, s.pinOut + off, std::min(kPiece, bytes - off));
        }
        FA_HIP(hipEventSynchronize(s.done));
    };
    size_t finished = 0;
    try {
        for (size_t c = 0; c < nChunks; ++c) {
            Slot& s = p.slots[c % kSlots];
            if (c >= (size_t)kSlots) { finish(c - kSlots); finished = c - kSlots + 1; }
            const size_t k = count(c), inBytes = k * inSliceBytes, outBytes = k * outSliceBytes;
            const char* from = src + c * nzc * inSliceBytes;
            for (size_t off = 0; off < inBytes; off += kPiece) {
                const size_t len = std::min(kPiece, inBytes - off);
                p.copier.copy(s.pinIn + off, from + off, len);
                FA_HIP(hipMemcpyAsync(s.dRawIn + off, s.pinIn + off, len, hipMemcpyHostToDevice, s.stream));
            }
            fn(s.dRawIn, s.dRawOut, s.dFIn, s.dFOut, k, s.stream);
            for (size_t off = 0, q = 0; off < outBytes; off += kPiece, ++q) {
                FA_HIP(hipMemcpyAsync(s.pinOut + off, s.dRawOut + off, std::min(kPiece, outBytes - off), hipMemcpyDeviceToHost, s.stream));
                FA_HIP(hipEventRecord(s.piece[q], s.stream));
            }
            FA_HIP(hipEventRecord(s.done, s.stream));
        }
        for (size_t c = finished; c < nChunks; ++c) finish(c);
    } catch (...) {
        for (Slot& s : p.slots) (void)hipStreamSynchronize(s.stream);  // nothing of this call may still be in flight
        throw;
    }
    return true;
}

namespace {
constexpr size_t kStagedCopyMin = (size_t)16 << 20;  // below this a plain copy is as fast
}

// hipMemcpyAsync(HostToDevice) for caller-owned (pageable) memory: large copies go through the pinned ring with parallel
// memcpy (about 2x the pageable rate) and have landed when this returns; small ones are enqueued on `stream` as before
void host_to_device(void* d_dst, const void* h_src, size_t bytes, hipStream_t stream)
{
    if (bytes == 0) return;
    if (bytes < kStagedCopyMin || tuning("HOST_PIPE", 1) == 0) {
        FA_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, stream));
        return;
    }
    int device = 0;
    FA_HIP(hipGetDevice(&device));
    PipeLease lease(device);
    HostPipe& p = *lease.pipe;
    const size_t nChunks = (bytes + kPinBytes - 1) / kPinBytes;
    for (size_t c = 0; c < std::min<size_t>(nChunks, kSlots); ++c) p.ensure(p.slots[c], HostPipe::PinIn);
    try {
        for (size_t c = 0; c < nChunks; ++c) {
            Slot& s = p.slots[c % kSlots];
            if (c >= (size_t)kSlots) FA_HIP(hipEventSynchronize(s.done));
            const size_t off = c * kPinBytes, len = std::min(kPinBytes, bytes - off);
            p.copier.copy(s.pinIn, static_cast<const char*>(h_src) + off, len);
            FA_HIP(hipMemcpyAsync(static_cast<char*>(d_dst) + off, s.pinIn, len, hipMemcpyHostToDevice, s.stream));
            FA_HIP(hipEventRecord(s.done, s.stream));
        }
        for (Slot& s : p.slots) FA_HIP(hipStreamSynchronize(s.stream));
    } catch (...) {
        for (Slot& s : p.slots) (void)hipStreamSynchronize(s.stream);
        throw;
    }
}

// the reverse; waits for the work queued on `stream` first (it produced d_src) when it takes the staged path
void device_to_host(void* h_dst, const void* d_src, size_t bytes, hipStream_t stream)
{
    if (bytes == 0) return;
    if (bytes < kStagedCopyMin || tuning("HOST_PIPE", 1) == 0) {
        FA_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, stream));
        return;
    }
    FA_HIP(hipStreamSynchronize(stream));
    int device = 0;
    FA_HIP(hipGetDevice(&device));
    PipeLease lease(device);
    HostPipe& p = *lease.pipe;
    const size_t nChunks = (bytes + kPinBytes - 1) / kPinBytes;
    for (size_t c = 0; c < std::min<size_t>(nChunks, kSlots); ++c) p.ensure(p.slots[c], HostPipe::PinOut);
    auto finish = [&](size_t c) {
        Slot& s = p.slots[c % kSlots];
        FA_HIP(hipEventSynchronize(s.done));
        const size_t off = c * kPinBytes;
        p.copier.copy(static_cast<char*>(h_dst) + off, s.pinOut, std::min(kPinBytes, bytes - off));
    };
    size_t finished = 0;
    try {
        for (size_t c = 0; c < nChunks; ++c) {
            Slot& s = p.slots[c % kSlots];
            if (c >= (size_t)kSlots) { finish(c - kSlots); finished = c - kSlots + 1; }
            const size_t off = c * kPinBytes;
            FA_HIP(hipMemcpyAsync(s.pinOut, static_cast<const char*>(d_src) + off, std::min(kPinBytes, bytes - off), hipMemcpyDeviceToHost, s.stream));
            FA_HIP(hipEventRecord(s.done, s.stream));
        }
        for (size_t c = finished; c < nChunks; ++c) finish(c);
    } catch (...) {
        for (Slot& s : p.slots) (void)hipStreamSynchronize(s.stream);
        throw;
    }
}

// frees the idle pipes (pinned host memory, device staging, copy threads); pipes in use are freed when their call returns and
// the cache is full, or by the next call of this
void release_host_pipes()
{
    std::vector<HostPipe*> idle;
    {
        std::lock_guard<std::mutex> l(g_poolMutex);
        idle.swap(g_pool);
    }
    for (HostPipe* p : idle) {
        int prev = 0;
        (void)hipGetDevice(&prev);
        (void)hipSetDevice(p->device);
        delete p;
        (void)hipSetDevice(prev);
    }
}

}  // namespace fimex_amd
