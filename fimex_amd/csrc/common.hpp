// Shared plumbing of libfimex_amd.so: error reporting, device buffers, launch geometry.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <mutex>
#include <set>
#include <string>
#include <utility>

#include "../../include/fimex_amd.h"

namespace fimex_amd {

// thrown inside the library, converted to FIMEX_AMD_ERROR + last_error at the C boundary
struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

void set_last_error(const std::string& msg);

#define FA_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t fa_err_ = (expr);                                                              \
        if (fa_err_ != hipSuccess)                                                                \
            throw ::fimex_amd::Error(std::string(#expr) + ": " + hipGetErrorString(fa_err_));     \
    } while (0)

#define FA_REQUIRE(cond, msg)                                                                     \
    do {                                                                                          \
        if (!(cond)) throw ::fimex_amd::Error(msg);                                               \
    } while (0)

// runs fn(), maps exceptions to the C return convention
template <typename F>
inline int c_guard(F&& fn) noexcept
{
    try {
        fn();
        return FIMEX_AMD_OK;
    } catch (const std::exception& e) {
        set_last_error(e.what());
    } catch (...) {
        set_last_error("unknown error");
    }
    return FIMEX_AMD_ERROR;
}

// hipMalloc'ed array, freed on scope exit
template <typename T>
class DeviceArray {
public:
    DeviceArray() = default;
    explicit DeviceArray(size_t n) { allocate(n); }
    DeviceArray(const DeviceArray&) = delete;
    DeviceArray& operator=(const DeviceArray&) = delete;
    DeviceArray(DeviceArray&& o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
    DeviceArray& operator=(DeviceArray&& o) noexcept
    {
        if (this != &o) { release(); p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; }
        return *this;
    }
    ~DeviceArray() { release(); }
    void allocate(size_t n)
    {
        release();
        if (n) FA_HIP(hipMalloc(reinterpret_cast<void**>(&p_), n * sizeof(T)));
        n_ = n;
    }
    void release() noexcept
    {
        if (p_) (void)hipFree(p_);
        p_ = nullptr;
        n_ = 0;
    }
    T* get() const { return p_; }
    size_t size() const { return n_; }
    size_t bytes() const { return n_ * sizeof(T); }

private:
    T* p_ = nullptr;
    size_t n_ = 0;
};

// stream owned for the duration of one *_host call (re-entrancy: one stream per call)
class ScopedStream {
public:
    ScopedStream() { FA_HIP(hipStreamCreateWithFlags(&s_, hipStreamNonBlocking)); }
    ~ScopedStream() { if (s_) (void)hipStreamDestroy(s_); }
    ScopedStream(const ScopedStream&) = delete;
    ScopedStream& operator=(const ScopedStream&) = delete;
    hipStream_t get() const { return s_; }
    void sync() const { FA_HIP(hipStreamSynchronize(s_)); }

private:
    hipStream_t s_ = nullptr;
};

// switches the calling thread to `device` for one call and back afterwards
class ScopedDevice {
public:
    explicit ScopedDevice(int device)
    {
        FA_HIP(hipGetDevice(&prev_));
        if (prev_ != device) { FA_HIP(hipSetDevice(device)); changed_ = true; }
    }
    ~ScopedDevice() { if (changed_) (void)hipSetDevice(prev_); }

private:
    int prev_ = 0;
    bool changed_ = false;
};

int current_device_checked();                 // throws when no gfx950 device is usable
void require_current_device(int planDevice);  // *_device calls must run on the plan's device

inline hipStream_t as_stream(void* s) { return static_cast<hipStream_t>(s); }

// diagnostics that make a kernel skip its loads or stores exist in the tuning build only
#ifdef FIMEX_AMD_TUNING
constexpr bool kTuningBuild = true;
#else
constexpr bool kTuningBuild = false;
#endif

constexpr int kWave = 64;    // CDNA wavefront
constexpr int kXcds = 8;     // MI355X accelerator complex dies, one L2 each
constexpr int kBlock = 256;  // 4 waves, one per SIMD

inline size_t ceil_div(size_t a, size_t b) { return (a + b - 1) / b; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: set it once per device and kernel
// (a process may drive several devices through fimex_amd_set_device; the calls are re-entrant)
inline void allow_dynamic_lds(const void* kernel, size_t bytes)
{
    static std::mutex mtx;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    FA_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mtx);
    if (done.count({dev, kernel})) return;
    FA_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    done.insert({dev, kernel});
}

}  // namespace fimex_amd
