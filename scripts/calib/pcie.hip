// PCIe / host-memory calibration for the *_host entry points: pageable vs pinned copies, registration cost, duplex.
// build on the box: hipcc --offload-arch=gfx950 -O2 -o /tmp/pcie scripts/calib/pcie.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t GB = 1ull << 30, n = 1 * GB;
    char *pag = (char*)malloc(n), *pag2 = (char*)malloc(n), *pin, *pin2, *d, *d2;
    memset(pag, 1, n); memset(pag2, 2, n);
    CK(hipHostMalloc((void**)&pin, n)); CK(hipHostMalloc((void**)&pin2, n));
    memset(pin, 1, n); memset(pin2, 1, n);
    CK(hipMalloc((void**)&d, n)); CK(hipMalloc((void**)&d2, n));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    CK(hipMemcpy(d, pin, n, hipMemcpyHostToDevice));
    double t;
    t = now(); CK(hipMemcpy(d, pag, n, hipMemcpyHostToDevice)); printf("{\"pageable_h2d_GBps\": %.1f}\n", n / (now() - t) / 1e9);
    t = now(); CK(hipMemcpy(pag2, d, n, hipMemcpyDeviceToHost)); printf("{\"pageable_d2h_GBps\": %.1f}\n", n / (now() - t) / 1e9);
    t = now(); CK(hipMemcpy(d, pin, n, hipMemcpyHostToDevice)); printf("{\"pinned_h2d_GBps\": %.1f}\n", n / (now() - t) / 1e9);
    t = now(); CK(hipMemcpy(pin2, d, n, hipMemcpyDeviceToHost)); printf("{\"pinned_d2h_GBps\": %.1f}\n", n / (now() - t) / 1e9);
    t = now(); CK(hipMemcpyAsync(d, pin, n, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(pin2, d2, n, hipMemcpyDeviceToHost, s2));
    CK(hipDeviceSynchronize()); printf("{\"pinned_duplex_GBps_total\": %.1f}\n", 2 * n / (now() - t) / 1e9);
    { // pageable duplex from two threads
        t = now();
        std::thread th([&] { (void)hipMemcpy(pag2, d2, n, hipMemcpyDeviceToHost); });
        (void)hipMemcpy(d, pag, n, hipMemcpyHostToDevice);
        th.join();
        printf("{\"pageable_duplex_2threads_GBps_total\": %.1f}\n", 2 * n / (now() - t) / 1e9);
    }
    t = now(); CK(hipHostRegister(pag, n, hipHostRegisterDefault)); double tr = now() - t;
    t = now(); CK(hipMemcpy(d, pag, n, hipMemcpyHostToDevice)); double tc = now() - t;
    t = now(); CK(hipHostUnregister(pag)); double tu = now() - t;
    printf("{\"register_ms_per_GB\": %.1f, \"registered_h2d_GBps\": %.1f, \"unregister_ms_per_GB\": %.1f}\n", tr * 1e3, n / tc / 1e9, tu * 1e3);
    { // host memcpy into pinned staging, 1 and 8 threads
        t = now(); memcpy(pin, pag2, n); printf("{\"memcpy_1thread_GBps\": %.1f}\n", n / (now() - t) / 1e9);
        t = now();
        std::thread ths[8];
        for (int k = 0; k < 8; ++k) ths[k] = std::thread([&, k] { memcpy(pin + k * (n / 8), pag2 + k * (n / 8), n / 8); });
        for (auto& x : ths) x.join();
        printf("{\"memcpy_8threads_GBps\": %.1f}\n", n / (now() - t) / 1e9);
    }
    return 0;
}
