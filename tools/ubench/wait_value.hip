// wait_value.hip -- does hipStreamWaitValue32 [BETA] gate a stream on a value that a KERNEL of another stream writes?
// (the question behind "one k_finalize launch per band, the parts signalled from inside the kernel": profiles/rd5l)
//   stream A: a kernel that spins ~2 ms, then writes 7 to the signal word (system-scope fence first) and its end time
//   stream B: hipStreamWaitValue32(sig == 7), then a kernel that writes its start time
// prints the attribute, the API's return codes and whether B's kernel started after A's write; a watchdog ends a hang.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            std::printf("{\"error\": \"%s: %s\"}\n", #x, hipGetErrorString(e_));         \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

__global__ void k_spin_then_signal(uint32_t *sig, unsigned long long *out, unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    out[0] = wall_clock64();
    __threadfence_system();
    atomicExch(sig, 7u);
}

__global__ void k_stamp(unsigned long long *out) { out[1] = wall_clock64(); }

int main()
{
    int can = -1;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    unsigned long long *out = nullptr;
    CK(hipMalloc(&out, 16));
    CK(hipMemset(out, 0, 16));
    int rate_khz = 0;
    CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    for (int variant = 0; variant < 3; ++variant) {  // 0: signal memory, 1: plain device memory, 2: pinned host memory
        uint32_t *sig = nullptr;
        hipError_t ea = variant == 0   ? hipExtMallocWithFlags((void **)&sig, 8, hipMallocSignalMemory)
                        : variant == 1 ? hipMalloc((void **)&sig, 8)
                                       : hipHostMalloc((void **)&sig, 8, hipHostMallocDefault);
        if (ea != hipSuccess) {
            std::printf("{\"variant\": %d, \"alloc\": \"%s\"}\n", variant, hipGetErrorString(ea));
            continue;
        }
        if (variant == 2) *sig = 0;
        else CK(hipMemset(sig, 0, 8));
        CK(hipMemset(out, 0, 16));
        CK(hipDeviceSynchronize());
        const hipError_t ew = hipStreamWaitValue32(b, sig, 7u, hipStreamWaitValueEq, 0xFFFFFFFFu);
        hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, b, out);
        hipLaunchKernelGGL(k_spin_then_signal, dim3(1), dim3(1), 0, a, sig, out, (unsigned long long)rate_khz * 2ull);  // 2 ms
        bool done = false;
        std::thread wd([&] {
            for (int i = 0; i < 500 && !done; ++i) std::this_thread::sleep_for(std::chrono::milliseconds(10));
            if (!done) {
                std::printf("{\"variant\": %d, \"wait_rc\": \"%s\", \"result\": \"HANG (watchdog)\"}\n", variant, hipGetErrorString(ew));
                std::fflush(stdout);
                std::_Exit(3);
            }
        });
        const hipError_t es = hipStreamSynchronize(b);
        (void)hipStreamSynchronize(a);
        done = true;
        wd.join();
        unsigned long long h[2] = {0, 0};
        CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
        std::printf("{\"variant\": %d, \"memory\": \"%s\", \"can_use_attr\": %d, \"wait_rc\": \"%s\", \"sync_rc\": \"%s\", \"a_wrote_at\": %llu, \"b_started_at\": %llu, "
                    "\"b_after_a\": %s, \"gap_us\": %.1f}\n",
                    variant, variant == 0 ? "hipMallocSignalMemory" : variant == 1 ? "hipMalloc" : "hipHostMalloc", can, hipGetErrorString(ew),
                    hipGetErrorString(es), h[0], h[1], h[1] >= h[0] && h[0] ? "true" : "false", h[0] ? (double)((long long)(h[1] - h[0])) * 1e3 / rate_khz : -1.0);
        std::fflush(stdout);
    }
    return 0;
}
