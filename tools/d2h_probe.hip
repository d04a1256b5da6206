// tools/d2h_probe.hip — how should rt_fire_all_rays bring a 24.9 MB image back to a pageable caller buffer?
// hipcc --offload-arch=gfx950 -O2 -o tools/d2h_probe tools/d2h_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t n = 1920ull * 1080 * 3 * 4;
    float* d; CK(hipMalloc(&d, n)); CK(hipMemset(d, 1, n));
    char* pageable = (char*)malloc(n); memset(pageable, 0, n);
    char* pinned; CK(hipHostMalloc(&pinned, n, hipHostMallocDefault));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now(); CK(hipMemcpy(pageable, d, n, hipMemcpyDeviceToHost)); double t1 = now();
        printf("pageable hipMemcpy            %.3f ms\n", t1 - t0);
        t0 = now(); CK(hipMemcpyAsync(pinned, d, n, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); t1 = now();
        printf("pinned D2H                    %.3f ms\n", t1 - t0);
        t0 = now(); memcpy(pageable, pinned, n); t1 = now();
        printf("memcpy pinned->pageable 1 thr %.3f ms\n", t1 - t0);
        for (int T : {2, 4, 8}) {
            t0 = now();
            std::vector<std::thread> th;
            for (int i = 0; i < T; ++i) th.emplace_back([=] { memcpy(pageable + n / T * i, pinned + n / T * i, n / T); });
            for (auto& t : th) t.join();
            t1 = now();
            printf("memcpy pinned->pageable %d thr %.3f ms\n", T, t1 - t0);
        }
        // chunked pipeline: D2H chunk c async, copy thread per chunk waits its event
        {
            const int K = 4; hipEvent_t ev[K];
            for (int c = 0; c < K; ++c) CK(hipEventCreateWithFlags(&ev[c], hipEventDisableTiming));
            t0 = now();
            for (int c = 0; c < K; ++c) { CK(hipMemcpyAsync(pinned + n / K * c, (char*)d + n / K * c, n / K, hipMemcpyDeviceToHost, st)); CK(hipEventRecord(ev[c], st)); }
            std::vector<std::thread> th;
            for (int c = 0; c < K; ++c) th.emplace_back([=] { (void)hipEventSynchronize(ev[c]); memcpy(pageable + n / K * c, pinned + n / K * c, n / K); });
            for (auto& t : th) t.join();
            t1 = now();
            printf("pipelined 4 chunks + 4 thr    %.3f ms\n", t1 - t0);
            for (int c = 0; c < K; ++c) (void)hipEventDestroy(ev[c]);
        }
    }
    // allocation costs
    double t0 = now(); float* x; CK(hipMalloc(&x, n)); double t1 = now(); CK(hipFree(x)); double t2 = now();
    printf("hipMalloc 24.9MB %.3f ms, hipFree %.3f ms\n", t1 - t0, t2 - t1);
    t0 = now(); for (int i = 0; i < 10; ++i) { CK(hipMalloc(&x, 4096)); CK(hipFree(x)); } t1 = now();
    printf("10x hipMalloc+hipFree 4KB %.3f ms\n", t1 - t0);
    return 0;
}
