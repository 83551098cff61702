// Device-to-host copy rate of one 1080p running sum (24.9 MB) into three kinds of host memory:
//   hipcc tools/diag/d2h_rate.cpp -o /tmp/d2h_rate && /tmp/d2h_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main()
{
    const size_t bytes = 1920ull * 1080 * 12;
    float* d; CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 0, bytes));
    hipStream_t s; CK(hipStreamCreate(&s));
    float* pinned; CK(hipHostMalloc((void**)&pinned, bytes, hipHostMallocDefault));
    std::vector<char> heap(bytes + 4096);
    char* aligned = (char*)(((uintptr_t)heap.data() + 4095) & ~(uintptr_t)4095);
    std::vector<char> heap2(bytes + 64);
    char* odd = heap2.data() + 16;
    hipError_t ra = hipHostRegister(aligned, bytes, hipHostRegisterDefault);
    printf("hipHostRegister of page-aligned heap memory: %s\n", hipGetErrorString(ra));
    hipError_t ro = hipHostRegister(odd, bytes, hipHostRegisterDefault);
    printf("hipHostRegister 16 bytes past an allocation's start: %s\n", hipGetErrorString(ro));
    (void)hipGetLastError();
    struct { const char* name; void* p; } t[] = { {"hipHostMalloc", pinned}, {ra == hipSuccess ? "registered, page aligned" : "pageable, page aligned", aligned},
                                                 {ro == hipSuccess ? "registered, odd start" : "pageable, odd start", odd} };
    for (auto& k : t) {
        for (int w = 0; w < 3; ++w) { CK(hipMemcpyAsync(k.p, d, bytes, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); }
        auto t0 = std::chrono::steady_clock::now();
        const int n = 50;
        for (int i = 0; i < n; ++i) { CK(hipMemcpyAsync(k.p, d, bytes, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); }
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / n;
        printf("%-50s %.3f ms per copy, %.1f GB/s\n", k.name, ms, bytes / ms / 1e6);
    }
    {   /* host-side copy out of the pinned buffer, one thread */
        auto t0 = std::chrono::steady_clock::now();
        const int n = 50;
        for (int i = 0; i < n; ++i) { memcpy(aligned, pinned, bytes); ((volatile char*)aligned)[i] = 1; }
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / n;
        printf("%-50s %.3f ms per copy, %.1f GB/s\n", "memcpy pinned -> heap, one thread", ms, bytes / ms / 1e6);
    }
    return 0;
}
