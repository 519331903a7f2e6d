// H2D and D2H copies on two streams: do they overlap?  registered (hipHostRegister) vs allocated (hipHostMalloc) pinned memory, 1-D vs 2-D.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t N = (size_t)1 << 30;        // 1 GiB each way
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    char *d1, *d2;
    CK(hipMalloc(&d1, N)); CK(hipMalloc(&d2, N));
    for (int mode = 0; mode < 2; mode++) {
        char *h1, *h2;
        if (mode == 0) { h1 = (char *)aligned_alloc(4096, N); h2 = (char *)aligned_alloc(4096, N); for (size_t i = 0; i < N; i += 4096) { h1[i] = 1; h2[i] = 2; } CK(hipHostRegister(h1, N, hipHostRegisterDefault)); CK(hipHostRegister(h2, N, hipHostRegisterDefault)); }
        else { CK(hipHostMalloc((void **)&h1, N, hipHostMallocDefault)); CK(hipHostMalloc((void **)&h2, N, hipHostMallocDefault)); }
        for (int dim2 = 0; dim2 < 2; dim2++) {
            const size_t w = 131072, rows = N / w / 2;         // 2-D: rows of 128 KB, every other row (pitch 256 KB) -> half the bytes
            auto h2d = [&]() { if (dim2) CK(hipMemcpy2DAsync(d1, w, h1, 2 * w, w, rows, hipMemcpyHostToDevice, s1)); else CK(hipMemcpyAsync(d1, h1, N / 2, hipMemcpyHostToDevice, s1)); };
            auto d2h = [&]() { if (dim2) CK(hipMemcpy2DAsync(h2, 2 * w, d2, w, w, rows, hipMemcpyDeviceToHost, s2)); else CK(hipMemcpyAsync(h2, d2, N / 2, hipMemcpyDeviceToHost, s2)); };
            h2d(); d2h(); CK(hipDeviceSynchronize());
            double t0 = now(); h2d(); CK(hipDeviceSynchronize()); double ta = now() - t0;
            t0 = now(); d2h(); CK(hipDeviceSynchronize()); double tb = now() - t0;
            t0 = now(); h2d(); d2h(); CK(hipDeviceSynchronize()); double tc = now() - t0;
            printf("%s %s: H2D %.2f ms (%.1f GB/s)  D2H %.2f ms (%.1f GB/s)  both at once %.2f ms\n", mode ? "hipHostMalloc  " : "hipHostRegister", dim2 ? "2-D rows" : "1-D     ",
                   ta, N / 2 / ta / 1e6, tb, N / 2 / tb / 1e6, tc);
        }
    }
    return 0;
}
