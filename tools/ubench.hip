// Micro-benchmark: issue cost (cycles per wave-instruction per SIMD) of the instructions the sweeps are made of, on gfx950.
// One workgroup per CU; W waves per SIMD; each wave runs N iterations of 8 independent instances of one instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(1024) void k(double *out, long long *cyc, int iters, const float *tab)
{
    extern __shared__ float2 lds[];
    for (int i = threadIdx.x; i < 10002; i += blockDim.x) lds[i] = make_float2(tab[i & 1023], 1.0f);
    __syncthreads();
    double d0 = threadIdx.x * 1e-3, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;
    double m = 1.0000001, a = 1e-9;
    float f0 = threadIdx.x * 1e-3f, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    float fm = 1.0000001f, fa = 1e-9f;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
    unsigned r = threadIdx.x * 2654435761u + blockIdx.x;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#define R8(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), \
          "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7), \
          "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(m), "v"(a), "v"(fm), "v"(fa))
        if constexpr (OP == 0) {
#define I(n) "v_fma_f64 %" #n ", %" #n ", %24, %25\n"
            R8(I);
#undef I
        } else if constexpr (OP == 1) {
#define I(n) "v_add_f64 %" #n ", %" #n ", %25\n"
            R8(I);
#undef I
        } else if constexpr (OP == 2) {
#define I(n) "v_mul_f64 %" #n ", %" #n ", %24\n"
            R8(I);
#undef I
        } else if constexpr (OP == 3) {     // cvt f32 -> f64 (8 = f0 ..)
            asm volatile("v_cvt_f64_f32 %0, %8\nv_cvt_f64_f32 %1, %9\nv_cvt_f64_f32 %2, %10\nv_cvt_f64_f32 %3, %11\n"
                         "v_cvt_f64_f32 %4, %12\nv_cvt_f64_f32 %5, %13\nv_cvt_f64_f32 %6, %14\nv_cvt_f64_f32 %7, %15\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
                         : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7));
        } else if constexpr (OP == 4) {     // fma f32
            asm volatile("v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fm), "v"(fa));
        } else if constexpr (OP == 5) {     // int max
            asm volatile("v_max_i32 %0, %0, %8\nv_max_i32 %1, %1, %8\nv_max_i32 %2, %2, %8\nv_max_i32 %3, %3, %8\n"
                         "v_max_i32 %4, %4, %8\nv_max_i32 %5, %5, %8\nv_max_i32 %6, %6, %8\nv_max_i32 %7, %7, %8\n"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(it));
        } else if constexpr (OP == 6) {     // cvt f64 -> i32
            asm volatile("v_cvt_i32_f64 %0, %8\nv_cvt_i32_f64 %1, %9\nv_cvt_i32_f64 %2, %10\nv_cvt_i32_f64 %3, %11\n"
                         "v_cvt_i32_f64 %4, %12\nv_cvt_i32_f64 %5, %13\nv_cvt_i32_f64 %6, %14\nv_cvt_i32_f64 %7, %15\n"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7)
                         : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7));
        } else if constexpr (OP == 7) {     // cvt f32 -> i32
            asm volatile("v_cvt_i32_f32 %0, %8\nv_cvt_i32_f32 %1, %9\nv_cvt_i32_f32 %2, %10\nv_cvt_i32_f32 %3, %11\n"
                         "v_cvt_i32_f32 %4, %12\nv_cvt_i32_f32 %5, %13\nv_cvt_i32_f32 %6, %14\nv_cvt_i32_f32 %7, %15\n"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7)
                         : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7));
        } else if constexpr (OP == 8 || OP == 9) {     // LDS gather of 8-byte entries: 8 random (8) / lane-consecutive (9) reads, then one wait
            float2 e[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                r = r * 1664525u + 1013904223u;
                const unsigned idx = OP == 8 ? (r >> 8) % 10001u : (threadIdx.x + q * 64 + it) % 10001u;
                e[q] = lds[idx];
            }
#pragma unroll
            for (int q = 0; q < 8; q++) f0 += e[q].x;
        } else if constexpr (OP == 10) {    // cvt f64 -> f32
            asm volatile("v_cvt_f32_f64 %0, %8\nv_cvt_f32_f64 %1, %9\nv_cvt_f32_f64 %2, %10\nv_cvt_f32_f64 %3, %11\n"
                         "v_cvt_f32_f64 %4, %12\nv_cvt_f32_f64 %5, %13\nv_cvt_f32_f64 %6, %14\nv_cvt_f32_f64 %7, %15\n"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)
                         : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7));
        } else if constexpr (OP == 11) {    // cndmask
            asm volatile("v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(it) : "vcc");
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    const int iters = 2000, ncu = 256;
    double *out; long long *cyc; float *tab;
    CHK(hipMalloc(&out, sizeof(double) * ncu * 1024));
    CHK(hipMalloc(&cyc, sizeof(long long) * ncu));
    CHK(hipMalloc(&tab, 4096));
    CHK(hipMemset(tab, 0, 4096));
    const char *names[] = {"v_fma_f64", "v_add_f64", "v_mul_f64", "v_cvt_f64_f32", "v_fma_f32", "v_max_i32", "v_cvt_i32_f64", "v_cvt_i32_f32",
                           "lds gather b64 random x8", "lds read b64 consecutive x8", "v_cvt_f32_f64", "v_cndmask_b32"};
    void (*ks[])(double *, long long *, int, const float *) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>, k<10>, k<11>};
    for (int op = 0; op < 12; op++) {
        CHK(hipFuncSetAttribute((const void *)ks[op], hipFuncAttributeMaxDynamicSharedMemorySize, 81920));
        for (int wps : {1, 2, 4}) {
            const int threads = 64 * 4 * wps;
            hipEvent_t e0, e1;
            CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
            hipLaunchKernelGGL(ks[op], dim3(ncu), dim3(threads), 81920, 0, out, cyc, iters, tab);
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(ks[op], dim3(ncu), dim3(threads), 81920, 0, out, cyc, iters, tab);
            CHK(hipEventRecord(e1));
            CHK(hipDeviceSynchronize());
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<long long> h(ncu);
            CHK(hipMemcpy(h.data(), cyc, sizeof(long long) * ncu, hipMemcpyDeviceToHost));
            double avg = 0; for (auto v : h) avg += v; avg /= ncu;
            // s_memtime ticks at 100 MHz on gfx9: report wall-based numbers too
            const double instr_per_simd = (double)iters * 8 * wps;
            printf("%-30s waves/SIMD %d: %8.3f ms  memtime ticks/instr/SIMD %.3f  ns/instr/SIMD %.3f\n", names[op], wps, ms, avg / instr_per_simd, ms * 1e6 / instr_per_simd);
        }
    }
    return 0;
}
