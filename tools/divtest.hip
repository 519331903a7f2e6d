// accuracy of the reciprocal-based quotients used for table indices (k_layer's fdiv) against the IEEE division, on the device
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ double fdiv2(double a, double b)
{
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    r = fma(fma(-b, r, 1.0), r, r);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}
__device__ __forceinline__ double fdiv1(double a, double b)
{
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}
__device__ unsigned long long rng(unsigned long long &s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
__global__ void k(unsigned long long *out, int iters)
{
    unsigned long long s = 0x9E3779B97F4A7C15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    unsigned long long bad_q2 = 0, bad_q1 = 0, bad_i2 = 0, bad_i1 = 0, max_rcp = 0;
    const double bpade = 1.0 / 0.278;
    for (int i = 0; i < iters; i++) {
        const double u = (double)(rng(s) >> 11) * (1.0 / 9007199254740992.0);
        const double od = 0.06 * exp(u * 12.0);            // 0.06 .. 1e4
        const double b = bpade + od;
        const double qe = od / b, q2 = fdiv2(od, b), q1 = fdiv1(od, b);
        bad_q2 += q2 != qe; bad_q1 += q1 != qe;
        const int ie = (int)(10000.0 * qe + 0.5), i2 = (int)(10000.0 * q2 + 0.5), i1 = (int)(10000.0 * q1 + 0.5);
        bad_i2 += i2 != ie; bad_i1 += i1 != ie;
        const double r0 = __builtin_amdgcn_rcp(b);
        const double e = fabs(fma(-b, r0, 1.0));
        const unsigned long long eb = (unsigned long long)(e * 1.8446744073709552e19);   // e * 2^64
        if (eb > max_rcp) max_rcp = eb;
    }
    atomicAdd(&out[0], bad_q2); atomicAdd(&out[1], bad_q1); atomicAdd(&out[2], bad_i2); atomicAdd(&out[3], bad_i1); atomicMax(&out[4], max_rcp);
}
int main()
{
    unsigned long long *d, h[5] = {0, 0, 0, 0, 0};
    hipMalloc(&d, sizeof(h)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    const int blocks = 4096, threads = 256, iters = 2000;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const double n = (double)blocks * threads * iters;
    printf("samples %.3g\nquotient != IEEE: two Newton steps %llu (%.3g), one %llu (%.3g)\ntable index != IEEE index: two steps %llu, one step %llu\nmax |1 - b rcp(b)| = %.3g (2^%.1f)\n",
           n, h[0], h[0] / n, h[1], h[1] / n, h[2], h[3], h[4] / 1.8446744073709552e19, log2(h[4] / 1.8446744073709552e19));
    return 0;
}
