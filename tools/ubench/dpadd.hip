// Micro-benchmark: sustained v_add_f64 issue rate on gfx950 (independent accumulators, no memory).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NACC>
__global__ __launch_bounds__(64) void dpadd(double* out, double seed, int iters) {
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = seed * (i + 1) + threadIdx.x;
    double p = seed;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = acc[i] + p;
        p = p + 1e-9;
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int NACC>
void run(int waves_per_simd) {
    int blocks = 256 * 4 * waves_per_simd;
    double* out;
    hipMalloc(&out, blocks * 64 * sizeof(double));
    int iters = 20000;
    dpadd<NACC><<<blocks, 64>>>(out, 1.0, 10);
    hipDeviceSynchronize();
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    dpadd<NACC><<<blocks, 64>>>(out, 1.0, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double adds = (double)blocks * iters * (NACC + 1);   // wave-instructions
    double per_simd = adds / 1024.0;
    printf("NACC=%d waves/SIMD=%d: %.3f ms, %.2f Tadd/s (lane adds), %.2f ns per wave-instr per SIMD (x2.4GHz = %.2f cycles)\n", NACC,
           waves_per_simd, ms, adds * 64 / (ms * 1e-3) / 1e12, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
    hipFree(out);
}
int main() {
    run<51>(1); run<51>(2); run<51>(3); run<26>(4); run<8>(8);
    return 0;
}
