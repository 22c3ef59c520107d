// Issue rate of a few VALU instructions on gfx950, one wave per SIMD and eight: every thread runs a long unrolled stream of
// INDEPENDENT instances (16 accumulators) of one instruction; cycles per wave-instruction per SIMD = waves_on_simd * clocks
// / instructions.  tools/ab/ubench.sh runs it; the numbers are quoted in DESIGN.md section 4.5.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed, long long* clocks) {
    uint32_t a[16];
    double d[16];
    for (int i = 0; i < 16; i++) a[i] = seed * (threadIdx.x + 3 + i), d[i] = (double)a[i];
    uint32_t b = seed + threadIdx.x;
    uint64_t acc = 0;
    double bd = (double)b;
    const uint64_t msk = __ballot((threadIdx.x ^ seed) & 1);
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#define U32MIN(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define U32MAX(i) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define F64MIN(i) asm volatile("v_min_f64 %0, %0, %1" : "+v"(d[i]) : "v"(bd));
#define F64FMA(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(bd));
#define F32FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
#define F32ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define U32ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(msk));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[i]) : "v"(bd));
#define CMPF64(i) { uint64_t m_; asm volatile("v_cmp_gt_f64 %0, %1, %2" : "=s"(m_) : "v"(d[i]), "v"(bd)); acc ^= m_; }
#define MIN3(i) asm volatile("v_min3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
#define ADDF64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(bd));
#define MOVDPP(i) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
#define LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b));
#define MADU24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define CMPF32(i) { uint64_t m_; asm volatile("v_cmp_ge_f32 %0, %1, %2" : "=s"(m_) : "v"(a[i]), "v"(b)); acc ^= m_; }
#define CMPU32(i) { uint64_t m_; asm volatile("v_cmp_gt_u32 %0, %1, %2" : "=s"(m_) : "v"(a[i]), "v"(b)); acc ^= m_; }
// the compare alone: results to vcc, nothing reads them until the end of the block of 16
#define CMPF64V(i) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(d[i]), "v"(bd) : "vcc");
#define CMPF32V(i) asm volatile("v_cmp_ge_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
// compare + the conditional move that consumes it (what an admission test followed by a select costs)
#define CMPSEL64(i) asm volatile("v_cmp_gt_f64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(a[i]) : "v"(d[i]), "v"(bd), "v"(b) : "vcc");
#define CMPSEL32(i) asm volatile("v_cmp_ge_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
        if constexpr (OP == 0) { REP16(U32MIN) REP16(U32MAX) }
        if constexpr (OP == 1) { REP16(F64MIN) REP16(F64MIN) }
        if constexpr (OP == 2) { REP16(F64FMA) REP16(F64FMA) }
        if constexpr (OP == 3) { REP16(F32FMA) REP16(F32FMA) }
        if constexpr (OP == 4) { REP16(F32ADD) REP16(F32ADD) }
        if constexpr (OP == 5) { REP16(U32ADD) REP16(U32ADD) }
        if constexpr (OP == 6) { REP16(CNDMASK) REP16(CNDMASK) }
        if constexpr (OP == 7) { REP16(PKFMA) REP16(PKFMA) }
        if constexpr (OP == 8) { REP16(CMPF64) REP16(CMPF64) }
        if constexpr (OP == 9) { REP16(MOVDPP) REP16(MOVDPP) }
        if constexpr (OP == 10) { REP16(LSHLOR) REP16(LSHLOR) }
        if constexpr (OP == 11) { REP16(MADU24) REP16(MADU24) }
        if constexpr (OP == 12) { REP16(MULLO) REP16(MULLO) }
        if constexpr (OP == 13) { REP16(MIN3) REP16(MIN3) }
        if constexpr (OP == 14) { REP16(ADDF64) REP16(ADDF64) }
        if constexpr (OP == 15) { REP16(CMPF32) REP16(CMPF32) }
        if constexpr (OP == 16) { REP16(CMPU32) REP16(CMPU32) }
        if constexpr (OP == 17) { REP16(CMPF64V) REP16(CMPF64V) }
        if constexpr (OP == 18) { REP16(CMPF32V) REP16(CMPF32V) }
        if constexpr (OP == 19) { REP16(CMPSEL64) REP16(CMPSEL64) }
        if constexpr (OP == 20) { REP16(CMPSEL32) REP16(CMPSEL32) }
    }
    const long long t1 = clock64();
    uint32_t s = 0;
    for (int i = 0; i < 16; i++) s += a[i] + (uint32_t)d[i];
    s += (uint32_t)acc;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clocks[0] = t1 - t0;
}

template <int OP>
static void run(const char* name, int waves_per_simd) {
    const int iters = 4096, blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = one per SIMD of a CU
    uint32_t* out;
    long long* clk;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipMalloc(&clk, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(out, 16, 12345u, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, iters, 12345u, clk);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long c = 0;
    hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double insts_per_wave = (double)iters * 32;
    // wall-clock view: instructions per SIMD = waves_per_simd * insts_per_wave, in ms at an assumed 2.4 GHz ceiling
    const double cyc_wall_24 = ms * 1e-3 * 2.4e9 / (waves_per_simd * insts_per_wave);
    printf("%-16s waves/SIMD %d: %.3f ms, %.2f cycles/inst/SIMD if 2.4 GHz; s_memtime-clock view %.2f ticks/inst/wave (100 MHz counter: x clock/100MHz)\n", name,
           waves_per_simd, ms, cyc_wall_24, (double)c / insts_per_wave);
    hipFree(out), hipFree(clk);
}

int main() {
    for (int w : {2, 8}) {
        run<0>("v_min/max_u32", w);
        run<1>("v_min_f64", w);
        run<2>("v_fma_f64", w);
        run<3>("v_fma_f32", w);
        run<4>("v_add_f32", w);
        run<5>("v_add_u32", w);
        run<6>("v_cndmask_b32", w);
        run<7>("v_pk_fma_f32", w);
        run<8>("v_cmp_gt_f64", w);
        run<9>("v_mov_b32_dpp", w);
        run<10>("v_lshl_or_b32", w);
        run<11>("v_mad_u32_u24", w);
        run<12>("v_mul_lo_u32", w);
        run<13>("v_min3_u32", w);
        run<14>("v_add_f64", w);
        run<15>("v_cmp_ge_f32 -> sgpr, s_xor", w);
        run<16>("v_cmp_gt_u32 -> sgpr, s_xor", w);
        run<17>("v_cmp_gt_f64 -> vcc, unread", w);
        run<18>("v_cmp_ge_f32 -> vcc, unread", w);
        run<19>("v_cmp_gt_f64 + v_cndmask (2 inst)", w);
        run<20>("v_cmp_ge_f32 + v_cndmask (2 inst)", w);
    }
    return 0;
}
