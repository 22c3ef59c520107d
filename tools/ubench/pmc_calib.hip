// What do FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc, gfx950) report for the access widths the line-search kernels use?
// MI355X_MICROARCH.md calibrates only the 16 B/lane streaming read (FETCH_SIZE = half the bytes) and says to calibrate other
// widths and WRITE_SIZE on a known byte count.  Each kernel below moves exactly BYTES (1 GiB, four times the Infinity Cache)
// in one pattern: reads of 16 / 8 / 4 / 2 / 1 bytes per lane, stores of 16 / 8 / 4 / 2 bytes per lane, and the verify kernel's
// store shape (one 512-byte row per wave, rows of one wave 16 KiB apart).  tools/pmc_calib.sh runs it under the two passes
// and prints counter / bytes.  Build: hipcc -O2 --offload-arch=gfx950 -o pmc_calib pmc_calib.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static constexpr size_t BYTES = (size_t)1 << 30;

template <typename T>
__device__ inline unsigned fold(T v);
template <> __device__ inline unsigned fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }
template <> __device__ inline unsigned fold(double v) { return (unsigned)__double_as_longlong(v) ^ (unsigned)(__double_as_longlong(v) >> 32); }
template <> __device__ inline unsigned fold(float v) { return __float_as_uint(v); }
template <> __device__ inline unsigned fold(unsigned short v) { return v; }
template <> __device__ inline unsigned fold(unsigned char v) { return v; }

// grid-stride streaming read, one element of T per lane and trip; the fold keeps the loads alive
template <typename T>
__global__ void __launch_bounds__(256) calib_read(const T* __restrict__ src, size_t n, unsigned* __restrict__ sink, unsigned magic) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= fold<T>(src[i]);
    if (acc == magic) sink[0] = acc;  // (a run-time value: the loads cannot be proven dead)
}

template <typename T>
__device__ inline T make(unsigned v);
template <> __device__ inline uint4 make(unsigned v) { return uint4{v, v, v, v}; }
template <> __device__ inline double make(unsigned v) { return (double)v; }
template <> __device__ inline float make(unsigned v) { return (float)v; }
template <> __device__ inline unsigned short make(unsigned v) { return (unsigned short)v; }

template <typename T>
__global__ void __launch_bounds__(256) calib_write(T* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = make<T>((unsigned)i);
}

// linesearch_verify_kernel's store: wave w owns rows w, w + waves, ...; a row is 64 doubles (512 B) at  row * ld  doubles
// (ld = 2048: M[q][g][64] with 32 groups), so consecutive stores of a wave are 16 KiB apart and every row is written once
__global__ void __launch_bounds__(64) calib_write_rows(double* __restrict__ dst, size_t rows, size_t ld, unsigned groups) {
    const unsigned lane = threadIdx.x;
    for (size_t r = blockIdx.x; r < rows * groups; r += gridDim.x) {
        const size_t q = r / groups, g = r % groups;
        dst[q * ld + g * 64 + lane] = (double)r;
    }
}

template <typename T>
static void run_read(const void* buf, unsigned* sink, const char* name) {
    const size_t n = BYTES / sizeof(T);
    hipLaunchKernelGGL(calib_read<T>, dim3(256 * 16), dim3(256), 0, 0, (const T*)buf, n, sink, 0x35u);
    CK(hipDeviceSynchronize());
    printf("%s: %zu bytes\n", name, BYTES);
}
template <typename T>
static void run_write(void* buf, const char* name) {
    const size_t n = BYTES / sizeof(T);
    hipLaunchKernelGGL(calib_write<T>, dim3(256 * 16), dim3(256), 0, 0, (T*)buf, n);
    CK(hipDeviceSynchronize());
    printf("%s: %zu bytes\n", name, BYTES);
}

int main() {
    void* buf;
    unsigned* sink;
    CK(hipMalloc(&buf, BYTES));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, BYTES));
    CK(hipDeviceSynchronize());
    run_read<uint4>(buf, sink, "calib_read<uint4> 16 B/lane");
    run_read<double>(buf, sink, "calib_read<double> 8 B/lane");
    run_read<float>(buf, sink, "calib_read<float> 4 B/lane");
    run_read<unsigned short>(buf, sink, "calib_read<u16> 2 B/lane");
    run_read<unsigned char>(buf, sink, "calib_read<u8> 1 B/lane");
    run_write<uint4>(buf, "calib_write<uint4> 16 B/lane");
    run_write<double>(buf, "calib_write<double> 8 B/lane");
    run_write<float>(buf, "calib_write<float> 4 B/lane");
    run_write<unsigned short>(buf, "calib_write<u16> 2 B/lane");
    // 65536 queries x 32 groups x 512 B = 1 GiB
    hipLaunchKernelGGL(calib_write_rows, dim3(256 * 32), dim3(64), 0, 0, (double*)buf, (size_t)65536, (size_t)2048, 32u);
    CK(hipDeviceSynchronize());
    printf("calib_write_rows 512 B per wave, 16 KiB apart: %zu bytes\n", BYTES);
    return 0;
}
