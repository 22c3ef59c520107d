// How fast does a caller-owned 2 GB host matrix reach HBM?  (a) hipHostRegister the whole range, one hipMemcpyAsync; (b) register +
// copy in chunks, a few threads registering ahead; (c) two pinned 32 MB slabs filled by 12 memcpy threads (what
// DeviceDataset::create does); (d) pageable hipMemcpy.  Build: hipcc -O2 -o hostreg hostreg.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const size_t bytes = (argc > 1 ? atol(argv[1]) : 2067) * (size_t)1000000;
    char* h = (char*)malloc(bytes);
    memset(h, 1, bytes);  // touch every page, like a numpy array that was written
    char* d;
    CK(hipMalloc(&d, bytes));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    CK(hipMemcpy(d, h, 1 << 20, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now();
        CK(hipHostRegister(h, bytes, hipHostRegisterDefault));
        double t1 = now();
        CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        double t2 = now();
        CK(hipHostUnregister(h));
        double t3 = now();
        printf("(a) whole range: register %.1f ms, copy %.1f ms (%.1f GB/s), unregister %.1f ms, total %.1f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3,
               bytes / (t2 - t1) / 1e9, (t3 - t2) * 1e3, (t3 - t0) * 1e3);
    }
    for (size_t chunk_mb : {64, 256}) {
        for (int nth : {1, 4, 8}) {
            const size_t chunk = chunk_mb << 20, nch = (bytes + chunk - 1) / chunk;
            double t0 = now();
            std::vector<std::thread> pool;
            std::vector<char> ready(nch, 0);
            // thread t registers chunks t, t + nth, ...; the main thread copies chunk i as soon as it is registered
            for (int t = 0; t < nth; t++)
                pool.emplace_back([&, t]() {
                    for (size_t i = t; i < nch; i += nth) {
                        const size_t off = i * chunk, len = std::min(chunk, bytes - off);
                        if (hipHostRegister(h + off, len, hipHostRegisterDefault) != hipSuccess) printf("register failed\n");
                        __atomic_store_n(&ready[i], 1, __ATOMIC_RELEASE);
                    }
                });
            for (size_t i = 0; i < nch; i++) {
                while (!__atomic_load_n(&ready[i], __ATOMIC_ACQUIRE)) std::this_thread::yield();
                const size_t off = i * chunk, len = std::min(chunk, bytes - off);
                CK(hipMemcpyAsync(d + off, h + off, len, hipMemcpyHostToDevice, st));
            }
            CK(hipStreamSynchronize(st));
            double t1 = now();
            for (auto& th : pool) th.join();
            for (size_t i = 0; i < nch; i++) CK(hipHostUnregister(h + i * chunk));
            double t2 = now();
            printf("(b) chunks of %zu MB, %d registering threads: data on the device after %.1f ms (%.1f GB/s), unregister %.1f ms\n", chunk_mb, nth,
                   (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9, (t2 - t1) * 1e3);
        }
    }
    {
        const size_t slab = 32 << 20;
        char* s[2];
        CK(hipHostMalloc((void**)&s[0], slab));
        CK(hipHostMalloc((void**)&s[1], slab));
        hipEvent_t ev[2];
        CK(hipEventCreate(&ev[0]));
        CK(hipEventCreate(&ev[1]));
        double t0 = now();
        int turn = 0;
        bool used[2] = {false, false};
        for (size_t off = 0; off < bytes; off += slab, turn ^= 1) {
            const size_t len = std::min(slab, bytes - off);
            if (used[turn]) CK(hipEventSynchronize(ev[turn]));
            std::vector<std::thread> pool;
            for (int t = 0; t < 12; t++)
                pool.emplace_back([&, t]() { memcpy(s[turn] + len * t / 12, h + off + len * t / 12, len * (t + 1) / 12 - len * t / 12); });
            for (auto& th : pool) th.join();
            CK(hipMemcpyAsync(d + off, s[turn], len, hipMemcpyHostToDevice, st));
            CK(hipEventRecord(ev[turn], st));
            used[turn] = true;
        }
        CK(hipStreamSynchronize(st));
        double t1 = now();
        printf("(c) two pinned 32 MB slabs, 12 memcpy threads: %.1f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
    }
    {
        double t0 = now();
        CK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
        double t1 = now();
        printf("(d) pageable hipMemcpy: %.1f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
    }
    return 0;
}
