// Which compute units does a stream created with hipExtStreamCreateWithCUMask reach on MI355X (8 XCDs x 32 CUs)?
// Every workgroup records (XCC id, shader engine, compute unit) from the hardware-id registers; the histogram per mask
// shows how the mask's bits map onto XCDs.  tools/ab/cumask.sh runs it.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ void who(unsigned* out, int spin) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // keep the slot for a while so that the grid spreads over everything the queue may use
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2] = xcc;
        out[blockIdx.x * 2 + 1] = hw;
    }
}

static void run(const char* name, const std::vector<uint32_t>& mask) {
    hipStream_t st;
    hipError_t e = mask.empty() ? hipStreamCreate(&st) : hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) {
        printf("%s: stream creation failed: %s\n", name, hipGetErrorString(e));
        return;
    }
    const int nb = 16384;
    unsigned* d;
    hipMalloc(&d, nb * 2 * sizeof(unsigned));
    hipLaunchKernelGGL(who, dim3(nb), dim3(64), 0, st, d, 20000);
    hipStreamSynchronize(st);
    std::vector<unsigned> h(nb * 2);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<unsigned, std::map<unsigned, int>> per;  // xcc -> (se << 8 | cu) -> count
    for (int b = 0; b < nb; b++) {
        const unsigned xcc = h[b * 2] & 0xf, hw = h[b * 2 + 1];
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
        per[xcc][(se << 8) | (sh << 4) | cu]++;
    }
    printf("%s:", name);
    int total = 0;
    for (auto& x : per) {
        printf("  xcc%u:%zu CUs", x.first, x.second.size());
        total += (int)x.second.size();
    }
    printf("  -> %d distinct CUs\n", total);
    hipFree(d);
    hipStreamDestroy(st);
}

int main() {
    run("no mask", {});
    run("all 256 bits", std::vector<uint32_t>(8, 0xffffffffu));
    run("first 32 bits", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0});
    run("first 8 bits", {0xffu, 0, 0, 0, 0, 0, 0, 0});
    run("bits 0,8,16,..,248 (every 8th)", std::vector<uint32_t>(8, 0x01010101u));
    run("all but the first 8 bits", {0xffffff00u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu});
    run("all but the last 8 bits", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x00ffffffu});
    run("all but bits 0..15", {0xffff0000u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu});
    run("32-bit mask only (size 1) all ones", {0xffffffffu});
    run("32-bit mask only (size 1) 0xffffff00", {0xffffff00u});
    return 0;
}
