// Instantiations of fullrank_verify_kernel (kernels_fullverify.inc), one slice of the size classes per object:
// compiled FV_PARTS times with -DFV_PART=0 .. FV_PARTS-1 (fastrank_amd/_build.py), so that the fully unrolled sorting
// networks (up to 1056 + 336 compare-exchanges per lane) build in parallel.  Class ci belongs to part ci % FV_PARTS.
#include "fullverify.hpp"

#include "device.hpp"

#ifndef FV_PART
#error "compile with -DFV_PART=<0..FV_PARTS-1>"
#endif

namespace frdev {

#include "kernels_sortnet.inc"
#include "kernels_fullverify.inc"

template <int NL, int PL, int MODE, bool TABLDS, bool DUP = false>
static bool fv_launch_one(const FVArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    // (more than the default 64 KB of dynamic LDS has to be asked for; per call: the attribute lives with the device's
    // copy of the function, and several devices may be in use)
    if (lds > (size_t(48) << 10) &&
        hipFuncSetAttribute((const void*)fullrank_verify_kernel<NL, PL, MODE, TABLDS, DUP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return false;
    fullrank_verify_kernel<NL, PL, MODE, TABLDS, DUP><<<grid, dim3(64 * FV_BLOCK_WAVES), lds, st>>>(a);
    return true;
}

// (DUP: the duplicate-group rule, chosen by the host when some query holds bit-identical rows with different gain classes)
template <int NL, int PL>
static bool fv_launch_class(int mode, bool tablds, const FVArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    const bool dup = a.dup != 0u;
    if (mode == FV_AP) return dup ? fv_launch_one<NL, PL, FV_AP, false, true>(a, grid, lds, st) : fv_launch_one<NL, PL, FV_AP, false>(a, grid, lds, st);
    if (mode == FV_NDCG_CUT)
        return !tablds && (dup ? fv_launch_one<NL, PL, FV_NDCG_CUT, false, true>(a, grid, lds, st) : fv_launch_one<NL, PL, FV_NDCG_CUT, false>(a, grid, lds, st));
    if (tablds) {
        if constexpr (NL * PL <= 1024)
            return dup ? fv_launch_one<NL, PL, FV_NDCG, true, true>(a, grid, lds, st) : fv_launch_one<NL, PL, FV_NDCG, true>(a, grid, lds, st);
        else return false;
    }
    return dup ? fv_launch_one<NL, PL, FV_NDCG, false, true>(a, grid, lds, st) : fv_launch_one<NL, PL, FV_NDCG, false>(a, grid, lds, st);
}

template <int CI>
static bool fv_try(int ci, int mode, bool tablds, const FVArgs& a, dim3 grid, size_t lds, hipStream_t st) {
#ifdef FV_ONLY_NL
    if constexpr (true) {
#else
    if constexpr (CI >= FV_NCLASSES) {
#endif
        return false;
    } else {
        if (ci == CI) return fv_launch_class<(int)FV_CLASSES[CI].nl, (int)FV_CLASSES[CI].pl>(mode, tablds, a, grid, lds, st);
        return fv_try<CI + FV_PARTS>(ci, mode, tablds, a, grid, lds, st);
    }
}

#ifdef FV_ONLY_NL  // tools/isa_fv.sh: one instantiation for looking at its ISA
template __global__ void fullrank_verify_kernel<FV_ONLY_NL, FV_ONLY_PL, FV_ONLY_MODE, FV_ONLY_MODE == FV_NDCG>(FVArgs);
#endif

#define FV_CAT2(a, b) a##b
#define FV_CAT(a, b) FV_CAT2(a, b)
bool FV_CAT(fv_launch_part, FV_PART)(int ci, int mode, bool tablds, const FVArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    return fv_try<FV_PART>(ci, mode, tablds, a, grid, lds, st);
}

}  // namespace frdev
