// Interface between device.hip (DeviceDataset, which launches) and fullverify.hip (which holds the instantiations of
// fullrank_verify_kernel, kernels_fullverify.inc).  fullverify.hip is compiled several times (-DFV_PART=k), one object
// per slice of the size classes, so that the many fully unrolled sorting networks compile in parallel.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace frdev {

struct FVArgs {
    const float4* xb;
    const uint32_t* qstart;
    const uint32_t* qlen;
    const uint32_t* qnpos;    // [nq] documents with gain > 0
    const uint32_t* qlist;    // queries of this size class
    const uint32_t* gcls;     // [np] gain class per position
    const uint32_t* gfeat;    // [G]
    const double* gw;         // [G][4*dq] zero padded
    const double* gcand;      // [G][64]
    const uint32_t* gncand;   // [G]
    const double* eps2;       // [G][64]
    const double* termtab;    // [ncls][tablen]
    const double* norms;      // NDCG ideal DCG (NaN = None) / AP number of relevant documents (0 = absent)
    uint32_t* redo_count;
    uint32_t* redo_list;      // q * G + g
    double* res_cur;          // resident sums, see LSArgs (rs_slot == nullptr: A is formed from the tiles)
    const int32_t* rs_slot;
    const double* rs_par;
    const uint32_t* rs_updf;
    double* M;                // [nq][ldm]
    int* flags;
    uint64_t relmask;         // bit c set: gain class c is relevant (gain > 0)
    uint32_t G, ldm, dq, d, np, tablen, cls_mask;
    uint32_t padcls;          // class id of the padding keys (= number of gain classes; its termtab row is all zero)
    // Duplicate groups (the DUP instantiations; dup != 0 when some query holds bit-identical rows with different gain classes):
    // a key's low bits then hold class | group id << cls_bits (cls_mask covers both, cls_only_mask the class), inverted in a
    // negative key -- see kernels_verify.inc, VERIFY_BITS_SRC -- and a pair of ONE group stands in the reference's tie-break
    // order as sorted.  gkey[p] = class | group << key_cls_bits as the NDCG@k kernel keeps it (class ids without the padding class)
    const uint16_t* gkey;
    uint32_t dup, key_cls_bits, cls_bits, cls_only_mask;
    uint32_t nqc;             // queries in qlist
    uint32_t qb;              // queries a block sorts together (their candidates fill the lanes of its waves)
    uint32_t nrec;            // query records a block keeps in LDS: qb * (most batches any block walks)
    uint32_t dbuf;            // 1: two word buffers (phase 1 of the next batch overlaps the sorting passes), 0: one
    int measure, depth;
};

enum : int { FV_NDCG = 0, FV_NDCG_CUT = 1, FV_AP = 2 };  // MODE: NDCG over the whole list / NDCG with a depth shorter than the query / AP

// A size class: every query of at most nl * pl documents is sorted as pl adjacent lanes of nl register-resident keys
// per candidate.  `cost` = modelled VALU wave-instructions per (query, 64 candidate lanes), used to pick a query's class.
struct FVClass {
    uint32_t nl, pl;
};
// ascending in nl * pl.  Lanes per candidate are powers of two: those groups are aligned, so their exchanges run through
// DPP.  (Measured at the 30K shape, round 3: classes of 3 / 5 / 7 / 10 / 12 lanes, which need ds_bpermute, cost as much
// per query as the next power of two -- 96x3 1.53 us against 96x4 1.38 us, 64x7 2.93 against 64x8 2.95 -- so the finer
// steps come from the keys per lane instead: 64 / 80 / 96.)
static constexpr FVClass FV_CLASSES[] = {{16, 1}, {32, 1}, {48, 1}, {64, 1}, {80, 1}, {96, 1}, {64, 2}, {80, 2}, {96, 2}, {64, 4},
                                         {80, 4}, {96, 4}, {64, 8}, {80, 8}, {96, 8}, {64, 16}, {80, 16}, {96, 16}, {64, 32}};
static constexpr int FV_NCLASSES = (int)(sizeof(FV_CLASSES) / sizeof(FV_CLASSES[0]));
static constexpr int FV_PARTS = 8;       // translation units the classes are dealt over (fullverify.hip -DFV_PART=k)
static constexpr int FV_BLOCK_WAVES = 4;  // waves per workgroup

// waves per SIMD the kernel is compiled for (VGPR budget 512 / n): the keys alone take 2 * nl registers
// up to this many keys per lane the next batch's documents are prefetched across a sorting pass (kernels_fullverify.inc)
#ifndef FV_PRE_MAX_NL
#define FV_PRE_MAX_NL 80
#endif
#ifndef FV_WPS64
#define FV_WPS64 2
#endif
#ifndef FV_WPS96
#define FV_WPS96 2
#endif
constexpr int fv_waves_per_simd(int nl) { return nl <= 16 ? 5 : (nl <= 32 ? 4 : (nl <= 48 ? 3 : (nl <= 64 ? FV_WPS64 : (nl <= 80 ? 2 : FV_WPS96)))); }

// LDS of one workgroup (bytes): the words of 2 x qb queries, the records of the nrec queries the block walks, the
// candidate table, the term table.  tab_doubles = doubles of the metric's table in LDS (0: terms are gathered from global memory).
inline size_t fv_lds_bytes(uint32_t nl, uint32_t pl, uint32_t qb, uint32_t nrec, size_t tab_doubles, bool dbuf) {
    const size_t wstride = (size_t)pl * (nl + 1);
    return (dbuf ? 2 : 1) * (size_t)qb * wstride * 16 + (size_t)nrec * 16 + (((size_t)nrec + 1) & ~(size_t)1) * 8 + 2 * 64 * 8 + tab_doubles * 8 +
           3 * (size_t)qb * 4 + (size_t)qb * 64 * 4;
}

// launches class index `ci` (into FV_CLASSES); false if that instantiation does not exist (tablds for a class that has none)
bool fv_launch(int ci, int mode, bool tablds, const FVArgs& a, dim3 grid, size_t lds, hipStream_t st);
// does class ci have an LDS-table instantiation?
inline bool fv_has_tablds(int ci) { return FV_CLASSES[ci].nl * FV_CLASSES[ci].pl <= 1024; }

}  // namespace frdev
