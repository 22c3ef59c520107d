// HIP kernels + launchers for the fastrank hot path on MI355X (gfx950, wave64).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off  (contract=off is REQUIRED: the
// reference's dot product is an unfused f64 multiply-then-add, src/dense_dataset.rs:71-74).
//
// Kernels
//   score_linear_kernel      lane = document; exact ordered f64 dot product for B weight vectors
//   linesearch_ndcg_kernel   the fused coordinate-ascent line search: one wave per
//                            (query, line group); phase S (lane = document) scores all <=64
//                            candidates of the group with a shared prefix sum; an LDS transpose
//                            turns lanes into candidates; phase K keeps each candidate's top-K
//                            list in registers by ordered insertion; NDCG@k per (query, candidate)
//   metric_sort_kernel       general evaluator: LDS bitonic sort of one query's (score, position)
//                            keys with the reference's 3-key order, then NDCG / AP / RR
//   segment_sum_kernel +     mean over queries with a fixed two-level summation shape
//   final_mean_kernel        (256-query segments summed in order, then segment partials in order)
//   tree_ensemble_kernel     batched tree traversal, document rows staged in LDS
#include "device.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <thread>

namespace frdev {

// ----------------------------------------------------------------------------------------------
// plumbing
// ----------------------------------------------------------------------------------------------

#define FR_HIP(expr)                                                                        \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            if (err) *err = std::string("HIP error: ") + hipGetErrorString(_e) + " at " #expr; \
            return false;                                                                   \
        }                                                                                   \
    } while (0)

static std::mutex g_prof_mu;
static bool g_prof_on = false;
struct ProfRec {
    const char* name;
    hipEvent_t a, b;
};
static std::vector<ProfRec> g_prof_recs;
static std::map<std::string, KernelStat> g_prof_done;

void profile_enable(bool on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on;
}

static void prof_collect_locked() {
    for (auto& r : g_prof_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            auto& s = g_prof_done[r.name];
            s.name = r.name;
            s.launches++;
            s.total_ms += (double)ms;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_prof_recs.clear();
}

void profile_reset() {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    prof_collect_locked();
    g_prof_done.clear();
}

std::vector<KernelStat> profile_stats() {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    prof_collect_locked();
    std::vector<KernelStat> out;
    for (auto& kv : g_prof_done) out.push_back(kv.second);
    return out;
}

struct ProfScope {
    hipStream_t st;
    bool on = false;
    ProfRec rec{};
    ProfScope(const char* name, hipStream_t s) : st(s) {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (!g_prof_on) return;
        if (hipEventCreate(&rec.a) != hipSuccess) return;
        if (hipEventCreate(&rec.b) != hipSuccess) { (void)hipEventDestroy(rec.a); return; }
        rec.name = name;
        on = true;
        (void)hipEventRecord(rec.a, st);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(rec.b, st);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof_recs.push_back(rec);
        if (g_prof_recs.size() > 4096) prof_collect_locked();
    }
};

int device_count(std::string* err) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        if (err) *err = std::string("HIP error: ") + hipGetErrorString(e);
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

bool set_device(int ordinal, std::string* err) {
    FR_HIP(hipSetDevice(ordinal));
    return true;
}

bool device_synchronize(std::string* err) {
    FR_HIP(hipDeviceSynchronize());
    return true;
}

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    bool ensure(size_t n, std::string* err) {
        if (n <= cap) return true;
        release();
        size_t want = n;
        FR_HIP(hipMalloc((void**)&p, want * sizeof(T)));
        cap = want;
        return true;
    }
    size_t bytes() const { return cap * sizeof(T); }
};

// ----------------------------------------------------------------------------------------------
// kernels
// ----------------------------------------------------------------------------------------------

constexpr int WAVE = 64;
constexpr uint32_t IDX_INVALID = 0xFFFFFFFFu;

// Blocked feature layout ("tiles"): documents live in a padded position space p; tile = p >> 6
// holds 64 documents; inside a tile features are grouped by four:
//     float index(p, j) = ((p >> 6) * dq + (j >> 2)) * 256 + (p & 63) * 4 + (j & 3),  dq = ceil(D / 4)
// so lane = document reads 16 B (four consecutive features) per load and a wave reads 1 KiB
// contiguous.  Feature columns D..4*dq-1 and padding documents are zero.
__host__ __device__ inline size_t xb_index(uint32_t p, uint32_t j, uint32_t dq) {
    return ((size_t)(p >> 6) * dq + (j >> 2)) * 256 + (size_t)(p & 63) * 4 + (j & 3);
}

// scores[b*np + p] = sum_j f64(x[p][j]) * w[b][j], j ascending, unfused (dense_dataset.rs:67-76).
// Weights arrive zero-padded to 4*dq: a padded column contributes x*0 = +0.0, which never changes
// the running sum (the sum starts at +0.0 and can therefore never be -0.0).
template <int BT>
__global__ __launch_bounds__(256) void score_linear_kernel(const float4* __restrict__ xb, uint32_t np, uint32_t dq,
                                                           const double* __restrict__ w, uint32_t B,
                                                           double* __restrict__ scores) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t b0 = blockIdx.y * BT;
    if (p >= np) return;
    double acc[BT];
#pragma unroll
    for (int t = 0; t < BT; t++) acc[t] = 0.0;
    const float4* xp = xb + (size_t)(p >> 6) * dq * 64 + (p & 63);
    const uint32_t dp = dq * 4;
#pragma unroll 2
    for (uint32_t j4 = 0; j4 < dq; j4++) {
        float4 x4 = xp[(size_t)j4 * 64];
        const float xs[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
        for (int u = 0; u < 4; u++) {
            double x = (double)xs[u];
#pragma unroll
            for (int t = 0; t < BT; t++) {
                uint32_t b = b0 + t < B ? b0 + t : B - 1;
                double prod = x * w[(size_t)b * dp + j4 * 4 + u];
                acc[t] = acc[t] + prod;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < BT; t++)
        if (b0 + t < B) scores[(size_t)(b0 + t) * np + p] = acc[t];
}

// SingleFeatureModel (src/model.rs:35-40): dir * f64(x[fid])
__global__ void score_single_feature_kernel(const float* __restrict__ xb, uint32_t np, uint32_t dq, uint32_t fid,
                                            double dir, double* __restrict__ scores) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < np) scores[p] = dir * (double)xb[xb_index(p, fid, dq)];
}

__global__ void fill_kernel(double* __restrict__ a, uint32_t n, double v) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) a[p] = v;
}

// acc = acc + w * t (unfused), WeightedEnsemble::score (src/model.rs:104-112)
__global__ void axpy_unfused_kernel(double* __restrict__ acc, const double* __restrict__ t, uint32_t n,
                                    double w) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) {
        double prod = w * t[p];
        acc[p] = acc[p] + prod;
    }
}

struct TreeNodeDev {
    double split;  // threshold, or the leaf value when fid < 0
    int32_t fid;   // < 0: leaf
    int32_t lhs;   // left child; the right child is lhs + 1 (children are stored adjacently)
};

// Batched tree-ensemble scoring (src/model.rs:64-84,104-112; config 5 of BASELINE.json).
// One thread = one document, one 64-document tile per block.  The block first stages its
// documents' feature rows from the tile into LDS (16-byte loads in, row stride 4*dq+1 dwords so
// lane-strided accesses spread over the banks).  Each lane then walks TI trees at once (TI
// independent pointer chases hide the L1/L2 latency of the 16-byte node records; all lanes walk
// the same trees, so the records stay cache resident) and adds the leaves in tree order:
// out += w_t * leaf_t, unfused (src/model.rs:104-112).
template <bool ROWS_IN_LDS, int TI>
__global__ __launch_bounds__(64) void tree_ensemble_kernel(const float* __restrict__ xb, uint32_t np, uint32_t dq,
                                                           uint32_t d, const TreeNodeDev* __restrict__ nodes,
                                                           const int32_t* __restrict__ roots,
                                                           const double* __restrict__ tw, uint32_t ntrees,
                                                           int raw_single, double* __restrict__ scores) {
    extern __shared__ float rows[];  // [64][4*dq+1]
    const uint32_t tid = threadIdx.x;
    const uint32_t p = blockIdx.x * blockDim.x + tid;  // np is a multiple of 64
    const uint32_t rs = dq * 4 + 1;
    if (ROWS_IN_LDS) {
        const float4* xp = (const float4*)xb + (size_t)(p >> 6) * dq * 64 + (p & 63);
        for (uint32_t j4 = 0; j4 < dq; j4++) {
            float4 v = xp[(size_t)j4 * 64];
            float* r = rows + tid * rs + j4 * 4;
            r[0] = v.x;
            r[1] = v.y;
            r[2] = v.z;
            r[3] = v.w;
        }
        __syncthreads();
    }
    const float* myrow = rows + tid * rs;
    double acc = 0.0;
    for (uint32_t t0 = 0; t0 < ntrees; t0 += TI) {
        TreeNodeDev nd[TI];
#pragma unroll
        for (int k = 0; k < TI; k++) nd[k] = nodes[roots[t0 + k < ntrees ? t0 + k : ntrees - 1]];
        bool any_inner = true;
        while (any_inner) {
            any_inner = false;
#pragma unroll
            for (int k = 0; k < TI; k++) {
                if (nd[k].fid >= 0) {
                    float xv;
                    if ((uint32_t)nd[k].fid < d) {
                        xv = ROWS_IN_LDS ? myrow[(uint32_t)nd[k].fid] : xb[xb_index(p, (uint32_t)nd[k].fid, dq)];
                    } else {
                        xv = 0.0f;  // Features::get -> None -> unwrap_or(0.0) (src/model.rs:72-73)
                    }
                    const int32_t next = ((double)xv <= nd[k].split) ? nd[k].lhs : nd[k].lhs + 1;
                    nd[k] = nodes[next];
                    any_inner |= nd[k].fid >= 0;
                }
            }
            any_inner = __any(any_inner);
        }
#pragma unroll
        for (int k = 0; k < TI; k++) {
            if (t0 + k < ntrees) {
                if (raw_single) {
                    acc = nd[k].split;
                } else {
                    double prod = tw[t0 + k] * nd[k].split;
                    acc = acc + prod;
                }
            }
        }
    }
    if (p < np) scores[p] = acc;
}

// Compact forest for the LDS walk.  One 8-byte word per node; children adjacent (right = left + 1):
//   internal: { float thr; u16 fid; u16 left }   thr = largest f32 <= split, so for every f32 x
//             (f64(x) <= split)  <=>  (x <= thr)          (src/model.rs:75-79)
//   leaf:     { 0; 0xFFFF; u16 index into this tree's f64 leaf table }
constexpr uint32_t LDS_TREES_PER_BATCH = 64;  // upper bound on trees per LDS batch

struct PackedNode {
    float thr;
    uint16_t fid;
    uint16_t ref;
};

// Batched tree-ensemble scoring, LDS-resident trees (the fast path of config 5).
// Block = `blockDim.x / 64` tiles of documents, one thread per document: feature rows staged in LDS
// (row stride 4*dq+1 dwords), then the forest streams through LDS in batches (words + f64 leaves,
// coalesced copy); every lane walks TI trees of the batch concurrently, `levels` steps each
// (leaves stay put), and adds the leaves in tree order: out += w_t * leaf_t, unfused.
template <int TI>
__global__ __launch_bounds__(256) void tree_ensemble_lds_kernel(
    const float* __restrict__ xb, uint32_t np, uint32_t dq, uint32_t d, const uint64_t* __restrict__ forest,
    const uint32_t* __restrict__ batch_off /*[nbatch+1] in 8-byte words*/, const uint32_t* __restrict__ tree_meta
    /*[ntrees][3]: word offset of root inside its batch, word offset of its leaf table, levels*/,
    const uint32_t* __restrict__ batch_first /*[nbatch+1] first tree of each batch*/,
    const double* __restrict__ tw, uint32_t nbatch, int raw_single, uint32_t tree_words_cap,
    double* __restrict__ scores) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw8[];
    uint64_t* trees = (uint64_t*)lds_raw8;                   // [tree_words_cap]
    double* bw = (double*)(trees + tree_words_cap);          // [LDS_TREES_PER_BATCH] tree weights of the batch
    uint32_t* bmeta = (uint32_t*)(bw + LDS_TREES_PER_BATCH); // [LDS_TREES_PER_BATCH][4]
    float* rows = (float*)(bmeta + LDS_TREES_PER_BATCH * 4); // [blockDim.x][4*dq+1]
    const uint32_t tid = threadIdx.x;
    const uint32_t p = blockIdx.x * blockDim.x + tid;        // np is a multiple of 64; p may exceed np in the last block
    const uint32_t pc = p < np ? p : np - 1;
    const uint32_t rs = dq * 4 + 1;
    {
        const float4* xp = (const float4*)xb + (size_t)(pc >> 6) * dq * 64 + (pc & 63);
        for (uint32_t j4 = 0; j4 < dq; j4++) {
            float4 v = xp[(size_t)j4 * 64];
            float* r = rows + tid * rs + j4 * 4;
            r[0] = v.x;
            r[1] = v.y;
            r[2] = v.z;
            r[3] = v.w;
        }
    }
    const float* myrow = rows + tid * rs;
    double acc = 0.0;
    for (uint32_t b = 0; b < nbatch; b++) {
        __syncthreads();  // previous batch fully consumed (and rows staged, first time round)
        const uint32_t w0 = batch_off[b], wn = batch_off[b + 1] - w0;
        for (uint32_t i = tid; i < wn; i += blockDim.x) trees[i] = forest[w0 + i];
        const uint32_t t_begin = batch_first[b], t_end = batch_first[b + 1];
        for (uint32_t i = tid; i < t_end - t_begin; i += blockDim.x) {
            bw[i] = tw[t_begin + i];
            bmeta[i * 4 + 0] = tree_meta[(t_begin + i) * 3 + 0];
            bmeta[i * 4 + 1] = tree_meta[(t_begin + i) * 3 + 1];
            bmeta[i * 4 + 2] = tree_meta[(t_begin + i) * 3 + 2];
        }
        __syncthreads();
        for (uint32_t t0 = t_begin; t0 < t_end; t0 += TI) {
            uint32_t cur[TI], root[TI], leaftab[TI];
            uint32_t levels = 0;
#pragma unroll
            for (int k = 0; k < TI; k++) {
                const uint32_t tl = (t0 + k < t_end ? t0 + k : t_end - 1) - t_begin;
                root[k] = bmeta[tl * 4 + 0];
                leaftab[k] = bmeta[tl * 4 + 1];
                cur[k] = root[k];
                const uint32_t lv = bmeta[tl * 4 + 2];
                levels = lv > levels ? lv : levels;
            }
            for (uint32_t lv = 0; lv < levels; lv++) {
                // branch-free step so that the TI walks' LDS reads are all in flight together
                uint64_t word[TI];
#pragma unroll
                for (int k = 0; k < TI; k++) word[k] = trees[cur[k]];
                float xv[TI];
#pragma unroll
                for (int k = 0; k < TI; k++) {
                    const uint32_t fid = (uint32_t)(word[k] >> 32) & 0xFFFFu;
                    xv[k] = myrow[fid < d ? fid : 0u];
                }
#pragma unroll
                for (int k = 0; k < TI; k++) {
                    const uint32_t hi = (uint32_t)(word[k] >> 32);
                    const uint32_t fid = hi & 0xFFFFu;
                    const float thr = __uint_as_float((uint32_t)word[k]);
                    const float x = fid < d ? xv[k] : 0.0f;  // Features::get -> None -> 0.0
                    const uint32_t left = root[k] + (hi >> 16);  // children relative to the root word
                    const uint32_t next = (x <= thr) ? left : left + 1;
                    cur[k] = (fid == 0xFFFFu) ? cur[k] : next;
                }
            }
#pragma unroll
            for (int k = 0; k < TI; k++) {
                if (t0 + k < t_end) {
                    const uint32_t t = t0 + k;
                    const uint64_t word = trees[cur[k]];
                    const uint32_t leaf = (uint32_t)(word >> 48);
                    const double val = __longlong_as_double((long long)trees[leaftab[k] + leaf]);
                    if (raw_single) {
                        acc = val;
                    } else {
                        double prod = bw[t - t_begin] * val;
                        acc = acc + prod;
                    }
                }
            }
        }
    }
    if (p < np) scores[p] = acc;
}

// a precedes b in the reference order?  Positions are in reverse tie-break layout, so among equal
// scores the HIGHER position ranks first (src/evaluators.rs:34-49).  Invalid (padding) keys last.
__device__ __forceinline__ bool key_before(double sa, uint32_t ia, double sb, uint32_t ib) {
    if (ia == IDX_INVALID) return false;
    if (ib == IDX_INVALID) return true;
    if (sa > sb) return true;
    if (sa < sb) return false;
    return ia > ib;
}

// General evaluator for one (query, score slot): LDS bitonic sort + metric.
// dynamic LDS: double keys[npad]; uint32 idx[npad]; (npad = pow2 >= longest query)
__global__ __launch_bounds__(256) void metric_sort_kernel(
    const double* __restrict__ scores, uint32_t np, const uint32_t* __restrict__ qstart,
    const uint32_t* __restrict__ qlen, const uint32_t* __restrict__ qtight, const double* __restrict__ gexp,
    const float* __restrict__ gain, const double* __restrict__ disc, const double* __restrict__ norms, int measure,
    int depth, uint32_t B, double* __restrict__ M, uint32_t* __restrict__ rank_out, const uint32_t* __restrict__ perm,
    int* __restrict__ flags, uint32_t npad_max, const uint32_t* __restrict__ qlist) {
    extern __shared__ double lds_raw[];
    double* keys = lds_raw;
    uint32_t* idx = (uint32_t*)(keys + npad_max);
    // queries are launched per size class so that the LDS footprint matches the query length
    const uint32_t q = qlist[blockIdx.x], b = blockIdx.y, tid = threadIdx.x, nt = blockDim.x;
    const uint32_t base = qstart[q], n = qlen[q];
    uint32_t npad = 1;
    while (npad < n) npad <<= 1;
    const double* sc = scores + (size_t)b * np + base;
    bool nan_seen = false;
    for (uint32_t i = tid; i < npad; i += nt) {
        if (i < n) {
            double s = sc[i];
            nan_seen |= (s != s);
            keys[i] = s;
            idx[i] = i;
        } else {
            keys[i] = 0.0;
            idx[i] = IDX_INVALID;
        }
    }
    if (nan_seen) atomicOr(flags, FLAG_NAN_SCORE);
    __syncthreads();
    for (uint32_t k = 2; k <= npad; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < (npad >> 1); t += nt) {
                uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                uint32_t l = i | j;
                bool up = (i & k) == 0;
                double si = keys[i], sl = keys[l];
                uint32_t ii = idx[i], il = idx[l];
                bool swap = up ? key_before(sl, il, si, ii) : key_before(si, ii, sl, il);
                if (swap) {
                    keys[i] = sl;
                    keys[l] = si;
                    idx[i] = il;
                    idx[l] = ii;
                }
            }
            __syncthreads();
        }
    }
    if (rank_out != nullptr && b == 0) {
        const uint32_t tb = qtight[q];
        for (uint32_t i = tid; i < n; i += nt) rank_out[tb + i] = perm[base + idx[i]];
    }
    double result = 0.0;
    if (measure == M_NDCG) {
        // src/evaluators.rs:255-272,350-380: terms in rank order, sequential sum from 0.0
        uint32_t L = depth >= 0 ? ((uint32_t)depth < n ? (uint32_t)depth : n) : n;
        for (uint32_t i = tid; i < L; i += nt) keys[i] = gexp[base + idx[i]] / disc[i];
        __syncthreads();
        if (tid == 0) {
            double norm = norms[q];
            if (norm == norm) {
                double dcg = 0.0;
                for (uint32_t i = 0; i < L; i++) dcg = dcg + keys[i];
                if (dcg > norm) atomicOr(flags, FLAG_ACTUAL_GT_IDEAL);
                result = dcg / norm;
            }
        }
    } else {
        // relevance flags in rank order (src/evaluators.rs:94-96 is_relevant: gain > 0)
        for (uint32_t i = tid; i < n; i += nt) keys[i] = gain[base + idx[i]] > 0.0f ? 1.0 : 0.0;
        __syncthreads();
        if (tid == 0) {
            if (measure == M_AP) {
                // src/evaluators.rs:422-447, step 1: recall_points at every relevant rank
                uint32_t c = 0;
                for (uint32_t i = 0; i < n; i++) {
                    c += keys[i] != 0.0;
                    idx[i] = keys[i] != 0.0 ? c : 0u;
                }
            } else {
                // src/evaluators.rs:239-252
                for (uint32_t i = 0; i < n; i++) {
                    if (keys[i] != 0.0) {
                        result = 1.0 / (double)(i + 1);
                        break;
                    }
                }
            }
        }
        if (measure == M_AP) {
            __syncthreads();
            // step 2 (all threads): precision at each relevant rank, the reference's operands exactly
            for (uint32_t i = tid; i < n; i += nt)
                if (idx[i]) keys[i] = (double)(int)idx[i] / (double)(i + 1);
            __syncthreads();
            if (tid == 0) {
                // step 3: ordered sum in rank order, then / num_relevant
                uint32_t num_rel = (uint32_t)norms[q];
                uint32_t in_list = 0;
                double sum_precision = 0.0;
                for (uint32_t i = 0; i < n; i++) {
                    if (idx[i]) {
                        sum_precision += keys[i];
                        in_list++;
                    }
                }
                if (num_rel == 0) num_rel = in_list;
                if (num_rel != 0) result = sum_precision / (double)num_rel;
            }
        }
    }
    if (tid == 0) M[(size_t)q * B + b] = result;
}

// Mean over queries with a FIXED two-level summation shape (the reference sums in a fresh
// HashMap's iteration order, i.e. unspecified: src/evaluators.rs:173-184 + dense_dataset.rs:96-109):
//   partial[s][c] = sequential sum of M[q][c] over segment s = queries [s*MEAN_SEG, (s+1)*MEAN_SEG)
//   mean[c]       = (sequential sum of partial[s][c] over s) / nq
// For nq <= MEAN_SEG this is the plain sequential sum in query order.
constexpr uint32_t MEAN_SEG = 256;

__global__ __launch_bounds__(64) void segment_sum_kernel(const double* __restrict__ M, uint32_t ldm, uint32_t ncols,
                                                         uint32_t nq, double* __restrict__ partial) {
    const uint32_t c = blockIdx.y * blockDim.x + threadIdx.x;
    const uint32_t s = blockIdx.x;
    if (c >= ncols) return;
    const uint32_t q0 = s * MEAN_SEG;
    const uint32_t q1 = (q0 + MEAN_SEG < nq) ? q0 + MEAN_SEG : nq;
    const double* p = M + (size_t)q0 * ldm + c;
    double sum = 0.0;
    uint32_t q = q0;
    for (; q + 16 <= q1; q += 16) {
        double v[16];
#pragma unroll
        for (int t = 0; t < 16; t++) v[t] = p[(size_t)t * ldm];
#pragma unroll
        for (int t = 0; t < 16; t++) sum += v[t];
        p += (size_t)16 * ldm;
    }
    for (; q < q1; q++) {
        sum += *p;
        p += ldm;
    }
    partial[(size_t)s * ldm + c] = sum;
}

__global__ void final_mean_kernel(const double* __restrict__ partial, uint32_t ldm, uint32_t ncols, uint32_t nseg,
                                  uint32_t nq, double* __restrict__ means) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncols) return;
    double sum = 0.0;
    for (uint32_t s = 0; s < nseg; s++) sum += partial[(size_t)s * ldm + c];
    means[c] = nq ? sum / (double)nq : 0.0;
}

struct LSArgs {
    const float4* xb;         // feature tiles
    const uint32_t* gcls;     // [np] gain class of each document
    const double* dcgtab;     // [ncls][LS_KT]: (2^gain - 1) / log2(i + 2), divided on the host
    const uint32_t* qstart;   // [nq] padded start position
    const uint32_t* qlen;     // [nq]
    const uint32_t* run_q0;   // [nruns] first query of the run
    const uint32_t* run_q1;   // [nruns] one past its last query
    const uint32_t* run_pos;  // [nruns] first position (multiple of 64)
    const uint32_t* run_docs; // [nruns] documents in the run
    const uint32_t* run_order;// [nruns] longest-first schedule
    const double* norms;
    const double* disc;
    const uint32_t* gfeat;    // [G]
    const double* gw;         // [G][4*dq] zero padded
    const double* gcand;      // [G][64]
    const uint32_t* gncand;   // [G]
    double* M;                // [nq][ldm]
    int* flags;
    unsigned long long* dbg_counters;  // [4] rows, batches, insertion rows, documents (debug bit 16)
    uint32_t dq, d, nruns, G, ldm;
    int depth;
    int debug;  // tuning knob (env FR_LS_DEBUG): 1 = skip phase K, 2 = no threshold filter, 16 = count rows
};

constexpr int LS_ROWPAD = 65;  // LDS row stride in doubles: 130 dwords -> 16 lanes hit 16 distinct even banks
constexpr int LS_KT = 20;      // ranks covered by the per-gain-class DCG term table
#ifndef LS_WAVES_PER_SIMD
#define LS_WAVES_PER_SIMD 3
#endif

__device__ __forceinline__ uint64_t lane_range_mask(uint32_t lo, uint32_t hi) {
    const uint64_t up = (hi >= 64u) ? ~0ull : ((1ull << hi) - 1ull);
    return up & ~((1ull << lo) - 1ull);
}

// The fused line search.  One wave per (run of consecutive queries, line group); CT = candidate
// tile (accumulators per lane in phase S), K = top-K list length (>= depth), RB = documents per
// transpose batch.
//
// phase S (lane = document, one 64-document tile per step; tiles are full because a run packs
//   several queries): exact ordered f64 dot products for all CT candidates at once.  The prefix
//   sum over features < f is shared by every candidate and so are the products x_j*w_j (j > f):
//   a candidate costs one v_add_f64 per feature.  Features arrive four at a time (16-byte loads,
//   prefetched two groups ahead); weights come from LDS where two masked copies (w_j for j < f,
//   w_j for j > f, zero elsewhere) remove every per-feature branch: a masked product is +-0.0 and
//   adding it changes nothing, because a sum that starts at +0.0 can never be -0.0.
// filter: a document can only matter if it ties/beats the current K-th best score of at least one
//   candidate; thresholds are re-published after every transpose batch.
// phase K (lane = candidate): surviving documents of the current query are transposed through LDS
//   in batches of RB rows and inserted, in document order, into each candidate's sorted top-K
//   list (registers).  At a query boundary inside the tile the lists are turned into NDCG@k.
template <int K, int CT, int RB>
__global__ __launch_bounds__(WAVE, LS_WAVES_PER_SIMD) void linesearch_ndcg_kernel(LSArgs a) {
    extern __shared__ __attribute__((aligned(16))) double wdyn[];  // wpre[4*dq], wsuf[4*dq]
    __shared__ double tr[RB * LS_ROWPAD];
    __shared__ double thr[WAVE];
    __shared__ double cwl[WAVE];  // this group's candidate weights for feature f
    __shared__ uint32_t rowcls[RB];
    const uint32_t lane = threadIdx.x;
    // XCD-aware block -> (run, group): blocks b, b+8, b+16.. run on one XCD (observed dispatch
    // b % 8), so all groups of a run share that XCD's L2 for the run's feature tiles.
    const uint32_t blk = blockIdx.x;
    const uint32_t xcd = blk & 7u, seq = blk >> 3;
    const uint32_t g = seq % a.G;
    const uint32_t ri = (seq / a.G) * 8u + xcd;
    if (ri >= a.nruns) return;
    const uint32_t r = a.run_order[ri];
    uint32_t q = a.run_q0[r];
    const uint32_t q1 = a.run_q1[r];
    const uint32_t pos = a.run_pos[r];
    const uint32_t run_end = pos + a.run_docs[r];
    const uint32_t f = a.gfeat[g];
    const uint32_t ncand = a.gncand[g];
    const uint32_t dq = a.dq, dp = a.dq * 4, d = a.d;
    const double* __restrict__ w = a.gw + (size_t)g * dp;
    const double NEG_INF = -__builtin_huge_val();
    double* wpre = wdyn;
    double* wsuf = wdyn + dp;
    for (uint32_t j = lane; j < dp; j += WAVE) {
        const double wv = j < d ? w[j] : 0.0;
        wpre[j] = j < f ? wv : 0.0;
        wsuf[j] = j > f ? wv : 0.0;
    }
    thr[lane] = NEG_INF;  // nothing is filtered until a candidate's list is full
    cwl[lane] = a.gcand[(size_t)g * 64 + lane];
    __syncthreads();

    double slot_s[K];
    uint32_t slot_c[K];  // gain class of the document in each slot
#pragma unroll
    for (int m = 0; m < K; m++) {
        slot_s[m] = NEG_INF;
        slot_c[m] = 0;
    }
    bool nan_seen = false;
    uint32_t dbg_rows = 0, dbg_batches = 0, dbg_ins = 0;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint32_t qn = a.qlen[q];
    uint32_t q_end = pos + qn;  // one past the current query's last position
    const uint32_t ngp = (f + 3) >> 2;   // tile groups that hold prefix features
    const uint32_t sq0 = (f + 1) >> 2;   // first tile group that holds a suffix feature
    const uint32_t fgrp = f >> 2, fsub = f & 3;

    for (uint32_t pb = pos; pb < run_end; pb += WAVE) {
        // ---------------- phase S: lane = document ----------------
        const float4* __restrict__ tile = a.xb + (size_t)(pb >> 6) * dq * 64 + lane;
        const uint32_t mycls = a.gcls[pb + lane];
        double P = 0.0;  // shared prefix: features < f in order (dense_dataset.rs:71-74)
        if (ngp > 0) {
            float4 xa[4], xc[4];
#pragma unroll
            for (int u = 0; u < 4; u++) xa[u] = tile[(size_t)((uint32_t)u < ngp ? (uint32_t)u : ngp - 1) * 64];
            for (uint32_t j4 = 0; j4 < ngp; j4 += 4) {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t jn = j4 + 4 + u;
                    xc[u] = tile[(size_t)(jn < ngp ? jn : ngp - 1) * 64];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (j4 + u < ngp) {
                        const double* wp = wpre + (j4 + u) * 4;
                        double p0 = (double)xa[u].x * wp[0];
                        P = P + p0;
                        double p1 = (double)xa[u].y * wp[1];
                        P = P + p1;
                        double p2 = (double)xa[u].z * wp[2];
                        P = P + p2;
                        double p3 = (double)xa[u].w * wp[3];
                        P = P + p3;
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) xa[u] = xc[u];
            }
        }
        const double xf = (double)((const float*)(tile + (size_t)fgrp * 64))[fsub];
        // two tile groups of the suffix are requested before the candidate initialisation
        float4 x0 = tile[(size_t)(sq0 < dq ? sq0 : dq - 1) * 64];
        double sc[CT];
#pragma unroll
        for (int c = 0; c < CT; c++) {
            double prod = xf * cwl[c];
            sc[c] = P + prod;
        }
        for (uint32_t j4 = sq0; j4 < dq; j4++) {
            const float4 x2 = tile[(size_t)(j4 + 1 < dq ? j4 + 1 : dq - 1) * 64];
            const double* wp = wsuf + j4 * 4;
            const float xs[4] = {x0.x, x0.y, x0.z, x0.w};
#pragma unroll
            for (int u = 0; u < 4; u++) {
                double prod = (double)xs[u] * wp[u];
#pragma unroll
                for (int c = 0; c < CT; c++) sc[c] = sc[c] + prod;
            }
            x0 = x2;
        }
        // ---------------- filter + transpose + phase K: lane = candidate ----------------
        const uint32_t nvalid = (run_end - pb) < (uint32_t)WAVE ? (run_end - pb) : (uint32_t)WAVE;
        uint32_t lo = 0;
        while (lo < nvalid) {
            const uint32_t seg_hi = (q_end - pb) < nvalid ? (q_end - pb) : nvalid;
            uint64_t remaining = lane_range_mask(lo, seg_hi);
            if (a.debug & 1) remaining = 0ull;
            while (remaining != 0ull) {
                if (!(a.debug & 2)) {
                    bool p = false;
#pragma unroll
                    for (int c = 0; c < CT; c++) p |= (sc[c] >= thr[c]) | (sc[c] != sc[c]);
                    remaining &= __ballot(p);
                    if (remaining == 0ull) break;
                }
                const bool mine = (remaining >> lane) & 1ull;
                const uint32_t myrank = __popcll(remaining & lt_mask);
                const bool in_batch = mine && myrank < (uint32_t)RB;
                if (in_batch) {
                    double* row = tr + myrank * LS_ROWPAD;
#pragma unroll
                    for (int c = 0; c < CT; c++) row[c] = sc[c];
                    rowcls[myrank] = mycls;
                }
                const uint64_t batch_mask = __ballot(in_batch);
                const uint32_t nb = __popcll(batch_mask);
                remaining &= ~batch_mask;
                dbg_rows += nb;
                dbg_batches++;
                __syncthreads();
                for (uint32_t rr = 0; rr < nb; rr++) {
                    const double e = lane < (uint32_t)CT ? tr[rr * LS_ROWPAD + lane] : NEG_INF;
                    const uint32_t ep = rowcls[rr];
                    nan_seen |= (e != e);
                    if (__ballot(e >= slot_s[K - 1]) != 0ull) {
                        dbg_ins++;
                        // ordered insertion: e goes above every slot it ties or beats (later document
                        // wins ties = reference tie-break in the reverse layout).  Empty slots hold
                        // -inf and therefore lose against every non-NaN score, including -inf itself.
                        bool beat[K];
#pragma unroll
                        for (int m = 0; m < K; m++) beat[m] = (e >= slot_s[m]);
#pragma unroll
                        for (int m = K - 1; m >= 1; m--) {
                            slot_s[m] = beat[m - 1] ? slot_s[m - 1] : (beat[m] ? e : slot_s[m]);
                            slot_c[m] = beat[m - 1] ? slot_c[m - 1] : (beat[m] ? ep : slot_c[m]);
                        }
                        slot_s[0] = beat[0] ? e : slot_s[0];
                        slot_c[0] = beat[0] ? ep : slot_c[0];
                    }
                }
                // publish each candidate's K-th best score; unused candidate lanes never admit a document
                thr[lane] = (lane < ncand) ? slot_s[K - 1] : __builtin_huge_val();
                __syncthreads();
            }
            if (q_end - pb > (uint32_t)WAVE) break;  // the query continues in the next tile
            // ---- the current query is complete: NDCG@k (src/evaluators.rs:255-272,350-380) ----
            if (lane < ncand) {
                const uint32_t L = (uint32_t)a.depth < qn ? (uint32_t)a.depth : qn;
                double dcg = 0.0;
#pragma unroll
                for (int i = 0; i < K; i++) {
                    if ((uint32_t)i < L) {
                        double term = a.dcgtab[(size_t)slot_c[i] * LS_KT + i];
                        dcg = dcg + term;
                    }
                }
                const double norm = a.norms[q];
                double val = 0.0;
                int fl = 0;
                if (norm == norm) {
                    if (dcg > norm) fl |= FLAG_ACTUAL_GT_IDEAL;
                    val = dcg / norm;
                }
                if (fl && !a.debug) atomicOr(a.flags, fl);
                a.M[(size_t)q * a.ldm + (size_t)g * 64 + lane] = val;
            }
            lo = seg_hi;
            q++;
            if (q >= q1) break;
#pragma unroll
            for (int m = 0; m < K; m++) {
                slot_s[m] = NEG_INF;
                slot_c[m] = 0;
            }
            qn = a.qlen[q];
            q_end += qn;
            __syncthreads();  // every lane has read thr[] for the finished query
            thr[lane] = NEG_INF;
            __syncthreads();
        }
    }
    if (nan_seen && !a.debug) atomicOr(a.flags, FLAG_NAN_SCORE);
    if ((a.debug & 16) && lane == 0) {
        atomicAdd(a.dbg_counters + 0, (unsigned long long)dbg_rows);
        atomicAdd(a.dbg_counters + 1, (unsigned long long)dbg_batches);
        atomicAdd(a.dbg_counters + 2, (unsigned long long)dbg_ins);
        atomicAdd(a.dbg_counters + 3, (unsigned long long)(run_end - pos));
    }
}

// ----------------------------------------------------------------------------------------------
// Full-ranking line search (AP / RR / NDCG without depth or with depth > 20): two kernels per
// chunk of line groups.  Kernel A scores every candidate exactly like phase S above and writes one
// 512-byte row per (document, group): rows[(p * GC + gl) * 64 + c].  Kernel B (lane = candidate)
// ranks the documents that can contribute to the metric by counting, for each of them, the
// documents that precede it in the reference order, then walks the ranks in order.
// ----------------------------------------------------------------------------------------------

struct FSArgs {
    const float4* xb;
    const uint32_t* run_pos;
    const uint32_t* run_docs;
    const uint32_t* run_order;
    const uint32_t* gfeat;   // [GC]
    const double* gw;        // [GC][4*dq]
    const double* gcand;     // [GC][64]
    double* rows;            // [np][GC][64]
    int* flags;
    uint32_t dq, d, nruns, GC;
};

template <int CT, int RB>
__global__ __launch_bounds__(WAVE, LS_WAVES_PER_SIMD) void linesearch_scores_kernel(FSArgs a) {
    extern __shared__ __attribute__((aligned(16))) double wdyn[];  // wpre[4*dq], wsuf[4*dq]
    __shared__ double tr[RB * LS_ROWPAD];
    __shared__ double cwl[WAVE];
    const uint32_t lane = threadIdx.x;
    const uint32_t blk = blockIdx.x;
    const uint32_t xcd = blk & 7u, seq = blk >> 3;
    const uint32_t g = seq % a.GC;
    const uint32_t ri = (seq / a.GC) * 8u + xcd;
    if (ri >= a.nruns) return;
    const uint32_t r = a.run_order[ri];
    const uint32_t pos = a.run_pos[r];
    const uint32_t run_end = pos + a.run_docs[r];
    const uint32_t f = a.gfeat[g];
    const uint32_t dq = a.dq, dp = a.dq * 4, d = a.d;
    const double* __restrict__ w = a.gw + (size_t)g * dp;
    double* wpre = wdyn;
    double* wsuf = wdyn + dp;
    for (uint32_t j = lane; j < dp; j += WAVE) {
        const double wv = j < d ? w[j] : 0.0;
        wpre[j] = j < f ? wv : 0.0;
        wsuf[j] = j > f ? wv : 0.0;
    }
    cwl[lane] = a.gcand[(size_t)g * 64 + lane];
    __syncthreads();
    bool nan_seen = false;
    const uint32_t ngp = (f + 3) >> 2;
    const uint32_t sq0 = (f + 1) >> 2;
    const uint32_t fgrp = f >> 2, fsub = f & 3;
    for (uint32_t pb = pos; pb < run_end; pb += WAVE) {
        const float4* __restrict__ tile = a.xb + (size_t)(pb >> 6) * dq * 64 + lane;
        double P = 0.0;
        for (uint32_t j4 = 0; j4 < ngp; j4++) {
            const float4 x = tile[(size_t)j4 * 64];
            const double* wp = wpre + j4 * 4;
            double p0 = (double)x.x * wp[0];
            P = P + p0;
            double p1 = (double)x.y * wp[1];
            P = P + p1;
            double p2 = (double)x.z * wp[2];
            P = P + p2;
            double p3 = (double)x.w * wp[3];
            P = P + p3;
        }
        const double xf = (double)((const float*)(tile + (size_t)fgrp * 64))[fsub];
        float4 x0 = tile[(size_t)(sq0 < dq ? sq0 : dq - 1) * 64];
        double sc[CT];
#pragma unroll
        for (int c = 0; c < CT; c++) {
            double prod = xf * cwl[c];
            sc[c] = P + prod;
        }
        for (uint32_t j4 = sq0; j4 < dq; j4++) {
            const float4 x2 = tile[(size_t)(j4 + 1 < dq ? j4 + 1 : dq - 1) * 64];
            const double* wp = wsuf + j4 * 4;
            const float xs[4] = {x0.x, x0.y, x0.z, x0.w};
#pragma unroll
            for (int u = 0; u < 4; u++) {
                double prod = (double)xs[u] * wp[u];
#pragma unroll
                for (int c = 0; c < CT; c++) sc[c] = sc[c] + prod;
            }
            x0 = x2;
        }
        // transpose RB rows at a time and store them as coalesced 512-byte rows
        const uint32_t nvalid = (run_end - pb) < (uint32_t)WAVE ? (run_end - pb) : (uint32_t)WAVE;
        for (uint32_t b0 = 0; b0 < nvalid; b0 += RB) {
            if (lane >= b0 && lane < b0 + RB) {
                double* row = tr + (lane - b0) * LS_ROWPAD;
#pragma unroll
                for (int c = 0; c < CT; c++) row[c] = sc[c];
            }
            __syncthreads();
            const uint32_t nb = (nvalid - b0) < (uint32_t)RB ? (nvalid - b0) : (uint32_t)RB;
            for (uint32_t rr = 0; rr < nb; rr++) {
                const double e = lane < (uint32_t)CT ? tr[rr * LS_ROWPAD + lane] : 0.0;
                nan_seen |= (e != e);
                a.rows[((size_t)(pb + b0 + rr) * a.GC + g) * 64 + lane] = e;
            }
            __syncthreads();
        }
    }
    if (nan_seen) atomicOr(a.flags, FLAG_NAN_SCORE);
}

struct RMArgs {
    const double* rows;        // [np][GC][64]
    const uint32_t* qstart;
    const uint32_t* qlen;
    const uint32_t* qnpos;     // [nq] documents with gain > 0 (a prefix of the query: gain-descending layout)
    const uint32_t* qnneg;     // [nq] documents with gain < 0 (a suffix)
    const uint32_t* gcls;      // [np]
    const double* termtab;     // [ncls][tablen]: (2^gain - 1) / log2(rank + 2)
    const double* norms;
    const uint32_t* gncand;    // [GC]
    const uint32_t* qlist;     // queries of this size class
    double* M;                 // [nq][ldm]
    int* flags;
    uint32_t GC, ldm, col0, tablen;
    int measure, depth;
};

constexpr int RM_AB = 32;  // documents ranked per sweep over the query
constexpr int RM_PF = 8;   // rows loaded together while sweeping

// One wave per (query, group), lane = candidate.  For each block of RM_AB "interesting" documents
// (relevant ones for AP/RR, non-zero-gain ones for NDCG) the wave sweeps every document row of the
// query once and counts, per lane, how many documents precede each of the RM_AB in the reference
// order (score desc; ties: later position first = gain asc, id asc; src/evaluators.rs:34-49).
// Ranks are then scattered into a per-lane rank table in LDS and consumed in rank order, so the
// floating-point sums are formed exactly like the reference forms them.
__global__ __launch_bounds__(512) void rank_metric_kernel(RMArgs a) {
    extern __shared__ unsigned char at_rank[];  // [npad][64] class id (NDCG) or relevance flag (AP)
    __shared__ uint32_t best_shared[WAVE];
    // long queries get several waves per block: the waves split the blocks of documents to rank and
    // share the per-lane rank table; wave 0 then walks the ranks
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const uint32_t q = a.qlist[blockIdx.x];
    const uint32_t g = blockIdx.y;
    const uint32_t base = a.qstart[q], n = a.qlen[q];
    const uint32_t npos = a.qnpos[q], nneg = a.qnneg[q];
    const uint32_t ncand = a.gncand[g];
    const size_t rstride = (size_t)a.GC * 64;
    const double* __restrict__ rows = a.rows + ((size_t)base * a.GC + g) * 64 + lane;
    const bool ndcg = a.measure == M_NDCG;
    // documents whose rank matters: [0, npos) and, for NDCG, also [n - nneg, n)
    const uint32_t ninteresting = npos + (ndcg ? nneg : 0u);
    for (uint32_t r = wave; r < n; r += nwaves) at_rank[r * 64 + lane] = 0xFF;
    if (wave == 0) best_shared[lane] = 0xFFFFFFFFu;
    __syncthreads();
    uint32_t best_rank = 0xFFFFFFFFu;  // RR: rank of the best relevant document
    for (uint32_t a0 = wave * RM_AB; a0 < ninteresting; a0 += nwaves * RM_AB) {
        double sa[RM_AB];
        uint32_t cnt[RM_AB], apos[RM_AB];
#pragma unroll
        for (int t = 0; t < RM_AB; t++) {
            uint32_t ai = a0 + t < ninteresting ? a0 + t : ninteresting - 1;
            apos[t] = ai < npos ? ai : (n - nneg) + (ai - npos);
            sa[t] = rows[(size_t)apos[t] * rstride];
            cnt[t] = 0;
        }
        const uint32_t amin = apos[0], amax = apos[RM_AB - 1];
        // Sweep the query's rows RM_PF at a time (independent loads in flight).  Documents stored
        // before the block precede a only if s_k > s_a (a later document wins ties); documents stored
        // after it also win ties (s_k >= s_a); inside the block's position range the rule is per pair.
        for (uint32_t k0 = 0; k0 < n; k0 += RM_PF) {
            double sk[RM_PF];
#pragma unroll
            for (int u = 0; u < RM_PF; u++) sk[u] = rows[(size_t)(k0 + u < n ? k0 + u : n - 1) * rstride];
            if (k0 + RM_PF <= amin) {
#pragma unroll
                for (int u = 0; u < RM_PF; u++)
#pragma unroll
                    for (int t = 0; t < RM_AB; t++) cnt[t] += (sk[u] > sa[t]) ? 1u : 0u;
            } else if (k0 > amax && k0 + RM_PF <= n) {
#pragma unroll
                for (int u = 0; u < RM_PF; u++)
#pragma unroll
                    for (int t = 0; t < RM_AB; t++) cnt[t] += (sk[u] >= sa[t]) ? 1u : 0u;
            } else {
#pragma unroll
                for (int u = 0; u < RM_PF; u++) {
                    const uint32_t k = k0 + u;
                    if (k < n) {
#pragma unroll
                        for (int t = 0; t < RM_AB; t++) {
                            const bool before =
                                (k > apos[t]) ? (sk[u] >= sa[t]) : ((k < apos[t]) ? (sk[u] > sa[t]) : false);
                            cnt[t] += before ? 1u : 0u;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < RM_AB; t++) {
            if (a0 + t < ninteresting) {
                if (ndcg) {
                    at_rank[cnt[t] * 64 + lane] = (unsigned char)a.gcls[base + apos[t]];
                } else {
                    at_rank[cnt[t] * 64 + lane] = 1;
                    best_rank = cnt[t] < best_rank ? cnt[t] : best_rank;
                }
            }
        }
    }
    if (!ndcg && a.measure != M_AP && best_rank != 0xFFFFFFFFu) atomicMin(&best_shared[lane], best_rank);
    __syncthreads();
    if (wave != 0) return;
    best_rank = best_shared[lane];
    double val = 0.0;
    int fl = 0;
    if (ndcg) {
        // src/evaluators.rs:255-272,350-380: terms in rank order from 0.0; zero-gain ranks add +0.0 (skipped)
        const uint32_t L = a.depth >= 0 ? ((uint32_t)a.depth < n ? (uint32_t)a.depth : n) : n;
        double dcg = 0.0;
        for (uint32_t r = 0; r < L; r++) {
            const uint32_t c = at_rank[r * 64 + lane];
            if (c != 0xFF) {
                double term = a.termtab[(size_t)c * a.tablen + r];
                dcg = dcg + term;
            }
        }
        const double norm = a.norms[q];
        if (norm == norm) {
            if (dcg > norm) fl |= FLAG_ACTUAL_GT_IDEAL;
            val = dcg / norm;
        }
    } else if (a.measure == M_AP) {
        // src/evaluators.rs:422-447
        uint32_t num_rel = (uint32_t)a.norms[q];
        if (num_rel == 0) num_rel = npos;
        if (num_rel != 0) {
            int recall_points = 0;
            double sum_precision = 0.0;
            for (uint32_t r = 0; r < n; r++) {
                if (at_rank[r * 64 + lane] != 0xFF) {
                    recall_points += 1;
                    sum_precision += (double)recall_points / (double)(r + 1);
                }
            }
            val = sum_precision / (double)num_rel;
        }
    } else {
        // src/evaluators.rs:239-252
        if (best_rank != 0xFFFFFFFFu) val = 1.0 / (double)(best_rank + 1);
    }
    if (lane < ncand) {
        if (fl) atomicOr(a.flags, fl);
        a.M[(size_t)q * a.ldm + a.col0 + (size_t)g * 64 + lane] = val;
    }
}

// ----------------------------------------------------------------------------------------------
// DeviceDataset
// ----------------------------------------------------------------------------------------------

struct DeviceDataset::Impl {
    int device = 0;
    hipStream_t stream = nullptr;
    size_t n = 0, d = 0, nq = 0, np = 0, dq = 0, maxlen = 0, nruns = 0;
    bool nonfinite = false;            // X holds inf/NaN: zero-weight masking is not exact -> no fused path
    std::vector<uint32_t> perm_host;   // [np] original instance id or IDX_INVALID (padding)
    DevBuf<float> xb, gain;
    DevBuf<double> gexp, disc;
    DevBuf<uint32_t> qstart, qlen, qtight, perm, rank, run_q0, run_q1, run_pos, run_docs, run_order, gcls, qlist;
    struct SizeClass {
        uint32_t npad, offset, count;
    };
    std::vector<SizeClass> size_classes;  // queries bucketed by next power of two of their length
    DevBuf<uint32_t> qnpos, qnneg;
    DevBuf<double> termtab, rows;
    size_t ncls = 0, tablen = 0;
    DevBuf<double> dcgtab;
    DevBuf<int> flags;
    DevBuf<unsigned long long> dbgc;
    // work buffers
    DevBuf<double> scores, acc, weights, M, means, partial, norms, gw, gcand;
    DevBuf<uint32_t> gfeat, gncand;
    DevBuf<TreeNodeDev> nodes;
    DevBuf<int32_t> roots;
    DevBuf<double> tweights;
    DevBuf<uint64_t> forest;
    DevBuf<uint32_t> batch_off, batch_first, tree_meta;
    size_t scores_slots = 0;
    size_t last_ldm = 0, last_cols = 0;
    int host_flags = 0;
    std::mutex mu;

    bool bind(std::string* err) {
        FR_HIP(hipSetDevice(device));
        return true;
    }
    bool pull_flags(std::string* err) {
        int v = 0;
        FR_HIP(hipMemcpyAsync(&v, flags.p, sizeof(int), hipMemcpyDeviceToHost, stream));
        FR_HIP(hipStreamSynchronize(stream));
        if (v) {
            host_flags |= v;
            FR_HIP(hipMemsetAsync(flags.p, 0, sizeof(int), stream));
        }
        return true;
    }
};

DeviceDataset::DeviceDataset() : impl_(new Impl()) {}
DeviceDataset::~DeviceDataset() {
    if (impl_) {
        (void)hipSetDevice(impl_->device);
        if (impl_->stream) {
            (void)hipStreamSynchronize(impl_->stream);
            (void)hipStreamDestroy(impl_->stream);
        }
        delete impl_;
    }
}

size_t DeviceDataset::n() const { return impl_->n; }
size_t DeviceDataset::d() const { return impl_->d; }
size_t DeviceDataset::nq() const { return impl_->nq; }
size_t DeviceDataset::max_query_len() const { return impl_->maxlen; }
size_t DeviceDataset::last_ldm() const { return impl_->last_ldm; }
size_t DeviceDataset::hbm_bytes() const {
    const Impl& m = *impl_;
    return m.xb.bytes() + m.gain.bytes() + m.gexp.bytes() + m.disc.bytes() + m.qstart.bytes() + m.qlen.bytes() +
           m.qtight.bytes() + m.perm.bytes();
}

int DeviceDataset::take_flags() {
    std::lock_guard<std::mutex> lk(impl_->mu);
    int v = impl_->host_flags;
    impl_->host_flags = 0;
    return v;
}

template <typename T>
static bool upload(DevBuf<T>& buf, const std::vector<T>& host, std::string* err) {
    if (!buf.ensure(std::max<size_t>(host.size(), 1), err)) return false;
    if (!host.empty()) FR_HIP(hipMemcpy(buf.p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return true;
}

std::shared_ptr<DeviceDataset> DeviceDataset::create(const HostCSR& csr, std::string* err) {
    auto fail = [&](const std::string& m) {
        if (err) *err = m;
        return std::shared_ptr<DeviceDataset>();
    };
    std::string e2;
    if (device_count(&e2) <= 0)
        return fail("no MI355X/HIP device available for the fastrank_amd compute path (" +
                    (e2.empty() ? std::string("device count is 0") : e2) + ")");
    if (csr.n == 0 || csr.d == 0 || csr.nq == 0) return fail("empty dataset");
    if (csr.n >= 0xF0000000ull) return fail("dataset too large for 32-bit document positions");
    std::shared_ptr<DeviceDataset> ds(new DeviceDataset());
    Impl& m = *ds->impl_;
    if (hipGetDevice(&m.device) != hipSuccess) return fail("hipGetDevice failed");
    auto chk = [&](hipError_t e, const char* what) {
        if (e == hipSuccess) return true;
        if (err) *err = std::string("HIP error: ") + hipGetErrorString(e) + " at " + what;
        return false;
    };
    if (!chk(hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking), "hipStreamCreate")) return nullptr;
    m.n = csr.n;
    m.d = csr.d;
    m.nq = csr.nq;
    m.dq = (csr.d + 3) / 4;

    // ---- runs: consecutive queries packed into whole 64-document tiles ----------------------
    size_t target = 768;
    if (const char* t = getenv("FR_RUN_DOCS")) target = std::max<size_t>(64, (size_t)atoll(t));
    std::vector<uint32_t> qstart(m.nq), qlen(m.nq), qtight(m.nq + 1), run_q0, run_q1, run_pos, run_docs;
    {
        size_t pos = 0, cur_docs = 0;
        uint32_t cur_q0 = 0;
        auto close_run = [&](uint32_t q_end) {
            run_q0.push_back(cur_q0);
            run_q1.push_back(q_end);
            run_pos.push_back((uint32_t)(pos - cur_docs));
            run_docs.push_back((uint32_t)cur_docs);
            pos = (pos + 63) / 64 * 64;
            cur_docs = 0;
            cur_q0 = q_end;
        };
        for (size_t q = 0; q < m.nq; q++) {
            size_t len = csr.qoff[q + 1] - csr.qoff[q];
            m.maxlen = std::max(m.maxlen, len);
            if (cur_docs > 0 && cur_docs + len > target) close_run((uint32_t)q);
            qstart[q] = (uint32_t)pos;
            qlen[q] = (uint32_t)len;
            qtight[q] = csr.qoff[q];
            pos += len;
            cur_docs += len;
        }
        if (cur_docs > 0) close_run((uint32_t)m.nq);
        qtight[m.nq] = csr.qoff[m.nq];
        m.np = pos;
        if (m.np >= 0xFFFFFF00ull) return fail("dataset too large for 32-bit document positions");
    }
    m.nruns = run_q0.size();
    std::vector<uint32_t> run_order(m.nruns);
    for (size_t r = 0; r < m.nruns; r++) run_order[r] = (uint32_t)r;
    // longest-first schedule so the biggest runs do not form the tail of a launch
    std::stable_sort(run_order.begin(), run_order.end(), [&](uint32_t x, uint32_t y) { return run_docs[x] > run_docs[y]; });

    // ---- size classes for the general (sort) evaluator: LDS sized per class, not per dataset maximum
    std::vector<uint32_t> qlist(m.nq);
    {
        auto npad_of = [](uint32_t len) {
            uint32_t p2 = 64;
            while (p2 < len) p2 <<= 1;
            return p2;
        };
        for (size_t q = 0; q < m.nq; q++) qlist[q] = (uint32_t)q;
        std::stable_sort(qlist.begin(), qlist.end(), [&](uint32_t x, uint32_t y) { return npad_of(qlen[x]) < npad_of(qlen[y]); });
        for (size_t k = 0; k < m.nq;) {
            uint32_t np2 = npad_of(qlen[qlist[k]]);
            size_t e = k;
            while (e < m.nq && npad_of(qlen[qlist[e]]) == np2) e++;
            m.size_classes.push_back({np2, (uint32_t)k, (uint32_t)(e - k)});
            k = e;
        }
    }

    // ---- padded per-position arrays ------------------------------------------------------------
    m.perm_host.assign(m.np, IDX_INVALID);
    std::vector<float> gain(m.np, 0.0f);
    std::vector<double> gexp(m.np, 0.0);
    for (size_t q = 0; q < m.nq; q++) {
        for (uint32_t k = 0; k < qlen[q]; k++) {
            size_t p = (size_t)qstart[q] + k, t = (size_t)csr.qoff[q] + k;
            m.perm_host[p] = csr.perm[t];
            gain[p] = csr.gain[t];
            // (2^g - 1) with the platform libm, exactly like 2.0_f64.powf(gain) - 1.0
            // (src/evaluators.rs:266-270); g is the f32 gain widened to f64.
            gexp[p] = std::pow(2.0, (double)csr.gain[t]) - 1.0;
        }
    }

    // ---- gain classes and the per-class DCG term table: term(c, i) = (2^g_c - 1) / log2(i + 2), the
    // exact expression of src/evaluators.rs:266-270 evaluated once per (class, rank) on the host
    std::vector<uint32_t> gcls(m.np, 0);
    std::vector<double> dcgtab;
    {
        std::map<uint32_t, uint32_t> cls_of_bits;
        std::vector<float> cls_gain;
        for (size_t p = 0; p < m.np; p++) {
            if (m.perm_host[p] == IDX_INVALID) continue;
            float gv = gain[p] == 0.0f ? 0.0f : gain[p];  // -0.0 and +0.0 are one class
            uint32_t bits;
            std::memcpy(&bits, &gv, sizeof(bits));
            auto it = cls_of_bits.find(bits);
            if (it == cls_of_bits.end()) {
                it = cls_of_bits.emplace(bits, (uint32_t)cls_gain.size()).first;
                cls_gain.push_back(gv);
            }
            gcls[p] = it->second;
        }
        if (cls_gain.empty()) cls_gain.push_back(0.0f);
        dcgtab.resize(cls_gain.size() * LS_KT);
        for (size_t c = 0; c < cls_gain.size(); c++)
            for (int i = 0; i < LS_KT; i++)
                dcgtab[c * LS_KT + i] = (std::pow(2.0, (double)cls_gain[c]) - 1.0) / std::log2((double)i + 2.0);
    }

    // per-query counts of positive / negative gains (documents are stored gain-descending, so these are a
    // prefix / suffix of the query) and the full-depth term table for the rank-counting evaluator
    std::vector<uint32_t> qnpos(m.nq, 0), qnneg(m.nq, 0);
    std::vector<double> termtab;
    {
        for (size_t q = 0; q < m.nq; q++)
            for (uint32_t k = 0; k < qlen[q]; k++) {
                float gv = gain[(size_t)qstart[q] + k];
                qnpos[q] += gv > 0.0f;
                qnneg[q] += gv < 0.0f;
            }
        m.ncls = dcgtab.size() / LS_KT;
        m.tablen = std::max<size_t>(m.maxlen, 1);
        if (m.ncls <= 255 && m.ncls * m.tablen <= (size_t(64) << 20)) {
            termtab.resize(m.ncls * m.tablen);
            for (size_t c = 0; c < m.ncls; c++) {
                const double ge = dcgtab[c * LS_KT] * std::log2(2.0);  // = 2^g - 1 (term at rank 0, log2(2) = 1)
                for (size_t r = 0; r < m.tablen; r++) termtab[c * m.tablen + r] = ge / std::log2((double)r + 2.0);
            }
        }
    }

    // ---- feature tiles: built on the host in slabs (threads over tiles), uploaded slab by slab.
    // One-time cost; SURVEY 8d excludes it from evals/s and bench.py reports it separately.
    const size_t ntiles = m.np / 64;
    const size_t tile_floats = m.dq * 256;
    if (!m.xb.ensure(ntiles * tile_floats, err)) return nullptr;
    {
        const size_t slab_tiles = std::max<size_t>(1, (size_t(256) << 20) / (tile_floats * sizeof(float)));
        std::vector<float> slab(std::min(slab_tiles, ntiles) * tile_floats);
        unsigned hw = std::thread::hardware_concurrency();
        const size_t nthreads = std::max<size_t>(1, std::min<size_t>(hw ? hw : 1, 32));
        std::vector<char> bad(nthreads, 0);
        for (size_t t0 = 0; t0 < ntiles; t0 += slab_tiles) {
            const size_t tn = std::min(slab_tiles, ntiles - t0);
            std::fill(slab.begin(), slab.begin() + tn * tile_floats, 0.0f);
            auto work = [&](size_t tid) {
                for (size_t t = t0 + tid; t < t0 + tn; t += nthreads) {
                    float* tile = slab.data() + (t - t0) * tile_floats;
                    for (size_t l = 0; l < 64; l++) {
                        uint32_t id = m.perm_host[t * 64 + l];
                        if (id == IDX_INVALID) continue;
                        const float* row = csr.x + (size_t)id * csr.d;
                        for (size_t j = 0; j < csr.d; j++) {
                            float v = row[j];
                            if (!std::isfinite(v)) bad[tid] = 1;
                            tile[(j >> 2) * 256 + l * 4 + (j & 3)] = v;
                        }
                    }
                }
            };
            std::vector<std::thread> pool;
            for (size_t tid = 1; tid < nthreads; tid++) pool.emplace_back(work, tid);
            work(0);
            for (auto& th : pool) th.join();
            if (!chk(hipMemcpy(m.xb.p + t0 * tile_floats, slab.data(), tn * tile_floats * sizeof(float),
                               hipMemcpyHostToDevice),
                     "upload feature tiles"))
                return nullptr;
        }
        for (char b : bad) m.nonfinite = m.nonfinite || b;
    }
    {
        size_t nd = std::max<size_t>(m.maxlen, 64);
        std::vector<double> disc(nd);
        for (size_t i = 0; i < nd; i++) disc[i] = std::log2((double)i + 2.0);
        if (!upload(m.disc, disc, err) || !upload(m.gexp, gexp, err) || !upload(m.gain, gain, err) ||
            !upload(m.qstart, qstart, err) || !upload(m.qlen, qlen, err) || !upload(m.qtight, qtight, err) ||
            !upload(m.perm, m.perm_host, err) || !upload(m.run_q0, run_q0, err) || !upload(m.run_q1, run_q1, err) ||
            !upload(m.run_pos, run_pos, err) || !upload(m.run_docs, run_docs, err) ||
            !upload(m.run_order, run_order, err) || !upload(m.gcls, gcls, err) || !upload(m.dcgtab, dcgtab, err) ||
            !upload(m.qlist, qlist, err) || !upload(m.qnpos, qnpos, err) || !upload(m.qnneg, qnneg, err) ||
            !upload(m.termtab, termtab, err))
            return nullptr;
        if (!m.flags.ensure(1, err) || !m.dbgc.ensure(4, err)) return nullptr;
        if (!chk(hipMemset(m.flags.p, 0, sizeof(int)), "clear flags")) return nullptr;
    }
    return ds;
}

static bool launch_means(const double* M, size_t ldm, size_t ncols, size_t nq, DevBuf<double>& partial,
                         DevBuf<double>& means, hipStream_t st, std::string* err) {
    const size_t nseg = (nq + MEAN_SEG - 1) / MEAN_SEG;
    if (!partial.ensure(std::max<size_t>(1, nseg) * ldm, err) || !means.ensure(ldm, err)) return false;
    {
        ProfScope ps("segment_sum_kernel", st);
        dim3 grid((unsigned)nseg, (unsigned)((ncols + 63) / 64));
        segment_sum_kernel<<<grid, 64, 0, st>>>(M, (uint32_t)ldm, (uint32_t)ncols, (uint32_t)nq, partial.p);
    }
    {
        ProfScope ps("final_mean_kernel", st);
        final_mean_kernel<<<dim3((unsigned)((ncols + 63) / 64)), 64, 0, st>>>(partial.p, (uint32_t)ldm, (uint32_t)ncols,
                                                                              (uint32_t)nseg, (uint32_t)nq, means.p);
    }
    FR_HIP(hipGetLastError());
    return true;
}

static inline dim3 grid1d(size_t n, unsigned bs) { return dim3((unsigned)((n + bs - 1) / bs)); }

bool DeviceDataset::score_linear(size_t B, const double* weights, std::string* err) {
    Impl& m = *impl_;
    std::lock_guard<std::mutex> lk(m.mu);
    if (!m.bind(err)) return false;
    if (B == 0) return true;
    const size_t dp = m.dq * 4;
    if (!m.scores.ensure(B * m.np, err) || !m.weights.ensure(B * dp, err)) return false;
    m.scores_slots = B;
    std::vector<double> wpad(B * dp, 0.0);
    for (size_t b = 0; b < B; b++) std::memcpy(&wpad[b * dp], weights + b * m.d, m.d * sizeof(double));
    FR_HIP(hipMemcpyAsync(m.weights.p, wpad.data(), wpad.size() * sizeof(double), hipMemcpyHostToDevice, m.stream));
    FR_HIP(hipStreamSynchronize(m.stream));  // wpad is a local
    {
        ProfScope ps("score_linear_kernel", m.stream);
        if (B >= 8) {
            dim3 grid((unsigned)((m.np + 255) / 256), (unsigned)((B + 7) / 8));
            score_linear_kernel<8><<<grid, 256, 0, m.stream>>>((const float4*)m.xb.p, (uint32_t)m.np, (uint32_t)m.dq,
                                                               m.weights.p, (uint32_t)B, m.scores.p);
        } else {
            dim3 grid((unsigned)((m.np + 255) / 256), (unsigned)B);
            score_linear_kernel<1><<<grid, 256, 0, m.stream>>>((const float4*)m.xb.p, (uint32_t)m.np, (uint32_t)m.dq,
                                                               m.weights.p, (uint32_t)B, m.scores.p);
        }
    }
    FR_HIP(hipGetLastError());
    return true;
}

bool DeviceDataset::score_single_feature(uint32_t fid, double dir, std::string* err) {
    Impl& m = *impl_;
    std::lock_guard<std::mutex> lk(m.mu);
    if (!m.bind(err)) return false;
    if (!m.scores.ensure(m.np, err)) return false;
    m.scores_slots = 1;
    if (fid >= m.d) {
        // Features::get -> None -> unwrap_or(0.0) for loaded data; dir * 0.0
        fill_kernel<<<grid1d(m.np, 256), 256, 0, m.stream>>>(m.scores.p, (uint32_t)m.np, dir * 0.0);
    } else {
        score_single_feature_kernel<<<grid1d(m.np, 256), 256, 0, m.stream>>>(m.xb.p, (uint32_t)m.np, (uint32_t)m.dq, fid,
                                                                             dir, m.scores.p);
    }
    FR_HIP(hipGetLastError());
    return true;
}

// Largest f32 <= split: (f64(x) <= split) <=> (x <= thr) for every non-NaN f32 x.
static float floor_to_f32(double split) {
    float t = (float)split;
    if ((double)t > split) t = std::nextafterf(t, -std::numeric_limits<float>::infinity());
    return t;
}

// Packs the forest for tree_ensemble_lds_kernel and launches it.  Returns false with an empty
// *err when the forest does not fit the compact encoding (caller falls back to the L2 walk).
bool DeviceDataset::try_score_trees_lds(const FlatTrees& t, std::string* err) {
    Impl& m = *impl_;
    if (err) err->clear();
    const size_t nt = t.root.size();
    if (nt == 0 || m.d >= 0xFFFF) return false;
    const size_t row_floats = m.dq * 4 + 1;
    unsigned bs = 0;
    const size_t lds_cap = 160 * 1024;
    size_t tree_bytes_cap = 0;
    for (unsigned cand : {256u, 128u, 64u}) {
        size_t rows_b = (size_t)cand * row_floats * sizeof(float);
        if (rows_b + 8 * 1024 + LDS_TREES_PER_BATCH * 24 <= lds_cap) {
            bs = cand;
            tree_bytes_cap = std::min<size_t>(24 * 1024, (lds_cap - rows_b - LDS_TREES_PER_BATCH * 24) / 16 * 16);
            break;
        }
    }
    if (bs == 0) return false;
    const size_t words_cap = tree_bytes_cap / 8;
    struct Packed {
        std::vector<uint64_t> words;  // node words followed by the f64 leaf table
        uint32_t nodes = 0, levels = 0;
    };
    std::vector<Packed> packed(nt);
    for (size_t k = 0; k < nt; k++) {
        Packed& pk = packed[k];
        std::vector<uint64_t> nodes;
        std::vector<double> leaves;
        struct Item { int32_t src; uint32_t dst; uint32_t depth; };
        std::vector<Item> work;
        nodes.push_back(0);
        work.push_back({t.root[k], 0u, 0u});
        while (!work.empty()) {
            Item it = work.back();
            work.pop_back();
            pk.levels = std::max(pk.levels, it.depth);
            if (t.fid[it.src] < 0) {
                if (leaves.size() >= 0xFFFF) return false;
                uint64_t word = ((uint64_t)leaves.size() << 48) | ((uint64_t)0xFFFFu << 32);
                nodes[it.dst] = word;
                leaves.push_back(t.split[it.src]);
            } else {
                if (nodes.size() + 2 > 0xFFFF || (uint32_t)t.fid[it.src] >= 0xFFFFu) return false;
                uint32_t left = (uint32_t)nodes.size();
                float thr = floor_to_f32(t.split[it.src]);
                uint32_t tb;
                std::memcpy(&tb, &thr, 4);
                nodes[it.dst] = ((uint64_t)left << 48) | ((uint64_t)(uint32_t)t.fid[it.src] << 32) | tb;
                nodes.push_back(0);
                nodes.push_back(0);
                work.push_back({t.lhs[it.src], left, it.depth + 1});
                work.push_back({t.rhs[it.src], left + 1, it.depth + 1});
            }
        }
        pk.nodes = (uint32_t)nodes.size();
        pk.words = nodes;
        for (double v : leaves) {
            uint64_t bits;
            std::memcpy(&bits, &v, 8);
            pk.words.push_back(bits);
        }
        if (pk.words.size() > words_cap) return false;
    }
    std::vector<uint64_t> forest;
    std::vector<uint32_t> batch_off(1, 0), batch_first(1, 0), meta(nt * 3);
    size_t cur_words = 0;
    for (size_t k = 0; k < nt; k++) {
        if (cur_words + packed[k].words.size() > words_cap || k - batch_first.back() >= LDS_TREES_PER_BATCH) {
            batch_off.push_back((uint32_t)forest.size());
            batch_first.push_back((uint32_t)k);
            cur_words = 0;
        }
        meta[k * 3 + 0] = (uint32_t)cur_words;
        meta[k * 3 + 1] = (uint32_t)(cur_words + packed[k].nodes);
        meta[k * 3 + 2] = packed[k].levels;
        forest.insert(forest.end(), packed[k].words.begin(), packed[k].words.end());
        cur_words += packed[k].words.size();
    }
    batch_off.push_back((uint32_t)forest.size());
    batch_first.push_back((uint32_t)nt);
    const size_t nbatch = batch_off.size() - 1;
    std::vector<double> tw = t.weight;
    tw.resize(nt, 1.0);
    if (!m.forest.ensure(forest.size(), err) || !m.batch_off.ensure(batch_off.size(), err) ||
        !m.batch_first.ensure(batch_first.size(), err) || !m.tree_meta.ensure(meta.size(), err) ||
        !m.tweights.ensure(nt, err))
        return false;
    FR_HIP(hipMemcpyAsync(m.forest.p, forest.data(), forest.size() * 8, hipMemcpyHostToDevice, m.stream));
    FR_HIP(hipMemcpyAsync(m.batch_off.p, batch_off.data(), batch_off.size() * 4, hipMemcpyHostToDevice, m.stream));
    FR_HIP(hipMemcpyAsync(m.batch_first.p, batch_first.data(), batch_first.size() * 4, hipMemcpyHostToDevice, m.stream));
    FR_HIP(hipMemcpyAsync(m.tree_meta.p, meta.data(), meta.size() * 4, hipMemcpyHostToDevice, m.stream));
    FR_HIP(hipMemcpyAsync(m.tweights.p, tw.data(), nt * sizeof(double), hipMemcpyHostToDevice, m.stream));
    FR_HIP(hipStreamSynchronize(m.stream));
    const size_t lds = words_cap * 8 + LDS_TREES_PER_BATCH * 24 + (size_t)bs * row_floats * sizeof(float);
    {
        ProfScope ps("tree_ensemble_kernel", m.stream);
        FR_HIP(hipFuncSetAttribute((const void*)tree_ensemble_lds_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
        tree_ensemble_lds_kernel<8><<<grid1d(m.np, bs), bs, lds, m.stream>>>(
            m.xb.p, (uint32_t)m.np, (uint32_t)m.dq, (uint32_t)m.d, m.forest.p, m.batch_off.p, m.tree_meta.p,
            m.batch_first.p, m.tweights.p, (uint32_t)nbatch, t.raw_single ? 1 : 0, (uint32_t)words_cap, m.scores.p);
    }
    FR_HIP(hipGetLastError());
    return true;
}

bool DeviceDataset::score_trees(const FlatTrees& t, std::string* err) {
    Impl& m = *impl_;
    std::lock_guard<std::mutex> lk(m.mu);
    if (!m.bind(err)) return false;
    if (!m.scores.ensure(m.np, err)) return false;
    m.scores_slots = 1;
    if (try_score_trees_lds(t, err)) return true;
    if (err && !err->empty()) return false;
    const size_t nt = t.root.size();
    // re-lay the forest out so that the two children of a node are adjacent (rhs = lhs + 1)
    std::vector<TreeNodeDev> nodes;
    std::vector<int32_t> roots(nt);
    nodes.reserve(t.fid.size() + nt);
    {
        std::vector<std::pair<int32_t, int32_t>> work;  // (source node, destination slot)
        for (size_t k = 0; k < nt; k++) {
            roots[k] = (int32_t)nodes.size();
            nodes.push_back(TreeNodeDev{0.0, -1, 0});
            work.emplace_back(t.root[k], roots[k]);
            while (!work.empty()) {
                auto [src, dst] = work.back();
                work.pop_back();
                nodes[dst].split = t.split[src];
                nodes[dst].fid = t.fid[src];
                nodes[dst].lhs = 0;
                if (t.fid[src] >= 0) {
                    int32_t kids = (int32_t)nodes.size();
                    nodes[dst].lhs = kids;
                    nodes.push_back(TreeNodeDev{0.0, -1, 0});
                    nodes.push_back(TreeNodeDev{0.0, -1, 0});
                    work.emplace_back(t.lhs[src], kids);
                    work.emplace_back(t.rhs[src], kids + 1);
                }
            }
        }
    }
    const size_t nn = nodes.size();
    if (!m.nodes.ensure(std::max<size_t>(nn, 1), err) || !m.roots.ensure(std::max<size_t>(nt, 1), err) ||
        !m.tweights.ensure(std::max<size_t>(nt, 1), err))
        return false;
    FR_HIP(hipMemcpyAsync(m.nodes.p, nodes.data(), nn * sizeof(TreeNodeDev), hipMemcpyHostToDevice, m.stream));
    FR_HIP(hipMemcpyAsync(m.roots.p, roots.data(), nt * sizeof(int32_t), hipMemcpyHostToDevice, m.stream));
    std::vector<double> tw = t.weight;
    tw.resize(nt, 1.0);
    FR_HIP(hipMemcpyAsync(m.tweights.p, tw.data(), nt * sizeof(double), hipMemcpyHostToDevice, m.stream));
    FR_HIP(hipStreamSynchronize(m.stream));
    const unsigned bs = 64;  // one tile per block (np is a multiple of 64)
    size_t lds = (size_t)bs * (m.dq * 4 + 1) * sizeof(float);
    {
        ProfScope ps("tree_ensemble_kernel", m.stream);
        if (lds <= 150 * 1024) {
            FR_HIP(hipFuncSetAttribute((const void*)tree_ensemble_kernel<true, 8>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            tree_ensemble_kernel<true, 8><<<grid1d(m.np, bs), bs, lds, m.stream>>>(
                m.xb.p, (uint32_t)m.np, (uint32_t)m.dq, (uint32_t)m.d, m.nodes.p, m.roots.p, m.tweights.p, (uint32_t)nt,
                t.raw_single ? 1 : 0, m.scores.p);
        } else {
            tree_ensemble_kernel<false, 8><<<grid1d(m.np, bs), bs, 0, m.stream>>>(
                m.xb.p, (uint32_t)m.np, (uint32_t)m.dq, (uint32_t)m.d, m.nodes.p, m.roots.p, m.tweights.p, (uint32_t)nt,
                t.raw_single ? 1 : 0, m.scores.p);
        }
    }
    FR_HIP(hipGetLastError());
    return true;
}

bool DeviceDataset::ensemble_begin(std::string* err) {
    Impl& m = *impl_;
    std::lock_guard<std::mutex> lk(m.mu);
    if (!m.bind(err)) return false;
    if (!m.acc.ensure(m.np, err)) return false;
    fill_kernel<<<grid1d(m.np, 256), 256, 0, m.stream>>>(m.acc.p, (uint32_t)m.np, 0.0);
    FR_HIP(hipGetLastError());
    return true;
}

bool DeviceDataset::ensemble_accumulate(double w, std::string* err) {
    Impl& m = *impl_;
    std::lock_guard<std::mutex> lk(m.mu);
    if (!m.bind(err)) return false;
    axpy_unfused_kernel<<<grid1d(m.np, 256), 256, 0, m.stream>>>(m.acc.p, m.scores.p, (uint32_t)m.np, w);
    FR_HIP(hipGetLastError());
    return true;
}

bool DeviceDataset::ensemble_finish(std::string* err) {
    Impl& m = *impl_;
    std::lock_guard<std::mutex> lk(m.mu);
    if (!m.bind(err)) return false;
    if (!m.scores.ensure(m.np, err)) return false;
    FR_HIP(hipMemcpyAsync(m.scores.p, m.acc.p, m.np * sizeof(double), hipMemcpyDeviceToDevice, m.stream));
    m.scores_slots = 1;
    return true;
}

bool DeviceDataset::download_scores(size_t b, double* out, size_t out_len, std::string* err) {
    Impl& m = *impl_;
    std::lock_guard<std::mutex> lk(m.mu);
    if (!m.bind(err)) return false;
    if (b >= m.scores_slots) {
        if (err) *err = "score slot out of range";
        return false;
    }
    std::vector<double> tmp(m.np);
    FR_HIP(hipMemcpyAsync(tmp.data(), m.scores.p + b * m.np, m.np * sizeof(double), hipMemcpyDeviceToHost, m.stream));
    FR_HIP(hipStreamSynchronize(m.stream));
    for (size_t p = 0; p < m.np; p++) {
        size_t id = m.perm_host[p];
        if (id != IDX_INVALID && id < out_len) out[id] = tmp[p];
    }
    return true;
}

bool DeviceDataset::metric_from_scores(int measure, int64_t depth, const double* norms, size_t B, bool want_rank,
                                       std::string* err) {
    Impl& m = *impl_;
    std::lock_guard<std::mutex> lk(m.mu);
    if (!m.bind(err)) return false;
    if (B == 0 || B > m.scores_slots) {
        if (err) *err = "metric_from_scores: no scores resident";
        return false;
    }
    size_t npad = 1;
    while (npad < m.maxlen) npad <<= 1;
    size_t lds = npad * (sizeof(double) + sizeof(uint32_t));
    if (lds > 160 * 1024 - 1024) {
        if (err)
            *err = "query with " + std::to_string(m.maxlen) +
                   " documents exceeds the LDS sort capacity of the MI355X path (8192 documents per query)";
        return false;
    }
    if (!m.M.ensure(m.nq * B, err) || !m.norms.ensure(m.nq, err)) return false;
    if (want_rank && !m.rank.ensure(m.n, err)) return false;
    FR_HIP(hipMemcpyAsync(m.norms.p, norms, m.nq * sizeof(double), hipMemcpyHostToDevice, m.stream));
    FR_HIP(hipStreamSynchronize(m.stream));
    FR_HIP(hipFuncSetAttribute((const void*)metric_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
    int dd = depth < 0 ? -1 : (depth > 0x7fffffff ? 0x7fffffff : (int)depth);
    {
        ProfScope ps("metric_sort_kernel", m.stream);
        for (const auto& sc : m.size_classes) {
            const size_t cls_lds = (size_t)sc.npad * (sizeof(double) + sizeof(uint32_t));
            dim3 grid((unsigned)sc.count, (unsigned)B);
            unsigned bs = sc.npad >= 512 ? 256 : 64;
            metric_sort_kernel<<<grid, bs, cls_lds, m.stream>>>(m.scores.p, (uint32_t)m.np, m.qstart.p, m.qlen.p,
                                                                m.qtight.p, m.gexp.p, m.gain.p, m.disc.p, m.norms.p,
                                                                measure, dd, (uint32_t)B, m.M.p,
                                                                want_rank ? m.rank.p : nullptr, m.perm.p, m.flags.p,
                                                                sc.npad, m.qlist.p + sc.offset);
        }
    }
    FR_HIP(hipGetLastError());
    m.last_ldm = B;
    m.last_cols = B;
    return m.pull_flags(err);
}

bool DeviceDataset::download_per_query(size_t B, double* out, std::string* err) {
    Impl& m = *impl_;
    std::lock_guard<std::mutex> lk(m.mu);
    if (!m.bind(err)) return false;
    if (B != m.last_ldm) {
        if (err) *err = "download_per_query: shape mismatch";
        return false;
    }
    FR_HIP(hipMemcpyAsync(out, m.M.p, m.nq * B * sizeof(double), hipMemcpyDeviceToHost, m.stream));
    FR_HIP(hipStreamSynchronize(m.stream));
    return true;
}

bool DeviceDataset::download_last_matrix(std::vector<double>* out, size_t* ldm, std::string* err) {
    Impl& m = *impl_;
    std::lock_guard<std::mutex> lk(m.mu);
    if (!m.bind(err)) return false;
    out->resize(m.nq * m.last_ldm);
    *ldm = m.last_ldm;
    FR_HIP(hipMemcpyAsync(out->data(), m.M.p, out->size() * sizeof(double), hipMemcpyDeviceToHost, m.stream));
    FR_HIP(hipStreamSynchronize(m.stream));
    return true;
}

bool DeviceDataset::download_rank(uint32_t* out, std::string* err) {
    Impl& m = *impl_;
    std::lock_guard<std::mutex> lk(m.mu);
    if (!m.bind(err)) return false;
    if (m.rank.cap < m.n) {
        if (err) *err = "no rank order resident";
        return false;
    }
    FR_HIP(hipMemcpyAsync(out, m.rank.p, m.n * sizeof(uint32_t), hipMemcpyDeviceToHost, m.stream));
    FR_HIP(hipStreamSynchronize(m.stream));
    return true;
}

bool DeviceDataset::reduce_means(size_t ncols, double* out, std::string* err) {
    Impl& m = *impl_;
    std::lock_guard<std::mutex> lk(m.mu);
    if (!m.bind(err)) return false;
    if (ncols == 0 || ncols > m.last_ldm) {
        if (err) *err = "reduce_means: shape mismatch";
        return false;
    }
    if (!launch_means(m.M.p, m.last_ldm, ncols, m.nq, m.partial, m.means, m.stream, err)) return false;
    FR_HIP(hipMemcpyAsync(out, m.means.p, ncols * sizeof(double), hipMemcpyDeviceToHost, m.stream));
    FR_HIP(hipStreamSynchronize(m.stream));
    return true;
}

bool DeviceDataset::linesearch_supported(int measure, int64_t depth) const {
    // zero-weight masking is exact only for finite features (inf * 0 = NaN)
    return measure == M_NDCG && depth >= 0 && depth <= 20 && !impl_->nonfinite && impl_->dq * 4 <= 2048;
}

template <int K, int CT>
static void launch_linesearch(const LSArgs& a, unsigned nblocks, size_t lds, hipStream_t st) {
    linesearch_ndcg_kernel<K, CT, 16><<<dim3(nblocks), dim3(WAVE), lds, st>>>(a);
}

template <int K>
static void dispatch_ct(const LSArgs& a, unsigned nblocks, size_t maxc, size_t lds, hipStream_t st) {
    if (maxc <= 4) launch_linesearch<K, 4>(a, nblocks, lds, st);
    else if (maxc <= 16) launch_linesearch<K, 16>(a, nblocks, lds, st);
    else if (maxc <= 32) launch_linesearch<K, 32>(a, nblocks, lds, st);
    else if (maxc <= 51) launch_linesearch<K, 51>(a, nblocks, lds, st);
    else launch_linesearch<K, 64>(a, nblocks, lds, st);
}

bool DeviceDataset::linesearch_ndcg(int64_t depth, const double* norms, const std::vector<LineGroup>& groups,
                                    std::vector<double>* means, std::string* err) {
    Impl& m = *impl_;
    std::lock_guard<std::mutex> lk(m.mu);
    if (!m.bind(err)) return false;
    if (!linesearch_supported(M_NDCG, depth)) {
        if (err) *err = "linesearch_ndcg: unsupported depth, feature count or non-finite features";
        return false;
    }
    const size_t G = groups.size();
    means->assign(G * 64, 0.0);
    if (G == 0) return true;
    const size_t dp = m.dq * 4;
    size_t maxc = 0;
    std::vector<uint32_t> gfeat(G), gncand(G);
    std::vector<double> gw(G * dp, 0.0), gcand(G * 64, 0.0);
    for (size_t g = 0; g < G; g++) {
        const LineGroup& lg = groups[g];
        if (lg.feature >= m.d || lg.weights.size() != m.d || lg.candidates.empty() || lg.candidates.size() > 64) {
            if (err) *err = "linesearch_ndcg: malformed line group";
            return false;
        }
        gfeat[g] = lg.feature;
        gncand[g] = (uint32_t)lg.candidates.size();
        maxc = std::max(maxc, lg.candidates.size());
        std::memcpy(&gw[g * dp], lg.weights.data(), m.d * sizeof(double));
        std::memcpy(&gcand[g * 64], lg.candidates.data(), lg.candidates.size() * sizeof(double));
    }
    const size_t ldm = G * 64;
    if (!m.M.ensure(m.nq * ldm, err) || !m.norms.ensure(m.nq, err) || !m.gfeat.ensure(G, err) ||
        !m.gncand.ensure(G, err) || !m.gw.ensure(G * dp, err) || !m.gcand.ensure(G * 64, err) ||
        !m.means.ensure(ldm, err))
        return false;
    FR_HIP(hipMemcpyAsync(m.norms.p, norms, m.nq * sizeof(double), hipMemcpyHostToDevice, m.stream));
    FR_HIP(hipMemcpyAsync(m.gfeat.p, gfeat.data(), G * sizeof(uint32_t), hipMemcpyHostToDevice, m.stream));
    FR_HIP(hipMemcpyAsync(m.gncand.p, gncand.data(), G * sizeof(uint32_t), hipMemcpyHostToDevice, m.stream));
    FR_HIP(hipMemcpyAsync(m.gw.p, gw.data(), G * dp * sizeof(double), hipMemcpyHostToDevice, m.stream));
    FR_HIP(hipMemcpyAsync(m.gcand.p, gcand.data(), G * 64 * sizeof(double), hipMemcpyHostToDevice, m.stream));
    LSArgs a;
    a.xb = (const float4*)m.xb.p;
    a.gcls = m.gcls.p;
    a.dcgtab = m.dcgtab.p;
    a.qstart = m.qstart.p;
    a.qlen = m.qlen.p;
    a.run_q0 = m.run_q0.p;
    a.run_q1 = m.run_q1.p;
    a.run_pos = m.run_pos.p;
    a.run_docs = m.run_docs.p;
    a.run_order = m.run_order.p;
    a.norms = m.norms.p;
    a.disc = m.disc.p;
    a.gfeat = m.gfeat.p;
    a.gw = m.gw.p;
    a.gcand = m.gcand.p;
    a.gncand = m.gncand.p;
    a.M = m.M.p;
    a.flags = m.flags.p;
    a.dbg_counters = m.dbgc.p;
    a.dq = (uint32_t)m.dq;
    a.d = (uint32_t)m.d;
    a.nruns = (uint32_t)m.nruns;
    a.G = (uint32_t)G;
    a.ldm = (uint32_t)ldm;
    a.depth = (int)depth;
    {
        const char* dbg = getenv("FR_LS_DEBUG");
        a.debug = dbg ? atoi(dbg) : 0;
        if (a.debug & 16) FR_HIP(hipMemsetAsync(m.dbgc.p, 0, 4 * sizeof(unsigned long long), m.stream));
    }
    const size_t nblocks = ((m.nruns + 7) / 8) * 8 * G;
    if (nblocks > 0x7fffffffull) {
        if (err) *err = "linesearch_ndcg: grid too large";
        return false;
    }
    const size_t lds = 2 * dp * sizeof(double);
    {
        ProfScope ps("linesearch_ndcg_kernel", m.stream);
        if (depth <= 5) dispatch_ct<5>(a, (unsigned)nblocks, maxc, lds, m.stream);
        else if (depth <= 10) dispatch_ct<10>(a, (unsigned)nblocks, maxc, lds, m.stream);
        else dispatch_ct<20>(a, (unsigned)nblocks, maxc, lds, m.stream);
    }
    FR_HIP(hipGetLastError());
    if (!launch_means(m.M.p, ldm, ldm, m.nq, m.partial, m.means, m.stream, err)) return false;
    FR_HIP(hipMemcpyAsync(means->data(), m.means.p, ldm * sizeof(double), hipMemcpyDeviceToHost, m.stream));
    m.last_ldm = ldm;
    m.last_cols = ldm;
    if (a.debug & 16) {
        unsigned long long c[4];
        FR_HIP(hipMemcpyAsync(c, m.dbgc.p, sizeof(c), hipMemcpyDeviceToHost, m.stream));
        FR_HIP(hipStreamSynchronize(m.stream));
        fprintf(stderr, "[FR_LS_DEBUG] docs=%llu rows=%llu (%.3f of docs) batches=%llu insertion_rows=%llu\n", c[3], c[0],
                (double)c[0] / (double)c[3], c[1], c[2]);
    }
    return m.pull_flags(err);
}

bool DeviceDataset::fullrank_supported(int measure, int64_t depth) const {
    const Impl& m = *impl_;
    (void)depth;
    if (m.nonfinite || m.dq * 4 > 2048) return false;
    if (m.maxlen > 2048) return false;                       // per-lane rank table must fit LDS (64 B per rank)
    if (measure == M_NDCG && m.termtab.cap < m.ncls * m.tablen) return false;  // too many gain classes
    return measure == M_NDCG || measure == M_AP || measure == M_RR;
}

template <int CT>
static void launch_scores(const FSArgs& a, unsigned nblocks, size_t lds, hipStream_t st) {
    linesearch_scores_kernel<CT, 16><<<dim3(nblocks), dim3(WAVE), lds, st>>>(a);
}

// Full-ranking line search: every candidate of every group, any measure.  means[g*64 + c].
bool DeviceDataset::linesearch_fullrank(int measure, int64_t depth, const double* norms,
                                        const std::vector<LineGroup>& groups, std::vector<double>* means,
                                        std::string* err) {
    Impl& m = *impl_;
    std::lock_guard<std::mutex> lk(m.mu);
    if (!m.bind(err)) return false;
    if (!fullrank_supported(measure, depth)) {
        if (err) *err = "linesearch_fullrank: unsupported dataset or measure";
        return false;
    }
    const size_t G = groups.size();
    means->assign(G * 64, 0.0);
    if (G == 0) return true;
    const size_t dp = m.dq * 4;
    const size_t ldm = G * 64;
    size_t maxc = 0;
    for (const auto& lg : groups) {
        if (lg.feature >= m.d || lg.weights.size() != m.d || lg.candidates.empty() || lg.candidates.size() > 64) {
            if (err) *err = "linesearch_fullrank: malformed line group";
            return false;
        }
        maxc = std::max(maxc, lg.candidates.size());
    }
    // chunk the groups so that the score rows stay within ~8 GiB of HBM
    const size_t row_bytes = 64 * sizeof(double);
    size_t GC = std::max<size_t>(1, std::min<size_t>(G, (size_t(8) << 30) / (m.np * row_bytes)));
    if (!m.M.ensure(m.nq * ldm, err) || !m.norms.ensure(m.nq, err) || !m.rows.ensure(m.np * GC * 64, err) ||
        !m.gfeat.ensure(GC, err) || !m.gncand.ensure(GC, err) || !m.gw.ensure(GC * dp, err) ||
        !m.gcand.ensure(GC * 64, err) || !m.means.ensure(ldm, err))
        return false;
    FR_HIP(hipMemcpyAsync(m.norms.p, norms, m.nq * sizeof(double), hipMemcpyHostToDevice, m.stream));
    int dd = depth < 0 ? -1 : (depth > 0x7fffffff ? 0x7fffffff : (int)depth);
    for (size_t g0 = 0; g0 < G; g0 += GC) {
        const size_t gc = std::min(GC, G - g0);
        std::vector<uint32_t> gfeat(gc), gncand(gc);
        std::vector<double> gw(gc * dp, 0.0), gcand(gc * 64, 0.0);
        for (size_t g = 0; g < gc; g++) {
            const LineGroup& lg = groups[g0 + g];
            gfeat[g] = lg.feature;
            gncand[g] = (uint32_t)lg.candidates.size();
            std::memcpy(&gw[g * dp], lg.weights.data(), m.d * sizeof(double));
            std::memcpy(&gcand[g * 64], lg.candidates.data(), lg.candidates.size() * sizeof(double));
        }
        FR_HIP(hipMemcpyAsync(m.gfeat.p, gfeat.data(), gc * sizeof(uint32_t), hipMemcpyHostToDevice, m.stream));
        FR_HIP(hipMemcpyAsync(m.gncand.p, gncand.data(), gc * sizeof(uint32_t), hipMemcpyHostToDevice, m.stream));
        FR_HIP(hipMemcpyAsync(m.gw.p, gw.data(), gc * dp * sizeof(double), hipMemcpyHostToDevice, m.stream));
        FR_HIP(hipMemcpyAsync(m.gcand.p, gcand.data(), gc * 64 * sizeof(double), hipMemcpyHostToDevice, m.stream));
        FR_HIP(hipStreamSynchronize(m.stream));  // the staging vectors are locals
        FSArgs fa;
        fa.xb = (const float4*)m.xb.p;
        fa.run_pos = m.run_pos.p;
        fa.run_docs = m.run_docs.p;
        fa.run_order = m.run_order.p;
        fa.gfeat = m.gfeat.p;
        fa.gw = m.gw.p;
        fa.gcand = m.gcand.p;
        fa.rows = m.rows.p;
        fa.flags = m.flags.p;
        fa.dq = (uint32_t)m.dq;
        fa.d = (uint32_t)m.d;
        fa.nruns = (uint32_t)m.nruns;
        fa.GC = (uint32_t)gc;
        const size_t nblocks = ((m.nruns + 7) / 8) * 8 * gc;
        {
            ProfScope ps("linesearch_scores_kernel", m.stream);
            const size_t lds = 2 * dp * sizeof(double);
            if (maxc <= 16) launch_scores<16>(fa, (unsigned)nblocks, lds, m.stream);
            else if (maxc <= 51) launch_scores<51>(fa, (unsigned)nblocks, lds, m.stream);
            else launch_scores<64>(fa, (unsigned)nblocks, lds, m.stream);
        }
        FR_HIP(hipGetLastError());
        RMArgs ra;
        ra.rows = m.rows.p;
        ra.qstart = m.qstart.p;
        ra.qlen = m.qlen.p;
        ra.qnpos = m.qnpos.p;
        ra.qnneg = m.qnneg.p;
        ra.gcls = m.gcls.p;
        ra.termtab = m.termtab.p;
        ra.norms = m.norms.p;
        ra.gncand = m.gncand.p;
        ra.M = m.M.p;
        ra.flags = m.flags.p;
        ra.GC = (uint32_t)gc;
        ra.ldm = (uint32_t)ldm;
        ra.col0 = (uint32_t)(g0 * 64);
        ra.tablen = (uint32_t)m.tablen;
        ra.measure = measure;
        ra.depth = dd;
        {
            ProfScope ps("rank_metric_kernel", m.stream);
            FR_HIP(hipFuncSetAttribute((const void*)rank_metric_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       128 * 1024));
            // longest queries first (their O(n^2) sweeps are the tail), several waves per long query
            for (size_t ci = m.size_classes.size(); ci-- > 0;) {
                const auto& sc = m.size_classes[ci];
                ra.qlist = m.qlist.p + sc.offset;
                dim3 grid((unsigned)sc.count, (unsigned)gc);
                const unsigned waves = sc.npad <= 128 ? 1u : (sc.npad <= 256 ? 2u : (sc.npad <= 512 ? 4u : 8u));
                rank_metric_kernel<<<grid, WAVE * waves, (size_t)sc.npad * 64, m.stream>>>(ra);
            }
        }
        FR_HIP(hipGetLastError());
    }
    if (!launch_means(m.M.p, ldm, ldm, m.nq, m.partial, m.means, m.stream, err)) return false;
    FR_HIP(hipMemcpyAsync(means->data(), m.means.p, ldm * sizeof(double), hipMemcpyDeviceToHost, m.stream));
    m.last_ldm = ldm;
    m.last_cols = ldm;
    return m.pull_flags(err);
}

}  // namespace frdev
