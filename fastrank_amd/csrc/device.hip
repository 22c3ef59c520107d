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
#include <map>
#include <mutex>

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

// scores[b*ld + p] = sum_j f64(x[p][j]) * w[b][j], j ascending, unfused (dense_dataset.rs:67-76)
template <int BT>
__global__ __launch_bounds__(256) void score_linear_kernel(const float* __restrict__ xt, uint32_t ld,
                                                           uint32_t n, uint32_t d,
                                                           const double* __restrict__ w, uint32_t B,
                                                           double* __restrict__ scores) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t b0 = blockIdx.y * BT;
    if (p >= n) return;
    double acc[BT];
#pragma unroll
    for (int t = 0; t < BT; t++) acc[t] = 0.0;
    const float* xp = xt + p;
#pragma unroll 4
    for (uint32_t j = 0; j < d; j++) {
        double x = (double)xp[(size_t)j * ld];
#pragma unroll
        for (int t = 0; t < BT; t++) {
            uint32_t b = b0 + t < B ? b0 + t : B - 1;
            double prod = x * w[(size_t)b * d + j];
            acc[t] = acc[t] + prod;
        }
    }
#pragma unroll
    for (int t = 0; t < BT; t++)
        if (b0 + t < B) scores[(size_t)(b0 + t) * ld + p] = acc[t];
}

// SingleFeatureModel (src/model.rs:35-40): dir * f64(x[fid])
__global__ void score_single_feature_kernel(const float* __restrict__ xt, uint32_t ld, uint32_t n,
                                            uint32_t fid, double dir, double* __restrict__ scores) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) scores[p] = dir * (double)xt[(size_t)fid * ld + p];
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
    double split;  // or leaf value
    int32_t fid;   // <0: leaf
    int32_t lhs, rhs;
    int32_t pad;
};

// Batched tree-ensemble scoring (src/model.rs:64-84,104-112; config 5 of BASELINE.json).
// One thread = one document.  With ROWS_IN_LDS the block first stages its documents' feature
// rows from the column-major matrix into LDS (coalesced 256-B column segments in, row stride
// d+1 dwords so lane-strided accesses spread over the banks), then every tree walk reads
// features from LDS; node records come from L1/L2 (all lanes walk the same tree).
template <bool ROWS_IN_LDS>
__global__ __launch_bounds__(128) void tree_ensemble_kernel(const float* __restrict__ xt, uint32_t ld,
                                                            uint32_t n, uint32_t d,
                                                            const TreeNodeDev* __restrict__ nodes,
                                                            const int32_t* __restrict__ roots,
                                                            const double* __restrict__ tw, uint32_t ntrees,
                                                            int raw_single, double* __restrict__ scores) {
    extern __shared__ float rows[];  // [blockDim.x][d+1]
    const uint32_t tid = threadIdx.x;
    const uint32_t p = blockIdx.x * blockDim.x + tid;
    const uint32_t pc = p < n ? p : n - 1;
    const uint32_t rs = d + 1;
    if (ROWS_IN_LDS) {
        for (uint32_t j = 0; j < d; j++) rows[tid * rs + j] = xt[(size_t)j * ld + pc];
        __syncthreads();
    }
    double acc = 0.0;
    for (uint32_t t = 0; t < ntrees; t++) {
        int32_t node = roots[t];
        TreeNodeDev nd = nodes[node];
        while (nd.fid >= 0) {
            float xv;
            if ((uint32_t)nd.fid < d) {
                xv = ROWS_IN_LDS ? rows[tid * rs + (uint32_t)nd.fid] : xt[(size_t)nd.fid * ld + pc];
            } else {
                xv = 0.0f;  // Features::get -> None -> unwrap_or(0.0) (src/model.rs:72-73)
            }
            node = ((double)xv <= nd.split) ? nd.lhs : nd.rhs;
            nd = nodes[node];
        }
        if (raw_single) {
            acc = nd.split;
        } else {
            double prod = tw[t] * nd.split;
            acc = acc + prod;
        }
    }
    if (p < n) scores[p] = acc;
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
    const double* __restrict__ scores, uint32_t ld, const uint32_t* __restrict__ qoff,
    const double* __restrict__ gexp, const float* __restrict__ gain, const double* __restrict__ disc,
    const double* __restrict__ norms, int measure, int depth, uint32_t B, double* __restrict__ M,
    uint32_t* __restrict__ rank_out, const uint32_t* __restrict__ perm, int* __restrict__ flags,
    uint32_t npad_max) {
    extern __shared__ double lds_raw[];
    double* keys = lds_raw;
    uint32_t* idx = (uint32_t*)(keys + npad_max);
    const uint32_t q = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, nt = blockDim.x;
    const uint32_t base = qoff[q], n = qoff[q + 1] - base;
    uint32_t npad = 1;
    while (npad < n) npad <<= 1;
    const double* sc = scores + (size_t)b * ld + base;
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
    if (rank_out != nullptr && b == 0)
        for (uint32_t i = tid; i < n; i += nt) rank_out[base + i] = perm[base + idx[i]];
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
                // src/evaluators.rs:422-447
                uint32_t num_rel = (uint32_t)norms[q];
                if (num_rel == 0)
                    for (uint32_t i = 0; i < n; i++) num_rel += keys[i] != 0.0;
                if (num_rel != 0) {
                    int recall_points = 0;
                    double sum_precision = 0.0;
                    for (uint32_t i = 0; i < n; i++) {
                        if (keys[i] != 0.0) {
                            recall_points += 1;
                            sum_precision += (double)recall_points / (double)(i + 1);
                        }
                    }
                    result = sum_precision / (double)num_rel;
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
    const float* xt;
    const double* gexp;
    const uint32_t* qoff;
    const uint32_t* qorder;
    const double* norms;
    const double* disc;
    const uint32_t* gfeat;  // [G]
    const double* gw;       // [G][d]
    const double* gcand;    // [G][64]
    const uint32_t* gncand; // [G]
    double* M;              // [nq][ldm]
    int* flags;
    uint64_t* dbg_counters;  // [4] rows, batches, insertion rows, documents (debug bit 16)
    uint32_t ld, d, nq, G, ldm;
    int depth;
    int debug;  // tuning knob (env FR_LS_DEBUG): 1 = skip phase K, 2 = no threshold filter, 4 = skip suffix adds, 8 = skip prefix
};

constexpr int LS_ROWPAD = 65;  // LDS row stride in doubles: 130 dwords -> 16 lanes hit 16 distinct even banks
constexpr int LS_JB = 4;       // features per software-pipelined block of phase S
constexpr int LS_PB = 16;      // features per load batch of the shared-prefix chain
constexpr int LS_MAXD = 1016;  // widest matrix the fused kernel stages weights for (LDS)

// The fused line search.  One wave per (query, group); CT = candidate tile (accumulators per
// lane in phase S), K = top-K list length (>= depth), RB = documents per transpose batch.
//
// phase S (lane = document, 64 documents per chunk): exact ordered f64 dot products for all CT
//   candidates at once.  The prefix sum over features < f is shared by every candidate; the
//   products x_j*w_j for j > f are shared too, so a candidate costs one v_add_f64 per feature.
//   Feature columns are read from the column-major matrix (256 B per wave per feature) through
//   a register double buffer so the adds of block b hide the loads of block b+1.
// filter: a document can only matter if it ties/beats the current K-th best score of at least
//   one candidate; thresholds are re-published after every transpose batch.
// phase K (lane = candidate): surviving documents are transposed through LDS in batches of RB
//   rows and inserted, in document order, into each candidate's sorted top-K list (registers).
template <int K, int CT, int RB>
__global__ __launch_bounds__(WAVE) void linesearch_ndcg_kernel(LSArgs a) {
    __shared__ double tr[RB * LS_ROWPAD];
    __shared__ double thr[WAVE];
    __shared__ double wl[LS_MAXD + LS_JB];  // this group's base weights, zero padded
    __shared__ double cwl[WAVE];            // this group's candidate weights for feature f
    const uint32_t lane = threadIdx.x;
    // XCD-aware block -> (query, group): blocks b, b+8, b+16.. run on one XCD (observed dispatch
    // b % 8), so all groups of a query share that XCD's L2 for the query's feature columns.
    const uint32_t blk = blockIdx.x;
    const uint32_t xcd = blk & 7u, seq = blk >> 3;
    const uint32_t g = seq % a.G;
    const uint32_t qi = (seq / a.G) * 8u + xcd;
    if (qi >= a.nq) return;
    const uint32_t q = a.qorder[qi];
    const uint32_t base = a.qoff[q];
    const uint32_t n = a.qoff[q + 1] - base;
    const uint32_t f = a.gfeat[g];
    const uint32_t ncand = a.gncand[g];
    const double* __restrict__ w = a.gw + (size_t)g * a.d;
    const double* __restrict__ cw = a.gcand + (size_t)g * 64;
    const uint32_t d = a.d;
    const size_t ld = a.ld;
    const double NEG_INF = -__builtin_huge_val();

    // stage the weights in LDS: every later use is a broadcast ds_read that the compiler can
    // issue ahead of time with counted waits (no scalar-cache round trip inside the hot loops)
    for (uint32_t j = lane; j < d + LS_JB; j += WAVE) wl[j] = j < d ? w[j] : 0.0;
    thr[lane] = NEG_INF;  // nothing is filtered until a candidate's list is full
    cwl[lane] = cw[lane];
    __syncthreads();

    double slot_s[K];
    uint32_t slot_p[K];
#pragma unroll
    for (int m = 0; m < K; m++) {
        slot_s[m] = NEG_INF;
        slot_p[m] = base;
    }
    bool nan_seen = false;
    uint32_t dbg_rows = 0, dbg_batches = 0, dbg_ins = 0;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    for (uint32_t c0 = 0; c0 < n; c0 += WAVE) {
        // ---------------- phase S: lane = document ----------------
        const uint32_t nchunk = (n - c0) < (uint32_t)WAVE ? (n - c0) : (uint32_t)WAVE;
        const uint32_t pl = base + c0 + (lane < nchunk ? lane : nchunk - 1);
        const float* __restrict__ xp = a.xt + pl;
        double P = 0.0;  // shared prefix: features < f in order (dense_dataset.rs:71-74)
        if (!(a.debug & 8)) {
            // the accumulators are not live yet, so the prefix can keep two 16-feature batches of
            // loads in flight (xa is consumed while xc is on its way)
            float xa[LS_PB], xc[LS_PB];
#pragma unroll
            for (int t = 0; t < LS_PB; t++) {
                uint32_t jj = ((uint32_t)t < f) ? (uint32_t)t : 0;
                xa[t] = xp[(size_t)jj * ld];
            }
            for (uint32_t j = 0; j < f; j += LS_PB) {
#pragma unroll
                for (int t = 0; t < LS_PB; t++) {
                    uint32_t jn = j + LS_PB + t;
                    uint32_t jj = (jn < f) ? jn : 0;
                    xc[t] = xp[(size_t)jj * ld];
                }
#pragma unroll
                for (int t = 0; t < LS_PB; t++) {
                    if (j + t < f) {
                        double prod = (double)xa[t] * wl[j + t];
                        P = P + prod;
                    }
                }
#pragma unroll
                for (int t = 0; t < LS_PB; t++) xa[t] = xc[t];
            }
        }
        const double xf = (double)xp[(size_t)f * ld];
        // first block of the suffix is requested before the candidate initialisation below
        float xb[LS_JB], xn[LS_JB];
        uint32_t j0 = f + 1;
#pragma unroll
        for (int t = 0; t < LS_JB; t++) {
            uint32_t jj = (j0 + t < d) ? j0 + t : d - 1;
            xb[t] = xp[(size_t)jj * ld];
        }
        double sc[CT];
#pragma unroll
        for (int c = 0; c < CT; c++) {
            double prod = xf * cwl[c];
            sc[c] = P + prod;
        }
        for (; j0 < d; j0 += LS_JB) {
            const uint32_t jn = j0 + LS_JB;
#pragma unroll
            for (int t = 0; t < LS_JB; t++) {
                uint32_t jj = (jn + t < d) ? jn + t : d - 1;
                xn[t] = xp[(size_t)jj * ld];
            }
#pragma unroll
            for (int t = 0; t < LS_JB; t++) {
                if (j0 + t < d && !(a.debug & 4)) {
                    double prod = (double)xb[t] * wl[j0 + t];
#pragma unroll
                    for (int c = 0; c < CT; c++) sc[c] = sc[c] + prod;
                }
            }
#pragma unroll
            for (int t = 0; t < LS_JB; t++) xb[t] = xn[t];
        }
        // ---------------- filter + transpose + phase K: lane = candidate ----------------
        uint64_t remaining = (nchunk >= 64u) ? ~0ull : ((1ull << nchunk) - 1ull);
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
            }
            uint64_t batch_mask = __ballot(in_batch);
            const uint32_t nb = __popcll(batch_mask);
            remaining &= ~batch_mask;
            dbg_rows += nb;
            dbg_batches++;
            __syncthreads();
            for (uint32_t r = 0; r < nb; r++) {
                const double e = lane < (uint32_t)CT ? tr[r * LS_ROWPAD + lane] : NEG_INF;
                // document position of the r-th set bit of the batch (rows are in lane order)
                const uint32_t bit = (uint32_t)__ffsll((unsigned long long)batch_mask) - 1u;
                batch_mask &= batch_mask - 1ull;
                const uint32_t ep = base + c0 + bit;
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
                        slot_p[m] = beat[m - 1] ? slot_p[m - 1] : (beat[m] ? ep : slot_p[m]);
                    }
                    slot_s[0] = beat[0] ? e : slot_s[0];
                    slot_p[0] = beat[0] ? ep : slot_p[0];
                }
            }
            // publish each candidate's K-th best score; unused candidate lanes never admit a document
            thr[lane] = (lane < ncand) ? slot_s[K - 1] : __builtin_huge_val();
            __syncthreads();
        }
    }

    if ((a.debug & 16) && lane == 0) {
        atomicAdd((unsigned long long*)a.dbg_counters + 0, (unsigned long long)dbg_rows);
        atomicAdd((unsigned long long*)a.dbg_counters + 1, (unsigned long long)dbg_batches);
        atomicAdd((unsigned long long*)a.dbg_counters + 2, (unsigned long long)dbg_ins);
        atomicAdd((unsigned long long*)a.dbg_counters + 3, (unsigned long long)n);
    }
    if (lane < ncand) {
        // src/evaluators.rs:255-272,350-380
        const uint32_t L = (uint32_t)a.depth < n ? (uint32_t)a.depth : n;
        double dcg = 0.0;
#pragma unroll
        for (int i = 0; i < K; i++) {
            if ((uint32_t)i < L) {
                double term = a.gexp[slot_p[i]] / a.disc[i];
                dcg = dcg + term;
            }
        }
        const double norm = a.norms[q];
        double val = 0.0;
        int fl = nan_seen ? FLAG_NAN_SCORE : 0;
        if (norm == norm) {
            if (dcg > norm) fl |= FLAG_ACTUAL_GT_IDEAL;
            val = dcg / norm;
        }
        if (fl && !a.debug) atomicOr(a.flags, fl);
        a.M[(size_t)q * a.ldm + (size_t)g * 64 + lane] = val;
    }
}

// ----------------------------------------------------------------------------------------------
// DeviceDataset
// ----------------------------------------------------------------------------------------------

struct DeviceDataset::Impl {
    int device = 0;
    hipStream_t stream = nullptr;
    size_t n = 0, d = 0, nq = 0, ld = 0, maxlen = 0;
    std::vector<uint32_t> perm_host;
    DevBuf<float> xt, gain;
    DevBuf<double> gexp, disc;
    DevBuf<uint32_t> qoff, qorder, perm, rank;
    DevBuf<int> flags;
    DevBuf<uint64_t> dbgc;
    // work buffers
    DevBuf<double> scores, acc, weights, M, means, partial, norms, gw, gcand;
    DevBuf<uint32_t> gfeat, gncand;
    DevBuf<TreeNodeDev> nodes;
    DevBuf<int32_t> roots;
    DevBuf<double> tweights;
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
    return m.xt.bytes() + m.gain.bytes() + m.gexp.bytes() + m.disc.bytes() + m.qoff.bytes() +
           m.qorder.bytes() + m.perm.bytes();
}

int DeviceDataset::take_flags() {
    std::lock_guard<std::mutex> lk(impl_->mu);
    int v = impl_->host_flags;
    impl_->host_flags = 0;
    return v;
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
    if (csr.n >= 0xFFFFFF00ull) return fail("dataset too large for 32-bit instance ids");
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
    m.ld = (csr.n + 63) / 64 * 64;
    m.perm_host = csr.perm;
    for (size_t q = 0; q < csr.nq; q++) m.maxlen = std::max<size_t>(m.maxlen, csr.qoff[q + 1] - csr.qoff[q]);

    if (!m.xt.ensure(m.d * m.ld, err) || !m.gain.ensure(m.n, err) || !m.gexp.ensure(m.n, err) ||
        !m.qoff.ensure(m.nq + 1, err) || !m.qorder.ensure(m.nq, err) || !m.perm.ensure(m.n, err) ||
        !m.flags.ensure(1, err))
        return nullptr;
    // Host-side transpose in column panels, uploaded panel by panel (one-time cost; SURVEY 8d
    // excludes it from evals/s, bench.py reports it separately).
    {
        const size_t PANEL = 8;
        std::vector<float> panel(PANEL * m.ld, 0.0f);
        for (size_t j0 = 0; j0 < m.d; j0 += PANEL) {
            size_t jn = std::min(PANEL, m.d - j0);
            for (size_t p = 0; p < m.n; p++) {
                const float* row = csr.x + (size_t)csr.perm[p] * csr.d + j0;
                for (size_t jj = 0; jj < jn; jj++) panel[jj * m.ld + p] = row[jj];
            }
            if (!chk(hipMemcpy(m.xt.p + j0 * m.ld, panel.data(), jn * m.ld * sizeof(float),
                               hipMemcpyHostToDevice),
                     "upload X panel"))
                return nullptr;
        }
    }
    {
        std::vector<double> gexp(m.n);
        // (2^g - 1) with the platform libm, exactly like 2.0_f64.powf(gain) - 1.0
        // (src/evaluators.rs:266-270); g is the f32 gain widened to f64.
        for (size_t p = 0; p < m.n; p++) gexp[p] = std::pow(2.0, (double)csr.gain[p]) - 1.0;
        if (!chk(hipMemcpy(m.gexp.p, gexp.data(), m.n * sizeof(double), hipMemcpyHostToDevice), "upload gexp"))
            return nullptr;
        if (!chk(hipMemcpy(m.gain.p, csr.gain.data(), m.n * sizeof(float), hipMemcpyHostToDevice), "upload gain"))
            return nullptr;
        size_t nd = std::max<size_t>(m.maxlen, 64);
        std::vector<double> disc(nd);
        for (size_t i = 0; i < nd; i++) disc[i] = std::log2((double)i + 2.0);
        if (!m.disc.ensure(nd, err)) return nullptr;
        if (!chk(hipMemcpy(m.disc.p, disc.data(), nd * sizeof(double), hipMemcpyHostToDevice), "upload disc"))
            return nullptr;
        if (!chk(hipMemcpy(m.qoff.p, csr.qoff.data(), (m.nq + 1) * sizeof(uint32_t), hipMemcpyHostToDevice),
                 "upload qoff"))
            return nullptr;
        if (!chk(hipMemcpy(m.perm.p, csr.perm.data(), m.n * sizeof(uint32_t), hipMemcpyHostToDevice),
                 "upload perm"))
            return nullptr;
        // longest-first schedule so the 1k-document queries do not form the tail of a launch
        std::vector<uint32_t> order(m.nq);
        for (size_t q = 0; q < m.nq; q++) order[q] = (uint32_t)q;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
            return (csr.qoff[x + 1] - csr.qoff[x]) > (csr.qoff[y + 1] - csr.qoff[y]);
        });
        if (!chk(hipMemcpy(m.qorder.p, order.data(), m.nq * sizeof(uint32_t), hipMemcpyHostToDevice),
                 "upload qorder"))
            return nullptr;
        if (!chk(hipMemset(m.flags.p, 0, sizeof(int)), "clear flags")) return nullptr;
    }
    return ds;
}

static bool launch_means(DeviceDataset* self, const double* M, size_t ldm, size_t ncols, size_t nq,
                         DevBuf<double>& partial, DevBuf<double>& means, hipStream_t st, std::string* err) {
    (void)self;
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
    if (!m.scores.ensure(B * m.ld, err) || !m.weights.ensure(B * m.d, err)) return false;
    m.scores_slots = B;
    FR_HIP(hipMemcpyAsync(m.weights.p, weights, B * m.d * sizeof(double), hipMemcpyHostToDevice, m.stream));
    FR_HIP(hipStreamSynchronize(m.stream));  // `weights` is caller memory
    {
        ProfScope ps("score_linear_kernel", m.stream);
        if (B >= 8) {
            dim3 grid((unsigned)((m.n + 255) / 256), (unsigned)((B + 7) / 8));
            score_linear_kernel<8><<<grid, 256, 0, m.stream>>>(m.xt.p, (uint32_t)m.ld, (uint32_t)m.n,
                                                               (uint32_t)m.d, m.weights.p, (uint32_t)B, m.scores.p);
        } else {
            dim3 grid((unsigned)((m.n + 255) / 256), (unsigned)B);
            score_linear_kernel<1><<<grid, 256, 0, m.stream>>>(m.xt.p, (uint32_t)m.ld, (uint32_t)m.n,
                                                               (uint32_t)m.d, m.weights.p, (uint32_t)B, m.scores.p);
        }
    }
    FR_HIP(hipGetLastError());
    return true;
}

bool DeviceDataset::score_single_feature(uint32_t fid, double dir, std::string* err) {
    Impl& m = *impl_;
    std::lock_guard<std::mutex> lk(m.mu);
    if (!m.bind(err)) return false;
    if (!m.scores.ensure(m.ld, err)) return false;
    m.scores_slots = 1;
    if (fid >= m.d) {
        // Features::get -> None -> unwrap_or(0.0) for loaded data; dir * 0.0
        fill_kernel<<<grid1d(m.n, 256), 256, 0, m.stream>>>(m.scores.p, (uint32_t)m.n, dir * 0.0);
    } else {
        score_single_feature_kernel<<<grid1d(m.n, 256), 256, 0, m.stream>>>(m.xt.p, (uint32_t)m.ld, (uint32_t)m.n,
                                                                            fid, dir, m.scores.p);
    }
    FR_HIP(hipGetLastError());
    return true;
}

bool DeviceDataset::score_trees(const FlatTrees& t, std::string* err) {
    Impl& m = *impl_;
    std::lock_guard<std::mutex> lk(m.mu);
    if (!m.bind(err)) return false;
    if (!m.scores.ensure(m.ld, err)) return false;
    m.scores_slots = 1;
    size_t nn = t.fid.size(), nt = t.root.size();
    std::vector<TreeNodeDev> nodes(nn);
    for (size_t k = 0; k < nn; k++) {
        nodes[k].split = t.split[k];
        nodes[k].fid = t.fid[k];
        nodes[k].lhs = t.lhs[k];
        nodes[k].rhs = t.rhs[k];
        nodes[k].pad = 0;
    }
    if (!m.nodes.ensure(std::max<size_t>(nn, 1), err) || !m.roots.ensure(std::max<size_t>(nt, 1), err) ||
        !m.tweights.ensure(std::max<size_t>(nt, 1), err))
        return false;
    FR_HIP(hipMemcpyAsync(m.nodes.p, nodes.data(), nn * sizeof(TreeNodeDev), hipMemcpyHostToDevice, m.stream));
    FR_HIP(hipMemcpyAsync(m.roots.p, t.root.data(), nt * sizeof(int32_t), hipMemcpyHostToDevice, m.stream));
    std::vector<double> tw = t.weight;
    tw.resize(nt, 1.0);
    FR_HIP(hipMemcpyAsync(m.tweights.p, tw.data(), nt * sizeof(double), hipMemcpyHostToDevice, m.stream));
    FR_HIP(hipStreamSynchronize(m.stream));
    const unsigned bs = 128;
    size_t lds = (size_t)bs * (m.d + 1) * sizeof(float);
    {
        ProfScope ps("tree_ensemble_kernel", m.stream);
        if (lds <= 150 * 1024) {
            FR_HIP(hipFuncSetAttribute((const void*)tree_ensemble_kernel<true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            tree_ensemble_kernel<true><<<grid1d(m.n, bs), bs, lds, m.stream>>>(
                m.xt.p, (uint32_t)m.ld, (uint32_t)m.n, (uint32_t)m.d, m.nodes.p, m.roots.p, m.tweights.p,
                (uint32_t)nt, t.raw_single ? 1 : 0, m.scores.p);
        } else {
            tree_ensemble_kernel<false><<<grid1d(m.n, bs), bs, 0, m.stream>>>(
                m.xt.p, (uint32_t)m.ld, (uint32_t)m.n, (uint32_t)m.d, m.nodes.p, m.roots.p, m.tweights.p,
                (uint32_t)nt, t.raw_single ? 1 : 0, m.scores.p);
        }
    }
    FR_HIP(hipGetLastError());
    return true;
}

bool DeviceDataset::ensemble_begin(std::string* err) {
    Impl& m = *impl_;
    std::lock_guard<std::mutex> lk(m.mu);
    if (!m.bind(err)) return false;
    if (!m.acc.ensure(m.ld, err)) return false;
    fill_kernel<<<grid1d(m.n, 256), 256, 0, m.stream>>>(m.acc.p, (uint32_t)m.n, 0.0);
    FR_HIP(hipGetLastError());
    return true;
}

bool DeviceDataset::ensemble_accumulate(double w, std::string* err) {
    Impl& m = *impl_;
    std::lock_guard<std::mutex> lk(m.mu);
    if (!m.bind(err)) return false;
    axpy_unfused_kernel<<<grid1d(m.n, 256), 256, 0, m.stream>>>(m.acc.p, m.scores.p, (uint32_t)m.n, w);
    FR_HIP(hipGetLastError());
    return true;
}

bool DeviceDataset::ensemble_finish(std::string* err) {
    Impl& m = *impl_;
    std::lock_guard<std::mutex> lk(m.mu);
    if (!m.bind(err)) return false;
    if (!m.scores.ensure(m.ld, err)) return false;
    FR_HIP(hipMemcpyAsync(m.scores.p, m.acc.p, m.n * sizeof(double), hipMemcpyDeviceToDevice, m.stream));
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
    std::vector<double> tmp(m.n);
    FR_HIP(hipMemcpyAsync(tmp.data(), m.scores.p + b * m.ld, m.n * sizeof(double), hipMemcpyDeviceToHost, m.stream));
    FR_HIP(hipStreamSynchronize(m.stream));
    for (size_t p = 0; p < m.n; p++) {
        size_t id = m.perm_host[p];
        if (id < out_len) out[id] = tmp[p];
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
        dim3 grid((unsigned)m.nq, (unsigned)B);
        unsigned bs = npad >= 512 ? 256 : (npad >= 128 ? 64 : 64);
        metric_sort_kernel<<<grid, bs, lds, m.stream>>>(m.scores.p, (uint32_t)m.ld, m.qoff.p, m.gexp.p, m.gain.p,
                                                        m.disc.p, m.norms.p, measure, dd, (uint32_t)B, m.M.p,
                                                        want_rank ? m.rank.p : nullptr, m.perm.p, m.flags.p,
                                                        (uint32_t)npad);
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
    if (!launch_means(this, m.M.p, m.last_ldm, ncols, m.nq, m.partial, m.means, m.stream, err)) return false;
    FR_HIP(hipMemcpyAsync(out, m.means.p, ncols * sizeof(double), hipMemcpyDeviceToHost, m.stream));
    FR_HIP(hipStreamSynchronize(m.stream));
    return true;
}

bool DeviceDataset::linesearch_supported(int measure, int64_t depth) {
    return measure == M_NDCG && depth >= 0 && depth <= 20;
}

size_t DeviceDataset::linesearch_max_features() { return LS_MAXD; }

template <int K, int CT>
static void launch_linesearch(const LSArgs& a, unsigned nblocks, hipStream_t st) {
    linesearch_ndcg_kernel<K, CT, 16><<<dim3(nblocks), dim3(WAVE), 0, st>>>(a);
}

template <int K>
static void dispatch_ct(const LSArgs& a, unsigned nblocks, size_t maxc, hipStream_t st) {
    if (maxc <= 4) launch_linesearch<K, 4>(a, nblocks, st);
    else if (maxc <= 16) launch_linesearch<K, 16>(a, nblocks, st);
    else if (maxc <= 32) launch_linesearch<K, 32>(a, nblocks, st);
    else if (maxc <= 52) launch_linesearch<K, 52>(a, nblocks, st);
    else launch_linesearch<K, 64>(a, nblocks, st);
}

bool DeviceDataset::linesearch_ndcg(int64_t depth, const double* norms, const std::vector<LineGroup>& groups,
                                    std::vector<double>* means, std::string* err) {
    Impl& m = *impl_;
    std::lock_guard<std::mutex> lk(m.mu);
    if (!m.bind(err)) return false;
    if (!linesearch_supported(M_NDCG, depth) || m.d > (size_t)LS_MAXD) {
        if (err) *err = "linesearch_ndcg: unsupported depth or feature count";
        return false;
    }
    const size_t G = groups.size();
    means->assign(G * 64, 0.0);
    if (G == 0) return true;
    size_t maxc = 0;
    std::vector<uint32_t> gfeat(G), gncand(G);
    std::vector<double> gw(G * m.d), gcand(G * 64, 0.0);
    for (size_t g = 0; g < G; g++) {
        const LineGroup& lg = groups[g];
        if (lg.feature >= m.d || lg.weights.size() != m.d || lg.candidates.empty() || lg.candidates.size() > 64) {
            if (err) *err = "linesearch_ndcg: malformed line group";
            return false;
        }
        gfeat[g] = lg.feature;
        gncand[g] = (uint32_t)lg.candidates.size();
        maxc = std::max(maxc, lg.candidates.size());
        std::memcpy(&gw[g * m.d], lg.weights.data(), m.d * sizeof(double));
        std::memcpy(&gcand[g * 64], lg.candidates.data(), lg.candidates.size() * sizeof(double));
    }
    const size_t ldm = G * 64;
    if (!m.M.ensure(m.nq * ldm, err) || !m.norms.ensure(m.nq, err) || !m.gfeat.ensure(G, err) ||
        !m.gncand.ensure(G, err) || !m.gw.ensure(G * m.d, err) || !m.gcand.ensure(G * 64, err) ||
        !m.means.ensure(ldm, err))
        return false;
    FR_HIP(hipMemcpyAsync(m.norms.p, norms, m.nq * sizeof(double), hipMemcpyHostToDevice, m.stream));
    FR_HIP(hipMemcpyAsync(m.gfeat.p, gfeat.data(), G * sizeof(uint32_t), hipMemcpyHostToDevice, m.stream));
    FR_HIP(hipMemcpyAsync(m.gncand.p, gncand.data(), G * sizeof(uint32_t), hipMemcpyHostToDevice, m.stream));
    FR_HIP(hipMemcpyAsync(m.gw.p, gw.data(), G * m.d * sizeof(double), hipMemcpyHostToDevice, m.stream));
    FR_HIP(hipMemcpyAsync(m.gcand.p, gcand.data(), G * 64 * sizeof(double), hipMemcpyHostToDevice, m.stream));
    LSArgs a;
    a.xt = m.xt.p;
    a.gexp = m.gexp.p;
    a.qoff = m.qoff.p;
    a.qorder = m.qorder.p;
    a.norms = m.norms.p;
    a.disc = m.disc.p;
    a.gfeat = m.gfeat.p;
    a.gw = m.gw.p;
    a.gcand = m.gcand.p;
    a.gncand = m.gncand.p;
    a.M = m.M.p;
    a.flags = m.flags.p;
    if (!m.dbgc.ensure(4, err)) return false;
    a.dbg_counters = m.dbgc.p;
    a.ld = (uint32_t)m.ld;
    a.d = (uint32_t)m.d;
    a.nq = (uint32_t)m.nq;
    a.G = (uint32_t)G;
    a.ldm = (uint32_t)ldm;
    a.depth = (int)depth;
    {
        const char* dbg = getenv("FR_LS_DEBUG");
        a.debug = dbg ? atoi(dbg) : 0;
        if (a.debug & 16) FR_HIP(hipMemsetAsync(m.dbgc.p, 0, 4 * sizeof(uint64_t), m.stream));
    }
    const size_t nblocks = ((m.nq + 7) / 8) * 8 * G;
    if (nblocks > 0x7fffffffull) {
        if (err) *err = "linesearch_ndcg: grid too large";
        return false;
    }
    {
        ProfScope ps("linesearch_ndcg_kernel", m.stream);
        if (depth <= 5) dispatch_ct<5>(a, (unsigned)nblocks, maxc, m.stream);
        else if (depth <= 10) dispatch_ct<10>(a, (unsigned)nblocks, maxc, m.stream);
        else dispatch_ct<20>(a, (unsigned)nblocks, maxc, m.stream);
    }
    FR_HIP(hipGetLastError());
    if (!launch_means(this, m.M.p, ldm, ldm, m.nq, m.partial, m.means, m.stream, err)) return false;
    FR_HIP(hipMemcpyAsync(means->data(), m.means.p, ldm * sizeof(double), hipMemcpyDeviceToHost, m.stream));
    m.last_ldm = ldm;
    m.last_cols = ldm;
    if (a.debug & 16) {
        uint64_t c[4];
        FR_HIP(hipMemcpyAsync(c, m.dbgc.p, sizeof(c), hipMemcpyDeviceToHost, m.stream));
        FR_HIP(hipStreamSynchronize(m.stream));
        fprintf(stderr, "[FR_LS_DEBUG] docs=%llu rows=%llu (%.3f of docs) batches=%llu insertion_rows=%llu\n",
                (unsigned long long)c[3], (unsigned long long)c[0], (double)c[0] / (double)c[3], (unsigned long long)c[1],
                (unsigned long long)c[2]);
    }
    return m.pull_flags(err);
}

}  // namespace frdev
