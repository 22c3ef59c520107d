// Device-side interface of the MI355X hot path (no HIP types leak through this header).
//
// A DeviceDataset is the HBM-resident form of one reference `RankingDataset` view
// (src/dense_dataset.rs:11-20, src/dataset.rs:101-106):
//   * documents regrouped by query (CSR), and inside each query stored in REVERSE tie-break
//     order (gain desc, instance-id desc) so that "later document wins score ties" reproduces
//     the reference's (score desc, gain asc, id asc) total order (src/evaluators.rs:34-49);
//   * consecutive queries are packed into "runs" of whole 64-document tiles; features live in
//     tiles [tile][D/4][64 docs][4 features] f32, so lane = document reads 16 B per load;
//   * per-document gain (f32), 2^gain-1 (f64, host libm like the reference), relevance flag;
//   * per-query offsets, a longest-first query schedule, log2(i+2) discount table.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace frdev {

// Environment switches, three kinds (INTEGRATION.md lists the first):
//   * runtime configuration of the shipped library is read with std::getenv where it is used: FR_DEVICES, FR_RESTART_QUEUE,
//     FR_RESTART_SLOTS, FR_MIN_RESTARTS_PER_DEVICE, FR_XCOL, FR_VERIFY_ORDER, FR_LS_EXACT, FR_VERIFY_AUDIT, FR_LS_PIPELINE,
//     FR_UPLOAD_TIMING, FR_HOST_TIMING;
//   * path_env: selectors of ANOTHER BIT-EXACT path (the exact kernels instead of bound-and-verify, the generic sort
//     evaluator instead of the fused one, tiles instead of resident sums, ...).  The parity tests use them to reach every
//     path with the same inputs; whatever they are set to, results are the reference's;
//   * pricing_env: timing experiments and tuning sweeps, some of which return WRONG numbers on purpose (a kernel phase
//     skipped to price the rest).  They exist only in a library built with -DFR_PRICING (FR_BUILD_FLAGS=-DFR_PRICING):
//     the shipped library cannot be talked into wrong scores through its environment.
const char* path_env(const char* name);
inline const char* pricing_env(const char* name) {
#ifdef FR_PRICING
    return path_env(name);
#else
    (void)name;
    return nullptr;
#endif
}

enum Measure : int { M_NDCG = 0, M_AP = 1, M_RR = 2 };

// Walk tiles of the resident NDCG@k verify kernel (kernels_order.inc).  Every run's positions are cut, greedily and in query
// order, into stretches of at most WALK_TILE positions such that a query of up to WALK_TILE documents is never cut; a longer
// one is cut every WALK_TILE documents from its start and what is left of it shares a tile with the queries behind it.
//   wt_start[i] = first position of tile i, ascending; one more entry = np (behind a run's last tile comes padding)
//   run_wt0[r]  = the tile run r starts in (tiles never span runs)
//   seg[p]      = first slot | (one past the last slot) << 8 of position p's (query, tile) segment, relative to the tile's
//                 start; 0 for positions that hold no document
//   wofs[p]     = p's offset inside its tile
constexpr uint32_t WALK_TILE = 128;
struct WalkTileLayout {
    std::vector<uint32_t> wt_start, run_wt0;
    std::vector<uint16_t> seg;
    std::vector<uint8_t> wofs;
};
inline WalkTileLayout build_walk_tiles(const std::vector<uint32_t>& run_pos, const std::vector<uint32_t>& run_q0, const std::vector<uint32_t>& run_q1,
                                       const std::vector<uint32_t>& qstart, const std::vector<uint32_t>& qlen, size_t np) {
    constexpr uint32_t WT = WALK_TILE;
    WalkTileLayout out;
    std::vector<uint32_t>& wts = out.wt_start;
    const size_t nruns = run_pos.size(), nq = qlen.size();
    out.run_wt0.assign(nruns, 0);
    for (size_t r = 0; r < nruns; r++) {
        out.run_wt0[r] = (uint32_t)wts.size();
        uint32_t start = run_pos[r], len = 0;
        auto close = [&]() {
            if (len == 0) return;
            wts.push_back(start);
            start += len;
            len = 0;
        };
        for (uint32_t q = run_q0[r]; q < run_q1[r]; q++) {
            uint32_t n = qlen[q];
            if (n > WT) {
                close();
                for (; n > WT; n -= WT) {
                    len = WT;
                    close();
                }
                len = n;
            } else {
                if (len + n > WT) close();
                len += n;
            }
        }
        close();
    }
    const size_t nwt = wts.size();
    wts.push_back((uint32_t)np);
    out.seg.assign(np, 0);
    out.wofs.assign(np, 0);
    size_t t = 0;
    for (size_t q = 0; q < nq; q++) {
        const size_t b = qstart[q], e = b + qlen[q];
        while (t + 1 < nwt && wts[t + 1] <= b) t++;
        for (size_t u = t; u < nwt && wts[u] < e; u++) {
            const size_t t0 = wts[u], t1 = std::min<size_t>((size_t)wts[u + 1], t0 + WT);
            const size_t lo = std::max(b, t0) - t0, hi = std::min(e, t1) - t0;
            for (size_t p = t0 + lo; p < t0 + hi; p++) {
                out.seg[p] = (uint16_t)(lo | (hi << 8));
                out.wofs[p] = (uint8_t)(p - t0);
            }
        }
    }
    return out;
}

// Error bound E >= |R - sum_j x_j v_j| of a trainer's resident sums, u = 2^-53, T from column
// maxima.  One place for the constants: the trainer (host.hpp), compute_eps2 (device_dataset.inc) and the CPU test
// that replays the device's update arithmetic against extended precision (tests/test_error_bound.py, through
// fr_debug_resident_bound) all use these.
//   after an exact refresh (score_linear: ordered unfused f64 sums of D products): gamma_D * T
inline double resident_err_refresh(uint32_t d, double T) { return 1.1 * (double)(d + 1) * 0x1p-53 * T; }
//   after R' = fma(x_f, cand, fma(-x_f, base_f, R * fl(1/norm))) with base = fl(v / norm), v' = base with [f] = cand:
//   the old error shrinks by 1/norm; the separately rounded divisions of l1_normalize (u T), the reciprocal and the
//   product R * inv (2 u T), and the two FMA roundings (u (|x_f base_f| + T) + u T) add at most 8 u T with
//   T = |base_f| X_f + sum_j |v'_j| X_j
inline double resident_err_update(double err, double norm, double T) { return 1.002 * err / norm + 8.0 * 0x1p-53 * T * (1.0 + 1e-6); }
//   what A = fma(-x_f, base_f, R * fl(1/norm)) adds to a candidate's key error (T = sum_j |base_j| X_j incl. j = f)
inline double resident_eps_extra(double err, double norm, double T) { return 1.01 * err / norm + 10.0 * 0x1p-53 * T; }

// error bits raised by kernels (the reference panics in these cases)
enum : int {
    FLAG_NAN_SCORE = 1,        // src/model.rs:49  "Model.predict -> NaN"
    FLAG_ACTUAL_GT_IDEAL = 2,  // src/evaluators.rs:368-374
};

struct HostCSR {
    size_t n = 0, d = 0, nq = 0;
    const float* x = nullptr;     // row-major base matrix, stride d
    std::vector<uint32_t> perm;   // CSR position -> row in x (original InstanceId)
    std::vector<uint32_t> qoff;   // [nq+1]
    std::vector<float> gain;      // [n] by CSR position
};

struct FlatTrees {
    // node k: fid<0 => leaf with value `split`; else go lhs if f64(x[fid]) <= split else rhs
    std::vector<int32_t> fid, lhs, rhs;
    std::vector<double> split;
    std::vector<int32_t> root;    // per tree
    std::vector<double> weight;   // per tree (ignored when raw_single)
    bool raw_single = false;      // a bare DecisionTree model: output = leaf value
};

// One line-search group: every candidate shares (feature f, base weights w) and differs only in
// the weight of f (src/coordinate_ascent.rs:131-171).
struct LineGroup {
    uint32_t feature = 0;
    std::vector<double> weights;     // [d] base weights (entry `feature` ignored)
    std::vector<double> candidates;  // <= 64 values for w[feature], in evaluation order
    // Optional (bound-and-verify path only): a resident per-document sum R ~ sum_j x_j * v_j for some
    // vector v (the trainer's un-normalised best weights) lives in slot `resident_slot`; the base dot
    // product is then formed as  A = R / resident_norm - x_f * resident_base_f  instead of from the
    // feature tiles (weights == v / resident_norm up to rounding).  `resident_err` bounds |R - sum_j x_j v_j|
    // for every document.  A pending update (the previous tick accepted a candidate, so v changed in one
    // coordinate after being normalised) is applied first:
    //   R <- fma(x_uf, upd_cand, fma(-x_uf, upd_base_f, R / upd_norm))
    int resident_slot = -1;
    uint64_t resident_owner = 0;  // ticket from resident_reserve(); a stale ticket means "form A from the tiles"
    double resident_norm = 1.0, resident_base_f = 0.0, resident_err = 0.0;
    bool has_update = false;
    uint32_t upd_feature = 0;
    double upd_norm = 1.0, upd_base_f = 0.0, upd_cand = 0.0;
};

struct KernelStat {
    std::string name;
    uint64_t launches = 0;
    double total_ms = 0.0;
};

class DeviceDataset {
  public:
    static std::shared_ptr<DeviceDataset> create(const HostCSR& csr, std::string* err);
    // A view of `parent` restricted to some of its queries (src/sampling.rs:117-133 with_queries): shares the parent's
    // feature tiles and per-document arrays in HBM (no second copy of X) and only builds its own query / run tables.
    // csr = the view's own CSR (same documents per query, same order inside a query as the parent);
    // parent_query[q] = index of the view's query q in the parent.
    static std::shared_ptr<DeviceDataset> create_view(const std::shared_ptr<DeviceDataset>& parent, const HostCSR& csr,
                                                      const std::vector<uint32_t>& parent_query, std::string* err);
    // A copy of `src` (which must own its matrix) in the HBM of device `device`, made device to device (hipMemcpyPeer);
    // the same ordinal as the source gives a second, independent context on that device.
    static std::shared_ptr<DeviceDataset> replicate(const std::shared_ptr<DeviceDataset>& src, int device, std::string* err);
    int device_ordinal() const;
    bool shares_parent_matrix() const;
    ~DeviceDataset();

    size_t n() const;
    size_t d() const;
    size_t nq() const;
    size_t max_query_len() const;
    size_t hbm_bytes() const;

    // --- scoring into the internal score buffer (slot b of B, CSR order) ---------------------
    // weights: B vectors of length d (row-major), exact reference dot product.
    bool score_linear(size_t B, const double* weights, std::string* err);
    bool score_single_feature(uint32_t fid, double dir, std::string* err);
    bool score_trees(const FlatTrees& trees, std::string* err);
    // out += w * tmp (unfused), used for mixed ensembles (src/model.rs:104-112)
    bool ensemble_begin(std::string* err);                 // acc = 0
    bool ensemble_accumulate(double w, std::string* err);  // acc = acc + w * slot0
    bool ensemble_finish(std::string* err);                // slot0 = acc

    // copy slot b back, scattered to original instance ids: out[perm[p]] = score[p]
    bool download_scores(size_t b, double* out_by_instance, size_t out_len, std::string* err);

    // --- metrics -------------------------------------------------------------------------------
    // per-query metric of the B score slots -> M[q][b]; norms[nq]: NDCG ideal (NaN=None) /
    // AP num_relevant (0=absent).  depth<0 = None.
    bool metric_from_scores(int measure, int64_t depth, const double* norms, size_t B,
                            bool want_rank, std::string* err);
    bool download_per_query(size_t B, double* out /*[nq*B], q-major*/, std::string* err);
    bool download_rank(uint32_t* out_instance_ids /*[n] grouped by query, rank order*/, std::string* err);
    // mean over queries of each of the last result's columns (sequential sum in query order)
    bool reduce_means(size_t ncols, double* out_means, std::string* err);
    // query-sharded training: every reduction over queries (reduce_means, the line searches) returns the
    // SUM over this dataset's queries, in the fixed two-level shape, instead of the mean
    void set_sums_only(bool on);

    // --- fused line search (NDCG@k, k <= 20) --------------------------------------------------
    // false for non-NDCG@k measures, k > 20, and datasets with non-finite features
    bool linesearch_supported(int measure, int64_t depth) const;
    // evaluates every candidate of every group; means[g*64 + c]
    bool linesearch_ndcg(int64_t depth, const double* norms, const std::vector<LineGroup>& groups,
                         std::vector<double>* means, std::string* err);
    // The same in parts, for callers that keep several independent sets of groups in flight (ctx 0 ..
    // LINESEARCH_CONTEXTS-1: each has its own stream and buffers, so the host's work between two line searches of
    // one set overlaps the kernels of the others).  submit queues everything and returns; collect waits and
    // returns means[g*64 + c].
    static constexpr int LINESEARCH_CONTEXTS = 4;
    bool linesearch_ndcg_submit(int ctx, int64_t depth, const double* norms, const std::vector<LineGroup>& groups, std::string* err);
    bool linesearch_ndcg_collect(int ctx, std::vector<double>* means, std::string* err);
    // Reciprocal rank on resident sums, same contexts: *queued = false means "not applicable right now" (nothing is
    // pending; pending resident updates were applied) and the caller evaluates the groups with linesearch_fullrank.
    bool linesearch_rr_submit(int ctx, const std::vector<LineGroup>& groups, bool* queued, std::string* err);
    bool linesearch_rr_collect(int ctx, std::vector<double>* means, std::string* err);
    // The same protocol for every full-ranking measure: reciprocal rank as above; NDCG of any depth and AP by sorting
    // approximate keys in registers and verifying the gaps between neighbours of different gain class
    // (kernels_fullverify.inc; works from resident sums or, for stateless callers, from the feature tiles).
    bool linesearch_fullrank_submit(int ctx, int measure, int64_t depth, const double* norms,
                                    const std::vector<LineGroup>& groups, bool* queued, std::string* err);
    bool linesearch_fullrank_collect(int ctx, std::vector<double>* means, std::string* err);
    // resident per-document sums for LineGroup::resident_slot: `slots` double-buffered arrays of np doubles
    // Returns an owner ticket (0 on failure).  A later reserve by someone else takes the buffers over: groups
    // and stores that carry the old ticket are then treated as non-resident / refused.
    uint64_t resident_reserve(size_t slots, std::string* err);
    // slot <- the scores of score slot b of the last score_linear() call (exact ordered sums); false with an
    // empty *err when the ticket is stale
    bool resident_store_from_scores(uint64_t owner, size_t slot, size_t b, std::string* err);
    const std::vector<double>& column_absmax() const;  // per-column max |x|
    // running totals: (run, group) pairs given to the bound-and-verify kernel / recomputed exactly
    void verify_counters(unsigned long long* pairs, unsigned long long* redone) const;
    // line searches evaluated by the exact kernels alone (NDCG@k: every group of the line search was routed there; the other
    // measures: a recent line search had > 25 % of its pairs redone)
    unsigned long long exact_fallbacks() const;
    // NDCG@k: group line searches routed to the exact kernel (a restart whose last verified line search left > 25 % of its
    // pairs undecided goes there for 4, 8, 16 line searches; the other groups of the tick stay on the verify kernel), and
    // the (query, group, 16-candidate slice) entries the verify kernel listed for recomputation
    void routing_counters(unsigned long long* exact_groups, unsigned long long* redo_entries) const;
    // NDCG@k verify kernel: documents that made a wave run its insertion chain, and the (document, group) visits they are out of
    // ... and the restarts whose R ranks were found worth keeping / not worth it (device_dataset.inc: slot_rank_mode)
    void chain_counters(unsigned long long* runs, unsigned long long* visits, unsigned long long* ranked_on = nullptr,
                        unsigned long long* ranked_off = nullptr) const;
    // FR_VERIFY_AUDIT=1: values re-derived by the exact kernel after a bound-and-verify line search / how many differed
    void audit_counters(unsigned long long* values, unsigned long long* mismatches) const;
    // --- full-ranking line search (AP, RR, NDCG of any depth): scores kernel + rank-counting kernel ----
    bool fullrank_supported(int measure, int64_t depth) const;
    bool linesearch_fullrank(int measure, int64_t depth, const double* norms, const std::vector<LineGroup>& groups,
                             std::vector<double>* means, std::string* err);
    // column stride of the last linesearch result (for download_per_query-style inspection)
    size_t last_ldm() const;
    bool download_last_matrix(std::vector<double>* out, size_t* ldm, std::string* err);

    // --- random-forest training (src/random_forest.rs:211-408; kernels_rf.inc) ---------------------------------
    // A batch of trees is grown level by level.  rf_begin: tree t's sampled instances are root_ids[root_off[t] ..
    // root_off[t+1]) (original instance ids, in the order the sample is iterated), its sampled features
    // feats[t*nf .. t*nf+nf); every instance starts in node key t.
    struct RfActive { uint32_t tree, key, n; };
    struct RfCand { double position, importance; uint32_t ids_i, flags, pos_l, pos_r; double sum_l, sum_r; };
    struct RfSplit { int32_t fslot; uint32_t pos, left, right; };
    // File-loaded datasets: bit f of bits_by_instance[id * words + f / 32] = instance id HOLDS feature f (the reference's
    // FeatureStats skips absent values, src/normalizers.rs:24-29, while the sort reads them as 0.0).  nullptr: all held.
    bool rf_set_presence(const uint32_t* bits_by_instance, size_t words, size_t n_instances, std::string* err);
    bool rf_begin(const std::vector<uint32_t>& root_off, const std::vector<uint32_t>& root_ids, uint32_t nf,
                  const std::vector<uint32_t>& feats, std::string* err, const uint32_t* positions = nullptr);
    // host only, callable from another thread while the device works: positions[g] = where instance root_ids[g] sits in the
    // tiled layout (what rf_begin otherwise works out itself); hand the result to rf_begin
    bool rf_positions(const std::vector<uint32_t>& root_ids, uint32_t* positions /*[root_ids.size()]*/, std::string* err);
    // compute_output of every tree's whole sample (random_forest.rs:344-351: a root that does not split)
    bool rf_root_outputs(std::vector<double>* out, std::string* err);
    // one level: for every active node (slot = its index in `active`; slot_of_key maps node keys to slots, IDX for
    // closed nodes) and every feature slot, the k-1 split candidates: cands[(slot*nf + fi)*(k-1) + c-1];
    // label_minmax[slot*2 .. +2)
    bool rf_level(const std::vector<RfActive>& active, const std::vector<uint32_t>& slot_of_key, uint32_t k, int method,
                  uint32_t min_leaf, std::vector<RfCand>* cands, std::vector<float>* label_minmax, std::string* err);
    // the host's decisions for the level just evaluated: instances move to the children's keys; child_out[slot*2 + side]
    // = compute_output of that child (random_forest.rs:32-41)
    bool rf_split(const std::vector<RfSplit>& splits, std::vector<double>* child_out, std::string* err);
    void rf_end();
    // device bytes per (sampled instance x sampled feature) of a batch: two key and two payload arrays (8 + 8 + 4 + 4), the sorted
    // gains and values of this level and the one before (4 x 4), the side byte of the stable partition
    size_t rf_bytes_per_item() const { return 41; }

    int take_flags();  // returns and clears the accumulated kernel error bits

  private:
    DeviceDataset();
    bool try_score_trees_rank(const FlatTrees& trees, std::string* err);
    bool try_score_trees_lds(const FlatTrees& trees, std::string* err);
    bool try_score_trees_lds_shape(const FlatTrees& trees, const void* shape, std::string* err);
    struct Impl;
    Impl* impl_;
};

// process-wide helpers -------------------------------------------------------------------------
int device_count(std::string* err);
bool set_device(int ordinal, std::string* err);
void warm_device(int ordinal);  // pays the runtime's first-stream cost on that device once per process
void profile_enable(bool on);
void profile_reset();
std::vector<KernelStat> profile_stats();
bool device_synchronize(std::string* err);
// one timed device-to-device copy src -> dst as replicate() makes them (device_plumbing.inc)
// The job's one exchange as a single-process RCCL all-gather over `devices` (rccl_exchange.inc): rank i contributes
// blocks[i * block_len .. (i + 1) * block_len), every rank receives all blocks in rank order; *gathered = rank 0's buffer.
// rep->ran = false with a reason when it does not apply (one rank; ranks sharing a GPU); with N distinct GPUs any failure --
// and any gathered buffer that differs from the blocks -- returns false.
struct RcclReport {
    bool ran = false, matches_host_gather = false;
    int ranks = 0;
    std::vector<int> devices;
    size_t block_doubles = 0;
    double init_us = 0.0, first_us = 0.0, us = 0.0;  // ncclCommInitAll; the first all-gather (lazy set-up included); the second
    std::string reason;                               // why it did not run
    std::string library;                              // the librccl.so that ran (next to this library's own HIP runtime)
};
bool rccl_allgather(const std::vector<int>& devices, const double* blocks, size_t block_len, std::vector<double>* gathered, RcclReport* rep,
                    std::string* err, size_t min_ranks = 2);  // (min_ranks = 1: the self-test -- a one-rank communicator really runs)
bool peer_copy_probe(int src, int dst, size_t bytes, int* can_access, int* enabled, double* ms, std::string* err);
size_t device_free_bytes();
// page-locked host memory (nullptr when none is left: the caller falls back to ordinary memory); for staging uploads that
// a helper thread prepares -- a copy from ordinary memory ran at under 1 GB/s on some of the boxes this was measured on
void* pinned_alloc(size_t bytes);
void pinned_free(void* p);
// size class of the full-ranking kernel a query of `len` documents is sorted in: *nl keys per lane, *pl lanes per candidate
// (no device needed; fullverify.hpp FV_CLASSES)
void fullrank_class_of(uint32_t len, uint32_t* nl, uint32_t* pl);  // free HBM on the current device (0 if unknown)

}  // namespace frdev
