// Random-forest training on the MI355X path (src/random_forest.rs:127-408, src/sampling.rs:38-66).
//
// The reference grows every tree recursively on the CPU (rayon over trees): per node and sampled feature it sorts
// the node's instances by feature value, places k-1 thresholds on the value range and scores each split by a sum over
// both sides.  Here a batch of trees is grown LEVEL BY LEVEL on the device (kernels_rf.inc): one radix sort and one
// candidate-evaluation launch per level cover every open node of every tree of the batch; this file keeps the
// reference's sequential decisions (sampling with Rand64, which candidate wins, when a node becomes a leaf) on the host.
// Where the reference leaves an order unspecified (HashMap iteration, sort_unstable among equal keys) the order is the
// one the test oracle fixes, so trees are bit-identical to the oracle's:
//   * a tree's sampled instances are iterated query by query in the dataset's query order, instance ids ascending;
//   * equal feature values sort by that index; equal importances: the last candidate wins.
#pragma once
#include <chrono>
#include <cmath>
#include <exception>
#include <future>
#include <thread>

#include "host.hpp"

namespace fr {

struct RFParams {  // src/random_forest.rs:127-157
    uint64_t seed = 0;
    bool quiet = false;
    uint32_t num_trees = 100;
    bool weight_trees = false;
    int split_method = 0;  // 0 SquaredError, 1 BinaryGiniImpurity, 2 InformationGain, 3 TrueVarianceReduction (:14-20)
    double instance_sampling_rate = 0.5;
    double feature_sampling_rate = 0.25;
    uint32_t min_leaf_support = 10;
    uint32_t split_candidates = 3;
    uint32_t max_depth = 8;

    static const char* method_name(int m) {
        static const char* names[] = {"SquaredError", "BinaryGiniImpurity", "InformationGain", "TrueVarianceReduction"};
        return names[m & 3];
    }
    static RFParams from_json(const Value& v) {
        RFParams p;
        p.seed = json_u64(json_field(v, "seed"), "seed");
        p.quiet = json_bool(json_field(v, "quiet"), "quiet");
        p.num_trees = json_u32(json_field(v, "num_trees"), "num_trees");
        p.weight_trees = json_bool(json_field(v, "weight_trees"), "weight_trees");
        {
            // serde: a zero-field TUPLE variant {"SquaredError": []}; a bare string is what a unit variant would be
            const Value& sm = json_field(v, "split_method");
            std::string name;
            if (sm.is_string()) {
                fail_raw("Error(\"invalid type: unit variant, expected tuple variant\", line: 1, column: 1)");
            } else {
                const auto& var = json_variant(sm, "SplitSelectionStrategy");
                name = var.first;
                if (!var.second.is_array() || !var.second.arr.empty())
                    fail_raw("Error(\"invalid length " + std::to_string(var.second.is_array() ? var.second.arr.size() : 1) +
                             ", expected tuple variant SplitSelectionStrategy::" + name + " with 0 elements\", line: 1, column: 1)");
            }
            int m = -1;
            for (int i = 0; i < 4; i++)
                if (name == method_name(i)) m = i;
            if (m < 0)
                fail_raw("Error(\"unknown variant `" + name +
                         "`, expected one of `SquaredError`, `BinaryGiniImpurity`, `InformationGain`, `TrueVarianceReduction`\", line: 1, column: 1)");
            p.split_method = m;
        }
        p.instance_sampling_rate = json_f64(json_field(v, "instance_sampling_rate"), "instance_sampling_rate");
        p.feature_sampling_rate = json_f64(json_field(v, "feature_sampling_rate"), "feature_sampling_rate");
        p.min_leaf_support = json_u32(json_field(v, "min_leaf_support"), "min_leaf_support");
        p.split_candidates = json_u32(json_field(v, "split_candidates"), "split_candidates");
        p.max_depth = json_u32(json_field(v, "max_depth"), "max_depth");
        return p;
    }
};

struct RFStats {
    double t_sample = 0, t_begin = 0, t_level = 0, t_select = 0, t_split = 0, t_weights = 0;  // host wall seconds per stage (FR_RF_TIMING)
    uint32_t trees = 0, batches = 0, levels = 0, devices = 1;
    uint64_t nodes = 0, candidates = 0, sorted_items = 0;
    double seconds = 0.0;
};

// random_forest.rs:52-87 (host libm log2, as in the reference and the oracle)
inline double rf_gini(uint32_t n, uint32_t positive) {
    if (n == 0) return 0.0;
    const double count = (double)n, pos = (double)positive;
    const double p_yes = pos / count, p_no = (count - pos) / count;
    return p_yes * (1.0 - p_yes) + p_no * (1.0 - p_no);
}
inline double rf_plogp(double x) { return x == 0.0 ? 0.0 : x * std::log2(x); }
inline double rf_entropy(uint32_t n, uint32_t positive) {
    if (n == 0) return 0.0;
    const double count = (double)n, pos = (double)positive;
    const double p_yes = pos / count, p_no = (count - pos) / count;
    return -rf_plogp(p_yes) - rf_plogp(p_no);
}

class RFTrainer {
  public:
    RFTrainer(std::shared_ptr<DatasetView> view, Evaluator ev, RFParams p) : view_(std::move(view)), ev_(std::move(ev)), p_(p) {}

    // One device-side copy of the view and the trees it grows (train_model spreads a forest's trees over the GPUs of a
    // node like a coordinate-ascent request's restarts; the reference grows them with rayon over the host's cores,
    // random_forest.rs:301-331).  slot / device: DatasetView::device_ptr.
    struct DevicePart {
        int slot = 0, device = -1;
        uint32_t t_begin = 0, t_end = 0;
    };

    // random_forest.rs:288-342 learn_ensemble
    Model learn(std::vector<DevicePart> parts = {}) {
        const frdev::HostCSR& csr = view_->host_csr();
        const DataCore& core = *view_->core;
        if (parts.empty()) parts.push_back(DevicePart{0, -1, 0u, p_.num_trees});
        if (!p_.quiet && parts.size() > 1) {  // (the progress table is printed in tree order: one device)
            parts.resize(1);
            parts[0].t_begin = 0, parts[0].t_end = p_.num_trees;
        }
        // sampling.rs:40-48: features ascending, query-id STRINGS ascending
        features_ = view_->features;
        std::sort(features_.begin(), features_.end());
        queries_.resize(csr.nq);  // CSR query indices, ordered by their qid string
        for (size_t q = 0; q < csr.nq; q++) queries_[q] = (uint32_t)q;
        std::sort(queries_.begin(), queries_.end(), [&](uint32_t a, uint32_t b) {
            return core.qnames[view_->csr_query[a]] < core.qnames[view_->csr_query[b]];
        });
        // sampling.rs:49-50: max(1, (len as f64 * rate) as usize), then take(n) from a list of len items.  Rust's cast
        // saturates (NaN / negative -> 0); a C++ cast of such a value is undefined behaviour
        auto sample_count = [](size_t len, double rate) {
            const double x = (double)len * rate;
            const size_t c = (x != x || x <= 0.0) ? 0 : (x >= (double)len ? len : (size_t)x);
            return std::min(len, std::max<size_t>(1, c));
        };
        n_features_ = sample_count(features_.size(), p_.feature_sampling_rate);
        n_queries_ = sample_count(queries_.size(), p_.instance_sampling_rate);
        if (features_.empty()) fail_str("assertion failed: !features.is_empty()");
        if (queries_.empty()) fail_str("assertion failed: !data.queries().is_empty()");
        // instance ids of each query in ascending order (the device layout inside a query is the ranking order)
        qids_sorted_.assign(csr.nq, std::vector<uint32_t>());
        for (size_t qi = 0; qi < csr.nq; qi++) {
            std::vector<uint32_t>& ids = qids_sorted_[qi];
            ids.assign(csr.perm.begin() + csr.qoff[qi], csr.perm.begin() + csr.qoff[qi + 1]);
            std::sort(ids.begin(), ids.end());
        }
        Rand64 rand(p_.seed);
        seeds_.resize(p_.num_trees);
        for (uint32_t t = 0; t < p_.num_trees; t++) seeds_[t] = rand.rand_u64();

        Model out;
        out.kind = Model::Ensemble;
        out.members.resize(p_.num_trees);
        out.ens_weights.assign(p_.num_trees, 1.0);
        if (!p_.quiet) {
            printf("-----------------------\n|%7s|%7s|%7s|\n-----------------------\n", "Tree", "Depth", ev_.name.c_str());
        }
        if (parts.size() == 1) {
            learn_part(parts[0], out, stats_);
        } else {
            std::vector<RFStats> st(parts.size());
            std::vector<std::exception_ptr> errors(parts.size());
            auto work = [&](size_t i) {
                try {
                    learn_part(parts[i], out, st[i]);  // (disjoint slots of out.members / out.ens_weights)
                } catch (...) {
                    errors[i] = std::current_exception();
                }
            };
            std::vector<std::thread> pool;
            for (size_t i = 1; i < parts.size(); i++) pool.emplace_back(work, i);
            work(0);
            for (auto& th : pool) th.join();
            for (auto& e : errors)
                if (e) std::rethrow_exception(e);
            for (const RFStats& x : st) {
                stats_.t_sample += x.t_sample, stats_.t_begin += x.t_begin, stats_.t_level += x.t_level, stats_.t_select += x.t_select;
                stats_.t_split += x.t_split, stats_.t_weights += x.t_weights;
                stats_.batches += x.batches, stats_.levels += x.levels, stats_.nodes += x.nodes, stats_.candidates += x.candidates;
                stats_.sorted_items += x.sorted_items;
            }
        }
        if (frdev::pricing_env("FR_RF_TIMING"))
            fprintf(stderr, "[rf] sample %.2f s, begin %.2f s, levels %.2f s, select %.2f s, split %.2f s, weights %.2f s\n", stats_.t_sample,
                    stats_.t_begin, stats_.t_level, stats_.t_select, stats_.t_split, stats_.t_weights);
        if (!p_.quiet) printf("-----------------------\n");
        stats_.trees = p_.num_trees;
        stats_.devices = (uint32_t)parts.size();
        return out;
    }

    // trees [t_begin, t_end) on one device-side copy of the view
    void learn_part(const DevicePart& part, Model& out, RFStats& stats) {
        if (part.device >= 0) {  // (this host thread's current device)
            std::string derr;
            if (!frdev::set_device(part.device, &derr)) fail_str(derr);
        }
        frdev::DeviceDataset& dev = view_->device(part.slot, part.device);
        const frdev::HostCSR& csr = view_->host_csr();
        const DataCore& core = *view_->core;
        {
            std::string perr;
            if (!dev.rf_set_presence(core.present_bits.empty() ? nullptr : core.present_bits.data(), core.present_words, core.n, &perr)) fail_str(perr);
        }
        // batches sized by device memory: rf_bytes_per_item per (sampled instance x sampled feature), at most 30 GB and at
        // most 40 % of what is free right now (the sort's scratch, the candidate tables and the per-tree score buffers
        // come on top)
        size_t budget = (size_t)30 << 30;
        {
            const size_t free_b = frdev::device_free_bytes();
            if (free_b != 0) budget = std::min(budget, std::max<size_t>((size_t)64 << 20, free_b / 5 * 2));
        }
        if (const char* e = frdev::path_env("FR_RF_BATCH_BYTES")) budget = std::max<size_t>(1 << 20, (size_t)atoll(e));
        struct EndGuard {  // the batch buffers (GBs) go back to the device also when a batch fails
            frdev::DeviceDataset& d;
            ~EndGuard() { d.rf_end(); }
        } end_guard{dev};
        // A batch's host side -- the trees' samples (sampling.rs:38-60) and where their instances sit on the device -- is made
        // by another thread while the device grows the batch before it: 0.12 s per batch of eleven trees at the 30K shape,
        // a quarter of the wall time when it ran in line.
        struct Batch {
            uint32_t t0 = 0, t1 = 0;
            std::vector<uint32_t> root_off, root_ids, feats;
            uint32_t* positions = nullptr;  // in one of the two page-locked slabs below (or in positions_own)
            std::vector<uint32_t> positions_own;
            double seconds = 0.0;
        };
        // two page-locked slabs for the batches' instance positions (the one big upload of a batch), used in turn: one is on
        // the wire while the helper thread fills the other
        struct Slab {
            uint32_t* p = nullptr;
            size_t cap = 0;
            ~Slab() { frdev::pinned_free(p); }
            uint32_t* get(size_t count) {
                if (count > cap) {
                    frdev::pinned_free(p);
                    p = static_cast<uint32_t*>(frdev::pinned_alloc(std::max<size_t>(count, 1) * sizeof(uint32_t)));
                    cap = p ? count : 0;
                }
                return p;
            }
        } slabs[2];
        uint32_t batch_no = 0;
        auto prepare = [this, &dev, &csr, &slabs, budget, t_end = part.t_end](uint32_t t0, uint32_t which) {
            Batch b;
            const auto ts0 = std::chrono::steady_clock::now();
            b.t0 = t0;
            b.root_off.assign(1, 0);
            uint32_t t1 = t0;
            while (t1 < t_end) {
                Rand64 local(seeds_[t1]);
                std::vector<uint32_t> f = features_, q = queries_;
                shuffle(f, local);  // randutil.rs:14-18: shuffle all, take the first n
                f.resize(n_features_);
                shuffle(q, local);
                q.resize(n_queries_);
                std::vector<char> chosen(csr.nq, 0);
                for (uint32_t qi : q) chosen[qi] = 1;
                const size_t before = b.root_ids.size();
                for (size_t qi = 0; qi < csr.nq; qi++) {  // sampling.rs:56-60 in the dataset's query order
                    if (!chosen[qi]) continue;
                    const std::vector<uint32_t>& ids = qids_sorted_[qi];
                    b.root_ids.insert(b.root_ids.end(), ids.begin(), ids.end());
                }
                const size_t items = b.root_ids.size() * n_features_;
                if (t1 > t0 && (items * dev.rf_bytes_per_item() > budget || items >= (size_t(1) << 31))) {
                    b.root_ids.resize(before);  // this tree opens the next batch
                    break;
                }
                if (items >= (size_t(1) << 31)) fail_str("random forest: one tree's sample exceeds the device sort's index range");
                b.feats.insert(b.feats.end(), f.begin(), f.end());
                b.root_off.push_back((uint32_t)b.root_ids.size());
                t1++;
            }
            b.t1 = t1;
            std::string perr;
            b.positions = slabs[which & 1u].get(b.root_ids.size());
            if (b.positions == nullptr) {
                b.positions_own.resize(b.root_ids.size());
                b.positions = b.positions_own.data();
            }
            if (!dev.rf_positions(b.root_ids, b.positions, &perr)) fail_str(perr);
            b.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - ts0).count();
            return b;
        };
        if (part.t_begin >= part.t_end) return;
        std::future<Batch> next = std::async(std::launch::async, prepare, part.t_begin, batch_no++);
        for (;;) {
            Batch b = next.get();
            const bool more = b.t1 < part.t_end;
            if (more) next = std::async(std::launch::async, prepare, b.t1, batch_no++);
            stats.t_sample += b.seconds;
            grow_batch(dev, b.t0, b.t1, b.root_off, b.root_ids, (uint32_t)n_features_, b.feats, b.positions, out, stats);
            stats.batches++;
            if (!more) break;
        }
    }

    const RFStats& stats() const { return stats_; }

  private:
    struct Open {          // a node whose split is being searched this level
        uint32_t tree;     // index inside the batch
        uint32_t key;      // device node key
        uint32_t n;
        TreeNode* node;    // where the result goes (a leaf until it splits)
        double output;     // compute_output of this node (becomes the leaf value)
        uint32_t depth;
    };

    static uint32_t tree_depth(const TreeNode& n) { return n.leaf ? 1 : 1 + std::max(tree_depth(*n.lhs), tree_depth(*n.rhs)); }

    void grow_batch(frdev::DeviceDataset& dev, uint32_t t0, uint32_t t1, const std::vector<uint32_t>& root_off,
                    const std::vector<uint32_t>& root_ids, uint32_t nf, const std::vector<uint32_t>& feats,
                    const uint32_t* positions, Model& out, RFStats& stats_) {
        const uint32_t T = t1 - t0;
        std::string err;
        auto tnow = [] { return std::chrono::steady_clock::now(); };
        auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
        auto tb0 = tnow();
        if (!dev.rf_begin(root_off, root_ids, nf, feats, &err, positions)) fail_str(err);
        stats_.t_begin += secs(tb0, tnow());
        std::vector<std::shared_ptr<TreeNode>> roots(T);
        std::vector<Open> open;
        uint32_t next_key = T;  // keys 0..T-1 are the roots
        for (uint32_t t = 0; t < T; t++) {
            roots[t] = std::make_shared<TreeNode>();
            roots[t]->leaf = true;
            const uint32_t n = root_off[t + 1] - root_off[t];
            if (enterable(n, 1)) open.push_back({t, t, n, roots[t].get(), 0.0, 1});
        }
        const uint32_t k = p_.split_candidates;
        // split_candidates < 2: `(1..k)` is empty (random_forest.rs:236), no feature yields a candidate and every node
        // stays the leaf it is -- nothing to ask the device
        if (k < 2) open.clear();
        while (!open.empty()) {
            std::vector<frdev::DeviceDataset::RfActive> active(open.size());
            std::vector<uint32_t> slot_of_key(next_key, 0xFFFFFFFFu);
            for (size_t a = 0; a < open.size(); a++) {
                active[a] = {open[a].tree, open[a].key, open[a].n};
                slot_of_key[open[a].key] = (uint32_t)a;
                stats_.sorted_items += (uint64_t)open[a].n * nf;
            }
            std::vector<frdev::DeviceDataset::RfCand> cands;
            std::vector<float> lab;
            auto tl0 = tnow();
            if (!dev.rf_level(active, slot_of_key, k, p_.split_method, p_.min_leaf_support, &cands, &lab, &err)) fail_str(err);
            auto tl1 = tnow();
            stats_.t_level += secs(tl0, tl1);
            stats_.levels++;
            stats_.candidates += cands.size();
            std::vector<frdev::DeviceDataset::RfSplit> splits(open.size());
            std::vector<Open> next;
            struct Pending { size_t a; uint32_t left_n, right_n; double sum_l, sum_r; };
            // SquaredError evaluates a candidate by summing the gains of both sides in the segment's order -- the same sums
            // compute_output of the children divides (random_forest.rs:32-41, 395-398): taken from the chosen candidate, no
            // second pass over the node (FR_RF_CHILDSUM=1: the separate pass, for comparison)
            const bool force_childsum = frdev::path_env("FR_RF_CHILDSUM") != nullptr;
            const bool sums_from_cands = p_.split_method == 0 && !force_childsum;
            std::vector<Pending> pend;
            const uint32_t km1 = k >= 2 ? k - 1 : 0;
            for (size_t a = 0; a < open.size(); a++) {
                splits[a] = {-1, 0, 0, 0};
                const Open& o = open[a];
                // label_stats: all labels equal -> no feature yields a candidate (random_forest.rs:218-221)
                if (km1 == 0 || lab[a * 2] == lab[a * 2 + 1]) continue;
                bool have = false;
                double best_imp = 0.0, best_split = 0.0, best_sl = 0.0, best_sr = 0.0;
                uint32_t best_pos = 0, best_fi = 0;
                for (uint32_t fi = 0; fi < nf; fi++) {
                    bool fhave = false;
                    double fimp = 0.0, fsplit = 0.0, fsl = 0.0, fsr = 0.0;
                    uint32_t fpos = 0;
                    for (uint32_t c = 0; c < km1; c++) {
                        const auto& cd = cands[((size_t)a * nf + fi) * km1 + c];
                        if (!(cd.flags & 1)) continue;
                        const uint32_t nl = cd.ids_i, nr = o.n - cd.ids_i;
                        double imp;
                        switch (p_.split_method) {  // random_forest.rs:89-125
                            case 1: imp = -(rf_gini(nl, cd.pos_l) * (double)nl + rf_gini(nr, cd.pos_r) * (double)nr); break;
                            case 2: imp = -(rf_entropy(nl, cd.pos_l) * (double)nl + rf_entropy(nr, cd.pos_r) * (double)nr); break;
                            default: imp = cd.importance; break;
                        }
                        if (imp != imp)  // NotNan::new(..).expect / label_stats(..).unwrap() on None
                            fail_str(p_.split_method == 3 ? "called `Option::unwrap()` on a `None` value (variance of fewer than two labels; "
                                                            "raise min_leaf_support)"
                                                          : "importance was NaN");
                        if (!fhave || imp >= fimp) {  // sort_unstable_by_key(importance).last(), random_forest.rs:275-276
                            fhave = true;
                            fimp = imp;
                            fsplit = cd.position;
                            fpos = cd.ids_i;
                            fsl = cd.sum_l, fsr = cd.sum_r;
                        }
                    }
                    if (fhave && (!have || fimp >= best_imp)) {  // random_forest.rs:391-392
                        have = true;
                        best_imp = fimp;
                        best_split = fsplit;
                        best_pos = fpos;
                        best_fi = fi;
                        best_sl = fsl, best_sr = fsr;
                    }
                }
                if (!have) continue;  // NoFeatureSplitCandidates: stays the leaf it is
                TreeNode* nd = o.node;
                nd->leaf = false;
                nd->fid = feats[(size_t)o.tree * nf + best_fi];
                nd->value = best_split;
                nd->lhs.reset(new TreeNode());
                nd->rhs.reset(new TreeNode());
                splits[a] = {(int32_t)best_fi, best_pos, next_key, next_key + 1};
                next_key += 2;
                pend.push_back({a, best_pos, o.n - best_pos, best_sl, best_sr});
                stats_.nodes += 2;
            }
            std::vector<double> child_out;
            auto tp0 = tnow();
            stats_.t_select += secs(tl1, tp0);
            if (!dev.rf_split(splits, sums_from_cands ? nullptr : &child_out, &err)) fail_str(err);
            stats_.t_split += secs(tp0, tnow());
            for (const Pending& pd : pend) {
                const Open& o = open[pd.a];
                TreeNode* kids[2] = {o.node->lhs.get(), o.node->rhs.get()};
                const uint32_t ns[2] = {pd.left_n, pd.right_n};
                for (int side = 0; side < 2; side++) {
                    kids[side]->leaf = true;
                    // random_forest.rs:395-398 (gain_sum / len, 0.0 for an empty side: rf_childsum_kernel's formula)
                    kids[side]->value = sums_from_cands ? (ns[side] ? (side ? pd.sum_r : pd.sum_l) / (double)ns[side] : 0.0) : child_out[pd.a * 2 + side];
                    const uint32_t key = side ? splits[pd.a].right : splits[pd.a].left;
                    if (enterable(ns[side], o.depth + 1)) next.push_back({o.tree, key, ns[side], kids[side], kids[side]->value, o.depth + 1});
                }
            }
            open.swap(next);
        }
        {
            // random_forest.rs:348-351: a root that did not split is LeafNode(to_output(dataset)) -- rare, so the sequential
            // mean over the whole sample is only computed when some tree needs it
            bool need = false;
            for (uint32_t t = 0; t < T; t++) need = need || roots[t]->leaf;
            if (need) {
                std::vector<double> root_out;
                if (!dev.rf_root_outputs(&root_out, &err)) fail_str(err);
                for (uint32_t t = 0; t < T; t++)
                    if (roots[t]->leaf) roots[t]->value = root_out[t];
            }
        }
        for (uint32_t t = 0; t < T; t++) {
            Model& mm = out.members[t0 + t];
            mm.kind = Model::DecisionTree;
            mm.tree = roots[t];
            stats_.nodes += 1;
            if (p_.weight_trees || !p_.quiet) {
                // random_forest.rs:315: the tree's evaluate_mean over the whole training dataset
                score_model(*view_, mm, &dev);
                std::string e2;
                double mean = 0.0;
                frdev::DeviceDataset& d2 = dev;
                if (!d2.metric_from_scores(ev_.measure, ev_.depth, ev_.norms.data(), 1, false, &e2)) fail_str(e2);
                if (!d2.reduce_means(1, &mean, &e2)) fail_str(e2);
                check_flags(d2);
                if (p_.weight_trees) out.ens_weights[t0 + t] = mean;
                if (!p_.quiet) printf("|%7u|%7u|%7.3f|\n", t0 + t + 1, tree_depth(*roots[t]), mean);
            }
        }
    }

    // random_forest.rs:366-377: may learn_recursive look for a split at all?
    bool enterable(uint32_t n, uint32_t depth) const {
        if (n == 0) return false;                        // StepDone (features are never empty here)
        if (depth >= p_.max_depth) return false;         // DepthExceeded
        if (n < p_.min_leaf_support) return false;       // SplitTooSmall
        return n > 1;                                    // FeatureStats / label_stats need two elements
    }

    std::shared_ptr<DatasetView> view_;
    Evaluator ev_;
    RFParams p_;
    RFStats stats_;
    // shared, read-only while the parts run
    std::vector<uint32_t> features_, queries_;
    size_t n_features_ = 0, n_queries_ = 0;
    std::vector<std::vector<uint32_t>> qids_sorted_;
    std::vector<uint64_t> seeds_;
};

}  // namespace fr
