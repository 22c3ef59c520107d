// Host side of the fastrank hot path: data model, evaluator set-up, model scoring dispatch and
// the coordinate-ascent trainer.  All arithmetic that decides a ranking runs on the device
// (device.hip); the host keeps the reference's *sequential control semantics*
// (src/coordinate_ascent.rs:87-254) and feeds the device batches of candidates.
#pragma once
#include <algorithm>
#include <atomic>
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "device.hpp"
#include "json.hpp"

namespace fr {

using frjson::Value;

// Error carried to the C boundary; `debug` is already in Rust `{:?}` form (src/ffi.rs:45-74).
struct FrError {
    std::string debug;
};
[[noreturn]] inline void fail_str(const std::string& msg) { throw FrError{frjson::rust_debug_str(msg)}; }
[[noreturn]] inline void fail_raw(const std::string& debug) { throw FrError{debug}; }

// ---------------------------------------------------------------------------------------------
// Rand64 -- oorandom =11.1.0 (Cargo.toml:18-19), third-party, source absent from the reference
// tree: restated from the crate's published PCG algorithm.  Seed->trajectory parity with the
// Rust build is UNPINNED (DESIGN.md "Parity status").
// ---------------------------------------------------------------------------------------------
class Rand64 {
  public:
    explicit Rand64(uint64_t seed) {
        state_ = 0;
        inc_ = (default_inc() << 1) | 1;
        (void)rand_u64();
        state_ += (u128)seed;
        (void)rand_u64();
    }
    uint64_t rand_u64() {
        u128 old = state_;
        state_ = old * multiplier() + inc_;
        uint64_t xorshifted = (uint64_t)(((old >> 29) ^ old) >> 58);
        uint32_t rot = (uint32_t)(old >> 122);
        return (xorshifted >> rot) | (xorshifted << ((64 - rot) & 63));
    }
    double rand_float() {
        uint64_t u = rand_u64() >> (64 - 54);
        return (double)u * (1.0 / 18014398509481984.0);
    }
    uint64_t rand_range(uint64_t lo, uint64_t hi) {
        uint64_t s = hi - lo;
        u128 m = (u128)rand_u64() * (u128)s;
        uint64_t leftover = (uint64_t)m;
        if (leftover < s) {
            uint64_t threshold = (0 - s) % s;
            while (leftover < threshold) {
                m = (u128)rand_u64() * (u128)s;
                leftover = (uint64_t)m;
            }
        }
        return (uint64_t)(m >> 64) + lo;
    }

  private:
    typedef unsigned __int128 u128;
    static u128 multiplier() { return (((u128)2549297995355413924ULL) << 64) | (u128)4865540595714422341ULL; }
    static u128 default_inc() { return (((u128)0x2FE0E169FFBD06E3ULL) << 64) | (u128)0x5BC307BD4D2F814FULL; }
    u128 state_, inc_;
};

// src/randutil.rs:21-27
template <typename T>
inline void shuffle(std::vector<T>& v, Rand64& rand) {
    uint64_t n = v.size();
    for (uint64_t i = 0; i < n; i++) {
        uint64_t j = rand.rand_range(i, n);
        std::swap(v[i], v[j]);
    }
}

// ---------------------------------------------------------------------------------------------
// Models (src/model.rs:10-112) and their serde wire form
// ---------------------------------------------------------------------------------------------
struct TreeNode {
    bool leaf = true;
    double value = 0.0;  // leaf value or split threshold
    uint32_t fid = 0;
    std::unique_ptr<TreeNode> lhs, rhs;
};

struct Model {
    enum Kind { SingleFeature, Linear, DecisionTree, Ensemble } kind = Linear;
    uint32_t fid = 0;
    double dir = 0.0;
    std::vector<double> weights;
    std::shared_ptr<TreeNode> tree;
    std::vector<double> ens_weights;
    std::vector<Model> members;
};

inline double json_f64(const Value& v, const char* what) {
    if (!v.is_number()) fail_raw(std::string("Error(\"invalid type: expected f64 for ") + what + "\", line: 0, column: 0)");
    return v.as_double();
}
inline uint64_t json_u64(const Value& v, const char* what) {
    if (v.kind == Value::UInt) return v.u;
    if (v.kind == Value::Int && v.i >= 0) return (uint64_t)v.i;
    fail_raw(std::string("Error(\"invalid type: expected unsigned integer for ") + what + "\", line: 0, column: 0)");
}
// serde rejects integers that do not fit the field's type instead of truncating them
inline uint32_t json_u32(const Value& v, const char* what) {
    const uint64_t x = json_u64(v, what);
    if (x > (uint64_t)UINT32_MAX)
        fail_raw("Error(\"invalid value: integer `" + std::to_string(x) + "`, expected u32\", line: 0, column: 0)");
    return (uint32_t)x;
}
inline bool json_bool(const Value& v, const char* what) {
    if (v.kind != Value::Bool) fail_raw(std::string("Error(\"invalid type: expected a boolean for ") + what + "\", line: 0, column: 0)");
    return v.b;
}
inline const Value& json_field(const Value& obj, const char* key) {
    const Value* f = obj.find(key);
    if (!f) fail_raw(std::string("Error(\"missing field `") + key + "`\", line: 0, column: 0)");
    return *f;
}
// serde externally-tagged enum: {"Variant": payload}
inline const frjson::Member& json_variant(const Value& v, const char* what) {
    if (!v.is_object() || v.obj.size() != 1)
        fail_raw(std::string("Error(\"expected a single-key map for enum ") + what + "\", line: 0, column: 0)");
    return v.obj[0];
}

inline std::unique_ptr<TreeNode> tree_from_json(const Value& v) {
    const auto& var = json_variant(v, "TreeNode");
    auto node = std::make_unique<TreeNode>();
    if (var.first == "LeafNode") {
        node->leaf = true;
        node->value = json_f64(var.second, "LeafNode");
    } else if (var.first == "FeatureSplit") {
        node->leaf = false;
        node->fid = json_u32(json_field(var.second, "fid"), "fid");
        node->value = json_f64(json_field(var.second, "split"), "split");
        node->lhs = tree_from_json(json_field(var.second, "lhs"));
        node->rhs = tree_from_json(json_field(var.second, "rhs"));
    } else {
        fail_raw("Error(\"unknown variant `" + var.first + "`, expected `FeatureSplit` or `LeafNode`\", line: 0, column: 0)");
    }
    return node;
}

inline Value tree_to_json(const TreeNode& n) {
    Value v = Value::object();
    if (n.leaf) {
        v.set("LeafNode", Value::number(n.value));
    } else {
        Value fs = Value::object();
        fs.set("fid", Value::uint(n.fid));
        fs.set("split", Value::number(n.value));
        fs.set("lhs", tree_to_json(*n.lhs));
        fs.set("rhs", tree_to_json(*n.rhs));
        v.set("FeatureSplit", std::move(fs));
    }
    return v;
}

inline Model model_from_json(const Value& v) {
    const auto& var = json_variant(v, "ModelEnum");
    Model m;
    if (var.first == "Linear") {
        m.kind = Model::Linear;
        const Value& w = json_field(var.second, "weights");
        if (!w.is_array()) fail_raw("Error(\"invalid type: expected a sequence for weights\", line: 0, column: 0)");
        for (const auto& x : w.arr) m.weights.push_back(json_f64(x, "weights"));
    } else if (var.first == "SingleFeature") {
        m.kind = Model::SingleFeature;
        m.fid = json_u32(json_field(var.second, "fid"), "fid");
        m.dir = json_f64(json_field(var.second, "dir"), "dir");
    } else if (var.first == "DecisionTree") {
        m.kind = Model::DecisionTree;
        m.tree = std::shared_ptr<TreeNode>(tree_from_json(var.second).release());
    } else if (var.first == "Ensemble") {
        m.kind = Model::Ensemble;
        const Value& w = json_field(var.second, "weights");
        const Value& ms = json_field(var.second, "models");
        if (!w.is_array() || !ms.is_array())
            fail_raw("Error(\"invalid type: expected sequences in Ensemble\", line: 0, column: 0)");
        for (const auto& x : w.arr) m.ens_weights.push_back(json_f64(x, "weights"));
        for (const auto& x : ms.arr) m.members.push_back(model_from_json(x));
    } else {
        fail_raw("Error(\"unknown variant `" + var.first +
                 "`, expected one of `SingleFeature`, `Linear`, `DecisionTree`, `Ensemble`\", line: 0, column: 0)");
    }
    return m;
}

inline Value model_to_json(const Model& m) {
    Value v = Value::object();
    switch (m.kind) {
        case Model::Linear: {
            Value inner = Value::object();
            Value w = Value::array();
            for (double x : m.weights) w.push(Value::number(x));
            inner.set("weights", std::move(w));
            v.set("Linear", std::move(inner));
            break;
        }
        case Model::SingleFeature: {
            Value inner = Value::object();
            inner.set("fid", Value::uint(m.fid));
            inner.set("dir", Value::number(m.dir));
            v.set("SingleFeature", std::move(inner));
            break;
        }
        case Model::DecisionTree:
            v.set("DecisionTree", tree_to_json(*m.tree));
            break;
        case Model::Ensemble: {
            Value inner = Value::object();
            Value w = Value::array();
            for (double x : m.ens_weights) w.push(Value::number(x));
            Value ms = Value::array();
            for (const auto& mm : m.members) ms.push(model_to_json(mm));
            inner.set("weights", std::move(w));
            inner.set("models", std::move(ms));
            v.set("Ensemble", std::move(inner));
            break;
        }
    }
    return v;
}

inline int32_t flatten_tree(const TreeNode& n, frdev::FlatTrees& out) {
    int32_t idx = (int32_t)out.fid.size();
    out.fid.push_back(-1);
    out.split.push_back(n.value);
    out.lhs.push_back(-1);
    out.rhs.push_back(-1);
    if (!n.leaf) {
        out.fid[idx] = (int32_t)n.fid;
        int32_t l = flatten_tree(*n.lhs, out);
        int32_t r = flatten_tree(*n.rhs, out);
        out.lhs[idx] = l;
        out.rhs[idx] = r;
    }
    return idx;
}

// ---------------------------------------------------------------------------------------------
// Judgments (src/qrel.rs)
// ---------------------------------------------------------------------------------------------
struct QRel {
    // qid -> (docid -> gain); insertion order kept for stable JSON
    std::vector<std::pair<std::string, std::vector<std::pair<std::string, float>>>> queries;
    std::unordered_map<std::string, size_t> index;

    const std::vector<std::pair<std::string, float>>* get(const std::string& qid) const {
        auto it = index.find(qid);
        return it == index.end() ? nullptr : &queries[it->second].second;
    }
    void insert(const std::string& qid, const std::string& docid, float gain) {
        auto it = index.find(qid);
        size_t k;
        if (it == index.end()) {
            k = queries.size();
            index.emplace(qid, k);
            queries.emplace_back(qid, std::vector<std::pair<std::string, float>>());
        } else {
            k = it->second;
        }
        for (auto& dg : queries[k].second)
            if (dg.first == docid) {
                dg.second = gain;
                return;
            }
        queries[k].second.emplace_back(docid, gain);
    }
    void ensure_query(const std::string& qid) {
        if (index.find(qid) == index.end()) {
            index.emplace(qid, queries.size());
            queries.emplace_back(qid, std::vector<std::pair<std::string, float>>());
        }
    }
};

inline QRel qrel_from_json(const Value& v) {
    if (!v.is_object()) fail_raw("Error(\"invalid type: expected a map\", line: 1, column: 1)");
    QRel q;
    for (const auto& qm : v.obj) {
        if (!qm.second.is_object()) fail_raw("Error(\"invalid type: expected a map\", line: 1, column: 1)");
        q.ensure_query(qm.first);
        for (const auto& dm : qm.second.obj) {
            double g = json_f64(dm.second, "gain");
            q.insert(qm.first, dm.first, (float)g);
        }
    }
    return q;
}

inline Value qrel_query_to_json(const std::vector<std::pair<std::string, float>>& docs) {
    Value o = Value::object();
    for (const auto& dg : docs) o.obj.emplace_back(dg.first, Value::number32(dg.second));
    return o;
}

inline Value qrel_to_json(const QRel& q) {
    Value o = Value::object();
    for (const auto& qq : q.queries) o.obj.emplace_back(qq.first, qrel_query_to_json(qq.second));
    return o;
}

// ---------------------------------------------------------------------------------------------
// Datasets
// ---------------------------------------------------------------------------------------------
struct DataCore {
    size_t n = 0, d = 0;          // instances, n_dim (matrix columns)
    const float* x = nullptr;     // row-major n x d (borrowed or -> x_own)
    std::vector<float> x_own;
    std::vector<float> gain;      // f32 gain per instance (dense_dataset.rs:114-123: ys[i] as f32)
    std::vector<uint32_t> qix;    // per-instance query index
    std::vector<std::string> qnames;  // query index -> qid string (first-appearance order)
    bool has_docids = false;
    std::vector<std::string> docids;
    std::vector<uint8_t> doc_present;
    std::vector<uint32_t> features;   // feature ids present (ascending)
    std::map<uint32_t, std::string> feature_names;
    bool is_dense_borrowed = false;
    // File-loaded datasets: which features each instance HOLDS (src/instance.rs:64-74: a Dense32 row holds every index
    // below its length, a Sparse32 row the listed ones).  An absent value reads 0.0 wherever a value is needed, but
    // FeatureStats skips it (src/normalizers.rs:24-29) -- random-forest training needs the difference.  Bits
    // [n][present_words] (feature f: word f / 32, bit f % 32); empty when every instance holds every feature.
    std::vector<uint32_t> present_bits;
    size_t present_words = 0;
    // One device job at a time per DATASET (every view of a core shares its device forms and their work buffers);
    // calls on different datasets run concurrently -- each on the devices it was given.
    std::mutex api_mu;

    std::string feature_name(uint32_t fid) const {
        auto it = feature_names.find(fid);
        return it == feature_names.end() ? std::to_string(fid) : it->second;
    }
};

struct Evaluator {
    int measure = frdev::M_NDCG;
    int64_t depth = -1;
    std::string name;
    std::vector<double> norms;  // per CSR query
};

struct DatasetView {
    std::shared_ptr<DataCore> core;
    std::vector<uint32_t> features;   // feature ids visible to trainers
    std::vector<uint32_t> instances;  // instance ids of this view, iteration order
    bool sampled = false;
    // The view this one was sampled from (dataset_query_sampling / dataset_feature_sampling).  On the device a sampled
    // view does not get a matrix of its own: a feature sample uses its parent's DeviceDataset as it is (feature lists
    // only steer the trainers), a query sample gets query / run tables over the parent's tiles (DeviceDataset::create_view).
    std::shared_ptr<DatasetView> parent;
    bool same_instances_as_parent = false;

    // lazily built device form
    std::mutex mu;
    bool csr_built = false;
    frdev::HostCSR csr;
    std::vector<uint32_t> csr_query;  // CSR query -> core query index
    std::shared_ptr<frdev::DeviceDataset> dev;

    uint32_t n_dim() const { return sampled ? (uint32_t)features.size() : (uint32_t)core->d; }

    void build_csr() {
        if (csr_built) return;
        const DataCore& c = *core;
        const auto t_csr0 = std::chrono::steady_clock::now();
        for (uint32_t id : instances)  // (the reference panics on a NaN label: src/dense_dataset.rs:114-123)
            if (c.gain[id] != c.gain[id]) fail_str("NaN in ys[" + std::to_string(id) + "]");
        // queries in first-appearance order over this view's instances (counting pass, then fill)
        std::vector<int32_t> slot(c.qnames.size(), -1);  // core query index -> CSR query
        std::vector<uint32_t> count;
        for (uint32_t id : instances) {
            const uint32_t qi = c.qix[id];
            if (slot[qi] < 0) {
                slot[qi] = (int32_t)count.size();
                count.push_back(0);
                csr_query.push_back(qi);
            }
            count[(size_t)slot[qi]]++;
        }
        csr.n = instances.size();
        csr.d = c.d;
        csr.nq = count.size();
        csr.x = c.x;
        csr.qoff.assign(csr.nq + 1, 0);
        for (size_t q = 0; q < csr.nq; q++) csr.qoff[q + 1] = csr.qoff[q] + count[q];
        csr.perm.assign(csr.n, 0);
        {
            std::vector<uint32_t> fill(csr.qoff.begin(), csr.qoff.end() - 1);
            for (uint32_t id : instances) csr.perm[fill[(size_t)slot[c.qix[id]]]++] = id;
        }
        // reverse tie-break layout inside each query: gain desc, instance id desc (device.hpp header comment)
        {
            unsigned hw = std::thread::hardware_concurrency();
            const size_t nthreads = std::max<size_t>(1, std::min<size_t>(hw ? hw : 1, csr.n > 200000 ? 16 : 1));
            auto work = [&](size_t tid) {
                for (size_t q = tid; q < csr.nq; q += nthreads)
                    std::sort(csr.perm.begin() + csr.qoff[q], csr.perm.begin() + csr.qoff[q + 1], [&](uint32_t a2, uint32_t b2) {
                        const float ga = c.gain[a2], gb = c.gain[b2];
                        if (ga != gb) return ga > gb;
                        return a2 > b2;
                    });
            };
            std::vector<std::thread> pool;
            for (size_t tid = 1; tid < nthreads; tid++) pool.emplace_back(work, tid);
            work(0);
            for (auto& th : pool) th.join();
        }
        csr.gain.resize(csr.n);
        for (size_t p = 0; p < csr.n; p++) csr.gain[p] = c.gain[csr.perm[p]];
        csr_built = true;
        if (getenv("FR_UPLOAD_TIMING"))
            fprintf(stderr, "[upload] %-28s %7.1f ms\n", "build_csr (group, sort)",
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_csr0).count());
    }

    // the view that owns the device matrix this one's data lives in, and whether this view covers all of its instances
    DatasetView* matrix_owner(bool* all_instances) {
        DatasetView* v = this;
        bool all = true;
        while (v->parent) {
            all = all && v->same_instances_as_parent;
            v = v->parent.get();
        }
        if (all_instances) *all_instances = all;
        return v;
    }
    // queries of this view -> queries of the view that owns the matrix, through the core's query index (false: some
    // query of this view is not one of the owner's)
    bool owner_query_map(DatasetView* owner, std::vector<uint32_t>& pq) {
        std::vector<int32_t> owner_q(core->qnames.size(), -1);
        for (size_t q = 0; q < owner->csr_query.size(); q++) owner_q[owner->csr_query[q]] = (int32_t)q;
        pq.resize(csr.nq);
        for (size_t q = 0; q < csr.nq; q++) {
            if (owner_q[csr_query[q]] < 0) return false;
            pq[q] = (uint32_t)owner_q[csr_query[q]];
        }
        return true;
    }
    std::shared_ptr<frdev::DeviceDataset> device_ptr() {
        std::lock_guard<std::mutex> lk(mu);
        if (dev) return dev;
        build_csr();
        for (size_t p = 0; p < csr.n; p++)
            if (csr.gain[p] != csr.gain[p]) fail_str("NaN in ys[" + std::to_string(csr.perm[p]) + "]");
        std::string err;
        bool all = true;
        DatasetView* owner = matrix_owner(&all);
        if (owner != this && !frdev::path_env("FR_VIEW_COPIES")) {  // (FR_VIEW_COPIES=1: every view tiles its own matrix, as in round 1)
            std::shared_ptr<frdev::DeviceDataset> pdev = owner->device_ptr();
            if (all) {
                dev = pdev;  // same documents: the parent's device dataset as it is
            } else {
                // a view the parent's layout cannot express (a query the owner does not hold, documents in another
                // order) tiles its own matrix below, as every view did in round 1, instead of failing
                std::vector<uint32_t> pq;
                if (owner_query_map(owner, pq)) dev = frdev::DeviceDataset::create_view(pdev, csr, pq, &err);
            }
            if (dev) return dev;
            err.clear();
        }
        dev = frdev::DeviceDataset::create(csr, &err);
        if (!dev) fail_str(err);
        return dev;
    }
    // The device form of this view in another context: slot 0 is device_ptr() (built on the device that was current at
    // first use); slot k > 0 lives on device `device` and is a device-to-device copy of slot 0's matrix
    // (DeviceDataset::replicate), or -- for a sampled view -- the same kind of view over its owner's copy there.
    // train_model keeps one slot per device it spreads a request's restarts over (capi.cpp).
    std::vector<std::shared_ptr<frdev::DeviceDataset>> replicas;  // [slot - 1]
    std::shared_ptr<frdev::DeviceDataset> device_ptr(int slot, int device) {
        if (slot <= 0) return device_ptr();
        std::shared_ptr<frdev::DeviceDataset> primary = device_ptr();
        bool all = true;
        DatasetView* owner = matrix_owner(&all);
        std::shared_ptr<frdev::DeviceDataset> owner_rep;
        const bool aliases_owner = owner != this && (primary.get() == owner->device_ptr().get() || primary->shares_parent_matrix());
        if (aliases_owner) owner_rep = owner->device_ptr(slot, device);
        {
            std::lock_guard<std::mutex> lk(mu);
            if (replicas.size() < (size_t)slot) replicas.resize((size_t)slot);
            const std::shared_ptr<frdev::DeviceDataset>& have = replicas[(size_t)slot - 1];
            if (have && have->device_ordinal() == device) return have;
        }
        // (the copy itself runs outside the lock: train_model makes the copies for all devices of its list at the same
        // time -- every peer has its own xGMI link to the first device -- and no two callers ever fill the same slot)
        std::shared_ptr<frdev::DeviceDataset> r;
        std::string err;
        if (aliases_owner) {
            if (!primary->shares_parent_matrix()) {
                r = owner_rep;
            } else {
                std::vector<uint32_t> pq;
                if (!owner_query_map(owner, pq)) fail_str("sampled view: query not in the dataset it was sampled from");
                r = frdev::DeviceDataset::create_view(owner_rep, csr, pq, &err);
            }
        } else {
            r = frdev::DeviceDataset::replicate(primary, device, &err);
        }
        if (!r) fail_str(err.empty() ? "could not copy the dataset to device " + std::to_string(device) : err);
        std::lock_guard<std::mutex> lk(mu);
        if (replicas.size() < (size_t)slot) replicas.resize((size_t)slot);
        replicas[(size_t)slot - 1] = r;
        return r;
    }
    // drops the copies on other devices / in other contexts (slot 0 stays); returns how many there were
    size_t release_replicas() {
        std::lock_guard<std::mutex> lk(mu);
        size_t n = 0;
        for (auto& r : replicas) n += r ? 1 : 0;
        replicas.clear();
        return n;
    }
    // ordinal of the device the first device form lives on; -1 while there is none yet
    int built_on_device() {
        std::lock_guard<std::mutex> lk(mu);
        return dev ? dev->device_ordinal() : -1;
    }
    frdev::DeviceDataset& device() { return *device_ptr(); }
    frdev::DeviceDataset& device(int slot, int dev_ordinal) { return *device_ptr(slot, dev_ordinal); }
    const frdev::HostCSR& host_csr() {
        std::lock_guard<std::mutex> lk(mu);
        build_csr();
        return csr;
    }
};

// src/dense_dataset.rs:28-55
inline std::shared_ptr<DatasetView> make_dense(size_t n, size_t d, const float* x, const double* y,
                                               const int64_t* qids) {
    auto core = std::make_shared<DataCore>();
    core->n = n;
    core->d = d;
    core->x = x;
    core->is_dense_borrowed = true;
    core->gain.resize(n);
    core->qix.resize(n);
    std::unordered_map<uint32_t, uint32_t> seen;
    for (size_t i = 0; i < n; i++) {
        int64_t q = qids[i];
        if (q < 0 || q > (int64_t)UINT32_MAX) fail_raw("TryFromIntError(())");  // u32::try_from(i64)?
        auto it = seen.find((uint32_t)q);
        if (it == seen.end()) {
            it = seen.emplace((uint32_t)q, (uint32_t)core->qnames.size()).first;
            core->qnames.push_back(std::to_string((uint32_t)q));
        }
        core->qix[i] = it->second;
        core->gain[i] = (float)y[i];
    }
    for (size_t j = 0; j < d; j++) core->features.push_back((uint32_t)j);
    auto view = std::make_shared<DatasetView>();
    view->core = core;
    view->features = core->features;
    view->instances.resize(n);
    for (size_t i = 0; i < n; i++) view->instances[i] = (uint32_t)i;
    return view;
}

// src/evaluators.rs:255-272 on gains already sorted descending; depth<0 = None
inline double ideal_dcg_sorted_desc(const std::vector<float>& g, int64_t depth) {
    // zero-padding up to `depth` adds (2^0 - 1)/log2(i+2) = +0.0 terms, which leave the sum unchanged:
    // stop at the list's end (a depth of 10^18 must not loop)
    size_t len = depth >= 0 ? std::min<size_t>((size_t)depth, g.size()) : g.size();
    double dcg = 0.0;
    for (size_t i = 0; i < len; i++) {
        double gain = (double)g[i];
        double term = (std::pow(2.0, gain) - 1.0) / std::log2((double)i + 2.0);
        dcg = dcg + term;
    }
    return dcg;
}

// src/evaluators.rs:132-155 (+ NDCG::new :303-340, AveragePrecision::new :389-419)
inline Evaluator make_evaluator(DatasetView& view, const std::string& orig_name, const QRel* qrel) {
    Evaluator ev;
    std::string name = orig_name;
    size_t at = orig_name.find('@');
    if (at != std::string::npos) {
        std::string rhs = orig_name.substr(at);
        std::string num = rhs.substr(1);
        bool ok = !num.empty();
        size_t k = (num.size() > 1 && num[0] == '+') ? 1 : 0;
        for (size_t t = k; t < num.size(); t++) ok = ok && num[t] >= '0' && num[t] <= '9';
        // usize::from_str: any digit count as long as the value fits 64 bits (leading zeros are fine)
        uint64_t depth = 0;
        if (ok) {
            auto r = std::from_chars(num.data() + k, num.data() + num.size(), depth);
            ok = r.ec == std::errc() && r.ptr == num.data() + num.size();
        }
        if (!ok) fail_str("Couldn't parse after the @ in \"" + orig_name + "\": " + rhs);
        // depths beyond any possible query length behave like "longer than the list": clamp for the i64 field
        ev.depth = (int64_t)std::min<uint64_t>(depth, (uint64_t)INT64_MAX / 2);
        name = orig_name.substr(0, at);
    }
    for (auto& ch : name) ch = (char)std::tolower((unsigned char)ch);
    if (name == "ap" || name == "map") {
        ev.measure = frdev::M_AP;
        ev.name = "AP";
    } else if (name == "rr" || name == "mrr") {
        ev.measure = frdev::M_RR;
        ev.name = "RR";
    } else if (name == "ndcg") {
        ev.measure = frdev::M_NDCG;
        ev.name = ev.depth >= 0 ? "NDCG@" + std::to_string(ev.depth) : "NDCG";
    } else {
        fail_str("Invalid training measure: \"" + orig_name + "\"");
    }
    const frdev::HostCSR& csr = view.host_csr();
    const DataCore& c = *view.core;
    ev.norms.assign(csr.nq, 0.0);
    for (size_t q = 0; q < csr.nq; q++) {
        const std::string& qid = c.qnames[view.csr_query[q]];
        const std::vector<std::pair<std::string, float>>* judged = qrel ? qrel->get(qid) : nullptr;
        if (ev.measure == frdev::M_NDCG) {
            std::vector<float> gains;
            if (judged) {
                for (const auto& dg : *judged)
                    if (dg.second > 0.0f) gains.push_back(dg.second);  // qrel.rs:33-39 gain_vector
                std::sort(gains.begin(), gains.end(), [](float a, float b) { return a > b; });
            } else {
                // the view's docs are stored gain-descending already
                gains.assign(csr.gain.begin() + csr.qoff[q], csr.gain.begin() + csr.qoff[q + 1]);
            }
            size_t pos = 0;
            for (float g : gains) pos += g > 0.0f;
            ev.norms[q] = pos == 0 ? std::nan("") : ideal_dcg_sorted_desc(gains, ev.depth);
        } else if (ev.measure == frdev::M_AP) {
            uint32_t nr = 0;
            if (judged) {
                for (const auto& dg : *judged) nr += dg.second > 0.0f;
            } else {
                for (uint32_t p = csr.qoff[q]; p < csr.qoff[q + 1]; p++) nr += csr.gain[p] > 0.0f;
            }
            ev.norms[q] = (double)nr;  // 0 => kernel falls back to the ranked list's own count
        }
    }
    return ev;
}

inline void check_flags(frdev::DeviceDataset& dev) {
    int fl = dev.take_flags();
    if (fl & frdev::FLAG_NAN_SCORE) fail_str("Model.predict -> NaN");  // src/model.rs:49
    if (fl & frdev::FLAG_ACTUAL_GT_IDEAL)
        fail_str("actual DCG exceeds ideal DCG for some query (the reference panics here, src/evaluators.rs:368-374)");
}

#define FR_DEV(call)                      \
    do {                                  \
        std::string _err;                 \
        if (!(call)) fr::fail_str(_err);  \
    } while (0)

// Score `model` for every document of the view into device score slot 0.
inline void score_model(DatasetView& view, const Model& m, frdev::DeviceDataset* on = nullptr) {
    frdev::DeviceDataset& dev = on ? *on : view.device();  // (on: one of the view's device-side copies, DatasetView::device_ptr(slot, ..))
    std::string _err;
    switch (m.kind) {
        case Model::Linear: {
            std::vector<double> w(dev.d(), 0.0);
            for (size_t j = 0; j < w.size() && j < m.weights.size(); j++) w[j] = m.weights[j];
            if (!dev.score_linear(1, w.data(), &_err)) fail_str(_err);
            break;
        }
        case Model::SingleFeature:
            if (!dev.score_single_feature(m.fid, m.dir, &_err)) fail_str(_err);
            break;
        case Model::DecisionTree: {
            frdev::FlatTrees ft;
            ft.raw_single = true;
            ft.root.push_back(flatten_tree(*m.tree, ft));
            ft.weight.push_back(1.0);
            if (!dev.score_trees(ft, &_err)) fail_str(_err);
            break;
        }
        case Model::Ensemble: {
            bool all_trees = !m.members.empty();
            for (const auto& mm : m.members) all_trees = all_trees && mm.kind == Model::DecisionTree;
            if (all_trees) {
                frdev::FlatTrees ft;
                for (size_t t = 0; t < m.members.size(); t++) {
                    ft.root.push_back(flatten_tree(*m.members[t].tree, ft));
                    ft.weight.push_back(t < m.ens_weights.size() ? m.ens_weights[t] : 0.0);
                }
                // zip(weights, models) stops at the shorter (src/model.rs:106)
                size_t nt = std::min(m.members.size(), m.ens_weights.size());
                ft.root.resize(nt);
                ft.weight.resize(nt);
                if (!dev.score_trees(ft, &_err)) fail_str(_err);
            } else {
                if (!dev.ensemble_begin(&_err)) fail_str(_err);
                size_t nt = std::min(m.members.size(), m.ens_weights.size());
                for (size_t t = 0; t < nt; t++) {
                    const Model& mm = m.members[t];
                    if (mm.kind == Model::Ensemble) {
                        bool trees = !mm.members.empty();
                        for (const auto& x : mm.members) trees = trees && x.kind == Model::DecisionTree;
                        if (!trees)
                            fail_str("nested mixed ensembles are not supported by the MI355X scoring path");
                    }
                    score_model(view, mm, &dev);
                    if (!dev.ensemble_accumulate(m.ens_weights[t], &_err)) fail_str(_err);
                }
                if (!dev.ensemble_finish(&_err)) fail_str(_err);
            }
            break;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Coordinate ascent (src/coordinate_ascent.rs)
// ---------------------------------------------------------------------------------------------
struct CAParams {
    uint32_t num_restarts = 5;
    uint32_t num_max_iterations = 25;
    double step_base = 0.05;
    double step_scale = 2.0;
    double tolerance = 0.001;
    uint64_t seed = 0;
    bool normalize = true;
    bool quiet = false;
    bool init_random = true;
    bool output_ensemble = false;

    static CAParams defaults() {
        CAParams p;
        Rand64 rand(0xdeadbeefULL);
        p.seed = rand.rand_u64();  // coordinate_ascent.rs:27,34
        return p;
    }
    static CAParams from_json(const Value& v) {
        CAParams p;
        p.num_restarts = json_u32(json_field(v, "num_restarts"), "num_restarts");
        p.num_max_iterations = json_u32(json_field(v, "num_max_iterations"), "num_max_iterations");
        p.step_base = json_f64(json_field(v, "step_base"), "step_base");
        p.step_scale = json_f64(json_field(v, "step_scale"), "step_scale");
        p.tolerance = json_f64(json_field(v, "tolerance"), "tolerance");
        p.seed = json_u64(json_field(v, "seed"), "seed");
        p.normalize = json_bool(json_field(v, "normalize"), "normalize");
        p.quiet = json_bool(json_field(v, "quiet"), "quiet");
        p.init_random = json_bool(json_field(v, "init_random"), "init_random");
        p.output_ensemble = json_bool(json_field(v, "output_ensemble"), "output_ensemble");
        return p;
    }
    Value to_json() const {
        Value o = Value::object();
        o.set("num_restarts", Value::uint(num_restarts));
        o.set("num_max_iterations", Value::uint(num_max_iterations));
        o.set("step_base", Value::number(step_base));
        o.set("step_scale", Value::number(step_scale));
        o.set("tolerance", Value::number(tolerance));
        o.set("seed", Value::uint(seed));
        o.set("normalize", Value::boolean(normalize));
        o.set("quiet", Value::boolean(quiet));
        o.set("init_random", Value::boolean(init_random));
        o.set("output_ensemble", Value::boolean(output_ensemble));
        return o;
    }
};

struct RestartResult {
    uint32_t restart_id = 0;
    double score = 0.0;
    std::vector<double> weights;
};

// The exchange records of the job's one all-gather (src/coordinate_ascent.rs:232-252 selects from them): a
// rank's restarts as a fixed-size block of `cap` records of 3 + d doubles -- valid (1.0 / 0.0 padding), restart id, score,
// weights[d] -- the same layout native.gather_restarts puts through torch.distributed and rccl_allgather puts through RCCL.
inline size_t restart_record_len(size_t d) { return 3 + d; }
inline void pack_restart_records(const std::vector<RestartResult>& mine, size_t cap, size_t d, double* out) {
    if (mine.size() > cap) fail_str("pack_restart_records: " + std::to_string(mine.size()) + " restarts for a block of " + std::to_string(cap));
    const size_t rl = restart_record_len(d);
    std::fill(out, out + cap * rl, 0.0);
    for (size_t k = 0; k < mine.size(); k++) {
        if (mine[k].weights.size() > d) fail_str("pack_restart_records: a restart has more weights than the record holds");
        double* r = out + k * rl;
        r[0] = 1.0;
        r[1] = (double)mine[k].restart_id;
        r[2] = mine[k].score;
        std::copy(mine[k].weights.begin(), mine[k].weights.end(), r + 3);
    }
}
// every valid record of `n` records, in restart order; each id must occur exactly once when `expect` > 0 (= the ids 0..expect-1)
inline std::vector<RestartResult> unpack_restart_records(const double* in, size_t n, size_t d, size_t expect = 0) {
    const size_t rl = restart_record_len(d);
    std::vector<RestartResult> out;
    for (size_t k = 0; k < n; k++) {
        const double* r = in + k * rl;
        if (r[0] != 1.0) continue;
        RestartResult x;
        x.restart_id = (uint32_t)r[1];
        x.score = r[2];
        x.weights.assign(r + 3, r + 3 + d);
        out.push_back(std::move(x));
    }
    std::sort(out.begin(), out.end(), [](const RestartResult& a, const RestartResult& b) { return a.restart_id < b.restart_id; });
    if (expect > 0) {
        if (out.size() != expect) fail_str("internal error: the exchange carried " + std::to_string(out.size()) + " of " + std::to_string(expect) + " restarts");
        for (size_t r = 0; r < expect; r++)
            if (out[r].restart_id != r) fail_str("internal error: restart " + std::to_string(r) + " was not trained exactly once");
    }
    return out;
}

struct TrainStats {
    uint64_t useful_evals = 0;  // evaluate_mean calls the sequential reference would have made
    uint64_t raw_evals = 0;     // candidates actually evaluated (incl. speculative ones)
    uint64_t ticks = 0;         // batched launches
    uint64_t groups = 0;
    double seconds = 0.0;
    std::string path;           // "fused_linesearch" | "fused_fullrank" | "generic_sort"
    uint32_t restarts = 0;
    // fused_linesearch: (run, group) pairs evaluated by the bound-and-verify kernel, and how many of
    // them it could not verify (recomputed by the exact kernel)
    uint64_t verify_pairs = 0, verify_redone = 0;
    uint64_t line_searches = 0;  // batched line searches submitted (one per tick in lock step, one per set and tick when pipelined)
    uint64_t audit_values = 0, audit_mismatches = 0;  // FR_VERIFY_AUDIT=1 (see include/fastrank.h)
    uint64_t exact_ticks = 0;  // line searches evaluated by the exact kernels alone (every group routed there / after a tick with > 25 % redone pairs)
    uint64_t exact_groups = 0;         // NDCG@k: group line searches routed to the exact kernel (of `groups`)
    uint64_t verify_redo_entries = 0;  // NDCG@k: (query, group, 16-candidate slice) entries the exact kernel recomputed
    uint64_t chain_runs = 0, chain_visits = 0;  // NDCG@k verify kernel: insertion-chain runs out of (document, group) visits
    uint64_t rank_slots_on = 0, rank_slots_off = 0;  // restarts whose R ranks were found worth keeping / not (chain runs of their first line searches)
    uint32_t devices = 1;      // devices train_model spread the restarts over (ticks = the longest device's)
    uint32_t refills = 0;      // times converged restarts handed their places to the next ids of the restart queue
    int device = -1;           // ordinal this trainer ran on (per-device entries of train_model's statistics)
    // train_model over several devices: the exchange of the restarts' records as an RCCL all-gather (rccl_exchange.inc)
    bool rccl_set = false;
    frdev::RcclReport rccl;
};

// adds the exact-only line searches of a scope to the trainer's statistics (also when the scope unwinds)
struct ExactTickCount {
    frdev::DeviceDataset& dev;
    TrainStats& st;
    unsigned long long base, av0 = 0, am0 = 0, eg0 = 0, re0 = 0, cr0 = 0, cv0 = 0, on0 = 0, off0 = 0;
    ExactTickCount(frdev::DeviceDataset& d, TrainStats& s) : dev(d), st(s), base(d.exact_fallbacks()) {
        d.audit_counters(&av0, &am0);
        d.routing_counters(&eg0, &re0);
        d.chain_counters(&cr0, &cv0, &on0, &off0);
    }
    ~ExactTickCount() {
        unsigned long long av1 = 0, am1 = 0, eg1 = 0, re1 = 0, cr1 = 0, cv1 = 0, on1 = 0, off1 = 0;
        dev.audit_counters(&av1, &am1);
        dev.routing_counters(&eg1, &re1);
        dev.chain_counters(&cr1, &cv1, &on1, &off1);
        st.rank_slots_on += on1 - on0;
        st.rank_slots_off += off1 - off0;
        st.chain_runs += cr1 - cr0;
        st.chain_visits += cv1 - cv0;
        st.exact_groups += eg1 - eg0;
        st.verify_redo_entries += re1 - re0;
        st.exact_ticks += dev.exact_fallbacks() - base;
        st.audit_values += av1 - av0;
        st.audit_mismatches += am1 - am0;
    }
};

// coordinate_ascent.rs:72-82
inline void l1_normalize(std::vector<double>& w) {
    double sum = 0.0;
    for (double x : w) sum += std::fabs(x);
    if (sum > 0.0)
        for (double& x : w) x /= sum;
}

// Candidate weights of one line search in evaluation order (coordinate_ascent.rs:145-171):
// block 0 = dir 0 (1 candidate), block 1 = dir -1, block 2 = dir +1.
inline void line_candidates(double orig, const CAParams& p, std::vector<double>& out, uint32_t block_len[3]) {
    static const int SIGN[3] = {0, -1, 1};
    out.clear();
    for (int s = 0; s < 3; s++) {
        double dir = (double)SIGN[s];
        double step = p.step_base * dir;
        if (orig != 0.0 && std::fabs(step) > 0.5 * std::fabs(orig)) step = p.step_base * std::fabs(orig) * dir;
        double total = step;
        uint32_t iters = p.num_max_iterations;
        if (SIGN[s] == 0) {
            iters = 1;
            total = -orig;
        }
        block_len[s] = iters;
        for (uint32_t it = 0; it < iters; it++) {
            out.push_back(orig + total);
            step *= p.step_scale;
            total += step;
        }
    }
}

// Batched evaluate_mean of arbitrary weight vectors through the general sort path.
inline void evaluate_means_generic(frdev::DeviceDataset& dev, const Evaluator& ev, const std::vector<double>& weights,
                                   size_t B, std::vector<double>& means) {
    const size_t d = dev.d();
    means.assign(B, 0.0);
    // bound the score scratch to ~2 GiB
    size_t ld = (dev.n() + 63) / 64 * 64;
    size_t chunk = std::max<size_t>(1, std::min<size_t>(512, (size_t(2) << 30) / (ld * sizeof(double))));
    for (size_t b0 = 0; b0 < B; b0 += chunk) {
        size_t bn = std::min(chunk, B - b0);
        std::string _err;
        if (!dev.score_linear(bn, weights.data() + b0 * d, &_err)) fail_str(_err);
        if (!dev.metric_from_scores(ev.measure, ev.depth, ev.norms.data(), bn, false, &_err)) fail_str(_err);
        if (!dev.reduce_means(bn, means.data() + b0, &_err)) fail_str(_err);
        check_flags(dev);
    }
}

// learn(): coordinate_ascent.rs:198-253, restarts [rbegin, rend) of num_restarts.  All live
// restarts advance in lock step: one device launch per "feature tick" evaluates every
// candidate of every live restart's current line search; the host then replays the
// reference's sequential accept / early-break logic on the returned means.
// Query-sharded training (SURVEY 8e fallback for fewer restarts than GPUs): this process holds a
// contiguous block of the queries; `allreduce` must replace values[i] by the sum over all ranks,
// added in rank order (so every rank gets the same bits), and `total_queries` is the global count.
struct QueryShard {
    std::function<void(double* values, size_t n)> allreduce;
    uint64_t total_queries = 0;
};

// The restart ids of one request that no trainer has started yet.  The reference hands its restarts to rayon's
// work-stealing pool (src/coordinate_ascent.rs:215-225): whichever worker is free takes the next one.  Here the workers
// are the trainers of train_model's devices (capi.cpp, train_ca_devices): each keeps a bounded number of restarts live
// and pulls the next id from this one counter whenever one of its restarts has converged.  Which trainer runs a restart
// does not matter -- its trajectory depends only on its child seed (:211-213) -- so the model is the single trainer's.
class RestartQueue {
  public:
    RestartQueue(uint32_t begin, uint32_t end) : next_(begin), end_(std::max(begin, end)) {}
    bool pop(uint32_t* id) {
        uint32_t v = next_.load(std::memory_order_relaxed);
        while (v < end_)
            if (next_.compare_exchange_weak(v, v + 1, std::memory_order_relaxed)) {
                *id = v;
                return true;
            }
        return false;
    }
    bool empty() const { return next_.load(std::memory_order_relaxed) >= end_; }

  private:
    std::atomic<uint32_t> next_;
    const uint32_t end_;
};

// accumulates the wall time of a scope in microseconds (FR_HOST_TIMING=1 prints the trainer's totals when it is destroyed)
struct HostTimer {
    double& acc;
    std::chrono::steady_clock::time_point t0;
    explicit HostTimer(double& a) : acc(a), t0(std::chrono::steady_clock::now()) {}
    ~HostTimer() { acc += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); }
};

class CATrainer {
  public:
    ~CATrainer() {
        if (getenv("FR_HOST_TIMING") && host_calls_)
            fprintf(stderr, "[host timing] %llu set-ticks: stage + submit %.1f us (of which staging the groups %.1f), wait + collect %.1f us, replay %.1f us per set and tick\n",
                    (unsigned long long)host_calls_, host_us_[0] / host_calls_, host_us_[3] / host_calls_, host_us_[1] / host_calls_, host_us_[2] / host_calls_);
    }
    // slot / device: which device-side copy of the view this trainer runs on (DatasetView::device_ptr(slot, device);
    // slot 0 = the view's first device form).  train_model gives every device its own trainer.
    // This form trains the fixed range [rbegin, rend), all of it live from the start.
    CATrainer(std::shared_ptr<DatasetView> view, Evaluator ev, const CAParams& p, uint32_t rbegin, uint32_t rend,
              QueryShard shard = QueryShard(), int slot = 0, int device = -1)
        : CATrainer(std::move(view), std::move(ev), p, std::make_shared<RestartQueue>(rbegin, std::min(rend, p.num_restarts)),
                    UINT32_MAX, std::move(shard), slot, device) {}

    // This form keeps at most `capacity` restarts live and takes ids from `queue` (shared with other trainers) -- at the
    // start, and again whenever restarts of its own have converged (refill()).
    CATrainer(std::shared_ptr<DatasetView> view, Evaluator ev, const CAParams& p, std::shared_ptr<RestartQueue> queue,
              uint32_t capacity, QueryShard shard = QueryShard(), int slot = 0, int device = -1)
        : view_(std::move(view)), ev_(std::move(ev)), p_(p), fids_(view_->features), shard_(std::move(shard)), slot_(slot), device_(device),
          queue_(std::move(queue)) {
        if (fids_.empty()) fail_str("assertion failed: data.n_dim() > 0");
        if (view_->instances.empty()) fail_str("assertion failed: !data.instances().is_empty()");
        frdev::DeviceDataset& dev = view_->device(slot_, device_);
        d_ = dev.d();
        model_dim_ = *std::max_element(fids_.begin(), fids_.end()) + 1;  // :93-98
        if (model_dim_ > d_) fail_str("feature id out of range for this dataset");

        Rand64 master(p_.seed);
        child_.resize(p_.num_restarts);
        for (uint32_t r = 0; r < p_.num_restarts; r++) child_[r] = master.rand_u64();  // :211-213
        {
            uint32_t id = 0;
            while (rs_.size() < (size_t)std::max<uint32_t>(capacity, 1) && queue_->pop(&id)) rs_.emplace_back(id, child_[id]);
        }
        refill_min_ = std::max<size_t>(1, rs_.size() / 8);
        bool can_fused = dev.linesearch_supported(ev_.measure, ev_.depth);
        bool can_fullrank = dev.fullrank_supported(ev_.measure, ev_.depth) && !frdev::path_env("FR_FORCE_GENERIC");
        if (shard_.allreduce) {
            // Query shards: what a shard's device form supports depends on that shard's data (a non-finite
            // feature, a query beyond the rank-table size, the number of gain classes), and the vector each
            // tick hands to the exchange is laid out per path (groups*64 sums on the fused paths, one sum per
            // candidate on the generic one).  Every rank must therefore take the same path: the one all of
            // them support.  Counts are small integers, so the rank-ordered sum is exact.
            double caps[3] = {can_fused ? 1.0 : 0.0, can_fullrank ? 1.0 : 0.0, 1.0};
            shard_.allreduce(caps, 3);
            can_fused = caps[0] == caps[2];
            can_fullrank = caps[1] == caps[2];
        }
        fused_ = can_fused;
        fullrank_ = !fused_ && can_fullrank;
        stats_.path = fused_ ? "fused_linesearch" : (fullrank_ ? "fused_fullrank" : "generic_sort");
        stats_.restarts = (uint32_t)rs_.size();
        if (rs_.empty()) return;
        const size_t R = rs_.size();
        std::vector<size_t> all(R);
        for (size_t k = 0; k < R; k++) all[k] = k;
        init_restarts(all);
        // Resident base sums for the bound-and-verify kernel (device.hpp LineGroup): one slot per restart holds
        // R ~ sum_j x_j * best_w_j for every document, so a tick reads 24 bytes per document and restart instead
        // of the whole feature row.  FR_LS_RESIDENT=0 turns it off (every tick then forms the sums from the tiles).
        const char* res_env = frdev::path_env("FR_LS_RESIDENT");
        if ((fused_ || fullrank_) && !(res_env && res_env[0] == '0')) {
            std::string _err;
            res_owner_ = dev.resident_reserve(R, &_err);
            if (res_owner_ != 0) {
                resident_ = true;
                if (const char* e = frdev::path_env("FR_RESIDENT_REFRESH")) res_refresh_ = (uint32_t)std::max(1, atoi(e));
                for (size_t k = 0; k < R; k++) rs_[k].slot = (int)k;
                refresh_resident(all);
            }
        }
    }

    // every restart this trainer holds has converged and the queue has no more for it
    bool done() const {
        for (const Restart& r : rs_)
            if (!r.done) return false;
        return !queue_ || queue_->empty();
    }

    // One lock-step tick.  Returns false when every restart had already converged.
    bool tick() {
        refill();
        frdev::DeviceDataset& dev = view_->device(slot_, device_);
        ExactTickCount etc_(dev, stats_);
        size_t gen_B = 0;
        if (!build_groups(-1, groups_, &gen_B)) return false;
        stats_.line_searches++;
        dev.set_sums_only((bool)shard_.allreduce);  // the dataset object may be shared with other callers
        if (fused_) {
            std::string _err;
            unsigned long long p0 = 0, r0 = 0, p1 = 0, r1 = 0;
            dev.verify_counters(&p0, &r0);
            if (!dev.linesearch_ndcg(ev_.depth, ev_.norms.data(), groups_, &means_, &_err)) fail_str(_err);
            dev.verify_counters(&p1, &r1);
            stats_.verify_pairs += p1 - p0;
            stats_.verify_redone += r1 - r0;
            check_flags(dev);
        } else if (fullrank_) {
            std::string _err;
            unsigned long long p0 = 0, r0 = 0, p1 = 0, r1 = 0;
            dev.verify_counters(&p0, &r0);
            if (!dev.linesearch_fullrank(ev_.measure, ev_.depth, ev_.norms.data(), groups_, &means_, &_err))
                fail_str(_err);
            dev.verify_counters(&p1, &r1);
            stats_.verify_pairs += p1 - p0;
            stats_.verify_redone += r1 - r0;
            check_flags(dev);
        } else {
            evaluate_means_generic(dev, ev_, gen_w_, gen_B, means_);
        }
        global_means(means_);
        stats_.ticks++;
        stats_.groups += (fused_ || fullrank_) ? groups_.size() : gen_B;
        apply_results(-1, means_);
        return true;
    }

    // Up to max_ticks lock-step ticks; *ticks_done = how many happened.  Returns false when every restart had
    // already converged before the last of them.  On the fused NDCG@k path the restarts are stepped as a few
    // sets with one line search of each in flight (DeviceDataset::linesearch_ndcg_submit): while the host
    // replays the accept logic of one set and stages its next tick, the device works on the others.  A
    // restart's trajectory does not depend on what it is batched with, so the results are the same as tick()'s;
    // every submitted line search is collected and applied before this returns.
    bool run(uint64_t max_ticks, uint64_t* ticks_done) {
        uint64_t n = 0;
        bool alive = true;
        {  // how many sets this call keeps in flight (FR_LS_PIPELINE=n; 0 or 1: plain lock step); read per call
            const char* pe = getenv("FR_LS_PIPELINE");
            long want = pe ? atol(pe) : 3;
            want = std::min<long>(want, frdev::DeviceDataset::LINESEARCH_CONTEXTS);
            want = std::min<long>(want, (long)rs_.size());
            const bool fr_resident = fullrank_ && resident_;  // (MRR, NDCG of any depth, MAP: bound-and-verify on resident sums)
            parts_ = ((fused_ || fr_resident) && !shard_.allreduce && want >= 2) ? (int)want : 1;
        }
        if (parts_ < 2) {
            while (n < max_ticks && (alive = tick())) n++;
            if (ticks_done) *ticks_done = n;
            return alive;
        }
        frdev::DeviceDataset& dev = view_->device(slot_, device_);
        ExactTickCount etc_(dev, stats_);
        constexpr int MAXP = frdev::DeviceDataset::LINESEARCH_CONTEXTS;
        // One round = the sets stepped with one line search of each in flight until the tick budget is spent, every
        // restart has converged, or enough restarts have converged for a refill from the queue to be worth draining the
        // pipeline (refill_wanted()); then the next round starts with the new restarts in the freed places.
        while (n < max_ticks) {
            refill();
            const uint64_t budget = max_ticks - n;
            uint64_t steps[MAXP] = {};
            bool inflight[MAXP] = {};
            bool ready[MAXP] = {};  // reciprocal rank: the set was evaluated in lock step (means_h_ already holds the result)
            bool stop = false;
            auto submit = [&](int h) {
                size_t unused = 0;
                HostTimer ht_(host_us_[0]);
                {
                    HostTimer hb_(host_us_[3]);
                    if (stop || steps[h] >= budget || !build_groups(h, groups_h_[h], &unused)) return;
                }
                stats_.line_searches++;
                dev.set_sums_only(false);
                std::string _err;
                ready[h] = false;
                if (fused_) {
                    if (!dev.linesearch_ndcg_submit(h, ev_.depth, ev_.norms.data(), groups_h_[h], &_err)) fail_str(_err);
                } else {
                    bool queued = false;
                    if (!dev.linesearch_fullrank_submit(h, ev_.measure, ev_.depth, ev_.norms.data(), groups_h_[h], &queued, &_err))
                        fail_str(_err);
                    if (!queued) {  // not applicable this tick (the device applied the pending resident updates): exact kernels
                        for (frdev::LineGroup& lg : groups_h_[h]) lg.has_update = false;
                        unsigned long long p0 = 0, r0 = 0, p1 = 0, r1 = 0;
                        dev.verify_counters(&p0, &r0);
                        if (!dev.linesearch_fullrank(ev_.measure, ev_.depth, ev_.norms.data(), groups_h_[h], &means_h_[h], &_err))
                            fail_str(_err);
                        dev.verify_counters(&p1, &r1);
                        stats_.verify_pairs += p1 - p0;
                        stats_.verify_redone += r1 - r0;
                        ready[h] = true;
                    }
                }
                inflight[h] = true;
            };
            auto collect = [&](int h) {
                std::string _err;
                if (ready[h]) return;
                if (fused_) {
                    if (!dev.linesearch_ndcg_collect(h, &means_h_[h], &_err)) fail_str(_err);
                } else {
                    if (!dev.linesearch_fullrank_collect(h, &means_h_[h], &_err)) fail_str(_err);
                }
            };
            auto drain = [&]() {  // an error is on its way out: leave no submitted line search behind
                for (int h = 0; h < parts_; h++)
                    if (inflight[h] && !ready[h]) {
                        std::string _e;
                        std::vector<double> tmp;
                        if (fused_) (void)dev.linesearch_ndcg_collect(h, &tmp, &_e);
                        else (void)dev.linesearch_fullrank_collect(h, &tmp, &_e);
                        inflight[h] = false;
                    }
            };
            try {
                for (int h = 0; h < parts_; h++) submit(h);
                for (bool any = true; any;) {
                    any = false;
                    for (int h = 0; h < parts_; h++) {
                        if (!inflight[h]) continue;
                        any = true;
                        unsigned long long p0 = 0, r0 = 0, p1 = 0, r1 = 0;
                        dev.verify_counters(&p0, &r0);
                        inflight[h] = false;
                        {
                            HostTimer ht_(host_us_[1]);
                            collect(h);
                        }
                        dev.verify_counters(&p1, &r1);
                        stats_.verify_pairs += p1 - p0;
                        stats_.verify_redone += r1 - r0;
                        check_flags(dev);
                        stats_.groups += groups_h_[h].size();
                        {
                            HostTimer ht_(host_us_[2]);
                            apply_results(h, means_h_[h]);
                        }
                        host_calls_++;
                        steps[h]++;
                        if (newly_done_ > 0) done_wait_++;
                        if (refill_wanted()) stop = true;
                        submit(h);
                    }
                }
            } catch (...) {
                drain();
                throw;
            }
            uint64_t round = 0;
            for (int h = 0; h < parts_; h++) round = std::max(round, steps[h]);
            n += round;
            stats_.ticks += round;
            if (round == 0) break;  // nothing live, nothing left to start
        }
        if (ticks_done) *ticks_done = n;
        return n == max_ticks;
    }

    // Restarts that have converged hand their place (and resident slot) to the next ids of the queue: their results are
    // kept, the new restarts get their initial weights, first evaluation and exact resident sums (init_restarts).
    // Called with nothing in flight on the device.  Returns how many restarts were started.
    size_t refill() {
        newly_done_ = 0;
        done_wait_ = 0;
        if (!queue_ || queue_->empty()) return 0;
        std::vector<size_t> fresh;
        for (size_t k = 0; k < rs_.size(); k++) {
            if (!rs_[k].done) continue;
            uint32_t id = 0;
            if (!queue_->pop(&id)) break;
            finished_.push_back(result_of(rs_[k]));
            const int slot = rs_[k].slot;
            rs_[k] = Restart(id, child_[id]);
            rs_[k].slot = slot;
            fresh.push_back(k);
        }
        if (fresh.empty()) return 0;
        stats_.restarts += (uint32_t)fresh.size();
        stats_.refills++;
        view_->device(slot_, device_).set_sums_only((bool)shard_.allreduce);
        init_restarts(fresh);
        if (resident_) refresh_resident(fresh);
        return fresh.size();
    }

  private:
    // part -1: every restart; otherwise the part-th of run()'s parts_ contiguous sets
    bool in_part(size_t k, int part) const {
        if (part < 0) return true;
        const size_t R = rs_.size(), P = (size_t)parts_;
        return k >= R * (size_t)part / P && k < R * ((size_t)part + 1) / P;
    }

    // Stages the next line search of every live restart of the half: shuffles at the start of a pass
    // (coordinate_ascent.rs:113-116), normalises (:131-143), lists the candidates (:145-171).  Returns false when
    // the half has no live restart.
    bool build_groups(int part, std::vector<frdev::LineGroup>& groups, size_t* gen_B_out) {
        groups.clear();
        gen_w_.clear();
        size_t gen_B = 0;
        bool any = false;
        if (resident_) {
            std::vector<size_t> stale;
            for (size_t k = 0; k < rs_.size(); k++)
                if (in_part(k, part) && !rs_[k].done && rs_[k].res_updates >= res_refresh_) stale.push_back(k);
            if (!stale.empty()) refresh_resident(stale);
        }
        for (size_t k = 0; k < rs_.size(); k++) {
            Restart& r = rs_[k];
            if (r.done || !in_part(k, part)) continue;
            any = true;
            if (r.pos == 0 && r.order.empty()) {
                r.order = fids_;
                shuffle(r.order, r.rand);  // :113-116
                r.successes = 0;
                if (!p_.quiet)
                    printf("[restart %u] shuffle features and optimize (%s=%.6f)\n", r.id, ev_.name.c_str(), r.best_score);
            }
            uint32_t f = r.order[r.pos];
            r.start_score = r.best_score;
            r.base = r.best_w;
            r.norm = 1.0;
            if (p_.normalize) {  // l1_normalize (entries >= model_dim are 0 and stay 0), keeping the divisor
                double sum = 0.0;
                for (double x : r.base) sum += std::fabs(x);
                if (sum > 0.0) {
                    for (double& x : r.base) x /= sum;
                    r.norm = sum;
                }
            }
            double orig = r.base[f];
            line_candidates(orig, p_, r.cands, r.block_len);
            if (fused_ || fullrank_) {
                r.first_group = groups.size();
                for (size_t c0 = 0; c0 < r.cands.size(); c0 += 64) {
                    frdev::LineGroup lg;
                    lg.feature = f;
                    lg.weights = r.base;
                    lg.candidates.assign(r.cands.begin() + c0, r.cands.begin() + std::min(r.cands.size(), c0 + 64));
                    if (resident_) {
                        lg.resident_slot = r.slot;
                        lg.resident_owner = res_owner_;
                        lg.resident_norm = r.norm;
                        lg.resident_base_f = orig;
                        lg.resident_err = r.res_err;
                        lg.has_update = r.pend;
                        lg.upd_feature = r.pend_f;
                        lg.upd_norm = r.pend_norm;
                        lg.upd_base_f = r.pend_base_f;
                        lg.upd_cand = r.pend_cand;
                    }
                    groups.push_back(std::move(lg));
                }
                // the device applies the pending update of this restart with the line search staged here (the verify
                // kernel, or resident_update_kernel when the exact kernels run instead)
                if (resident_) r.pend = false;
            } else {
                r.first_group = gen_B;
                for (double cw : r.cands) {
                    size_t off = gen_w_.size();
                    gen_w_.insert(gen_w_.end(), r.base.begin(), r.base.end());
                    gen_w_[off + f] = cw;
                    gen_B++;
                }
            }
        }
        *gen_B_out = gen_B;
        return any;
    }

    // Replays coordinate_ascent.rs:145-186 over the batched results of the half's line searches.
    void apply_results(int part, const std::vector<double>& means) {
        frdev::DeviceDataset& dev = view_->device(slot_, device_);
        for (size_t k = 0; k < rs_.size(); k++) {
            Restart& r = rs_[k];
            if (r.done || !in_part(k, part)) continue;
            uint32_t f = r.order[r.pos];
            size_t c = 0;
            long accepted = -1;  // index of the last accepted candidate of this line search
            for (int s = 0; s < 3; s++) {
                for (uint32_t it = 0; it < r.block_len[s]; it++, c++) {
                    double sc = (fused_ || fullrank_) ? means[(r.first_group + c / 64) * 64 + (c % 64)]
                                                      : means[r.first_group + c];
                    stats_.useful_evals++;
                    if (sc == sc && sc > r.best_score) {  // core.rs:57-66: NaN rejected, strict >
                        r.best_score = sc;
                        r.best_w = r.base;
                        r.best_w[f] = r.cands[c];
                        accepted = (long)c;
                        if (!p_.quiet)
                            printf("%4u|%-16s|%9.3f|%9.3f\n", r.id, view_->core->feature_name(f).c_str(), r.cands[c], sc);
                    }
                }
                if (r.best_score - r.start_score > p_.tolerance) break;  // :174
            }
            stats_.raw_evals += r.cands.size();
            if (resident_ && accepted >= 0) {
                // best_w became base with [f] = cand: the resident sum follows on the device at the next visit,
                //   R' = fma(x_f, cand, fma(-x_f, base_f, R * (1/norm)))  ~  sum_j x_j * best_w'_j,
                // and its error bound follows here: the old one shrinks by 1/norm, the reciprocal, the two products
                // and the two roundings add at most 8 u * T (T = sum_j |x_j best_w'_j| + |x_f base_f|, via column maxima)
                const std::vector<double>& X = dev.column_absmax();
                double T = std::fabs(r.base[f]) * X[f];
                for (size_t j = 0; j < d_; j++) T += std::fabs(r.best_w[j]) * X[j];
                r.res_err = frdev::resident_err_update(r.res_err, r.norm, T);
                r.res_updates++;
                r.pend = true;
                r.pend_f = f;
                r.pend_norm = r.norm;
                r.pend_base_f = r.base[f];
                r.pend_cand = r.cands[(size_t)accepted];
            }
            if (r.best_score - r.start_score > p_.tolerance) r.successes++;  // :179-182
            r.pos++;
            if (r.pos == r.order.size()) {
                if (r.successes == 0) {
                    r.done = true;  // :185
                    newly_done_++;
                } else {
                    r.pos = 0;
                    r.order.clear();
                }
            }
        }
    }

  public:
    // every restart this trainer has run or is running, in restart order
    std::vector<RestartResult> results() const {
        std::vector<RestartResult> out = finished_;
        for (const Restart& r : rs_) out.push_back(result_of(r));
        std::sort(out.begin(), out.end(), [](const RestartResult& a, const RestartResult& b) { return a.restart_id < b.restart_id; });
        return out;
    }

    TrainStats& stats() { return stats_; }
    const CAParams& params() const { return p_; }

  private:
    // initial weights + initial evaluate_mean (:104-111) of the restarts rs_[which[..]]
    void init_restarts(const std::vector<size_t>& which) {
        frdev::DeviceDataset& dev = view_->device(slot_, device_);
        const size_t R = which.size();
        std::vector<double> w0(R * d_, 0.0);
        for (size_t i = 0; i < R; i++) {
            Restart& r = rs_[which[i]];
            r.best_w.assign(d_, 0.0);
            if (p_.init_random) {
                for (uint32_t f : fids_) r.best_w[f] = r.rand.rand_float() * 2.0 - 1.0;  // :50-54
            } else {
                for (uint32_t f : fids_) r.best_w[f] = 1.0 / (double)fids_.size();  // :60-70
            }
            std::copy(r.best_w.begin(), r.best_w.end(), w0.begin() + i * d_);
        }
        std::vector<double> means;
        dev.set_sums_only((bool)shard_.allreduce);
        evaluate_means_generic(dev, ev_, w0, R, means);
        global_means(means);
        for (size_t i = 0; i < R; i++) {
            if (means[i] != means[i]) fail_str("NaN found!");  // core.rs:50-55 Scored::new
            rs_[which[i]].best_score = means[i];
            stats_.useful_evals++;
            stats_.raw_evals++;
        }
    }
    // enough restarts have converged since the last refill (or a few have been waiting for 48 line searches), and the
    // queue still holds ids
    bool refill_wanted() const { return (newly_done_ >= refill_min_ || done_wait_ >= 48) && queue_ && !queue_->empty(); }

    // Exact resident sums for the given restarts: score_linear (ordered f64 sums of best_w) -> slot.
    void refresh_resident(const std::vector<size_t>& which) {
        frdev::DeviceDataset& dev = view_->device(slot_, device_);
        const std::vector<double>& X = dev.column_absmax();
        const size_t ld = (dev.n() + 63) / 64 * 64;
        const size_t chunk = std::max<size_t>(1, std::min<size_t>(512, (size_t(2) << 30) / (ld * sizeof(double))));
        std::vector<double> w;
        for (size_t b0 = 0; b0 < which.size(); b0 += chunk) {
            const size_t bn = std::min(chunk, which.size() - b0);
            w.assign(bn * d_, 0.0);
            for (size_t k = 0; k < bn; k++) std::copy(rs_[which[b0 + k]].best_w.begin(), rs_[which[b0 + k]].best_w.end(), w.begin() + k * d_);
            std::string _err;
            if (!dev.score_linear(bn, w.data(), &_err)) fail_str(_err);
            for (size_t k = 0; k < bn; k++) {
                Restart& r = rs_[which[b0 + k]];
                if (!dev.resident_store_from_scores(res_owner_, (size_t)r.slot, k, &_err)) {
                    if (!_err.empty()) fail_str(_err);
                    resident_ = false;  // another trainer took the buffers over: form the sums from the tiles from now on
                    return;
                }
                double T = 0.0;
                for (size_t j = 0; j < d_; j++) T += std::fabs(r.best_w[j]) * X[j];
                r.res_err = frdev::resident_err_refresh((uint32_t)d_, T);  // gamma_D * T of an ordered sum
                r.res_updates = 0;
                r.pend = false;
            }
        }
    }
    // query-sharded mode: `v` holds this shard's sums; make them global means
    void global_means(std::vector<double>& v) {
        if (!shard_.allreduce) return;
        shard_.allreduce(v.data(), v.size());
        const double q = (double)shard_.total_queries;
        for (double& x : v) x = shard_.total_queries ? x / q : 0.0;
    }
    struct Restart {
        uint32_t id;
        Rand64 rand;
        std::vector<double> best_w;  // length d (entries >= model_dim stay 0)
        double best_score = 0.0;
        std::vector<uint32_t> order;
        size_t pos = 0;
        size_t successes = 0;
        bool done = false;
        // current line search
        std::vector<double> base;
        std::vector<double> cands;
        uint32_t block_len[3] = {0, 0, 0};
        size_t first_group = 0;
        double start_score = 0.0;
        // resident base sum of this restart on the device (see the constructor)
        int slot = -1;
        double norm = 1.0;         // l1 norm the current base was divided by
        double res_err = 0.0;      // bound on |R - sum_j x_j best_w_j| over all documents (incl. a pending update)
        uint32_t res_updates = 0;  // incremental updates since the last exact refresh
        bool pend = false;         // best_w changed last tick: the device still has to update R
        uint32_t pend_f = 0;
        double pend_norm = 1.0, pend_base_f = 0.0, pend_cand = 0.0;
        Restart(uint32_t i, uint64_t seed) : id(i), rand(seed) {}
    };
    std::shared_ptr<DatasetView> view_;
    Evaluator ev_;
    CAParams p_;
    std::vector<uint32_t> fids_;
    QueryShard shard_;
    size_t d_ = 0;
    uint32_t model_dim_ = 0;
    bool fused_ = false;
    bool fullrank_ = false;
    int slot_ = 0, device_ = -1;
    bool resident_ = false;
    uint64_t res_owner_ = 0;
    uint32_t res_refresh_ = 256;  // incremental updates of a resident sum between exact refreshes
    RestartResult result_of(const Restart& r) const {
        RestartResult rr;
        rr.restart_id = r.id;
        rr.score = r.best_score;
        rr.weights.assign(r.best_w.begin(), r.best_w.begin() + model_dim_);
        return rr;
    }
    std::vector<Restart> rs_;
    std::shared_ptr<RestartQueue> queue_;
    std::vector<uint64_t> child_;           // child seeds of all num_restarts restarts, drawn in order (:211-213)
    std::vector<RestartResult> finished_;   // restarts whose place was handed to a later one
    size_t newly_done_ = 0, refill_min_ = 1, done_wait_ = 0;
    double host_us_[4] = {0.0, 0.0, 0.0, 0.0};  // stage + submit / wait + collect / replay (FR_HOST_TIMING)
    unsigned long long host_calls_ = 0;
    std::vector<frdev::LineGroup> groups_, groups_h_[frdev::DeviceDataset::LINESEARCH_CONTEXTS];
    std::vector<double> means_h_[frdev::DeviceDataset::LINESEARCH_CONTEXTS];
    int parts_ = 1;  // sets of restarts run() keeps in flight
    std::vector<double> gen_w_, means_;
    TrainStats stats_;
};

// coordinate_ascent.rs:232-252 selection over a restart-ordered history
inline Model ca_select(const std::vector<RestartResult>& hist, bool output_ensemble) {
    if (hist.empty()) fail_str("Should be at least 1 restart!");
    if (output_ensemble && hist.size() > 1) {
        Model ens;
        ens.kind = Model::Ensemble;
        for (const auto& h : hist) {
            Model lin;
            lin.kind = Model::Linear;
            lin.weights = h.weights;
            l1_normalize(lin.weights);
            ens.ens_weights.push_back(h.score);
            ens.members.push_back(std::move(lin));
        }
        return ens;
    }
    size_t best = 0;
    for (size_t k = 0; k < hist.size(); k++)
        if (hist[k].score >= hist[best].score) best = k;  // Iterator::max = last maximum
    Model lin;
    lin.kind = Model::Linear;
    lin.weights = hist[best].weights;
    return lin;
}

}  // namespace fr
