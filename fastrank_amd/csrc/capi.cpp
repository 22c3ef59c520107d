// extern "C" boundary of libfastrank_amd.so -- see include/fastrank.h for the contract and the
// reference file:line each symbol replaces (src/lib.rs, src/ffi.rs, src/json_api.rs).
#include <atomic>
#include <limits>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

#include "../../include/fastrank.h"
#include "host.hpp"
#include "loader.hpp"
#include "rf_train.hpp"

using fr::FrError;
using frjson::Value;

struct CDataset {
    std::shared_ptr<fr::DatasetView> view;
};
struct CModel {
    fr::Model actual;
};
struct CQRel {
    fr::QRel actual;
};

namespace {

// Device jobs are serialised per dataset (DataCore::api_mu: the views of one core share device forms and work buffers),
// not per process: two host threads that train on different datasets -- each on its own devices -- do not wait for
// each other.  (Round 2 had one process-wide lock.)
std::mutex g_stats_mu;  // fr_last_train_stats
static std::mutex& api_mu_of(const CDataset& ds) { return ds.view->core->api_mu; }
fr::TrainStats g_last_stats;

// src/ffi.rs:40-43 return_string
const void* return_string(const std::string& s) {
    char* p = (char*)malloc(s.size() + 1);
    memcpy(p, s.c_str(), s.size() + 1);
    return p;
}

std::string error_envelope(const std::string& debug_context) {
    Value o = Value::object();
    o.set("error", Value::string("error"));
    o.set("context", Value::string(debug_context));
    return frjson::dump(o);
}

// src/ffi.rs:29-37 accept_str
std::string accept_str(const char* name, const void* input) {
    if (!input) fr::fail_str(std::string("NULL pointer: ") + name);
    return std::string((const char*)input);
}

Value parse_json_or_fail(const std::string& text) {
    try {
        return frjson::parse(text.c_str());
    } catch (const frjson::ParseError& e) {
        fr::fail_raw(e.debug());
    }
}

// src/ffi.rs:45-55 result_to_json
template <typename F>
const void* json_call(F&& body) {
    std::string out;
    try {
        out = body();
    } catch (const FrError& e) {
        out = error_envelope(e.debug);
    } catch (const std::exception& e) {
        out = error_envelope(frjson::rust_debug_str(e.what()));
    }
    return return_string(out);
}

// src/ffi.rs:57-74 result_to_c
template <typename T, typename F>
const CResult* c_call(F&& body) {
    CResult* r = (CResult*)malloc(sizeof(CResult));
    r->error_message = nullptr;
    r->success = nullptr;
    try {
        T* item = body();
        r->success = item;
    } catch (const FrError& e) {
        r->error_message = return_string(error_envelope(e.debug));
    } catch (const std::exception& e) {
        r->error_message = return_string(error_envelope(frjson::rust_debug_str(e.what())));
    }
    return r;
}

// NULL on success, else envelope string
template <typename F>
const void* status_call(F&& body) {
    try {
        body();
        return nullptr;
    } catch (const FrError& e) {
        return return_string(error_envelope(e.debug));
    } catch (const std::exception& e) {
        return return_string(error_envelope(frjson::rust_debug_str(e.what())));
    }
}

const CDataset& require_dataset(const void* p) {
    if (!p) fr::fail_str("Dataset pointer is null!");
    return *(const CDataset*)p;
}
const CModel& require_model(const void* p) {
    if (!p) fr::fail_str("Model pointer is null!");
    return *(const CModel*)p;
}

Value unknown_cmd(const char* kind, const std::string& cmd) {
    Value o = Value::object();
    o.set("error", Value::string(kind));
    o.set("context", Value::string(cmd));
    return o;
}

// instances_by_query of a view: qid -> instance ids in view order (ascending for unsampled data)
std::vector<std::pair<std::string, std::vector<uint32_t>>> instances_by_query(const fr::DatasetView& v) {
    std::vector<std::pair<std::string, std::vector<uint32_t>>> out;
    std::unordered_map<uint32_t, size_t> slot;
    for (uint32_t id : v.instances) {
        uint32_t qi = v.core->qix[id];
        auto it = slot.find(qi);
        if (it == slot.end()) {
            it = slot.emplace(qi, out.size()).first;
            out.emplace_back(v.core->qnames[qi], std::vector<uint32_t>());
        }
        out[it->second].second.push_back(id);
    }
    return out;
}

Value rccl_report_to_json(const frdev::RcclReport& r) {
    Value o = Value::object();
    o.set("ran", Value::boolean(r.ran));
    o.set("ranks", Value::uint((uint64_t)r.ranks));
    Value d = Value::array();
    for (int x : r.devices) d.push(Value::uint((uint64_t)x));
    o.set("devices", std::move(d));
    o.set("block_doubles", Value::uint((uint64_t)r.block_doubles));
    if (r.ran) {
        o.set("us", Value::number(r.us));
        o.set("first_us", Value::number(r.first_us));
        o.set("init_us", Value::number(r.init_us));
        o.set("matches_host_gather", Value::boolean(r.matches_host_gather));
        o.set("library", Value::string(r.library));
    } else {
        o.set("reason", Value::string(r.reason));
    }
    return o;
}

Value stats_to_json(const fr::TrainStats& s) {
    Value o = Value::object();
    if (s.rccl_set) o.set("rccl", rccl_report_to_json(s.rccl));
    o.set("useful_evals", Value::uint(s.useful_evals));
    o.set("raw_evals", Value::uint(s.raw_evals));
    o.set("ticks", Value::uint(s.ticks));
    o.set("groups", Value::uint(s.groups));
    o.set("seconds", Value::number(s.seconds));
    o.set("path", Value::string(s.path));
    o.set("restarts", Value::uint(s.restarts));
    o.set("verify_pairs", Value::uint(s.verify_pairs));
    o.set("verify_redone", Value::uint(s.verify_redone));
    o.set("exact_ticks", Value::uint(s.exact_ticks));
    o.set("exact_groups", Value::uint(s.exact_groups));
    o.set("verify_redo_entries", Value::uint(s.verify_redo_entries));
    o.set("chain_runs", Value::uint(s.chain_runs));
    o.set("chain_visits", Value::uint(s.chain_visits));
    o.set("rank_slots_on", Value::uint(s.rank_slots_on));
    o.set("rank_slots_off", Value::uint(s.rank_slots_off));
    o.set("line_searches", Value::uint(s.line_searches));
    o.set("audit_values", Value::uint(s.audit_values));
    o.set("audit_mismatches", Value::uint(s.audit_mismatches));
    o.set("devices", Value::uint(s.devices));
    o.set("refills", Value::uint(s.refills));
    if (s.device >= 0) o.set("device", Value::uint((uint64_t)s.device));
    return o;
}

struct ParsedRequest {
    std::string measure;
    bool is_ca = false;
    fr::CAParams ca;
    fr::RFParams rf;
    bool has_qrel = false;
    fr::QRel qrel;
};

// src/json_api.rs:13-34 TrainRequest
ParsedRequest parse_train_request(const std::string& text) {
    Value v = parse_json_or_fail(text);
    if (!v.is_object()) fr::fail_raw("Error(\"invalid type: expected struct TrainRequest\", line: 1, column: 1)");
    ParsedRequest rq;
    const Value& m = fr::json_field(v, "measure");
    if (!m.is_string()) fr::fail_raw("Error(\"invalid type: expected a string for measure\", line: 1, column: 1)");
    rq.measure = m.s;
    const auto& var = fr::json_variant(fr::json_field(v, "params"), "FastRankModelParams");
    if (var.first == "CoordinateAscent") {
        rq.is_ca = true;
        rq.ca = fr::CAParams::from_json(var.second);
    } else if (var.first == "RandomForest") {
        rq.is_ca = false;
        rq.rf = fr::RFParams::from_json(var.second);
    } else {
        fr::fail_raw("Error(\"unknown variant `" + var.first +
                     "`, expected `CoordinateAscent` or `RandomForest`\", line: 1, column: 1)");
    }
    const Value* j = v.find("judgments");
    if (j && !j->is_null()) {
        rq.has_qrel = true;
        rq.qrel = fr::qrel_from_json(*j);
    }
    return rq;
}

Value random_forest_defaults_json() {
    // src/random_forest.rs:141-157 (+ :14-20 tuple-variant wire form {"SquaredError":[]})
    fr::Rand64 rand(0xdeadbeefULL);
    Value p = Value::object();
    p.set("seed", Value::uint(rand.rand_u64()));
    p.set("quiet", Value::boolean(false));
    p.set("num_trees", Value::uint(100));
    p.set("weight_trees", Value::boolean(false));
    Value sm = Value::object();
    sm.set("SquaredError", Value::array());
    p.set("split_method", std::move(sm));
    p.set("instance_sampling_rate", Value::number(0.5));
    p.set("feature_sampling_rate", Value::number(0.25));
    p.set("min_leaf_support", Value::uint(10));
    p.set("split_candidates", Value::uint(3));
    p.set("max_depth", Value::uint(8));
    return p;
}

// Rust's Display for f64: shortest round-trip digits, never exponent form
std::string rust_display_f64(double v) {
    if (v != v) return "NaN";
    if (std::isinf(v)) return v > 0 ? "inf" : "-inf";
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof(buf), v, std::chars_format::scientific);
    std::string sci(buf, r.ptr);
    bool neg = sci[0] == '-';
    if (neg) sci.erase(0, 1);
    size_t epos = sci.find('e');
    std::string digits;
    for (char c : sci.substr(0, epos))
        if (c != '.') digits += c;
    int exp10 = atoi(sci.c_str() + epos + 1);
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    std::string out = neg ? "-" : "";
    int kk = exp10 + 1;
    if (digits == "0") return out + "0";
    if (kk <= 0) {
        out += "0.";
        out.append((size_t)(-kk), '0');
        out += digits;
    } else if ((int)digits.size() <= kk) {
        out += digits;
        out.append((size_t)(kk - (int)digits.size()), '0');
    } else {
        out.append(digits, 0, (size_t)kk);
        out += '.';
        out.append(digits, (size_t)kk, std::string::npos);
    }
    return out;
}

std::vector<fr::TrainStats> g_last_per_device;  // one entry per device of the last multi-device train_model call
static void set_last_stats(const fr::TrainStats& st, std::vector<fr::TrainStats> per_device = {}) {
    std::lock_guard<std::mutex> lk(g_stats_mu);
    g_last_stats = st;
    g_last_per_device = std::move(per_device);
}

// Per-trainer bound on the restarts kept live at once (FR_RESTART_SLOTS, default 64): a request with more restarts than
// that per device runs them through the restart queue, converged restarts handing their places to the next ids -- the
// resident sums, result matrices and launch width stay those of 64 restarts however many the request names.
static uint32_t restart_slots_max() {
    const char* e = std::getenv("FR_RESTART_SLOTS");
    const long v = e ? std::atol(e) : 64;
    return (uint32_t)std::min<long>(std::max<long>(v, 1), 1 << 20);
}

// the restarts `queue` still holds (at most `capacity` live at a time) on one device-side copy of the view (slot, device:
// DatasetView::device_ptr)
static std::vector<fr::RestartResult> train_ca_queue(const std::shared_ptr<fr::DatasetView>& view, const ParsedRequest& rq,
                                                     const fr::Evaluator& ev, std::shared_ptr<fr::RestartQueue> queue, uint32_t capacity,
                                                     int slot, int device, fr::TrainStats* stats_out) {
    auto t0 = std::chrono::steady_clock::now();
    fr::CATrainer trainer(view, ev, rq.ca, std::move(queue), capacity, fr::QueryShard(), slot, device);
    while (trainer.run(64, nullptr)) {
    }
    trainer.stats().seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    trainer.stats().device = device;
    if (stats_out) *stats_out = trainer.stats();
    return trainer.results();
}

fr::Model train_ca(const std::shared_ptr<fr::DatasetView>& view, const ParsedRequest& rq, uint32_t rbegin,
                   uint32_t rend, std::vector<fr::RestartResult>* hist_out) {
    fr::Evaluator ev = fr::make_evaluator(*view, rq.measure, rq.has_qrel ? &rq.qrel : nullptr);
    if (view->host_csr().nq == 0) fr::fail_str("assertion failed: !data.queries().is_empty()");
    fr::TrainStats st;
    rend = std::min(rend, rq.ca.num_restarts);
    std::vector<fr::RestartResult> hist =
        train_ca_queue(view, rq, ev, std::make_shared<fr::RestartQueue>(rbegin, rend), restart_slots_max(), 0, -1, &st);
    set_last_stats(st);
    fr::Model m;
    if (hist_out) {
        *hist_out = hist;
    } else {
        m = fr::ca_select(hist, rq.ca.output_ensemble);
    }
    return m;
}

// ---- train_model over the devices of a node -------------------------------------------------------------------------
// The reference fans a request's restarts out over the host's cores inside the call (rayon, src/coordinate_ascent.rs:
// 215-225).  Here the fan-out is over GPUs: FR_DEVICES (a comma list of ordinals; the same ordinal twice = two contexts
// on one device) or, if unset, the device chosen with fr_set_device or else every visible device.  Restart ids are
// block-partitioned (the child seeds are drawn in order from the master generator by every trainer, :211-213), each
// device gets its own host thread, trainer and device-side copy of the dataset -- made device to device from the first
// one (DeviceDataset::replicate, hipMemcpyPeer over xGMI) -- and the results are gathered in host memory and selected
// like a single trainer's history (last maximum, or the score-weighted ensemble: :232-251).  No collective: this is
// the in-process form of native.train_model_distributed.
static std::atomic<int> g_pinned_device{-1};  // fr_set_device: an explicit choice of ONE device for this process

static std::vector<int> parse_device_list(const char* e, int count) {
    std::vector<int> devs;
    const char* p = e;
    while (*p) {
        while (*p == ',' || *p == ' ') p++;
        if (!*p) break;
        char* end = nullptr;
        const long v = std::strtol(p, &end, 10);
        if (end == p) fr::fail_str(std::string("FR_DEVICES: not a list of device ordinals: ") + e);
        if (v < 0 || v >= count) fr::fail_str("FR_DEVICES: no device " + std::to_string(v) + " (" + std::to_string(count) + " visible)");
        devs.push_back((int)v);
        p = end;
    }
    if (devs.empty()) fr::fail_str("FR_DEVICES is empty");
    return devs;
}

std::vector<int> fr_train_devices() {
    const int count = frdev::device_count(nullptr);
    if (const char* e = std::getenv("FR_DEVICES")) return parse_device_list(e, count);
    if (const int pinned = g_pinned_device.load(); pinned >= 0) return {pinned};
    std::vector<int> devs;
    for (int d = 0; d < count; d++) devs.push_back(d);
    return devs;
}

// contiguous block partition of restart ids (first parts take the remainder), as native.shard_bounds
static void restart_block(uint32_t R, uint32_t part, uint32_t parts, uint32_t* b, uint32_t* e) {
    const uint32_t base = R / parts, rem = R % parts;
    *b = part * base + std::min(part, rem);
    *e = *b + base + (part < rem ? 1u : 0u);
}

// which entry of the device list trains which restarts, and on which device-side copy (slot 0 = the copy that exists
// on `primary_dev`, if that device is listed)
struct DevicePlan {
    std::vector<int> devs, slot;
    std::vector<uint32_t> begin, end;
};
static DevicePlan plan_devices(std::vector<int> devs, uint32_t R, int primary_dev) {
    DevicePlan pl;
    if (devs.size() > R) devs.resize(std::max<uint32_t>(R, 1));
    const size_t k = devs.size();
    pl.devs = devs;
    pl.slot.assign(k, -1);
    pl.begin.assign(k, 0);
    pl.end.assign(k, 0);
    int next_slot = 1;
    bool primary_used = false;
    for (size_t i = 0; i < k; i++) {
        if (!primary_used && devs[i] == primary_dev) pl.slot[i] = 0, primary_used = true;
        else pl.slot[i] = next_slot++;
        restart_block(R, (uint32_t)i, (uint32_t)k, &pl.begin[i], &pl.end[i]);
    }
    return pl;
}

// Fewest restarts a device must get before the DEFAULT device list (every visible device) adds it to a request
// (FR_MIN_RESTARTS_PER_DEVICE, default 4; an explicit FR_DEVICES is taken as given).  Measured on one MI355X at the 30K
// shape (tools/tick_sweep.py, profiles/r04_tick_vs_groups.json): a tick of G line groups takes 0.11 + 0.083 G ms
// (0.21 / 0.28 / 0.34 / 0.43 / 0.74 / 2.75 ms at G = 1 / 2 / 3 / 4 / 8 / 32), i.e. a device with one restart delivers 40 %
// of the evaluations per second it delivers with 32, with four 80 %; every further device also costs a device-to-device
// copy of the dataset the first time (2.2 GB) and the runtime's start-up there.  With the floor at 4 the reference's
// default request (5 restarts, coordinate_ascent.rs:29) stays on one GPU and 32 restarts spread over 8.
static uint32_t min_restarts_per_device() {
    const char* e = std::getenv("FR_MIN_RESTARTS_PER_DEVICE");
    const long v = e ? std::atol(e) : 4;
    return (uint32_t)std::min<long>(std::max<long>(v, 1), 1 << 20);
}

// The device list of one train_model call over `units` independent pieces of work (restarts, trees), with the view's
// device-side copies made: an empty / one-entry plan means "train on one device" (slot 0 = the view's first device
// form; slot 1 = a copy on the one device an explicit list names when the first form lives elsewhere).
// slots_max (coordinate ascent): the restarts one trainer keeps live; the list is cut to the entries that get restarts when
// every trainer takes its share from the one queue at the start (5 restarts over 4 devices are 2 + 2 + 1, the
// fourth entry would have made a dataset copy and an empty trainer).
static DevicePlan devices_for_request(const std::shared_ptr<fr::DatasetView>& view, uint32_t units, uint32_t min_units_per_device = 1,
                                      uint32_t slots_max = 0) {
    std::vector<int> devs = fr_train_devices();
    const bool explicit_list = std::getenv("FR_DEVICES") != nullptr;
    if (!explicit_list) {
        // a small matrix is not worth a context on every GPU (each costs a device-to-device copy and, the first time, the
        // runtime's per-device start-up), and neither is a device that would get fewer than the floor of units
        if (view->instances.size() * (size_t)view->core->d < (size_t(1) << 23)) devs.resize(std::min<size_t>(devs.size(), 1));
        const size_t by_floor = std::max<size_t>(1, units / std::max<uint32_t>(min_units_per_device, 1));
        devs.resize(std::min(devs.size(), by_floor));
    }
    if (devs.size() > units) devs.resize(std::max<uint32_t>(units, 1));
    if (slots_max > 0 && devs.size() > 1) {
        const uint32_t cap = std::max<uint32_t>(1, std::min<uint32_t>(slots_max, (uint32_t)((units + devs.size() - 1) / devs.size())));
        devs.resize(std::min<size_t>(devs.size(), (units + cap - 1) / cap));
    }
    auto single = [&](const std::vector<int>& d, bool pin) {
        if (d.empty()) return plan_devices(d, units, -1);
        if (pin) {
            std::string err;
            if (!frdev::set_device(d[0], &err)) fr::fail_str(err);
        }
        // an explicit choice of a device other than the one the first device form already lives on: train on a copy there
        const int have = view->built_on_device();
        const int primary = (pin && have >= 0 && have != d[0]) ? have : d[0];
        return plan_devices(std::vector<int>(1, d[0]), units, primary);
    };
    if (devs.size() <= 1) return single(devs, explicit_list || g_pinned_device.load() >= 0);
    if (view->host_csr().nq == 0) return single(devs, false);  // (the trainer reports the empty dataset its own way)
    // slot 0 is the device form the view already has (or builds now, on the first listed device); every other entry of
    // the list gets the next slot
    {
        std::string err;
        if (!frdev::set_device(devs[0], &err)) fr::fail_str(err);
    }
    // The copies are made at the same time (one host thread per entry), then the training runs concurrently.
    // A device the dataset cannot be copied to is an error when FR_DEVICES named it; of the default "every visible
    // device" it is simply left out (the request then trains on the devices that took a copy).
    const int primary_dev = view->device_ptr()->device_ordinal();
    for (;;) {
        DevicePlan pl = plan_devices(devs, units, primary_dev);
        std::vector<int> kept;
        std::exception_ptr first_error;
        // every entry's copy at the same time (each on its own device and stream, reading the first device over its own
        // link); the slots are disjoint
        std::vector<std::exception_ptr> errs(pl.devs.size());
        {
            std::vector<std::thread> copiers;
            auto copy_to = [&](size_t i) {
                try {
                    (void)view->device_ptr(pl.slot[i], pl.devs[i]);
                } catch (...) {
                    errs[i] = std::current_exception();
                }
            };
            (void)view->device_ptr();  // (the first copy exists before anything reads it)
            for (size_t i = 1; i < pl.devs.size(); i++) copiers.emplace_back(copy_to, i);
            copy_to(0);
            for (auto& th : copiers) th.join();
        }
        for (size_t i = 0; i < pl.devs.size(); i++) {
            if (!errs[i]) kept.push_back(pl.devs[i]);
            else if (!first_error) first_error = errs[i];
        }
        if (!first_error) return pl;
        if (explicit_list || kept.empty()) std::rethrow_exception(first_error);
        devs = kept;
        if (devs.size() <= 1) return single(devs, true);
    }
}

fr::Model train_ca_devices(const std::shared_ptr<fr::DatasetView>& view, const ParsedRequest& rq) {
    const uint32_t R = rq.ca.num_restarts;
    auto t0 = std::chrono::steady_clock::now();
    const DevicePlan pl = devices_for_request(view, R, min_restarts_per_device(), restart_slots_max());
    if (pl.devs.empty() || (pl.devs.size() == 1 && pl.slot[0] == 0)) return train_ca(view, rq, 0, R, nullptr);
    fr::Evaluator ev = fr::make_evaluator(*view, rq.measure, rq.has_qrel ? &rq.qrel : nullptr);
    if (view->host_csr().nq == 0) fr::fail_str("assertion failed: !data.queries().is_empty()");
    const size_t k = pl.devs.size();
    // One queue of restart ids for all devices (the reference's rayon pool, src/coordinate_ascent.rs:215-225): every
    // trainer starts with its share and, once restarts of its own have converged, takes the next ids nobody has started.
    // FR_RESTART_QUEUE=0: the static block partition of round 3 (entry i trains block i of the ids) -- same model.
    const char* qe = std::getenv("FR_RESTART_QUEUE");
    const bool shared_queue = !(qe && qe[0] == '0');
    const uint32_t capacity = std::min<uint32_t>(restart_slots_max(), (uint32_t)((R + k - 1) / k));
    auto queue = std::make_shared<fr::RestartQueue>(0, R);
    std::vector<std::vector<fr::RestartResult>> parts(k);
    std::vector<fr::TrainStats> stats(k);
    std::vector<std::exception_ptr> errors(k);
    auto work = [&](size_t i) {
        try {
            std::string err;
            if (!frdev::set_device(pl.devs[i], &err)) fr::fail_str(err);
            auto q = shared_queue ? queue : std::make_shared<fr::RestartQueue>(pl.begin[i], pl.end[i]);
            parts[i] = train_ca_queue(view, rq, ev, std::move(q), shared_queue ? capacity : restart_slots_max(), pl.slot[i], pl.devs[i], &stats[i]);
        } catch (...) {
            errors[i] = std::current_exception();
        }
    };
    std::vector<std::thread> pool;
    for (size_t i = 1; i < k; i++) pool.emplace_back(work, i);
    work(0);
    for (auto& th : pool) th.join();
    for (size_t i = 0; i < k; i++)
        if (errors[i]) std::rethrow_exception(errors[i]);
    // ---- the job's one exchange.  Every entry of the device list contributes its restarts as a block of records; the
    // blocks meet in host memory (the threads share an address space) AND, when the entries are distinct GPUs, in ONE
    // RCCL all-gather across them (frdev::rccl_allgather: ncclCommInitAll over the list, a grouped ncclAllGather).  The
    // selection below reads what RCCL delivered to rank 0, which must equal the host-side concatenation bit for bit; ranks
    // that share a GPU (FR_DEVICES=0,0: contexts, not devices) cannot form a communicator and keep the host gather, with the
    // reason recorded.  With N distinct GPUs a failed exchange fails the request.
    size_t dim = 0, cap = 1;
    for (size_t i = 0; i < k; i++) {
        cap = std::max(cap, parts[i].size());
        for (const auto& h : parts[i]) dim = std::max(dim, h.weights.size());
    }
    const size_t block_len = cap * fr::restart_record_len(dim);
    std::vector<double> blocks(k * block_len);
    for (size_t i = 0; i < k; i++) fr::pack_restart_records(parts[i], cap, dim, blocks.data() + i * block_len);
    frdev::RcclReport rccl;
    std::vector<double> via_rccl;
    {
        std::string err;
        if (!frdev::rccl_allgather(pl.devs, blocks.data(), block_len, &via_rccl, &rccl, &err)) fr::fail_str(err);
    }
    const std::vector<double>& exchanged = rccl.ran ? via_rccl : blocks;
    std::vector<fr::RestartResult> hist = fr::unpack_restart_records(exchanged.data(), k * cap, dim, R);
    fr::TrainStats total = stats[0];
    total.rccl_set = true;
    total.rccl = rccl;
    for (size_t i = 0; i < k; i++) {
        if (i == 0) continue;
        total.useful_evals += stats[i].useful_evals, total.raw_evals += stats[i].raw_evals;
        total.ticks = std::max(total.ticks, stats[i].ticks), total.groups += stats[i].groups;
        total.restarts += stats[i].restarts, total.refills += stats[i].refills;
        total.verify_pairs += stats[i].verify_pairs, total.verify_redone += stats[i].verify_redone;
        total.line_searches += stats[i].line_searches, total.exact_ticks += stats[i].exact_ticks;
        total.exact_groups += stats[i].exact_groups, total.verify_redo_entries += stats[i].verify_redo_entries;
        total.chain_runs += stats[i].chain_runs, total.chain_visits += stats[i].chain_visits;
        total.rank_slots_on += stats[i].rank_slots_on, total.rank_slots_off += stats[i].rank_slots_off;
        total.audit_values += stats[i].audit_values, total.audit_mismatches += stats[i].audit_mismatches;
    }
    total.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    total.devices = (uint32_t)k;
    total.device = -1;
    set_last_stats(total, stats);
    return fr::ca_select(hist, rq.ca.output_ensemble);
}

// json_api.rs:41-48 -> random_forest::learn_ensemble
fr::Model train_rf(const std::shared_ptr<fr::DatasetView>& view, const ParsedRequest& rq) {
    auto t0 = std::chrono::steady_clock::now();
    fr::Evaluator ev = fr::make_evaluator(*view, rq.measure, rq.has_qrel ? &rq.qrel : nullptr);
    fr::RFTrainer trainer(view, std::move(ev), rq.rf);
    // the trees of a forest are independent given their seeds (random_forest.rs:301-331 grows them with rayon): spread
    // over FR_DEVICES like a coordinate-ascent request's restarts, block i of the trees on entry i of the list
    std::vector<fr::RFTrainer::DevicePart> parts;
    if (rq.rf.num_trees > 0) {
        // (a forest that prints its progress table grows on one device -- one unit of work, so no copies are made -- but an
        // explicit FR_DEVICES / fr_set_device choice of that device is still honoured)
        const DevicePlan pl = devices_for_request(view, rq.rf.quiet ? rq.rf.num_trees : 1);
        // several entries, or ONE entry that names a device other than the one the dataset was built on (slot 1: its copy there)
        if (pl.devs.size() > 1 || (pl.devs.size() == 1 && pl.slot[0] != 0))
            for (size_t i = 0; i < pl.devs.size(); i++)
                parts.push_back(fr::RFTrainer::DevicePart{pl.slot[i], pl.devs[i], rq.rf.quiet ? pl.begin[i] : 0u, rq.rf.quiet ? pl.end[i] : rq.rf.num_trees});
    }
    fr::Model m = trainer.learn(std::move(parts));
    fr::TrainStats st;
    st.devices = trainer.stats().devices;
    st.path = "random_forest";
    st.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    st.restarts = trainer.stats().trees;
    st.ticks = trainer.stats().levels;
    st.groups = trainer.stats().batches;
    st.raw_evals = trainer.stats().candidates;  // split candidates evaluated
    st.useful_evals = trainer.stats().nodes;    // tree nodes produced
    set_last_stats(st);
    return m;
}

Value restarts_to_json(const std::vector<fr::RestartResult>& hist) {
    Value arr = Value::array();
    for (const auto& h : hist) {
        Value r = Value::object();
        r.set("restart_id", Value::uint(h.restart_id));
        r.set("score", Value::number(h.score));
        Value w = Value::array();
        for (double x : h.weights) w.push(Value::number(x));
        r.set("weights", std::move(w));
        arr.push(std::move(r));
    }
    return arr;
}

}  // namespace

extern "C" {

void free_str(void* p) { free(p); }
void free_c_result(CResult* p) { free(p); }
void free_dataset(CDataset* p) { delete p; }
void free_model(CModel* p) { delete p; }
void free_cqrel(CQRel* p) { delete p; }

const CResult* load_cqrel(const void* data_path) {
    return c_call<CQRel>([&]() {
        std::string path = accept_str("data_path", data_path);
        auto* q = new CQRel();
        try {
            q->actual = fr::load_qrel_file(path);
        } catch (...) {
            delete q;
            throw;
        }
        return q;
    });
}

const CResult* cqrel_from_json(const void* json_str) {
    return c_call<CQRel>([&]() {
        Value v = parse_json_or_fail(accept_str("json_str", json_str));
        auto* q = new CQRel();
        try {
            q->actual = fr::qrel_from_json(v);
        } catch (...) {
            delete q;
            throw;
        }
        return q;
    });
}

const void* cqrel_query_json(const CQRel* cqrel, const void* query_str) {
    return json_call([&]() {
        if (!cqrel) fr::fail_str("cqrel pointer is null!");
        std::string cmd = accept_str("query_str", query_str);
        if (cmd == "to_json") return frjson::dump(fr::qrel_to_json(cqrel->actual));
        if (cmd == "queries") {
            Value a = Value::array();
            for (const auto& q : cqrel->actual.queries) a.push(Value::string(q.first));
            return frjson::dump(a);
        }
        const auto* docs = cqrel->actual.get(cmd);
        if (!docs) fr::fail_str("Unknown request: " + cmd);
        return frjson::dump(fr::qrel_query_to_json(*docs));
    });
}

const CResult* load_ranksvm_format(void* data_path, void* feature_names_path) {
    return c_call<CDataset>([&]() {
        std::string path = accept_str("data_path", data_path);
        std::map<uint32_t, std::string> names;
        bool has_names = false;
        if (feature_names_path) {
            names = fr::load_feature_names(accept_str("feature_names_path", feature_names_path));
            has_names = true;
        }
        auto* d = new CDataset();
        try {
            d->view = fr::load_ranksvm(path, has_names ? &names : nullptr);
        } catch (...) {
            delete d;
            throw;
        }
        return d;
    });
}

const CResult* dataset_query_sampling(CDataset* dataset, const void* queries_json_list) {
    return c_call<CDataset>([&]() {
        const CDataset& ds = require_dataset(dataset);
        Value v = parse_json_or_fail(accept_str("queries_json_list", queries_json_list));
        if (!v.is_array()) fr::fail_raw("Error(\"invalid type: expected a sequence\", line: 1, column: 1)");
        std::set<std::string> wanted;
        for (const auto& x : v.arr) {
            if (!x.is_string()) fr::fail_raw("Error(\"invalid type: expected a string\", line: 1, column: 1)");
            wanted.insert(x.s);
        }
        // src/sampling.rs:117-133 with_queries
        auto child = std::make_shared<fr::DatasetView>();
        child->core = ds.view->core;
        child->features = ds.view->features;
        child->sampled = true;
        child->parent = ds.view;
        for (auto& qi : instances_by_query(*ds.view))
            if (wanted.count(qi.first)) child->instances.insert(child->instances.end(), qi.second.begin(), qi.second.end());
        auto* out = new CDataset();
        out->view = child;
        return out;
    });
}

const CResult* dataset_feature_sampling(CDataset* dataset, const void* feature_json_list) {
    return c_call<CDataset>([&]() {
        const CDataset& ds = require_dataset(dataset);
        Value v = parse_json_or_fail(accept_str("feature_json_list", feature_json_list));
        if (!v.is_array()) fr::fail_raw("Error(\"invalid type: expected a sequence\", line: 1, column: 1)");
        // src/sampling.rs:91-115 with_features
        std::set<uint32_t> valid(ds.view->features.begin(), ds.view->features.end());
        std::set<uint32_t> keep, missing;
        for (const auto& x : v.arr) {
            uint32_t fid = (uint32_t)fr::json_u64(x, "FeatureId");
            (valid.count(fid) ? keep : missing).insert(fid);
        }
        if (!missing.empty()) {
            std::string s = "Missing Features: {";
            bool first = true;
            for (uint32_t f : missing) {
                if (!first) s += ", ";
                s += "FeatureId(" + std::to_string(f) + ")";
                first = false;
            }
            fr::fail_str(s + "}");
        }
        if (keep.empty()) fr::fail_str("No Features!");
        auto child = std::make_shared<fr::DatasetView>();
        child->core = ds.view->core;
        child->features.assign(keep.begin(), keep.end());
        child->instances = ds.view->instances;
        child->sampled = true;
        child->parent = ds.view;
        child->same_instances_as_parent = true;
        auto* out = new CDataset();
        out->view = child;
        return out;
    });
}

const void* dataset_query_json(void* dataset, void* json_cmd_str) {
    return json_call([&]() {
        const CDataset& ds = require_dataset(dataset);
        const fr::DatasetView& v = *ds.view;
        std::string cmd = accept_str("dataset_query_json", json_cmd_str);
        // src/ffi.rs:144-183
        if (cmd == "is_sampled") return std::string(v.sampled ? "true" : "false");
        if (cmd == "num_features") return std::to_string(v.n_dim());
        if (cmd == "feature_ids") {
            Value a = Value::array();
            for (uint32_t f : v.features) a.push(Value::uint(f));
            return frjson::dump(a);
        }
        if (cmd == "num_instances") return std::to_string(v.instances.size());
        if (cmd == "queries") {
            Value a = Value::array();
            for (auto& qi : instances_by_query(v)) a.push(Value::string(qi.first));
            return frjson::dump(a);
        }
        if (cmd == "instances_by_query") {
            Value o = Value::object();
            for (auto& qi : instances_by_query(v)) {
                Value a = Value::array();
                for (uint32_t id : qi.second) a.push(Value::uint(id));
                o.obj.emplace_back(qi.first, std::move(a));
            }
            return frjson::dump(o);
        }
        if (cmd == "feature_names") {
            Value a = Value::array();
            for (uint32_t f : v.features) a.push(Value::string(v.core->feature_name(f)));
            return frjson::dump(a);
        }
        return frjson::dump(unknown_cmd("unknown_dataset_query_str", cmd));
    });
}

const void* query_json(const void* json_cmd_str) {
    return json_call([&]() {
        std::string cmd = accept_str("query_json_str", json_cmd_str);
        // src/ffi.rs:215-236
        if (cmd == "coordinate_ascent_defaults" || cmd == "random_forest_defaults") {
            Value o = Value::object();
            o.set("measure", Value::string("ndcg"));
            Value params = Value::object();
            if (cmd == "coordinate_ascent_defaults")
                params.set("CoordinateAscent", fr::CAParams::defaults().to_json());
            else
                params.set("RandomForest", random_forest_defaults_json());
            o.set("params", std::move(params));
            o.set("judgments", Value::null());
            return frjson::dump(o);
        }
        return frjson::dump(unknown_cmd("unknown_query_str", cmd));
    });
}

const CResult* make_dense_dataset_f32_f64_i64(size_t n, size_t d, const float* x, const double* y,
                                              const int64_t* qids) {
    return c_call<CDataset>([&]() {
        if ((n && (!x || !y || !qids))) fr::fail_str("NULL pointer: dense dataset arrays");
        auto* out = new CDataset();
        try {
            out->view = fr::make_dense(n, d, x, y, qids);
        } catch (...) {
            delete out;
            throw;
        }
        return out;
    });
}

const CResult* train_model(void* train_request_json, void* dataset) {
    return c_call<CModel>([&]() {
        // src/lib.rs:249-252: the dataset pointer is checked after the request is parsed lazily;
        // result_train_model reports the null dataset first (src/ffi.rs:189-193)
        const CDataset& ds = require_dataset(dataset);
        ParsedRequest rq = parse_train_request(accept_str("train_request_json", train_request_json));
        std::lock_guard<std::mutex> lk(api_mu_of(ds));
        auto* out = new CModel();
        try {
            out->actual = rq.is_ca ? train_ca_devices(ds.view, rq) : train_rf(ds.view, rq);
        } catch (...) {
            delete out;
            throw;
        }
        return out;
    });
}

const CResult* model_from_json(const void* json_str) {
    return c_call<CModel>([&]() {
        Value v = parse_json_or_fail(accept_str("json_str", json_str));
        auto* out = new CModel();
        try {
            out->actual = fr::model_from_json(v);
        } catch (...) {
            delete out;
            throw;
        }
        return out;
    });
}

const void* model_query_json(const void* model, const void* json_cmd_str) {
    return json_call([&]() {
        const CModel& m = require_model(model);
        std::string cmd = accept_str("query_json", json_cmd_str);
        if (cmd == "to_json") return frjson::dump(fr::model_to_json(m.actual));
        return frjson::dump(unknown_cmd("unknown_dataset_query_str", cmd));  // sic: src/ffi.rs:206-209
    });
}

const void* evaluate_by_query(const CModel* model, const CDataset* dataset, const CQRel* qrel,
                              const void* evaluator) {
    return json_call([&]() {
        const CModel& m = require_model(model);
        const CDataset& ds = require_dataset(dataset);
        std::string name = accept_str("evaluator_name", evaluator);
        std::lock_guard<std::mutex> lk(api_mu_of(ds));
        fr::DatasetView& view = *ds.view;
        fr::Evaluator ev = fr::make_evaluator(view, name, qrel ? &qrel->actual : nullptr);
        Value o = Value::object();
        if (view.instances.empty()) return frjson::dump(o);
        frdev::DeviceDataset& dev = view.device();
        fr::score_model(view, m.actual);
        std::string err;
        if (!dev.metric_from_scores(ev.measure, ev.depth, ev.norms.data(), 1, false, &err)) fr::fail_str(err);
        fr::check_flags(dev);
        std::vector<double> vals(dev.nq());
        if (!dev.download_per_query(1, vals.data(), &err)) fr::fail_str(err);
        for (size_t q = 0; q < vals.size(); q++)
            o.obj.emplace_back(view.core->qnames[view.csr_query[q]], Value::number(vals[q]));
        return frjson::dump(o);
    });
}

const void* predict_scores(const CModel* model, const CDataset* dataset) {
    return json_call([&]() {
        const CModel& m = require_model(model);
        const CDataset& ds = require_dataset(dataset);
        std::lock_guard<std::mutex> lk(api_mu_of(ds));
        fr::DatasetView& view = *ds.view;
        if (view.instances.empty()) return std::string("{}");
        frdev::DeviceDataset& dev = view.device();
        fr::score_model(view, m.actual);
        std::vector<double> scores(view.core->n, 0.0);
        std::string err;
        if (!dev.download_scores(0, scores.data(), scores.size(), &err)) fr::fail_str(err);
        for (uint32_t id : view.instances)
            if (scores[id] != scores[id]) fr::fail_str("Model.predict -> NaN");
        // src/json_api.rs:53-72: {"<instance index>": score}
        std::string out = "{";
        bool first = true;
        std::vector<uint32_t> ids = view.instances;
        std::sort(ids.begin(), ids.end());
        for (uint32_t id : ids) {
            if (!first) out += ',';
            first = false;
            out += '"';
            out += std::to_string(id);
            out += "\":";
            frjson::write_double(out, scores[id]);
        }
        out += '}';
        return out;
    });
}

const void* predict_to_trecrun(const CModel* model, const CDataset* dataset, const void* output_path,
                               const void* system_name, size_t depth) {
    return json_call([&]() {
        const CModel& m = require_model(model);
        const CDataset& ds = require_dataset(dataset);
        std::string path = accept_str("output_path", output_path);
        std::string sysname = accept_str("system_name", system_name);
        std::lock_guard<std::mutex> lk(api_mu_of(ds));
        fr::DatasetView& view = *ds.view;
        // src/json_api.rs:75-120
        std::ofstream out(path);
        if (!out) fr::fail_str("could not open " + path + " for writing");
        size_t written = 0;
        if (view.instances.empty()) return std::to_string(written);
        frdev::DeviceDataset& dev = view.device();
        fr::score_model(view, m.actual);
        std::string err;
        std::vector<double> norms(dev.nq(), 0.0);
        if (!dev.metric_from_scores(frdev::M_RR, -1, norms.data(), 1, true, &err)) fr::fail_str(err);
        fr::check_flags(dev);
        std::vector<uint32_t> rank(dev.n());
        if (!dev.download_rank(rank.data(), &err)) fr::fail_str(err);
        std::vector<double> scores(view.core->n, 0.0);
        if (!dev.download_scores(0, scores.data(), scores.size(), &err)) fr::fail_str(err);
        const frdev::HostCSR& csr = view.host_csr();
        const fr::DataCore& c = *view.core;
        for (size_t q = 0; q < csr.nq; q++) {
            const std::string& qid = c.qnames[view.csr_query[q]];
            for (uint32_t p = csr.qoff[q]; p < csr.qoff[q + 1]; p++) {
                size_t rk = p - csr.qoff[q] + 1;
                if (depth > 0 && rk > depth) break;
                uint32_t id = rank[p];
                if (!c.has_docids || id >= c.doc_present.size() || !c.doc_present[id])
                    fr::fail_str("Dataset does not contain document ids and therefore cannot save to trecrun!");
                out << qid << " Q0 " << c.docids[id] << " " << rk << " " << rust_display_f64(scores[id]) << " "
                    << sysname << "\n";
                written++;
            }
            out.flush();
        }
        return std::to_string(written);
    });
}

// ------------------------------------------------------------------------------------------
// extensions
// ------------------------------------------------------------------------------------------

int fr_device_count(void) { return frdev::device_count(nullptr); }

// keys per lane << 16 | lanes per candidate of the full-ranking kernel's size class for a query of `len` documents
uint32_t fr_debug_fullrank_class(uint32_t len) {
    uint32_t nl = 0, pl = 0;
    frdev::fullrank_class_of(len, &nl, &pl);
    return (nl << 16) | pl;
}

// The restart queue without a device: n_workers "trainers" of `capacity` places each, restart r converges after
// lengths[r % n_lengths] ticks, places are refilled at the end of a tick (include/fastrank.h).
const void* fr_debug_restart_queue(uint32_t num_restarts, uint32_t n_workers, uint32_t capacity, const uint32_t* lengths,
                                   uint32_t n_lengths) {
    return json_call([&]() {
        if (n_workers == 0 || capacity == 0 || !lengths || n_lengths == 0) fr::fail_str("fr_debug_restart_queue: bad arguments");
        fr::RestartQueue queue(0, num_restarts);
        struct Place { uint32_t id, left; };
        std::vector<std::vector<Place>> live(n_workers);
        std::vector<std::vector<uint32_t>> order(n_workers);
        std::vector<uint64_t> ticks(n_workers, 0);
        auto fill = [&](uint32_t w) {
            uint32_t id = 0;
            while (live[w].size() < capacity && queue.pop(&id)) {
                live[w].push_back(Place{id, std::max<uint32_t>(lengths[id % n_lengths], 1)});
                order[w].push_back(id);
            }
        };
        for (uint32_t w = 0; w < n_workers; w++) fill(w);
        for (bool any = true; any;) {
            any = false;
            for (uint32_t w = 0; w < n_workers; w++) {
                if (live[w].empty()) continue;
                any = true;
                ticks[w]++;
                for (auto& p : live[w]) p.left--;
                live[w].erase(std::remove_if(live[w].begin(), live[w].end(), [](const Place& p) { return p.left == 0; }), live[w].end());
                fill(w);
            }
        }
        Value o = Value::object(), ord = Value::array(), tk = Value::array();
        for (uint32_t w = 0; w < n_workers; w++) {
            Value a = Value::array();
            for (uint32_t id : order[w]) a.push(Value::uint(id));
            ord.push(std::move(a));
            tk.push(Value::uint(ticks[w]));
        }
        o.set("order", std::move(ord));
        o.set("ticks", std::move(tk));
        return frjson::dump(o);
    });
}

size_t fr_dataset_release_replicas(const CDataset* dataset) {
    if (!dataset || !dataset->view) return 0;
    std::lock_guard<std::mutex> lk(api_mu_of(*dataset));
    size_t n = 0;
    for (fr::DatasetView* v = dataset->view.get(); v; v = v->parent.get()) n += v->release_replicas();
    return n;
}

// The device plan of a request of `num_restarts` units over the devices of `devices_csv` (the FR_DEVICES syntax) on a
// node with `device_count` devices whose first device form lives on `primary_device`: {"devices","slots","blocks"}.
// No device is touched (the CPU tests check the partition and the list parsing with it).
const void* fr_debug_device_plan(const void* devices_csv, int device_count, uint32_t num_restarts, int primary_device) {
    return json_call([&]() {
        const std::string csv = accept_str("devices_csv", devices_csv);
        const DevicePlan pl = plan_devices(parse_device_list(csv.c_str(), device_count), num_restarts, primary_device);
        Value o = Value::object(), d = Value::array(), sl = Value::array(), bl = Value::array();
        for (size_t i = 0; i < pl.devs.size(); i++) {
            d.push(Value::uint((uint64_t)pl.devs[i]));
            sl.push(Value::uint((uint64_t)pl.slot[i]));
            Value b = Value::array();
            b.push(Value::uint(pl.begin[i]));
            b.push(Value::uint(pl.end[i]));
            bl.push(std::move(b));
        }
        o.set("devices", std::move(d));
        o.set("slots", std::move(sl));
        o.set("blocks", std::move(bl));
        return frjson::dump(o);
    });
}

// One timed device-to-device copy of `bytes` bytes src -> dst, made like the dataset copies of train_model's fan-out:
// {"src","dst","bytes","can_access","enabled","ms","gbps"}.
const void* fr_debug_peer_copy(int src_device, int dst_device, size_t bytes) {
    return json_call([&]() {
        int can = 0, en = 0;
        double ms = 0.0;
        std::string err;
        if (!frdev::peer_copy_probe(src_device, dst_device, bytes, &can, &en, &ms, &err)) fr::fail_str(err);
        Value o = Value::object();
        o.set("src", Value::uint((uint64_t)src_device));
        o.set("dst", Value::uint((uint64_t)dst_device));
        o.set("bytes", Value::uint((uint64_t)bytes));
        o.set("can_access", Value::boolean(can != 0));
        o.set("enabled", Value::boolean(en != 0));
        o.set("ms", Value::number(ms));
        o.set("gbps", Value::number(ms > 0.0 ? (double)bytes / (ms * 1e-3) / 1e9 : 0.0));
        return frjson::dump(o);
    });
}

// a JSON list of {"restart_id","score","weights"} (what fr_train_model_shard / fr_ca_state return)
static std::vector<fr::RestartResult> restarts_from_json_text(const std::string& text) {
    Value v = parse_json_or_fail(text);
    if (!v.is_array()) fr::fail_raw("Error(\"invalid type: expected a sequence\", line: 1, column: 1)");
    std::vector<fr::RestartResult> hist;
    for (const auto& r : v.arr) {
        fr::RestartResult h;
        h.restart_id = (uint32_t)fr::json_u64(fr::json_field(r, "restart_id"), "restart_id");
        h.score = fr::json_f64(fr::json_field(r, "score"), "score");
        for (const auto& x : fr::json_field(r, "weights").arr) h.weights.push_back(fr::json_f64(x, "weights"));
        hist.push_back(std::move(h));
    }
    return hist;
}

// The exchange records (host.hpp: pack_restart_records) for callers that gather outside the library.
const void* fr_pack_restart_records(const void* restarts_json, size_t cap, size_t dim, double* out) {
    return status_call([&]() {
        if (out == nullptr) fr::fail_str("fr_pack_restart_records: null output buffer");
        fr::pack_restart_records(restarts_from_json_text(accept_str("restarts_json", restarts_json)), cap, dim, out);
    });
}

const void* fr_unpack_restart_records(const double* records, size_t n_records, size_t dim) {
    return json_call([&]() {
        if (records == nullptr && n_records > 0) fr::fail_str("fr_unpack_restart_records: null input buffer");
        return frjson::dump(restarts_to_json(fr::unpack_restart_records(records, n_records, dim)));
    });
}

// One single-process RCCL all-gather of `n` blocks of block_len doubles over `devices` (rccl_exchange.inc): out (optional)
// receives rank 0's gathered buffer, n * block_len doubles.  Returns the report as JSON ({"ran": false, "reason"} when it
// does not apply), or the error envelope when N distinct GPUs were named and the exchange failed.
const void* fr_rccl_allgather(const int* devices, size_t n, const double* blocks, size_t block_len, double* out) {
    return json_call([&]() {
        if ((devices == nullptr || blocks == nullptr) && n > 0) fr::fail_str("fr_rccl_allgather: null argument");
        std::vector<int> devs(devices, devices + n);
        frdev::RcclReport rep;
        std::vector<double> got;
        std::string err;
        if (!frdev::rccl_allgather(devs, blocks, block_len, &got, &rep, &err)) fr::fail_str(err);
        if (rep.ran && out != nullptr) std::memcpy(out, got.data(), got.size() * sizeof(double));
        return frjson::dump(rccl_report_to_json(rep));
    });
}

// The walk tiles the device form of a dataset would get for these runs and queries (frdev::build_walk_tiles; no device).
const void* fr_debug_walk_tiles(const uint32_t* run_pos, const uint32_t* run_q0, const uint32_t* run_q1, size_t nruns, const uint32_t* qstart,
                                const uint32_t* qlen, size_t nq, size_t np) {
    return json_call([&]() {
        if (nruns > 0 && (!run_pos || !run_q0 || !run_q1)) fr::fail_str("fr_debug_walk_tiles: null run table");
        if (nq > 0 && (!qstart || !qlen)) fr::fail_str("fr_debug_walk_tiles: null query table");
        for (size_t q = 0; q < nq; q++)
            if ((size_t)qstart[q] + qlen[q] > np) fr::fail_str("fr_debug_walk_tiles: a query lies beyond np");
        for (size_t r = 0; r < nruns; r++)
            if (run_q0[r] > run_q1[r] || run_q1[r] > nq) fr::fail_str("fr_debug_walk_tiles: a run names queries that do not exist");
        const frdev::WalkTileLayout wl = frdev::build_walk_tiles(std::vector<uint32_t>(run_pos, run_pos + nruns), std::vector<uint32_t>(run_q0, run_q0 + nruns),
                                                                std::vector<uint32_t>(run_q1, run_q1 + nruns), std::vector<uint32_t>(qstart, qstart + nq),
                                                                std::vector<uint32_t>(qlen, qlen + nq), np);
        Value o = Value::object();
        auto list = [](const auto& v) {
            Value a = Value::array();
            for (auto x : v) a.push(Value::uint((uint64_t)x));
            return a;
        };
        o.set("walk_tile", Value::uint(frdev::WALK_TILE));
        o.set("wt_start", list(wl.wt_start));
        o.set("run_wt0", list(wl.run_wt0));
        o.set("seg", list(wl.seg));
        o.set("wofs", list(wl.wofs));
        return frjson::dump(o);
    });
}

// The same call sequence on ONE device (a one-rank communicator): librccl.so opens, its symbols bind, ncclCommInitAll /
// grouped ncclAllGather / ncclCommDestroy run and the data comes back.  What a one-GPU box can check of the exchange.
const void* fr_debug_rccl_selftest(int device) {
    return json_call([&]() {
        std::vector<double> block(16), got;
        for (size_t i = 0; i < block.size(); i++) block[i] = 0.5 + (double)i;
        frdev::RcclReport rep;
        std::string err;
        if (!frdev::rccl_allgather(std::vector<int>(1, device), block.data(), block.size(), &got, &rep, &err, 1)) fr::fail_str(err);
        return frjson::dump(rccl_report_to_json(rep));
    });
}

int fr_set_device(int ordinal) {
    std::string err;
    if (!frdev::set_device(ordinal, &err)) return 1;
    frdev::warm_device(ordinal);
    g_pinned_device = ordinal;  // train_model then stays on this device unless FR_DEVICES says otherwise
    return 0;
}

const char* fr_version(void) { return "fastrank_amd 0.1.0 (fastrank C ABI 0.9.0-dev / python 0.7.0)"; }

const void* fr_train_model_shard(const void* train_request_json, const CDataset* dataset, uint32_t restart_begin,
                                 uint32_t restart_end) {
    return json_call([&]() {
        const CDataset& ds = require_dataset(dataset);
        ParsedRequest rq = parse_train_request(accept_str("train_request_json", train_request_json));
        if (!rq.is_ca) fr::fail_str("fr_train_model_shard: only CoordinateAscent shards by restart");
        std::lock_guard<std::mutex> lk(api_mu_of(ds));
        std::vector<fr::RestartResult> hist;
        train_ca(ds.view, rq, restart_begin, restart_end, &hist);
        Value o = Value::object();
        o.set("restarts", restarts_to_json(hist));
        {
            std::lock_guard<std::mutex> slk(g_stats_mu);
            o.set("stats", stats_to_json(g_last_stats));
        }
        return frjson::dump(o);
    });
}

struct FrTrainer {
    std::unique_ptr<fr::CATrainer> t;
    std::chrono::steady_clock::time_point t0;
    std::shared_ptr<fr::DataCore> core;  // (its api_mu serialises the calls on this handle with other jobs on the dataset)
};

void* fr_ca_begin(const void* train_request_json, const CDataset* dataset, uint32_t restart_begin,
                  uint32_t restart_end, const void** error_out) {
    if (error_out) *error_out = nullptr;
    FrTrainer* h = nullptr;
    const void* st = status_call([&]() {
        const CDataset& ds = require_dataset(dataset);
        ParsedRequest rq = parse_train_request(accept_str("train_request_json", train_request_json));
        if (!rq.is_ca) fr::fail_str("fr_ca_begin: only CoordinateAscent");
        std::lock_guard<std::mutex> lk(api_mu_of(ds));
        fr::Evaluator ev = fr::make_evaluator(*ds.view, rq.measure, rq.has_qrel ? &rq.qrel : nullptr);
        if (ds.view->host_csr().nq == 0) fr::fail_str("assertion failed: !data.queries().is_empty()");
        auto holder = std::make_unique<FrTrainer>();
        holder->t0 = std::chrono::steady_clock::now();
        holder->t = std::make_unique<fr::CATrainer>(ds.view, std::move(ev), rq.ca, restart_begin, restart_end);
        holder->core = ds.view->core;
        h = holder.release();
    });
    if (st) {
        if (error_out) *error_out = st; else free((void*)st);
        return nullptr;
    }
    return h;
}

void* fr_ca_begin_query_shard(const void* train_request_json, const CDataset* dataset, uint64_t total_queries,
                              fr_allreduce_sum_fn allreduce, void* ctx, const void** error_out) {
    if (error_out) *error_out = nullptr;
    FrTrainer* h = nullptr;
    const void* st = status_call([&]() {
        const CDataset& ds = require_dataset(dataset);
        ParsedRequest rq = parse_train_request(accept_str("train_request_json", train_request_json));
        if (!rq.is_ca) fr::fail_str("fr_ca_begin_query_shard: only CoordinateAscent");
        if (!allreduce) fr::fail_str("fr_ca_begin_query_shard: allreduce callback is null!");
        std::lock_guard<std::mutex> lk(api_mu_of(ds));
        fr::Evaluator ev = fr::make_evaluator(*ds.view, rq.measure, rq.has_qrel ? &rq.qrel : nullptr);
        if (total_queries == 0) fr::fail_str("assertion failed: !data.queries().is_empty()");
        fr::QueryShard shard;
        shard.total_queries = total_queries;
        shard.allreduce = [allreduce, ctx](double* v, size_t n) {
            if (allreduce(ctx, v, n) != 0) fr::fail_str("query-shard allreduce callback failed");
        };
        auto holder = std::make_unique<FrTrainer>();
        holder->t0 = std::chrono::steady_clock::now();
        holder->t = std::make_unique<fr::CATrainer>(ds.view, std::move(ev), rq.ca, 0u, rq.ca.num_restarts, std::move(shard));
        holder->core = ds.view->core;
        h = holder.release();
    });
    if (st) {
        if (error_out) *error_out = st; else free((void*)st);
        return nullptr;
    }
    return h;
}

const void* fr_ca_step(void* trainer, uint64_t max_ticks, uint64_t* ticks_done, int* finished) {
    return status_call([&]() {
        if (!trainer) fr::fail_str("trainer pointer is null!");
        FrTrainer* h = (FrTrainer*)trainer;
        std::lock_guard<std::mutex> lk(h->core->api_mu);
        uint64_t n = 0;
        const bool alive = h->t->run(max_ticks, &n);
        if (ticks_done) *ticks_done = n;
        if (finished) *finished = (!alive || h->t->done()) ? 1 : 0;
    });
}

const void* fr_ca_state(void* trainer) {
    return json_call([&]() {
        if (!trainer) fr::fail_str("trainer pointer is null!");
        FrTrainer* h = (FrTrainer*)trainer;
        std::lock_guard<std::mutex> lk(h->core->api_mu);  // fr_ca_step may be mutating the trainer on another thread
        h->t->stats().seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - h->t0).count();
        Value o = Value::object();
        o.set("restarts", restarts_to_json(h->t->results()));
        o.set("stats", stats_to_json(h->t->stats()));
        o.set("finished", Value::boolean(h->t->done()));
        return frjson::dump(o);
    });
}

void fr_ca_free(void* trainer) { delete (FrTrainer*)trainer; }

const CResult* fr_select_model(const void* restarts_json, int output_ensemble) {
    return c_call<CModel>([&]() {
        std::vector<fr::RestartResult> hist = restarts_from_json_text(accept_str("restarts_json", restarts_json));
        // restart order defines "last maximum" (src/coordinate_ascent.rs:244-251)
        std::stable_sort(hist.begin(), hist.end(),
                         [](const fr::RestartResult& a, const fr::RestartResult& b) { return a.restart_id < b.restart_id; });
        auto* out = new CModel();
        try {
            out->actual = fr::ca_select(hist, output_ensemble != 0);
        } catch (...) {
            delete out;
            throw;
        }
        return out;
    });
}

const void* fr_last_train_stats(void) {
    return json_call([&]() {
        std::lock_guard<std::mutex> slk(g_stats_mu);
        Value o = stats_to_json(g_last_stats);
        if (!g_last_per_device.empty()) {  // what every device of the call did (ticks above = the longest device's)
            Value a = Value::array();
            for (const fr::TrainStats& st : g_last_per_device) a.push(stats_to_json(st));
            o.set("per_device", std::move(a));
        }
        return frjson::dump(o);
    });
}

const void* fr_predict_scores_dense(const CModel* model, const CDataset* dataset, double* out, size_t out_len) {
    return status_call([&]() {
        const CModel& m = require_model(model);
        const CDataset& ds = require_dataset(dataset);
        std::lock_guard<std::mutex> lk(api_mu_of(ds));
        fr::DatasetView& view = *ds.view;
        if (view.instances.empty()) return;
        frdev::DeviceDataset& dev = view.device();
        fr::score_model(view, m.actual);
        std::string err;
        if (!dev.download_scores(0, out, out_len, &err)) fr::fail_str(err);
    });
}

const void* fr_evaluate_dense(const CModel* model, const CDataset* dataset, const CQRel* qrel,
                              const void* evaluator_name, double* out_values, size_t out_len,
                              const void** out_qids_json) {
    return status_call([&]() {
        const CModel& m = require_model(model);
        const CDataset& ds = require_dataset(dataset);
        std::string name = accept_str("evaluator_name", evaluator_name);
        std::lock_guard<std::mutex> lk(api_mu_of(ds));
        fr::DatasetView& view = *ds.view;
        fr::Evaluator ev = fr::make_evaluator(view, name, qrel ? &qrel->actual : nullptr);
        frdev::DeviceDataset& dev = view.device();
        if (out_len < dev.nq()) fr::fail_str("fr_evaluate_dense: output buffer too small");
        fr::score_model(view, m.actual);
        std::string err;
        if (!dev.metric_from_scores(ev.measure, ev.depth, ev.norms.data(), 1, false, &err)) fr::fail_str(err);
        fr::check_flags(dev);
        if (!dev.download_per_query(1, out_values, &err)) fr::fail_str(err);
        if (out_qids_json) {
            Value a = Value::array();
            for (size_t q = 0; q < dev.nq(); q++) a.push(Value::string(view.core->qnames[view.csr_query[q]]));
            *out_qids_json = return_string(frjson::dump(a));
        }
    });
}

const void* fr_rank_order(const CModel* model, const CDataset* dataset, uint32_t* out_instance_ids, size_t n,
                          uint64_t* out_offsets, size_t nq_plus_1) {
    return status_call([&]() {
        const CModel& m = require_model(model);
        const CDataset& ds = require_dataset(dataset);
        std::lock_guard<std::mutex> lk(api_mu_of(ds));
        fr::DatasetView& view = *ds.view;
        frdev::DeviceDataset& dev = view.device();
        if (n < dev.n() || nq_plus_1 < dev.nq() + 1) fr::fail_str("fr_rank_order: output buffers too small");
        fr::score_model(view, m.actual);
        std::string err;
        std::vector<double> norms(dev.nq(), 0.0);
        if (!dev.metric_from_scores(frdev::M_RR, -1, norms.data(), 1, true, &err)) fr::fail_str(err);
        fr::check_flags(dev);
        if (!dev.download_rank(out_instance_ids, &err)) fr::fail_str(err);
        const frdev::HostCSR& csr = view.host_csr();
        for (size_t q = 0; q <= csr.nq; q++) out_offsets[q] = csr.qoff[q];
    });
}

size_t fr_dataset_num_queries(const CDataset* dataset) {
    if (!dataset) return 0;
    // host_csr() validates the labels (a NaN label is an error): nothing may unwind through extern "C".
    // SIZE_MAX = "this dataset cannot be grouped"; the compute calls report the reason in their envelope.
    try {
        return dataset->view->host_csr().nq;
    } catch (...) {
        return SIZE_MAX;
    }
}

const void* fr_dataset_device_info(const CDataset* dataset) {
    return json_call([&]() {
        const CDataset& ds = require_dataset(dataset);
        std::lock_guard<std::mutex> lk(api_mu_of(ds));
        std::shared_ptr<frdev::DeviceDataset> devp = ds.view->device_ptr();
        frdev::DeviceDataset& dev = *devp;
        fr::DatasetView* owner = ds.view->matrix_owner(nullptr);
        const bool is_parents = owner != ds.view.get() && owner->device_ptr() == devp;  // a feature sample: the parent's object itself
        Value o = Value::object();
        o.set("hbm_bytes_owned", Value::uint(is_parents ? 0 : dev.hbm_bytes()));
        o.set("shares_parent_matrix", Value::boolean(is_parents || dev.shares_parent_matrix()));
        o.set("is_parent_device_dataset", Value::boolean(is_parents));
        o.set("queries", Value::uint(dev.nq()));
        o.set("instances", Value::uint(dev.n()));
        return frjson::dump(o);
    });
}

size_t fr_dataset_num_instances(const CDataset* dataset) {
    if (!dataset) return 0;
    return dataset->view->instances.size();
}

const void* fr_evaluate_candidates(const CDataset* dataset, const CQRel* qrel, const void* evaluator_name,
                                   size_t n_groups, const uint32_t* features, const double* base_weights,
                                   const uint32_t* n_cand, const double* candidates, double* out_means,
                                   double* out_per_query) {
    return status_call([&]() {
        const CDataset& ds = require_dataset(dataset);
        std::string name = accept_str("evaluator_name", evaluator_name);
        std::lock_guard<std::mutex> lk(api_mu_of(ds));
        fr::DatasetView& view = *ds.view;
        fr::Evaluator ev = fr::make_evaluator(view, name, qrel ? &qrel->actual : nullptr);
        frdev::DeviceDataset& dev = view.device();
        dev.set_sums_only(false);
        const size_t d = dev.d();
        std::string err;
        for (size_t g = 0; g < n_groups; g++)
            if (n_cand[g] == 0 || n_cand[g] > 64 || features[g] >= d)
                fr::fail_str("fr_evaluate_candidates: malformed group");
        const bool topk = dev.linesearch_supported(ev.measure, ev.depth);
        const bool full = !topk && dev.fullrank_supported(ev.measure, ev.depth) && !frdev::path_env("FR_FORCE_GENERIC");
        if (topk || full) {
            std::vector<frdev::LineGroup> groups(n_groups);
            for (size_t g = 0; g < n_groups; g++) {
                groups[g].feature = features[g];
                groups[g].weights.assign(base_weights + g * d, base_weights + (g + 1) * d);
                groups[g].candidates.assign(candidates + g * 64, candidates + g * 64 + n_cand[g]);
            }
            std::vector<double> means;
            if (topk) {
                if (!dev.linesearch_ndcg(ev.depth, ev.norms.data(), groups, &means, &err)) fr::fail_str(err);
            } else {
                if (!dev.linesearch_fullrank(ev.measure, ev.depth, ev.norms.data(), groups, &means, &err)) fr::fail_str(err);
            }
            fr::check_flags(dev);
            std::copy(means.begin(), means.end(), out_means);
            if (out_per_query) {
                std::vector<double> M;
                size_t ldm = 0;
                if (!dev.download_last_matrix(&M, &ldm, &err)) fr::fail_str(err);
                std::copy(M.begin(), M.end(), out_per_query);
            }
        } else {
            std::vector<double> w;
            std::vector<size_t> slot;
            for (size_t g = 0; g < n_groups; g++)
                for (uint32_t c = 0; c < n_cand[g]; c++) {
                    size_t off = w.size();
                    w.insert(w.end(), base_weights + g * d, base_weights + (g + 1) * d);
                    w[off + features[g]] = candidates[g * 64 + c];
                    slot.push_back(g * 64 + c);
                }
            if (out_per_query) fr::fail_str("fr_evaluate_candidates: per-query output is not available on the general sort path");
            std::vector<double> means;
            fr::evaluate_means_generic(view.device(), ev, w, slot.size(), means);
            for (size_t g = 0; g < n_groups * 64; g++) out_means[g] = 0.0;
            for (size_t k = 0; k < slot.size(); k++) out_means[slot[k]] = means[k];
        }
    });
}

void fr_profile_enable(int on) { frdev::profile_enable(on != 0); }
void fr_profile_reset(void) { frdev::profile_reset(); }

double fr_debug_resident_bound(int which, double a, double b, double c) {
    switch (which) {
        case 0: return frdev::resident_err_refresh((uint32_t)a, b);
        case 1: return frdev::resident_err_update(a, b, c);
        case 2: return frdev::resident_eps_extra(a, b, c);
        default: return std::numeric_limits<double>::quiet_NaN();
    }
}

const void* fr_profile_json(void) {
    return json_call([&]() {
        Value a = Value::array();
        for (const auto& s : frdev::profile_stats()) {
            Value o = Value::object();
            o.set("kernel", Value::string(s.name));
            o.set("launches", Value::uint(s.launches));
            o.set("total_ms", Value::number(s.total_ms));
            a.push(std::move(o));
        }
        return frjson::dump(a);
    });
}

int fr_synchronize(void) {
    std::string err;
    return frdev::device_synchronize(&err) ? 0 : 1;
}

}  // extern "C"
