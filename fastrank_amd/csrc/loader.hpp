// Text loaders that feed the device path: ranksvm/libsvm feature files and TREC qrels.
// Behaviour follows src/libsvm.rs:131-189, src/instance.rs:104-130, src/dataset.rs:211-256 and
// src/qrel.rs:65-102.  Every loaded dataset is densified to an n x n_dim f32 matrix (a missing
// feature reads 0.0, exactly what Features::get -> None -> unwrap_or(0.0) and the sparse dot
// product give) so that it runs through the same HIP kernels as a numpy-made DenseDataset.
#pragma once
#include <zlib.h>

#include <cstdlib>
#include <cstring>

#include "host.hpp"

namespace fr {

// io_helper.rs:18-34 sniffs .gz/.bz2/.zst; zlib's gz* API reads both plain and gzip streams.
class LineReader {
  public:
    explicit LineReader(const std::string& path) {
        auto ends = [&](const char* suf) {
            size_t n = strlen(suf);
            return path.size() >= n && path.compare(path.size() - n, n, suf) == 0;
        };
        if (ends(".bz2") || ends(".zst"))
            fail_str(path + ": bzip2/zstd inputs are not supported by the MI355X build (use .gz or plain text)");
        f_ = gzopen(path.c_str(), "rb");
        if (!f_) {
            fail_raw("Os { code: " + std::to_string(errno) + ", kind: " + (errno == ENOENT ? "NotFound" : "Other") +
                     ", message: " + frjson::rust_debug_str(strerror(errno)) + " }");
        }
        gzbuffer(f_, 1 << 20);
    }
    ~LineReader() {
        if (f_) gzclose(f_);
    }
    bool next(std::string& line) {
        line.clear();
        char buf[1 << 16];
        bool got = false;
        while (gzgets(f_, buf, sizeof(buf))) {
            got = true;
            size_t len = strlen(buf);
            line.append(buf, len);
            if (len && buf[len - 1] == '\n') break;
        }
        return got;
    }

  private:
    gzFile f_ = nullptr;
};

inline std::vector<std::string> split_ws(const std::string& s) {
    std::vector<std::string> out;
    size_t i = 0, n = s.size();
    while (i < n) {
        while (i < n && isspace((unsigned char)s[i])) i++;
        size_t j = i;
        while (j < n && !isspace((unsigned char)s[j])) j++;
        if (j > i) out.emplace_back(s, i, j - i);
        i = j;
    }
    return out;
}

inline std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) a++;
    while (b > a && isspace((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}

inline bool parse_f64_strict(const std::string& t, double* out) {
    if (t.empty()) return false;
    char* end = nullptr;
    *out = strtod(t.c_str(), &end);
    return end == t.c_str() + t.size();
}
inline bool parse_f32_strict(const std::string& t, float* out) {
    if (t.empty()) return false;
    char* end = nullptr;
    *out = strtof(t.c_str(), &end);  // correctly rounded, like fast_float::parse::<f32>
    return end == t.c_str() + t.size();
}

// dataset.rs:14-25 load_feature_names_json: {"1": "name", ...}
inline std::map<uint32_t, std::string> load_feature_names(const std::string& path) {
    LineReader rd(path);
    std::string text, line;
    while (rd.next(line)) text += line;
    Value v;
    try {
        v = frjson::parse(text.c_str());
    } catch (const frjson::ParseError& e) {
        fail_raw(e.debug());
    }
    if (!v.is_object()) fail_raw("Error(\"invalid type: expected a map\", line: 1, column: 1)");
    std::map<uint32_t, std::string> out;
    for (const auto& m : v.obj) {
        char* end = nullptr;
        unsigned long long k = strtoull(m.first.c_str(), &end, 10);
        if (m.first.empty() || end != m.first.c_str() + m.first.size())
            fail_raw("ParseIntError { kind: InvalidDigit }");
        if (!m.second.is_string()) fail_raw("Error(\"invalid type: expected a string\", line: 1, column: 1)");
        out[(uint32_t)k] = m.second.s;
    }
    return out;
}

inline std::shared_ptr<DatasetView> load_ranksvm(const std::string& path,
                                                 const std::map<uint32_t, std::string>* names) {
    struct Row {
        std::vector<std::pair<uint32_t, float>> feats;
        uint32_t dense_len = 0;  // >0: Dense32 of this length (instance.rs:108-115)
    };
    auto core = std::make_shared<DataCore>();
    std::vector<Row> rows;
    std::unordered_map<std::string, uint32_t> qslot;
    std::set<uint32_t> present;
    bool any_docid = false;
    auto lerr = [&](uint64_t line_no, const std::string& kind) {
        fail_str(path + ": LineParseError(" + std::to_string(line_no) + ", " + kind + ")");
    };
    try {
        LineReader rd(path);
        std::string line;
        uint64_t line_no = 0;
        while (rd.next(line)) {
            line_no++;
            std::string data = line, comment;
            bool has_comment = false;
            size_t hash = line.find('#');
            if (hash != std::string::npos) {
                data = line.substr(0, hash);
                comment = trim(line.substr(hash + 1));
                has_comment = true;
            }
            std::vector<std::string> toks = split_ws(data);
            if (toks.empty()) lerr(line_no, "EmptyLine");
            double label64;
            if (!parse_f64_strict(toks[0], &label64)) lerr(line_no, "Label(ParseFloatError { kind: Invalid })");
            float label = (float)label64;
            if (label != label) lerr(line_no, "LabelIsNan(FloatIsNan)");
            size_t t = 1;
            std::string qid;
            bool has_q = false;
            if (t < toks.size() && toks[t].compare(0, 4, "qid:") == 0) {
                qid = toks[t];
                while (qid.compare(0, 4, "qid:") == 0) qid.erase(0, 4);  // trim_start_matches
                has_q = true;
                t++;
            }
            Row row;
            for (; t < toks.size(); t++) {
                size_t colon = toks[t].find(':');
                if (colon == std::string::npos) lerr(line_no, "FeatureNoColon");
                std::string fs = toks[t].substr(0, colon), vs = toks[t].substr(colon + 1);
                char* end = nullptr;
                unsigned long long fid = strtoull(fs.c_str(), &end, 10);
                if (fs.empty() || end != fs.c_str() + fs.size() || fid > 0xFFFFFFFFull || fs[0] == '-')
                    lerr(line_no, "FeatureNum(ParseIntError { kind: InvalidDigit })");
                float val;
                if (!parse_f32_strict(vs, &val)) lerr(line_no, "FeatureValNotFloat(Error)");
                row.feats.emplace_back((uint32_t)fid, val);
            }
            if (row.feats.empty()) lerr(line_no, "NoFeatures");
            bool needs_sort = false;
            for (size_t k = 0; k + 1 < row.feats.size(); k++)
                if (row.feats[k].first >= row.feats[k + 1].first) needs_sort = true;
            if (needs_sort) {
                std::sort(row.feats.begin(), row.feats.end(),
                          [](const auto& a, const auto& b) { return a.first < b.first; });
                for (size_t k = 0; k + 1 < row.feats.size(); k++)
                    if (row.feats[k].first == row.feats[k + 1].first)
                        lerr(line_no, "MultipleDefinitions(Feature { idx: " + std::to_string(row.feats[k].first) + " })");
            }
            if (!has_q) fail_str(path + ": \"Missing qid\"");
            // instance.rs:104-130: dense when len / max_feature >= 0.5
            uint32_t max_feature = 0;
            for (const auto& fv : row.feats) max_feature = std::max(max_feature, fv.first);
            double density = (double)row.feats.size() / (double)max_feature;
            if (density >= 0.5) {
                row.dense_len = max_feature + 1;
                for (uint32_t j = 0; j <= max_feature; j++) present.insert(j);
            } else {
                for (const auto& fv : row.feats) present.insert(fv.first);
            }
            auto it = qslot.find(qid);
            if (it == qslot.end()) {
                it = qslot.emplace(qid, (uint32_t)core->qnames.size()).first;
                core->qnames.push_back(qid);
            }
            core->qix.push_back(it->second);
            core->gain.push_back(label);
            core->docids.push_back(comment);
            core->doc_present.push_back(has_comment ? 1 : 0);  // document_name() is Some only then
            any_docid = any_docid || has_comment;
            rows.push_back(std::move(row));
        }
    } catch (FrError&) {
        throw;
    }
    if (rows.empty() || present.empty()) fail_str(path + ": No features defined!");
    core->n = rows.size();
    core->d = (size_t)(*present.rbegin()) + 1;
    core->features.assign(present.begin(), present.end());
    core->x_own.assign(core->n * core->d, 0.0f);
    for (size_t i = 0; i < rows.size(); i++)
        for (const auto& fv : rows[i].feats) core->x_own[i * core->d + fv.first] = fv.second;
    core->x = core->x_own.data();
    {
        // which features every row holds (see DataCore::present_bits); kept only if some row lacks some feature
        const size_t pw = (core->d + 31) / 32;
        std::vector<uint32_t> bits(core->n * pw, 0u);
        bool any_absent = false;
        for (size_t i = 0; i < rows.size(); i++) {
            uint32_t* b = bits.data() + i * pw;
            size_t held = 0;
            if (rows[i].dense_len > 0) {
                for (uint32_t j = 0; j < rows[i].dense_len && j < core->d; j++) b[j >> 5] |= 1u << (j & 31), held++;
            } else {
                for (const auto& fv : rows[i].feats) b[fv.first >> 5] |= 1u << (fv.first & 31), held++;
            }
            any_absent = any_absent || held != core->d;
        }
        if (any_absent) {
            core->present_bits = std::move(bits);
            core->present_words = pw;
        }
    }
    core->has_docids = any_docid;
    if (names) core->feature_names = *names;
    auto view = std::make_shared<DatasetView>();
    view->core = core;
    view->features = core->features;
    view->instances.resize(core->n);
    for (size_t i = 0; i < core->n; i++) view->instances[i] = (uint32_t)i;
    return view;
}

// src/qrel.rs:65-102
inline QRel load_qrel_file(const std::string& path) {
    QRel q;
    LineReader rd(path);
    std::string line;
    uint64_t num = 0;
    while (rd.next(line)) {
        num++;
        std::vector<std::string> row = split_ws(line);
        if (row.size() < 4) fail_str(path + ":" + std::to_string(num) + ": expected 4 columns");
        float gain;
        if (!parse_f32_strict(row[3], &gain))
            fail_str(path + ":" + std::to_string(num) + ": Invalid relevance judgment " + row[3]);
        if (gain != gain) fail_str(path + ":" + std::to_string(num) + ": NaN relevance judgment.");
        q.insert(row[0], row[2], gain);
    }
    return q;
}

}  // namespace fr
