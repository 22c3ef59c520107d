// Minimal JSON value / parser / writer for the fastrank C-ABI (JSON strings over pointers).
//
// The reference speaks serde_json (src/ffi.rs, src/json_api.rs).  This file re-creates the
// wire behaviour that matters for a drop-in: u64 integers survive exactly (seeds), floats
// print in shortest round-trip form with serde/ryu's layout ("1.0", "0.05", "1e-7"),
// non-finite floats print as null, object key order is preserved.
#pragma once
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace frjson {

struct ParseError : std::runtime_error {
    size_t line, column;
    ParseError(const std::string& m, size_t l, size_t c) : std::runtime_error(m), line(l), column(c) {}
    // serde_json's Debug form: Error("msg", line: L, column: C)
    std::string debug() const {
        std::string s = "Error(\"";
        s += what();
        s += "\", line: " + std::to_string(line) + ", column: " + std::to_string(column) + ")";
        return s;
    }
};

struct Value;
using Member = std::pair<std::string, Value>;

struct Value {
    enum Kind { Null, Bool, UInt, Int, Float, String, Array, Object } kind = Null;
    bool b = false;
    uint64_t u = 0;
    int64_t i = 0;
    double f = 0.0;
    bool f32 = false;  // print with float (not double) shortest round-trip digits
    std::string s;
    std::vector<Value> arr;
    std::vector<Member> obj;

    Value() = default;
    static Value null() { return Value(); }
    static Value boolean(bool v) { Value x; x.kind = Bool; x.b = v; return x; }
    static Value uint(uint64_t v) { Value x; x.kind = UInt; x.u = v; return x; }
    static Value sint(int64_t v) { Value x; x.kind = Int; x.i = v; return x; }
    static Value number(double v) { Value x; x.kind = Float; x.f = v; return x; }
    static Value number32(float v) { Value x; x.kind = Float; x.f = (double)v; x.f32 = true; return x; }
    static Value string(std::string v) { Value x; x.kind = String; x.s = std::move(v); return x; }
    static Value array() { Value x; x.kind = Array; return x; }
    static Value object() { Value x; x.kind = Object; return x; }

    bool is_null() const { return kind == Null; }
    bool is_number() const { return kind == UInt || kind == Int || kind == Float; }
    bool is_object() const { return kind == Object; }
    bool is_array() const { return kind == Array; }
    bool is_string() const { return kind == String; }

    double as_double() const {
        switch (kind) {
            case UInt: return (double)u;
            case Int: return (double)i;
            case Float: return f;
            default: throw std::runtime_error("expected a number");
        }
    }
    const Value* find(const std::string& key) const {
        if (kind != Object) return nullptr;
        for (const auto& m : obj)
            if (m.first == key) return &m.second;
        return nullptr;
    }
    Value& set(const std::string& key, Value v) {
        for (auto& m : obj)
            if (m.first == key) { m.second = std::move(v); return m.second; }
        obj.emplace_back(key, std::move(v));
        return obj.back().second;
    }
    void push(Value v) { arr.push_back(std::move(v)); }
};

class Parser {
  public:
    explicit Parser(const char* text) : p_(text), begin_(text) {}
    Value parse_document() {
        skip_ws();
        Value v = parse_value();
        skip_ws();
        if (*p_ != '\0') fail("trailing characters");
        return v;
    }

  private:
    const char* p_;
    const char* begin_;
    int depth_ = 0;
    // serde_json refuses documents nested deeper than 128 ("recursion limit exceeded"); so do we,
    // which also bounds every recursive walk over a parsed Value (tree models included).
    static constexpr int MAX_DEPTH = 128;
    struct Nest {
        Parser& p;
        explicit Nest(Parser& q) : p(q) {
            if (++p.depth_ > MAX_DEPTH) p.fail("recursion limit exceeded");
        }
        ~Nest() { --p.depth_; }
    };

    [[noreturn]] void fail(const std::string& msg) const {
        size_t line = 1, col = 0;
        for (const char* q = begin_; q < p_; ++q) {
            if (*q == '\n') { line++; col = 0; } else col++;
        }
        throw ParseError(msg, line, col + 1);
    }
    void skip_ws() {
        while (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r') ++p_;
    }
    Value parse_value() {
        switch (*p_) {
            case '\0': fail("EOF while parsing a value");
            case '{': return parse_object();
            case '[': return parse_array();
            case '"': return Value::string(parse_string());
            case 't': expect_word("true"); return Value::boolean(true);
            case 'f': expect_word("false"); return Value::boolean(false);
            case 'n': expect_word("null"); return Value::null();
            default:
                if (*p_ == '-' || (*p_ >= '0' && *p_ <= '9')) return parse_number();
                fail("expected value");
        }
    }
    void expect_word(const char* w) {
        size_t n = strlen(w);
        if (strncmp(p_, w, n) != 0) fail("expected ident");
        p_ += n;
    }
    Value parse_number() {
        const char* start = p_;
        bool is_float = false;
        if (*p_ == '-') ++p_;
        if (!(*p_ >= '0' && *p_ <= '9')) fail("invalid number");
        while (*p_ >= '0' && *p_ <= '9') ++p_;
        if (*p_ == '.') {
            is_float = true;
            ++p_;
            if (!(*p_ >= '0' && *p_ <= '9')) fail("invalid number");
            while (*p_ >= '0' && *p_ <= '9') ++p_;
        }
        if (*p_ == 'e' || *p_ == 'E') {
            is_float = true;
            ++p_;
            if (*p_ == '+' || *p_ == '-') ++p_;
            if (!(*p_ >= '0' && *p_ <= '9')) fail("invalid number");
            while (*p_ >= '0' && *p_ <= '9') ++p_;
        }
        std::string tok(start, p_);
        if (!is_float) {
            if (tok[0] == '-') {
                int64_t v = 0;
                auto r = std::from_chars(tok.data(), tok.data() + tok.size(), v);
                if (r.ec == std::errc() && r.ptr == tok.data() + tok.size()) return Value::sint(v);
            } else {
                uint64_t v = 0;
                auto r = std::from_chars(tok.data(), tok.data() + tok.size(), v);
                if (r.ec == std::errc() && r.ptr == tok.data() + tok.size()) return Value::uint(v);
            }
        }
        // std::from_chars: locale-independent, no hex floats; out-of-range magnitudes are an error
        // like serde_json's "number out of range" (tiny magnitudes flush to 0.0 as in serde)
        double v = 0.0;
        auto r = std::from_chars(tok.data(), tok.data() + tok.size(), v);
        if (r.ec == std::errc::result_out_of_range) {
            // underflow -> +-0.0 (like serde_json), overflow -> error: decided by the decimal exponent of the first
            // non-zero digit (digits before the point count up, zeros after it count down, plus the written exponent) --
            // "0.<400 zeros>1" and "0.<400 zeros>1e+5" are tiny although they carry no / a positive exponent
            const size_t e = tok.find_first_of("eE");
            const std::string mant = tok.substr(0, e);
            long long written = 0;
            if (e != std::string::npos) {
                const bool neg = tok.size() > e + 1 && tok[e + 1] == '-';
                size_t k = e + 1 + ((tok.size() > e + 1 && (tok[e + 1] == '-' || tok[e + 1] == '+')) ? 1 : 0);
                for (; k < tok.size() && written < 100000000; k++) written = written * 10 + (tok[k] - '0');
                if (neg) written = -written;
            }
            const size_t point = mant.find('.');
            const size_t int_end = point == std::string::npos ? mant.size() : point;
            long long lead = 0;  // decimal exponent of the first non-zero digit, before `written`
            bool found = false;
            for (size_t k = (mant[0] == '-' ? 1 : 0); k < mant.size() && !found; k++) {
                if (mant[k] == '.' || mant[k] == '0') continue;
                lead = k < int_end ? (long long)(int_end - k) - 1 : -(long long)(k - int_end);
                found = true;
            }
            if (found && lead + written >= 0) fail("number out of range");
            return Value::number(tok[0] == '-' ? -0.0 : 0.0);
        }
        if (r.ec != std::errc() || r.ptr != tok.data() + tok.size()) fail("invalid number");
        return Value::number(v);
    }
    static void append_utf8(std::string& out, uint32_t cp) {
        if (cp < 0x80) out += (char)cp;
        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) {
            out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F));
        } else {
            out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F));
            out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F));
        }
    }
    uint32_t parse_hex4() {
        uint32_t v = 0;
        for (int k = 0; k < 4; k++) {
            char c = *p_++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
            else fail("invalid escape");
        }
        return v;
    }
    std::string parse_string() {
        ++p_;  // opening quote
        std::string out;
        for (;;) {
            char c = *p_;
            if (c == '\0') fail("EOF while parsing a string");
            if (c == '"') { ++p_; return out; }
            if (c == '\\') {
                ++p_;
                char e = *p_++;
                switch (e) {
                    case '"': out += '"'; break;
                    case '\\': out += '\\'; break;
                    case '/': out += '/'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'n': out += '\n'; break;
                    case 'r': out += '\r'; break;
                    case 't': out += '\t'; break;
                    case 'u': {
                        uint32_t cp = parse_hex4();
                        if (cp >= 0xD800 && cp < 0xDC00 && p_[0] == '\\' && p_[1] == 'u') {
                            p_ += 2;
                            uint32_t lo = parse_hex4();
                            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                        }
                        append_utf8(out, cp);
                        break;
                    }
                    default: fail("invalid escape");
                }
            } else {
                out += c;
                ++p_;
            }
        }
    }
    Value parse_array() {
        Nest nest(*this);
        ++p_;
        Value v = Value::array();
        skip_ws();
        if (*p_ == ']') { ++p_; return v; }
        for (;;) {
            skip_ws();
            v.arr.push_back(parse_value());
            skip_ws();
            if (*p_ == ',') { ++p_; continue; }
            if (*p_ == ']') { ++p_; return v; }
            fail(*p_ == '\0' ? "EOF while parsing a list" : "expected `,` or `]`");
        }
    }
    Value parse_object() {
        Nest nest(*this);
        ++p_;
        Value v = Value::object();
        skip_ws();
        if (*p_ == '}') { ++p_; return v; }
        for (;;) {
            skip_ws();
            if (*p_ != '"') fail("key must be a string");
            std::string k = parse_string();
            skip_ws();
            if (*p_ != ':') fail("expected `:`");
            ++p_;
            skip_ws();
            v.obj.emplace_back(std::move(k), parse_value());
            skip_ws();
            if (*p_ == ',') { ++p_; continue; }
            if (*p_ == '}') { ++p_; return v; }
            fail(*p_ == '\0' ? "EOF while parsing an object" : "expected `,` or `}`");
        }
    }
};

inline Value parse(const char* text) { return Parser(text).parse_document(); }

// Shortest round-trip digits laid out the way ryu's "pretty" printer (used by serde_json) does.
inline void write_double(std::string& out, double v, bool as_f32 = false) {
    if (!std::isfinite(v)) { out += "null"; return; }
    if (v == 0.0) { out += std::signbit(v) ? "-0.0" : "0.0"; return; }
    char buf[64];
    auto r = as_f32 ? std::to_chars(buf, buf + sizeof(buf), (float)v, std::chars_format::scientific)
                    : std::to_chars(buf, buf + sizeof(buf), v, std::chars_format::scientific);
    std::string sci(buf, r.ptr);  // d[.ddd]e[+-]XX
    bool neg = sci[0] == '-';
    if (neg) sci.erase(0, 1);
    size_t epos = sci.find('e');
    std::string mant = sci.substr(0, epos);
    int exp10 = atoi(sci.c_str() + epos + 1);
    std::string digits;
    for (char c : mant)
        if (c != '.') digits += c;
    int len = (int)digits.size();
    int kk = exp10 + 1;  // position of the decimal point relative to the first digit
    // ryu's pretty printer switches to exponent form outside (1e-5, 1e16) for f64 and
    // outside (1e-6, 1e13) for f32: 1e13f -> "1e13", 1e-6f -> "0.000001"
    const int kk_hi = as_f32 ? 13 : 16, kk_lo = as_f32 ? -6 : -5;
    if (neg) out += '-';
    if (len <= kk && kk <= kk_hi) {  // 1234e7 -> 12340000000.0
        out += digits;
        out.append((size_t)(kk - len), '0');
        out += ".0";
    } else if (0 < kk && kk <= kk_hi) {  // 1234e-2 -> 12.34
        out.append(digits, 0, (size_t)kk);
        out += '.';
        out.append(digits, (size_t)kk, std::string::npos);
    } else if (kk_lo < kk && kk <= 0) {  // 1234e-6 -> 0.001234
        out += "0.";
        out.append((size_t)(-kk), '0');
        out += digits;
    } else {  // exponent form: 1.234e30 / 1e-7
        out += digits[0];
        if (len > 1) {
            out += '.';
            out.append(digits, 1, std::string::npos);
        }
        out += 'e';
        out += std::to_string(kk - 1);
    }
}

inline void write_string(std::string& out, const std::string& s) {
    out += '"';
    for (unsigned char c : s) {
        switch (c) {
            case '"': out += "\\\""; break;
            case '\\': out += "\\\\"; break;
            case '\b': out += "\\b"; break;
            case '\f': out += "\\f"; break;
            case '\n': out += "\\n"; break;
            case '\r': out += "\\r"; break;
            case '\t': out += "\\t"; break;
            default:
                if (c < 0x20) {
                    char b[8];
                    snprintf(b, sizeof(b), "\\u%04x", c);
                    out += b;
                } else {
                    out += (char)c;
                }
        }
    }
    out += '"';
}

inline void write(std::string& out, const Value& v) {
    switch (v.kind) {
        case Value::Null: out += "null"; break;
        case Value::Bool: out += v.b ? "true" : "false"; break;
        case Value::UInt: out += std::to_string(v.u); break;
        case Value::Int: out += std::to_string(v.i); break;
        case Value::Float: write_double(out, v.f, v.f32); break;
        case Value::String: write_string(out, v.s); break;
        case Value::Array: {
            out += '[';
            for (size_t k = 0; k < v.arr.size(); k++) {
                if (k) out += ',';
                write(out, v.arr[k]);
            }
            out += ']';
            break;
        }
        case Value::Object: {
            out += '{';
            for (size_t k = 0; k < v.obj.size(); k++) {
                if (k) out += ',';
                write_string(out, v.obj[k].first);
                out += ':';
                write(out, v.obj[k].second);
            }
            out += '}';
            break;
        }
    }
}

inline std::string dump(const Value& v) {
    std::string s;
    write(s, v);
    return s;
}

// Rust's `{:?}` of a str: quoted with escapes.  Used for the error-envelope "context".
inline std::string rust_debug_str(const std::string& s) {
    std::string out = "\"";
    for (unsigned char c : s) {
        switch (c) {
            case '"': out += "\\\""; break;
            case '\\': out += "\\\\"; break;
            case '\n': out += "\\n"; break;
            case '\r': out += "\\r"; break;
            case '\t': out += "\\t"; break;
            default: out += (char)c;
        }
    }
    out += '"';
    return out;
}

}  // namespace frjson
