// json_min.hpp -- the few nlohmann::json operations the Env / ISerializable interface of the reference needs
// (common/serializable.hpp:11-15, env/env_normalize.hpp:134-146, common/running_statistics.hpp:61-85).
// If the reference's vendored ../json.hpp (nlohmann 3.6.1) is on the include path it is used instead.
#pragma once

#if defined(INCLUDE_NLOHMANN_JSON_HPP_)
#define PPO_HAVE_NLOHMANN 1            // the real nlohmann::json is already in this translation unit
#elif defined(__has_include)
#if __has_include("json.hpp") && !defined(PPO_FORCE_JSON_MIN)
#include "json.hpp"
#define PPO_HAVE_NLOHMANN 1
#endif
#endif

#ifndef PPO_HAVE_NLOHMANN
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

namespace nlohmann {

class json {
public:
    enum Kind { Null, Number, String, Array, Object, Bool };
    json() : kind_(Null), num_(0), b_(false) {}
    json(double x) : kind_(Number), num_(x), b_(false) {}
    json(float x) : kind_(Number), num_(x), b_(false) {}
    json(int x) : kind_(Number), num_(x), b_(false) {}
    json(long x) : kind_(Number), num_((double)x), b_(false) {}
    json(bool x) : kind_(Bool), num_(0), b_(x) {}
    json(const char* s) : kind_(String), num_(0), b_(false), str_(s) {}
    json(const std::string& s) : kind_(String), num_(0), b_(false), str_(s) {}
    template <class T>
    json(const std::vector<T>& v) : kind_(Array), num_(0), b_(false) { for (const T& e : v) arr_.push_back(json(e)); }

    json& operator[](const std::string& key) {
        if (kind_ == Null) kind_ = Object;
        if (kind_ != Object) throw std::runtime_error("json: not an object");
        return obj_[key];
    }
    const json& at(const std::string& key) const {
        auto it = obj_.find(key);
        if (kind_ != Object || it == obj_.end()) throw std::out_of_range("json: missing key " + key);
        return it->second;
    }
    bool contains(const std::string& key) const { return kind_ == Object && obj_.count(key); }
    Kind kind() const { return kind_; }
    size_t size() const { return kind_ == Array ? arr_.size() : kind_ == Object ? obj_.size() : 0; }

    template <class T>
    T get() const { return get_impl((T*)nullptr); }

    std::string dump(int indent = -1) const { std::ostringstream o; write(o, indent, 0); return o.str(); }
    static json parse(const std::string& text) { size_t p = 0; json j = parse_value(text, p); skip(text, p); if (p != text.size()) throw std::runtime_error("json: trailing characters"); return j; }

private:
    Kind kind_; double num_; bool b_; std::string str_; std::vector<json> arr_; std::map<std::string, json> obj_;

    double get_impl(double*) const { need(Number); return num_; }
    float get_impl(float*) const { need(Number); return (float)num_; }
    int get_impl(int*) const { need(Number); return (int)num_; }
    bool get_impl(bool*) const { need(Bool); return b_; }
    std::string get_impl(std::string*) const { need(String); return str_; }
    template <class T>
    std::vector<T> get_impl(std::vector<T>*) const { need(Array); std::vector<T> v; for (const json& e : arr_) v.push_back(e.get<T>()); return v; }
    void need(Kind k) const { if (kind_ != k) throw std::runtime_error("json: wrong type"); }

    void write(std::ostream& o, int indent, int depth) const {
        auto nl = [&](int d) { if (indent >= 0) { o << '\n'; for (int i = 0; i < indent * d; ++i) o << ' '; } };
        switch (kind_) {
            case Null: o << "null"; break;
            case Bool: o << (b_ ? "true" : "false"); break;
            case Number: {
                if (!(num_ == num_) || num_ - num_ != 0.0) { o << "null"; break; }   // NaN / inf have no JSON spelling (nlohmann::json dumps null too)
                char buf[40]; snprintf(buf, sizeof buf, "%.17g", num_); o << buf; break;
            }
            case String: write_string(o, str_); break;
            case Array: { o << '['; bool f = true; for (const json& e : arr_) { if (!f) o << ','; f = false; nl(depth + 1); e.write(o, indent, depth + 1); } if (!arr_.empty()) nl(depth); o << ']'; break; }
            case Object: { o << '{'; bool f = true; for (const auto& kv : obj_) { if (!f) o << ','; f = false; nl(depth + 1); write_string(o, kv.first); o << ':'; if (indent >= 0) o << ' '; kv.second.write(o, indent, depth + 1); } if (!obj_.empty()) nl(depth); o << '}'; break; }
        }
    }
    static void write_string(std::ostream& o, const std::string& t) {
        o << '"';
        for (const char ch : t) {
            const unsigned char c = (unsigned char)ch;
            if (c == '"') o << "\\\"";
            else if (c == '\\') o << "\\\\";
            else if (c == '\n') o << "\\n";
            else if (c == '\t') o << "\\t";
            else if (c == '\r') o << "\\r";
            else if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o << b; }
            else o << ch;
        }
        o << '"';
    }
    static std::string parse_string(const std::string& s, size_t& p) {       // s[p] == '"'
        std::string out;
        for (++p; p < s.size(); ++p) {
            const char c = s[p];
            if (c == '"') { ++p; return out; }
            if (c != '\\') { out += c; continue; }
            if (++p >= s.size()) break;
            switch (s[p]) {
                case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
                case 'b': out += '\b'; break; case 'f': out += '\f'; break;
                case 'u': { if (p + 4 >= s.size()) throw std::runtime_error("json: bad \\u escape"); out += (char)strtol(s.substr(p + 1, 4).c_str(), nullptr, 16); p += 4; break; }
                default: out += s[p];
            }
        }
        throw std::runtime_error("json: unterminated string");
    }
    static void skip(const std::string& s, size_t& p) { while (p < s.size() && isspace((unsigned char)s[p])) ++p; }
    static json parse_value(const std::string& s, size_t& p) {
        skip(s, p);
        if (p >= s.size()) throw std::runtime_error("json: unexpected end");
        const char c = s[p];
        if (c == '{') {
            json j; j.kind_ = Object; ++p; skip(s, p);
            if (s[p] == '}') { ++p; return j; }
            for (;;) {
                skip(s, p); json k = parse_value(s, p); skip(s, p);
                if (s[p] != ':') throw std::runtime_error("json: expected ':'");
                ++p; j.obj_[k.get<std::string>()] = parse_value(s, p); skip(s, p);
                if (s[p] == ',') { ++p; continue; }
                if (s[p] == '}') { ++p; return j; }
                throw std::runtime_error("json: expected ',' or '}'");
            }
        }
        if (c == '[') {
            json j; j.kind_ = Array; ++p; skip(s, p);
            if (s[p] == ']') { ++p; return j; }
            for (;;) {
                j.arr_.push_back(parse_value(s, p)); skip(s, p);
                if (s[p] == ',') { ++p; continue; }
                if (s[p] == ']') { ++p; return j; }
                throw std::runtime_error("json: expected ',' or ']'");
            }
        }
        if (c == '"') { json j(parse_string(s, p)); return j; }
        if (!s.compare(p, 4, "true")) { p += 4; return json(true); }
        if (!s.compare(p, 5, "false")) { p += 5; return json(false); }
        if (!s.compare(p, 4, "null")) { p += 4; return json(); }
        char* end = nullptr; const double x = strtod(s.c_str() + p, &end);
        if (end == s.c_str() + p) throw std::runtime_error("json: bad token");
        p = (size_t)(end - s.c_str()); return json(x);
    }
};

}  // namespace nlohmann
#endif
