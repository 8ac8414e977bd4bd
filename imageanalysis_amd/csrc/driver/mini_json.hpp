// A small JSON reader for the driver's config files: objects, arrays, strings, numbers,
// true/false/null, plus the // and /* */ comments the reference's configs contain
// (test/configs/*.json; the reference reads them with jsoncpp, which accepts comments).
// Accessors follow jsoncpp's forgiving style: a missing member is a null value, and
// asInt/asFloat/asBool/asString of null give 0 / 0 / false / "".
#pragma once
#include <cctype>
#include <cstdlib>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace mini_json {

class Value {
public:
    enum Kind { Null, Bool, Number, String, Array, Object };
    Kind kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<Value> arr;
    std::vector<std::pair<std::string, Value>> obj;  // insertion order kept

    bool isNull() const { return kind == Null; }
    size_t size() const { return kind == Array ? arr.size() : kind == Object ? obj.size() : 0; }
    const Value &operator[](const std::string &key) const {
        static const Value null_value;
        if (kind != Object) return null_value;
        for (auto &kv : obj)
            if (kv.first == key) return kv.second;
        return null_value;
    }
    const Value &operator[](size_t i) const {
        static const Value null_value;
        return (kind == Array && i < arr.size()) ? arr[i] : null_value;
    }
    const Value &operator[](int i) const { return (*this)[(size_t)i]; }
    bool isMember(const std::string &key) const { return !(*this)[key].isNull(); }
    int asInt() const { return kind == Number ? (int)num : kind == Bool ? (int)b : 0; }
    float asFloat() const { return kind == Number ? (float)num : kind == Bool ? (float)b : 0.0f; }
    double asDouble() const { return kind == Number ? num : kind == Bool ? (double)b : 0.0; }
    bool asBool() const { return kind == Bool ? b : kind == Number ? num != 0 : false; }
    std::string asString() const {
        if (kind == String) return str;
        if (kind == Bool) return b ? "true" : "false";
        if (kind == Number) {
            char buf[64];
            snprintf(buf, sizeof buf, "%.17g", num);
            return buf;
        }
        return "";
    }
};

class Parser {
public:
    explicit Parser(const std::string &text) : s(text) {}
    Value parse() {
        Value v = value();
        skip();
        if (pos != s.size()) fail("trailing characters");
        return v;
    }

private:
    const std::string &s;
    size_t pos = 0;

    [[noreturn]] void fail(const char *what) const {
        size_t line = 1;
        for (size_t i = 0; i < pos && i < s.size(); i++) line += s[i] == '\n';
        throw std::runtime_error(std::string("JSON: ") + what + " at line " + std::to_string(line));
    }
    void skip() {
        for (;;) {
            while (pos < s.size() && isspace((unsigned char)s[pos])) pos++;
            if (pos + 1 < s.size() && s[pos] == '/' && s[pos + 1] == '/') {
                while (pos < s.size() && s[pos] != '\n') pos++;
            } else if (pos + 1 < s.size() && s[pos] == '/' && s[pos + 1] == '*') {
                size_t e = s.find("*/", pos + 2);
                if (e == std::string::npos) fail("unterminated comment");
                pos = e + 2;
            } else {
                return;
            }
        }
    }
    Value value() {
        skip();
        if (pos >= s.size()) fail("unexpected end");
        const char c = s[pos];
        if (c == '{') return object();
        if (c == '[') return array();
        if (c == '"') {
            Value v;
            v.kind = Value::String;
            v.str = string();
            return v;
        }
        if (s.compare(pos, 4, "true") == 0) { pos += 4; Value v; v.kind = Value::Bool; v.b = true; return v; }
        if (s.compare(pos, 5, "false") == 0) { pos += 5; Value v; v.kind = Value::Bool; v.b = false; return v; }
        if (s.compare(pos, 4, "null") == 0) { pos += 4; return Value(); }
        if (c == '-' || c == '+' || isdigit((unsigned char)c) || c == '.') {
            const char *b = s.c_str() + pos;
            char *e = nullptr;
            Value v;
            v.kind = Value::Number;
            v.num = strtod(b, &e);
            if (e == b) fail("bad number");
            pos += (size_t)(e - b);
            return v;
        }
        fail("unexpected character");
    }
    std::string string() {
        std::string out;
        pos++;  // opening quote
        while (pos < s.size() && s[pos] != '"') {
            char c = s[pos++];
            if (c == '\\') {
                if (pos >= s.size()) fail("bad escape");
                char e = s[pos++];
                switch (e) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'u': {
                    if (pos + 4 > s.size()) fail("bad \\u escape");
                    unsigned cp = (unsigned)strtoul(s.substr(pos, 4).c_str(), nullptr, 16);
                    pos += 4;
                    if (cp < 0x80) out += (char)cp;
                    else if (cp < 0x800) { out += (char)(0xc0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3f)); }
                    else { out += (char)(0xe0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3f)); out += (char)(0x80 | (cp & 0x3f)); }
                    break;
                }
                default: out += e; break;  // \" \\ \/
                }
            } else {
                out += c;
            }
        }
        if (pos >= s.size()) fail("unterminated string");
        pos++;  // closing quote
        return out;
    }
    Value array() {
        Value v;
        v.kind = Value::Array;
        pos++;
        skip();
        if (pos < s.size() && s[pos] == ']') { pos++; return v; }
        for (;;) {
            v.arr.push_back(value());
            skip();
            if (pos < s.size() && s[pos] == ',') { pos++; skip(); if (pos < s.size() && s[pos] == ']') { pos++; return v; } continue; }
            if (pos < s.size() && s[pos] == ']') { pos++; return v; }
            fail("expected , or ]");
        }
    }
    Value object() {
        Value v;
        v.kind = Value::Object;
        pos++;
        skip();
        if (pos < s.size() && s[pos] == '}') { pos++; return v; }
        for (;;) {
            skip();
            if (pos >= s.size() || s[pos] != '"') fail("expected a member name");
            std::string key = string();
            skip();
            if (pos >= s.size() || s[pos] != ':') fail("expected :");
            pos++;
            Value member = value();
            bool replaced = false;
            for (auto &kv : v.obj)
                if (kv.first == key) { kv.second = member; replaced = true; }
            if (!replaced) v.obj.emplace_back(key, member);
            skip();
            if (pos < s.size() && s[pos] == ',') { pos++; skip(); if (pos < s.size() && s[pos] == '}') { pos++; return v; } continue; }
            if (pos < s.size() && s[pos] == '}') { pos++; return v; }
            fail("expected , or }");
        }
    }
};

inline Value parse(const std::string &text) { return Parser(text).parse(); }

}  // namespace mini_json
