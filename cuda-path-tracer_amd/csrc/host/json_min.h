/*
 * json_min.h — a small recursive-descent JSON reader for scene files.
 *
 * The reference parses scenes with nlohmann::json (external/include/json.hpp),
 * whose objects are std::map, i.e. iterate in lexicographic key order
 * (json.hpp:3399-3400); material ids depend on that order
 * (src/scene.cpp:576-762).  Object members are therefore kept in a std::map
 * here as well.
 */
#pragma once
#include <cctype>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace ptjson {

struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Value> arr;
    std::map<std::string, Value> obj;

    bool isNumber() const { return kind == Number; }
    bool isArray() const { return kind == Array; }
    bool isObject() const { return kind == Object; }
    bool contains(const std::string& k) const { return kind == Object && obj.count(k) != 0; }
    const Value& at(const std::string& k) const
    {
        auto it = obj.find(k);
        if (kind != Object || it == obj.end()) throw std::runtime_error("json: missing key '" + k + "'");
        return it->second;
    }
    const Value& at(size_t i) const
    {
        if (kind != Array || i >= arr.size()) throw std::runtime_error("json: index out of range");
        return arr[i];
    }
    float f() const
    {
        if (kind != Number) throw std::runtime_error("json: number expected");
        return (float)num;
    }
    int i() const
    {
        if (kind != Number) throw std::runtime_error("json: number expected");
        return (int)num;
    }
    const std::string& s() const
    {
        if (kind != String) throw std::runtime_error("json: string expected");
        return str;
    }
};

class Parser {
public:
    explicit Parser(const std::string& text) : t(text), p(0) {}
    Value parse()
    {
        Value v = value();
        ws();
        if (p != t.size()) fail("trailing characters");
        return v;
    }

private:
    const std::string& t;
    size_t p;
    [[noreturn]] void fail(const char* m) const
    {
        throw std::runtime_error(std::string("json: ") + m + " at offset " + std::to_string(p));
    }
    void ws()
    {
        while (p < t.size() && (t[p] == ' ' || t[p] == '\t' || t[p] == '\n' || t[p] == '\r')) ++p;
    }
    bool eat(char c)
    {
        ws();
        if (p < t.size() && t[p] == c) { ++p; return true; }
        return false;
    }
    Value value()
    {
        ws();
        if (p >= t.size()) fail("unexpected end");
        char c = t[p];
        if (c == '{') return object();
        if (c == '[') return array();
        if (c == '"') { Value v; v.kind = Value::String; v.str = string(); return v; }
        if (c == 't' && t.compare(p, 4, "true") == 0) { p += 4; Value v; v.kind = Value::Bool; v.b = true; return v; }
        if (c == 'f' && t.compare(p, 5, "false") == 0) { p += 5; Value v; v.kind = Value::Bool; v.b = false; return v; }
        if (c == 'n' && t.compare(p, 4, "null") == 0) { p += 4; return Value(); }
        return number();
    }
    Value number()
    {
        const char* s = t.c_str() + p;
        char* e = nullptr;
        double d = strtod(s, &e);
        if (e == s) fail("bad number");
        p += (size_t)(e - s);
        Value v; v.kind = Value::Number; v.num = d;
        return v;
    }
    std::string string()
    {
        ++p; /* opening quote */
        std::string out;
        while (p < t.size() && t[p] != '"') {
            char c = t[p++];
            if (c == '\\') {
                if (p >= t.size()) fail("bad escape");
                char e = t[p++];
                switch (e) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'u': {
                    if (p + 4 > t.size()) fail("bad \\u escape");
                    unsigned cp = (unsigned)strtoul(t.substr(p, 4).c_str(), nullptr, 16);
                    p += 4;
                    if (cp < 0x80) out += (char)cp;
                    else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                    else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                    break;
                }
                default: out += e; break;
                }
            } else out += c;
        }
        if (p >= t.size()) fail("unterminated string");
        ++p;
        return out;
    }
    Value array()
    {
        ++p;
        Value v; v.kind = Value::Array;
        if (eat(']')) return v;
        do { v.arr.push_back(value()); } while (eat(','));
        if (!eat(']')) fail("']' expected");
        return v;
    }
    Value object()
    {
        ++p;
        Value v; v.kind = Value::Object;
        if (eat('}')) return v;
        do {
            ws();
            if (p >= t.size() || t[p] != '"') fail("key expected");
            std::string k = string();
            if (!eat(':')) fail("':' expected");
            v.obj[k] = value();
        } while (eat(','));
        if (!eat('}')) fail("'}' expected");
        return v;
    }
};

} // namespace ptjson
