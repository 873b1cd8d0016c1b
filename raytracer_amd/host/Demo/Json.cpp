#include "Json.h"

#include <stdlib.h>
#include <string.h>

namespace helpers {
namespace json {

namespace {

struct Parser
{
    const char* p; const char* end; std::string error;
    void SkipWs() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
    bool Fail(const char* what) { if (error.empty()) error = what; return false; }

    bool ParseString(std::string& out)
    {
        out.clear();
        if (p >= end || *p != '"') return Fail("Missing a name for object member.");
        ++p;
        while (p < end && *p != '"')
        {
            if (*p == '\\')
            {
                if (++p >= end) return Fail("Missing a closing quotation mark in string.");
                switch (*p)
                {
                case '"': out += '"'; break; case '\\': out += '\\'; break; case '/': out += '/'; break;
                case 'b': out += '\b'; break; case 'f': out += '\f'; break; case 'n': out += '\n'; break;
                case 'r': out += '\r'; break; case 't': out += '\t'; break;
                case 'u':
                {
                    if (end - p < 5) return Fail("Incorrect hex digit after \\u escape in string.");
                    const unsigned code = (unsigned)strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16);
                    if (code < 0x80) out += (char)code;
                    else if (code < 0x800) { out += (char)(0xC0 | (code >> 6)); out += (char)(0x80 | (code & 0x3F)); }
                    else { out += (char)(0xE0 | (code >> 12)); out += (char)(0x80 | ((code >> 6) & 0x3F)); out += (char)(0x80 | (code & 0x3F)); }
                    p += 4;
                    break;
                }
                default: return Fail("Invalid escape character in string.");
                }
                ++p;
            }
            else out += *p++;
        }
        if (p >= end) return Fail("Missing a closing quotation mark in string.");
        ++p;
        return true;
    }

    // objects / arrays nest by recursion: a bounded depth keeps a hostile file from running the stack out (scene files nest 4 deep)
    static constexpr int MaxDepth = 128;
    int depth = 0;
    struct DepthGuard { int& d; explicit DepthGuard(int& d_) : d(d_) { ++d; } ~DepthGuard() { --d; } };

    bool ParseValue(Value& v)
    {
        SkipWs();
        if (p >= end) return Fail("The document is empty.");
        DepthGuard guard(depth);
        if (depth > MaxDepth) return Fail("The document nests too deeply.");
        if (*p == '{')
        {
            v.type = Value::Type::Object; ++p; SkipWs();
            if (p < end && *p == '}') { ++p; return true; }
            for (;;)
            {
                SkipWs();
                std::string name;
                if (!ParseString(name)) return false;
                SkipWs();
                if (p >= end || *p != ':') return Fail("Missing a colon after a name of object member.");
                ++p;
                v.members.emplace_back(name, Value());
                if (!ParseValue(v.members.back().second)) return false;
                SkipWs();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; return true; }
                return Fail("Missing a comma or '}' after an object member.");
            }
        }
        if (*p == '[')
        {
            v.type = Value::Type::Array; ++p; SkipWs();
            if (p < end && *p == ']') { ++p; return true; }
            for (;;)
            {
                v.array.emplace_back();
                if (!ParseValue(v.array.back())) return false;
                SkipWs();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; return true; }
                return Fail("Missing a comma or ']' after an array element.");
            }
        }
        if (*p == '"') { v.type = Value::Type::String; return ParseString(v.string); }
        if (end - p >= 4 && strncmp(p, "true", 4) == 0) { v.type = Value::Type::Bool; v.boolean = true; p += 4; return true; }
        if (end - p >= 5 && strncmp(p, "false", 5) == 0) { v.type = Value::Type::Bool; v.boolean = false; p += 5; return true; }
        if (end - p >= 4 && strncmp(p, "null", 4) == 0) { v.type = Value::Type::Null; p += 4; return true; }
        if (*p == '-' || (*p >= '0' && *p <= '9'))
        {
            const char* s = p;
            if (*p == '-') ++p;
            size_t digits = 0;
            while (p < end && *p >= '0' && *p <= '9') { ++p; ++digits; }
            if (digits == 0) return Fail("Invalid value.");
            bool real = false;
            if (p < end && *p == '.') { real = true; ++p; while (p < end && *p >= '0' && *p <= '9') ++p; }
            if (p < end && (*p == 'e' || *p == 'E')) { real = true; ++p; if (p < end && (*p == '+' || *p == '-')) ++p; while (p < end && *p >= '0' && *p <= '9') ++p; }
            v.type = Value::Type::Number;
            v.number = strtod(std::string(s, p).c_str(), nullptr);
            v.numberIsReal = real || digits > 19 || (*s == '-' && v.number == 0.0);   // rapidjson falls back to double for these
            return true;
        }
        return Fail("Invalid value.");
    }
};

const Value kNull;

} // namespace

bool Value::HasMember(const char* name) const
{
    for (const auto& m : members) if (m.first == name) return true;
    return false;
}

const Value& Value::operator[](const char* name) const
{
    for (const auto& m : members) if (m.first == name) return m.second;
    return kNull;
}

bool Parse(const std::string& text, Value& out, std::string& error)
{
    Parser parser = { text.data(), text.data() + text.size(), std::string() };
    out = Value();
    // UTF-8 byte order mark
    if (text.size() >= 3 && (unsigned char)text[0] == 0xEF && (unsigned char)text[1] == 0xBB && (unsigned char)text[2] == 0xBF) parser.p += 3;
    if (!parser.ParseValue(out)) { error = parser.error; return false; }
    parser.SkipWs();
    if (parser.p != parser.end) { error = "The document root must not be followed by other values."; return false; }
    return true;
}

} // namespace json
} // namespace helpers
