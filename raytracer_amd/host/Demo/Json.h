// A small JSON document model for helpers::LoadScene.  The reference parses scene files with rapidjson (vendored
// under External/, a third-party dependency); SceneLoader only needs objects, arrays, strings, booleans and numbers,
// plus rapidjson's distinction between numbers written as integers and as reals: Value::IsFloat() is true only
// for tokens that were parsed as doubles (a '.', an exponent, or too many digits for 64 bits), IsInt() only for
// tokens that fit an int -- SceneLoader's TryParseFloat / TryParseInt reject the other kind (SceneLoader.cpp:101-145).
#pragma once

#include <map>
#include <memory>
#include <string>
#include <vector>

namespace helpers {
namespace json {

class Value
{
public:
    enum class Type { Null, Bool, Number, String, Array, Object };
    Type type = Type::Null;
    bool boolean = false;
    double number = 0.0;
    bool numberIsReal = false;     // written with '.', 'e' or 'E'
    std::string string;
    std::vector<Value> array;
    std::vector<std::pair<std::string, Value>> members;   // document order; lookups return the first match

    bool IsObject() const { return type == Type::Object; }
    bool IsArray() const { return type == Type::Array; }
    bool IsString() const { return type == Type::String; }
    bool IsBool() const { return type == Type::Bool; }
    bool IsFloat() const { return type == Type::Number && numberIsReal && number >= -3.4028234e38 && number <= 3.4028234e38; }
    bool IsInt() const { return type == Type::Number && !numberIsReal && number >= -2147483648.0 && number <= 2147483647.0; }
    float GetFloat() const { return (float)number; }       // static_cast<float>(GetDouble())
    int GetInt() const { return (int)number; }
    bool GetBool() const { return boolean; }
    const char* GetString() const { return string.c_str(); }
    size_t Size() const { return array.size(); }
    const Value& operator[](size_t i) const { return array[i]; }
    bool HasMember(const char* name) const;
    const Value& operator[](const char* name) const;       // a Null value when absent
};

// false + message when the text is not a JSON document
bool Parse(const std::string& text, Value& out, std::string& error);

} // namespace json
} // namespace helpers
