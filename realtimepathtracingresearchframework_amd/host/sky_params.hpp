// sky_params.hpp -- the fitted sky of a named configuration from package data (data/sky_params.json, written by tools/gen_sky_params.py
// with the reference's own Hosek-Wilkie code compiled in place: the model's data tables are third-party data this package does not
// carry). What scenes.py: Scene.scene_params does, for the C++ hosts. In a drop-in build the adapter runs the reference's
// update_sky_light instead (vulkan/render_sky.cpp:25-72).
#pragma once
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rptr_hip.h"

namespace rptr {

// a JSON value, just enough for the package's own data files (objects, arrays, numbers, strings, true / false / null)
struct Json {
    enum Kind { Null, Number, String, Array, Object, Bool } kind = Null;
    double number = 0.0;
    std::string string;
    std::vector<Json> array;
    std::map<std::string, Json> object;
    const Json &at(const std::string &key) const {
        auto it = object.find(key);
        if (kind != Object || it == object.end()) throw std::runtime_error("JSON: no member \"" + key + "\"");
        return it->second;
    }
    const Json &at(size_t i) const {
        if (kind != Array || i >= array.size()) throw std::runtime_error("JSON: index out of range");
        return array[i];
    }
    float f() const {
        if (kind != Number) throw std::runtime_error("JSON: number expected");
        return (float)number;
    }
};
class JsonParser {
public:
    explicit JsonParser(const std::string &text) : s_(text) {}
    Json parse() {
        Json v = value();
        ws();
        if (at_ != s_.size()) fail("trailing characters");
        return v;
    }

private:
    const std::string &s_;
    size_t at_ = 0;
    [[noreturn]] void fail(const char *what) const { throw std::runtime_error(std::string("JSON: ") + what + " at offset " + std::to_string(at_)); }
    void ws() {
        while (at_ < s_.size() && std::isspace((unsigned char)s_[at_])) ++at_;
    }
    Json value() {
        ws();
        if (at_ >= s_.size()) fail("unexpected end");
        const char c = s_[at_];
        Json v;
        if (c == '{') {
            v.kind = Json::Object;
            ++at_;
            ws();
            if (at_ < s_.size() && s_[at_] == '}') { ++at_; return v; }
            for (;;) {
                ws();
                if (at_ >= s_.size() || s_[at_] != '"') fail("member name expected");
                const std::string key = str();
                ws();
                if (at_ >= s_.size() || s_[at_] != ':') fail("':' expected");
                ++at_;
                v.object[key] = value();
                ws();
                if (at_ < s_.size() && s_[at_] == ',') { ++at_; continue; }
                if (at_ < s_.size() && s_[at_] == '}') { ++at_; return v; }
                fail("',' or '}' expected");
            }
        }
        if (c == '[') {
            v.kind = Json::Array;
            ++at_;
            ws();
            if (at_ < s_.size() && s_[at_] == ']') { ++at_; return v; }
            for (;;) {
                v.array.push_back(value());
                ws();
                if (at_ < s_.size() && s_[at_] == ',') { ++at_; continue; }
                if (at_ < s_.size() && s_[at_] == ']') { ++at_; return v; }
                fail("',' or ']' expected");
            }
        }
        if (c == '"') {
            v.kind = Json::String;
            v.string = str();
            return v;
        }
        if (!s_.compare(at_, 4, "true")) { at_ += 4; v.kind = Json::Bool; v.number = 1; return v; }
        if (!s_.compare(at_, 5, "false")) { at_ += 5; v.kind = Json::Bool; return v; }
        if (!s_.compare(at_, 4, "null")) { at_ += 4; return v; }
        if (!s_.compare(at_, 3, "NaN")) { at_ += 3; v.kind = Json::Number; v.number = std::strtod("nan", nullptr); return v; } // (Python's json writes these)
        if (!s_.compare(at_, 8, "Infinity")) { at_ += 8; v.kind = Json::Number; v.number = std::strtod("inf", nullptr); return v; }
        if (!s_.compare(at_, 9, "-Infinity")) { at_ += 9; v.kind = Json::Number; v.number = -std::strtod("inf", nullptr); return v; }
        char *end = nullptr;
        v.number = std::strtod(s_.c_str() + at_, &end);
        if (end == s_.c_str() + at_) fail("value expected");
        at_ = (size_t)(end - s_.c_str());
        v.kind = Json::Number;
        return v;
    }
    std::string str() {
        std::string out;
        ++at_; // opening quote
        while (at_ < s_.size() && s_[at_] != '"') {
            if (s_[at_] == '\\' && at_ + 1 < s_.size()) {
                const char e = s_[at_ + 1];
                out += e == 'n' ? '\n' : e == 't' ? '\t' : e;
                at_ += 2;
            } else
                out += s_[at_++];
        }
        if (at_ >= s_.size()) fail("unterminated string");
        ++at_;
        return out;
    }
};

inline Json load_json(const std::string &path) {
    std::FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    std::string text;
    char buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, n);
    std::fclose(f);
    return JsonParser(text).parse();
}

// SkyModelParams + sun of the configuration `key` ("default", "grid", "low_sun", "forest", "night"); sun_radiance.w = the probability of
// sampling the sun in next-event estimation: 0.5 x with triangle lights in the scene, 1 x without (render_sky.cpp:67-70)
inline RptrSceneParams load_sky_params(const std::string &json_path, const std::string &key, bool has_lights) {
    const Json doc = load_json(json_path);
    const Json &e = doc.at("entries").at(key);
    RptrSceneParams sp;
    std::memset(&sp, 0, sizeof(sp));
    for (size_t i = 0; i < 9; ++i)
        for (size_t c = 0; c < e.at("configs").at(i).array.size() && c < 4; ++c) sp.sky_params.configs[i][c] = e.at("configs").at(i).at(c).f();
    for (size_t c = 0; c < e.at("radiances").array.size() && c < 4; ++c) sp.sky_params.radiances[c] = e.at("radiances").at(c).f();
    for (size_t c = 0; c < 3; ++c) sp.sun_dir[c] = e.at("sun_dir").at(c).f();
    sp.sun_cos_angle = e.at("sun_cos_angle").f();
    const Json &sun = e.at(has_lights ? "sun_radiance_lights" : "sun_radiance_nolights");
    for (size_t c = 0; c < 4; ++c) sp.sun_radiance[c] = sun.at(c).f();
    sp.normal_z_scale = 1.0f;
    return sp;
}

} // namespace rptr
