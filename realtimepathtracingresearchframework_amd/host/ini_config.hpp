// ini_config.hpp -- reader for the reference's configuration / keyframe files (--config, --keyframe; cmdline.cpp:19-29,296-474,
// main.cpp:121-149). The reference serialises its UI state through Dear ImGui's settings mechanism (imstate.cpp:226-330,576-598):
//
//   [Application][<target>]        an object; <target> "" = the application state, otherwise a scene id
//   name=value [value ...]         an attribute of the current (sub)object; floats are written %e, ints / bools %d
//   [.][Name]  /  [.][*Name]       open the sub-object Name (a collapsing header such as Camera, Sensor, Tonemapping, Sun, Scene;
//   ..                             close it        "*" marks a combo box whose selected entry is the attribute "<entry>=1")
//
// This reader keeps what a headless run of the path-traced hot path consumes: camera (position / direction / up), the RenderParams
// sliders (batch spp, max path depth, rr path depth, glossy-only mode, pixel radius, output channel / moment, exposure, tone mapping
// operator, variance radius), LightSamplingConfig (light bin size, light mis angle), target spp, the BVH policy (force bvh rebuild,
// rebuild triangle budget), the integrator variant and bump scale, and the Sun sliders (height, angle, turbidity, Color:
// libapp/scene_state.h:79-96) -- the host refits the sky for them (host/sky_fit.hpp) when it was told where the Hosek-Wilkie data
// headers are (--sky-data / RPTR_SKY_DATA); without the data the scene file's sky is kept and a note says so.
// Every [Application] block of a file is one keyframe when the file is given with --keyframe.
#pragma once
#include "../../include/rptr_hip.h"
#include "pointsets.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace rptr {

struct IniObject { // attributes and sub-objects of one [Application][target] block
    std::map<std::string, std::string> attributes;
    std::map<std::string, IniObject> children;
    const IniObject *child(const std::string &name) const {
        auto it = children.find(name);
        return it == children.end() ? nullptr : &it->second;
    }
    bool get(const std::string &name, float *v, int n) const {
        auto it = attributes.find(name);
        if (it == attributes.end()) return false;
        const char *p = it->second.c_str();
        for (int i = 0; i < n; ++i) {
            char *end = nullptr;
            const float x = std::strtof(p, &end);
            if (end == p) break;
            v[i] = x;
            p = end;
        }
        return true;
    }
    bool get(const std::string &name, int *v) const {
        auto it = attributes.find(name);
        if (it == attributes.end()) return false;
        *v = (int)std::strtol(it->second.c_str(), nullptr, 10);
        return true;
    }
    // the selected entry of a combo box object ([.][*name]: "<entry>=1")
    std::string selected() const {
        for (auto &kv : attributes)
            if (std::strtol(kv.second.c_str(), nullptr, 10) != 0) return kv.first;
        return std::string();
    }
};

struct IniBlock {
    std::string target; // "" = application state
    IniObject root;
};

inline std::vector<IniBlock> parse_ini(const std::string &path) {
    FILE *f = std::fopen(path.c_str(), "r");
    if (!f) throw std::runtime_error("Cannot find config file: " + path); // main.cpp:131-133
    std::vector<IniBlock> blocks;
    std::vector<IniObject *> stack;
    char line[4096];
    while (std::fgets(line, sizeof(line), f)) {
        size_t n = std::strlen(line);
        while (n && (line[n - 1] == '\n' || line[n - 1] == '\r')) line[--n] = 0;
        if (!n || line[0] == ';' || line[0] == '#') continue;
        if (line[0] == '[') {
            const char *close = std::strchr(line, ']');
            if (!close || close[1] != '[') { // another ImGui settings type ([Window][..], [Table][..]): not ours
                stack.clear();
                continue;
            }
            const std::string type((const char *)line + 1, close);
            const char *name_begin = close + 2, *name_end = std::strchr(name_begin, ']');
            std::string name(name_begin, name_end ? name_end : name_begin + std::strlen(name_begin));
            if (type == "Application") {
                blocks.emplace_back();
                blocks.back().target = name;
                stack.clear();
                stack.push_back(&blocks.back().root);
            } else if (type == "." && !stack.empty()) {
                if (!name.empty() && name[0] == '*') name.erase(0, 1);
                stack.push_back(&stack.back()->children[name]);
            } else
                stack.clear();
            continue;
        }
        if (stack.empty()) continue;
        if (line[0] == '.' && line[1] == '.') {
            if (stack.size() > 1) stack.pop_back();
            continue;
        }
        const char *eq = std::strchr(line, '=');
        if (!eq) continue;
        stack.back()->attributes[std::string((const char *)line, eq)] = std::string(eq + 1);
    }
    std::fclose(f);
    return blocks;
}

// what a configuration changes: the fields are only touched when the file names them
struct HostConfig {
    RptrRenderParams params;
    RptrLightSamplingConfig lighting;
    RptrCamera camera;
    int target_spp = -1;
    int variant = -1; // -1: not set
    int rng_variant = -1; // RNG_VARIANT_*, -1: not set
    int force_bvh_rebuild = 0, rebuild_triangle_budget = 500000; // RBO_rebuild_triangle_budget_DEFAULT
    bool bvh_policy_set = false; // a file named one of the two
    float bump_scale = 0.f; // 0: not set
    bool sun_changed = false;
    // the Sun header of the scene state (libapp/scene_state.h:79-96); NAN = not named by any file so far
    float sun_height = NAN, sun_angle = NAN, turbidity = NAN, albedo[3] = {NAN, NAN, NAN};
    std::vector<std::string> notes;
};

inline void apply_ini_object(const IniObject &o, HostConfig &c) {
    // application state (libapp/app_state.cpp:17-115)
    o.get("target spp", &c.target_spp);
    o.get("batch spp", &c.params.batch_spp);
    o.get("max path depth", &c.params.max_path_depth);
    o.get("rr path depth", &c.params.rr_path_depth);
    o.get("glossy-only mode", &c.params.glossy_only_mode);
    if (o.attributes.count("force bvh rebuild") || o.attributes.count("rebuild triangle budget")) c.bvh_policy_set = true;
    o.get("force bvh rebuild", &c.force_bvh_rebuild);
    o.get("rebuild triangle budget", &c.rebuild_triangle_budget);
    o.get("pixel radius", &c.params.pixel_radius, 1);
    o.get("output moment", &c.params.output_moment);
    o.get("variance radius", &c.params.variance_radius, 1);
    if (const IniObject *ch = o.child("output channel")) { // OUTPUT_CHANNEL_NAMES, render_params.glsl.h:45-54
        const std::string s = ch->selected();
        static const char *names[] = {"COLOR", "ALBEDO_ROUGHNESS", "NORMAL_DEPTH", "MOTION_JITTER"};
        for (int i = 0; i < 4; ++i)
            if (s.find(names[i]) != std::string::npos) c.params.output_channel = i;
    }
    if (const IniObject *ch = o.child("variant")) {
        const std::string s = ch->selected();
        if (s.find("transmission") != std::string::npos) c.variant = RPTR_VARIANT_GLTF_TRANSMISSION;
        else if (s.find("diffuse") != std::string::npos || s.find("simple") != std::string::npos) c.variant = RPTR_VARIANT_SIMPLE;
        else if (!s.empty()) c.variant = RPTR_VARIANT_GLTF;
    }
    if (const IniObject *ch = o.child("pointset")) {
        const std::string s = ch->selected();
        const int v = rng_variant_from_name(s);
        if (v >= 0) c.rng_variant = v;
        else if (!s.empty()) c.notes.push_back("pointset " + s + ": not one of RNG_VARIANT_NAMES");
    }
    // scene state (libapp/camera_state.h:19-40, libapp/scene_state.h:45-101)
    if (const IniObject *cam = o.child("Camera")) {
        cam->get("position", c.camera.pos, 3);
        cam->get("direction", c.camera.dir, 3);
        cam->get("up", c.camera.up, 3);
    }
    if (const IniObject *s = o.child("Sensor")) {
        s->get("aperture radius", &c.params.aperture_radius, 1);
        s->get("focal distance", &c.params.focus_distance, 1);
        s->get("focal length", &c.params.focal_length, 1);
        s->get("light bin size", &c.lighting.bin_size);
        s->get("light mis angle", &c.lighting.light_mis_angle, 1);
    }
    if (const IniObject *t = o.child("Tonemapping")) {
        t->get("exposure", &c.params.exposure, 1);
        if (const IniObject *op = t->child("operator")) { // COMPATIBILITY_TONEMAPPING_OPERATOR_NAMES + TONEMAPPING_MODES_NAMES (postprocess/tonemapping.h)
            const std::string s = op->selected();
            if (s.find("NEUTRAL") != std::string::npos || s.find("LOG") != std::string::npos) c.params.early_tone_mapping_mode = 1;
            else if (s.find("FAST") != std::string::npos) c.params.early_tone_mapping_mode = 2;
            else if (s.find("NO_TONE") != std::string::npos || s.find("LINEAR") != std::string::npos) c.params.early_tone_mapping_mode = 0;
        }
    }
    if (const IniObject *fl = o.child("Filtering")) { // libapp/app_state.cpp:148-195
        if (const IniObject *op = fl->child("reprojection")) { // REPROJECTION_MODE_NAMES (postprocess/reprojection.h)
            const std::string s = op->selected();
            if (s.find("DISCARD_HISTORY") != std::string::npos) c.params.reprojection_mode = 1;
            else if (s.find("ACCUMULATE") != std::string::npos) {
                c.params.reprojection_mode = 0;
                c.notes.push_back("reprojection ACCUMULATE belongs to ENABLE_REALTIME_RESOLVE builds: NONE is used");
            } else if (s.find("NONE") != std::string::npos) c.params.reprojection_mode = 0;
        }
        int upscale = c.params.render_upscale_factor == 2 ? 1 : 0, taa = c.params.enable_raster_taa != 0 ? 1 : 0, unjittered = c.params.enable_raster_taa < 0 ? 1 : 0;
        fl->get("use 2x upscaling", &upscale);
        fl->get("raster TAA pattern", &taa);
        fl->get("unjittered raster pattern", &unjittered);
        c.params.render_upscale_factor = upscale ? 2 : 1;
        c.params.enable_raster_taa = unjittered ? -1 : (taa ? 1 : 0);
    }
    if (const IniObject *s = o.child("Sun")) {
        if (!s->attributes.empty()) c.sun_changed = true;
        s->get("height", &c.sun_height, 1);
        s->get("angle", &c.sun_angle, 1);
        s->get("turbidity", &c.turbidity, 1);
        s->get("Color", c.albedo, 3);
    }
    if (const IniObject *s = o.child("Scene")) s->get("bump scale", &c.bump_scale, 1);
}

// applies every block of `path` (all targets: the headless host has one scene); returns the number of [Application] blocks
inline int load_config(const std::string &path, HostConfig &c) {
    const std::vector<IniBlock> blocks = parse_ini(path);
    for (const IniBlock &b : blocks) apply_ini_object(b.root, c);
    return (int)blocks.size();
}

} // namespace rptr
