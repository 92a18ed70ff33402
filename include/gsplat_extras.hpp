// gsplat_extras.hpp — header-only C++ for the rows either side of the training path, on top
// of gsplat_shim.hpp's ModelSplatsHost:
//   * the text `.gobj` splat format of the reference's save / load menu entries (src/ui/UiFrame.cpp:333-358, :373-450)
//   * `settings.json`, the Project's serialised members (src/Project.h:64-73, src/ui/UiFrame.cpp:323-331, :360-371)
//   * the three start fields: grid, mono, one splat per OBJ triangle (src/ui/UiFrame.cpp:137-264)
// No GUI, no progress dialogs.  Same file format, same constants, same error behaviour
// (std::runtime_error("Inconsistent SH degree!"), "Unexpected vertex count in face list!", the five-vector
// constructor's dimension checks).
//
// Structure (this file's own, mirrored by gaussian-splatterer_amd/io.py and fields.py):
//   GobjSchema      the five record kinds of a splat as a table {tag, floats per splat, array}: the writer and the
//                   reader are loops over that table
//   SplatRecord     one splat's five attribute groups, the unit every field initialiser emits
//   lattice / mono / triangles   generators of SplatRecords; makeField() pushes them into a ModelSplatsHost
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <istream>
#include <memory>
#include <ostream>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "gsplat_shim.hpp"

namespace gsplat_shim {

constexpr int SPLATS_LIMIT = 1000000;  // src/Config.h:17
constexpr int SPLATS_SH_DEGREE = 1;    // src/Config.h:19
constexpr int SPLATS_SH_COEF = 4;      // src/Config.h:20

// ---------------------------------------------------------------------------------------------------------
// .gobj
// ---------------------------------------------------------------------------------------------------------
// One record kind: the line tag, how many floats a splat has of it (0 = "3 * shCoeffs", the only variable one) and
// where they live in a host model.
struct GobjField {
    const char* tag;
    int fixedWidth;
    float* ModelSplatsHost::*array;
    int width(int shCoeffs) const { return fixedWidth ? fixedWidth : 3 * shCoeffs; }
};
// File order within a splat: v, sh, s, a, r (src/ui/UiFrame.cpp:343-357).
inline const std::array<GobjField, 5>& gobjSchema() {
    static const std::array<GobjField, 5> schema = { {
        { "v", 3, &ModelSplatsHost::locations },
        { "sh", 0, &ModelSplatsHost::shs },
        { "s", 3, &ModelSplatsHost::scales },
        { "a", 1, &ModelSplatsHost::opacities },
        { "r", 4, &ModelSplatsHost::rotations },
    } };
    return schema;
}

// "<tag> f0 f1 ...\n" at the stream's default float formatting (6 significant digits: the format is lossy)
inline void writeRecord(std::ostream& os, const char* tag, const float* values, int n) {
    os << tag;
    for (int k = 0; k < n; k++) os << ' ' << values[k];
    os << '\n';
}

inline void writeSplats(std::ostream& os, const ModelSplatsHost& model) {
    for (int i = 0; i < model.count; i++)
        for (const GobjField& f : gobjSchema()) {
            const int w = f.width(model.shCoeffs);
            writeRecord(os, f.tag, (model.*(f.array)) + (size_t)i * w, w);
        }
}
inline void saveSplats(const std::string& path, const ModelSplatsHost& model) {
    std::ofstream file(path);
    if (!file) throw std::runtime_error("Failed to open splats file \"" + path + "\" for writing!");
    writeSplats(file, model);
}

// Reader: every line is "<tag> numbers..."; a known tag appends up to its width to that attribute's column (the
// variable-width `sh` record takes every number on the line and all `sh` lines must agree), unknown tags and blank
// lines are skipped, as the reference's prefix chain does.  The columns then go through the five-vector constructor,
// which owns the dimension checks.
// One number the way `iss >> x` (float) reads it: leading blanks skipped, then the longest run of decimal-float characters;
// "nan", "inf" and hexadecimal forms are not numbers to a stream.  Returns false (q unchanged) when there is none.
inline bool streamFloat(const char*& q, float& v) {
    const char* p = q;
    while (*p == ' ' || *p == '\t' || *p == '\r') p++;
    char buf[64];
    size_t n = 0;
    while (n + 1 < sizeof(buf) && *p && std::strchr("+-0123456789.eE", *p)) buf[n++] = *p++;
    buf[n] = 0;
    if (n == 0) return false;
    char* end = nullptr;
    v = std::strtof(buf, &end);
    if (end == buf) return false;
    q += (p - n - q) + (end - buf);
    return true;
}

inline std::unique_ptr<ModelSplatsHost> readSplats(std::istream& is) {
    const auto& schema = gobjSchema();
    std::array<std::vector<float>, 5> column;
    int shPerSplat = -1;
    std::string line;
    while (std::getline(is, line)) {
        const char* p = line.c_str();
        while (*p == ' ' || *p == '\t') p++;
        const char* tagEnd = p;
        while (*tagEnd && *tagEnd != ' ' && *tagEnd != '\t' && *tagEnd != '\r') tagEnd++;
        const std::string tag(p, tagEnd);
        for (size_t k = 0; k < schema.size(); k++) {
            if (tag != schema[k].tag) continue;
            const char* q = tagEnd;
            if (schema[k].fixedWidth) {
                // the reference pushes exactly 3 / 1 / 4 values per v, s / a / r line whatever the line holds
                // (src/ui/UiFrame.cpp:404-434: `float x; iss >> x; push_back(x)`; a failed extraction stores 0)
                bool ok = true;
                for (int f = 0; f < schema[k].fixedWidth; f++) {
                    float v = 0.0f;
                    ok = ok && streamFloat(q, v);
                    column[k].push_back(ok ? v : 0.0f);
                }
            } else {
                int got = 0;
                float v;
                while (streamFloat(q, v)) { column[k].push_back(v); got++; }   // `while (iss >> x)`
                if (shPerSplat < 0) shPerSplat = got;
                else if (shPerSplat != got) throw std::runtime_error("Inconsistent SH degree!");
            }
        }
    }
    return std::make_unique<ModelSplatsHost>(column[0], column[1], column[2], column[3], column[4]);
}
inline std::unique_ptr<ModelSplatsHost> loadSplats(const std::string& path) {
    std::ifstream file(path);
    if (!file) throw std::runtime_error("Failed to load splats file at \"" + path + "\"!");
    return readSplats(file);
}

// ---------------------------------------------------------------------------------------------------------
// settings.json — UiFrame::saveSettings / loadSettings (src/ui/UiFrame.cpp:323-331, :360-371): the Project through
// nlohmann's NLOHMANN_DEFINE_TYPE_INTRUSIVE_WITH_DEFAULT (src/Project.h:22,64-73).  What that macro pair fixes and this
// section reproduces without the library: keys = member names; `file << j` writes ONE compact line whose keys are sorted
// (nlohmann::json's object is a std::map); floats are stored as double(float) and written with the shortest digits that
// read back to that double; from_json WITH_DEFAULT gives a key missing from the file the value of a default-constructed
// Project (not the loaded-into object's current one); unknown keys are ignored; a value of the wrong JSON type throws.
// The table below is the serialised member list in the reference's order; writer and reader are loops over it
// (mirrored by gaussian-splatterer_amd/io.py, byte-identical output: tests/test_next_rows.py).
// ---------------------------------------------------------------------------------------------------------
struct JsonValue {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool boolean = false;
    double number = 0.0;
    bool integral = false;      // the literal had no fraction / exponent
    std::string text;
    std::vector<JsonValue> items;
    std::vector<std::pair<std::string, JsonValue>> members;
    const JsonValue* find(const std::string& key) const {
        const JsonValue* hit = nullptr;
        for (const auto& m : members) if (m.first == key) hit = &m.second;     // a repeated key: the last one wins, as in nlohmann
        return hit;
    }
};

class JsonReader {
public:
    explicit JsonReader(const std::string& src) : s(src) {}
    JsonValue parse() {
        JsonValue v = value();
        blanks();
        if (i != s.size()) fail("trailing characters");
        return v;
    }
private:
    const std::string& s;
    size_t i = 0;
    [[noreturn]] void fail(const char* what) const { throw std::runtime_error(std::string("settings.json: ") + what + " at offset " + std::to_string(i)); }
    void blanks() { while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r')) i++; }
    bool eat(const char* word) { const size_t n = std::strlen(word); if (s.compare(i, n, word) == 0) { i += n; return true; } return false; }
    static void utf8(std::string& out, unsigned cp) {
        if (cp < 0x80) out += (char)cp;
        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
        else { out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
    }
    unsigned hex4() {
        if (i + 4 > s.size()) fail("short \\u escape");
        unsigned v = 0;
        for (int k = 0; k < 4; k++) {
            const char c = s[i++];
            v = v * 16 + (c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : (fail("bad \\u escape"), 0));
        }
        return v;
    }
    std::string string() {
        std::string out;
        i++;  // opening quote
        while (true) {
            if (i >= s.size()) fail("unterminated string");
            const char c = s[i++];
            if (c == '"') return out;
            if (c != '\\') { out += c; continue; }
            if (i >= s.size()) fail("unterminated escape");
            const char e = s[i++];
            switch (e) {
                case '"': out += '"'; break; case '\\': out += '\\'; break; case '/': out += '/'; break;
                case 'b': out += '\b'; break; case 'f': out += '\f'; break; case 'n': out += '\n'; break;
                case 'r': out += '\r'; break; case 't': out += '\t'; break;
                case 'u': {
                    unsigned cp = hex4();
                    if (cp >= 0xD800 && cp < 0xDC00 && s.compare(i, 2, "\\u") == 0) { i += 2; const unsigned lo = hex4(); cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00); }
                    utf8(out, cp);
                    break;
                }
                default: fail("unknown escape");
            }
        }
    }
    JsonValue value() {
        blanks();
        if (i >= s.size()) fail("unexpected end");
        JsonValue v;
        const char c = s[i];
        if (c == '{') {
            v.kind = JsonValue::Object;
            i++; blanks();
            if (i < s.size() && s[i] == '}') { i++; return v; }
            while (true) {
                blanks();
                if (i >= s.size() || s[i] != '"') fail("expected a key");
                std::string key = string();
                blanks();
                if (i >= s.size() || s[i] != ':') fail("expected ':'");
                i++;
                v.members.emplace_back(std::move(key), value());
                blanks();
                if (i < s.size() && s[i] == ',') { i++; continue; }
                if (i < s.size() && s[i] == '}') { i++; return v; }
                fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            v.kind = JsonValue::Array;
            i++; blanks();
            if (i < s.size() && s[i] == ']') { i++; return v; }
            while (true) {
                v.items.push_back(value());
                blanks();
                if (i < s.size() && s[i] == ',') { i++; continue; }
                if (i < s.size() && s[i] == ']') { i++; return v; }
                fail("expected ',' or ']'");
            }
        }
        if (c == '"') { v.kind = JsonValue::String; v.text = string(); return v; }
        if (eat("true")) { v.kind = JsonValue::Bool; v.boolean = true; return v; }
        if (eat("false")) { v.kind = JsonValue::Bool; return v; }
        if (eat("null")) return v;
        const size_t start = i;
        while (i < s.size() && std::strchr("+-0123456789.eE", s[i])) i++;
        if (i == start) fail("unexpected character");
        const std::string lit = s.substr(start, i - start);
        char* end = nullptr;
        v.kind = JsonValue::Number;
        v.number = std::strtod(lit.c_str(), &end);
        if (end != lit.c_str() + lit.size()) { i = start; fail("bad number"); }
        v.integral = lit.find_first_of(".eE") == std::string::npos;
        return v;
    }
};

// A JSON number for a double: the shortest digits that read back to it (what nlohmann's writer and Python's repr produce),
// laid out by Python's rule — fixed notation for 1e-4 <= |x| < 1e16 with ".0" on integral values, else d.ddde+XX — so that
// io.py writes the same bytes.  Non-finite values become null, as nlohmann writes them.
inline std::string jsonNumber(double x) {
    if (!std::isfinite(x)) return "null";
    if (x == 0.0) return std::signbit(x) ? "-0.0" : "0.0";
    char buf[40];
    int prec = 1;
    for (; prec <= 17; prec++) {      // shortest precision that round-trips (%.{p}e has p + 1 significant digits)
        std::snprintf(buf, sizeof buf, "%.*e", prec - 1, x);
        if (std::strtod(buf, nullptr) == x) break;
    }
    std::string digits;
    const char* p = buf;
    const bool neg = *p == '-';
    if (neg) p++;
    for (; *p && *p != 'e'; p++) if (*p != '.') digits += *p;
    const int exp10 = std::atoi(p + 1);
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    std::string out = neg ? "-" : "";
    if (exp10 >= -4 && exp10 < 16) {
        if (exp10 < 0) out += "0." + std::string((size_t)(-exp10 - 1), '0') + digits;
        else if ((int)digits.size() <= exp10 + 1) out += digits + std::string((size_t)(exp10 + 1 - (int)digits.size()), '0') + ".0";
        else out += digits.substr(0, (size_t)exp10 + 1) + "." + digits.substr((size_t)exp10 + 1);
    } else {
        out += digits.substr(0, 1);
        if (digits.size() > 1) out += "." + digits.substr(1);
        char e[8];
        std::snprintf(e, sizeof e, "e%c%02d", exp10 < 0 ? '-' : '+', std::abs(exp10));
        out += e;
    }
    return out;
}
inline std::string jsonString(const std::string& v) {
    std::string out = "\"";
    for (const unsigned char c : v) {
        switch (c) {
            case '"': out += "\\\""; break; case '\\': out += "\\\\"; break; case '\b': out += "\\b"; break; case '\f': out += "\\f"; break;
            case '\n': out += "\\n"; break; case '\r': out += "\\r"; break; case '\t': out += "\\t"; break;
            default:
                if (c < 0x20) { char u[8]; std::snprintf(u, sizeof u, "\\u%04x", c); out += u; }
                else out += (char)c;      // UTF-8 passes through, as nlohmann's dump() leaves it
        }
    }
    return out + "\"";
}

// One serialised member: how it is written and how it is taken from a parsed value.
struct SettingsField {
    std::string key;
    std::function<std::string(const Project&)> write;
    std::function<void(Project&, const JsonValue&)> read;
};
namespace settings_detail {
[[noreturn]] inline void wrongType(const std::string& key, const char* want) { throw std::runtime_error("settings.json: \"" + key + "\" must be " + want); }
inline double numberOf(const std::string& key, const JsonValue& v) { if (v.kind != JsonValue::Number) wrongType(key, "a number"); return v.number; }
inline SettingsField field(const char* key, float Project::*m) {
    return { key, [m](const Project& p) { return jsonNumber((double)(p.*m)); }, [m, k = std::string(key)](Project& p, const JsonValue& v) { p.*m = (float)numberOf(k, v); } };
}
inline SettingsField field(const char* key, int Project::*m) {
    return { key, [m](const Project& p) { return std::to_string(p.*m); }, [m, k = std::string(key)](Project& p, const JsonValue& v) { p.*m = (int)numberOf(k, v); } };
}
inline SettingsField field(const char* key, bool Project::*m) {
    return { key, [m](const Project& p) { return std::string(p.*m ? "true" : "false"); },
             [m, k = std::string(key)](Project& p, const JsonValue& v) { if (v.kind != JsonValue::Bool) wrongType(k, "true or false"); p.*m = v.boolean; } };
}
inline SettingsField field(const char* key, std::string Project::*m) {
    return { key, [m](const Project& p) { return jsonString(p.*m); },
             [m, k = std::string(key)](Project& p, const JsonValue& v) { if (v.kind != JsonValue::String) wrongType(k, "a string"); p.*m = v.text; } };
}
inline SettingsField field(const char* key, CameraSphere Project::*m) {     // NLOHMANN_DEFINE_TYPE_INTRUSIVE_WITH_DEFAULT(CameraSphere, ...), src/Project.h:22
    return { key,
             [m](const Project& p) {
                 const CameraSphere& c = p.*m;     // keys sorted, like the outer object
                 return "{\"count\":" + std::to_string(c.count) + ",\"distance\":" + jsonNumber((double)c.distance) + ",\"fovDeg\":" + jsonNumber((double)c.fovDeg) +
                        ",\"rotX\":" + jsonNumber((double)c.rotX) + ",\"rotY\":" + jsonNumber((double)c.rotY) + "}";
             },
             [m, k = std::string(key)](Project& p, const JsonValue& v) {
                 if (v.kind != JsonValue::Object) wrongType(k, "an object");
                 CameraSphere c;      // members the file lacks: CameraSphere's defaults
                 if (const JsonValue* x = v.find("count")) c.count = (int)numberOf(k + ".count", *x);
                 if (const JsonValue* x = v.find("distance")) c.distance = (float)numberOf(k + ".distance", *x);
                 if (const JsonValue* x = v.find("fovDeg")) c.fovDeg = (float)numberOf(k + ".fovDeg", *x);
                 if (const JsonValue* x = v.find("rotX")) c.rotX = (float)numberOf(k + ".rotX", *x);
                 if (const JsonValue* x = v.find("rotY")) c.rotY = (float)numberOf(k + ".rotY", *x);
                 p.*m = c;
             } };
}
}  // namespace settings_detail

// src/Project.h:64-73, in the macro's order
inline const std::vector<SettingsField>& settingsSchema() {
    using settings_detail::field;
    static const std::vector<SettingsField> schema = {
        field("perspective", &Project::perspective), field("pathModel", &Project::pathModel), field("pathTextureDiffuse", &Project::pathTextureDiffuse),
        field("sphere1", &Project::sphere1), field("sphere2", &Project::sphere2), field("rtSamples", &Project::rtSamples),
        field("lrLocation", &Project::lrLocation), field("lrSh", &Project::lrSh), field("lrScale", &Project::lrScale), field("lrOpacity", &Project::lrOpacity),
        field("lrRotation", &Project::lrRotation), field("paramScaleMax", &Project::paramScaleMax), field("paramCullOpacity", &Project::paramCullOpacity),
        field("paramCullSize", &Project::paramCullSize), field("paramDensifyVariance", &Project::paramDensifyVariance), field("paramSplitSize", &Project::paramSplitSize),
        field("paramSplitDistance", &Project::paramSplitDistance), field("paramSplitScale", &Project::paramSplitScale), field("paramCloneDistance", &Project::paramCloneDistance),
        field("iterations", &Project::iterations), field("intervalCapture", &Project::intervalCapture), field("intervalDensify", &Project::intervalDensify),
        field("previewTimer", &Project::previewTimer), field("previewRtSamples", &Project::previewRtSamples), field("previewSplatScale", &Project::previewSplatScale),
        field("previewTruth", &Project::previewTruth), field("previewTruthIndex", &Project::previewTruthIndex), field("previewFreeOrbit", &Project::previewFreeOrbit),
        field("previewFreeOrbitSpeed", &Project::previewFreeOrbitSpeed), field("previewFreeDistance", &Project::previewFreeDistance),
        field("previewFreeFovDeg", &Project::previewFreeFovDeg), field("previewFreeRotX", &Project::previewFreeRotX), field("previewFreeRotY", &Project::previewFreeRotY),
        field("renderResX", &Project::renderResX), field("renderResY", &Project::renderResY),
    };
    return schema;
}

// nlohmann::to_json(j, project); file << j   — one compact line, keys in std::map order
inline void writeSettings(std::ostream& os, const Project& project) {
    std::vector<const SettingsField*> sorted;
    for (const SettingsField& f : settingsSchema()) sorted.push_back(&f);
    std::sort(sorted.begin(), sorted.end(), [](const SettingsField* a, const SettingsField* b) { return a->key < b->key; });
    os << '{';
    for (size_t k = 0; k < sorted.size(); k++) os << (k ? "," : "") << jsonString(sorted[k]->key) << ':' << sorted[k]->write(project);
    os << '}';
}
inline void saveSettings(const std::string& path, const Project& project) {
    std::ofstream file(path);
    if (!file) throw std::runtime_error("Failed to open settings file \"" + path + "\" for writing!");
    writeSettings(file, project);
}
// file >> j; nlohmann::from_json(j, project)   — WITH_DEFAULT: every member the file lacks gets a fresh Project's value
inline void readSettings(std::istream& is, Project& project) {
    std::stringstream all;
    all << is.rdbuf();
    const std::string text = all.str();
    const JsonValue root = JsonReader(text).parse();
    if (root.kind != JsonValue::Object) throw std::runtime_error("settings.json: the top level must be an object");
    Project loaded;
    for (const SettingsField& f : settingsSchema())
        if (const JsonValue* v = root.find(f.key)) f.read(loaded, *v);
    project = loaded;
}
// The reference shows a dialog and returns when the file is missing (src/ui/UiFrame.cpp:361-365); without a GUI that is an exception.
inline void loadSettings(const std::string& path, Project& project) {
    std::ifstream file(path);
    if (!file) throw std::runtime_error("Failed to load settings file at \"" + path + "\"!");
    readSettings(file, project);
}

// ---------------------------------------------------------------------------------------------------------
// start fields
// ---------------------------------------------------------------------------------------------------------
struct SplatRecord {
    float location[3];
    float scale[3];
    float opacity;
    float rotation[4];  // the four floats a memcpy of the reference's glm::quat yields (layout: see quatMemory)
};

// glm::angleAxis(angle, axis) as stored by the reference (ModelSplatsHost::pushBack memcpy's the glm::quat object,
// src/ModelSplatsHost.cpp:74).  glm's default member order is {x, y, z, w}; xyzw = false models
// GLM_FORCE_QUAT_DATA_WXYZ builds.  The axis is used as given (glm does not normalise it).
inline void quatMemory(float angle, const float axis[3], bool xyzw, float out[4]) {
    const float half = angle * 0.5f, s = std::sin(half), w = std::cos(half);
    const float v[3] = { axis[0] * s, axis[1] * s, axis[2] * s };
    if (xyzw) { out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = w; }
    else { out[0] = w; out[1] = v[0]; out[2] = v[1]; out[3] = v[2]; }
}

using SplatSink = std::function<void(const SplatRecord&)>;
using SplatSource = std::function<void(const SplatSink&)>;

// A fresh model of the compile-time limits (src/Config.h:17-20) filled by a generator; SH all zero (grey 0.5).
inline std::unique_ptr<ModelSplatsHost> makeField(const SplatSource& source) {
    auto model = std::make_unique<ModelSplatsHost>(SPLATS_LIMIT, SPLATS_SH_DEGREE, SPLATS_SH_COEF);
    const std::vector<float> grey(3 * (size_t)model->shCoeffs, 0.0f);
    source([&](const SplatRecord& r) { model->pushBack(r.location, grey, r.scale, r.opacity, r.rotation); });
    return model;
}

// src/ui/UiFrame.cpp:137-160 — a cubic lattice over [-4, 4]^3 with pitch 0.5 (17^3 splats, z fastest), isotropic
// scale pitch / 10, opaque, identity rotation.  Pitch and extent are exact binary fractions, so indexing the lattice
// gives the same coordinates as the reference's accumulating float loops.
inline std::unique_ptr<ModelSplatsHost> initFieldGrid(bool quatXYZW = true) {
    constexpr float extent = 4.0f, pitch = 0.5f;
    constexpr int perAxis = (int)(2.0f * extent / pitch) + 1;
    return makeField([&](const SplatSink& emit) {
        SplatRecord r{};
        const float up[3] = { 0.0f, 1.0f, 0.0f };
        quatMemory(0.0f, up, quatXYZW, r.rotation);
        r.opacity = 1.0f;
        r.scale[0] = r.scale[1] = r.scale[2] = pitch * 0.1f;
        for (int index = 0; index < perAxis * perAxis * perAxis; index++) {
            const int cell[3] = { index / (perAxis * perAxis), (index / perAxis) % perAxis, index % perAxis };
            for (int a = 0; a < 3; a++) r.location[a] = -extent + pitch * (float)cell[a];
            emit(r);
        }
    });
}

// src/ui/UiFrame.cpp:162-176 — one large splat at the origin
inline std::unique_ptr<ModelSplatsHost> initFieldMono(bool quatXYZW = true) {
    return makeField([&](const SplatSink& emit) {
        SplatRecord r{};
        const float up[3] = { 0.0f, 1.0f, 0.0f };
        quatMemory(0.0f, up, quatXYZW, r.rotation);
        r.opacity = 1.0f;
        r.scale[0] = r.scale[1] = r.scale[2] = 0.3f;
        emit(r);
    });
}

// The reference's OBJ subset (src/ui/UiFrame.cpp:185-232): `v x y z`; `f` with 3 or 4 corners `vi[/...]` (1-based;
// quads are fanned from their first corner); every other line ignored; any other corner count is an error.
struct ObjMesh {
    std::vector<std::array<float, 3>> vertices;
    std::vector<std::array<int, 3>> triangles;
};
inline ObjMesh parseObj(std::istream& is) {
    ObjMesh mesh;
    std::string line;
    while (std::getline(is, line)) {
        std::istringstream words(line);
        std::string kind;
        if (!(words >> kind)) continue;
        if (kind == "v") {
            std::array<float, 3> v{};
            words >> v[0] >> v[1] >> v[2];
            mesh.vertices.push_back(v);
        } else if (kind == "f") {
            std::vector<int> corner;
            for (std::string word; words >> word;) corner.push_back(std::atoi(word.substr(0, word.find('/')).c_str()) - 1);
            if (corner.size() != 3 && corner.size() != 4)
                throw std::runtime_error("Unexpected vertex count in face list!" + std::to_string(corner.size()));
            for (size_t k = 2; k < corner.size(); k++) mesh.triangles.push_back({ corner[0], corner[k - 1], corner[k] });
        }
    }
    return mesh;
}

// src/ui/UiFrame.cpp:234-259 — a triangle becomes a thin splat at its centroid: the two edges leaving corner 0 give
// the in-plane extents, 0.005 the thickness (all x 0.2), and +Z is turned onto the face normal by
// angleAxis(acos(n.z), +Z x n).
inline SplatRecord triangleSplat(const std::array<float, 3>& a, const std::array<float, 3>& b, const std::array<float, 3>& c, bool quatXYZW) {
    SplatRecord r{};
    float e1[3], e2[3];
    for (int k = 0; k < 3; k++) {
        r.location[k] = (a[k] + b[k] + c[k]) / 3.0f;
        e1[k] = b[k] - a[k];
        e2[k] = c[k] - a[k];
    }
    auto length = [](const float v[3]) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); };
    r.scale[0] = length(e1) * 0.2f; r.scale[1] = length(e2) * 0.2f; r.scale[2] = 0.005f * 0.2f;
    float n[3] = { e1[1] * e2[2] - e2[1] * e1[2], e1[2] * e2[0] - e2[2] * e1[0], e1[0] * e2[1] - e2[0] * e1[1] };
    const float inv = 1.0f / length(n);
    for (float& x : n) x *= inv;
    const float axis[3] = { -n[1], n[0], 0.0f };  // (0, 0, 1) x n
    quatMemory(std::acos(n[2]), axis, quatXYZW, r.rotation);
    r.opacity = 1.0f;
    return r;
}

// src/ui/UiFrame.cpp:178-264 — one splat per triangle of an OBJ mesh
inline std::unique_ptr<ModelSplatsHost> initFieldModel(std::istream& obj, bool quatXYZW = true) {
    const ObjMesh mesh = parseObj(obj);
    return makeField([&](const SplatSink& emit) {
        for (const auto& t : mesh.triangles) emit(triangleSplat(mesh.vertices.at(t[0]), mesh.vertices.at(t[1]), mesh.vertices.at(t[2]), quatXYZW));
    });
}
inline std::unique_ptr<ModelSplatsHost> initFieldModel(const std::string& objPath, bool quatXYZW = true) {
    std::ifstream file(objPath);
    if (!file) throw std::runtime_error("Failed to load model file at \"" + objPath + "\"!");
    return initFieldModel(file, quatXYZW);
}

// ---------------------------------------------------------------------------------------------------------------------
// The caller of the path: the "Auto Train" loop of UiFrame::update (src/ui/UiFrame.cpp:266-298) without the GUI — at most
// AUTO_TRAIN_BUDGET = 100 iterations per second (src/Config.h:10; the budget never accumulates beyond one iteration), truth
// re-capture with randomly re-rotated camera spheres every `intervalCapture` iterations (UiPanelToolsTruth::onButtonRandomRotate
// / onButtonCapture, src/ui/tools/UiPanelToolsTruth.cpp:186-197), densify every `intervalDensify` iterations.
// TrainerT needs train(ProjectT&, bool) and captureTruths(cameras, framesWhite, framesBlack) (gsplat_shim::Trainer has both);
// the truth renderer stays the caller's (the reference's is OptiX): capture(cameras, framesWhite, framesBlack) fills one RGBA8
// image per camera and background.  random01: a uniform [0, 1) source (the reference calls rand() / RAND_MAX).
template <class TrainerT, class ProjectT>
class AutoTrainer {
public:
    using Frames = std::vector<std::vector<uint32_t>>;
    using Capture = std::function<void(const std::vector<Camera>&, Frames& white, Frames& black)>;
    static constexpr float kBudgetPerSecond = 100.0f;  // AUTO_TRAIN_BUDGET
    bool autoTraining = true;
    float autoTrainingBudget = 0.0f;

    AutoTrainer(TrainerT& trainerArg, ProjectT& projectArg, Capture captureArg, std::function<float()> random01 = {})
        : trainer(trainerArg), project(projectArg), capture(std::move(captureArg)),
          random(random01 ? std::move(random01) : std::function<float()>([] { return (float)std::rand() / ((float)RAND_MAX + 1.0f); })) {}

    void randomRotate() {  // src/ui/tools/UiPanelToolsTruth.cpp:192-197
        project.sphere1.rotX = random() * 360.0f; project.sphere1.rotY = random() * 360.0f;
        project.sphere2.rotX = random() * 360.0f; project.sphere2.rotY = random() * 360.0f;
    }
    void captureTruths() {  // :186-190 -> Trainer::captureTruths with the project's cameras
        const std::vector<Camera> cameras = Camera::getCameras(project);
        Frames white, black;
        capture(cameras, white, black);
        trainer.captureTruths(cameras, white, black);
    }
    // one iteration: the body of `if(autoTrainingBudget >= 1.0f)` (src/ui/UiFrame.cpp:279-296); returns {captured, densified}
    std::pair<bool, bool> step() {
        const bool cap = project.intervalCapture > 0 && project.iterations % project.intervalCapture == 0;
        const bool densify = project.intervalDensify > 0 && project.iterations % project.intervalDensify == 0;
        if (cap) { randomRotate(); captureTruths(); }
        trainer.train(project, densify);
        return { cap, densify };
    }
    // UiFrame::update (:266-298), called from the host's idle loop with the seconds since the last call; true when an iteration ran
    bool update(float delta) {
        project.previewTimer += delta;
        if (!autoTraining) return false;
        autoTrainingBudget = std::min(1.0f, autoTrainingBudget + delta * kBudgetPerSecond);
        if (autoTrainingBudget < 1.0f) return false;
        autoTrainingBudget = 0.0f;
        step();
        return true;
    }

private:
    TrainerT& trainer;
    ProjectT& project;
    Capture capture;
    std::function<float()> random;
};

}  // namespace gsplat_shim
