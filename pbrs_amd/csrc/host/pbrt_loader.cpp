// pbrs_amd/csrc/host/pbrt_loader.cpp — pbrt-v3 scene front-end: text -> tokens -> syntax tree -> pbrs_scene_spec.
//
// Restates, for the subset the reference supports, scene_parser/src/{token.rs:3-175, lexer.rs:6-60, parser.rs:14-391,
// ast.rs:7-125} and scene/src/loader.rs:41-879 (with scene/src/plyloader.rs:10-256 and texture/src/lib.rs:173-209 for the
// two binary formats it reads).  The output is the same plain-data scene the synthetic builders produce, so everything
// downstream (flattener, kernels, oracle) is shared.  What the reference leaves `unimplemented!()` / `todo!()` / panicking
// (object instancing :781, CoordinateSystem / Transform / ConcatTransform :800, spot and projection lights, `spectrum` colours
// given as numbers :762, ASCII PLY) is reported as an error here instead of aborting.  `blackbody` colours (:763) and metal
// `eta` / `k` from `.spd` files (:548-570, :858-879) go through host/spectrum.h (radiometry/src/spectrum.rs, math/src/spline.rs).
// `Shape "loopsubdiv"` IS implemented upstream (:332-379 over shape/src/subdivision.rs:76-219) but is mesh pre-processing outside
// this path: an error here, stated as such.
// The reference holds no tests or scene files for this layer ("parity unpinned"); tests/test_pbrt_loader.py checks it
// against scenes assembled directly through the spec.
#include <zlib.h>

#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/pbrs_host.h"
#include "../../../include/pbrs_numeric.h"
#include "spectrum.h"

namespace {

struct LoadError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
[[noreturn]] void fail(const std::string& m) { throw LoadError(m); }

// ---- token.rs / lexer.rs -------------------------------------------------------------------------------------------
struct Tok {
    enum Kind { End, Word, LBracket, RBracket, Number, String } kind = End;
    std::string text;  // keyword or string contents
    float number = 0.0f;
};
const char* const kKeywords[] = {"Include", "LookAt", "Camera", "Integrator", "Accelerator", "Sampler", "Film", "PixelFilter", "Filter",
                                 "WorldBegin", "WorldEnd", "AttributeBegin", "AttributeEnd", "TransformBegin", "TransformEnd",
                                 "LightSource", "AreaLightSource", "Material", "Shape", "Texture", "Identity", "Translate", "Scale",
                                 "Rotate", "CoordinateSystem", "CoordSysTransform", "Transform", "ConcatTransform", "ReverseOrientation",
                                 "MediumInterface", "NamedMedium", "MakeNamedMedium", "NamedMaterial", "MakeNamedMaterial", "ObjectBegin",
                                 "ObjectEnd", "ObjectInstance"};
bool is_keyword(const std::string& w) {
    for (const char* k : kKeywords)
        if (w == k) return true;
    return false;
}
std::string read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) fail("cannot open " + path);
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}
std::string dir_of(const std::string& path) {
    size_t k = path.find_last_of('/');
    return k == std::string::npos ? std::string(".") : path.substr(0, k);
}
// token.rs:3-118.  Numbers are `[-+]?\d+(\.\d*)?` or `[-+]?\.\d+` (no exponents); strings `"[^"\n]+"`; `#` comments to
// the end of the line.  lexer.rs:33-58: `Include "file"` splices the tokens of the named file (relative to the root file).
void lex_file(const std::string& path, std::vector<Tok>& out, int depth) {
    if (depth > 16) fail("Include nesting too deep at " + path);
    const std::string src = read_file(path);
    const std::string root = dir_of(path);
    size_t i = 0;
    bool pending_include = false;
    auto digit = [&](size_t k) { return k < src.size() && src[k] >= '0' && src[k] <= '9'; };
    while (i < src.size()) {
        const char c = src[i];
        if (c == ' ' || c == '\t' || c == '\n' || c == '\f' || c == '\r') {
            ++i;
            continue;
        }
        if (c == '#') {
            while (i < src.size() && src[i] != '\n') ++i;
            continue;
        }
        Tok t;
        if (c == '[') {
            t.kind = Tok::LBracket;
            ++i;
        } else if (c == ']') {
            t.kind = Tok::RBracket;
            ++i;
        } else if (c == '"') {
            size_t e = i + 1;
            while (e < src.size() && src[e] != '"' && src[e] != '\n') ++e;
            if (e >= src.size() || src[e] != '"' || e == i + 1) fail("bad string literal in " + path);
            t.kind = Tok::String;
            t.text = src.substr(i + 1, e - i - 1);
            i = e + 1;
        } else if (digit(i) || ((c == '-' || c == '+' || c == '.') && (digit(i + 1) || (c != '.' && i + 1 < src.size() && src[i + 1] == '.' && digit(i + 2))))) {
            size_t e = i;
            if (src[e] == '-' || src[e] == '+') ++e;
            if (digit(e)) {
                while (digit(e)) ++e;
                if (e < src.size() && src[e] == '.') {
                    ++e;
                    while (digit(e)) ++e;
                }
            } else {  // `.\d+`
                ++e;
                while (digit(e)) ++e;
            }
            t.kind = Tok::Number;
            t.number = std::strtof(src.substr(i, e - i).c_str(), nullptr);  // str::parse::<f32>: correctly rounded, as strtof
            i = e;
        } else {
            size_t e = i;
            while (e < src.size() && ((src[e] >= 'A' && src[e] <= 'Z') || (src[e] >= 'a' && src[e] <= 'z'))) ++e;
            const std::string w = src.substr(i, e - i);
            if (e == i || !is_keyword(w)) fail("unrecognised token '" + src.substr(i, 16) + "' in " + path);
            t.kind = Tok::Word;
            t.text = w;
            i = e;
        }
        if (pending_include) {
            if (t.kind != Tok::String) fail("Include needs a file name");
            lex_file(root + "/" + t.text, out, depth + 1);
            pending_include = false;
        } else if (t.kind == Tok::Word && t.text == "Include") {
            pending_include = true;
        } else {
            out.push_back(t);
        }
    }
    if (pending_include) fail("Include without a file name");
}

// ---- ast.rs --------------------------------------------------------------------------------------------------------
struct Arg {
    enum Kind { Str, Nums, Num } kind = Num;
    std::string s;
    std::vector<float> v;
    float x = 0.0f;
};
struct Params {  // ParameterSet: a map keyed by the full `"type name"` string; a repeated key replaces the earlier value
    std::vector<std::pair<std::string, Arg>> kv;
    void insert(const std::string& k, const Arg& a) {
        for (auto& e : kv)
            if (e.first == k) {
                e.second = a;
                return;
            }
        kv.emplace_back(k, a);
    }
    bool extract(const std::string& key, Arg* out) {
        for (size_t i = 0; i < kv.size(); ++i)
            if (kv[i].first == key) {
                *out = kv[i].second;
                kv.erase(kv.begin() + (long)i);
                return true;
            }
        return false;
    }
    // ast.rs:57-70: the first key that has `pattern` as one of its space-separated parts
    bool extract_substr(const std::string& pattern, std::string* key, Arg* out) {
        for (size_t i = 0; i < kv.size(); ++i) {
            std::istringstream parts(kv[i].first);
            std::string part;
            while (std::getline(parts, part, ' '))
                if (part == pattern) {
                    *key = kv[i].first;
                    *out = kv[i].second;
                    kv.erase(kv.begin() + (long)i);
                    return true;
                }
        }
        return false;
    }
    bool lookup_f32(const std::string& key, float* out) const {  // :18-26
        for (const auto& e : kv)
            if (e.first == key) {
                if (e.second.kind == Arg::Num) *out = e.second.x;
                else if (e.second.kind == Arg::Nums && !e.second.v.empty()) *out = e.second.v[0];
                else return false;
                return true;
            }
        return false;
    }
    bool lookup_string(const std::string& key, std::string* out) const {
        for (const auto& e : kv)
            if (e.first == key && e.second.kind == Arg::Str) {
                *out = e.second.s;
                return true;
            }
        return false;
    }
};
struct Xform {
    enum Kind { Identity, Translate, Scale, Rotate, LookAt, CoordSys } kind = Identity;
    float a[9] = {0};
    float deg = 0.0f;
    std::string name;
};
struct Item {
    enum Kind { Transform, Shape, Material, Light, AreaLight, Texture, AttributeBlock, ObjectBlock, TransformBlock, MakeMaterial,
                MaterialInstance, ObjectInstance, ReverseOrientation } kind = Transform;
    Xform xf;
    std::string impl, tex_type, name;
    Params params;
    std::vector<Item> children;
};
struct Option {
    enum Kind { Camera, Film, Filter, Integrator, Accel, Transform, Sampler } kind = Camera;
    std::string impl;
    Params params;
    Xform xf;
};

// ---- parser.rs -----------------------------------------------------------------------------------------------------
struct Parser {
    const std::vector<Tok>& t;
    size_t p = 0;
    explicit Parser(const std::vector<Tok>& toks) : t(toks) {}
    const Tok& peek() const { return t[p]; }
    void next() {
        if (t[p].kind != Tok::End) ++p;
    }
    bool is_word(const char* w) const { return peek().kind == Tok::Word && peek().text == w; }
    void expect_word(const char* w) {
        if (!is_word(w)) fail(std::string("expected ") + w);
        next();
    }
    std::string quoted() {
        if (peek().kind != Tok::String) fail("expected quoted string");
        std::string s = peek().text;
        next();
        return s;
    }
    std::vector<float> numbers() {
        std::vector<float> v;
        while (peek().kind == Tok::Number) {
            v.push_back(peek().number);
            next();
        }
        return v;
    }
    static bool starts_transform(const Tok& k) {  // token.rs:122-137
        static const char* const w[] = {"Identity", "Translate", "Scale", "Rotate", "LookAt", "Transform", "ConcatTransform", "CoordSysTransform",
                                        "CoordinateSystem"};
        if (k.kind != Tok::Word) return false;
        for (const char* x : w)
            if (k.text == x) return true;
        return false;
    }
    static bool starts_scene_option(const Tok& k) {  // :139-151
        static const char* const w[] = {"Camera", "Sampler", "Film", "Filter", "Integrator", "Accelerator", "LookAt"};
        if (k.kind != Tok::Word) return false;
        for (const char* x : w)
            if (k.text == x) return true;
        return starts_transform(k);
    }
    static bool starts_world_item(const Tok& k) {  // :153-171
        static const char* const w[] = {"AttributeBegin", "ObjectBegin", "TransformBegin", "Shape", "LightSource", "AreaLightSource", "Material",
                                        "Texture", "MakeNamedMaterial", "NamedMaterial", "ObjectInstance", "NamedMedium", "MakeNamedMedium",
                                        "ReverseOrientation"};
        if (k.kind != Tok::Word) return false;
        for (const char* x : w)
            if (k.text == x) return true;
        return starts_transform(k);
    }
    Params parameter_list() {  // :212-257
        Params ps;
        while (peek().kind == Tok::String) {
            const std::string key = quoted();
            Arg a;
            if (peek().kind == Tok::LBracket) {
                next();
                if (peek().kind == Tok::Number) {
                    std::vector<float> v = numbers();
                    if (v.size() == 1) {
                        a.kind = Arg::Num;
                        a.x = v[0];
                    } else {
                        a.kind = Arg::Nums;
                        a.v = v;
                    }
                } else if (peek().kind == Tok::String) {
                    a.kind = Arg::Str;
                    a.s = quoted();
                } else {
                    fail("only numbers or quoted strings allowed in [ ]");
                }
                if (peek().kind != Tok::RBracket) fail("expected ]");
                next();
            } else if (peek().kind == Tok::String) {
                a.kind = Arg::Str;
                a.s = quoted();
            } else if (peek().kind == Tok::Number) {
                a.kind = Arg::Num;
                a.x = peek().number;
                next();
            } else {
                fail("unexpected token after parameter name " + key);
            }
            ps.insert(key, a);
        }
        return ps;
    }
    Xform transform() {  // :259-318
        const std::string kw = peek().text;
        next();
        Xform x;
        if (kw == "Identity") {
            x.kind = Xform::Identity;
        } else if (kw == "Translate" || kw == "Scale") {
            std::vector<float> v = numbers();
            if (v.size() != 3) fail("wrong number of numbers after " + kw);
            x.kind = kw == "Translate" ? Xform::Translate : Xform::Scale;
            std::memcpy(x.a, v.data(), 3 * sizeof(float));
        } else if (kw == "Rotate") {
            std::vector<float> v = numbers();
            if (v.size() != 4) fail("Rotate needs an angle and an axis");
            x.kind = Xform::Rotate;
            x.deg = v[0];
            std::memcpy(x.a, v.data() + 1, 3 * sizeof(float));
        } else if (kw == "LookAt") {
            std::vector<float> v = numbers();
            if (v.size() != 9) fail("wrong numbers of floats in LookAt");
            x.kind = Xform::LookAt;
            std::memcpy(x.a, v.data(), 9 * sizeof(float));
        } else if (kw == "CoordSysTransform") {
            x.kind = Xform::CoordSys;
            x.name = quoted();
        } else {
            fail(kw + " is not implemented by the reference (parser.rs:309-311)");
        }
        return x;
    }
    std::vector<Item> world_items() {
        std::vector<Item> items;
        while (starts_world_item(peek())) items.push_back(world_item());
        return items;
    }
    Item world_item() {  // :36-160
        Item it;
        if (starts_transform(peek())) {
            it.kind = Item::Transform;
            it.xf = transform();
            return it;
        }
        const std::string kw = peek().text;
        next();
        if (kw == "Shape" || kw == "Material" || kw == "LightSource" || kw == "AreaLightSource") {
            it.kind = kw == "Shape" ? Item::Shape : kw == "Material" ? Item::Material : kw == "LightSource" ? Item::Light : Item::AreaLight;
            it.impl = quoted();
            it.params = parameter_list();
        } else if (kw == "Texture") {
            it.kind = Item::Texture;
            it.name = quoted();
            it.tex_type = quoted();
            it.impl = quoted();
            it.params = parameter_list();
        } else if (kw == "MakeNamedMaterial") {
            it.kind = Item::MakeMaterial;
            it.name = quoted();
            it.params = parameter_list();
        } else if (kw == "ObjectInstance") {
            it.kind = Item::ObjectInstance;
            it.name = quoted();
        } else if (kw == "AttributeBegin") {
            if (is_word("AttributeEnd")) {
                it.kind = Item::AttributeBlock;  // parser.rs:104-105: an empty block; the AttributeEnd is left for the caller
            } else if (is_word("ObjectBegin")) {
                next();
                it.kind = Item::ObjectBlock;
                it.name = quoted();
                it.children = world_items();
                if (is_word("AttributeEnd")) {
                    next();
                    expect_word("ObjectEnd");
                } else if (is_word("ObjectEnd")) {
                    next();
                    expect_word("AttributeEnd");
                } else {
                    fail("unexpected token before ending building an object");
                }
            } else {
                it.kind = Item::AttributeBlock;
                it.children = world_items();
                expect_word("AttributeEnd");
            }
        } else if (kw == "ObjectBegin") {
            it.kind = Item::ObjectBlock;
            it.name = quoted();
            it.children = world_items();
            expect_word("ObjectEnd");
        } else if (kw == "TransformBegin") {
            it.kind = Item::TransformBlock;
            it.children = world_items();
            expect_word("TransformEnd");
        } else if (kw == "NamedMaterial") {
            it.kind = Item::MaterialInstance;
            it.name = quoted();
        } else if (kw == "ReverseOrientation") {
            it.kind = Item::ReverseOrientation;
        } else {
            fail("invalid token for starting an item: " + kw);
        }
        return it;
    }
    Option scene_option() {  // :170-210
        Option o;
        if (starts_transform(peek())) {
            o.kind = Option::Transform;
            o.xf = transform();
            return o;
        }
        const std::string kw = peek().text;
        next();
        o.kind = kw == "Camera" ? Option::Camera : kw == "Sampler" ? Option::Sampler : kw == "Film" ? Option::Film : kw == "Filter" ? Option::Filter
                 : kw == "Integrator" ? Option::Integrator : Option::Accel;
        o.impl = quoted();
        o.params = parameter_list();
        return o;
    }
    void scene(std::vector<Option>& options, std::vector<Item>& items) {  // :22-34
        while (starts_scene_option(peek())) options.push_back(scene_option());
        expect_word("WorldBegin");
        items = world_items();
        expect_word("WorldEnd");
    }
};

// ---- math the loader needs (math/src/hcm.rs, geometry/src/transform.rs) -----------------------------------------------
struct V3 {
    float x, y, z;
};
struct M4 {
    float m[16];  // m[4 * col + row], as pbrs_instance_spec
};
M4 m4_identity() {
    M4 r{};
    r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f;
    return r;
}
M4 m4_mul(const M4& a, const M4& b) {  // hcm.rs:546-556: column c of the product is a * b.cols[c], summed left to right
    M4 r{};
    for (int c = 0; c < 4; ++c)
        for (int row = 0; row < 4; ++row) {
            float acc = a.m[row] * b.m[4 * c];
            for (int k = 1; k < 4; ++k) acc = acc + a.m[4 * k + row] * b.m[4 * c + k];
            r.m[4 * c + row] = acc;
        }
    return r;
}
M4 m4_transpose(const M4& a) {
    M4 r{};
    for (int c = 0; c < 4; ++c)
        for (int row = 0; row < 4; ++row) r.m[4 * c + row] = a.m[4 * row + c];
    return r;
}
V3 m4_point(const M4& a, V3 p) {  // Mat4 * (x, y, z, 1), hcm.rs:539-544
    V3 r;
    r.x = a.m[0] * p.x + a.m[4] * p.y + a.m[8] * p.z + a.m[12] * 1.0f;
    r.y = a.m[1] * p.x + a.m[5] * p.y + a.m[9] * p.z + a.m[13] * 1.0f;
    r.z = a.m[2] * p.x + a.m[6] * p.y + a.m[10] * p.z + a.m[14] * 1.0f;
    return r;
}
V3 m4_vector(const M4& a, V3 v) {
    V3 r;
    r.x = a.m[0] * v.x + a.m[4] * v.y + a.m[8] * v.z;
    r.y = a.m[1] * v.x + a.m[5] * v.y + a.m[9] * v.z;
    r.z = a.m[2] * v.x + a.m[6] * v.y + a.m[10] * v.z;
    return r;
}
struct Affine {  // AffineTransform {forward, inverse}, transform.rs:16-19
    M4 fwd, inv;
};
Affine affine_identity() { return {m4_identity(), m4_identity()}; }
Affine affine_mul(const Affine& a, const Affine& b) { return {m4_mul(a.fwd, b.fwd), m4_mul(b.inv, a.inv)}; }  // :185-194
Affine translater(V3 t) {                                                                                     // :140-145
    Affine r = affine_identity();
    r.fwd.m[12] = t.x; r.fwd.m[13] = t.y; r.fwd.m[14] = t.z;
    r.inv.m[12] = -t.x; r.inv.m[13] = -t.y; r.inv.m[14] = -t.z;
    return r;
}
Affine scaler(V3 s) {  // :159-166
    Affine r = affine_identity();
    r.fwd.m[0] = s.x; r.fwd.m[5] = s.y; r.fwd.m[10] = s.z;
    r.inv.m[0] = 1.0f / s.x; r.inv.m[5] = 1.0f / s.y; r.inv.m[10] = 1.0f / s.z;
    return r;
}
Affine rotater(V3 axis, float angle) {  // :146-152 with Mat3::rotater (hcm.rs:409-421)
    float s, c;
    pn_sincos(angle, &s, &c);
    const float aa = axis.x * axis.x + axis.y * axis.y + axis.z * axis.z;
    const float inv_len = 1.0f / pn_sqrt(aa);
    const V3 ahat{axis.x * inv_len, axis.y * inv_len, axis.z * inv_len};
    Affine r = affine_identity();
    for (int i = 0; i < 3; ++i) {
        V3 base{i == 0 ? 1.0f : 0.0f, i == 1 ? 1.0f : 0.0f, i == 2 ? 1.0f : 0.0f};
        const float d = base.x * axis.x + base.y * axis.y + base.z * axis.z;
        const V3 vc{d * axis.x / aa, d * axis.y / aa, d * axis.z / aa};
        const V3 v1{base.x - vc.x, base.y - vc.y, base.z - vc.z};
        const V3 v2{v1.y * ahat.z - v1.z * ahat.y, v1.z * ahat.x - v1.x * ahat.z, v1.x * ahat.y - v1.y * ahat.x};
        r.fwd.m[4 * i] = vc.x + v1.x * c + v2.x * s;
        r.fwd.m[4 * i + 1] = vc.y + v1.y * c + v2.y * s;
        r.fwd.m[4 * i + 2] = vc.z + v1.z * c + v2.z * s;
    }
    r.inv = m4_transpose(r.fwd);
    return r;
}

// ---- binary inputs ---------------------------------------------------------------------------------------------------
struct RawMesh {
    std::vector<float> positions, normals, uvs;  // 3, 3, 2 per vertex
    std::vector<uint32_t> indices;               // 3 per triangle
};
// geometry/src/lib.rs:16-32
void compute_normals(RawMesh& m) {
    const size_t nv = m.positions.size() / 3;
    std::vector<float> n(3 * nv, 0.0f);
    for (size_t t = 0; t + 2 < m.indices.size(); t += 3) {
        const uint32_t i = m.indices[t], j = m.indices[t + 1], k = m.indices[t + 2];
        const float* p0 = &m.positions[3 * i];
        const float* p1 = &m.positions[3 * j];
        const float* p2 = &m.positions[3 * k];
        const float e1[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]}, e2[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
        const float c[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        for (uint32_t v : {i, j, k})
            for (int a = 0; a < 3; ++a) n[3 * v + a] += c[a];
    }
    for (size_t v = 0; v < nv; ++v) {
        const float inv = 1.0f / pn_sqrt(n[3 * v] * n[3 * v] + n[3 * v + 1] * n[3 * v + 1] + n[3 * v + 2] * n[3 * v + 2]);
        for (int a = 0; a < 3; ++a) n[3 * v + a] *= inv;  // Vec3::hat
    }
    m.normals = n;
}
// scene/src/plyloader.rs:69-256: binary PLY, float vertex properties x y z [nx ny nz] [u v], faces as index lists
// (triangles, or fans of larger polygons)
RawMesh load_ply(const std::string& path) {
    const std::string data = read_file(path);
    size_t pos = 0;
    auto line = [&]() {
        size_t e = data.find('\n', pos);
        if (e == std::string::npos) fail("PLY header ends early: " + path);
        std::string l = data.substr(pos, e - pos);
        pos = e + 1;
        while (!l.empty() && (l.back() == '\r' || l.back() == ' ')) l.pop_back();
        return l;
    };
    if (line() != "ply") fail("not a PLY file: " + path);
    const std::string fmt = line();
    bool big = false;
    if (fmt.rfind("format binary_little_endian", 0) == 0) big = false;
    else if (fmt.rfind("format binary_big_endian", 0) == 0) big = true;
    else fail("PLY format not supported (the reference reads binary PLY only): " + fmt);
    std::vector<std::string> props;
    size_t nv = 0, nf = 0, len_size = 0, idx_size = 0;
    auto type_size = [](const std::string& t) -> size_t {
        if (t == "uchar" || t == "uint8") return 1;
        if (t == "short") return 2;
        if (t == "int" || t == "uint") return 4;
        return 0;
    };
    for (;;) {
        const std::string l = line();
        if (l == "end_header") break;
        if (l.rfind("comment", 0) == 0) continue;
        std::istringstream ws(l);
        std::vector<std::string> w;
        std::string x;
        while (ws >> x) w.push_back(x);
        if (w.size() == 3 && w[0] == "element" && w[1] == "vertex") nv = std::stoul(w[2]);
        else if (w.size() == 3 && w[0] == "element" && w[1] == "face") nf = std::stoul(w[2]);
        else if (w.size() == 3 && w[0] == "property" && w[1] == "float") props.push_back(w[2]);
        else if (w.size() == 5 && w[0] == "property" && w[1] == "list" && w[4] == "vertex_indices") {
            len_size = type_size(w[2]);
            idx_size = type_size(w[3]);
        }
    }
    if (nv == 0 || nf == 0 || len_size == 0 || idx_size == 0 || props.empty()) fail("PLY header lacks vertices, faces or the index list: " + path);
    auto rd = [&](size_t bytes) -> uint32_t {
        if (pos + bytes > data.size()) fail("PLY data ends early: " + path);
        uint32_t v = 0;
        for (size_t b = 0; b < bytes; ++b) {
            const uint32_t byte = (uint8_t)data[pos + b];
            v |= big ? byte << (8 * (bytes - 1 - b)) : byte << (8 * b);
        }
        pos += bytes;
        return v;
    };
    const size_t stride = props.size();
    std::vector<float> vb(nv * stride);
    for (float& f : vb) f = pn_from_bits(rd(4));
    RawMesh m;
    for (size_t f = 0; f < nf; ++f) {
        const uint32_t n = rd(len_size);
        std::vector<uint32_t> face(n);
        for (uint32_t& i : face) i = rd(idx_size);
        for (uint32_t i : face)
            if (i >= nv) fail("PLY face index out of range: " + path);
        if (n == 3) {
            m.indices.insert(m.indices.end(), face.begin(), face.end());
        } else {
            for (uint32_t i = 1; i + 1 < n; ++i) m.indices.insert(m.indices.end(), {face[0], face[i], face[i + 1]});
        }
    }
    int off[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    const char* names[8] = {"x", "y", "z", "nx", "ny", "nz", "u", "v"};
    for (size_t i = 0; i < stride; ++i)
        for (int k = 0; k < 8; ++k)
            if (props[i] == names[k]) off[k] = (int)i;
    if (off[0] < 0 || off[1] < 0 || off[2] < 0) fail("PLY vertices lack x y z: " + path);
    const bool has_n = off[3] >= 0 && off[4] >= 0 && off[5] >= 0, has_uv = off[6] >= 0 && off[7] >= 0;
    for (size_t v = 0; v < nv; ++v) {
        const float* b = &vb[v * stride];
        m.positions.insert(m.positions.end(), {b[off[0]], b[off[1]], b[off[2]]});
        if (has_n) m.normals.insert(m.normals.end(), {b[off[3]], b[off[4]], b[off[5]]});
        if (has_uv) m.uvs.insert(m.uvs.end(), {b[off[6]], b[off[7]]});
    }
    if (!has_n) compute_normals(m);
    if (!has_uv) m.uvs.assign(2 * nv, 0.0f);
    return m;
}

// texture/src/lib.rs:179-209 through the `png` crate: 8-bit grey / RGB / RGBA, non-interlaced -> colours in [0, 1]
struct RawImage {
    uint32_t w = 0, h = 0;
    std::vector<float> rgb;
};
RawImage load_png(const std::string& path) {
    const std::string d = read_file(path);
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (d.size() < 8 || std::memcmp(d.data(), sig, 8) != 0) fail("not a PNG file: " + path);
    auto be32 = [&](size_t p) { return ((uint32_t)(uint8_t)d[p] << 24) | ((uint32_t)(uint8_t)d[p + 1] << 16) | ((uint32_t)(uint8_t)d[p + 2] << 8) | (uint8_t)d[p + 3]; };
    size_t p = 8;
    uint32_t w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
    std::string idat;
    while (p + 12 <= d.size()) {
        const uint32_t len = be32(p);
        const std::string type = d.substr(p + 4, 4);
        if (p + 12 + len > d.size()) fail("truncated PNG: " + path);
        if (type == "IHDR") {
            w = be32(p + 8);
            h = be32(p + 12);
            depth = (uint8_t)d[p + 16];
            ctype = (uint8_t)d[p + 17];
            interlace = (uint8_t)d[p + 20];
        } else if (type == "IDAT") {
            idat.append(d, p + 8, len);
        } else if (type == "IEND") {
            break;
        }
        p += 12 + len;
    }
    if (depth != 8) fail("non 8-bit image: " + path);  // lib.rs:188-190
    uint32_t ch = 0;
    if (ctype == 0) ch = 1;
    else if (ctype == 2) ch = 3;
    else if (ctype == 6) ch = 4;
    else fail("unhandled PNG colour type (indexed / grey-alpha): " + path);  // :196-197
    if (interlace) fail("interlaced PNG not supported: " + path);
    if (w == 0 || h == 0) fail("empty PNG: " + path);
    std::vector<unsigned char> raw((size_t)h * (1 + (size_t)w * ch));
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, reinterpret_cast<const Bytef*>(idat.data()), (uLong)idat.size()) != Z_OK || out_len != raw.size())
        fail("PNG data does not inflate: " + path);
    const size_t row_bytes = (size_t)w * ch;
    std::vector<unsigned char> img((size_t)h * row_bytes), zero(row_bytes, 0);
    for (uint32_t y = 0; y < h; ++y) {
        const unsigned char* in = &raw[(size_t)y * (1 + row_bytes)];
        unsigned char* cur = &img[(size_t)y * row_bytes];
        const unsigned char* up = y ? &img[(size_t)(y - 1) * row_bytes] : zero.data();
        const unsigned filter = in[0];
        for (size_t x = 0; x < row_bytes; ++x) {
            const int a = x >= ch ? cur[x - ch] : 0, b = up[x], c = x >= ch ? up[x - ch] : 0;
            int pred = 0;
            switch (filter) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) / 2; break;
                case 4: {
                    const int pp = a + b - c, pa = std::abs(pp - a), pb = std::abs(pp - b), pc = std::abs(pp - c);
                    pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                    break;
                }
                default: fail("bad PNG filter: " + path);
            }
            cur[x] = (unsigned char)(in[1 + x] + pred);
        }
    }
    RawImage r;
    r.w = w;
    r.h = h;
    r.rgb.resize((size_t)w * h * 3);
    for (size_t k = 0; k < (size_t)w * h; ++k) {
        const unsigned char* px = &img[k * ch];
        for (int a = 0; a < 3; ++a) r.rgb[3 * k + a] = (float)px[ch == 1 ? 0 : a] / 255.0f;  // Color::gray(u8 / 255) / Color::rgb
    }
    return r;
}

}  // namespace

// ---- geometry/src/fourier.rs:55-96, :167-221: a `.bsdf` file (SCATFUN, version 1) ---------------------------------------
struct RawBsdf {
    std::vector<float> mu, cdf, a;
    std::vector<int32_t> offset_and_length;
    uint32_t n_channels = 0;
    float eta = 1.0f;
};
static std::unique_ptr<RawBsdf> load_bsdf(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) fail("can't open the BSDF file " + path);
    unsigned char h[64];
    if (!f.read(reinterpret_cast<char*>(h), 64)) fail("BSDF file shorter than its header: " + path);
    auto i32_at = [&](size_t at) {
        int32_t v;
        std::memcpy(&v, h + at, 4);
        return v;
    };
    // read_header's asserts (:84-91): identifier, version 1, flags 1 (a BSDF, no harmonic extrapolation), finite eta / alpha
    if (std::memcmp(h, "SCATFUN", 7) != 0 || h[7] != 1 || i32_at(8) != 1) fail("not a version-1 SCATFUN BSDF file: " + path);
    const int32_t n_mu = i32_at(12), n_coeffs = i32_at(16), n_channels = i32_at(24);
    float eta, alpha[2];
    std::memcpy(&eta, h + 44, 4);
    std::memcpy(alpha, h + 48, 8);
    if (!std::isfinite(eta) || !std::isfinite(alpha[0]) || !std::isfinite(alpha[1])) fail("BSDF header holds a non-finite eta or alpha: " + path);
    if (n_mu < 3 || n_mu > 4096 || n_coeffs < 0 || (n_channels != 1 && n_channels != 3)) fail("BSDF header sizes are out of range: " + path);
    const size_t nn = (size_t)n_mu * (size_t)n_mu;
    {
        // the header is not trusted with memory: the file must be as long as it claims before anything is sized from it
        // (n_coeffs alone may ask for 8 GiB; a 64-byte file must not be able to exhaust the host)
        const uint64_t need = 64ull + 4ull * (uint64_t)n_mu + 12ull * (uint64_t)nn + 4ull * (uint64_t)n_coeffs;
        f.seekg(0, std::ios::end);
        const std::streamoff size = f.tellg();
        if (size < 0 || (uint64_t)size < need) fail("BSDF file is truncated: its header announces " + std::to_string(need) + " bytes: " + path);
        f.seekg(64, std::ios::beg);
    }
    auto b = std::make_unique<RawBsdf>();
    b->n_channels = (uint32_t)n_channels;
    b->eta = eta;
    b->mu.resize((size_t)n_mu);
    b->cdf.resize(nn);
    b->offset_and_length.resize(2 * nn);
    b->a.resize((size_t)n_coeffs);
    auto read_n = [&](void* dst, size_t bytes) {
        if (bytes && !f.read(reinterpret_cast<char*>(dst), (std::streamsize)bytes)) fail("BSDF file is truncated: " + path);
    };
    read_n(b->mu.data(), 4 * b->mu.size());
    read_n(b->cdf.data(), 4 * nn);
    read_n(b->offset_and_length.data(), 8 * nn);
    read_n(b->a.data(), 4 * b->a.size());
    return b;
}

// ---- loader.rs -------------------------------------------------------------------------------------------------------
struct pbrs_loaded_scene {
    pbrs_scene_spec spec{};
    std::vector<pbrs_mesh_spec> meshes;
    std::vector<std::unique_ptr<RawMesh>> mesh_data;
    std::vector<pbrs_shape_spec> shapes;
    std::vector<pbrs_material_spec> materials;
    std::vector<pbrs_instance_spec> instances;
    std::vector<pbrs_area_light_spec> area_lights;
    std::vector<pbrs_delta_light_spec> delta_lights;
    std::vector<pbrs_texture_spec> textures;
    std::vector<std::unique_ptr<RawImage>> images;
    std::vector<pbrs_fourier_table_spec> fourier_tables;
    std::vector<std::unique_ptr<RawBsdf>> bsdf_data;
    std::string filter;  // parsed, not used by this path
};

namespace {

struct Color {
    float r, g, b;
};
Color gray(float x) { return {x, x, x}; }

struct Loader {
    pbrs_loaded_scene& out;
    std::string root;
    std::vector<Affine> ctm{affine_identity()};
    int current_mtl = -1;
    bool has_arealight = false;
    Color arealight{0, 0, 0};
    std::map<std::string, int> named_textures, named_materials;
    bool have_env = false;

    Loader(pbrs_loaded_scene& o, std::string root_dir) : out(o), root(std::move(root_dir)) {}

    static Color constant_color(const std::string& key, const std::vector<float>& nums) {  // :758-766
        const std::string type = key.substr(0, key.find(' '));
        if (type == "blackbody") {  // `radiometry::spectrum::temperature_to_color(nums[0]) * nums[1]` (:763)
            if (nums.size() < 2) fail("colour parameter '" + key + "' needs a temperature and a scale");
            const spectrum::Rgb c = spectrum::temperature_to_color(nums[0]);
            return {c.r * nums[1], c.g * nums[1], c.b * nums[1]};
        }
        if (nums.size() < 3) fail("colour parameter '" + key + "' needs three numbers");
        if (type == "rgb" || type == "color") return {nums[0], nums[1], nums[2]};
        if (type == "xyz")  // Color::from_xyz, radiometry/src/color.rs:30-36
            return {3.240479f * nums[0] - 1.537150f * nums[1] - 0.498535f * nums[2], -0.969256f * nums[0] + 1.875991f * nums[1] + 0.041556f * nums[2],
                    0.055648f * nums[0] - 0.204043f * nums[1] + 1.057311f * nums[2]};
        if (type == "spectrum") fail("a spectrum colour given as numbers is `unimplemented!()` in the reference (scene/src/loader.rs:762)");
        fail("unrecognized spectrum type '" + type + "'");  // the reference panics (:764)
    }
    // color_from_spd_file (:858-879): lines of `lambda value` split at single spaces, `#` comments; then
    // sampled_spectrum_to_color.  What `.parse::<f32>().unwrap()` / the asserts would stop is an error.
    static Color color_from_spd_file(const std::string& path) {
        std::ifstream f(path);
        if (!f) fail("can't open the SPD file " + path);
        std::vector<std::pair<float, float>> samples;
        std::string line;
        while (std::getline(f, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();  // BufRead::lines strips "\r\n" too
            size_t b = 0;
            while (b < line.size() && std::isspace((unsigned char)line[b])) ++b;
            if (b < line.size() && line[b] == '#') continue;
            std::vector<float> numbers;
            size_t at = 0;
            for (;;) {  // `content.split(' ')`: every piece must parse, an empty one (two spaces, a blank line) does not
                const size_t sp = line.find(' ', at);
                const std::string piece = line.substr(at, sp == std::string::npos ? std::string::npos : sp - at);
                char* end = nullptr;
                const float v = std::strtof(piece.c_str(), &end);
                if (piece.empty() || end != piece.c_str() + piece.size()) fail("SPD file " + path + ": '" + line + "' is not `lambda value`");
                numbers.push_back(v);
                if (sp == std::string::npos) break;
                at = sp + 1;
            }
            if (numbers.size() < 2) fail("SPD file " + path + ": a line with fewer than two numbers");
            samples.push_back({numbers[0], numbers[1]});
        }
        try {
            const spectrum::Rgb c = spectrum::sampled_spectrum_to_color(samples);
            return {c.r, c.g, c.b};
        } catch (const spectrum::SpectrumError& e) {
            fail("SPD file " + path + ": " + e.what());
        }
    }
    // a colour parameter that is numbers, one number (grey) or absent
    Color color_param(Params& ps, const char* name, Color dflt, const char* what) {
        std::string key;
        Arg a;
        if (!ps.extract_substr(name, &key, &a)) return dflt;
        if (a.kind == Arg::Nums) return constant_color(key, a.v);
        if (a.kind == Arg::Num) return gray(a.x);
        fail(std::string("textured ") + name + " is not supported for " + what + " (as in the reference)");
    }
    float float_param(Params& ps, const char* name, float dflt) {
        std::string key;
        Arg a;
        if (!ps.extract_substr(name, &key, &a)) return dflt;
        if (a.kind != Arg::Num) fail(std::string(name) + " is not a number");
        return a.x;
    }
    bool bool_param(Params& ps, const char* name, bool dflt) {
        std::string key;
        Arg a;
        if (!ps.extract_substr(name, &key, &a)) return dflt;
        if (a.kind != Arg::Str || (a.s != "true" && a.s != "false")) fail(std::string(name) + " is not a boolean string");
        return a.s == "true";
    }
    // solid_or_image_tex (:738-756): colour in `c`, or a texture handle (index + 1) in `tex`
    void solid_or_image(const std::string& key, const Arg& a, Color* c, uint32_t* tex) {
        *tex = 0;
        if (a.kind == Arg::Nums) *c = constant_color(key, a.v);
        else if (a.kind == Arg::Num) *c = gray(a.x);
        else {
            auto it = named_textures.find(a.s);
            if (it == named_textures.end()) fail("texture '" + a.s + "' is not defined");
            *c = gray(0.0f);
            *tex = (uint32_t)it->second + 1u;
        }
    }
    int add_material(const pbrs_material_spec& m) {
        out.materials.push_back(m);
        return (int)out.materials.size() - 1;
    }
    static void put3(float* dst, Color c) {
        dst[0] = c.r;
        dst[1] = c.g;
        dst[2] = c.b;
    }
    int parse_material(const std::string& impl, Params ps) {  // :483-714
        pbrs_material_spec m{};
        std::string key;
        Arg a;
        if (impl == "glass") {
            const Color kr = color_param(ps, "Kr", gray(1.0f), "glass"), kt = color_param(ps, "Kt", gray(1.0f), "glass");
            m.kind = PBRS_MTL_DIELECTRIC;
            m.p[0] = float_param(ps, "eta", 1.5f);
            put3(m.p + 1, kr);
            put3(m.p + 4, kt);
        } else if (impl == "mirror") {
            m.kind = PBRS_MTL_MIRROR;
            put3(m.p, color_param(ps, "Kr", gray(0.9f), "mirror"));
        } else if (impl == "matte") {
            m.kind = PBRS_MTL_LAMBERTIAN;
            Color kd = gray(0.5f);
            if (ps.extract_substr("Kd", &key, &a)) solid_or_image(key, a, &kd, &m.tex[0]);
            put3(m.p, kd);  // `sigma` is read and ignored (:539-544)
        } else if (impl == "metal") {
            m.kind = PBRS_MTL_METAL;
            const float roughness = float_param(ps, "roughness", 0.01f);
            const Color copper_eta{0.195470f, 0.925682f, 1.102186f};  // preset.rs:488-493; the default k also takes `.0` (:559)
            auto ior = [&](const char* name) {
                if (!ps.extract_substr(name, &key, &a)) return copper_eta;
                if (a.kind == Arg::Nums) return constant_color(key, a.v);
                if (a.kind == Arg::Str) return color_from_spd_file(root + "/" + a.s);  // :554-556, :565-567
                fail(std::string("metal ") + name + " is neither numbers nor a file");  // `unimplemented!()` upstream
            };
            put3(m.p, ior("eta"));
            put3(m.p + 3, ior("k"));
            m.p[6] = roughness;
        } else if (impl == "plastic") {
            m.kind = PBRS_MTL_PLASTIC;
            put3(m.p, color_param(ps, "Kd", gray(0.25f), "plastic"));
            put3(m.p + 3, color_param(ps, "Ks", gray(0.25f), "plastic"));
            m.p[6] = float_param(ps, "roughness", 0.1f);
            if (bool_param(ps, "remaproughness", true)) m.flags |= PBRS_MTL_FLAG_REMAP_ROUGHNESS;
        } else if (impl == "uber") {
            m.kind = PBRS_MTL_UBER;
            Color kd = gray(0.25f), ks = gray(0.25f), kr = gray(0.0f), kt = gray(0.0f);
            if (ps.extract_substr("Kd", &key, &a)) solid_or_image(key, a, &kd, &m.tex[0]);
            if (ps.extract_substr("Ks", &key, &a)) solid_or_image(key, a, &ks, &m.tex[1]);
            if (ps.extract_substr("Kr", &key, &a)) {
                solid_or_image(key, a, &kr, &m.tex[2]);
                m.flags |= PBRS_MTL_FLAG_HAS_KR;
            }
            if (ps.extract_substr("Kt", &key, &a)) {
                solid_or_image(key, a, &kt, &m.tex[3]);
                m.flags |= PBRS_MTL_FLAG_HAS_KT;
            }
            const float ur = float_param(ps, "uroughness", 0.0f), vr = float_param(ps, "vroughness", 0.0f);
            const float roughness = float_param(ps, "roughness", 0.0f);
            const float eta = float_param(ps, "eta", 1.5f);
            const float opacity = float_param(ps, "eta", 1.0f);  // :641-645 reads "eta" again: opacity stays 1 unless eta is given twice
            if (bool_param(ps, "remaproughness", true)) m.flags |= PBRS_MTL_FLAG_REMAP_ROUGHNESS;
            put3(m.p, kd);
            put3(m.p + 3, ks);
            put3(m.p + 6, kr);
            put3(m.p + 9, kt);
            m.p[12] = ur == vr ? roughness : ur;  // Roughness::Iso(roughness) / UV((u, v)), :653-657
            m.p[13] = ur == vr ? roughness : vr;
            m.p[14] = eta;
            m.p[15] = opacity;
        } else if (impl == "substrate") {
            m.kind = PBRS_MTL_SUBSTRATE;
            put3(m.p, color_param(ps, "Kd", gray(0.5f), "substrate"));
            put3(m.p + 3, color_param(ps, "Ks", gray(0.5f), "substrate"));
        } else if (impl == "fourier") {  // :705-710, material::Fourier::from_file
            std::string file;
            if (!ps.lookup_string("string bsdffile", &file)) fail("string bsdffile");
            out.bsdf_data.push_back(load_bsdf(root + "/" + file));
            const RawBsdf& b = *out.bsdf_data.back();
            pbrs_fourier_table_spec t{};
            t.n_mu = (uint32_t)b.mu.size();
            t.n_channels = b.n_channels;
            t.n_coeffs = (uint32_t)b.a.size();
            t.eta = b.eta;
            t.mu = b.mu.data();
            t.cdf = b.cdf.data();
            t.offset_and_length = b.offset_and_length.data();
            t.a = b.a.data();
            m.kind = PBRS_MTL_FOURIER;
            m.tex[0] = (uint32_t)out.fourier_tables.size();
            out.fourier_tables.push_back(t);
        } else {
            fail("not recognized material: " + impl);
        }
        return add_material(m);
    }
    pbrs_delta_light_spec parse_light(const std::string& impl, Params ps) {  // :436-481
        pbrs_delta_light_spec d{};
        std::string key;
        Arg a;
        auto point3 = [&](const char* name, V3 dflt) {
            if (!ps.extract_substr(name, &key, &a)) return dflt;
            if (a.kind != Arg::Nums || a.v.size() < 3) fail(std::string("can't parse 3d point '") + name + "'");
            return V3{a.v[0], a.v[1], a.v[2]};
        };
        auto radiance = [&]() {
            if (!ps.extract_substr("L", &key, &a)) return gray(1.0f);
            if (a.kind == Arg::Nums) return constant_color(key, a.v);
            if (a.kind == Arg::Num) return gray(a.x);
            fail("can't parse radiance");
        };
        if (impl == "distant") {
            const V3 from = point3("from", {0, 0, 0}), to = point3("to", {0, 0, 1});
            const Color l = radiance();
            d.kind = PBRS_DELTA_DISTANT;
            d.v[0] = to.x - from.x; d.v[1] = to.y - from.y; d.v[2] = to.z - from.z;
            d.color[0] = l.r; d.color[1] = l.g; d.color[2] = l.b;
            d.world_radius = pn_inf();  // DeltaLight::distant(f32::INFINITY, ..), :456
        } else if (impl == "point") {
            const V3 from = point3("from", {0, 0, 0});
            const Color l = radiance();
            d.kind = PBRS_DELTA_POINT;
            d.v[0] = from.x; d.v[1] = from.y; d.v[2] = from.z;
            d.color[0] = l.r; d.color[1] = l.g; d.color[2] = l.b;
        } else {
            fail("light '" + impl + "' is unimplemented in the reference");
        }
        return d;
    }
    int add_texture_image(const std::string& file) {
        auto img = std::make_unique<RawImage>(load_png(root + "/" + file));
        pbrs_texture_spec t{};
        t.kind = PBRS_TEX_IMAGE;
        t.width = img->w;
        t.height = img->h;
        t.data = img->rgb.data();
        out.images.push_back(std::move(img));
        out.textures.push_back(t);
        return (int)out.textures.size() - 1;
    }
    int add_mesh(std::unique_ptr<RawMesh> m) {
        const size_t nv = m->positions.size() / 3;
        for (uint32_t i : m->indices)
            if (i >= nv) fail("triangle index out of range");
        pbrs_mesh_spec ms{};
        ms.n_vertices = (uint32_t)nv;
        ms.n_triangles = (uint32_t)(m->indices.size() / 3);
        ms.positions = m->positions.data();
        ms.normals = m->normals.data();
        ms.uvs = m->uvs.data();
        ms.indices = m->indices.data();
        out.mesh_data.push_back(std::move(m));
        out.meshes.push_back(ms);
        return (int)out.meshes.size() - 1;
    }
    pbrs_shape_spec parse_shape(const std::string& impl, Params ps) {  // :307-388
        pbrs_shape_spec s{};
        Arg a;
        if (impl == "sphere") {
            float radius = 1.0f;
            ps.lookup_f32("float radius", &radius);
            s.kind = PBRS_SHAPE_SPHERE;
            s.p[3] = radius;
        } else if (impl == "plymesh") {
            std::string file;
            if (!ps.lookup_string("string filename", &file)) fail("no ply file specified");
            s.kind = PBRS_SHAPE_MESH;
            s.mesh = (uint32_t)add_mesh(std::make_unique<RawMesh>(load_ply(root + "/" + file)));
        } else if (impl == "trianglemesh") {
            auto m = std::make_unique<RawMesh>();
            if (!ps.extract("point P", &a) || a.kind != Arg::Nums) fail("missing points");
            m->positions = a.v;
            m->positions.resize(a.v.size() / 3 * 3);
            const size_t nv = m->positions.size() / 3;
            if (ps.extract("float uv", &a) || ps.extract("float st", &a)) {
                if (a.kind != Arg::Nums) fail("incorrect format for uv coords");
                m->uvs = a.v;
            }
            m->uvs.resize(2 * nv, 0.0f);
            if (!ps.extract("integer indices", &a) || a.kind != Arg::Nums) fail("missing indices");
            for (size_t k = 0; k + 2 < a.v.size(); k += 3)
                for (int c = 0; c < 3; ++c) m->indices.push_back((uint32_t)a.v[k + c]);  // `as usize`
            std::string key;
            if (ps.extract_substr("normal", &key, &a)) {
                if (a.kind != Arg::Nums) fail("incorrect format for normals");
                m->normals = a.v;
            }
            m->normals.resize(3 * nv, 0.0f);  // missing normals are zero vectors: every hit falls back to the face normal
            s.kind = PBRS_SHAPE_MESH;
            s.mesh = (uint32_t)add_mesh(std::move(m));
        } else if (impl == "loopsubdiv") {
            fail("loopsubdiv: Loop subdivision is implemented upstream (shape/src/subdivision.rs) but is mesh pre-processing outside this path");
        } else {
            fail("shape of " + impl + " is unimplemented in the reference");
        }
        return s;
    }
    void add_instance(const pbrs_shape_spec& shape, int material, const Affine& t) {
        out.shapes.push_back(shape);
        pbrs_instance_spec in{};
        in.shape = (uint32_t)out.shapes.size() - 1;
        in.material = (uint32_t)material;
        std::memcpy(in.forward, t.fwd.m, sizeof in.forward);
        std::memcpy(in.inverse, t.inv.m, sizeof in.inverse);
        out.instances.push_back(in);
    }
    // SamplableShape::transformed_by (light/src/sample_shape.rs:46-62): uniform scale and rotation of the transform
    static float uniform_scale_of(const Affine& t) {
        const V3 tx = m4_vector(t.fwd, {1, 0, 0}), ty = m4_vector(t.fwd, {0, 1, 0}), tz = m4_vector(t.fwd, {0, 0, 1});
        const V3 c{tx.y * ty.z - tx.z * ty.y, tx.z * ty.x - tx.x * ty.z, tx.x * ty.y - tx.y * ty.x};
        const float scale = cbrtf(c.x * tz.x + c.y * tz.y + c.z * tz.z);
        if (!(scale > 0.0f)) fail("area light under a transform with non-positive scale");
        const float r = 1.0f / scale;
        const float R[3][3] = {{tx.x * r, ty.x * r, tz.x * r}, {tx.y * r, ty.y * r, tz.y * r}, {tx.z * r, ty.z * r, tz.z * r}};  // R[row][col]
        float frob = 0.0f;  // || R R^T - I ||_F^2
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                const float e = R[i][0] * R[j][0] + R[i][1] * R[j][1] + R[i][2] * R[j][2] - (i == j ? 1.0f : 0.0f);
                frob += e * e;
            }
        if (frob > 1e-3f) fail("area light under a transform that is not a uniform scale and a rotation");
        return scale;
    }
    void shape_item(const std::string& impl, Params ps) {  // :171-206
        Arg ignored;
        ps.extract("alpha", &ignored);
        const Affine t = ctm.back();
        if (has_arealight) {
            pbrs_material_spec lm{};
            lm.kind = PBRS_MTL_DIFFUSE_LIGHT;
            put3(lm.p, arealight);
            const int light_mtl = add_material(lm);
            auto light = [&](const pbrs_shape_spec& world_shape) {
                pbrs_area_light_spec al{};
                put3(al.emit, arealight);
                al.shape = world_shape;
                out.area_lights.push_back(al);
            };
            if (impl == "sphere") {  // parse_samplable_shape :399-402
                float radius = 1.0f;
                ps.lookup_f32("float radius", &radius);
                pbrs_shape_spec local{};
                local.kind = PBRS_SHAPE_SPHERE;
                local.p[3] = radius;
                pbrs_shape_spec world = local;
                const V3 c = m4_point(t.fwd, {0, 0, 0});
                world.p[0] = c.x; world.p[1] = c.y; world.p[2] = c.z;
                world.p[3] = radius * uniform_scale_of(t);
                light(world);
                add_instance(local, light_mtl, t);
            } else if (impl == "plymesh") {  // :403-431: one isolated triangle, light and instance, per face
                std::string file;
                if (!ps.lookup_string("string filename", &file)) fail("no ply file specified");
                const RawMesh m = load_ply(root + "/" + file);
                for (size_t k = 0; k + 2 < m.indices.size(); k += 3) {
                    pbrs_shape_spec local{}, world{};
                    local.kind = world.kind = PBRS_SHAPE_TRIANGLE;
                    for (int v = 0; v < 3; ++v) {
                        const float* p = &m.positions[3 * m.indices[k + v]];
                        const V3 w = m4_point(t.fwd, {p[0], p[1], p[2]});
                        local.p[3 * v] = p[0]; local.p[3 * v + 1] = p[1]; local.p[3 * v + 2] = p[2];
                        world.p[3 * v] = w.x; world.p[3 * v + 1] = w.y; world.p[3 * v + 2] = w.z;
                    }
                    light(world);
                    add_instance(local, light_mtl, t);
                }
            } else {
                fail("samplable shape '" + impl + "' is unimplemented in the reference");
            }
        } else if (current_mtl >= 0) {
            add_instance(parse_shape(impl, ps), current_mtl, t);
        } else {
            fail("Shape without a material or an area light in scope");
        }
    }
    static Affine parse_transform(const Xform& x) {  // :786-802
        switch (x.kind) {
            case Xform::Identity: return affine_identity();
            case Xform::Translate: return translater({x.a[0], x.a[1], x.a[2]});
            case Xform::Scale: return scaler({x.a[0], x.a[1], x.a[2]});
            case Xform::Rotate:
                // pbrt-v3 stores the rotation's bases row-major, i.e. applies the inverse: the reference negates the angle
                return rotater({x.a[0], x.a[1], x.a[2]}, -pn_to_radians(x.deg));
            case Xform::LookAt: fail("LookAt in the modelling step is unsupported");
            default: fail("CoordSysTransform is unimplemented in the reference");
        }
    }
    void world_item(const Item& it) {  // :164-305
        switch (it.kind) {
            case Item::Transform: ctm.back() = affine_mul(ctm.back(), parse_transform(it.xf)); break;
            case Item::Shape: shape_item(it.impl, it.params); break;
            case Item::Material: current_mtl = parse_material(it.impl, it.params); break;
            case Item::AttributeBlock:
                ctm.push_back(ctm.back());
                current_mtl = -1;  // the reference clears both on ENTRY and does not restore them on exit (:210-221)
                has_arealight = false;
                for (const Item& c : it.children) world_item(c);
                ctm.pop_back();
                break;
            case Item::TransformBlock:
                ctm.push_back(ctm.back());
                for (const Item& c : it.children) world_item(c);
                ctm.pop_back();
                break;
            case Item::ObjectBlock:
            case Item::ObjectInstance: fail("object instancing is `unimplemented!()` in the reference (loader.rs:768-784)");
            case Item::MakeMaterial: {
                Params ps = it.params;
                Arg type;
                if (!ps.extract("string type", &type) || type.kind != Arg::Str) fail("no material type specified for " + it.name);
                named_materials[it.name] = parse_material(type.s, ps);
                break;
            }
            case Item::Texture: {
                if (it.tex_type != "color" && it.tex_type != "spectrum") break;  // logged and skipped (:253-255)
                if (it.impl != "imagemap") fail("texture implementation '" + it.impl + "' is unimplemented in the reference");
                std::string file;
                if (!it.params.lookup_string("string filename", &file)) fail("missing file name for image map texture");
                named_textures[it.name] = add_texture_image(file);
                break;
            }
            case Item::MaterialInstance: {
                auto f = named_materials.find(it.name);
                current_mtl = f == named_materials.end() ? -1 : f->second;
                break;
            }
            case Item::Light: {
                Params ps = it.params;
                if (it.impl == "infinite") {  // :259-282
                    std::string key, map;
                    Arg a;
                    bool has_l = false;
                    Color l = gray(1.0f);
                    if (ps.extract_substr("L", &key, &a)) {
                        if (a.kind != Arg::Nums) fail("unrecognized luminance in infinite light");
                        l = constant_color(key, a.v);
                        has_l = true;
                    }
                    if (ps.lookup_string("string mapname", &map)) {
                        out.spec.env_kind = PBRS_ENV_IMAGE;
                        out.spec.env_texture = (uint32_t)add_texture_image(map);
                        put3(out.spec.env_scale, l);  // `multiplier.unwrap_or(Color::ONE)`
                    } else if (has_l) {
                        out.spec.env_kind = PBRS_ENV_CONSTANT;
                        put3(out.spec.env_constant, l);
                    } else {
                        fail("can't process the infinite light");
                    }
                } else {
                    out.delta_lights.push_back(parse_light(it.impl, ps));
                }
                break;
            }
            case Item::AreaLight: {
                if (it.impl != "diffuse") break;  // logged and skipped (:300-302)
                Params ps = it.params;
                std::string key;
                Arg a;
                if (!ps.extract_substr("L", &key, &a) || a.kind != Arg::Nums) fail("diffuse area light needs an L colour");
                arealight = constant_color(key, a.v);
                has_arealight = true;
                break;
            }
            case Item::ReverseOrientation: break;  // "unhandled world item" in the reference
        }
    }
    void run(std::vector<Option>& options, const std::vector<Item>& items) {  // traverse_tree :137-162, build_camera :91-135
        bool have_fov = false, have_pose = false;
        float fov_deg = 60.0f, w = 0.0f, h = 0.0f;
        bool have_w = false, have_h = false;
        float pose[9] = {0, 0, 0, 0, 0, 1, 0, 1, 0};
        Affine world = affine_identity();
        for (Option& o : options) {
            if (o.kind == Option::Camera) {
                Arg a;
                have_fov = true;
                fov_deg = 60.0f;
                if (o.params.extract("float fov", &a)) {
                    if (a.kind != Arg::Num) fail("complicated fov degree");
                    fov_deg = a.x;
                }
            } else if (o.kind == Option::Film) {
                have_w = o.params.lookup_f32("integer xresolution", &w);
                have_h = o.params.lookup_f32("integer yresolution", &h);
            } else if (o.kind == Option::Transform && o.xf.kind == Xform::LookAt) {
                std::memcpy(pose, o.xf.a, sizeof pose);
                have_pose = true;
            }
        }
        if (!(have_fov && have_w && have_h)) fail("the scene needs Camera \"perspective\" and Film x/yresolution");
        for (Option& o : options) {
            if (o.kind == Option::Transform && o.xf.kind != Xform::LookAt) world = affine_mul(world, parse_transform(o.xf));
            else if (o.kind == Option::Filter) out.filter = o.impl;
        }
        pbrs_camera_spec& cam = out.spec.camera;
        cam.width = (uint32_t)w;
        cam.height = (uint32_t)h;
        cam.fov_y_rad = pn_to_radians(fov_deg);
        (void)have_pose;  // without a LookAt the camera keeps Camera::new's pose: at the origin, looking down +z, y up
        std::memcpy(cam.from, pose, 3 * sizeof(float));
        std::memcpy(cam.target, pose + 3, 3 * sizeof(float));
        std::memcpy(cam.up, pose + 6, 3 * sizeof(float));
        for (const Item& it : items) world_item(it);
        for (pbrs_instance_spec& in : out.instances) {  // `instance.transform = world_transform * instance.transform` (:159-161)
            Affine t;
            std::memcpy(t.fwd.m, in.forward, sizeof in.forward);
            std::memcpy(t.inv.m, in.inverse, sizeof in.inverse);
            t = affine_mul(world, t);
            std::memcpy(in.forward, t.fwd.m, sizeof in.forward);
            std::memcpy(in.inverse, t.inv.m, sizeof in.inverse);
        }
    }
};

thread_local std::string g_load_error;

}  // namespace

extern "C" {

const char* pbrs_host_load_error(void) { return g_load_error.c_str(); }

int pbrs_host_load_pbrt(const char* path, pbrs_loaded_scene** out) {
    if (!path || !out) return PBRS_E_INVALID;
    *out = nullptr;
    try {
        std::vector<Tok> toks;
        lex_file(path, toks, 0);
        toks.push_back(Tok{});
        Parser parser(toks);
        std::vector<Option> options;
        std::vector<Item> items;
        parser.scene(options, items);
        auto ls = std::make_unique<pbrs_loaded_scene>();
        Loader loader(*ls, dir_of(path));
        loader.run(options, items);
        pbrs_scene_spec& s = ls->spec;
        s.n_meshes = (uint32_t)ls->meshes.size(); s.meshes = ls->meshes.data();
        s.n_shapes = (uint32_t)ls->shapes.size(); s.shapes = ls->shapes.data();
        s.n_materials = (uint32_t)ls->materials.size(); s.materials = ls->materials.data();
        s.n_instances = (uint32_t)ls->instances.size(); s.instances = ls->instances.data();
        s.n_area_lights = (uint32_t)ls->area_lights.size(); s.area_lights = ls->area_lights.data();
        s.n_delta_lights = (uint32_t)ls->delta_lights.size(); s.delta_lights = ls->delta_lights.data();
        s.n_textures = (uint32_t)ls->textures.size(); s.textures = ls->textures.data();
        s.n_fourier_tables = (uint32_t)ls->fourier_tables.size(); s.fourier_tables = ls->fourier_tables.data();
        if (s.n_instances == 0) fail("the scene has no shapes");
        *out = ls.release();
        return PBRS_OK;
    } catch (const std::exception& e) {
        g_load_error = e.what();
        return PBRS_E_INVALID;
    }
}
const pbrs_scene_spec* pbrs_loaded_scene_spec(const pbrs_loaded_scene* s) { return &s->spec; }
void pbrs_loaded_scene_free(pbrs_loaded_scene* s) { delete s; }

}  // extern "C"
