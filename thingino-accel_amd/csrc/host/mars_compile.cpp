/*
 * mars_compile.cpp -- the ONNX -> .mars compile step, host-only C++.
 *
 * The reference's compiler is a Rust program (mars-compiler/src/{main,onnx_parser,mars_format}.rs).  This file
 * restates its observable behaviour -- which bytes it writes for which ONNX graph -- behind the C-ABI of
 * include/mars_compile.h.  It is written from the behaviour, not from the Rust: one pass over the protobuf wire
 * format into flat tables, one pass over the nodes that appends tensor / layer records and weight bytes, a fixed-point
 * pass over the scales, one serialisation.  Reference lines are cited per function.  PARITY UNPINNED (see the header).
 *
 * Behaviours kept on purpose because a .mars consumer can observe them (each cited where it happens):
 *   - the compiler's layer numbering differs from the runtime header for three ops (mars_format.rs:50-71 writes
 *     Transpose = 15, FullyConnected = 16, Softmax = 17; include/mars.h reads SOFTMAX = 15, TRANSPOSE = 17);
 *   - pooling and batch-norm read H / W / C at the NCHW positions even under --nhwc (main.rs:942-944, :1025);
 *   - GlobalAveragePool becomes an AVGPOOL with the 2x2 / stride-2 defaults (main.rs:80, :927-928);
 *   - Resize takes its factors from input 2 only, Upsample-9 graphs therefore get 2x2 (main.rs:1273-1285);
 *   - attributes whose AttributeProto.type is missing are dropped (onnx_parser.rs:317-329);
 *   - bias bytes are copied verbatim as float32 whatever the quantisation mode (main.rs:785-800).
 * One deliberate difference: where the Rust iterates a HashMap (QDQ `_scale` initialisers, main.rs:156) the order is
 * unspecified there; here it is by name, so two initialisers that trim to the same base resolve the same way every run.
 */
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "mars.h"
#include "mars_compile.h"

namespace {

thread_local std::string g_err;

struct fail {
    std::string what;
};
[[noreturn]] void bail(const std::string &s) { throw fail{s}; }

/* ------------------------------------------------------------------ protobuf wire format ------------------------- */

struct field_t {
    uint32_t no, wt;
    uint64_t v;           /* varint / fixed value */
    const uint8_t *p;     /* length-delimited payload */
    size_t n;
};

struct reader_t {
    const uint8_t *p, *e;
    reader_t(const uint8_t *b, size_t n) : p(b), e(b + n) {}
    uint64_t varint()
    {
        uint64_t v = 0;
        for (int sh = 0; sh < 70; sh += 7) {
            if (p >= e) bail("ONNX protobuf: truncated varint");
            uint8_t b = *p++;
            if (sh < 64) v |= (uint64_t)(b & 0x7f) << sh;
            if (!(b & 0x80)) return v;
        }
        bail("ONNX protobuf: varint longer than 10 bytes");
    }
    bool next(field_t &f)
    {
        if (p >= e) return false;
        uint64_t key = varint();
        f.no = (uint32_t)(key >> 3);
        f.wt = (uint32_t)(key & 7);
        f.v = 0, f.p = nullptr, f.n = 0;
        if (f.no == 0) bail("ONNX protobuf: field number 0");
        switch (f.wt) {
        case 0: f.v = varint(); break;
        case 1:
            if (e - p < 8) bail("ONNX protobuf: truncated fixed64");
            memcpy(&f.v, p, 8), p += 8;
            break;
        case 2: {
            uint64_t n = varint();
            if (n > (uint64_t)(e - p)) bail("ONNX protobuf: length-delimited field runs past its message");
            f.p = p, f.n = (size_t)n, p += n;
            break;
        }
        case 5: {
            uint32_t w;
            if (e - p < 4) bail("ONNX protobuf: truncated fixed32");
            memcpy(&w, p, 4), p += 4, f.v = w;
            break;
        }
        default: bail("ONNX protobuf: unsupported wire type " + std::to_string(f.wt));
        }
        return true;
    }
};

void want(const field_t &f, uint32_t wt, const char *what)
{
    if (f.wt != wt) bail(std::string("ONNX protobuf: wrong wire type for ") + what);
}
/* a protobuf `string` must be well-formed UTF-8 (the reference's decoder refuses the file otherwise); `bytes` need not be */
bool utf8_ok(const uint8_t *p, size_t n)
{
    for (size_t i = 0; i < n;) {
        uint8_t c = p[i];
        size_t len = c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : (c >> 3) == 30 ? 4 : 0;
        if (!len || i + len > n) return false;
        uint32_t cp = len == 1 ? c : c & (0xff >> (len + 1));
        for (size_t k = 1; k < len; k++) {
            if ((p[i + k] & 0xc0) != 0x80) return false;
            cp = cp << 6 | (p[i + k] & 0x3f);
        }
        static const uint32_t lo[5] = {0, 0, 0x80, 0x800, 0x10000};
        if (cp < lo[len] || cp > 0x10ffff || (cp >= 0xd800 && cp <= 0xdfff)) return false;
        i += len;
    }
    return true;
}
std::string bytes_of(const field_t &f, const char *what)
{
    want(f, 2, what);
    return std::string((const char *)f.p, f.n);
}
std::string str_of(const field_t &f, const char *what)
{
    want(f, 2, what);
    if (!utf8_ok(f.p, f.n)) bail(std::string("ONNX protobuf: string is not UTF-8 in ") + what);
    return std::string((const char *)f.p, f.n);
}
/* repeated varint scalar: one value, or a packed run */
void rep_varint(const field_t &f, std::vector<int64_t> &out, const char *what)
{
    if (f.wt == 0) {
        out.push_back((int64_t)f.v);
    } else if (f.wt == 2) {
        reader_t r(f.p, f.n);
        while (r.p < r.e) out.push_back((int64_t)r.varint());
    } else {
        bail(std::string("ONNX protobuf: wrong wire type for ") + what);
    }
}
void rep_float(const field_t &f, std::vector<float> &out, const char *what)
{
    if (f.wt == 5) {
        uint32_t w = (uint32_t)f.v;
        float x;
        memcpy(&x, &w, 4), out.push_back(x);
    } else if (f.wt == 2) {
        if (f.n % 4) bail(std::string("ONNX protobuf: packed float run not a multiple of 4 bytes in ") + what);
        for (size_t i = 0; i < f.n; i += 4) {
            float x;
            memcpy(&x, f.p + i, 4), out.push_back(x);
        }
    } else {
        bail(std::string("ONNX protobuf: wrong wire type for ") + what);
    }
}

/* ------------------------------------------------------------------ ONNX tables (onnx_parser.rs:235-496) --------- */

enum { DT_UNDEFINED = 0, DT_FLOAT = 1, DT_UINT8 = 2, DT_INT8 = 3, DT_INT32 = 6, DT_INT64 = 7, DT_FLOAT16 = 10, DT_DOUBLE = 11 };

/* onnx_parser.rs:37-48: every code the compiler does not name collapses to Undefined */
int dtype_of(int32_t raw)
{
    switch (raw) {
    case 1: case 2: case 3: case 6: case 7: case 10: case 11: return raw;
    default: return DT_UNDEFINED;
    }
}

struct otensor_t {
    std::string name;
    std::vector<int64_t> dims;
    int dtype = DT_UNDEFINED;
    std::vector<uint8_t> data;      /* raw_data, else float_data / int64_data / int32_data re-serialised little-endian */
    std::vector<float> float_data;
};

/* TensorProto (onnx_parser.rs:79-95) -> OnnxTensor::from_proto (:246-284) */
otensor_t parse_tensor(const uint8_t *b, size_t n)
{
    otensor_t t;
    std::vector<int64_t> i32s, i64s;
    const uint8_t *raw = nullptr;
    size_t raw_n = 0;
    int32_t dt = 0;
    reader_t r(b, n);
    field_t f;
    while (r.next(f)) {
        switch (f.no) {
        case 1: rep_varint(f, t.dims, "TensorProto.dims"); break;
        case 2: want(f, 0, "TensorProto.data_type"), dt = (int32_t)f.v; break;
        case 4: rep_float(f, t.float_data, "TensorProto.float_data"); break;
        case 5: rep_varint(f, i32s, "TensorProto.int32_data"); break;
        case 7: rep_varint(f, i64s, "TensorProto.int64_data"); break;
        case 8: t.name = str_of(f, "TensorProto.name"); break;
        case 9: want(f, 2, "TensorProto.raw_data"), raw = f.p, raw_n = f.n; break;
        default: break;
        }
    }
    t.dtype = dtype_of(dt);
    if (t.dtype == DT_UNDEFINED && dt != 0) fprintf(stderr, "Warning: Unknown ONNX data_type %d for tensor %s\n", dt, t.name.c_str());
    if (raw_n) {
        t.data.assign(raw, raw + raw_n);
    } else if (!t.float_data.empty()) {
        t.data.resize(t.float_data.size() * 4);
        memcpy(t.data.data(), t.float_data.data(), t.data.size());
    } else if (!i64s.empty()) {
        t.data.resize(i64s.size() * 8);
        memcpy(t.data.data(), i64s.data(), t.data.size());
    } else if (!i32s.empty()) {
        t.data.resize(i32s.size() * 4);
        for (size_t i = 0; i < i32s.size(); i++) {
            int32_t v = (int32_t)i32s[i];
            memcpy(t.data.data() + 4 * i, &v, 4);
        }
    }
    return t;
}

struct attr_t {
    enum { NONE, INT, FLOAT, STRING, INTS, FLOATS, TENSOR } kind = NONE;
    int64_t i = 0;
    float f = 0;
    std::string s;
    std::vector<int64_t> ints;
    std::vector<float> floats;
};

struct onode_t {
    std::string name, op;
    std::vector<std::string> in, out;
    std::map<std::string, attr_t> attrs;
    const int64_t *get_int(const char *k) const
    {
        auto it = attrs.find(k);
        return it != attrs.end() && it->second.kind == attr_t::INT ? &it->second.i : nullptr;
    }
    const std::vector<int64_t> *get_ints(const char *k) const
    {
        auto it = attrs.find(k);
        return it != attrs.end() && it->second.kind == attr_t::INTS ? &it->second.ints : nullptr;
    }
    const float *get_float(const char *k) const
    {
        auto it = attrs.find(k);
        return it != attrs.end() && it->second.kind == attr_t::FLOAT ? &it->second.f : nullptr;
    }
    const std::string *get_string(const char *k) const
    {
        auto it = attrs.find(k);
        return it != attrs.end() && it->second.kind == attr_t::STRING ? &it->second.s : nullptr;
    }
};

/* AttributeProto (onnx_parser.rs:98-118) -> the typed value OnnxNode::from_proto keeps (:316-334) */
void parse_attr(const uint8_t *b, size_t n, onode_t &node)
{
    std::string name, s;
    bool has_f = false, has_i = false, has_s = false, has_t = false;
    float fv = 0;
    int64_t iv = 0;
    int32_t type = 0;
    std::vector<int64_t> ints;
    std::vector<float> floats;
    reader_t r(b, n);
    field_t f;
    while (r.next(f)) {
        switch (f.no) {
        case 1: name = str_of(f, "AttributeProto.name"); break;
        case 2: {
            want(f, 5, "AttributeProto.f");
            uint32_t w = (uint32_t)f.v;
            memcpy(&fv, &w, 4), has_f = true;
            break;
        }
        case 3: want(f, 0, "AttributeProto.i"), iv = (int64_t)f.v, has_i = true; break;
        case 4: s = bytes_of(f, "AttributeProto.s"), has_s = true; break;
        case 5: want(f, 2, "AttributeProto.t"), parse_tensor(f.p, f.n), has_t = true; break; /* validated, value never read by the compiler */
        case 7: rep_float(f, floats, "AttributeProto.floats"); break;
        case 8: rep_varint(f, ints, "AttributeProto.ints"); break;
        case 9: want(f, 2, "AttributeProto.strings"); break;
        case 20: want(f, 0, "AttributeProto.type"), type = (int32_t)f.v; break;
        default: break;
        }
    }
    attr_t a;
    switch (type) {
    case 1: if (has_f) a.kind = attr_t::FLOAT, a.f = fv; break;
    case 2: if (has_i) a.kind = attr_t::INT, a.i = iv; break;
    case 3: if (has_s) a.kind = attr_t::STRING, a.s = s; break;
    case 4: if (has_t) a.kind = attr_t::TENSOR; break;
    case 6: a.kind = attr_t::FLOATS, a.floats = floats; break;
    case 7: a.kind = attr_t::INTS, a.ints = ints; break;
    default: break;
    }
    if (a.kind != attr_t::NONE) node.attrs[name] = a;
}

onode_t parse_node(const uint8_t *b, size_t n)
{
    onode_t nd;
    reader_t r(b, n);
    field_t f;
    while (r.next(f)) {
        switch (f.no) {
        case 1: nd.in.push_back(str_of(f, "NodeProto.input")); break;
        case 2: nd.out.push_back(str_of(f, "NodeProto.output")); break;
        case 3: nd.name = str_of(f, "NodeProto.name"); break;
        case 4: nd.op = str_of(f, "NodeProto.op_type"); break;
        case 5: want(f, 2, "NodeProto.attribute"), parse_attr(f.p, f.n, nd); break;
        case 6: case 7: str_of(f, "NodeProto.doc_string / domain"); break;
        default: break;
        }
    }
    return nd;
}

struct oshape_t {
    std::string name;
    std::vector<int64_t> dims;
    bool has_shape = false; /* TensorShape::from_value_info returned Some (onnx_parser.rs:383-396) */
};

/* ValueInfoProto -> type.tensor_type.shape.dim[].dim_value, -1 where a dimension carries no value */
oshape_t parse_value_info(const uint8_t *b, size_t n)
{
    oshape_t s;
    reader_t r(b, n);
    field_t f;
    while (r.next(f)) {
        if (f.no == 1) {
            s.name = str_of(f, "ValueInfoProto.name");
        } else if (f.no == 2) { /* TypeProto */
            want(f, 2, "ValueInfoProto.type");
            reader_t rt(f.p, f.n);
            field_t ft;
            while (rt.next(ft)) {
                if (ft.no != 1) continue; /* tensor_type */
                want(ft, 2, "TypeProto.tensor_type");
                reader_t rtt(ft.p, ft.n);
                field_t fs;
                while (rtt.next(fs)) {
                    if (fs.no == 1) want(fs, 0, "TypeProto.Tensor.elem_type");
                    if (fs.no != 2) continue; /* shape */
                    want(fs, 2, "TypeProto.Tensor.shape");
                    s.has_shape = true;
                    reader_t rs(fs.p, fs.n);
                    field_t fd;
                    while (rs.next(fd)) {
                        if (fd.no != 1) continue; /* dim */
                        want(fd, 2, "TensorShapeProto.dim");
                        int64_t v = -1;
                        reader_t rd(fd.p, fd.n);
                        field_t fv;
                        while (rd.next(fv)) {
                            if (fv.no == 1) want(fv, 0, "Dimension.dim_value"), v = (int64_t)fv.v;
                            if (fv.no == 2) str_of(fv, "Dimension.dim_param");
                        }
                        s.dims.push_back(v);
                    }
                }
            }
        } else if (f.no == 3) {
            str_of(f, "ValueInfoProto.doc_string");
        }
    }
    return s;
}

struct omodel_t {
    std::string name, producer;
    int64_t opset = 11;
    std::vector<oshape_t> inputs, outputs;
    std::vector<onode_t> nodes;
    std::map<std::string, otensor_t> inits;
    std::map<std::string, std::vector<int64_t>> shape_info;
};

/* ModelProto.graph -> OnnxModel::from_proto (onnx_parser.rs:424-496) */
omodel_t parse_model(const uint8_t *b, size_t n)
{
    omodel_t m;
    std::vector<oshape_t> g_in, g_out, g_vi;
    std::vector<std::pair<std::string, std::vector<int64_t>>> init_dims;
    bool has_graph = false, has_opset = false;
    reader_t r(b, n);
    field_t f;
    while (r.next(f)) {
        if (f.no == 3 || f.no == 4 || f.no == 6) str_of(f, "ModelProto string");
        if (f.no == 1 || f.no == 5) want(f, 0, "ModelProto integer");
        if (f.no == 2) m.producer = str_of(f, "ModelProto.producer_name");
        if (f.no == 8) {
            want(f, 2, "ModelProto.opset_import");
            int64_t v = 0;
            reader_t ro(f.p, f.n);
            field_t fo;
            while (ro.next(fo))
                if (fo.no == 2) want(fo, 0, "OperatorSetIdProto.version"), v = (int64_t)fo.v;
            if (!has_opset) m.opset = v, has_opset = true;
        }
        if (f.no != 7) continue;
        want(f, 2, "ModelProto.graph");
        has_graph = true;
        reader_t rg(f.p, f.n);
        field_t fg;
        while (rg.next(fg)) {
            switch (fg.no) {
            case 1: want(fg, 2, "GraphProto.node"), m.nodes.push_back(parse_node(fg.p, fg.n)); break;
            case 2: m.name = str_of(fg, "GraphProto.name"); break;
            case 5: {
                want(fg, 2, "GraphProto.initializer");
                otensor_t t = parse_tensor(fg.p, fg.n);
                init_dims.push_back({t.name, t.dims});
                m.inits[t.name] = std::move(t);
                break;
            }
            case 10: str_of(fg, "GraphProto.doc_string"); break;
            case 11: want(fg, 2, "GraphProto.input"), g_in.push_back(parse_value_info(fg.p, fg.n)); break;
            case 12: want(fg, 2, "GraphProto.output"), g_out.push_back(parse_value_info(fg.p, fg.n)); break;
            case 13: want(fg, 2, "GraphProto.value_info"), g_vi.push_back(parse_value_info(fg.p, fg.n)); break;
            default: break;
            }
        }
    }
    if (!has_graph) bail("ONNX model has no graph");
    for (auto &s : g_in)
        if (s.has_shape && !m.inits.count(s.name)) m.inputs.push_back(s);
    for (auto &s : g_out)
        if (s.has_shape) m.outputs.push_back(s);
    for (auto *v : {&g_in, &g_out, &g_vi})
        for (auto &s : *v)
            if (s.has_shape) m.shape_info[s.name] = s.dims;
    for (auto &d : init_dims) m.shape_info[d.first] = d.second;
    return m;
}

/* ------------------------------------------------------------------ helpers -------------------------------------- */

/* the compiler's own layer numbering (mars_format.rs:50-71) */
enum {
    LT_CONV2D = 0, LT_DEPTHWISE = 1, LT_MAXPOOL = 2, LT_AVGPOOL = 3, LT_RELU = 5, LT_LEAKY_RELU = 7, LT_SIGMOID = 9,
    LT_CONCAT = 10, LT_ADD = 11, LT_MUL = 12, LT_UPSAMPLE = 13, LT_RESHAPE = 14, LT_TRANSPOSE = 15, LT_SOFTMAX = 17,
    LT_BATCHNORM = 18, LT_SKIP = -1,
};

/* main.rs:76-103 */
int map_op(const std::string &op)
{
    static const std::map<std::string, int> table = {
        {"Conv", LT_CONV2D}, {"MaxPool", LT_MAXPOOL}, {"AveragePool", LT_AVGPOOL}, {"GlobalAveragePool", LT_AVGPOOL},
        {"Relu", LT_RELU}, {"LeakyRelu", LT_LEAKY_RELU}, {"Sigmoid", LT_SIGMOID}, {"Mul", LT_MUL}, {"Add", LT_ADD},
        {"Concat", LT_CONCAT}, {"Resize", LT_UPSAMPLE}, {"Upsample", LT_UPSAMPLE}, {"Reshape", LT_RESHAPE},
        {"Transpose", LT_TRANSPOSE}, {"Softmax", LT_SOFTMAX}, {"BatchNormalization", LT_BATCHNORM},
    };
    static const char *quiet[] = {"Constant", "Shape", "Gather", "Slice", "Split", "Sub", "Div", "Unsqueeze", "Pow",
                                  "QuantizeLinear", "DequantizeLinear"};
    auto it = table.find(op);
    if (it != table.end()) return it->second;
    for (const char *q : quiet)
        if (op == q) return LT_SKIP;
    fprintf(stderr, "Warning: Unknown op type: %s\n", op.c_str());
    return LT_SKIP;
}

/* main.rs:20-46 */
float half_to_f32(uint16_t bits)
{
    uint32_t sign = (bits >> 15) & 1, exp = (bits >> 10) & 0x1f, mant = bits & 0x3ff;
    if (exp == 0) {
        if (mant == 0) return sign ? -0.0f : 0.0f;
        float f = (float)mant / 1024.0f * 6.103515625e-05f; /* 2^-14 */
        return sign ? -f : f;
    }
    if (exp == 31) {
        if (mant == 0) return sign ? -INFINITY : INFINITY;
        return NAN;
    }
    uint32_t w = (sign << 31) | ((exp + 127 - 15) << 23) | (mant << 13);
    float f;
    memcpy(&f, &w, 4);
    return f;
}

bool ends_with(const std::string &s, const char *suf)
{
    size_t n = strlen(suf);
    return s.size() >= n && memcmp(s.data() + s.size() - n, suf, n) == 0;
}
/* str::trim_end_matches: the suffix is removed as many times as it repeats */
std::string trim_end(std::string s, const char *suf)
{
    while (ends_with(s, suf) && *suf) s.resize(s.size() - strlen(suf));
    return s;
}
std::vector<float> bytes_to_f32(const std::vector<uint8_t> &d)
{
    std::vector<float> v(d.size() / 4);
    if (!v.empty()) memcpy(v.data(), d.data(), v.size() * 4);
    return v;
}
/* `x as u32` on a float: saturating, NaN -> 0 */
uint32_t f32_as_u32(float x)
{
    if (!(x > 0)) return 0;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)x;
}
bool is_default(float s, float tol) { return fabsf(s - 1.0f) < tol; }
int64_t at_or(const std::vector<int64_t> &v, size_t i, int64_t d) { return i < v.size() ? v[i] : d; }

/* ------------------------------------------------------------------ the compiler (main.rs:106-1523) -------------- */

struct compiler_t {
    const omodel_t &onnx;
    bool quantize, nhwc, verbose;
    std::vector<mars_tensor_t> tensors;
    std::vector<mars_layer_t> layers;
    std::vector<uint8_t> weights;
    std::map<std::string, uint32_t> tmap;
    std::map<std::string, float> qdq;
    bool has_qdq = false;

    compiler_t(const omodel_t &m, bool q, bool n, bool v) : onnx(m), quantize(q), nhwc(n), verbose(v) {}

    /* MarsTensor::new (mars_format.rs:171-184): int8, NHWC, 4 dims of 0, scale 1 */
    mars_tensor_t new_tensor(uint32_t id, const std::string &name)
    {
        mars_tensor_t t;
        memset(&t, 0, sizeof t);
        t.id = id;
        memcpy(t.name, name.data(), name.size() < sizeof t.name - 1 ? name.size() : sizeof t.name - 1);
        t.dtype = MARS_DTYPE_INT8, t.format = MARS_FORMAT_NHWC, t.ndims = 4, t.scale = 1.0f;
        return t;
    }
    mars_layer_t new_layer(int type)
    {
        mars_layer_t l;
        memset(&l, 0, sizeof l);
        l.id = (uint32_t)layers.size(), l.type = (mars_layer_type_t)type, l.num_inputs = 1, l.num_outputs = 1;
        for (int i = 0; i < 4; i++) l.input_tensor_ids[i] = l.output_tensor_ids[i] = 0xFFFFFFFFu;
        return l;
    }
    static void put_params(mars_layer_t &l, std::initializer_list<uint32_t> words)
    {
        size_t i = 0;
        for (uint32_t w : words) memcpy(l.params.raw + 4 * i++, &w, 4);
    }

    /* main.rs:137-214 */
    void parse_qdq_scales()
    {
        size_t n = 0;
        for (auto &nd : onnx.nodes) n += nd.op == "QuantizeLinear" || nd.op == "DequantizeLinear";
        if (!n) return;
        has_qdq = true;
        for (auto &kv : onnx.inits) {
            if (!ends_with(kv.first, "_scale")) continue;
            const otensor_t &t = kv.second;
            float s;
            if (t.data.size() >= 4) {
                memcpy(&s, t.data.data(), 4);
            } else if (t.data.size() >= 2) {
                uint16_t h;
                memcpy(&h, t.data.data(), 2), s = half_to_f32(h);
            } else if (t.data.empty() && !t.float_data.empty()) {
                s = t.float_data[0];
            } else {
                continue;
            }
            qdq[trim_end(kv.first, "_scale")] = s;
        }
        for (auto &nd : onnx.nodes) {
            if (nd.op != "QuantizeLinear" || nd.in.size() < 2) continue;
            auto it = qdq.find(trim_end(nd.in[1], "_scale"));
            if (it != qdq.end() && !qdq.count(nd.in[0])) {
                float s = it->second;
                qdq[nd.in[0]] = s;
            }
        }
        if (verbose) fprintf(stderr, "Loaded %zu QDQ scales (including shared)\n", qdq.size());
    }

    /* main.rs:217-260 */
    bool qdq_scale(const std::string &name, float &out) const
    {
        auto it = qdq.find(name);
        if (it != qdq.end()) return out = it->second, true;
        for (const char *suf : {"_DequantizeLinear_Output", "_QuantizeLinear_Output", "_QuantizeLinear_Input", "_quantized"}) {
            if (!ends_with(name, suf)) continue;
            it = qdq.find(trim_end(name, suf));
            if (it != qdq.end()) return out = it->second, true;
        }
        return false;
    }

    void set_feature_shape(mars_tensor_t &t, const std::vector<int64_t> &dims)
    {
        t.ndims = (uint32_t)dims.size();
        auto d = [&](size_t i) { return (int32_t)(dims[i] > 1 ? dims[i] : 1); };
        if (nhwc && dims.size() == 4) {
            t.shape[0] = d(0), t.shape[1] = d(2), t.shape[2] = d(3), t.shape[3] = d(1);
        } else {
            for (size_t i = 0; i < dims.size() && i < MARS_MAX_DIMS; i++) t.shape[i] = d(i);
        }
    }

    /* main.rs:407-458 */
    void create_inputs()
    {
        for (auto &in : onnx.inputs) {
            uint32_t id = (uint32_t)tensors.size();
            mars_tensor_t t = new_tensor(id, in.name);
            set_feature_shape(t, in.dims);
            t.format = nhwc && in.dims.size() == 4 ? MARS_FORMAT_NHWC : MARS_FORMAT_NCHW;
            t.dtype = quantize ? MARS_DTYPE_INT8 : MARS_DTYPE_FLOAT32;
            float qs;
            if (quantize) t.scale = qdq_scale(in.name, qs) ? qs : 1.0f / 255.0f;
            tmap[in.name] = id;
            tensors.push_back(t);
        }
    }

    /* main.rs:499-552 */
    uint32_t feature(const std::string &name)
    {
        auto it = tmap.find(name);
        if (it != tmap.end()) return it->second;
        uint32_t id = (uint32_t)tensors.size();
        mars_tensor_t t = new_tensor(id, name);
        t.dtype = quantize ? MARS_DTYPE_INT8 : MARS_DTYPE_FLOAT32;
        t.format = nhwc ? MARS_FORMAT_NHWC : MARS_FORMAT_NCHW;
        std::string key = name;
        if (!onnx.shape_info.count(name))
            for (const char *suf : {"_DequantizeLinear_Output", "_QuantizeLinear_Output", "_QuantizeLinear_Input"})
                if (ends_with(name, suf)) {
                    key = trim_end(name, suf);
                    break;
                }
        auto si = onnx.shape_info.find(key);
        if (si != onnx.shape_info.end()) set_feature_shape(t, si->second);
        float qs;
        if (quantize && qdq_scale(name, qs)) t.scale = qs;
        tmap[name] = id;
        tensors.push_back(t);
        return id;
    }

    float scale_of(uint32_t id) const { return id < tensors.size() ? tensors[id].scale : 1.0f; }
    void set_scale(uint32_t id, float s)
    {
        if (id < tensors.size()) tensors[id].scale = s;
    }
    /* main.rs:589-600: only a tensor that has no shape yet takes the computed one */
    void update_shape(uint32_t id, std::initializer_list<int32_t> shape)
    {
        if (id >= tensors.size()) return;
        mars_tensor_t &t = tensors[id];
        if (t.ndims != 0 && t.shape[0] != 0) return;
        t.ndims = (uint32_t)shape.size();
        size_t i = 0;
        for (int32_t d : shape) t.shape[i++] = d;
    }
    struct shape4 {
        int32_t d[4];
    };
    shape4 shape_of(uint32_t id) const
    {
        shape4 s = {{0, 0, 0, 0}};
        if (id < tensors.size())
            for (int i = 0; i < 4; i++) s.d[i] = tensors[id].shape[i];
        return s;
    }

    /* main.rs:611-619 */
    uint64_t add_weights(const std::vector<uint8_t> &d)
    {
        /* a hostile few-KB file can declare shapes that make every node allocate tens of MB: the blob of a model this
         * path can run is bounded by the reference runtime's addressable weights; refuse anything beyond 1 GiB in total */
        if (weights.size() + d.size() > ((size_t)1 << 30)) bail("weight blob exceeds 1 GiB");
        uint64_t off = weights.size();
        weights.insert(weights.end(), d.begin(), d.end());
        while (weights.size() % 4) weights.push_back(0);
        return off;
    }

    /* main.rs:621-677 */
    std::vector<uint8_t> quantize_weights(const otensor_t &w, float &scale) const
    {
        std::vector<float> f;
        if (w.dtype == DT_FLOAT16) {
            f.resize(w.data.size() / 2);
            for (size_t i = 0; i < f.size(); i++) {
                uint16_t h;
                memcpy(&h, w.data.data() + 2 * i, 2), f[i] = half_to_f32(h);
            }
        } else if (w.dtype == DT_INT8) {
            scale = 1.0f / 127.0f;
            return w.data;
        } else {
            if (w.dtype != DT_FLOAT) fprintf(stderr, "  Warning: Unknown dtype %d, trying as float32\n", w.dtype);
            f = bytes_to_f32(w.data);
        }
        float max_abs = 0.0f;
        for (float x : f) max_abs = fmaxf(max_abs, fabsf(x));
        scale = max_abs > 0.0f ? max_abs / 127.0f : 1.0f;
        std::vector<uint8_t> q(f.size());
        for (size_t i = 0; i < f.size(); i++) {
            float r = roundf(f[i] / scale); /* f32::round: halves away from zero */
            int8_t v = std::isnan(r) ? 0 : (int8_t)(r < -127.0f ? -127.0f : r > 127.0f ? 127.0f : r);
            q[i] = (uint8_t)v;
        }
        return q;
    }

    /* mars_format.rs:407-434 */
    static std::vector<uint8_t> oihw_to_ohwi(const std::vector<uint8_t> &w, size_t O, size_t I, size_t KH, size_t KW)
    {
        std::vector<uint8_t> out(O * I * KH * KW, 0);
        for (size_t o = 0; o < O; o++)
            for (size_t i = 0; i < I; i++)
                for (size_t h = 0; h < KH; h++)
                    for (size_t x = 0; x < KW; x++) {
                        size_t s = ((o * I + i) * KH + h) * KW + x, d = ((o * KH + h) * KW + x) * I + i;
                        if (s < w.size()) out[d] = w[s];
                    }
        return out;
    }

    static const std::string &arg(const std::vector<std::string> &v, size_t i, const char *what)
    {
        if (i >= v.size()) bail(what);
        return v[i];
    }

    /* main.rs:686-916 */
    void conv(const onode_t &nd)
    {
        uint32_t in_id = feature(arg(nd.in, 0, "Conv missing input"));
        const std::string &w_in = arg(nd.in, 1, "Conv missing weight");
        std::string w_name = w_in;
        const otensor_t *wt = nullptr;
        float qdq_w = 0;
        bool has_qdq_w = false;
        if (has_qdq) {
            std::string base = trim_end(w_in, "_DequantizeLinear_Output"), qn = base + "_quantized";
            auto it = onnx.inits.find(qn);
            if (it != onnx.inits.end()) {
                wt = &it->second, w_name = qn, has_qdq_w = qdq_scale(base, qdq_w);
            } else if ((it = onnx.inits.find(w_in)) != onnx.inits.end()) {
                wt = &it->second;
            } else {
                bail("Conv weight not found: " + w_in);
            }
        } else {
            auto it = onnx.inits.find(w_in);
            if (it == onnx.inits.end()) bail("Conv weight not found in initializers");
            wt = &it->second;
        }
        uint32_t oc = (uint32_t)at_or(wt->dims, 0, 1), ic = (uint32_t)at_or(wt->dims, 1, 1);
        uint32_t kh = (uint32_t)at_or(wt->dims, 2, 3), kw = (uint32_t)at_or(wt->dims, 3, 3);
        if ((uint64_t)oc * ic * kh * kw > ((uint64_t)1 << 28)) bail("Conv weight dims out of range"); /* the --nhwc re-order allocates that many bytes */
        /* ... and, where that allocation happens (quantised weights under --nhwc), only for weights that are really there: the element
         * count the re-order walks must be covered by the initializer's payload, or a few-KB file could ask for 256 MB per node
         * (ADVICE r3).  Elsewhere the payload is copied as it is, as the reference does -- weights with fewer than four dims (kh / kw
         * then default to 3) and initializers without an inline payload (external data) compile like there */
        if (quantize && nhwc && (uint64_t)oc * ic * kh * kw > (uint64_t)wt->data.size())
            bail("Conv weight initializer shorter than its dims");

        std::vector<uint8_t> wdata;
        float w_scale = 1.0f;
        mars_format_t w_fmt = MARS_FORMAT_OIHW;
        if (quantize) {
            if (wt->dtype == DT_INT8) {
                w_scale = has_qdq_w ? qdq_w : 1.0f / 127.0f;
                wdata = wt->data;
            } else {
                wdata = quantize_weights(*wt, w_scale);
            }
            if (nhwc) wdata = oihw_to_ohwi(wdata, oc, ic, kh, kw), w_fmt = MARS_FORMAT_OHWI;
        } else {
            wdata = wt->data;
        }
        uint64_t w_off = add_weights(wdata);
        uint32_t w_id = (uint32_t)tensors.size();
        mars_tensor_t w = new_tensor(w_id, w_name);
        w.dtype = quantize ? MARS_DTYPE_INT8 : MARS_DTYPE_FLOAT32, w.format = w_fmt;
        w.shape[0] = (int32_t)oc, w.shape[1] = (int32_t)ic, w.shape[2] = (int32_t)kh, w.shape[3] = (int32_t)kw;
        w.scale = w_scale, w.data_offset = w_off, w.data_size = wdata.size();
        tmap[w_name] = w_id;
        tensors.push_back(w);

        uint32_t b_id = 0xFFFFFFFFu;
        if (nd.in.size() > 2) {
            auto it = onnx.inits.find(nd.in[2]);
            if (it != onnx.inits.end()) {
                uint64_t b_off = add_weights(it->second.data);
                b_id = (uint32_t)tensors.size();
                mars_tensor_t b = new_tensor(b_id, nd.in[2]);
                b.dtype = MARS_DTYPE_FLOAT32, b.ndims = 1, b.shape[0] = (int32_t)oc;
                b.data_offset = b_off, b.data_size = it->second.data.size();
                tensors.push_back(b);
            }
        }
        const std::string &out_name = arg(nd.out, 0, "Conv missing output");
        uint32_t out_id = feature(out_name);

        static const std::vector<int64_t> one2 = {1, 1}, zero4 = {0, 0, 0, 0};
        const std::vector<int64_t> &st = nd.get_ints("strides") ? *nd.get_ints("strides") : one2;
        const std::vector<int64_t> &pd = nd.get_ints("pads") ? *nd.get_ints("pads") : zero4;
        const std::vector<int64_t> &dl = nd.get_ints("dilations") ? *nd.get_ints("dilations") : one2;
        uint32_t group = (uint32_t)(nd.get_int("group") ? *nd.get_int("group") : 1);
        int32_t sh = (int32_t)at_or(st, 0, 1), sw = (int32_t)at_or(st, 1, 1), dh = (int32_t)at_or(dl, 0, 1), dw = (int32_t)at_or(dl, 1, 1);
        int32_t pt = (int32_t)at_or(pd, 0, 0), pl = (int32_t)at_or(pd, 1, 0), pb = (int32_t)at_or(pd, 2, 0), pr = (int32_t)at_or(pd, 3, 0);
        if (sh == 0 || sw == 0) bail("Conv stride of 0"); /* the reference divides by it and aborts */
        shape4 is = shape_of(in_id);
        int64_t in_h = nhwc ? is.d[1] : is.d[2], in_w = nhwc ? is.d[2] : is.d[3];
        int32_t oh = (int32_t)((in_h + pt + pb - (int64_t)dh * ((int32_t)kh - 1) - 1) / sh + 1);
        int32_t ow = (int32_t)((in_w + pl + pr - (int64_t)dw * ((int32_t)kw - 1) - 1) / sw + 1);
        if (nhwc)
            update_shape(out_id, {is.d[0], oh, ow, (int32_t)oc});
        else
            update_shape(out_id, {is.d[0], (int32_t)oc, oh, ow});

        if (quantize) {
            float os;
            if (!qdq_scale(out_name, os)) {
                float fan_in = (float)(uint32_t)(ic * kh * kw);
                os = scale_of(in_id) * w_scale * fan_in; /* main.rs:865-868 */
            }
            set_scale(out_id, os);
        }
        bool all_zero = true;
        for (int64_t p : pd) all_zero &= p == 0;
        mars_layer_t l = new_layer(group > 1 && group == ic && group == oc ? LT_DEPTHWISE : LT_CONV2D);
        l.input_tensor_ids[0] = in_id, l.output_tensor_ids[0] = out_id;
        put_params(l, {kh, kw, (uint32_t)at_or(st, 0, 1), (uint32_t)at_or(st, 1, 1), (uint32_t)at_or(dl, 0, 1), (uint32_t)at_or(dl, 1, 1),
                       (uint32_t)(all_zero ? MARS_PAD_VALID : MARS_PAD_EXPLICIT), (uint32_t)at_or(pd, 0, 0), (uint32_t)at_or(pd, 2, 0),
                       (uint32_t)at_or(pd, 1, 0), (uint32_t)at_or(pd, 3, 0), group, (uint32_t)MARS_ACT_NONE, w_id, b_id});
        layers.push_back(l);
        if (verbose) fprintf(stderr, "Conv %u: in_ch=%u out_ch=%u k=%ux%u\n", l.id, ic, oc, kh, kw);
    }

    /* main.rs:918-972 */
    void pool(const onode_t &nd, int type)
    {
        uint32_t in_id = feature(arg(nd.in, 0, "Pool missing input")), out_id = feature(arg(nd.out, 0, "Pool missing output"));
        static const std::vector<int64_t> two2 = {2, 2}, zero4 = {0, 0, 0, 0};
        const std::vector<int64_t> &k = nd.get_ints("kernel_shape") ? *nd.get_ints("kernel_shape") : two2;
        const std::vector<int64_t> &st = nd.get_ints("strides") ? *nd.get_ints("strides") : two2;
        const std::vector<int64_t> &pd = nd.get_ints("pads") ? *nd.get_ints("pads") : zero4;
        int32_t kh = (int32_t)at_or(k, 0, 2), kw = (int32_t)at_or(k, 1, 2), sh = (int32_t)at_or(st, 0, 2), sw = (int32_t)at_or(st, 1, 2);
        int32_t pt = (int32_t)at_or(pd, 0, 0), pl = (int32_t)at_or(pd, 1, 0), pb = (int32_t)at_or(pd, 2, 0), pr = (int32_t)at_or(pd, 3, 0);
        if (sh == 0 || sw == 0) bail("Pool stride of 0");
        shape4 is = shape_of(in_id);
        int32_t oh = (int32_t)(((int64_t)is.d[2] + pt + pb - kh) / sh + 1), ow = (int32_t)(((int64_t)is.d[3] + pl + pr - kw) / sw + 1);
        update_shape(out_id, {is.d[0], is.d[1], oh, ow});
        if (quantize) set_scale(out_id, scale_of(in_id));
        bool all_zero = true;
        for (int64_t p : pd) all_zero &= p == 0;
        mars_layer_t l = new_layer(type);
        l.input_tensor_ids[0] = in_id, l.output_tensor_ids[0] = out_id;
        put_params(l, {(uint32_t)kh, (uint32_t)kw, (uint32_t)sh, (uint32_t)sw, (uint32_t)(all_zero ? MARS_PAD_VALID : MARS_PAD_EXPLICIT),
                       (uint32_t)pt, (uint32_t)pb, (uint32_t)pl, (uint32_t)pr});
        layers.push_back(l);
    }

    /* main.rs:974-1008 */
    void activation(const onode_t &nd, int type)
    {
        uint32_t in_id = feature(arg(nd.in, 0, "Activation missing input")), out_id = feature(arg(nd.out, 0, "Activation missing output"));
        shape4 s = shape_of(in_id);
        update_shape(out_id, {s.d[0], s.d[1], s.d[2], s.d[3]});
        if (quantize) set_scale(out_id, type == LT_SIGMOID ? 1.0f / 127.0f : scale_of(in_id));
        mars_layer_t l = new_layer(type);
        l.input_tensor_ids[0] = in_id, l.output_tensor_ids[0] = out_id;
        layers.push_back(l);
    }

    /* main.rs:1011-1140 */
    void batchnorm(const onode_t &nd)
    {
        uint32_t in_id = feature(arg(nd.in, 0, "BatchNorm missing input")), out_id = feature(arg(nd.out, 0, "BatchNorm missing output"));
        shape4 s = shape_of(in_id);
        update_shape(out_id, {s.d[0], s.d[1], s.d[2], s.d[3]});
        if (s.d[1] < 0 || s.d[1] > (1 << 20)) bail("BatchNorm channel count out of range"); /* six vectors of that many floats are allocated */
        size_t nc = (size_t)s.d[1];
        float eps = nd.get_float("epsilon") ? *nd.get_float("epsilon") : 1e-5f;
        auto operand = [&](size_t i, float dflt) {
            if (i < nd.in.size()) {
                auto it = onnx.inits.find(nd.in[i]);
                if (it != onnx.inits.end()) return bytes_to_f32(it->second.data);
            }
            return std::vector<float>(nc, dflt);
        };
        std::vector<float> gamma = operand(1, 1.0f), beta = operand(2, 0.0f), mean = operand(3, 0.0f), var = operand(4, 1.0f);
        std::vector<float> fs(nc, 1.0f), fb(nc, 0.0f);
        for (size_t i = 0; i < nc && i < gamma.size() && i < var.size(); i++) {
            float inv_std = 1.0f / sqrtf(var[i] + eps);
            fs[i] = gamma[i] * inv_std;
            fb[i] = (i < beta.size() ? beta[i] : 0.0f) - (i < mean.size() ? mean[i] : 0.0f) * fs[i];
        }
        uint32_t ids[2];
        const std::vector<float> *vals[2] = {&fs, &fb};
        const char *suffix[2] = {"_scale", "_bias"};
        for (int k = 0; k < 2; k++) {
            std::vector<uint8_t> bytes(nc * 4);
            if (nc) memcpy(bytes.data(), vals[k]->data(), nc * 4);
            uint64_t off = add_weights(bytes);
            ids[k] = (uint32_t)tensors.size();
            mars_tensor_t t = new_tensor(ids[k], nd.name + suffix[k]);
            t.dtype = MARS_DTYPE_FLOAT32, t.ndims = 1, t.shape[0] = (int32_t)nc, t.data_offset = off, t.data_size = bytes.size();
            tensors.push_back(t);
        }
        if (quantize) {
            float mx = 0.0f;
            for (float x : fs) mx = fmaxf(mx, fabsf(x));
            set_scale(out_id, scale_of(in_id) * fmaxf(mx, 0.1f));
        }
        mars_layer_t l = new_layer(LT_BATCHNORM);
        l.num_inputs = 3;
        l.input_tensor_ids[0] = in_id, l.input_tensor_ids[1] = ids[0], l.input_tensor_ids[2] = ids[1], l.output_tensor_ids[0] = out_id;
        layers.push_back(l);
    }

    /* main.rs:1142-1186 */
    void elementwise(const onode_t &nd, int type)
    {
        uint32_t a = feature(arg(nd.in, 0, "Elementwise missing input A")), b = feature(arg(nd.in, 1, "Elementwise missing input B"));
        uint32_t out_id = feature(arg(nd.out, 0, "Elementwise missing output"));
        shape4 s = shape_of(a);
        update_shape(out_id, {s.d[0], s.d[1], s.d[2], s.d[3]});
        if (quantize) {
            float sa = scale_of(a), sb = scale_of(b), os;
            if (type == LT_ADD)
                os = fmaxf(sa, sb);
            else
                os = is_default(sa, 0.001f) ? sb : is_default(sb, 0.001f) ? sa : fminf(sa, sb);
            set_scale(out_id, os);
        }
        mars_layer_t l = new_layer(type);
        l.num_inputs = 2;
        l.input_tensor_ids[0] = a, l.input_tensor_ids[1] = b, l.output_tensor_ids[0] = out_id;
        layers.push_back(l);
    }

    /* main.rs:1188-1261 */
    void concat(const onode_t &nd)
    {
        int64_t raw = nd.get_int("axis") ? *nd.get_int("axis") : 1;
        uint32_t axis = raw < 0 ? (uint32_t)(4 + raw) : (uint32_t)raw;
        if (axis > 3) axis = 3;
        if (nhwc && axis > 0) axis = axis == 1 ? 3 : axis == 2 ? 1 : 2;
        mars_layer_t l = new_layer(LT_CONCAT);
        l.num_inputs = (uint32_t)(nd.in.size() < 4 ? nd.in.size() : 4);
        int32_t total = 0;
        shape4 base = {{0, 0, 0, 0}};
        for (uint32_t i = 0; i < l.num_inputs; i++) {
            uint32_t tid = feature(nd.in[i]);
            l.input_tensor_ids[i] = tid;
            shape4 s = shape_of(tid);
            if (i == 0) base = s;
            total = (int32_t)((uint32_t)total + (uint32_t)s.d[axis]);
        }
        uint32_t out_id = feature(arg(nd.out, 0, "Concat missing output"));
        l.output_tensor_ids[0] = out_id;
        base.d[axis] = total;
        update_shape(out_id, {base.d[0], base.d[1], base.d[2], base.d[3]});
        if (quantize && is_default(scale_of(out_id), 0.0001f)) {
            float mx = 0.0f;
            for (uint32_t i = 0; i < l.num_inputs; i++) mx = fmaxf(mx, scale_of(l.input_tensor_ids[i]));
            if (mx > 0.0001f) set_scale(out_id, mx);
        }
        put_params(l, {axis, l.num_inputs});
        layers.push_back(l);
    }

    /* main.rs:1263-1324 */
    void upsample(const onode_t &nd)
    {
        uint32_t in_id = feature(arg(nd.in, 0, "Upsample missing input")), out_id = feature(arg(nd.out, 0, "Upsample missing output"));
        uint32_t fh = 2, fw = 2;
        if (nd.in.size() > 2) {
            auto it = onnx.inits.find(nd.in[2]);
            if (it != onnx.inits.end()) {
                std::vector<float> v = bytes_to_f32(it->second.data);
                fh = f32_as_u32(v.size() > 2 ? v[2] : 2.0f), fw = f32_as_u32(v.size() > 3 ? v[3] : 2.0f);
            }
        }
        shape4 s = shape_of(in_id);
        auto mul = [](int32_t a, uint32_t b) { return (int32_t)((uint32_t)a * b); };
        if (nhwc)
            update_shape(out_id, {s.d[0], mul(s.d[1], fh), mul(s.d[2], fw), s.d[3]});
        else
            update_shape(out_id, {s.d[0], s.d[1], mul(s.d[2], fh), mul(s.d[3], fw)});
        const std::string *mode = nd.get_string("mode");
        uint32_t mode_val = mode && (*mode == "bilinear" || *mode == "linear") ? 1 : 0;
        if (quantize) set_scale(out_id, scale_of(in_id));
        mars_layer_t l = new_layer(LT_UPSAMPLE);
        l.input_tensor_ids[0] = in_id, l.output_tensor_ids[0] = out_id;
        put_params(l, {fh, fw, mode_val});
        layers.push_back(l);
    }

    /* main.rs:1326-1378 */
    void reshape(const onode_t &nd)
    {
        uint32_t in_id = feature(arg(nd.in, 0, "Reshape missing input")), out_id = feature(arg(nd.out, 0, "Reshape missing output"));
        int32_t target[6] = {0, 0, 0, 0, 0, 0};
        uint32_t nd_ = 4;
        if (nd.in.size() > 1) {
            auto it = onnx.inits.find(nd.in[1]);
            if (it != onnx.inits.end()) {
                size_t n = it->second.data.size() / 8;
                nd_ = (uint32_t)(n < 6 ? n : 6);
                for (size_t i = 0; i < nd_; i++) {
                    int64_t d;
                    memcpy(&d, it->second.data.data() + 8 * i, 8), target[i] = (int32_t)d;
                }
            }
        }
        int32_t o[4] = {1, 1, 1, 1};
        for (uint32_t i = 0; i < 4 && i < nd_; i++) o[i] = target[i];
        update_shape(out_id, {o[0], o[1], o[2], o[3]});
        if (quantize) set_scale(out_id, scale_of(in_id));
        mars_layer_t l = new_layer(LT_RESHAPE);
        l.input_tensor_ids[0] = in_id, l.output_tensor_ids[0] = out_id;
        put_params(l, {(uint32_t)target[0], (uint32_t)target[1], (uint32_t)target[2], (uint32_t)target[3], (uint32_t)target[4],
                       (uint32_t)target[5], nd_});
        layers.push_back(l);
    }

    /* The reference's compiler and its runtime disagree on three layer-type numbers (quirk list at the top of this file);
     * the bytes are kept, the user is told: a file with such a layer runs a DIFFERENT op in the runtime */
    void numbering_warning(const char *op, int written, const char *runtime_reads)
    {
        fprintf(stderr, "mars_compile: warning: %s is written as layer type %d (the reference compiler's numbering); the runtime header "
                        "include/mars.h reads %d as %s -- the compiled file will not run this op as intended\n", op, written, written, runtime_reads);
    }

    /* main.rs:1380-1427 */
    void transpose(const onode_t &nd)
    {
        uint32_t in_id = feature(arg(nd.in, 0, "Transpose missing input")), out_id = feature(arg(nd.out, 0, "Transpose missing output"));
        static const std::vector<int64_t> ident = {0, 1, 2, 3};
        const std::vector<int64_t> &perm = nd.get_ints("perm") ? *nd.get_ints("perm") : ident;
        uint32_t pa[6] = {0, 0, 0, 0, 0, 0};
        for (size_t i = 0; i < perm.size() && i < 6; i++) pa[i] = (uint32_t)perm[i];
        shape4 s = shape_of(in_id);
        int32_t o[4] = {1, 1, 1, 1};
        for (size_t i = 0; i < 4 && i < perm.size(); i++)
            if ((uint64_t)perm[i] < 4) o[i] = s.d[perm[i]];
        update_shape(out_id, {o[0], o[1], o[2], o[3]});
        if (quantize) set_scale(out_id, scale_of(in_id));
        numbering_warning("Transpose", 15, "SOFTMAX");
        mars_layer_t l = new_layer(LT_TRANSPOSE);
        l.input_tensor_ids[0] = in_id, l.output_tensor_ids[0] = out_id;
        put_params(l, {pa[0], pa[1], pa[2], pa[3], pa[4], pa[5], (uint32_t)perm.size()});
        layers.push_back(l);
    }

    /* main.rs:1429-1460 */
    void softmax(const onode_t &nd)
    {
        uint32_t in_id = feature(arg(nd.in, 0, "Softmax missing input")), out_id = feature(arg(nd.out, 0, "Softmax missing output"));
        shape4 s = shape_of(in_id);
        update_shape(out_id, {s.d[0], s.d[1], s.d[2], s.d[3]});
        if (quantize) set_scale(out_id, 1.0f / 127.0f);
        int64_t axis = nd.get_int("axis") ? *nd.get_int("axis") : -1;
        numbering_warning("Softmax", 17, "TRANSPOSE");
        mars_layer_t l = new_layer(LT_SOFTMAX);
        l.input_tensor_ids[0] = in_id, l.output_tensor_ids[0] = out_id;
        put_params(l, {axis < 0 ? (uint32_t)(4 + axis) : (uint32_t)axis});
        layers.push_back(l);
    }

    /* main.rs:312-405: outputs still at the default scale take one from their inputs, at most five sweeps */
    void propagate_scales()
    {
        for (int sweep = 0; sweep < 5; sweep++) {
            bool any = false;
            for (auto &l : layers) {
                uint32_t out = l.output_tensor_ids[0];
                auto s_in = [&](uint32_t i) { return i < l.num_inputs ? scale_of(l.input_tensor_ids[i]) : 1.0f; };
                auto live = [](float s) { return fabsf(s - 1.0f) > 0.0001f; };
                if (live(scale_of(out))) continue;
                float ns = 0;
                bool have = false;
                switch ((int)l.type) {
                case LT_RESHAPE: case LT_TRANSPOSE: case LT_SOFTMAX: case LT_MAXPOOL: case LT_AVGPOOL: case LT_UPSAMPLE:
                    if (l.num_inputs < 1) bail("layer without inputs");
                    if (live(s_in(0))) ns = s_in(0), have = true;
                    break;
                case LT_CONCAT: {
                    float mx = 0.0f;
                    for (uint32_t i = 0; i < l.num_inputs && i < 4; i++)
                        if (live(s_in(i))) mx = fmaxf(mx, s_in(i));
                    if (mx > 0.0001f) ns = mx, have = true;
                    break;
                }
                case LT_ADD: {
                    float mx = fmaxf(s_in(0), s_in(1));
                    if (live(mx)) ns = mx, have = true;
                    break;
                }
                case LT_MUL: {
                    float a = s_in(0), b = s_in(1);
                    if (live(a) && live(b)) ns = a * b, have = true;
                    else if (live(a)) ns = a, have = true;
                    else if (live(b)) ns = b, have = true;
                    break;
                }
                default: break;
                }
                if (have && out < tensors.size()) tensors[out].scale = ns, any = true;
            }
            if (!any) break;
        }
    }

    void run()
    {
        parse_qdq_scales();
        create_inputs();
        for (auto &nd : onnx.nodes) {
            int t = map_op(nd.op);
            switch (t) {
            case LT_CONV2D: conv(nd); break;
            case LT_MAXPOOL: case LT_AVGPOOL: pool(nd, t); break;
            case LT_RELU: case LT_SIGMOID: case LT_LEAKY_RELU: activation(nd, t); break;
            case LT_ADD: case LT_MUL: elementwise(nd, t); break;
            case LT_CONCAT: concat(nd); break;
            case LT_UPSAMPLE: upsample(nd); break;
            case LT_BATCHNORM: batchnorm(nd); break;
            case LT_RESHAPE: reshape(nd); break;
            case LT_TRANSPOSE: transpose(nd); break;
            case LT_SOFTMAX: softmax(nd); break;
            default: break;
            }
        }
        propagate_scales();
    }

    /* main.rs:1463-1522 */
    std::vector<uint8_t> serialise() const
    {
        mars_header_t h;
        memset(&h, 0, sizeof h);
        h.magic = MARS_MAGIC, h.version_major = MARS_VERSION_MAJOR, h.version_minor = MARS_VERSION_MINOR;
        h.num_layers = (uint32_t)layers.size(), h.num_tensors = (uint32_t)tensors.size();
        h.num_inputs = (uint32_t)onnx.inputs.size(), h.num_outputs = (uint32_t)onnx.outputs.size();
        h.weights_offset = sizeof h + tensors.size() * sizeof(mars_tensor_t) + layers.size() * sizeof(mars_layer_t);
        h.weights_size = weights.size();
        for (int i = 0; i < 4; i++) h.input_tensor_ids[i] = h.output_tensor_ids[i] = 0xFFFFFFFFu;
        for (size_t i = 0; i < onnx.inputs.size() && i < 4; i++) {
            auto it = tmap.find(onnx.inputs[i].name);
            if (it != tmap.end()) h.input_tensor_ids[i] = it->second;
        }
        for (size_t i = 0; i < onnx.outputs.size() && i < 4; i++) {
            auto it = tmap.find(onnx.outputs[i].name);
            if (it == tmap.end()) it = tmap.find(onnx.outputs[i].name + "_QuantizeLinear_Input");
            if (it != tmap.end())
                h.output_tensor_ids[i] = it->second;
            else
                fprintf(stderr, "Warning: Output tensor %s not found in tensor_map\n", onnx.outputs[i].name.c_str());
        }
        std::vector<uint8_t> out((size_t)h.weights_offset + weights.size());
        uint8_t *p = out.data();
        memcpy(p, &h, sizeof h), p += sizeof h;
        if (!tensors.empty()) memcpy(p, tensors.data(), tensors.size() * sizeof(mars_tensor_t));
        p += tensors.size() * sizeof(mars_tensor_t);
        if (!layers.empty()) memcpy(p, layers.data(), layers.size() * sizeof(mars_layer_t));
        p += layers.size() * sizeof(mars_layer_t);
        if (!weights.empty()) memcpy(p, weights.data(), weights.size());
        return out;
    }
};

std::vector<uint8_t> compile_bytes(const void *onnx, size_t n, const mars_compile_opts_t *o)
{
    mars_compile_opts_t d = {0, 0, 0};
    if (!o) o = &d;
    if (!onnx && n) bail("null ONNX buffer");
    omodel_t m = parse_model((const uint8_t *)onnx, n);
    if (o->verbose)
        fprintf(stderr, "ONNX Model: %s (producer %s, opset %lld): %zu nodes, %zu initializers\n", m.name.c_str(), m.producer.c_str(),
                (long long)m.opset, m.nodes.size(), m.inits.size());
    compiler_t c(m, !o->float32, o->nhwc != 0, o->verbose != 0);
    c.run();
    if (o->verbose)
        fprintf(stderr, "  Layers: %zu  Tensors: %zu  Weights: %zu bytes\n", c.layers.size(), c.tensors.size(), c.weights.size());
    return c.serialise();
}

} // namespace

extern "C" size_t mars_compile_onnx(const void *onnx, size_t onnx_size, const mars_compile_opts_t *opts, void *out, size_t cap)
{
    g_err.clear();
    try {
        std::vector<uint8_t> b = compile_bytes(onnx, onnx_size, opts);
        if (out && cap >= b.size()) memcpy(out, b.data(), b.size());
        return b.size();
    } catch (const fail &f) {
        g_err = f.what;
    } catch (const std::exception &e) {
        g_err = e.what();
    }
    return 0;
}

extern "C" int mars_compile_file(const char *onnx_path, const char *mars_path, const mars_compile_opts_t *opts)
{
    g_err.clear();
    if (!onnx_path || !mars_path) return g_err = "null path", -1;
    FILE *f = fopen(onnx_path, "rb");
    if (!f) return g_err = std::string("Failed to read ONNX file: ") + onnx_path, -1;
    std::vector<uint8_t> in;
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) in.insert(in.end(), buf, buf + n);
    fclose(f);
    try {
        std::vector<uint8_t> b = compile_bytes(in.data(), in.size(), opts);
        FILE *o = fopen(mars_path, "wb");
        if (!o) return g_err = std::string("Failed to create output file: ") + mars_path, -1;
        bool ok = fwrite(b.data(), 1, b.size(), o) == b.size();
        ok &= fclose(o) == 0;
        if (!ok) return g_err = std::string("short write: ") + mars_path, -1;
        return 0;
    } catch (const fail &e) {
        g_err = e.what;
    } catch (const std::exception &e) {
        g_err = e.what();
    }
    return -1;
}

extern "C" const char *mars_compile_last_error(void) { return g_err.c_str(); }
