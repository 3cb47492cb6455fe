// mars_synth.cpp -- writer for well-formed synthetic .mars graphs.
//
// The reference repo does not ship yolov5s_int8.mars / yolov5s_float32.mars
// (reference .MISSING_LARGE_BLOBS:15-17) and its compiler is Rust (absent
// here), so the workloads BASELINE.json names are synthesised: same on-disk
// format (reference include/mars.h:103-221, cross-checked against
// mars-compiler/src/mars_format.rs:93-397), the YOLOv5 v6 layer sequence as
// exported to ONNX (SiLU = Sigmoid + Mul), seeded int8
// weights, and scales chosen so activations neither vanish nor saturate.
// Unlike the shipped files these graphs are well-formed for the executor:
// tag 7 (NHWC) activations, OHWI weights, real int32 biases, SAME padding
// (the only mode the executor honours, reference mars_runtime.c:592-598).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "mars_hip.h"

namespace {

struct Rng { // splitmix64: identical bytes on every platform
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    int range(int lo, int hi) { return lo + (int)(next() % (uint64_t)(hi - lo + 1)); }
    float unit() { return (float)((next() >> 40) * (1.0 / 16777216.0)); }
};

struct Builder {
    bool f32, nchw;
    Rng rng;
    std::vector<mars_tensor_t> tensors;
    std::vector<mars_layer_t> layers;
    std::vector<uint8_t> blob;
    // activation statistics model (see header comment): every post-SiLU tensor
    // shares one scale and has RMS ~ 33 int8 steps
    static constexpr float kConvOutScale = 0.03125f;  // q1 * this ~ N(0, 1.5)
    static constexpr float kSigScale = 1.0f / 127.0f;
    static constexpr float kActScale = 4.0f / 127.0f; // SiLU output
    static constexpr float kHeadScale = 0.05f;
    static constexpr float kInScale = 1.0f / 64.0f;

    bool vary = false; // per-convolution scales (x0.75 .. x1.35 of the nominal ones): every fused table differs
    float jit() { return vary ? 0.75f + 0.6f * rng.unit() : 1.0f; }
    Builder(bool f, bool n, unsigned seed) : f32(f), nchw(n || f), rng(seed) {}

    struct T { int id; int c, h, w; float rms; };

    int add_tensor(const std::string &name, mars_dtype_t dt, mars_format_t fmt,
                   std::vector<int> shape, float scale, const void *data, size_t bytes) {
        mars_tensor_t t;
        std::memset(&t, 0, sizeof(t));
        t.id = (uint32_t)tensors.size();
        std::snprintf(t.name, sizeof(t.name), "%s", name.c_str());
        t.dtype = dt;
        t.format = fmt;
        t.ndims = (uint32_t)shape.size();
        for (size_t i = 0; i < shape.size(); i++) t.shape[i] = shape[i];
        t.scale = scale;
        if (data) {
            while (blob.size() % 4) blob.push_back(0); // 4-byte aligned entries (main.rs:611-619)
            t.data_offset = blob.size();
            t.data_size = bytes;
            const uint8_t *p = (const uint8_t *)data;
            blob.insert(blob.end(), p, p + bytes);
        }
        tensors.push_back(t);
        return (int)t.id;
    }

    T act(const std::string &name, int c, int h, int w, float scale, float rms) {
        mars_dtype_t dt = f32 ? MARS_DTYPE_FLOAT32 : MARS_DTYPE_INT8;
        int id = nchw ? add_tensor(name, dt, MARS_FORMAT_NCHW, {1, c, h, w}, scale, nullptr, 0)
                      : add_tensor(name, dt, MARS_FORMAT_NHWC, {1, h, w, c}, scale, nullptr, 0);
        return T{id, c, h, w, rms};
    }

    mars_layer_t &layer(mars_layer_type_t type, std::vector<int> ins, int out) {
        mars_layer_t l;
        std::memset(&l, 0, sizeof(l));
        l.id = (uint32_t)layers.size();
        l.type = type;
        l.num_inputs = (uint32_t)ins.size();
        l.num_outputs = 1;
        for (size_t i = 0; i < ins.size() && i < 4; i++) l.input_tensor_ids[i] = (uint32_t)ins[i];
        l.output_tensor_ids[0] = (uint32_t)out;
        layers.push_back(l);
        return layers.back();
    }

    float scale_of(const T &t) const { return tensors[t.id].scale; }

    // conv (+ optional SiLU as Sigmoid+Mul).  relu: fused ReLU activation instead.
    T conv(const T &x, int cout, int k, int s, bool silu, bool relu, float out_scale_override = 0.f) {
        const int K = k * k * x.c;
        const int oh = (x.h + s - 1) / s, ow = (x.w + s - 1) / s;
        const std::string base = "conv" + std::to_string(layers.size());
        int wid, bid;
        float w_scale = 0.01f;
        const float out_scale = (out_scale_override > 0 ? out_scale_override : kConvOutScale) * jit();
        if (f32) {
            std::vector<float> w((size_t)cout * K);
            const float a = 1.7f / std::sqrt((float)K); // keeps unit-ish variance through the net
            for (auto &v : w) v = (rng.unit() * 2.f - 1.f) * a;
            std::vector<float> b(cout);
            for (auto &v : b) v = (rng.unit() * 2.f - 1.f) * 0.1f;
            wid = add_tensor(base + ".w", MARS_DTYPE_FLOAT32, MARS_FORMAT_OIHW, {cout, x.c, k, k}, 1.f,
                             w.data(), w.size() * 4);
            bid = add_tensor(base + ".b", MARS_DTYPE_FLOAT32, MARS_FORMAT_D1, {cout}, 1.f, b.data(),
                             b.size() * 4);
        } else {
            std::vector<int8_t> w((size_t)cout * K);
            for (auto &v : w) v = (int8_t)rng.range(-127, 127);
            // sigma(acc) = 73.3 * sqrt(K) * rms_in; aim the int8 result at sigma 48
            const float sigma_acc = 73.3f * std::sqrt((float)K) * x.rms;
            const float cs = 48.0f / sigma_acc;
            w_scale = cs * out_scale / scale_of(x);
            std::vector<int32_t> b(cout);
            const int half = (int)(sigma_acc * 0.5f) + 1;
            for (auto &v : b) v = rng.range(-half, half);
            if (nchw)
                wid = add_tensor(base + ".w", MARS_DTYPE_INT8, MARS_FORMAT_OIHW, {cout, x.c, k, k}, w_scale,
                                 w.data(), w.size());
            else
                wid = add_tensor(base + ".w", MARS_DTYPE_INT8, MARS_FORMAT_OHWI, {cout, k, k, x.c}, w_scale,
                                 w.data(), w.size());
            bid = add_tensor(base + ".b", MARS_DTYPE_INT32, MARS_FORMAT_D1, {cout}, 1.f, b.data(), b.size() * 4);
        }
        T q1 = act(base + ".out", cout, oh, ow, f32 ? 1.f : out_scale, 48.f);
        mars_layer_t &l = layer(MARS_LAYER_CONV2D, {x.id}, q1.id);
        mars_conv_params_t &cp = l.params.conv;
        cp.kernel_h = cp.kernel_w = (uint32_t)k;
        cp.stride_h = cp.stride_w = (uint32_t)s;
        cp.dilation_h = cp.dilation_w = 1;
        cp.padding = MARS_PAD_SAME;
        cp.pad_top = cp.pad_left = cp.pad_bottom = cp.pad_right = (uint32_t)(k / 2);
        cp.groups = 1;
        cp.activation = relu ? MARS_ACT_RELU : MARS_ACT_NONE;
        cp.weight_tensor_id = (uint32_t)wid;
        cp.bias_tensor_id = (uint32_t)bid;
        if (!silu) return q1;
        T q2 = act(base + ".sig", cout, oh, ow, f32 ? 1.f : kSigScale * jit(), 80.f);
        layer(MARS_LAYER_SIGMOID, {q1.id}, q2.id);
        T q3 = act(base + ".silu", cout, oh, ow, f32 ? 1.f : kActScale * jit(), 33.f);
        layer(MARS_LAYER_MUL, {q1.id, q2.id}, q3.id);
        return q3;
    }

    T add(const T &a, const T &b) {
        T o = act("add" + std::to_string(layers.size()), a.c, a.h, a.w, f32 ? 1.f : kActScale * jit(), 45.f);
        layer(MARS_LAYER_ADD, {a.id, b.id}, o.id);
        return o;
    }

    T concat(const std::vector<T> &xs) {
        int c = 0;
        float r2 = 0;
        std::vector<int> ids;
        for (auto &x : xs) { c += x.c; r2 += x.rms * x.rms * x.c; ids.push_back(x.id); }
        T o = act("cat" + std::to_string(layers.size()), c, xs[0].h, xs[0].w, scale_of(xs[0]),
                  std::sqrt(r2 / c));
        mars_layer_t &l = layer(MARS_LAYER_CONCAT, ids, o.id);
        l.params.concat.axis = nchw ? 1 : 3;
        l.params.concat.num_inputs = (uint32_t)xs.size();
        return o;
    }

    T maxpool(const T &x, int k) {
        T o = act("mp" + std::to_string(layers.size()), x.c, x.h, x.w, scale_of(x), x.rms * 1.3f);
        mars_layer_t &l = layer(MARS_LAYER_MAXPOOL, {x.id}, o.id);
        l.params.pool.kernel_h = l.params.pool.kernel_w = (uint32_t)k;
        l.params.pool.stride_h = l.params.pool.stride_w = 1;
        l.params.pool.padding = MARS_PAD_EXPLICIT; // carried; the executor ignores pool padding
        l.params.pool.pad_top = l.params.pool.pad_bottom = l.params.pool.pad_left = l.params.pool.pad_right = (uint32_t)(k / 2);
        return o;
    }

    T upsample(const T &x) {
        T o = act("up" + std::to_string(layers.size()), x.c, x.h * 2, x.w * 2, scale_of(x), x.rms);
        mars_layer_t &l = layer(MARS_LAYER_UPSAMPLE, {x.id}, o.id);
        l.params.upsample.scale_h = l.params.upsample.scale_w = 2;
        l.params.upsample.mode = 0;
        return o;
    }

    T bottleneck(const T &x, bool shortcut) {
        T y = conv(x, x.c, 1, 1, true, false);
        y = conv(y, x.c, 3, 1, true, false);
        return shortcut ? add(x, y) : y;
    }

    T c3(const T &x, int c2, int n, bool shortcut) {
        const int ch = c2 / 2;
        T a = conv(x, ch, 1, 1, true, false);
        T b = conv(x, ch, 1, 1, true, false);
        for (int i = 0; i < n; i++) a = bottleneck(a, shortcut);
        return conv(concat({a, b}), c2, 1, 1, true, false);
    }

    T sppf(const T &x, int c2) {
        T a = conv(x, x.c / 2, 1, 1, true, false);
        T y1 = maxpool(a, 5), y2 = maxpool(y1, 5), y3 = maxpool(y2, 5);
        return conv(concat({a, y1, y2, y3}), c2, 1, 1, true, false);
    }

    size_t serialise(const std::vector<int> &inputs, const std::vector<int> &outputs, void *buf, size_t cap) {
        mars_header_t h;
        std::memset(&h, 0, sizeof(h));
        h.magic = MARS_MAGIC;
        h.version_major = MARS_VERSION_MAJOR;
        h.version_minor = MARS_VERSION_MINOR;
        h.num_layers = (uint32_t)layers.size();
        h.num_tensors = (uint32_t)tensors.size();
        h.num_inputs = (uint32_t)inputs.size();
        h.num_outputs = (uint32_t)outputs.size();
        for (size_t i = 0; i < inputs.size() && i < 4; i++) h.input_tensor_ids[i] = (uint32_t)inputs[i];
        for (size_t i = 0; i < outputs.size() && i < 4; i++) h.output_tensor_ids[i] = (uint32_t)outputs[i];
        size_t off = sizeof(h) + tensors.size() * sizeof(mars_tensor_t) + layers.size() * sizeof(mars_layer_t);
        off = (off + 63) & ~(size_t)63;
        h.weights_offset = off;
        h.weights_size = blob.size();
        const size_t total = off + blob.size();
        if (!buf || cap < total) return total;
        uint8_t *p = (uint8_t *)buf;
        std::memset(p, 0, off);
        std::memcpy(p, &h, sizeof(h));
        p += sizeof(h);
        std::memcpy(p, tensors.data(), tensors.size() * sizeof(mars_tensor_t));
        p += tensors.size() * sizeof(mars_tensor_t);
        std::memcpy(p, layers.data(), layers.size() * sizeof(mars_layer_t));
        std::memcpy((uint8_t *)buf + off, blob.data(), blob.size());
        return total;
    }
};

size_t build_tiny(const mars_synth_opts_t &o, void *buf, size_t cap) {
    // shape of the shipped tiny_160 models: 3->16->32->64,
    // k3, ReLU between; here SAME-padded and well-formed
    Builder b(o.float32 != 0, o.nchw_int8 != 0, o.seed);
    b.vary = o.vary_scales != 0;
    const int hw = o.input_hw > 0 ? o.input_hw : 160;
    Builder::T x = b.act("input", 3, hw, hw, o.float32 ? 1.f : Builder::kInScale, 74.f);
    Builder::T y = b.conv(x, 16, 3, 1, false, true, Builder::kActScale);
    y.rms = 30.f;
    y = b.conv(y, 32, 3, 1, false, true, Builder::kActScale);
    y.rms = 30.f;
    y = b.conv(y, 64, 3, 1, false, false, Builder::kActScale);
    return b.serialise({x.id}, {y.id}, buf, cap);
}

size_t build_yolov5(const mars_synth_opts_t &o, void *buf, size_t cap) {
    Builder b(o.float32 != 0, o.nchw_int8 != 0, o.seed);
    b.vary = o.vary_scales != 0;
    const int hw = o.input_hw > 0 ? o.input_hw : 640;
    const int wm = o.width_x16 > 0 ? o.width_x16 : 8;
    const int dm = o.depth_x3 > 0 ? o.depth_x3 : 1;
    auto ch = [&](int c) { return c * wm / 16; };
    auto dep = [&](int n) { int v = (n * dm + 1) / 3; return v < 1 ? 1 : v; }; // round(n*depth), min 1
    using T = Builder::T;
    T x = b.act("images", 3, hw, hw, o.float32 ? 1.f : Builder::kInScale, 74.f);
    // backbone
    T p1 = b.conv(x, ch(64), 6, 2, true, false);
    T p2 = b.conv(p1, ch(128), 3, 2, true, false);
    T c2 = b.c3(p2, ch(128), dep(3), true);
    T p3 = b.conv(c2, ch(256), 3, 2, true, false);
    T c4 = b.c3(p3, ch(256), dep(6), true);
    T p4 = b.conv(c4, ch(512), 3, 2, true, false);
    T c6 = b.c3(p4, ch(512), dep(9), true);
    T p5 = b.conv(c6, ch(1024), 3, 2, true, false);
    T c8 = b.c3(p5, ch(1024), dep(3), true);
    T s9 = b.sppf(c8, ch(1024));
    // head
    T h10 = b.conv(s9, ch(512), 1, 1, true, false);
    T h13 = b.c3(b.concat({b.upsample(h10), c6}), ch(512), dep(3), false);
    T h14 = b.conv(h13, ch(256), 1, 1, true, false);
    T h17 = b.c3(b.concat({b.upsample(h14), c4}), ch(256), dep(3), false);
    T h18 = b.conv(h17, ch(256), 3, 2, true, false);
    T h20 = b.c3(b.concat({h18, h14}), ch(512), dep(3), false);
    T h21 = b.conv(h20, ch(512), 3, 2, true, false);
    T h23 = b.c3(b.concat({h21, h10}), ch(1024), dep(3), false);
    // detect: one 1x1 conv per scale, 3 anchors x 85 = 255 channels, shared output scale
    T d0 = b.conv(h17, 255, 1, 1, false, false, Builder::kHeadScale);
    T d1 = b.conv(h20, 255, 1, 1, false, false, Builder::kHeadScale);
    T d2 = b.conv(h23, 255, 1, 1, false, false, Builder::kHeadScale);
    // the exported graphs continue with Reshape/Transpose, which the executor
    // accepts and ignores (reference mars_runtime.c:1203-1213): keep one of each
    T r0 = b.act("reshape_out", 85, d2.h * 3, d2.w, Builder::kHeadScale, 1.f);
    b.layer(MARS_LAYER_RESHAPE, {d2.id}, r0.id);
    T r1 = b.act("transpose_out", 85, d2.h * 3, d2.w, Builder::kHeadScale, 1.f);
    b.layer(MARS_LAYER_TRANSPOSE, {r0.id}, r1.id);
    return b.serialise({x.id}, {d0.id, d1.id, d2.id}, buf, cap);
}

} // namespace

extern "C" size_t mars_synth_model(const mars_synth_opts_t *opts, void *buf, size_t cap) {
    if (!opts) return 0;
    if (opts->input_hw < 0 || (opts->tiny == 0 && opts->input_hw % 32 != 0)) return 0;
    return opts->tiny ? build_tiny(*opts, buf, cap) : build_yolov5(*opts, buf, cap);
}
