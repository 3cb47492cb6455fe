/*
 * mars_compile.h -- ONNX -> .mars compile step, C-ABI.
 *
 * Restates, in C++ (thingino-accel_amd/csrc/host/mars_compile.cpp), what the reference's Rust tool does
 * (mars-compiler/src/main.rs, mars_format.rs, onnx_parser.rs): protobuf decode of the ONNX ModelProto, operator
 * mapping (main.rs:76-103), QDQ scale harvest (:137-260), per-node translation with the shape / scale rules of
 * process_conv ... process_softmax (:677-1461), max-abs / 127 weight quantisation (:621-677), OIHW -> OHWI for
 * --nhwc (mars_format.rs:407-434), scale propagation (:312-405) and the writer (:1463-1522).  Host-only: no GPU needed.
 *
 * PARITY UNPINNED: the reference's compiler is Rust and cannot run in this image, and the reference tree holds no
 * (ONNX, .mars) pair made by the checked-in compiler (its shipped .mars files come from an older one).
 * What is pinned: the on-disk structs / enums (include/mars.h, which the runtime shares with the reference), the
 * operator table, the quantisation arithmetic on seeded tensors (tests/test_compile.py restates main.rs:621-677 in
 * numpy), and that the output loads and runs bit-identically on the oracle and on the GPU.
 */
#ifndef MARS_COMPILE_H
#define MARS_COMPILE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int float32; /* --float32: keep float weights, no int8 quantisation (main.rs:60-62) */
    int nhwc;    /* --nhwc: NHWC features and OHWI weights (main.rs:64-67); default NCHW / OIHW */
    int verbose; /* -v: progress on stderr */
} mars_compile_opts_t;

/* Compile an ONNX model held in memory.  Returns the size of the .mars file and writes it to `out` when it fits in `cap`
 * (call with out = NULL / cap = 0 to size); 0 = failure, text in mars_compile_last_error(). */
size_t mars_compile_onnx(const void *onnx, size_t onnx_size, const mars_compile_opts_t *opts, void *out, size_t cap);
/* File to file, the reference CLI's job (`mars -i in.onnx -o out.mars [--float32] [--nhwc] [-v]`): 0 = ok */
int mars_compile_file(const char *onnx_path, const char *mars_path, const mars_compile_opts_t *opts);
const char *mars_compile_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
