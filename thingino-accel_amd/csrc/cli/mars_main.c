/*
 * mars -- command-line front of the ONNX -> .mars compile step (include/mars_compile.h).
 * Same options as the reference's tool (mars-compiler/src/main.rs:48-73):
 *   mars -i|--input model.onnx -o|--output model.mars [-f|--float32] [--nhwc] [-v|--verbose]
 * Host-only: links the compile step alone, no GPU runtime.
 */
#include <stdio.h>
#include <string.h>

#include "mars_compile.h"

static int usage(const char *argv0)
{
    fprintf(stderr, "usage: %s -i <model.onnx> -o <model.mars> [-f|--float32] [--nhwc] [-v|--verbose]\n", argv0);
    return 2;
}

int main(int argc, char **argv)
{
    const char *in = NULL, *out = NULL;
    mars_compile_opts_t o = {0, 0, 0};
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if ((!strcmp(a, "-i") || !strcmp(a, "--input")) && i + 1 < argc)
            in = argv[++i];
        else if ((!strcmp(a, "-o") || !strcmp(a, "--output")) && i + 1 < argc)
            out = argv[++i];
        else if (!strcmp(a, "-f") || !strcmp(a, "--float32"))
            o.float32 = 1;
        else if (!strcmp(a, "--nhwc"))
            o.nhwc = 1;
        else if (!strcmp(a, "-v") || !strcmp(a, "--verbose"))
            o.verbose = 1;
        else
            return usage(argv[0]);
    }
    if (!in || !out) return usage(argv[0]);
    printf("Input:  %s\nOutput: %s\n", in, out);
    printf("Quantization: %s\n", o.float32 ? "FLOAT32 (no quantization)" : "INT8");
    printf("Feature format: %s\n", o.nhwc ? "NHWC (channels-last)" : "NCHW (channels-first)");
    if (mars_compile_file(in, out, &o) != 0) {
        fprintf(stderr, "Error: %s\n", mars_compile_last_error());
        return 1;
    }
    printf("Compilation complete: %s\n", out);
    return 0;
}
