/* ASan / UBSan driver for the ONNX -> .mars compile step: every file given is compiled whole, then under seeded
 * truncations, byte flips and varint splices.  A refusal (0 + error text) is fine; a sanitizer report or a crash is not.
 *   fuzz_compile <iterations> <seed> file.onnx ... */
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mars_compile.h"

static uint32_t rng_state;
static uint32_t rnd() { return rng_state = rng_state * 1664525u + 1013904223u; }

static int one(const std::vector<uint8_t> &b, int mode)
{
    mars_compile_opts_t o = {mode & 1, (mode >> 1) & 1, 0};
    size_t n = mars_compile_onnx(b.data(), b.size(), &o, nullptr, 0);
    if (n == 0) return mars_compile_last_error()[0] ? 0 : 1; /* a refusal must say why */
    std::vector<uint8_t> out(n);
    return mars_compile_onnx(b.data(), b.size(), &o, out.data(), n) == n ? 0 : 1;
}

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    int iters = atoi(argv[1]);
    rng_state = (uint32_t)atoi(argv[2]);
    int bad = 0, refused = 0;
    for (int a = 3; a < argc; a++) {
        FILE *f = fopen(argv[a], "rb");
        if (!f) return 2;
        std::vector<uint8_t> file;
        uint8_t buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + n);
        fclose(f);
        for (int mode = 0; mode < 4; mode++) bad += one(file, mode);
        for (int i = 0; i < iters; i++) {
            std::vector<uint8_t> m = file;
            switch (rnd() % 4) {
            case 0: m.resize(rnd() % (m.size() + 1)); break;
            case 1:
                for (int k = 0, nk = 1 + rnd() % 8; k < nk && !m.empty(); k++) m[rnd() % m.size()] ^= (uint8_t)(1u << (rnd() % 8));
                break;
            case 2: /* an over-long varint or a huge length somewhere */
                if (m.size() > 12) {
                    size_t at = rnd() % (m.size() - 11);
                    for (int k = 0; k < 10; k++) m[at + k] = 0xff;
                    m[at + 10] = (uint8_t)(rnd() % 4);
                }
                break;
            default: /* within the first 4 KiB, where the graph / node headers live */
                if (!m.empty()) m[rnd() % (m.size() < 4096 ? m.size() : 4096)] = (uint8_t)rnd();
                break;
            }
            int r = one(m, (int)(rnd() % 4));
            bad += r;
            refused += mars_compile_last_error()[0] != 0;
        }
    }
    printf("fuzz_compile: %d failures, %d refusals\n", bad, refused);
    return bad ? 1 : 0;
}
