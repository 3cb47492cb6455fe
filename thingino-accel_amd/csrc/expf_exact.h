/*
 * expf_exact.h -- float expf() restated after the algorithm glibc (>= 2.28) uses on x86-64
 * (Szabolcs Nagy's expf: double-precision range reduction by 1/32, 32-entry 2^(i/32) table,
 * degree-3 polynomial, one final rounding to float).  Purpose: the reference's float32
 * sigmoid layer (reference src/mars/mars_runtime.c:747) calls the HOST libm; byte-moving layers
 * downstream (max-pool on float BYTES, :919-957) make any last-bit difference chaotic, so the GPU
 * must reproduce libm's float result exactly, not approximately.
 * Shared by the HIP element-wise kernel and a host test that compares it with libm over a dense
 * sweep of all float inputs (tests/test_expf_exact.py).  The table is 2^(i/32) correctly rounded to
 * double with the exponent contribution of i pre-subtracted (generated, not copied).
 * Build with -ffp-contract=off.
 */
#ifndef EXPF_EXACT_H
#define EXPF_EXACT_H
#include <stdint.h>
#include <string.h>

#ifdef __HIPCC__
#define EXPF_EXACT_FN __host__ __device__ static inline
#define EXPF_EXACT_TAB __device__ __constant__ static const
#else
#define EXPF_EXACT_FN static inline
#define EXPF_EXACT_TAB static const
#endif

EXPF_EXACT_TAB uint64_t expf_exact_tab[32] = {
    0x3ff0000000000000ULL,
    0x3fefd9b0d3158574ULL,
    0x3fefb5586cf9890fULL,
    0x3fef9301d0125b51ULL,
    0x3fef72b83c7d517bULL,
    0x3fef54873168b9aaULL,
    0x3fef387a6e756238ULL,
    0x3fef1e9df51fdee1ULL,
    0x3fef06fe0a31b715ULL,
    0x3feef1a7373aa9cbULL,
    0x3feedea64c123422ULL,
    0x3feece086061892dULL,
    0x3feebfdad5362a27ULL,
    0x3feeb42b569d4f82ULL,
    0x3feeab07dd485429ULL,
    0x3feea47eb03a5585ULL,
    0x3feea09e667f3bcdULL,
    0x3fee9f75e8ec5f74ULL,
    0x3feea11473eb0187ULL,
    0x3feea589994cce13ULL,
    0x3feeace5422aa0dbULL,
    0x3feeb737b0cdc5e5ULL,
    0x3feec49182a3f090ULL,
    0x3feed503b23e255dULL,
    0x3feee89f995ad3adULL,
    0x3feeff76f2fb5e47ULL,
    0x3fef199bdd85529cULL,
    0x3fef3720dcef9069ULL,
    0x3fef5818dcfba487ULL,
    0x3fef7c97337b9b5fULL,
    0x3fefa4afa2a490daULL,
    0x3fefd0765b6e4540ULL
};

EXPF_EXACT_FN double expf_exact_asdouble(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
EXPF_EXACT_FN uint64_t expf_exact_asuint64(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
EXPF_EXACT_FN uint32_t expf_exact_asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

EXPF_EXACT_FN float expf_exact(float x, const uint64_t *tab) {
    const double InvLn2N = 0x1.71547652b82fep+0 * 32.0;
    const double Shift = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0;
    const double C1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0;
    const double C2 = 0x1.62e42ff0c52d6p-1 / 32.0;
    const uint32_t abstop = (expf_exact_asuint(x) >> 20) & 0x7ff;
    if (abstop >= 0x42b) { /* |x| >= 88 or NaN */
        if (expf_exact_asuint(x) == 0xff800000u) return 0.0f; /* -inf */
        if (abstop >= 0x7f8) return x + x;                   /* +inf, NaN */
        if (x > 0x1.62e42ep6f) return __builtin_inff();      /* overflow: x > log(0x1p128) */
        if (x < -0x1.9fe368p6f) return 0.0f;                 /* underflow: x < log(0x1p-150) */
    }
    const double xd = (double)x;
    double z;
    double kd = __builtin_fma(InvLn2N, xd, Shift);
    const uint64_t ki = expf_exact_asuint64(kd);
    kd = kd - Shift;
    const double r = __builtin_fma(InvLn2N, xd, -kd); /* exact residual: decides 2 of the 2^32 inputs */
    uint64_t t = tab[ki & 31];
    t += ki << (52 - 5);
    const double s = expf_exact_asdouble(t);
    /* every multiply-add is FUSED: on x86-64 hosts with FMA glibc dispatches (ifunc) to its -mfma
     * build of this routine, in which the compiler contracted them.  With fused steps this function
     * equals that libm for ALL 2^32 inputs (exhaustive check in the build container, 0 mismatches);
     * with separately rounded steps it differs for 2 inputs out of 2^32 (the residual r decides them) */
    z = __builtin_fma(C0, r, C1);
    const double r2 = r * r;
    double y = __builtin_fma(C2, r, 1.0);
    y = __builtin_fma(z, r2, y);
    y = y * s;
    return (float)y;
}
#endif
