// cosf / sinf with the results of glibc >= 2.28 (sysdeps/ieee754/flt-32/s_cosf.c, s_sinf.c, sincosf.h — the ARM optimized-routines
// algorithm), for |x| < 120.  Why it exists: computeOrbDescriptor rotates the 512 BRIEF sample offsets with
// a = (float)cos(angle), b = (float)sin(angle) (src/featureextractors/ORBextractor.cpp:117-119) and rounds x*b + y*a half-to-even;
// on the reference's platform those are libm's float routines, whose last bit is not that of a correctly rounded cosine.  One
// ulp in a or b flips a descriptor bit once per ~5 million descriptors (found by scripts/fuzz_parity.py), so the GPU
// evaluates the SAME published algorithm: reduction x - n*pi/2 in double with n from a scaled float-to-int conversion, then
// the degree-8 / degree-7 minimax polynomials in double, in glibc's operation order, without contraction.
// tests/test_orb_oracle.py compares this header, compiled for the host, with the C library on every 13th float of
// [0, 6.4] (the whole range was checked once, exhaustively: 1 087 163 598 values, zero differences, with and without FMA).
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define UH_SC_HD __host__ __device__ __forceinline__
#else
#define UH_SC_HD inline
#endif

namespace uh_sincosf {

struct Table { double sign[4]; double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3; };

UH_SC_HD Table table(int neg) {   // __sincosf_table[0] / [1] (negated cosine polynomial)
    const double sg = neg ? -1.0 : 1.0;
    return Table{{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0,
                 sg * 0x1p0, sg * -0x1.ffffffd0c621cp-2, sg * 0x1.55553e1068f19p-5, sg * -0x1.6c087e89a359dp-10, sg * 0x1.99343027bf8c3p-16,
                 -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13};
}

UH_SC_HD uint32_t abstop12(float x) {
    uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
    u = __float_as_uint(x);
#else
    std::memcpy(&u, &x, 4);
#endif
    return (u >> 20) & 0x7ff;
}

UH_SC_HD float poly(double x, double x2, const Table& p, int n) {   // sinf_poly
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double s1 = p.s2 + x2 * p.s3;
        const double x7 = x3 * x2;
        const double s = x + x3 * p.s1;
        return (float)(s + x7 * s1);
    }
    const double x4 = x2 * x2;
    const double c2 = p.c3 + x2 * p.c4;
    const double c1 = p.c1 + x2 * p.c2;
    const double x6 = x4 * x2;
    const double c = p.c0 + x2 * c1;
    return (float)(c + x6 * c2);
}

UH_SC_HD double reduce_fast(double x, const Table& p, int& n) {
    const double r = x * p.hpi_inv;
    n = ((int32_t)r + 0x800000) >> 24;
    return x - n * p.hpi;
}

// valid for |y| < 120 (the ORB angles are in [0, 2*pi))
UH_SC_HD float cosf_glibc(float y) {
    double x = y;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f)) return 1.0f;
        return poly(x, x * x, table(0), 1);
    }
    int n;
    x = reduce_fast(x, table(0), n);
    const Table p = table((n & 2) != 0);
    return poly(x * p.sign[n & 3], x * x, p, n ^ 1);
}

UH_SC_HD float sinf_glibc(float y) {
    double x = y;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f)) return y;
        return poly(x, x * x, table(0), 0);
    }
    int n;
    x = reduce_fast(x, table(0), n);
    const Table p = table((n & 2) != 0);
    return poly(x * p.sign[n & 3], x * x, p, n);
}

}  // namespace uh_sincosf
