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

// p.sign[n & 3] = {1, -1, -1, 1}[n & 3] as a select: indexing the by-value table dynamically put the whole struct into scratch memory
// (120 bytes per lane in describe_kernel: the write amplification the round-1 counters showed)
UH_SC_HD double quadrant_sign(int n) { return (((n & 3) == 1) || ((n & 3) == 2)) ? -1.0 : 1.0; }

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
    return poly(x * quadrant_sign(n), x * x, p, n ^ 1);
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
    return poly(x * quadrant_sign(n), x * x, p, n);
}

// logf with the results of glibc >= 2.28 (sysdeps/ieee754/flt-32/e_logf.c + e_logf_data.c): Frame::predictScale
// (src/map_types/frame.h:129-136) takes ceil(log(maxDist/dist) / log(scaleFactor)) with float arguments, i.e. libm's logf.
// Table lookup on the top 4 mantissa bits (16 centres c: invc ~ 1/c, logc ~ log c), r = z*invc - 1, degree-3 polynomial
// in double.  Positive normal arguments only (distances and ratios of distances); checked exhaustively against libm on all
// 2 130 706 432 positive normal floats, with and without FMA contraction: no difference.
UH_SC_HD float logf_glibc(float x) {
    const double T[16][2] = {
        {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2}, {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},
        {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3}, {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
        {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4}, {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5},
        {0x1p+0, 0x0p+0}, {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5}, {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
        {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3}, {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3}, {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},
        {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
    const double Ln2 = 0x1.62e42fefa39efp-1;
    const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
    uint32_t ix;
#if defined(__HIP_DEVICE_COMPILE__)
    ix = __float_as_uint(x);
#else
    std::memcpy(&ix, &x, 4);
#endif
    if (ix == 0x3f800000u) return 0.f;
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) % 16u);
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    float zf;
#if defined(__HIP_DEVICE_COMPILE__)
    zf = __uint_as_float(iz);
#else
    std::memcpy(&zf, &iz, 4);
#endif
    // the 16-entry table is indexed dynamically: select instead of a private array (which would live in scratch on the GPU)
    double invc = T[0][0], logc = T[0][1];
#pragma unroll
    for (int t = 1; t < 16; t++) { invc = i == t ? T[t][0] : invc; logc = i == t ? T[t][1] : logc; }
    const double z = zf;
    const double r = z * invc - 1;
    const double y0 = logc + (double)k * Ln2;
    const double r2 = r * r;
    double y = A1 * r + A2;
    y = A0 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}

}  // namespace uh_sincosf
