// Host check of csrc/glibc_sincosf.hpp against the C library (tests/test_orb_oracle.py): every `step`-th float of [0, 6.4].
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../ucoslam-cv3_amd/csrc/glibc_sincosf.hpp"
int main(int argc, char** argv) {
    const unsigned step = argc > 1 ? (unsigned)std::atoi(argv[1]) : 13;
    float top = 6.4f;
    uint32_t hi;
    std::memcpy(&hi, &top, 4);
    unsigned long long n = 0, bad = 0;
    for (uint32_t u = 0; u <= hi; u += step) {
        float x;
        std::memcpy(&x, &u, 4);
        const float a = cosf(x), b = uh_sincosf::cosf_glibc(x), c = sinf(x), d = uh_sincosf::sinf_glibc(x);
        if (std::memcmp(&a, &b, 4) || std::memcmp(&c, &d, 4)) { if (bad < 5) std::printf("x=%a cos %a/%a sin %a/%a\n", x, a, b, c, d); bad++; }
        n++;
    }
    for (uint32_t u = 0x00800000u; u < 0x7f800000u; u += 2 * step + 3) {   // positive normal floats: logf
        float x;
        std::memcpy(&x, &u, 4);
        const float a = logf(x), b = uh_sincosf::logf_glibc(x);
        if (std::memcmp(&a, &b, 4)) { if (bad < 5) std::printf("x=%a log %a/%a\n", x, a, b); bad++; }
        n++;
    }
    std::printf("checked %llu mismatches %llu\n", n, bad);
    return bad ? 1 : 0;
}
