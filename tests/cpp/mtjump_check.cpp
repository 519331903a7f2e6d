// Host check of rrtmg_lw_amd/csrc/mtjump.hpp (built and run by tests/test_mtjump.py): the characteristic polynomial of MT19937 found by
// Berlekamp-Massey has degree 19937 and 135 terms, and g(A) s with g = x^n mod phi equals n single steps of the recurrence, for jumps that
// are short, long, and longer than 2^32.
#include "mtjump.hpp"
#include <cstdio>
using namespace mtj;

static void step(uint32_t *st)
{
    const uint32_t nw = next_word(st[0], st[1], st[397]);
    memmove(st, st + 1, 4 * (NW - 1));
    st[NW - 1] = nw;
}
static int differ(const uint32_t *a, const uint32_t *b)
{
    int bad = 0;
    for (int i = 0; i < NW; i++) bad += (i == 0 ? ((a[i] ^ b[i]) & 0x80000000u) : (a[i] ^ b[i])) != 0;
    return bad;
}

int main()
{
    const Poly phi = char_poly();
    int deg = -1, cnt = 0;
    for (int i = 0; i < PW * 64; i++) if (bit(phi.data(), i)) { deg = i; cnt++; }
    printf("phi: degree %d, %d terms\n", deg, cnt);
    if (deg != DEG || cnt != 135) return 1;
    uint32_t s0[NW];
    s0[0] = 5489u;
    for (int i = 1; i < NW; i++) s0[i] = 1812433253u * (s0[i - 1] ^ (s0[i - 1] >> 30)) + (uint32_t)i;
    int rc = 0;
    uint32_t ref[NW];
    memcpy(ref, s0, sizeof(ref));
    unsigned long long done = 0;
    for (unsigned long long n : {0ull, 1ull, 63ull, 624ull, 19937ull, 300001ull}) {
        while (done < n) { step(ref); done++; }
        uint32_t st[NW];
        memcpy(st, s0, sizeof(st));
        jump_host(st, pow_x(n, phi));
        const int bad = differ(st, ref);
        printf("jump %llu: %d words differ\n", n, bad);
        rc |= bad != 0;
    }
    // x^(a + b) = x^a x^b: a jump beyond 2^32 as two jumps against one
    {
        const unsigned long long a = 5000000000ull, b = 123456789ull;
        uint32_t st1[NW], st2[NW];
        memcpy(st1, s0, sizeof(st1)); memcpy(st2, s0, sizeof(st2));
        jump_host(st1, pow_x(a, phi)); jump_host(st1, pow_x(b, phi));
        jump_host(st2, pow_x(a + b, phi));
        const int bad = differ(st1, st2);
        printf("jump %llu + %llu vs %llu: %d words differ\n", a, b, a + b, bad);
        rc |= bad != 0;
    }
    return rc;
}
