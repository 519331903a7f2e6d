// Host check of rrtmg_lw_amd/csrc/kissjump.hpp (built and run by tests/test_mtjump.py): the jump constants applied to a kissvec state give
// the state that n single steps give (src/mcica_subcol_gen_lw.f90:711-745), for seeds as the generator forms them (< 1e9), for arbitrary
// 32-bit words, and for the fixed points of the multiply-with-carry pair.
#include "kissjump.hpp"
#include <cstdio>
#include <random>

struct Kiss { unsigned a, b, c, d; };
static void step(Kiss &s)
{
    s.a = 69069u * s.a + 1327217885u;
    s.b ^= s.b << 13; s.b ^= s.b >> 17; s.b ^= s.b << 5;
    s.c = 18000u * (s.c & 65535u) + (s.c >> 16);
    s.d = 30903u * (s.d & 65535u) + (s.d >> 16);
}
static unsigned mwc_mul(unsigned s, unsigned P, unsigned m) { return s == m ? m : (unsigned)(((unsigned long long)s * P) % m); }
static void jump(Kiss &s, const KissJump &J)          // what kiss_jump does on the device (kernels.hip)
{
    if (J.n < 2u) { if (J.n == 1u) step(s); return; }
    s.a = J.A * s.a + J.B;
    unsigned b = 0u;
    for (int i = 0; i < 32; i++) if ((s.b >> i) & 1u) b ^= J.X[i];
    s.b = b;
    for (int q = 0; q < 2; q++) { s.c = 18000u * (s.c & 65535u) + (s.c >> 16); s.d = 30903u * (s.d & 65535u) + (s.d >> 16); }
    s.c = mwc_mul(s.c, J.Pc, KISS_MC);
    s.d = mwc_mul(s.d, J.Pd, KISS_MD);
}

int main()
{
    std::mt19937 rng(7);
    int bad = 0, cases = 0;
    const unsigned long long ns[] = {0, 1, 2, 3, 72, 144, 1152, 20163, 1000003};
    for (int trial = 0; trial < 40; trial++) {
        Kiss s0;
        if (trial < 20) s0 = {rng() % 1000000000u, rng() % 1000000000u, rng() % 1000000000u, rng() % 1000000000u};
        else if (trial < 36) s0 = {rng(), rng(), rng(), rng()};
        else s0 = {rng(), 0u, trial & 1 ? KISS_MC : 0u, trial & 2 ? KISS_MD : 0u};        // fixed points
        Kiss ref = s0;
        unsigned long long done = 0;
        for (unsigned long long n : ns) {
            while (done < n) { step(ref); done++; }
            Kiss s = s0;
            jump(s, kiss_jump_entry(n));
            cases++;
            if (s.a != ref.a || s.b != ref.b || s.c != ref.c || s.d != ref.d) { bad++; printf("trial %d n %llu differs\n", trial, n); }
        }
    }
    printf("kissvec jump: %d of %d cases differ\n", bad, cases);
    return bad != 0;
}
