// Jump-ahead constants of the kissvec stream (host side; plain C++, checked on the CPU by tests/test_kissjump.py).
//
// kissvec (src/mcica_subcol_gen_lw.f90:711-745) adds four generators, each a linear map of its own 32-bit state:
//   a  congruential (mod 2^32)          a -> A a + B with (A, B) the n-fold composition;
//   b  xorshift 13 / 17 / 5             linear over GF(2): the images X[i] of the 32 unit vectors under n steps;
//   c, d  multiply-with-carry (18000 / 30903, base 2^16)   the word s = carry 2^16 + value satisfies 2^16 s' = s (mod m),
//        m = multiplier 2^16 - 1, i.e. s' = multiplier s (mod m).  Two real steps bring any 32-bit word into [0, m]; 0 and m are fixed
//        points and every other word stays in [1, m-1], where the residue names the word: n >= 2 steps are two steps and a
//        multiplication by P = multiplier^(n-2) mod m.
// The device applies them (kernels.hip, kiss_jump).
#pragma once
#include <algorithm>
#include <cstdint>

struct KissJump { unsigned n, A, B, Pc, Pd, pad[3]; unsigned X[32]; };      // n draws ahead
constexpr unsigned KISS_MC = 18000u * 65536u - 1u, KISS_MD = 30903u * 65536u - 1u;

// n-fold composition of a -> 69069 a + 1327217885 (mod 2^32), by squaring
inline void kiss_lcg_pow(unsigned long long n, unsigned &A, unsigned &B)
{
    unsigned a = 69069u, b = 1327217885u;
    A = 1u; B = 0u;
    for (; n; n >>= 1) {
        if (n & 1ull) { A = a * A; B = a * B + b; }          // x -> a (A x + B) + b
        b = a * b + b; a = a * a;
    }
}
// the xorshift 13/17/5 step as a 32 x 32 matrix over GF(2), held as the images of the unit vectors; products and powers of it
struct Gf2 { unsigned col[32]; };
inline unsigned gf2_apply(const Gf2 &M, unsigned v) { unsigned r = 0u; for (int i = 0; i < 32; i++) if ((v >> i) & 1u) r ^= M.col[i]; return r; }
inline Gf2 gf2_mul(const Gf2 &A, const Gf2 &B) { Gf2 C; for (int i = 0; i < 32; i++) C.col[i] = gf2_apply(A, B.col[i]); return C; }
inline Gf2 kiss_xorshift_pow(unsigned long long n)
{
    Gf2 R, S;
    for (int i = 0; i < 32; i++) {
        R.col[i] = 1u << i;
        unsigned b = 1u << i;
        b ^= b << 13; b ^= b >> 17; b ^= b << 5;
        S.col[i] = b;
    }
    for (; n; n >>= 1) { if (n & 1ull) R = gf2_mul(S, R); S = gf2_mul(S, S); }
    return R;
}
inline unsigned kiss_modpow(unsigned base, unsigned long long n, unsigned m)
{
    unsigned long long r = 1ull, b = base % m;
    for (; n; n >>= 1) { if (n & 1ull) r = r * b % m; b = b * b % m; }
    return (unsigned)r;
}
inline KissJump kiss_jump_entry(unsigned long long n)
{
    KissJump J{};
    J.n = (unsigned)std::min<unsigned long long>(n, 0xffffffffull);
    kiss_lcg_pow(n, J.A, J.B);
    const Gf2 X = kiss_xorshift_pow(n);
    for (int i = 0; i < 32; i++) J.X[i] = X.col[i];
    J.Pc = n >= 2 ? kiss_modpow(18000u, n - 2, KISS_MC) : 1u;
    J.Pd = n >= 2 ? kiss_modpow(30903u, n - 2, KISS_MD) : 1u;
    return J;
}
