// Jump-ahead of the Mersenne Twister MT19937 (host side): characteristic polynomial, x^n modulo it.
//
// The reference's irng = 1 generator (src/mcica_random_numbers.f90:77-306) is ONE MT19937 stream over all (sub-column, column, layer)
// draws of a call.  The word sequence x_k of MT19937 is a linear recurrence over GF(2) on the 19937-bit state
//      x_{k+624} = x_{k+397} ^ twist(upper bit of x_k, lower 31 bits of x_{k+1}),
// so the state n steps on is g(A) applied to the state now, with A the one-step map and g(x) = x^n mod phi(x), phi the characteristic
// polynomial (degree 19937, primitive).  Horner's scheme evaluates g(A) s with deg g one-step advances and as many conditional XORs of
// the whole state (Haramoto, Matsumoto, Nishimura, Panneton, L'Ecuyer, "Efficient jump ahead for F2-linear random number
// generators", 2008).  The device applies the polynomials (kernels.hip, k_mt_jump); this header builds them:
//   phi        from 2 x 19937 bits of the sequence by Berlekamp-Massey (any non-zero bit sequence of a primitive recurrence has phi as
//              its minimal polynomial);
//   x^n mod phi  by squaring.
// Test infrastructure it is not: the product's irng = 1 path uses it (driver.hip, mt_states).
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <vector>

namespace mtj {

constexpr int NW = 624;            // words of the state
constexpr int DEG = 19937;         // degree of the characteristic polynomial
constexpr int PW = 312;            // 64-bit words of a polynomial of degree <= 19967
using Poly = std::array<uint64_t, PW>;

inline bool bit(const uint64_t *p, int i) { return (p[i >> 6] >> (i & 63)) & 1u; }
inline void flip(uint64_t *p, int i) { p[i >> 6] ^= 1ull << (i & 63); }

// one step of the recurrence on a state in canonical order (st[0] = x_k ... st[623] = x_{k+623})
inline uint32_t next_word(uint32_t x0, uint32_t x1, uint32_t x397)
{
    const uint32_t y = (x0 & 0x80000000u) | (x1 & 0x7fffffffu);
    return x397 ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// phi(x): Berlekamp-Massey on the lowest bit of 2 * DEG + 64 words of the sequence
inline Poly char_poly()
{
    const int nbits = 2 * DEG + 64;
    std::vector<uint32_t> x(NW + nbits + 1);
    x[0] = 19650218u;
    for (int i = 1; i < NW; i++) x[i] = 1812433253u * (x[i - 1] ^ (x[i - 1] >> 30)) + (uint32_t)i;
    for (int k = 0; k < nbits + 1; k++) x[k + NW] = next_word(x[k], x[k + 1], x[k + 397]);
    // connection polynomial C(x) = 1 + c_1 x + ... + c_L x^L with s[n] = sum_i c_i s[n-i]
    constexpr int W = PW + 2;
    std::vector<uint64_t> C(W, 0), B(W, 0), T(W, 0), R(W, 0);      // R bit i = s[n-1-i]
    C[0] = 1; B[0] = 1;
    int L = 0, m = 1;
    for (int n = 0; n < nbits; n++) {
        const unsigned sn = x[n + 1] & 1u;                          // (from x_1: the lower bits of x_0 are not part of the state)
        unsigned d = sn;
        {   // sum_{i=1..L} c_i s[n-i] = parity((C >> 1) & R)
            uint64_t acc = 0;
            for (int w = 0; w < W - 1; w++) acc ^= ((C[w] >> 1) | (C[w + 1] << 63)) & R[w];
            d ^= (unsigned)__builtin_parityll(acc);
        }
        if (d) {
            const bool grow = 2 * L <= n;
            if (grow) T = C;
            const int wo = m >> 6, bo = m & 63;                     // C ^= B << m
            for (int w = W - 1 - wo; w >= 0; w--) {
                C[w + wo] ^= B[w] << bo;
                if (bo && w + wo + 1 < W) C[w + wo + 1] ^= B[w] >> (64 - bo);
            }
            if (grow) { L = n + 1 - L; B = T; m = 1; } else m++;
        } else m++;
        for (int w = W - 1; w > 0; w--) R[w] = (R[w] << 1) | (R[w - 1] >> 63);   // R = (R << 1) | s[n]
        R[0] = (R[0] << 1) | sn;
    }
    Poly phi{};
    if (L != DEG) return phi;                                       // (all zero: the caller reports it)
    for (int i = 0; i <= DEG; i++) if (bit(C.data(), L - i)) flip(phi.data(), i);     // phi_i = c_{L-i}
    return phi;
}

// the exponents of phi below its leading term (phi has 135 terms): x^DEG = sum of x^e over them
inline std::vector<int> lower_terms(const Poly &phi)
{
    std::vector<int> e;
    for (int i = 0; i < DEG; i++) if (bit(phi.data(), i)) e.push_back(i);
    return e;
}

// p(x)^2 mod phi(x).  Squaring over GF(2) spreads the bits; the reduction replaces x^(DEG + s) by the sum of x^(e + s), 64 exponents s
// at a time from the top word down (every replacement lands strictly lower).
inline Poly sqr_mod(const Poly &p, const Poly &phi)
{
    static thread_local std::vector<int> terms;
    static thread_local Poly terms_of{};
    if (terms.empty() || terms_of != phi) { terms = lower_terms(phi); terms_of = phi; }
    uint64_t t[2 * PW + 1] = {0};
    for (int w = 0; w < PW; w++) {
        auto spread = [](uint64_t v) {                              // bit i -> bit 2 i (low 32 bits of v)
            v &= 0xffffffffull;
            v = (v | (v << 16)) & 0x0000ffff0000ffffull;
            v = (v | (v << 8)) & 0x00ff00ff00ff00ffull;
            v = (v | (v << 4)) & 0x0f0f0f0f0f0f0f0full;
            v = (v | (v << 2)) & 0x3333333333333333ull;
            v = (v | (v << 1)) & 0x5555555555555555ull;
            return v;
        };
        t[2 * w] = spread(p[w]);
        t[2 * w + 1] = spread(p[w] >> 32);
    }
    constexpr int WD = DEG >> 6, BD = DEG & 63;                     // the word and bit of x^DEG
    for (int wi = 2 * PW - 1; wi >= WD; wi--) {
        for (;;) {
            const uint64_t hw = wi == WD ? (t[wi] >> BD) << BD : t[wi];      // the part of this word at or above x^DEG
            if (!hw) break;
            t[wi] ^= hw;
            for (int e : terms) {
                const int off = 64 * wi - DEG + e;                  // bit b of hw is x^(64 wi + b) -> x^(off + b)
                if (off >= 0) {
                    const int w0 = off >> 6, b0 = off & 63;
                    t[w0] ^= hw << b0;
                    if (b0) t[w0 + 1] ^= hw >> (64 - b0);
                } else {
                    t[0] ^= hw >> (-off);                           // (only in the word of x^DEG, whose bits below BD are clear in hw)
                }
            }
        }
    }
    Poly r;
    std::memcpy(r.data(), t, sizeof(uint64_t) * PW);
    return r;
}

// x * p(x) mod phi(x)
inline Poly mulx_mod(const Poly &p, const Poly &phi)
{
    Poly r;
    for (int w = PW - 1; w > 0; w--) r[w] = (p[w] << 1) | (p[w - 1] >> 63);
    r[0] = p[0] << 1;
    if (bit(r.data(), DEG)) for (int w = 0; w < PW; w++) r[w] ^= phi[w];
    return r;
}

// x^n mod phi(x)
inline Poly pow_x(uint64_t n, const Poly &phi)
{
    Poly r{};
    r[0] = 1;
    for (int b = 63; b >= 0; b--) {
        r = sqr_mod(r, phi);
        if ((n >> b) & 1ull) r = mulx_mod(r, phi);
    }
    return r;
}

// g(A) st on the host (validation of the polynomials; the device does this in k_mt_jump)
inline void jump_host(uint32_t st[NW], const Poly &g)
{
    uint32_t acc[NW] = {0};
    int top = DEG;
    while (top > 0 && !bit(g.data(), top)) top--;
    for (int i = top; i >= 0; i--) {
        const uint32_t nw = next_word(acc[0], acc[1], acc[397]);
        std::memmove(acc, acc + 1, sizeof(uint32_t) * (NW - 1));
        acc[NW - 1] = nw;
        if (bit(g.data(), i)) for (int k = 0; k < NW; k++) acc[k] ^= st[k];
    }
    std::memcpy(st, acc, sizeof(acc));
}

}  // namespace mtj
