// mt19937 jump-ahead for the native torch-RNG stream (ls_torch_rng.cpp): the state `J` words further on WITHOUT running the recurrence
// over them, so that the generator threads of a long fill each start from their own jumped state and no thread has to walk the whole
// stream first (round 5's "scout": one sequential pass of the 624-word block update over every word of the fill, 0.26 ns per word --
// 1.35 ms per step at BEAT B = 256, above the step kernel's 0.79 ms).
//
// Mathematics (Haramoto, Matsumoto, Nishimura, Panneton, L'Ecuyer, "Efficient jump ahead for F2-linear random number generators", 2008 --
// restated, no code taken): the state sequence is linear over GF(2) with characteristic polynomial phi of degree 19937, so every bit of
// the untempered word sequence x[n] obeys  sum_i phi_i x[n + i] = 0,  and with  g_J(t) = t^J mod phi(t)
//         x[J + j] = XOR over { i : coefficient i of g_J is 1 } of x[i + j]              for every j.
// The state J words on is x[J .. J + 623]: generate X = x[0 .. 19936 + 623] once from the current state (33 block updates, 5 us) and
// XOR ~10 k shifted windows of it (0.1 ms with 512-bit registers; X is 82 KB and stays in L2).  The low 31 bits of x[0] never enter the
// recurrence; they only reach the low 31 bits of the new word 0, which are not read either.
//   phi        Berlekamp-Massey over 2 x 19937 bits of the sequence, once per process (~40 ms);
//   g_J        square-and-multiply modulo phi, once per distinct J (a few ms), then g_{(p+1) L} = g_{p L} g_L for a fill's pieces.
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace mtjump {

constexpr int kN = 624, kM = 397, kDeg = 19937;
constexpr int kPW = (2 * kDeg + 64 + 63) / 64;         // 64-bit words of a product before reduction
constexpr int kXBlocks = 33;                          // blocks of X: 33 * 624 = 20592 >= 19937 + 623 + 1
constexpr int kXWords = kXBlocks * kN;

typedef std::vector<uint64_t> Poly;                   // bit i of word i / 64 = coefficient of t^i

inline void block_update(uint32_t* st) {              // the reference recurrence (MT19937RNGEngine.h next_state), one whole block in place
    for (int i = 0; i < kN - kM; ++i) {
        const uint32_t y = (st[i] & 0x80000000u) | (st[i + 1] & 0x7fffffffu);
        st[i] = st[i + kM] ^ (y >> 1) ^ ((0u - (st[i + 1] & 1u)) & 0x9908b0dfu);
    }
    for (int i = kN - kM; i < kN - 1; ++i) {
        const uint32_t y = (st[i] & 0x80000000u) | (st[i + 1] & 0x7fffffffu);
        st[i] = st[i + kM - kN] ^ (y >> 1) ^ ((0u - (st[i + 1] & 1u)) & 0x9908b0dfu);
    }
    const uint32_t y = (st[kN - 1] & 0x80000000u) | (st[0] & 0x7fffffffu);
    st[kN - 1] = st[kM - 1] ^ (y >> 1) ^ ((0u - (st[0] & 1u)) & 0x9908b0dfu);
}

inline bool pbit(const Poly& p, int i) { return (p[(size_t)i >> 6] >> (i & 63)) & 1ull; }
inline void pflip(Poly& p, int i) { p[(size_t)i >> 6] ^= 1ull << (i & 63); }
inline int pdeg(const Poly& p) {
    for (int w = (int)p.size() - 1; w >= 0; --w)
        if (p[(size_t)w]) return 64 * w + 63 - __builtin_clzll(p[(size_t)w]);
    return -1;
}
// r ^= a << s (bits), r long enough
inline void xor_shifted(Poly& r, const Poly& a, int s, int awords) {
    const int ws = s >> 6, bs = s & 63;
    if (bs == 0) {
        for (int i = 0; i < awords; ++i) r[(size_t)(i + ws)] ^= a[(size_t)i];
    } else {
        uint64_t carry = 0;
        for (int i = 0; i < awords; ++i) {
            const uint64_t v = a[(size_t)i];
            r[(size_t)(i + ws)] ^= (v << bs) | carry;
            carry = v >> (64 - bs);
        }
        r[(size_t)(awords + ws)] ^= carry;
    }
}

// characteristic polynomial of the mt19937 state recurrence: Berlekamp-Massey on the top bit of 2 * 19937 consecutive state words
inline Poly compute_phi() {
    const int N = 2 * kDeg;
    std::vector<uint8_t> s((size_t)N);
    {
        uint32_t st[kN];
        st[0] = 5489u;                                                     // any non-degenerate state: init_genrand(5489)
        for (int j = 1; j < kN; ++j) st[j] = 1812433253u * (st[j - 1] ^ (st[j - 1] >> 30)) + (uint32_t)j;
        int n = 0;
        while (n < N) {
            block_update(st);
            for (int i = 0; i < kN && n < N; ++i) s[(size_t)n++] = (uint8_t)(st[i] >> 31);
        }
    }
    const int W = (kDeg + 2 + 63) / 64 + 2;
    Poly C((size_t)W, 0), B((size_t)W, 0), T((size_t)W, 0), win((size_t)W, 0);        // win bit i = s[n - i]
    C[0] = B[0] = 1;
    int L = 0, m = 1;
    for (int n = 0; n < N; ++n) {
        // win <<= 1, insert s[n]
        uint64_t carry = s[(size_t)n];
        for (int w = 0; w < W; ++w) { const uint64_t v = win[(size_t)w]; win[(size_t)w] = (v << 1) | carry; carry = v >> 63; }
        uint64_t acc = 0;
        const int lw = (L >> 6) + 1;
        for (int w = 0; w < lw && w < W; ++w) acc ^= C[(size_t)w] & win[(size_t)w];
        const int d = __builtin_parityll(acc);
        if (!d) { ++m; continue; }
        const int aw = W - (m >> 6) - 1;                                   // B << m stays inside W words (deg B + m <= 19937)
        if (2 * L <= n) {
            T = C;
            if (aw > 0) xor_shifted(C, B, m, aw);
            L = n + 1 - L; B = T; m = 1;
        } else {
            if (aw > 0) xor_shifted(C, B, m, aw);
            ++m;
        }
    }
    // phi(t) = t^L C(1 / t): coefficient i of phi = coefficient L - i of C
    Poly phi((size_t)((kDeg + 64) / 64 + 1), 0);
    if (L != kDeg) return Poly();                                          // cannot happen for mt19937; the caller then keeps the scout
    for (int i = 0; i <= L; ++i) if (pbit(C, L - i)) pflip(phi, i);
    return phi;
}

struct Field {
    Poly phi;                                          // degree kDeg
    int phiw;                                          // words of phi
    bool ok;
    Field() : phi(compute_phi()), phiw(0), ok(false) {
        if (phi.empty()) return;
        phiw = (kDeg + 64) / 64;
        ok = pdeg(phi) == kDeg;
    }
    // p (degree < 2 * kDeg, kPW words) mod phi, in place; result in the low words
    void reduce(Poly& p) const {
        for (int k = pdeg(p); k >= kDeg; --k)
            if (pbit(p, k)) xor_shifted(p, phi, k - kDeg, phiw);
    }
    Poly mul(const Poly& a, const Poly& b) const {     // a b mod phi; a, b of degree < kDeg
        Poly r((size_t)kPW + 2, 0);
        const int aw = (kDeg + 63) / 64;
        for (int w = 0; w < aw; ++w) {
            uint64_t v = a[(size_t)w];
            while (v) {
                const int bit = __builtin_ctzll(v);
                v &= v - 1;
                xor_shifted(r, b, 64 * w + bit, aw);
            }
        }
        reduce(r);
        r.resize((size_t)aw + 1);
        return r;
    }
    Poly sqr(const Poly& a) const {
        Poly r((size_t)kPW + 2, 0);
        const int aw = (kDeg + 63) / 64;
        for (int w = 0; w < aw; ++w) {                  // spread the bits: bit i -> bit 2 i
            uint64_t lo = a[(size_t)w] & 0xffffffffull, hi = a[(size_t)w] >> 32;
            auto spread = [](uint64_t x) {
                x = (x | (x << 16)) & 0x0000ffff0000ffffull;
                x = (x | (x << 8)) & 0x00ff00ff00ff00ffull;
                x = (x | (x << 4)) & 0x0f0f0f0f0f0f0f0full;
                x = (x | (x << 2)) & 0x3333333333333333ull;
                x = (x | (x << 1)) & 0x5555555555555555ull;
                return x;
            };
            r[(size_t)(2 * w)] = spread(lo);
            r[(size_t)(2 * w + 1)] = spread(hi);
        }
        reduce(r);
        r.resize((size_t)aw + 1);
        return r;
    }
    Poly xpow(unsigned long long J) const {            // t^J mod phi
        const int aw = (kDeg + 63) / 64;
        Poly r((size_t)aw + 1, 0), base((size_t)aw + 1, 0);
        r[0] = 1; base[0] = 2;                         // 1, t
        while (J) {
            if (J & 1ull) r = mul(r, base);
            J >>= 1;
            if (J) base = sqr(base);
        }
        return r;
    }
};

inline const Field& field() {
    static const Field f;
    return f;
}

// support of a jump polynomial: the exponents whose coefficient is 1, ascending
typedef std::vector<uint16_t> Support;
inline Support support_of(const Poly& g) {
    Support s;
    for (int i = 0; i < kDeg; ++i) if (pbit(g, i)) s.push_back((uint16_t)i);
    return s;
}

// the supports of g_{p L}, p = 1 .. count, for a piece length of L words; cached per L (a fill size fixes it), extended on demand
struct JumpSet { std::vector<std::shared_ptr<const Support>> sup; std::vector<Poly> g; };
inline std::shared_ptr<const Support> jump_support(unsigned long long L, int p) {
    static std::mutex mu;
    static std::map<unsigned long long, JumpSet> cache;
    const Field& f = field();
    if (!f.ok || p < 1) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (cache.size() > 64) cache.clear();                                  // fill sizes are few; a caller sweeping sizes does not grow this forever
    JumpSet& js = cache[L];
    while ((int)js.g.size() < p) {
        js.g.push_back(js.g.empty() ? f.xpow(L) : f.mul(js.g.back(), js.g.front()));
        js.sup.push_back(std::make_shared<const Support>(support_of(js.g.back())));
    }
    return js.sup[(size_t)p - 1];
}

// X = x[0 .. kXWords): the state `st` (= x[0 .. 623]) followed by the next 32 blocks of the untempered sequence
inline void expand(const uint32_t* st, uint32_t* X) {
    memcpy(X, st, kN * sizeof(uint32_t));
    for (int b = 1; b < kXBlocks; ++b) {
        memcpy(X + (size_t)b * kN, X + (size_t)(b - 1) * kN, kN * sizeof(uint32_t));
        block_update(X + (size_t)b * kN);
    }
}

// out[j] = XOR over i in sup of X[i + j], j = 0 .. 623: three column blocks of 208 words, so the accumulators of a block are 13 vector
// registers of 512 bits and X is read through unaligned loads
#define LS_MTJUMP_APPLY_BODY                                                                        \
    for (int c0 = 0; c0 < kN; c0 += 208) {                                                          \
        uint32_t acc[208];                                                                          \
        for (int j = 0; j < 208; ++j) acc[j] = 0;                                                   \
        for (size_t k = 0; k < n; ++k) {                                                            \
            const uint32_t* src = X + sup[k] + c0;                                                  \
            for (int j = 0; j < 208; ++j) acc[j] ^= src[j];                                         \
        }                                                                                           \
        for (int j = 0; j < 208; ++j) out[c0 + j] = acc[j];                                         \
    }
inline void apply_base(const uint32_t* X, const uint16_t* sup, size_t n, uint32_t* out) { LS_MTJUMP_APPLY_BODY }
#if defined(__x86_64__)
__attribute__((target("avx2"))) inline void apply_avx2(const uint32_t* X, const uint16_t* sup, size_t n, uint32_t* out) { LS_MTJUMP_APPLY_BODY }
__attribute__((target("avx512f"), min_vector_width(512))) inline void apply_avx512(const uint32_t* X, const uint16_t* sup, size_t n, uint32_t* out) { LS_MTJUMP_APPLY_BODY }
#else
inline void apply_avx2(const uint32_t* X, const uint16_t* sup, size_t n, uint32_t* out) { apply_base(X, sup, n, out); }
inline void apply_avx512(const uint32_t* X, const uint16_t* sup, size_t n, uint32_t* out) { apply_base(X, sup, n, out); }
#endif
inline void apply(const uint32_t* X, const Support& sup, uint32_t* out, int isa) {
    if (isa == 2) apply_avx512(X, sup.data(), sup.size(), out);
    else if (isa == 1) apply_avx2(X, sup.data(), sup.size(), out);
    else apply_base(X, sup.data(), sup.size(), out);
}

}  // namespace mtjump
