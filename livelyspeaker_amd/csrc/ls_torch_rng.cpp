// torch's CPU normal stream, restated natively: the host draws of the "identical seeds" mode (noise_source = 'torch_cpu') without
// torch's generator on the critical path.
//
// Contract being reproduced: the reference draws every normal of a sampling loop from torch's global CPU generator
// (scripts/diffusion/gaussian_diffusion.py:700-743: x_T, then per step randn_like(x); scripts/model/RAG.py:10-13, 120: randn_like of the
// style token in both CFG passes), so `torch.manual_seed(s)` fixes the sample.  torch (2.x, ATen/native/cpu/DistributionTemplates.h,
// ATen/core/DistributionsHelper.h, ATen/core/MT19937RNGEngine.h) makes those draws two ways:
//   * contiguous float tensor of >= 16 elements (randn(B,1,512), randn(*shape)): n 24-bit uniforms (one mt19937 word each), then
//     Box-Muller on 16-blocks -- element j pairs with j + 8: r = sqrt(-2 log(1 - u[j])), theta = 2 pi u[j+8] -- in FLOAT arithmetic
//     (std::log / cos / sin of float in torch's DEFAULT kernel; Cephes polynomials with compiler-contracted FMAs in its AVX2 / AVX512
//     kernels -- both restated below, the Python side picks the variant that reproduces torch on this machine, or keeps torch's generator);
//     a tail that is not a multiple of 16 re-draws the LAST 16 elements;
//   * anything else (randn_like of the model-output-shaped view whose memory order is [T][B][J][F]): one element at a time in MEMORY
//     order through normal_distribution<double>: two 53-bit uniforms (two mt19937 words each, first word = high half),
//     r = sqrt(-2 log1p(-u2)), theta = 2 pi u1, the cos branch is returned and the sin branch cached in the generator for the next
//     element (also across calls).
// The generator state travels as torch.get_rng_state()'s 5056-byte blob (layout probed in tests/test_torch_rng.py), updated in place,
// so torch.set_rng_state() leaves torch's generator exactly where the reference's draws would have left it.
//
// Speed: the mt19937 words are produced sequentially (they must be) but in bulk -- the state update and the tempering are plain array
// loops the compiler vectorises (an AVX2 clone is picked at run time) -- by the calling thread, which hands every job's transcendental
// part to a pool of worker threads and goes on generating the next job's words meanwhile.
#include "ls_hip.h"

// every product and sum below rounds where it is written: no contraction by THIS compiler (the FMAs of the restated kernels are explicit)
#pragma clang fp contract(off)

#include <array>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cstdio>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include "ls_mt_jump.h"
#if defined(__x86_64__)
#include <immintrin.h>
#define LS_CPU_PAUSE _mm_pause();
#else
#define LS_CPU_PAUSE
#endif

namespace {

constexpr int kN = 624, kM = 397;
constexpr size_t kStateBytes = 5056;
constexpr size_t kOffLeft = 8, kOffSeeded = 12, kOffNext = 16, kOffState = 24, kOffNormalY = 5024, kOffNormalValid = 5040;

struct Mt {
    uint32_t st[kN];
    int left;
    uint32_t next;
    double cached;
    int cached_valid;

    bool load(const uint8_t* blob) {
        int32_t l, seeded;
        uint64_t nx;
        memcpy(&l, blob + kOffLeft, 4);
        memcpy(&seeded, blob + kOffSeeded, 4);
        memcpy(&nx, blob + kOffNext, 8);
        if (!seeded || l <= 0 || l > kN || nx > (uint64_t)kN) return false;
        left = l; next = (uint32_t)nx;
        for (int i = 0; i < kN; ++i) { uint64_t v; memcpy(&v, blob + kOffState + 8 * (size_t)i, 8); st[i] = (uint32_t)v; }
        memcpy(&cached, blob + kOffNormalY, 8);
        int32_t cv;
        memcpy(&cv, blob + kOffNormalValid, 4);
        cached_valid = cv;
        return true;
    }
    void store(uint8_t* blob) const {
        int32_t l = left;
        uint64_t nx = next;
        memcpy(blob + kOffLeft, &l, 4);
        memcpy(blob + kOffNext, &nx, 8);
        for (int i = 0; i < kN; ++i) { uint64_t v = st[i]; memcpy(blob + kOffState + 8 * (size_t)i, &v, 8); }
        const double c = cached_valid ? cached : 0.0;
        memcpy(blob + kOffNormalY, &c, 8);
        int32_t cv = cached_valid ? 1 : 0;
        memcpy(blob + kOffNormalValid, &cv, 4);
    }
    static uint32_t twist(uint32_t u, uint32_t v) { return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u); }
    void next_state() {
        uint32_t* p = st;
        left = kN; next = 0;
        for (int j = kN - kM + 1; --j; p++) *p = p[kM] ^ twist(p[0], p[1]);
        for (int j = kM; --j; p++) *p = p[kM - kN] ^ twist(p[0], p[1]);
        *p = p[kM - kN] ^ twist(p[0], st[0]);
    }
    uint32_t word() {
        if (--left == 0) next_state();
        uint32_t y = st[next++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    // n words of the same stream in bulk.  `left - 1` words of the current block are still unread, at st[next ...]
    // Returns null when every word is in `out` on return, or the count of generator pieces still being written (see `gen`): whoever reads
    // `out` waits for it to reach zero (words_ready) -- the calling thread itself goes on to the next draw's state
    using Pending = std::shared_ptr<std::atomic<int>>;
    Pending fill_words(uint32_t* out, size_t n);
    // Generator threads for long fills (null: none).  The stream is sequential, but the 624-word state recurrence alone runs ~4x faster than
    // recurrence + tempering + the store of the words: on a long fill the calling thread only advances the STATE ("scout"), handing a
    // snapshot of it to a generator thread every few hundred blocks, and the generators produce the words of their pieces in parallel.
    class Pool* gen = nullptr;
    float uf() { return (float)(word() & ((1u << 24) - 1)) * (1.0f / (float)(1u << 24)); }                 // uniform_real<float>
    double ud() {                                                                                           // uniform_real<double>: random64
        const uint64_t hi = word(), lo = word();
        return (double)(((hi << 32) | lo) & ((1ull << 53) - 1)) * (1.0 / (double)(1ull << 53));
    }
};

// The block update and the tempering as array loops (same recurrence as next_state / word): in the first loop every read is ahead of
// the write, in the second the value read was written 227 iterations earlier, so both vectorise.
#define LS_MT_BLOCK_BODY                                                                                                    \
    for (int i = 0; i < kN - kM; ++i) {                                                                                     \
        const uint32_t y = (st[i] & 0x80000000u) | (st[i + 1] & 0x7fffffffu);                                               \
        st[i] = st[i + kM] ^ (y >> 1) ^ ((0u - (st[i + 1] & 1u)) & 0x9908b0dfu);                                            \
    }                                                                                                                       \
    for (int i = kN - kM; i < kN - 1; ++i) {                                                                                \
        const uint32_t y = (st[i] & 0x80000000u) | (st[i + 1] & 0x7fffffffu);                                               \
        st[i] = st[i + kM - kN] ^ (y >> 1) ^ ((0u - (st[i + 1] & 1u)) & 0x9908b0dfu);                                       \
    }                                                                                                                       \
    {                                                                                                                       \
        const uint32_t y = (st[kN - 1] & 0x80000000u) | (st[0] & 0x7fffffffu);                                              \
        st[kN - 1] = st[kM - 1] ^ (y >> 1) ^ ((0u - (st[0] & 1u)) & 0x9908b0dfu);                                           \
    }
#define LS_MT_TEMPER_BODY                                                                                                   \
    for (size_t i = 0; i < m; ++i) {                                                                                        \
        uint32_t y = src[i];                                                                                                \
        y ^= (y >> 11);                                                                                                     \
        y ^= (y << 7) & 0x9d2c5680u;                                                                                        \
        y ^= (y << 15) & 0xefc60000u;                                                                                       \
        y ^= (y >> 18);                                                                                                     \
        out[i] = y;                                                                                                         \
    }
void mt_block_base(uint32_t* __restrict__ st) { LS_MT_BLOCK_BODY }
void mt_temper_base(const uint32_t* __restrict__ src, uint32_t* __restrict__ out, size_t m) { LS_MT_TEMPER_BODY }
// ISA clones and run-time dispatch are x86 features: elsewhere the engine library still builds, the word loops run in their base form
// and the FMA-contracted float transforms (variants 1..4) report LS_EUNSUPPORTED (the Python side then keeps torch's own draws)
#if defined(__x86_64__)
#define LS_TRNG_X86 1
__attribute__((target("avx2"))) void mt_block_avx2(uint32_t* __restrict__ st) { LS_MT_BLOCK_BODY }
__attribute__((target("avx2"))) void mt_temper_avx2(const uint32_t* __restrict__ src, uint32_t* __restrict__ out, size_t m) { LS_MT_TEMPER_BODY }
// The AVX-512 block update is written out: 624 words = 39 vectors exactly.  Left to the vectoriser the loop whose operand was written
// 227 words earlier came out 128 bits wide with a 35-word scalar remainder in front of it (0.9 cycles per word on Zen 5); here the
// new vectors stay in registers and that operand is cut out of two of them (valignd by 13 = 397 - 24 * 16), so no load ever waits
// for a store of the same pass.  This update is the sequential part of the whole mode: the scout of Mt::fill_words runs nothing else.
__attribute__((target("avx512f"))) void mt_block_avx512(uint32_t* __restrict__ st) {
    const __m512i upper = _mm512_set1_epi32((int)0x80000000u), one = _mm512_set1_epi32(1), matrix = _mm512_set1_epi32((int)0x9908b0dfu);
    __m512i nv[40];                                    // nv[v + 1] = new words 16 v ... 16 v + 15; nv[0] = OLD words 608 ... 623
    nv[0] = _mm512_loadu_si512(st + 608);
#pragma unroll
    for (int v = 0; v < 39; ++v) {
        const __m512i a = _mm512_loadu_si512(st + 16 * v);
        __m512i b;
        if (v < 38) b = _mm512_loadu_si512(st + 16 * v + 1);
        else b = _mm512_mask_broadcastd_epi32(_mm512_maskz_loadu_epi32(0x7fff, st + 16 * v + 1), 0x8000, _mm512_castsi512_si128(nv[1]));   // word 624 = new word 0
        // word i + 397 (mod 624): old words for i < 227; for i >= 227 the new word i - 227
        const __m512i c = v < 14 ? _mm512_loadu_si512(st + 16 * v + kM) : _mm512_alignr_epi32(nv[v - 13], nv[v - 14], 13);
        const __m512i y = _mm512_ternarylogic_epi32(upper, a, b, 0xCA);                 // (a & upper) | (b & ~upper)
        __m512i r = _mm512_xor_si512(c, _mm512_srli_epi32(y, 1));
        r = _mm512_mask_xor_epi32(r, _mm512_test_epi32_mask(b, one), r, matrix);
        _mm512_storeu_si512(st + 16 * v, r);
        nv[v + 1] = r;
    }
}
__attribute__((target("avx512f"), min_vector_width(512))) void mt_temper_avx512(const uint32_t* __restrict__ src, uint32_t* __restrict__ out, size_t m) { LS_MT_TEMPER_BODY }
inline int mt_isa() { static const int isa = __builtin_cpu_supports("avx512f") ? 2 : __builtin_cpu_supports("avx2") ? 1 : 0; return isa; }
inline bool have_fma() { return __builtin_cpu_supports("fma"); }
#else
#define LS_TRNG_X86 0
inline void mt_block_avx2(uint32_t* st) { mt_block_base(st); }
inline void mt_temper_avx2(const uint32_t* src, uint32_t* out, size_t m) { mt_temper_base(src, out, m); }
inline void mt_block_avx512(uint32_t* st) { mt_block_base(st); }
inline void mt_temper_avx512(const uint32_t* src, uint32_t* out, size_t m) { mt_temper_base(src, out, m); }
inline int mt_isa() { return 0; }
inline bool have_fma() { return false; }
#endif

void mt_blocks_to_words(uint32_t* st, uint32_t* out, size_t nblocks, int isa) {       // nblocks block updates of `st`, every block's 624 words tempered to out
    for (size_t b = 0; b < nblocks; ++b, out += kN) {
        if (isa == 2) { mt_block_avx512(st); mt_temper_avx512(st, out, kN); }
        else if (isa == 1) { mt_block_avx2(st); mt_temper_avx2(st, out, kN); }
        else { mt_block_base(st); mt_temper_base(st, out, kN); }
    }
}
void pool_submit(class Pool* p, std::function<void()> job);
void pool_wait(class Pool* p);
int pool_threads(class Pool* p);
constexpr size_t kParBlocks = 256;           // whole blocks from which a fill is dealt to the generator threads (160 K words)

#ifdef LS_TRNG_TIMING
#include <chrono>
double g_t_scout = 0, g_t_genwait = 0, g_t_serial = 0, g_t_ringwait = 0, g_t_submit = 0, g_t_finalwait = 0;
inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define LS_T0 const double t0__ = now_s();
#define LS_T1(acc) acc += now_s() - t0__;
std::atomic<uint64_t> g_ns_gen{0}, g_ns_spin{0}, g_ns_xform{0};
#define LS_TA(acc) acc.fetch_add((uint64_t)((now_s() - t0__) * 1e9));
#else
#define LS_T0
#define LS_T1(acc)
#define LS_TA(acc)
#endif
inline void words_ready(const Mt::Pending& p) {
    if (!p) return;
    LS_T0
    for (int spins = 0; p->load(std::memory_order_acquire) != 0; ++spins) {
        if (spins < 64) { LS_CPU_PAUSE } else std::this_thread::yield();
    }
    LS_TA(g_ns_spin)
}
// ls_trng_set_jump: generator threads of a long fill start from JUMPED states (ls_mt_jump.h) instead of behind the sequential scout.
// Off by default -- measured on the GPU hosts (round 6, tools/rng_jump_ab.py, tools/seeds_time.py): the scout costs 0.2 ms per step at
// BEAT B = 256 there, not the 1.35 ms round 5 attributed to it; what a step's draws cost is the double-precision transforms (15.7 thread-ms)
// and, per call, starting and draining the pipeline, so the jump moves nothing in the loop (0.60-1.4 ms per step against 0.68-0.90, run to
// run) while persistent pools and 12-step segments brought the mode to the step kernel's own rate.  Kept: bitwise-tested, and the right
// tool on a host whose single-thread recurrence is the bound.
std::atomic<int> g_use_jump{0};
Mt::Pending Mt::fill_words(uint32_t* out, size_t n) {
    const int isa = mt_isa();
    const size_t n_entry = n;
    Pending pend;
    if (gen && n >= (kParBlocks + 2) * (size_t)kN) {
        // head: the rest of the current block
        if (left - 1 > 0) {
            const size_t m = (size_t)(left - 1);
            if (isa == 2) mt_temper_avx512(st + next, out, m); else if (isa == 1) mt_temper_avx2(st + next, out, m); else mt_temper_base(st + next, out, m);
            next += (uint32_t)m; left -= (int)m; out += m; n -= m;
        }
        // middle: whole blocks, in pieces.  A piece's generator starts from the state in front of it.
        const size_t nb = n / kN;
        const int nt = pool_threads(gen) > 0 ? pool_threads(gen) : 1;
        // (a) jump-ahead (ls_mt_jump.h): one piece per generator thread, each thread computes its own starting state from a shared
        // 33-block expansion of the current state and a precomputed jump polynomial (~0.1 ms), then produces its words; this thread only
        // jumps to the last whole piece and walks the remainder (< one piece) to leave the state where the fill ends.  The piece length
        // depends on the fill's size alone (not on where the stream stands inside a block), so a loop's fills reuse their polynomials.
        size_t jper = (n_entry / kN + (size_t)nt - 1) / (size_t)nt;
        if (jper < 64) jper = 64;
        bool jumped = false;
        if (g_use_jump.load(std::memory_order_relaxed) && nb >= 2 * jper && mtjump::field().ok) {
            size_t pfull = nb / jper, rem = nb % jper;
            if (rem == 0) { --pfull; rem = jper; }      // the state left behind comes out of real block updates (word 0's low bits are not part of a jumped state)
            const unsigned long long L = (unsigned long long)jper * kN;
            std::vector<std::shared_ptr<const mtjump::Support>> sup(pfull + 1);
            bool have = true;
            for (size_t p = 1; p <= pfull; ++p) have = have && (sup[p] = mtjump::jump_support(L, (int)p)) != nullptr;
            if (have) {
                LS_T0
                auto X = std::make_shared<std::vector<uint32_t>>((size_t)mtjump::kXWords);
                mtjump::expand(st, X->data());
                pend = std::make_shared<std::atomic<int>>((int)(pfull + 1));
                for (size_t p = 0; p < pfull; ++p) {
                    uint32_t* dst = out + p * jper * kN;
                    auto sp = sup[p];
                    const size_t cnt = jper;
                    pool_submit(gen, [X, sp, dst, cnt, isa, pend] {
                        LS_T0
                        alignas(64) uint32_t s0[kN];
                        if (sp) mtjump::apply(X->data(), *sp, s0, isa); else memcpy(s0, X->data(), sizeof s0);
                        mt_blocks_to_words(s0, dst, cnt, isa);
                        pend->fetch_sub(1, std::memory_order_acq_rel);
                        LS_TA(g_ns_gen)
                    });
                }
                alignas(64) uint32_t s1[kN];
                mtjump::apply(X->data(), *sup[pfull], s1, isa);
                {
                    auto snap = std::make_shared<std::array<uint32_t, kN>>();
                    memcpy(snap->data(), s1, sizeof s1);
                    uint32_t* dst = out + pfull * jper * kN;
                    const size_t cnt = rem;
                    pool_submit(gen, [snap, dst, cnt, isa, pend] { LS_T0 mt_blocks_to_words(snap->data(), dst, cnt, isa); pend->fetch_sub(1, std::memory_order_acq_rel); LS_TA(g_ns_gen) });
                }
                for (size_t b = 0; b < rem; ++b) { if (isa == 2) mt_block_avx512(s1); else if (isa == 1) mt_block_avx2(s1); else mt_block_base(s1); }
                memcpy(st, s1, sizeof s1);
                jumped = true;
                LS_T1(g_t_scout)
            }
        }
        // (b) the scout (this thread) runs the recurrence alone over each piece to reach the next snapshot
        if (!jumped) {
        size_t per = (nb + (size_t)(4 * nt) - 1) / (size_t)(4 * nt);
        if (per < 64) per = 64;
        pend = std::make_shared<std::atomic<int>>((int)((nb + per - 1) / per));
        { LS_T0
        for (size_t b0 = 0; b0 < nb; b0 += per) {
            const size_t cnt = b0 + per < nb ? per : nb - b0;
            auto snap = std::make_shared<std::array<uint32_t, kN>>();
            memcpy(snap->data(), st, sizeof st);
            uint32_t* dst = out + b0 * kN;
            pool_submit(gen, [snap, dst, cnt, isa, pend] { LS_T0 mt_blocks_to_words(snap->data(), dst, cnt, isa); pend->fetch_sub(1, std::memory_order_acq_rel); LS_TA(g_ns_gen) });
            for (size_t b = 0; b < cnt; ++b) { if (isa == 2) mt_block_avx512(st); else if (isa == 1) mt_block_avx2(st); else mt_block_base(st); }
        }
        LS_T1(g_t_scout) }
        }
        out += nb * kN; n -= nb * kN;
        left = 1; next = kN;                 // the last block is used up, exactly as after reading it word by word
    }
    LS_T0
    while (n > 0) {
        if (left - 1 == 0) {                 // word(): --left == 0 -> next_state(), and the word it then reads leaves left at kN
            if (isa == 2) mt_block_avx512(st); else if (isa == 1) mt_block_avx2(st); else mt_block_base(st);
            left = kN + 1; next = 0;
        }
        const size_t m = n < (size_t)(left - 1) ? n : (size_t)(left - 1);
        if (isa == 2) mt_temper_avx512(st + next, out, m); else if (isa == 1) mt_temper_avx2(st + next, out, m); else mt_temper_base(st + next, out, m);
        next += (uint32_t)m; left -= (int)m; out += m; n -= m;
    }
    LS_T1(g_t_serial)
    return pend;
}

// ---- the float transform of the contiguous path -------------------------------------------------------------------------------
// variant 0: normal_fill_16<float> as written (std::log / std::cos / std::sin of float): torch's DEFAULT-capability kernel.
// variants 1..4: normal_fill_16_AVX2, which torch's AVX2 AND AVX512 kernels use: Cephes' single-precision log and sincos in Julien
// Pommier's formulation (public: sse_mathfun / avx_mathfun, zlib licence), restated per lane here with the multiply-adds contracted to
// FMAs the way torch's compiler (GCC, -ffp-contract=fast) did.  Where a sum has two products to choose from the contraction is the
// compiler's pick, hence the variants: bit 0 = the log's  p(x) x z + e q1, bit 1 = the cosine's  y z - 0.5 z  (set: the second product
// is the fused one).  The Python side finds the variant that reproduces torch bit for bit on this machine (or none).
inline float as_f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline uint32_t as_u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

#if LS_TRNG_X86
#define LS_FMA_TARGET __attribute__((target("fma")))      // fmaf as one instruction (through libm it is a call per operation)
#else
#define LS_FMA_TARGET
#endif
// (always inlined: they take the ISA of the loop they are in -- the per-block form below, or the array loops of cephes_blocks_*)
__attribute__((always_inline)) inline float cephes_logf(float x, int variant) {
    const bool invalid = x <= 0.0f;
    x = x < as_f(0x00800000u) ? as_f(0x00800000u) : x;                   // cut off denormalized stuff
    int e_i = (int)(as_u(x) >> 23) - 0x7f;
    x = as_f((as_u(x) & ~0x7f800000u) | as_u(0.5f));         // keep only the fractional part
    float e = (float)e_i + 1.0f;
    const bool lt = x < 0.707106781186547524f;
    const float tmp0 = lt ? x : 0.0f;
    x = x - 1.0f;
    e = e - (lt ? 1.0f : 0.0f);
    x = x + tmp0;
    const float z = x * x;
    float y = 7.0376836292E-2f;
    y = __builtin_fmaf(y, x, -1.1514610310E-1f);
    y = __builtin_fmaf(y, x, 1.1676998740E-1f);
    y = __builtin_fmaf(y, x, -1.2420140846E-1f);
    y = __builtin_fmaf(y, x, 1.4249322787E-1f);
    y = __builtin_fmaf(y, x, -1.6668057665E-1f);
    y = __builtin_fmaf(y, x, 2.0000714765E-1f);
    y = __builtin_fmaf(y, x, -2.4999993993E-1f);
    y = __builtin_fmaf(y, x, 3.3333331174E-1f);
    y = y * x;
    if (variant & 1) y = __builtin_fmaf(e, -2.12194440e-4f, y * z);
    else y = __builtin_fmaf(y, z, e * -2.12194440e-4f);
    y = __builtin_fmaf(-z, 0.5f, y);
    x = __builtin_fmaf(e, 0.693359375f, x + y);
    return invalid ? as_f(0xffffffffu) : x;
}

__attribute__((always_inline)) inline void cephes_sincosf(float xin, int variant, float& s, float& c) {
    uint32_t sign_sin = as_u(xin) & 0x80000000u;
    float x = as_f(as_u(xin) & 0x7fffffffu);
    float y = x * 1.27323954473516f;                         // 4 / pi
    int j = (int)y;                                          // cvttps
    j = (j + 1) & ~1;
    y = (float)j;
    const uint32_t swap_sign_sin = ((uint32_t)(j & 4)) << 29;
    const bool poly = (j & 2) == 0;
    x = __builtin_fmaf(y, -0.78515625f, x);
    x = __builtin_fmaf(y, -2.4187564849853515625e-4f, x);
    x = __builtin_fmaf(y, -3.77489497744594108e-8f, x);
    const uint32_t sign_cos = ((uint32_t)(~(j - 2) & 4)) << 29;
    sign_sin ^= swap_sign_sin;
    const float z = x * x;
    float yc = 2.443315711809948E-005f;
    yc = __builtin_fmaf(yc, z, -1.388731625493765E-003f);
    yc = __builtin_fmaf(yc, z, 4.166664568298827E-002f);
    yc = yc * z;
    if (variant & 2) yc = __builtin_fmaf(-z, 0.5f, yc * z);
    else yc = __builtin_fmaf(yc, z, -(z * 0.5f));
    yc = yc + 1.0f;
    float ys = -1.9515295891E-4f;
    ys = __builtin_fmaf(ys, z, 8.3321608736E-3f);
    ys = __builtin_fmaf(ys, z, -1.6666654611E-1f);
    ys = ys * z;
    ys = __builtin_fmaf(ys, x, x);
    const float rs = poly ? ys : yc, rc = poly ? yc : ys;
    s = as_f(as_u(rs) ^ sign_sin);
    c = as_f(as_u(rc) ^ sign_cos);
}

inline float word_to_uf(uint32_t w) { return (float)(w & ((1u << 24) - 1)) * (1.0f / (float)(1u << 24)); }                 // uniform_real<float>

LS_FMA_TARGET void fill16_cephes(float* d, int variant) {
    const float two_pi = 2.0f * 3.14159265358979323846;
    for (int j = 0; j < 8; ++j) {
        const float u1 = 1.0f - d[j];
        const float u2 = d[j + 8];
        const float radius = std::sqrt(-2.0f * cephes_logf(u1, variant - 1));
        const float theta = two_pi * u2;
        float sn, cs;
        cephes_sincosf(theta, variant - 1, sn, cs);
        d[j] = __builtin_fmaf(radius * cs, 1.0f, 0.0f);
        d[j + 8] = __builtin_fmaf(radius * sn, 1.0f, 0.0f);
    }
}

// nb 16-blocks of mt19937 WORDS at `base` -> their normals, in place: the same arithmetic as fill16_cephes element for element (every
// operation is written out, contraction is off), laid out so that the eight pairs of a block are the lanes of a vector
#define LS_CEPHES_BLOCKS_BODY                                                                                                \
    const float two_pi = 2.0f * 3.14159265358979323846;                                                                      \
    const int v = variant - 1;                                                                                               \
    for (size_t b = 0; b < nb; ++b) {                                                                                        \
        uint32_t wv[16];                                                                                                     \
        float o[16];                                                                                                         \
        memcpy(wv, base + 16 * b, sizeof wv);                                                                                \
        for (int j = 0; j < 8; ++j) {                                                                                        \
            const float u1 = 1.0f - word_to_uf(wv[j]);                                                                       \
            const float u2 = word_to_uf(wv[j + 8]);                                                                          \
            const float radius = __builtin_elementwise_sqrt(-2.0f * cephes_logf(u1, v));                                     \
            const float theta = two_pi * u2;                                                                                 \
            float sn, cs;                                                                                                    \
            cephes_sincosf(theta, v, sn, cs);                                                                                \
            o[j] = __builtin_fmaf(radius * cs, 1.0f, 0.0f);                                                                  \
            o[j + 8] = __builtin_fmaf(radius * sn, 1.0f, 0.0f);                                                              \
        }                                                                                                                    \
        memcpy(base + 16 * b, o, sizeof o);                                                                                  \
    }
LS_FMA_TARGET void cephes_blocks_fma(float* base, size_t nb, int variant) { LS_CEPHES_BLOCKS_BODY }
#if LS_TRNG_X86
__attribute__((target("avx2,fma"))) void cephes_blocks_avx2(float* base, size_t nb, int variant) { LS_CEPHES_BLOCKS_BODY }
#else
inline void cephes_blocks_avx2(float* base, size_t nb, int variant) { cephes_blocks_fma(base, nb, variant); }
#endif

// in place on 16 uniforms
void fill16(float* d, int variant) {
    if (variant == 0) {
        for (int j = 0; j < 8; ++j) {
            const float u1 = 1 - d[j];
            const float u2 = d[j + 8];
            const float radius = std::sqrt(-2 * std::log(u1));
            const float theta = 2.0f * 3.14159265358979323846 * u2;
            d[j] = radius * std::cos(theta) * 1.0f + 0.0f;
            d[j + 8] = radius * std::sin(theta) * 1.0f + 0.0f;
        }
        return;
    }
    fill16_cephes(d, variant);
}

// ---- worker pool: the calling thread produces jobs (their words), the workers run the transcendental part ----------------------------
class Pool {
public:
    explicit Pool(int threads) {
        for (int t = 0; t < threads; ++t) th_.emplace_back([this] { work(); });
    }
    ~Pool() {
        wait();
        { std::lock_guard<std::mutex> l(m_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    // fn(a, b) over [0, n) in chunks; runs inline without workers or for small jobs.  One queue entry per call: the workers deal its
    // chunks among themselves through an atomic counter (a lock per chunk made the PRODUCER wait behind its own workers)
    void run(size_t n, size_t chunk, std::function<void(size_t, size_t)> fn) {
        if (th_.empty() || n <= chunk) { if (n) fn(0, n); return; }
        LS_T0
        auto b = std::make_shared<Batch>();
        b->fn = std::move(fn); b->n = n; b->chunk = chunk; b->njobs = (n + chunk - 1) / chunk;
        { std::lock_guard<std::mutex> l(m_); q_.push_back(std::move(b)); ++pending_; }
        cv_.notify_all();
        LS_T1(g_t_submit)
    }
    void wait() {
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [this] { return pending_ == 0; });
    }
    void submit(std::function<void()> job) {            // one job; runs inline without workers
        if (th_.empty()) { job(); return; }
        auto b = std::make_shared<Batch>();
        b->fn = [job = std::move(job)](size_t, size_t) { job(); };
        b->n = 1; b->chunk = 1; b->njobs = 1;
        { std::lock_guard<std::mutex> l(m_); q_.push_back(std::move(b)); ++pending_; }
        cv_.notify_one();
    }
    int threads() const { return (int)th_.size(); }
    bool take_failure() { return failed_.exchange(false); }
private:
    std::atomic<bool> failed_{false};
    struct Batch {
        std::function<void(size_t, size_t)> fn;
        size_t n = 0, chunk = 1, njobs = 0;
        std::atomic<size_t> next{0}, finished{0};
    };
    void work() {
        for (;;) {
            std::shared_ptr<Batch> b;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [this] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;
                b = q_.front();
            }
            for (;;) {
                const size_t i = b->next.fetch_add(1, std::memory_order_relaxed);
                if (i >= b->njobs) break;
                const size_t a = i * b->chunk;
                try {
                    b->fn(a, a + b->chunk < b->n ? a + b->chunk : b->n);
                } catch (...) {                          // (a failed allocation inside a job: recorded, the batch still retires, the entry point reports it)
                    failed_.store(true, std::memory_order_relaxed);
                }
                if (b->finished.fetch_add(1, std::memory_order_acq_rel) + 1 == b->njobs) {
                    std::lock_guard<std::mutex> l(m_);
                    if (--pending_ == 0) done_.notify_all();
                }
            }
            {   // every chunk of b is taken: whoever notices first retires it from the queue
                std::lock_guard<std::mutex> l(m_);
                if (!q_.empty() && q_.front() == b) q_.pop_front();
            }
        }
    }
    std::vector<std::thread> th_;
    std::deque<std::shared_ptr<Batch>> q_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    size_t pending_ = 0;                                  // batches not finished yet
    bool stop_ = false;
};

// The two pools of a draw call, kept between calls: a sampling loop in the identical-seeds mode calls ls_trng_fill_steps once per segment of
// a few steps, and starting ~30 threads per call cost more than the draws of a short segment.  One caller at a time owns them (the stream
// is sequential anyway); a concurrent caller gets pools of its own for the call.
struct PoolPair {
    std::unique_ptr<Pool> own_x, own_g;
    Pool* x = nullptr;
    Pool* g = nullptr;
    std::unique_lock<std::mutex> lock;
};
PoolPair acquire_pools(int nx, int ng) {
    static std::mutex mu;
    static std::unique_ptr<Pool> keep_x, keep_g;
    PoolPair pp;
    pp.lock = std::unique_lock<std::mutex>(mu, std::try_to_lock);
    if (pp.lock.owns_lock()) {
        if (!keep_x || keep_x->threads() != nx) keep_x.reset(new Pool(nx));
        if (!keep_g || keep_g->threads() != ng) keep_g.reset(new Pool(ng));
        pp.x = keep_x.get(); pp.g = keep_g.get();
    } else {
        pp.own_x.reset(new Pool(nx)); pp.own_g.reset(new Pool(ng));
        pp.x = pp.own_x.get(); pp.g = pp.own_g.get();
    }
    return pp;
}

void pool_submit(Pool* p, std::function<void()> job) { p->submit(std::move(job)); }
void pool_wait(Pool* p) { p->wait(); }
int pool_threads(Pool* p) { return p->threads(); }

inline double words_to_ud(uint32_t hi, uint32_t lo) {                                                                        // uniform_real<double>: random64
    return (double)((((uint64_t)hi << 32) | lo) & ((1ull << 53) - 1)) * (1.0 / (double)(1ull << 53));
}
// one pair of normal_distribution<double>: the cos branch is the sample, the sin branch the cached one
inline void normal_pair(const uint32_t* w, double& zc, double& zs) {
    const double u1 = words_to_ud(w[0], w[1]), u2 = words_to_ud(w[2], w[3]);
    const double r = ::sqrt(-2.0 * ::log1p(-u2));
    const double theta = 2.0 * 3.14159265358979323846 * u1;
    zs = r * ::sin(theta);
    zc = r * ::cos(theta);
}

// A contiguous float draw of n >= 16 elements: the words land in `out` itself (one per element) and are turned into uniforms and then
// normals in place, 16 at a time.  A length that is not a multiple of 16 re-draws the LAST 16 elements ("recompute the last 16 values").
void draw_contig(Mt& g, Pool& pool, float* out, size_t n, int variant) {
    uint32_t* w = reinterpret_cast<uint32_t*>(out);
    const Mt::Pending ready = g.fill_words(w, n);
    const size_t blocks = n / 16;
    auto block = [variant](float* d) {
        uint32_t* u = reinterpret_cast<uint32_t*>(d);
        float f[16];
        for (int i = 0; i < 16; ++i) f[i] = word_to_uf(u[i]);
        fill16(f, variant);
        memcpy(d, f, sizeof f);
    };
    if (n % 16) {                        // rare: finish this job before the tail overwrites the end of its last full block
        words_ready(ready);
        pool.wait();
        for (size_t i = 0; i < blocks; ++i) block(out + 16 * i);
        uint32_t tw[16];
        g.fill_words(tw, 16);
        float f[16];
        for (int i = 0; i < 16; ++i) f[i] = word_to_uf(tw[i]);
        fill16(f, variant);
        memcpy(out + n - 16, f, sizeof f);
        return;
    }
    if (variant > 0) {
        const bool avx2 = mt_isa() >= 1;
        pool.run(blocks, 2048, [out, variant, avx2, ready](size_t a, size_t b) {
            words_ready(ready);
            if (avx2) cephes_blocks_avx2(out + 16 * a, b - a, variant); else cephes_blocks_fma(out + 16 * a, b - a, variant);
        });
        return;
    }
    pool.run(blocks, 2048, [out, block, ready](size_t a, size_t b) { words_ready(ready); for (size_t i = a; i < b; ++i) block(out + 16 * i); });
}

// ---- normal_distribution<double> pairs whose samples are stored as FLOATS: a vectorised evaluation with a guard ----------------------
// torch evaluates r = sqrt(-2 log1p(-u2)), r cos(theta), r sin(theta) through libm in double and rounds the sample to float when it
// stores it.  libm's scalar calls are most of this mode's host time (~80 ns per pair), and the float only needs the double to ~2^-25:
// the pairs are evaluated HERE by branch-free array loops the compiler vectorises (fdlibm's published log / kernel_sin / kernel_cos
// polynomials, Cody-Waite reduction by pi/2 in two parts -- exact for the quadrants 0..4 that theta < 2 pi can reach), within
// 2^-48 r of libm's value (tests/test_torch_rng.py measures < 2^-50 r over 10^7 pairs).  A sample is taken from this evaluation only
// if every double within 2^-44 r of it rounds to the SAME float; otherwise (about one sample in 10^5) the pair is re-evaluated by
// normal_pair -- libm, as torch does -- so the stored floats are torch's bit for bit whatever the fast evaluation's last bits are.
constexpr int kFastChunk = 256;
#define LS_FAST_PAIRS_BODY(EXTRA)                                                                                                           \
    _Pragma("clang fp contract(fast)")                                                                                                  \
    for (int i = 0; i < np; ++i) {                                                                                                      \
        const uint64_t a = (((uint64_t)w[4 * i] << 32) | w[4 * i + 1]) & ((1ull << 53) - 1);                                            \
        const uint64_t b = (((uint64_t)w[4 * i + 2] << 32) | w[4 * i + 3]) & ((1ull << 53) - 1);                                        \
        const double u1 = (double)(int64_t)a * (1.0 / (double)(1ull << 53)), u2 = (double)(int64_t)b * (1.0 / (double)(1ull << 53));   \
        /* log(1 - u2): 1 - u2 is exact */                                                                                              \
        const double x = 1.0 - u2;                                                                                                      \
        uint64_t ix; memcpy(&ix, &x, 8);                                                                                                \
        const int64_t e = (int64_t)(ix >> 52) - 1023;                                                                                   \
        const uint64_t mant = ix & 0x000fffffffffffffull;                                                                               \
        const uint64_t up = (mant + 0x00095f6400000000ull) & 0x0010000000000000ull;                                                     \
        const uint64_t mb = mant | (up ^ 0x3ff0000000000000ull);                                                                        \
        double m; memcpy(&m, &mb, 8);                                                                                                   \
        const double dk = (double)(e + (int64_t)(up >> 52));                                                                            \
        const double f = m - 1.0, s = f / (2.0 + f), z = s * s, w2 = z * z;                                                             \
        const double t1 = w2 * (3.999999999940941908e-01 + w2 * (2.222219843214978396e-01 + w2 * 1.531383769920937332e-01));            \
        const double t2 = z * (6.666666666666735130e-01 + w2 * (2.857142874366239149e-01 + w2 * (1.818357216161805012e-01 + w2 * 1.479819860511658591e-01))); \
        const double R = t2 + t1, hfsq = 0.5 * f * f;                                                                                   \
        const double lg = dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);          \
        const double r = __builtin_elementwise_sqrt(-2.0 * lg);                                                                                   \
        /* sin / cos of theta = 2 pi u1 */                                                                                              \
        const double theta = 2.0 * 3.14159265358979323846 * u1;                                                                         \
        const int q = (int)(theta * 6.36619772367581382433e-01 + 0.5);                                                                  \
        const double dq = (double)q;                                                                                                    \
        const double y = (theta - dq * 1.57079632673412561417e+00) - dq * 6.07710050650619224932e-11;                                   \
        const double yy = y * y;                                                                                                        \
        const double ps = 8.33333333332248946124e-03 + yy * (-1.98412698298579493134e-04 + yy * (2.75573137070700676789e-06 +           \
                          yy * (-2.50507602534068634195e-08 + yy * 1.58969099521155010221e-10)));                                       \
        const double sy = y + y * yy * (-1.66666666666666324348e-01 + yy * ps);                                                         \
        const double pc = 4.16666666666666019037e-02 + yy * (-1.38888888888741095749e-03 + yy * (2.48015872894767294178e-05 +           \
                          yy * (-2.75573143513906633035e-07 + yy * (2.08757232129817482790e-09 + yy * -1.13596475577881948265e-11))));  \
        const double cy = (1.0 - 0.5 * yy) + yy * yy * pc;                                                                              \
        const bool odd = (q & 1) != 0;                                                                                                  \
        double sn = odd ? cy : sy, cs = odd ? sy : cy;                                                                                  \
        sn = (q & 2) ? -sn : sn;                                                                                                        \
        cs = ((q + 1) & 2) ? -cs : cs;                                                                                                  \
        const double vc = r * cs, vs = r * sn, dl = r * 0x1p-44;                                                                        \
        const float fc = (float)vc, fs = (float)vs;                                                                                     \
        const bool ok = (float)(vc - dl) == fc && (float)(vc + dl) == fc && (float)(vs - dl) == fs && (float)(vs + dl) == fs && r > 0.0; \
        zc[i] = fc; zs[i] = fs; redo[i] = ok ? 0 : 1; EXTRA                                                                                 \
    }
#define LS_FAST_ARGS const uint32_t* __restrict__ w, int np, float* __restrict__ zc, float* __restrict__ zs, uint8_t* __restrict__ redo
#define LS_FAST_DBG_ARGS LS_FAST_ARGS, double* __restrict__ dc, double* __restrict__ ds
#define LS_FAST_DBG_STORE dc[i] = vc; ds[i] = vs;
void fast_pairs_base(LS_FAST_ARGS) { LS_FAST_PAIRS_BODY() }
void fast_pairs_dbg_base(LS_FAST_DBG_ARGS) { LS_FAST_PAIRS_BODY(LS_FAST_DBG_STORE) }           // the *_dbg forms also hand out the doubles (ls_trng_pairs_debug)
#if LS_TRNG_X86
#define LS_FAST_T512 __attribute__((target("avx512f,avx512dq,avx512vl,avx512bw,fma"), min_vector_width(512)))
__attribute__((target("avx2,fma"))) void fast_pairs_avx2(LS_FAST_ARGS) { LS_FAST_PAIRS_BODY() }
__attribute__((target("avx2,fma"))) void fast_pairs_dbg_avx2(LS_FAST_DBG_ARGS) { LS_FAST_PAIRS_BODY(LS_FAST_DBG_STORE) }
LS_FAST_T512 void fast_pairs_avx512(LS_FAST_ARGS) { LS_FAST_PAIRS_BODY() }
LS_FAST_T512 void fast_pairs_dbg_avx512(LS_FAST_DBG_ARGS) { LS_FAST_PAIRS_BODY(LS_FAST_DBG_STORE) }
inline int fast_isa() {
    static const int isa = (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vl") &&
                            __builtin_cpu_supports("avx512bw")) ? 2 : (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) ? 1 : 0;
    return isa;
}
#else
inline void fast_pairs_avx2(LS_FAST_ARGS) { fast_pairs_base(w, np, zc, zs, redo); }
inline void fast_pairs_avx512(LS_FAST_ARGS) { fast_pairs_base(w, np, zc, zs, redo); }
inline void fast_pairs_dbg_avx2(LS_FAST_DBG_ARGS) { fast_pairs_dbg_base(w, np, zc, zs, redo, dc, ds); }
inline void fast_pairs_dbg_avx512(LS_FAST_DBG_ARGS) { fast_pairs_dbg_base(w, np, zc, zs, redo, dc, ds); }
inline int fast_isa() { return 0; }
#endif
std::atomic<uint64_t> g_redone{0}, g_pairs{0};           // how many pairs took the libm path (ls_trng_stats; tests and tools only)
// np <= kFastChunk pairs from 4 np words: zc[i] / zs[i] = the floats torch stores for the cos / sin branch of pair i
void pairs_to_floats(const uint32_t* w, int np, float* zc, float* zs) {
    uint8_t redo[kFastChunk];
    const int isa = fast_isa();
    if (isa == 2) fast_pairs_avx512(w, np, zc, zs, redo); else if (isa == 1) fast_pairs_avx2(w, np, zc, zs, redo); else fast_pairs_base(w, np, zc, zs, redo);
    int again = 0;
    for (int i = 0; i < np; ++i) {
        if (!redo[i]) continue;
        double c, s;
        normal_pair(w + 4 * i, c, s);
        zc[i] = (float)(c * 1.0 + 0.0);
        zs[i] = (float)(s * 1.0 + 0.0);
        ++again;
    }
    g_pairs.fetch_add((uint64_t)np, std::memory_order_relaxed);
    if (again) g_redone.fetch_add((uint64_t)again, std::memory_order_relaxed);
}

// An element-at-a-time double draw of n elements, element e stored at dst(e).  The generator's cached sample is consumed first and the
// one left over by an odd count is computed HERE (the next draw needs it before the workers have run).
struct SerialShape { int B, J, F, T; bool permuted; };    // permuted: element e is in MEMORY order [T][B][J][F] of a [B][J][F][T] tensor
inline size_t dst_index(const SerialShape& j, size_t e) {
    if (!j.permuted) return e;
    const size_t bjf = (size_t)j.B * j.J * j.F, t = e / bjf, r = e - t * bjf;       // r = (b * J + j) * F + f
    return r * j.T + t;
}
// Word buffers of the serial draws in flight: a ring of uninitialised arrays kept by the calling thread between calls (fresh 4 MB
// vectors per step cost a zero-fill and page faults on the producer's critical path); wrapping around waits for the workers.
struct WordRing {
    static constexpr int kSlots = 8;
    std::unique_ptr<uint32_t[]> buf[kSlots];
    size_t cap = 0;
    int used = 0;
    uint32_t* get(Pool& pool, size_t n) {
        if (n > cap || used == kSlots) {
            { LS_T0 pool.wait(); LS_T1(g_t_ringwait) }
            used = 0;
            if (n > cap) { for (auto& b : buf) b.reset(new uint32_t[n]); cap = n; }
        }
        return buf[used++].get();
    }
};
void draw_serial(Mt& g, Pool& pool, WordRing& ring, float* out, size_t n, SerialShape sh) {
    size_t e0 = 0;
    if (g.cached_valid && n > 0) { out[dst_index(sh, 0)] = (float)(g.cached * 1.0 + 0.0); g.cached_valid = 0; e0 = 1; }
    const size_t pairs = (n - e0 + 1) / 2;
    if (!pairs) return;
    uint32_t* w = ring.get(pool, 4 * pairs);
    const Mt::Pending ready = g.fill_words(w, 4 * pairs);
    if ((n - e0) & 1) {                  // the last pair's sin branch stays in the generator
        double zc, zs;
        words_ready(ready);
        normal_pair(w + 4 * (pairs - 1), zc, zs);
        g.cached = zs; g.cached_valid = 1;
    }
    if (!sh.permuted) {
        pool.run(pairs, 8192, [w, out, e0, n, ready](size_t a, size_t b) {
            words_ready(ready);
            float zc[kFastChunk], zs[kFastChunk];
            for (size_t p = a; p < b; p += kFastChunk) {
                const int np = (int)(b - p < (size_t)kFastChunk ? b - p : (size_t)kFastChunk);
                pairs_to_floats(w + 4 * p, np, zc, zs);
                for (int i = 0; i < np; ++i) {
                    const size_t e = e0 + 2 * (p + (size_t)i);
                    out[e] = zc[i];
                    if (e + 1 < n) out[e + 1] = zs[i];
                }
            }
        });
        return;
    }
    // Memory-order draw of a [B][J][F][T] tensor: element e = t * BJF + r lands at r * T + t.  The work is dealt by DESTINATION (a range
    // of r for every t), so that a worker owns whole cache lines of `out`; dealt by e, neighbouring t planes -- interleaved in memory --
    // would be written by different threads at the same time.  A pair that straddles two ranges is evaluated by both.
    const size_t bjf = (size_t)sh.B * sh.J * sh.F, T = (size_t)sh.T;
    pool.run(bjf, 256, [w, out, e0, bjf, T, ready](size_t ra, size_t rb) {
        words_ready(ready);
        LS_T0
        float zc[kFastChunk], zs[kFastChunk];
        for (size_t t = 0; t < T; ++t) {
            size_t e_first = t * bjf + ra;
            const size_t e_end = t * bjf + rb;          // <= n
            if (e_first < e0) ++e_first;                 // element 0 came from the generator's cached sample
            if (e_first >= e_end) continue;
            const size_t p_lo = (e_first - e0) >> 1, p_hi = (e_end - 1 - e0) >> 1;
            float* col = out + t;                        // element e of this plane: col[(e - t * bjf) * T]
            for (size_t p = p_lo; p <= p_hi; p += kFastChunk) {
                const int np = (int)(p_hi + 1 - p < (size_t)kFastChunk ? p_hi + 1 - p : (size_t)kFastChunk);
                pairs_to_floats(w + 4 * p, np, zc, zs);
                for (int i = 0; i < np; ++i) {
                    const size_t e = e0 + 2 * (p + (size_t)i);
                    if (e >= e_first) col[(e - t * bjf) * T] = zc[i];                    // e < e_end by the choice of p_hi
                    if (e + 1 >= e_first && e + 1 < e_end) col[(e + 1 - t * bjf) * T] = zs[i];
                }
            }
        }
        LS_TA(g_ns_xform)
    });
}

}  // namespace

extern "C" {

// torch.randn(n) (contiguous float32) / torch.randn_like of a [T][B][J][F]-memory-order view, from the state blob; see ls_hip.h
int ls_trng_randn(uint8_t* state, size_t state_bytes, float* out, size_t n, int variant, int n_threads) {
    if (!state || state_bytes != kStateBytes || (!out && n) || variant < 0 || variant > 4) return LS_EINVAL;
    if (variant > 0 && !have_fma()) return LS_EUNSUPPORTED;
    try {
    Mt g;
    if (!g.load(state)) return LS_EINVAL;
    {
        PoolPair pp = acquire_pools(n >= 65536 && n_threads > 1 ? n_threads : 0,
                                    n >= (kParBlocks + 2) * (size_t)kN && n_threads > 1 ? (n_threads < 12 ? n_threads / 2 : 6) : 0);      // word generators of long fills
        Pool& pool = *pp.x;
        Pool& gpool = *pp.g;
        g.gen = gpool.threads() ? &gpool : nullptr;
        static thread_local WordRing ring;
        ring.used = 0;
        if (n >= 16) draw_contig(g, pool, out, n, variant);
        else draw_serial(g, pool, ring, out, n, SerialShape{1, 1, 1, (int)n, false});
        pool.wait();
        gpool.wait();
        if (pool.take_failure() | gpool.take_failure()) return LS_ENOMEM;
    }
    g.store(state);
    return LS_OK;
    } catch (...) { return LS_ENOMEM; }     // nothing (a failed allocation of a word buffer, a thread that could not start) crosses the C ABI
}

int ls_trng_fill_steps(uint8_t* state, size_t state_bytes, int B, int D, int J, int F, int T, int n_steps, int first_contiguous, float* eps,
                       float* noise, int variant, int n_threads) {
    if (variant < 0 || variant > 4) return LS_EINVAL;
    if (variant > 0 && !have_fma()) return LS_EUNSUPPORTED;
    if (!state || state_bytes != kStateBytes || !eps || !noise || B < 1 || D < 1 || J < 1 || F < 1 || T < 1 || n_steps < 0) return LS_EINVAL;
    try {
    Mt g;
    if (!g.load(state)) return LS_EINVAL;
    const size_t ne = (size_t)B * D, nx = (size_t)B * J * F * T;
    {
        // The calling thread walks the steps in the reference's draw order producing words; the workers transform behind it.
#ifdef LS_TRNG_TIMING
        const double t_begin = now_s();
#endif
        // word generators of long fills (Mt::fill_words): behind the scout six keep up with it; with jump-ahead nobody walks the stream and
        // the generators are the producers, so more of them shorten a fill.  (Generator + transform threads <= 2 n_threads.)
        PoolPair pp = acquire_pools(n_threads > 1 ? n_threads : 0,
                                    n_threads > 1 ? (n_threads < 12 ? n_threads / 2 : (g_use_jump.load(std::memory_order_relaxed) ? 12 : 6)) : 0);
        Pool& pool = *pp.x;
        Pool& gpool = *pp.g;
        g.gen = gpool.threads() ? &gpool : nullptr;
        static thread_local WordRing ring;
        ring.used = 0;
        for (int k = 0; k < n_steps; ++k) {
            float* ec = eps + (size_t)(2 * k) * ne;
            float* nz = noise + (size_t)k * nx;
            for (float* p : {ec, ec + ne}) {
                if (ne >= 16) draw_contig(g, pool, p, ne, variant);
                else draw_serial(g, pool, ring, p, ne, SerialShape{1, 1, 1, (int)ne, false});      // never the case for D = 512; kept exact anyway
            }
            const bool first_c = k == 0 && first_contiguous;
            if (first_c && nx >= 16) draw_contig(g, pool, nz, nx, variant);
            else draw_serial(g, pool, ring, nz, nx, SerialShape{B, J, F, T, !first_c});             // a contiguous x of < 16 elements: serial, in order
        }
        { LS_T0 pool.wait(); gpool.wait(); LS_T1(g_t_finalwait) }
        if (pool.take_failure() | gpool.take_failure()) return LS_ENOMEM;
#ifdef LS_TRNG_TIMING
        fprintf(stderr, "trng timing (s): total %.4f scout %.4f genwait %.4f serial %.4f ringwait %.4f submit %.4f finalwait %.4f steps %d\n", now_s() - t_begin, g_t_scout, g_t_genwait, g_t_serial, g_t_ringwait, g_t_submit, g_t_finalwait, n_steps);
        fprintf(stderr, "   thread-ms: gen %.3f spin %.3f xform %.3f\n", g_ns_gen.exchange(0) * 1e-6, g_ns_spin.exchange(0) * 1e-6, g_ns_xform.exchange(0) * 1e-6);
        g_t_submit = 0;
        g_t_scout = g_t_genwait = g_t_serial = g_t_ringwait = g_t_finalwait = 0;
#endif
    }
    g.store(state);
    return LS_OK;
    } catch (...) { return LS_ENOMEM; }
}

// how many double pairs were evaluated, and how many of them went back to libm (process-wide; tests and tools)
// The mt19937 jump-ahead of long fills (ls_mt_jump.h): on (default) / off = round 5's sequential scout; returns the previous setting.
int ls_trng_set_jump(int on) { return g_use_jump.exchange(on ? 1 : 0); }

// Self-check of the jump-ahead: the state `words` (a multiple of 624) further on from init_genrand(seed), by the jump polynomial and by
// running the recurrence; 0 = identical in every bit that is part of the state (word 0's top bit, words 1 .. 623), 1 = different,
// negative = the characteristic polynomial could not be built.  *support = number of windows the jump XORs.
int ls_trng_jump_check(uint32_t seed, uint64_t words, int* support) {
    if (words == 0 || words % kN) return LS_EINVAL;
    const mtjump::Field& f = mtjump::field();
    if (!f.ok) return LS_EUNSUPPORTED;
    uint32_t st[kN], ref[kN], got[kN];
    st[0] = seed;
    for (int j = 1; j < kN; ++j) st[j] = 1812433253u * (st[j - 1] ^ (st[j - 1] >> 30)) + (uint32_t)j;
    memcpy(ref, st, sizeof st);
    for (uint64_t b = 0; b < words / kN; ++b) mt_block_base(ref);
    auto sup = mtjump::jump_support(words, 1);
    if (!sup) return LS_EUNSUPPORTED;
    if (support) *support = (int)sup->size();
    std::vector<uint32_t> X((size_t)mtjump::kXWords);
    mtjump::expand(st, X.data());
    for (int isa = 0; isa <= mt_isa(); ++isa) {
        mtjump::apply(X.data(), *sup, got, isa);
        if ((got[0] ^ ref[0]) & 0x80000000u) return 1;
        if (memcmp(got + 1, ref + 1, (kN - 1) * sizeof(uint32_t)) != 0) return 1;
    }
    return 0;
}

int ls_trng_stats(uint64_t* pairs, uint64_t* redone) {
    if (!pairs || !redone) return LS_EINVAL;
    *pairs = g_pairs.load(std::memory_order_relaxed);
    *redone = g_redone.load(std::memory_order_relaxed);
    return LS_OK;
}

// The vectorised evaluation by itself, for tests: np pairs from 4 np words -> the doubles r cos / r sin it computed, the floats it would
// store and the flag "this pair goes back to libm"; libm_c / libm_s = normal_pair's doubles (torch's).  isa: 0 base, 1 AVX2, 2 AVX-512
// (LS_EUNSUPPORTED if this machine lacks it), -1 = the one pairs_to_floats uses here.
int ls_trng_pairs_debug(const uint32_t* words, int np, int isa, double* fast_c, double* fast_s, float* zc, float* zs, uint8_t* redo,
                        double* libm_c, double* libm_s) {
    if (!words || np < 0 || !fast_c || !fast_s || !zc || !zs || !redo || !libm_c || !libm_s || isa < -1 || isa > 2) return LS_EINVAL;
    if (isa < 0) isa = fast_isa();
    if (isa > fast_isa()) return LS_EUNSUPPORTED;
    for (int p = 0; p < np; p += kFastChunk) {
        const int m = np - p < kFastChunk ? np - p : kFastChunk;
        if (isa == 2) fast_pairs_dbg_avx512(words + 4 * p, m, zc + p, zs + p, redo + p, fast_c + p, fast_s + p);
        else if (isa == 1) fast_pairs_dbg_avx2(words + 4 * p, m, zc + p, zs + p, redo + p, fast_c + p, fast_s + p);
        else fast_pairs_dbg_base(words + 4 * p, m, zc + p, zs + p, redo + p, fast_c + p, fast_s + p);
    }
    for (int p = 0; p < np; ++p) normal_pair(words + 4 * p, libm_c[p], libm_s[p]);
    return LS_OK;
}

}  // extern "C"
