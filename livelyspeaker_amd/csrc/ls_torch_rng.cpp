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
#include <cmath>
#include <cstdint>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

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
    void fill_words(uint32_t* out, size_t n);
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
__attribute__((target("avx512f,prefer-vector-width=512"))) void mt_block_avx512(uint32_t* __restrict__ st) { LS_MT_BLOCK_BODY }
__attribute__((target("avx512f,prefer-vector-width=512"))) void mt_temper_avx512(const uint32_t* __restrict__ src, uint32_t* __restrict__ out, size_t m) { LS_MT_TEMPER_BODY }
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

void Mt::fill_words(uint32_t* out, size_t n) {
    const int isa = mt_isa();
    if (gen && n >= (kParBlocks + 2) * (size_t)kN) {
        // head: the rest of the current block
        if (left - 1 > 0) {
            const size_t m = (size_t)(left - 1);
            if (isa == 2) mt_temper_avx512(st + next, out, m); else if (isa == 1) mt_temper_avx2(st + next, out, m); else mt_temper_base(st + next, out, m);
            next += (uint32_t)m; left -= (int)m; out += m; n -= m;
        }
        // middle: whole blocks, in pieces.  A piece's generator starts from a snapshot of the state in front of it; the scout (this thread)
        // runs the recurrence alone over the piece to reach the next snapshot
        const size_t nb = n / kN;
        const int nt = pool_threads(gen) > 0 ? pool_threads(gen) : 1;
        size_t per = (nb + (size_t)(4 * nt) - 1) / (size_t)(4 * nt);
        if (per < 64) per = 64;
        for (size_t b0 = 0; b0 < nb; b0 += per) {
            const size_t cnt = b0 + per < nb ? per : nb - b0;
            auto snap = std::make_shared<std::array<uint32_t, kN>>();
            memcpy(snap->data(), st, sizeof st);
            uint32_t* dst = out + b0 * kN;
            pool_submit(gen, [snap, dst, cnt, isa] { mt_blocks_to_words(snap->data(), dst, cnt, isa); });
            for (size_t b = 0; b < cnt; ++b) { if (isa == 2) mt_block_avx512(st); else if (isa == 1) mt_block_avx2(st); else mt_block_base(st); }
        }
        pool_wait(gen);
        out += nb * kN; n -= nb * kN;
        left = 1; next = kN;                 // the last block is used up, exactly as after reading it word by word
    }
    while (n > 0) {
        if (left - 1 == 0) {                 // word(): --left == 0 -> next_state(), and the word it then reads leaves left at kN
            if (isa == 2) mt_block_avx512(st); else if (isa == 1) mt_block_avx2(st); else mt_block_base(st);
            left = kN + 1; next = 0;
        }
        const size_t m = n < (size_t)(left - 1) ? n : (size_t)(left - 1);
        if (isa == 2) mt_temper_avx512(st + next, out, m); else if (isa == 1) mt_temper_avx2(st + next, out, m); else mt_temper_base(st + next, out, m);
        next += (uint32_t)m; left -= (int)m; out += m; n -= m;
    }
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
LS_FMA_TARGET inline float cephes_logf(float x, int variant) {
    const bool invalid = x <= 0.0f;
    x = std::fmax(x, as_f(0x00800000u));                    // cut off denormalized stuff
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
    y = fmaf(y, x, -1.1514610310E-1f);
    y = fmaf(y, x, 1.1676998740E-1f);
    y = fmaf(y, x, -1.2420140846E-1f);
    y = fmaf(y, x, 1.4249322787E-1f);
    y = fmaf(y, x, -1.6668057665E-1f);
    y = fmaf(y, x, 2.0000714765E-1f);
    y = fmaf(y, x, -2.4999993993E-1f);
    y = fmaf(y, x, 3.3333331174E-1f);
    y = y * x;
    if (variant & 1) y = fmaf(e, -2.12194440e-4f, y * z);
    else y = fmaf(y, z, e * -2.12194440e-4f);
    y = fmaf(-z, 0.5f, y);
    x = fmaf(e, 0.693359375f, x + y);
    return invalid ? as_f(0xffffffffu) : x;
}

LS_FMA_TARGET inline void cephes_sincosf(float xin, int variant, float& s, float& c) {
    uint32_t sign_sin = as_u(xin) & 0x80000000u;
    float x = as_f(as_u(xin) & 0x7fffffffu);
    float y = x * 1.27323954473516f;                         // 4 / pi
    int j = (int)y;                                          // cvttps
    j = (j + 1) & ~1;
    y = (float)j;
    const uint32_t swap_sign_sin = ((uint32_t)(j & 4)) << 29;
    const bool poly = (j & 2) == 0;
    x = fmaf(y, -0.78515625f, x);
    x = fmaf(y, -2.4187564849853515625e-4f, x);
    x = fmaf(y, -3.77489497744594108e-8f, x);
    const uint32_t sign_cos = ((uint32_t)(~(j - 2) & 4)) << 29;
    sign_sin ^= swap_sign_sin;
    const float z = x * x;
    float yc = 2.443315711809948E-005f;
    yc = fmaf(yc, z, -1.388731625493765E-003f);
    yc = fmaf(yc, z, 4.166664568298827E-002f);
    yc = yc * z;
    if (variant & 2) yc = fmaf(-z, 0.5f, yc * z);
    else yc = fmaf(yc, z, -(z * 0.5f));
    yc = yc + 1.0f;
    float ys = -1.9515295891E-4f;
    ys = fmaf(ys, z, 8.3321608736E-3f);
    ys = fmaf(ys, z, -1.6666654611E-1f);
    ys = ys * z;
    ys = fmaf(ys, x, x);
    const float rs = poly ? ys : yc, rc = poly ? yc : ys;
    s = as_f(as_u(rs) ^ sign_sin);
    c = as_f(as_u(rc) ^ sign_cos);
}

LS_FMA_TARGET void fill16_cephes(float* d, int variant) {
    const float two_pi = 2.0f * 3.14159265358979323846;
    for (int j = 0; j < 8; ++j) {
        const float u1 = 1.0f - d[j];
        const float u2 = d[j + 8];
        const float radius = std::sqrt(-2.0f * cephes_logf(u1, variant - 1));
        const float theta = two_pi * u2;
        float sn, cs;
        cephes_sincosf(theta, variant - 1, sn, cs);
        d[j] = fmaf(radius * cs, 1.0f, 0.0f);
        d[j + 8] = fmaf(radius * sn, 1.0f, 0.0f);
    }
}

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
    // fn(a, b) over [0, n) in chunks; runs inline without workers or for small jobs
    void run(size_t n, size_t chunk, std::function<void(size_t, size_t)> fn) {
        if (th_.empty() || n <= chunk) { if (n) fn(0, n); return; }
        auto f = std::make_shared<std::function<void(size_t, size_t)>>(std::move(fn));
        {
            std::lock_guard<std::mutex> l(m_);
            for (size_t a = 0; a < n; a += chunk) { q_.push_back([f, a, n, chunk] { (*f)(a, a + chunk < n ? a + chunk : n); }); ++pending_; }
        }
        cv_.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [this] { return pending_ == 0; });
    }
    void submit(std::function<void()> job) {            // one job; runs inline without workers
        if (th_.empty()) { job(); return; }
        { std::lock_guard<std::mutex> l(m_); q_.push_back(std::move(job)); ++pending_; }
        cv_.notify_one();
    }
    int threads() const { return (int)th_.size(); }
private:
    void work() {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [this] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;
                job = std::move(q_.front());
                q_.pop_front();
            }
            job();
            { std::lock_guard<std::mutex> l(m_); if (--pending_ == 0) done_.notify_all(); }
        }
    }
    std::vector<std::thread> th_;
    std::deque<std::function<void()>> q_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    size_t pending_ = 0;
    bool stop_ = false;
};

void pool_submit(Pool* p, std::function<void()> job) { p->submit(std::move(job)); }
void pool_wait(Pool* p) { p->wait(); }
int pool_threads(Pool* p) { return p->threads(); }

inline float word_to_uf(uint32_t w) { return (float)(w & ((1u << 24) - 1)) * (1.0f / (float)(1u << 24)); }                 // uniform_real<float>
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
    g.fill_words(w, n);
    const size_t blocks = n / 16;
    auto block = [variant](float* d) {
        uint32_t* u = reinterpret_cast<uint32_t*>(d);
        float f[16];
        for (int i = 0; i < 16; ++i) f[i] = word_to_uf(u[i]);
        fill16(f, variant);
        memcpy(d, f, sizeof f);
    };
    if (n % 16) {                        // rare: finish this job before the tail overwrites the end of its last full block
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
    pool.run(blocks, 2048, [out, block](size_t a, size_t b) { for (size_t i = a; i < b; ++i) block(out + 16 * i); });
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
            pool.wait();
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
    g.fill_words(w, 4 * pairs);
    if ((n - e0) & 1) {                  // the last pair's sin branch stays in the generator
        double zc, zs;
        normal_pair(w + 4 * (pairs - 1), zc, zs);
        g.cached = zs; g.cached_valid = 1;
    }
    if (!sh.permuted) {
        pool.run(pairs, 8192, [w, out, e0, n](size_t a, size_t b) {
            for (size_t p = a; p < b; ++p) {
                double zc, zs;
                normal_pair(w + 4 * p, zc, zs);
                const size_t e = e0 + 2 * p;
                out[e] = (float)(zc * 1.0 + 0.0);
                if (e + 1 < n) out[e + 1] = (float)(zs * 1.0 + 0.0);
            }
        });
        return;
    }
    // Memory-order draw of a [B][J][F][T] tensor: element e = t * BJF + r lands at r * T + t.  The work is dealt by DESTINATION (a range
    // of r for every t), so that a worker owns whole cache lines of `out`; dealt by e, neighbouring t planes -- interleaved in memory --
    // would be written by different threads at the same time.  A pair that straddles two ranges is evaluated by both.
    const size_t bjf = (size_t)sh.B * sh.J * sh.F, T = (size_t)sh.T;
    pool.run(bjf, 256, [w, out, e0, n, bjf, T](size_t ra, size_t rb) {
        for (size_t t = 0; t < T; ++t) {
            size_t e = t * bjf + ra;
            const size_t e_end = t * bjf + rb;          // <= n
            if (e < e0) ++e;                             // element 0 came from the generator's cached sample
            double zc, zs;
            if (e < e_end && ((e - e0) & 1)) {           // second member of a pair that starts in the previous range
                normal_pair(w + 4 * ((e - e0) >> 1), zc, zs);
                out[(e - t * bjf) * T + t] = (float)(zs * 1.0 + 0.0);
                ++e;
            }
            for (; e < e_end; e += 2) {
                normal_pair(w + 4 * ((e - e0) >> 1), zc, zs);
                out[(e - t * bjf) * T + t] = (float)(zc * 1.0 + 0.0);
                if (e + 1 < e_end) out[(e + 1 - t * bjf) * T + t] = (float)(zs * 1.0 + 0.0);
            }
        }
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
        Pool pool(n >= 65536 && n_threads > 1 ? n_threads : 0);
        Pool gpool(n >= (kParBlocks + 2) * (size_t)kN && n_threads > 1 ? (n_threads < 12 ? n_threads / 2 : 6) : 0);      // word generators of long fills
        g.gen = gpool.threads() ? &gpool : nullptr;
        static thread_local WordRing ring;
        ring.used = 0;
        if (n >= 16) draw_contig(g, pool, out, n, variant);
        else draw_serial(g, pool, ring, out, n, SerialShape{1, 1, 1, (int)n, false});
        pool.wait();
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
        Pool pool(n_threads > 1 ? n_threads : 0);
        Pool gpool(n_threads > 1 ? (n_threads < 12 ? n_threads / 2 : 6) : 0);      // word generators of long fills (Mt::fill_words)
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
        pool.wait();
    }
    g.store(state);
    return LS_OK;
    } catch (...) { return LS_ENOMEM; }
}

}  // extern "C"
