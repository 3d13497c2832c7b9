// Hand-off micro-probe for the sample-split step kernel (tools/probes, not part of the library): what one in-launch exchange among the F
// slice workgroups of a group costs on MI355X, by fan-in (2 / 4 / 8), placement (the F members on ONE XCD or on F different XCDs) and
// load (one group alone on the chip, or every CU in a group).  Two exchange shapes, the two of ls_coop_kernel.h:
//   stat  (SYNC1, LayerNorm partials): every member publishes 36 rows x 2 {tag, value} granules (relaxed agent-scope 8-byte stores) and
//         polls all F members' granules of every row until their tags match (sc1 loads)           -- ln_publish + ln_gather
//   rows  (SYNC2): every member writes a `bytes` payload with 16-byte write-through (sc1) stores, drains, barrier, publishes one flag
//         granule; polls the F flags, pulls the other F - 1 payloads global -> LDS by LDS-DMA (buffer_load ... lds, sc1), drains   -- the row hand-off
// A member's round r + 1 starts when its round r is complete: clocks per round = one hop of the chain (s_memtime, shader clocks).
//   hipcc --offload-arch=gfx950 -O3 -o handoff tools/probes/handoff.hip && ./handoff
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(1))) unsigned long long* gu64p;
typedef __attribute__((address_space(3))) void* lds_vp;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p) {
    const unsigned long long v = (unsigned long long)(uintptr_t)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((unsigned long long)hi << 32) | (unsigned long long)lo), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ unsigned long long gload(const unsigned long long* p) { return __hip_atomic_load((gu64p)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gstore(unsigned long long* p, unsigned tag, unsigned v) {
    __hip_atomic_store((gu64p)p, ((unsigned long long)tag << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct Args {
    unsigned long long* gran;     // [groups][2 parities][36 rows][8 members][2]
    float* pay;                   // [groups][2 parities][8 members][bytes / 4]
    unsigned long long* clk;      // [workgroups] clocks of the timed rounds
    unsigned* xcc;                // [workgroups]
    unsigned* bad;
    int F, same_xcd, active_groups, rounds, bytes, mode;   // mode 0 stat, 1 rows
};

constexpr int kRows = 36;

__global__ __launch_bounds__(512) void k_probe(const Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, bid = blockIdx.x;
    // blockIdx -> (group, member).  Observed placement: block b runs on XCD b % 8.
    //   same XCD:  the F members of a group are F blocks with the same b % 8:  b = x + 8 (F j + m), group = x + 8 j
    //   cross XCD: F consecutive blocks:                                          b = F g + m
    int g, m;
    if (a.same_xcd) { const int x = bid & 7, q = bid >> 3; m = q % a.F; g = x + 8 * (q / a.F); }
    else { g = bid / a.F; m = bid % a.F; }
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (tid == 0) a.xcc[bid] = xcc & 0xf;
    if (g >= a.active_groups) return;
    unsigned long long* gr = a.gran + (size_t)g * 2 * kRows * 8 * 2;
    float* pay = a.pay + (size_t)g * 2 * 8 * (a.bytes / 4);
    const __amdgpu_buffer_rsrc_t prs = rsrc(pay);
    unsigned bad = 0;
    unsigned long long t0 = 0;
    constexpr int kWarm = 8;
    for (int r = 0; r < a.rounds + kWarm; ++r) {
        if (r == kWarm) { __syncthreads(); t0 = __builtin_amdgcn_s_memtime(); }
        const unsigned tag = (unsigned)r + 1u;
        unsigned long long* ar = gr + (size_t)(r & 1) * kRows * 8 * 2;
        if (a.mode == 0) {
            // publish: thread `row` stores this member's two granules of the row (as ln_publish's tid < S threads do) ...
            if (tid < kRows) { gstore(ar + ((size_t)tid * 8 + m) * 2, tag, tid); gstore(ar + ((size_t)tid * 8 + m) * 2 + 1, tag, tid + 1); }
            // ... gather: thread (row = tid / F, member = tid % F) polls (as ln_gather)
            const int sl = tid % a.F, row = min(tid / a.F, kRows - 1);
            if (wv * (64 / a.F) < kRows) {
                for (unsigned spins = 0;; ++spins) {
                    const unsigned long long v0 = gload(ar + ((size_t)row * 8 + sl) * 2), v1 = gload(ar + ((size_t)row * 8 + sl) * 2 + 1);
                    if (__all((unsigned)(v0 >> 32) == tag && (unsigned)(v1 >> 32) == tag)) break;
                    if (spins > (1u << 20)) { bad = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
        } else {
            // payload: `bytes` of this member, 16-byte write-through stores over the workgroup; every wave drains; barrier; one flag granule
            float* mine = pay + ((size_t)(r & 1) * 8 + m) * (a.bytes / 4);
            const __amdgpu_buffer_rsrc_t mrs = rsrc(mine);
            for (int o = tid * 16; o < a.bytes; o += 512 * 16)
                __builtin_amdgcn_raw_buffer_store_b128((u4v){tag, tag, tag, tag}, mrs, o, 0, 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (tid == 0) gstore(ar + (size_t)m * 2, tag, 0);
            for (unsigned spins = 0;; ++spins) {
                const unsigned long long v = gload(ar + (size_t)(lane % a.F) * 2);
                if (__all((unsigned)(v >> 32) == tag)) break;
                if (spins > (1u << 20)) { bad = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            // pull the other members' payloads into LDS: 1 KiB chunks dealt over the 8 waves
            const int chunks = a.bytes / 1024;
            for (int i = 1; i < a.F; ++i) {
                const int s = (m + i) % a.F;
                const int base = ((r & 1) * 8 + s) * a.bytes;
                for (int ch = wv; ch < chunks; ch += 8)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(prs, (lds_vp)(smem + (size_t)(s * chunks + ch) * 256), 16, lane * 16, base + ch * 1024, 0, 16);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (a.F > 1 && __float_as_uint(smem[(size_t)(((m + 1) % a.F) * chunks) * 256 + tid]) != tag) bad = 2;      // the pulled data is this round's
        }
    }
    if (tid == 0) { a.clk[bid] = __builtin_amdgcn_s_memtime() - t0; if (bad) atomicOr(a.bad, bad); }
}

int main() {
    const int kCU = 256, rounds = 400, maxbytes = 36 * 256 * 4;     // up to one 256-channel slice: 36 KB
    Args a{};
    hipMalloc(&a.gran, (size_t)kCU * 2 * kRows * 8 * 2 * 8);
    hipMalloc(&a.pay, (size_t)kCU * 2 * 8 * maxbytes);
    hipMalloc(&a.clk, kCU * 8); hipMalloc(&a.xcc, kCU * 4); hipMalloc(&a.bad, 4);
    const int lds = 8 * maxbytes > 150000 ? 150000 : 8 * maxbytes;  // one workgroup per CU
    hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("# hand-off probe: %d timed rounds per configuration; clocks = s_memtime ticks per round (median over the active workgroups); us = kernel wall / rounds\n", rounds);
    printf("# mode  bytes/member  fan-in  placement  load        clocks/round   us/round   xcds-per-group\n");
    struct Cfg { int mode, bytes; };
    const Cfg cfgs[] = {{0, 0}, {1, 9216}, {1, 18432}, {1, 36864}};
    for (const Cfg& c : cfgs)
        for (int F : {2, 4, 8})
            for (int same = 0; same < 2; ++same)
                for (int loaded = 0; loaded < 2; ++loaded) {
                    if (c.mode == 1 && (size_t)F * c.bytes > (size_t)lds) continue;
                    a.mode = c.mode; a.bytes = c.mode ? c.bytes : 1024; a.F = F; a.same_xcd = same; a.rounds = rounds;
                    const int groups = kCU / F;
                    a.active_groups = loaded ? groups : 1;
                    hipMemset(a.gran, 0, (size_t)kCU * 2 * kRows * 8 * 2 * 8); hipMemset(a.bad, 0, 4); hipMemset(a.clk, 0, kCU * 8);
                    hipLaunchKernelGGL(k_probe, dim3(kCU), dim3(512), lds, 0, a);      // warm
                    hipDeviceSynchronize();
                    hipMemset(a.gran, 0, (size_t)kCU * 2 * kRows * 8 * 2 * 8); hipMemset(a.clk, 0, kCU * 8);
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(k_probe, dim3(kCU), dim3(512), lds, 0, a);
                    hipEventRecord(e1);
                    hipDeviceSynchronize();
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    std::vector<unsigned long long> clk(kCU); std::vector<unsigned> xcc(kCU); unsigned bad;
                    hipMemcpy(clk.data(), a.clk, kCU * 8, hipMemcpyDeviceToHost); hipMemcpy(xcc.data(), a.xcc, kCU * 4, hipMemcpyDeviceToHost);
                    hipMemcpy(&bad, a.bad, 4, hipMemcpyDeviceToHost);
                    std::vector<double> v;
                    for (int b = 0; b < kCU; ++b) if (clk[b]) v.push_back((double)clk[b] / rounds);
                    std::sort(v.begin(), v.end());
                    // XCDs the members of group 0 actually sat on
                    unsigned mask = 0;
                    for (int b = 0; b < kCU; ++b) {
                        int g = same ? ((b & 7) + 8 * ((b >> 3) / F)) : b / F;
                        if (g == 0) mask |= 1u << xcc[b];
                    }
                    printf("%-5s %8d %9d   %-9s  %-10s %12.0f %10.3f   %d%s\n", c.mode ? "rows" : "stat", c.mode ? c.bytes : 576, F, same ? "one XCD" : "F XCDs",
                           loaded ? "all CUs" : "one group", v.empty() ? 0.0 : v[v.size() / 2], ms * 1e3 / (rounds + 8), __builtin_popcount(mask), bad ? "  TIMEOUT/STALE" : "");
                }
    return 0;
}
