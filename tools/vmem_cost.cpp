// What one vector-memory instruction costs the fp32 matrix pipe on gfx950: a wave runs 64 MFMAs (+12 ds_read_b128) per iteration and
// NL extra instructions of one kind; the slowdown over NL = 0, divided by NL, is the cost per instruction in matrix-pipe cycles.
//   kinds: 0 global_load_dwordx4 (per-lane 64-bit address)   1 global_load_dword   2 buffer_load_dwordx4 (SGPR resource + VGPR offset)
//          3 ds_write_b128   4 ds_write_b32   5 global_load_dwordx4 of ONE address for the whole wave (no fan-out in the TA)
//          9 buffer_load_dwordx4 ... lds (LDS-DMA)  10 global_load_lds_dwordx4  11-13 the 16-byte LDS store as 2 x b64 / write2_b64 / 2 x write2_b32
//          14 / 15 v_exp_f32 / v_rcp_f32 x 8 (transcendental unit)  16 v_pk_fma_f32 x 4
//          6 v_fma_f32 x 8 (VALU)  7 global_store_dwordx4  8 v_lshl_add_u64 (the 64-bit pointer bump hipcc emits per load)
//   hipcc --offload-arch=gfx950 -O3 -x hip tools/vmem_cost.cpp -o variants/vmem_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)


template <int KIND, int NL>
__global__ __launch_bounds__(256, 3) void k_loop(const float* __restrict__ src, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float sA[64 * 40], sB[128 * 40];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, s16 = lane & 15, g = lane >> 4;
    for (int i = tid; i < 64 * 40; i += 256) sA[i] = src[i];
    for (int i = tid; i < 128 * 40; i += 256) sB[i] = src[i + 64 * 40];
    __syncthreads();
    f4 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
    f4 rg[NL > 0 ? NL : 1];
    for (int i = 0; i < (NL > 0 ? NL : 1); ++i) rg[i] = (f4){1.f, 2.f, 3.f, 4.f};
    const float* gp = src + (size_t)blockIdx.x * 16384 + (KIND == 5 ? wv * 4 : tid * 4);
    float* op = out + 1048576 + (size_t)blockIdx.x * 16384 + tid * 4;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)blockIdx.x * 16384), 0, 0x7fffffff, 0x00020000);
    typedef float f2v __attribute__((ext_vector_type(2)));
    f2v pk[NL > 0 ? NL : 1][2], pkc = {1.0001f, 0.5f};
    for (int i = 0; i < (NL > 0 ? NL : 1); ++i) { pk[i][0] = (f2v){1.f, 2.f}; pk[i][1] = (f2v){3.f, 4.f}; }
    int iv[NL > 0 ? NL : 1][4];
    for (int i = 0; i < (NL > 0 ? NL : 1); ++i) for (int e = 0; e < 4; ++e) iv[i][e] = tid + e;
    int soff = 0;
    const unsigned ldsw = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)&sB[wv * 1024]);
    unsigned long long gp64[NL > 0 ? NL : 1], inc64 = 128;
    for (int i = 0; i < (NL > 0 ? NL : 1); ++i) gp64[i] = (unsigned long long)(uintptr_t)gp + i;
    const float* pa = &sA[((wv >> 1) * 32 + s16) * 40 + 4 * g];
    const float* pb = &sB[((wv & 1) * 64 + s16) * 40 + 4 * g];
    for (int it = 0; it < iters; ++it) {
        gp += 32; soff = (soff + 128) & 0x3fff;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            // asm volatile: the compiler must neither sink the loads out of the loop (their results are only read at the very end) nor merge stores
            const float* p = gp + 1024 * i;
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2 lo2 = {rg[i][0], rg[i][1]}, hi2 = {rg[i][2], rg[i][3]};
            const unsigned la = (unsigned)(uintptr_t)&sB[((tid >> 3) + 32 * (i & 3)) * 40 + (tid & 7) * 4];
            if (KIND == 0 || KIND == 5) asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(rg[i]) : "v"(p));
            if (KIND == 1) asm volatile("global_load_dword %0, %1, off" : "+v"(rg[i][0]) : "v"(p));
            if (KIND == 2) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(rg[i]) : "v"(tid * 16 + 4096 * i), "s"(rs), "s"(soff));
            if (KIND == 3) asm volatile("ds_write_b128 %0, %1" :: "v"(la), "v"(rg[i]));
            if (KIND == 4) asm volatile("ds_write_b32 %0, %1" :: "v"(la), "v"(rg[i][0]));
            if (KIND == 6) {
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(rg[i][e & 3]) : "v"(1.0001f));
            }
            if (KIND == 7) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(op + 1024 * i), "v"(rg[i]));
            if (KIND == 9) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
                                        :: "v"(tid * 16 + 4096 * i), "s"(rs), "s"(ldsw + 1024 * (i & 3)), "s"(soff) : "memory");
            if (KIND == 10) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(p), "s"(ldsw + 1024 * (i & 3)) : "memory");
            if (KIND == 11) asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:8" :: "v"(la), "v"(lo2), "v"(hi2));
            if (KIND == 12) asm volatile("ds_write2_b64 %0, %1, %2 offset1:1" :: "v"(la), "v"(lo2), "v"(hi2));
            if (KIND == 13) asm volatile("ds_write2_b32 %0, %1, %2 offset1:1\n\tds_write2_b32 %0, %3, %4 offset0:2 offset1:3" :: "v"(la), "v"(rg[i][0]), "v"(rg[i][1]), "v"(rg[i][2]), "v"(rg[i][3]));
            if (KIND == 14) {
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("v_exp_f32 %0, %0" : "+v"(rg[i][e & 3]));
            }
            if (KIND == 15) {
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("v_rcp_f32 %0, %0" : "+v"(rg[i][e & 3]));
            }
            if (KIND == 16) {
#pragma unroll
                for (int e = 0; e < 4; ++e) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pk[i][e & 1]) : "v"(pkc));
            }
            if (KIND == 17) {
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("v_add_u32 %0, %0, %1" : "+v"(iv[i][e & 3]) : "v"(3));
            }
            if (KIND == 18) {
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("v_min_u32 %0, %0, %1" : "+v"(iv[i][e & 3]) : "v"(1000000));
            }
            if (KIND == 19) {
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("v_cmp_lt_f32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(rg[i][e & 3]) : "v"(0.25f) : "vcc");
            }
            if (KIND == 8) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(gp64[i]) : "v"(inc64));
        }
        __builtin_amdgcn_sched_barrier(0);
        f4 af[2][2], bf[2];
        af[0][0] = *reinterpret_cast<const f4*>(pa); af[0][1] = *reinterpret_cast<const f4*>(pa + 16 * 40);
        bf[0] = *reinterpret_cast<const f4*>(pb);
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            const int kk = st >> 2, i = st & 3;
            if (st + 1 < 8) bf[(st + 1) & 1] = *reinterpret_cast<const f4*>(pb + 16 * 40 * ((st + 1) & 3) + 16 * ((st + 1) >> 2));
            if (st == 0) { af[1][0] = *reinterpret_cast<const f4*>(pa + 16); af[1][1] = *reinterpret_cast<const f4*>(pa + 16 * 40 + 16); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = MFMA(bf[st & 1][e], af[kk][j][e], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    f4 s = rg[0];
    for (int i = 0; i < NL; ++i) s[1] += (float)(iv[i][0] + iv[i][3]);
    for (int i = 0; i < NL; ++i) s[0] += (float)(gp64[i] & 1) + pk[i][0][0] + pk[i][1][1];
    for (int i = 1; i < NL; ++i) s += rg[i];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + tid] = s[0] + s[1] + s[2] + s[3];
}

static float base_ms = 0.f;
template <int KIND, int NL>
static void run(const float* src, float* out, const char* what) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 768, iters = 2000;
    hipLaunchKernelGGL((k_loop<KIND, NL>), dim3(grid), dim3(256), 0, 0, src, out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_loop<KIND, NL>), dim3(grid), dim3(256), 0, 0, src, out, iters);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (NL == 0) base_ms = ms;
    const double flop = (double)grid * 4 * iters * 64 * 2048.0;
    // one iteration of one wave = 64 MFMAs = 2048 matrix-pipe cycles when the pipe is saturated; extra time per iteration, in those cycles:
    const double extra = (ms / base_ms - 1.0) * 2048.0;
    std::printf("%-34s x%-2d: %.3f ms  %.3f of peak  -> %+.1f pipe cycles per instruction\n", what, NL, ms, flop / ms * 1e-9 / 157.3, NL ? extra / NL : 0.0);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    float *src, *out;
    const size_t n = (size_t)768 * 16384 + 2000 * 32 + 65536;
    (void)hipMalloc(&src, n * 4); (void)hipMalloc(&out, (1048576 + n) * 4);
    (void)hipMemset(src, 0, n * 4);
    run<0, 0>(src, out, "baseline (MFMA + ds_read_b128)");
    run<0, 3>(src, out, "global_load_dwordx4"); run<0, 6>(src, out, "global_load_dwordx4"); run<0, 12>(src, out, "global_load_dwordx4");
    run<1, 6>(src, out, "global_load_dword"); run<1, 12>(src, out, "global_load_dword");
    run<2, 6>(src, out, "buffer_load_dwordx4 (s offset)"); run<2, 12>(src, out, "buffer_load_dwordx4 (s offset)");
    run<5, 6>(src, out, "global_load_dwordx4, 1 address/wave"); 
    run<3, 6>(src, out, "ds_write_b128"); run<3, 12>(src, out, "ds_write_b128");
    run<4, 6>(src, out, "ds_write_b32"); run<4, 12>(src, out, "ds_write_b32");
    run<6, 6>(src, out, "8 x v_fma_f32"); run<6, 12>(src, out, "8 x v_fma_f32");
    run<7, 6>(src, out, "global_store_dwordx4");
    run<9, 6>(src, out, "buffer_load_dwordx4 ... lds"); run<9, 12>(src, out, "buffer_load_dwordx4 ... lds");
    run<10, 6>(src, out, "global_load_lds_dwordx4");
    run<11, 6>(src, out, "2 x ds_write_b64"); run<12, 6>(src, out, "ds_write2_b64"); run<13, 6>(src, out, "2 x ds_write2_b32");
    run<14, 6>(src, out, "8 x v_exp_f32"); run<15, 6>(src, out, "8 x v_rcp_f32"); run<16, 6>(src, out, "4 x v_pk_fma_f32");
    run<17, 6>(src, out, "8 x v_add_u32"); run<18, 6>(src, out, "8 x v_min_u32"); run<19, 6>(src, out, "8 x (v_cmp_lt_f32 + v_cndmask_b32)");
    run<8, 6>(src, out, "v_lshl_add_u64"); run<8, 12>(src, out, "v_lshl_add_u64");
    return 0;
}
