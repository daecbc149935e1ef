// diag_mfma_coexec.hip — does INDEPENDENT vector work hide under fp32 MFMAs on gfx950?
//
// Round 2 measured "a wave's matrix and vector time add up" with tanh that DEPENDS on the MFMAs of the same 16-trajectory tile
// (tools/diag_mlp_step.hip).  MI355X_MICROARCH.md's constants table says an MFMA holds the SIMD's vector issue for 8 of its
// 32 cycles and up to 24 cycles of independent VALU issue hide per gap (measured there on v_mfma_f32_32x32x16_bf16).  This
// tool measures the same for v_mfma_f32_16x16x4_f32, the instruction of ctk_mlp.h:
//   part 1 (synthetic): a stream of MFMAs on two alternating accumulators with K independent fillers hand-placed behind each
//           (v_fma_f32 = 4 issue cycles, v_exp_f32 = 8), in-kernel s_memtime cycles per MFMA, one and two waves per SIMD;
//   part 2 (the network step): ctk_mlp.h's mlp_step for ONE tile per wave vs TWO independent tiles per wave, the second
//           tile's program shifted by half a step so that its tanh / reduce / cost work falls into the first tile's layer-2
//           MFMAs (and vice versa), placed with sched_group_barrier.  ns per tile-step at 1, 2, 4 tiles per SIMD.
// Counters: run under `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU` (own pass).
// build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -I control_toolkit_amd/csrc -o tools/diag_mfma_coexec tools/diag_mfma_coexec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "ctk_mlp.h"

#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)
constexpr int M_MFMA = 0x008, M_VALU = 0x002, M_TRANS = 0x400;

// ------------------------------------------------------------------------------------------------------------------
// part 1: MFMA stream + K_FMA independent v_fma and K_EXP independent v_exp behind every MFMA
// ------------------------------------------------------------------------------------------------------------------
template <int K_FMA, int K_EXP>
__global__ __launch_bounds__(256) void k_gap(float* out, int iters, unsigned long long* cyc) {
    f32x4 c0 = f32x4{0.f, 0.f, 0.f, 0.f}, c1 = c0;
    const float a = 1e-3f * (float)(threadIdx.x & 15), b = 0.25f;
    float v[8], e[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.001f * (float)(threadIdx.x + i);
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = 0.002f * (float)(threadIdx.x + i);
    const float p = 0.999f, q = 1e-4f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    // everything inside the loop is volatile asm: program order = issue order, nothing packed or moved by the compiler
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j & 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
#pragma unroll
            for (int f = 0; f < K_EXP; ++f) { const int r = (j * K_EXP + f) & 3; asm volatile("v_exp_f32 %0, %0" : "+v"(e[r])); }
#pragma unroll
            for (int f = 0; f < K_FMA; ++f) { const int r = (j * K_FMA + f) & 7; asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(p), "v"(q)); }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the MFMA results are read below (the compiler cannot see the asm's hazards)
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = c0[0] + c1[1];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += e[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// control experiment: the SAME harness on v_mfma_f32_16x16x32_bf16 (8 passes = 16 cycles; the guide's co-execution numbers are
// for the bf16 forms) — shows whether the method can see co-execution where it exists
typedef short bf16x8 __attribute__((ext_vector_type(8)));
template <int K_FMA, int K_EXP>
__global__ __launch_bounds__(256) void k_gap_bf16(float* out, int iters, unsigned long long* cyc) {
    f32x4 c0 = f32x4{0.f, 0.f, 0.f, 0.f}, c1 = c0;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3c00 + (threadIdx.x & 15)); b[i] = (short)0x3e80; }
    float v[8], e[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.001f * (float)(threadIdx.x + i);
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = 0.002f * (float)(threadIdx.x + i);
    const float p = 0.999f, q = 1e-4f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j & 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
#pragma unroll
            for (int f = 0; f < K_EXP; ++f) { const int r = (j * K_EXP + f) & 3; asm volatile("v_exp_f32 %0, %0" : "+v"(e[r])); }
#pragma unroll
            for (int f = 0; f < K_FMA; ++f) { const int r = (j * K_FMA + f) & 7; asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(p), "v"(q)); }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = c0[0] + c1[1];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += e[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// ------------------------------------------------------------------------------------------------------------------
// part 2: the network step.  TILES independent tiles per wave.
//   TILES = 1: mlp_step as the product runs it.
//   TILES = 2, PIPE = false: two calls, the compiler's schedule.
//   TILES = 2, PIPE = true : tile B runs half a step behind tile A (software pipeline, see mlp_step_x2 below).
// ------------------------------------------------------------------------------------------------------------------
// scalar tanh (no v_pk_*: a packed fp32 op beside MFMAs is an anti-lever, MI355X_MICROARCH.md constants table)
CTK_DEV float tanh_s(float x) {
    const float t = __builtin_amdgcn_exp2f(x * 2.885390081777927f);
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(t + 1.0f), 1.0f);
}
CTK_DEV f32x4 tanh4_s(f32x4 x) { return f32x4{tanh_s(x[0]), tanh_s(x[1]), tanh_s(x[2]), tanh_s(x[3])}; }

struct HalfA { f32x4 h1[2]; };                 // after layer 1 + tanh
struct HalfB { f32x4 c[2]; };                  // layer-2 accumulators (before tanh)

CTK_DEV HalfA step_front(const MlpFwdT& w, float sv, float u) {       // layer 1 (2 MFMA) + 8 tanh
    f32x4 a0, a1;
#pragma unroll
    for (int r = 0; r < 4; ++r) { a0[r] = fmaf(w.w1u[0][r], u, w.b1[0][r]); a1[r] = fmaf(w.w1u[1][r], u, w.b1[1][r]); }
    a0 = CTK_MFMA(w.w1s[0], sv, a0);
    a1 = CTK_MFMA(w.w1s[1], sv, a1);
    HalfA x; x.h1[0] = tanh4_s(a0); x.h1[1] = tanh4_s(a1);
    return x;
}
CTK_DEV HalfB step_mid(const MlpFwdT& w, const HalfA& x) {            // layer 2: 16 MFMA
    HalfB y; y.c[0] = w.b2[0]; y.c[1] = w.b2[1];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float b = x.h1[j >> 2][j & 3];
        y.c[0] = CTK_MFMA(w.w2[0][j], b, y.c[0]);
        y.c[1] = CTK_MFMA(w.w2[1][j], b, y.c[1]);
    }
    return y;
}
CTK_DEV float step_back(const MlpFwdT& w, const HalfB& y) {           // 8 tanh + layer 3 (8 small MFMA) + reduce-scatter
    f32x4 h2[2] = {tanh4_s(y.c[0]), tanh4_s(y.c[1])};
    f32x4 p0 = f32x4{0.f, 0.f, 0.f, 0.f}, p1 = p0;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        p0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w.w3n[j], h2[j >> 2][j & 3], p0, 0, 0, 0);
        p1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w.w3n[j + 1], h2[(j + 1) >> 2][(j + 1) & 3], p1, 0, 0, 0);
    }
    const f32x4 p = p0 + p1;
    const float s02 = swap_sum32(p[0], p[2]), s13 = swap_sum32(p[1], p[3]);
    return swap_sum16(s02, s13) + w.b3g;
}

// MODE 0: one tile (mlp_step of the product, packed tanh)   1: one tile, scalar tanh, front/mid/back
//      2: two tiles, two product calls per step             3: two tiles, pipelined: [A.mid || B.back, B.front] [B.mid || A.back, A.front]
//      4: as 3 with the interleave pinned by sched_group_barrier (1 MFMA : 3 VALU of which <= 1 transcendental)
template <int MODE>
__global__ __launch_bounds__(256) void k_step(const float* wperm, float* out, int H, unsigned long long* cyc) {
    const MlpFwdT w = mlp_load_fwd_thin(wperm);
    const int lane = threadIdx.x & 63, g = lane >> 4;
    float svA = 0.01f * (float)(lane & 15) + 0.1f * g, svB = svA + 0.37f;
    const MlpCostK ck = mlp_cost_coeffs(EnvK{}, g, false);
    float costA = 0.f, costB = 0.f;
    const float u = 0.1f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if constexpr (MODE == 0) {
        for (int h = 0; h < H; ++h) { costA += ctk_cosf_fast(svA); svA = mlp_step(w, svA, u, g); }
    } else if constexpr (MODE == 1) {
        for (int h = 0; h < H; ++h) { costA += ctk_cosf_fast(svA); svA = step_back(w, step_mid(w, step_front(w, svA, u))); }
    } else if constexpr (MODE == 2) {
        for (int h = 0; h < H; ++h) {
            costA += ctk_cosf_fast(svA); costB += ctk_cosf_fast(svB);
            svA = mlp_step(w, svA, u, g); svB = mlp_step(w, svB, u, g);
        }
    } else {
        // prologue: A's front; B's front + mid issued so that B is "half a step behind" in the loop
        HalfA fa = step_front(w, svA, u);
        costA += ctk_cosf_fast(svA);
        HalfA fb = step_front(w, svB, u);
        costB += ctk_cosf_fast(svB);
        HalfB mb = step_mid(w, fb);
        for (int h = 0; h < H; ++h) {
            // phase 1: A's layer 2 (16 MFMA)  ||  B: tanh, layer 3, reduce, cost, layer 1, tanh
            HalfB ma = step_mid(w, fa);
            svB = step_back(w, mb);
            costB += ctk_cosf_fast(svB);
            fb = step_front(w, svB, u);
            if constexpr (MODE == 4) {
#pragma unroll
                for (int j = 0; j < 16; ++j) { SGB(M_MFMA, 1); SGB(M_TRANS, 1); SGB(M_VALU, 3); }
            }
            __builtin_amdgcn_sched_barrier(0);
            // phase 2: B's layer 2  ||  A: tanh, layer 3, reduce, cost, layer 1, tanh
            mb = step_mid(w, fb);
            svA = step_back(w, ma);
            costA += ctk_cosf_fast(svA);
            fa = step_front(w, svA, u);
            if constexpr (MODE == 4) {
#pragma unroll
                for (int j = 0; j < 16; ++j) { SGB(M_MFMA, 1); SGB(M_TRANS, 1); SGB(M_VALU, 3); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        svB += mb.c[0][0];
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[blockIdx.x * 256 + threadIdx.x] = svA + svB + costA + costB + ck.A;
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <class F>
float time_ms(F&& launch, int reps = 20) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return ms / reps;
}

static unsigned long long median_cyc(unsigned long long* d, int n) {
    std::vector<unsigned long long> h(n);
    hipMemcpy(h.data(), d, n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    return h[n / 2];
}

template <int KF, int KE>
void run_gap(float* dout, unsigned long long* dcyc, int blocks) {
    const int iters = 2000;
    const float ms = time_ms([&] { hipLaunchKernelGGL((k_gap<KF, KE>), dim3(blocks), dim3(256), 0, 0, dout, iters, dcyc); }, 5);
    const double cyc = (double)median_cyc(dcyc, blocks * 4) / (iters * 16.0);
    printf("  fillers/MFMA: %d v_fma + %d v_exp (issue cost %2d cyc)  blocks %4d: %6.1f shader cycles per MFMA   %6.2f ns per MFMA (wall)\n", KF, KE,
           4 * KF + 8 * KE, blocks, cyc, ms * 1e6 / (iters * 16.0));
}

template <int KF, int KE>
void run_gap_bf16(float* dout, unsigned long long* dcyc, int blocks) {
    const int iters = 2000;
    const float ms = time_ms([&] { hipLaunchKernelGGL((k_gap_bf16<KF, KE>), dim3(blocks), dim3(256), 0, 0, dout, iters, dcyc); }, 5);
    const double cyc = (double)median_cyc(dcyc, blocks * 4) / (iters * 16.0);
    printf("  bf16 16x16x32, fillers/MFMA: %d v_fma + %d v_exp (issue cost %2d cyc)  blocks %4d: %6.1f shader cycles per MFMA   %6.2f ns per MFMA (wall)\n", KF, KE,
           4 * KF + 8 * KE, blocks, cyc, ms * 1e6 / (iters * 16.0));
}

template <int MODE>
void run_step(const char* name, int tiles_per_wave, const float* dw, float* dout, unsigned long long* dcyc, int blocks, int H) {
    const float ms = time_ms([&] { hipLaunchKernelGGL(k_step<MODE>, dim3(blocks), dim3(256), 0, 0, dw, dout, H, dcyc); });
    const double cyc = (double)median_cyc(dcyc, blocks * 4) / H / tiles_per_wave;
    const double tiles_per_simd = (double)blocks * 4 * tiles_per_wave / 1024.0;
    // a SIMD's time per tile-step: wall / H / tiles it owns
    printf("  %-34s blocks %4d (%4.1f tiles/SIMD): %7.1f ns per step of a wave, %6.1f ns of SIMD time per tile-step, %6.0f shader cycles per tile-step of one wave\n",
           name, blocks, tiles_per_simd, ms * 1e6 / H, ms * 1e6 / H / std::max(1.0, tiles_per_simd), cyc);
}

int main(int argc, char** argv) {
    const int part = argc > 1 ? atoi(argv[1]) : 0;   // 0 both, 1 synthetic only, 2 step only
    std::vector<float> w(64 * (MLP_FWD_PER_LANE + MLP_BWD_PER_LANE));
    for (size_t i = 0; i < w.size(); ++i) w[i] = 0.05f * (float)((i * 7919) % 13 - 6);
    float *dw, *dout; unsigned long long* dcyc;
    hipMalloc(&dw, w.size() * 4); hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&dout, 4096 * 256 * 4); hipMalloc(&dcyc, 4096 * 4 * 8);
    if (part == 0 || part == 1) {
        printf("part 1: v_mfma_f32_16x16x4_f32 stream, independent fillers behind each MFMA (256 blocks = one wave per SIMD, 512 = two)\n");
        for (int blocks : {256, 512}) {
            run_gap<0, 0>(dout, dcyc, blocks);
            run_gap<2, 0>(dout, dcyc, blocks);
            run_gap<4, 0>(dout, dcyc, blocks);
            run_gap<5, 0>(dout, dcyc, blocks);
            run_gap<6, 0>(dout, dcyc, blocks);
            run_gap<8, 0>(dout, dcyc, blocks);
            run_gap<0, 1>(dout, dcyc, blocks);
            run_gap<0, 2>(dout, dcyc, blocks);
            run_gap<0, 3>(dout, dcyc, blocks);
            run_gap<2, 1>(dout, dcyc, blocks);
            run_gap<4, 1>(dout, dcyc, blocks);
            run_gap<2, 2>(dout, dcyc, blocks);
            run_gap<3, 2>(dout, dcyc, blocks);
        }
    }
    if (part == 0 || part == 3) {
        printf("control: v_mfma_f32_16x16x32_bf16 stream in the same harness\n");
        for (int blocks : {256, 512}) {
            run_gap_bf16<0, 0>(dout, dcyc, blocks);
            run_gap_bf16<1, 0>(dout, dcyc, blocks);
            run_gap_bf16<2, 0>(dout, dcyc, blocks);
            run_gap_bf16<3, 0>(dout, dcyc, blocks);
            run_gap_bf16<4, 0>(dout, dcyc, blocks);
            run_gap_bf16<6, 0>(dout, dcyc, blocks);
            run_gap_bf16<0, 1>(dout, dcyc, blocks);
            run_gap_bf16<2, 1>(dout, dcyc, blocks);
            run_gap_bf16<0, 2>(dout, dcyc, blocks);
        }
    }
    if (part == 0 || part == 2) {
        const int H = 100;
        printf("part 2: the 5-32-32-4 network step + cos (H = %d)\n", H);
        for (int blocks : {256, 512, 1024}) {
            run_step<0>("1 tile/wave, product mlp_step", 1, dw, dout, dcyc, blocks, H);
            run_step<1>("1 tile/wave, scalar tanh", 1, dw, dout, dcyc, blocks, H);
        }
        for (int blocks : {128, 256, 512}) {
            run_step<2>("2 tiles/wave, two calls", 2, dw, dout, dcyc, blocks, H);
            run_step<3>("2 tiles/wave, half-step shifted", 2, dw, dout, dcyc, blocks, H);
            run_step<4>("2 tiles/wave, shifted + pinned 1:4", 2, dw, dout, dcyc, blocks, H);
        }
    }
    return 0;
}
