// Where does a CU put the waves of small workgroups?  (the two-tile form of the template MLP rollout.)
// A launch of G workgroups of W waves, each asking for L KiB of LDS, all resident at once (every wave spins ~30 us); every wave records its
// XCC / SE / CU / SIMD (s_getreg HW_ID, XCC_ID).  Per CU: how many of the launch's waves each SIMD holds.  A dependent fp32 MFMA chain per wave
// (the shape of a network rollout step) is timed with and without a SIMD neighbour.
//   hipcc -O3 --offload-arch=gfx950 tools/diag_wave_placement.hip -o tools/diag_wave_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Rec { unsigned hw, xcc; unsigned long long cycles; };

__global__ void place(Rec* out, int steps) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float a = 1e-3f * lane, b = 1.0f + 1e-3f * wave;
    const unsigned long long t0 = wall_clock64();
    for (int s = 0; s < steps; ++s) {                               // 14 dependent MFMAs + a little vector work: one network step
#pragma unroll
        for (int j = 0; j < 14; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        acc[0] = __expf(-acc[1] * 1e-9f);
    }
    const unsigned long long t1 = wall_clock64();
    if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + wave] = Rec{hw, xcc & 15, t1 - t0};
    if (acc[0] == 123.456f) lds[0] = acc[2];
}

static int run(int G, int W, int lds_kib, int steps) {
    Rec *d, *h = new Rec[G * W];
    CK(hipMalloc((void**)&d, sizeof(Rec) * G * W));
    hipLaunchKernelGGL(place, dim3(G), dim3(64 * W), (size_t)lds_kib * 1024, 0, d, steps);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, d, sizeof(Rec) * G * W, hipMemcpyDeviceToHost));
    std::map<unsigned, std::vector<int>> cu;                       // (xcc, se, sh, cu) -> waves per SIMD
    for (int i = 0; i < G * W; ++i) {
        const unsigned key = (h[i].xcc << 16) | (((h[i].hw >> 13) & 7) << 8) | (((h[i].hw >> 12) & 1) << 4) | ((h[i].hw >> 8) & 15);
        auto& v = cu[key];
        if (v.empty()) v.assign(4, 0);
        v[(h[i].hw >> 4) & 3]++;
    }
    int hist[5][9] = {};                                            // [idle SIMDs on the CU][waves on the fullest SIMD]
    for (auto& kv : cu) {
        int idle = 0, mx = 0;
        for (int s = 0; s < 4; ++s) { idle += kv.second[s] == 0; mx = kv.second[s] > mx ? kv.second[s] : mx; }
        hist[idle][mx < 8 ? mx : 8]++;
    }
    double alone = 0, shared = 0; int na = 0, ns = 0;
    for (int i = 0; i < G * W; ++i) {
        const unsigned key = (h[i].xcc << 16) | (((h[i].hw >> 13) & 7) << 8) | (((h[i].hw >> 12) & 1) << 4) | ((h[i].hw >> 8) & 15);
        const bool sh = cu[key][(h[i].hw >> 4) & 3] > 1;
        (sh ? shared : alone) += (double)h[i].cycles; (sh ? ns : na)++;
    }
    printf("%4d workgroups x %d waves, %3d KiB LDS each: %3zu CUs used;", G, W, lds_kib, cu.size());
    for (int idle = 0; idle < 5; ++idle)
        for (int mx = 0; mx < 9; ++mx)
            if (hist[idle][mx]) printf("  %d CUs [idle SIMDs %d, fullest SIMD %d waves]", hist[idle][mx], idle, mx);
    printf("\n      chain time per step: wave alone on its SIMD %.0f ns (%d waves)", na ? alone / na * 10.0 / steps : 0.0, na);
    printf(", sharing its SIMD %.0f ns (%d waves)\n", ns ? shared / ns * 10.0 / steps : 0.0, ns);
    delete[] h; hipFree(d);
    return 0;
}

int main() {
    const int steps = 60;
    if (run(256, 2, 60, steps)) return 1;      // one 2-wave workgroup per CU
    if (run(512, 2, 60, steps)) return 1;      // two 2-wave workgroups per CU: the template MLP rollout at N = 8192 before round 3
    if (run(256, 4, 120, steps)) return 1;     // the same waves as one 4-wave workgroup per CU: what it runs now
    if (run(1024, 1, 30, steps)) return 1;     // four 1-wave workgroups per CU
    if (run(4096, 1, 8, steps)) return 1;      // sixteen 1-wave workgroups per CU (the streaming MPPI kernel's regime)
    return 0;
}
