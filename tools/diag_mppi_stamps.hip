// Diagnostic build of the MPPI rollout kernel with s_memtime stamps (cdna_hip_programming.md §7
// "In-kernel stamps").  Reports the SHARES of the kernel's phases; its run time is not a benchmark.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCTK_STAMPS -I control_toolkit_amd/csrc tools/diag_mppi_stamps.hip -o tools/diag_mppi_stamps
#include <algorithm>
#include <cstdio>
#include <vector>
#include "../control_toolkit_amd/csrc/ctk_mppi.hip"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 1024, H = argc > 2 ? atoi(argv[2]) : 50, P = H;
    float params[CTK_P_COUNT] = {9.81f, 0.230f, 0.087f, 0.1975f, 2.62f, 4.77f, 2.5e-4f, 0.f, 1.f, 600.f, 20000.f, 80.f, 1.f, 1.f, 1.f, 0.198f, 0.f};
    EnvK k = derive_constants(params, 0.02f, 1);
    MppiK m{0.2121f, 0.4995f, 1.f, 0.5f, 1.f, -0.01f};
    std::vector<InterpEntry> tab(H);
    for (int t = 0; t < H; ++t) tab[t] = InterpEntry{std::min(t, P - 2), t < H - 1 ? 1.f : 0.f, t < H - 1 ? 0.f : 1.f};
    std::vector<float> noise((size_t)N * P);
    unsigned s = 1; for (auto& v : noise) { s = s * 1664525u + 1013904223u; v = ((s >> 8) * (1.f / 16777216.f) - 0.5f) * 3.f; }
    float *d_noise, *d_unom, *d_J, *d_parts; InterpEntry* d_tab; unsigned long long* d_st;
    const int nb = ctk_mppi_num_blocks(N, CTK_PRED_ODE);
    CK(hipMalloc(&d_noise, noise.size() * 4)); CK(hipMalloc(&d_unom, H * 4)); CK(hipMalloc(&d_J, N * 4));
    CK(hipMalloc(&d_parts, (size_t)nb * (2 + P) * 4)); CK(hipMalloc(&d_tab, H * sizeof(InterpEntry))); CK(hipMalloc(&d_st, nb * 16 * 8)); CK(hipMemset(d_st, 0, nb * 16 * 8));
    CK(hipMemcpy(d_noise, noise.data(), noise.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_unom, 0, H * 4)); CK(hipMemcpy(d_tab, tab.data(), H * sizeof(InterpEntry), hipMemcpyHostToDevice));
    RolloutArgs a{}; a.s0[0] = 0.05f; a.s0[1] = -0.1f; a.s0[2] = 2.8f; a.s0[3] = 0.4f; a.lo[0] = -1; a.hi[0] = 1; a.C = 1; a.N = N; a.H = H; a.P = P;
    a.inv_Hp1 = 1.f / (H + 1); a.p_magic = (uint32_t)((0x100000000ull + P - 1) / P); a.identity_interp = 1; a.interp = d_tab; a.J = d_J; a.stamps = d_st;
    unsigned* d_cnt; float *d_unom2, *d_u, *h_u; CK(hipMalloc(&d_cnt, 4)); CK(hipMemset(d_cnt, 0, 4)); CK(hipMalloc(&d_unom2, H * 4)); CK(hipMalloc(&d_u, 4)); CK(hipHostMalloc(&h_u, 64, hipHostMallocMapped));
    unsigned long long* d_ll; CK(hipMalloc(&d_ll, (size_t)nb * (2 + P) * 8)); CK(hipMemset(d_ll, 0, (size_t)nb * (2 + P) * 8));
    MppiFuse fz; fz.mode = argc > 3 ? atoi(argv[3]) : 1; fz.counter = d_cnt; fz.ll = (argc > 4 && atoi(argv[4]) == 0) ? nullptr : d_ll; fz.u_nom_out = d_unom2; fz.u_dev = d_u; fz.u_host = h_u;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int it = 0; it < 20; ++it) {
        CK(hipEventRecord(e0, 0));
        fz.seq = (uint32_t)(it + 1);
        CK(ctk_launch_mppi_rollout(0, CTK_PRED_ODE, a, k, m, d_noise, d_unom, nullptr, d_parts, false, fz));
        CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<unsigned long long> st(nb * 16);
    CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    const char* names[6] = {"tile load", "inputs (interp/clip/corr)", "recurrence (wave 0)", "softmin partial", "column sums + store", "fused tail (hand-off + merge)"};
    printf("N=%d H=%d blocks=%d  event time %.2f us (stamped build)\n", N, H, nb, ms * 1e3);
    for (int ph = 0; ph < 6; ++ph) {
        std::vector<double> d;
        for (int b = 0; b < nb; ++b) d.push_back((double)(st[b * 16 + ph + 1] - st[b * 16 + ph]));
        std::sort(d.begin(), d.end());
        printf("  %-40s median %8.0f cycles   max %8.0f\n", names[ph], d[d.size() / 2], d.back());
    }
    {   // inside the tile load: 0 -> 8 (pads zeroed, sample loads issued) -> 9 (table loads issued + stored) -> 10 (samples in LDS) -> 1 (barrier)
        const int seq[5] = {0, 8, 9, 10, 1};
        const char* nm[4] = {"  issue sample loads", "  table loads + LDS stores (early)", "  wait samples + LDS stores", "  barrier"};
        for (int i = 0; i < 4; ++i) {
            std::vector<double> d;
            for (int b = 0; b < nb; ++b) d.push_back((double)(st[b * 16 + seq[i + 1]] - st[b * 16 + seq[i]]));
            std::sort(d.begin(), d.end());
            printf("    %-38s median %8.0f cycles\n", nm[i], d[d.size() / 2]);
        }
    }
    unsigned long long mn = ~0ull, mx = 0;
    for (int b = 0; b < nb; ++b) { mn = std::min(mn, st[b * 16]); mx = std::max(mx, st[b * 16 + 6]); }
    printf("  first block start -> last block end: %llu cycles\n", mx - mn);
    return 0;
}
