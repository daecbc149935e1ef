// Diagnostic build of the one-launch CEM step with wall-clock stamps at its phase boundaries (100 MHz ticks -> ns).
// Reports where an outer iteration's time goes; its run time is not a benchmark.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCTK_CEM_STAMPS -mllvm -amdgpu-kernarg-preload-count=14 -I control_toolkit_amd/csrc tools/diag_cem_fused.hip -o tools/diag_cem_fused
#include <algorithm>
#include <cstdio>
#include <vector>
#include "../control_toolkit_amd/csrc/ctk_cem_fused.hip"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 4096, H = argc > 2 ? atoi(argv[2]) : 30, K = argc > 3 ? atoi(argv[3]) : 409, its = 3;
    float params[CTK_P_COUNT] = {9.81f, 0.230f, 0.087f, 0.1975f, 2.62f, 4.77f, 2.5e-4f, 0.f, 1.f, 600.f, 20000.f, 80.f, 1.f, 1.f, 1.f, 0.198f, 0.f};
    std::vector<float> noise((size_t)its * N * H);
    unsigned s = 1; for (auto& v : noise) { s = s * 1664525u + 1013904223u; v = ((s >> 8) * (1.f / 16777216.f) - 0.5f) * 3.f; }
    const int nb = ctk_cem_fused_blocks(N);
    float *d_noise, *d_mu, *d_sd, *d_J, *d_Q, *d_u, *h_u; int* d_idx; unsigned long long *d_ll, *d_st;
    CK(hipMalloc(&d_noise, noise.size() * 4)); CK(hipMemcpy(d_noise, noise.data(), noise.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_mu, H * 4)); CK(hipMalloc(&d_sd, H * 4)); CK(hipMalloc(&d_J, N * 4)); CK(hipMalloc(&d_Q, (size_t)N * H * 4)); CK(hipMalloc(&d_u, 16)); CK(hipMalloc(&d_idx, N * 4));
    CK(hipHostMalloc(&h_u, 64, hipHostMallocMapped));
    const size_t llw = ctk_cem_fused_ll_words(N, H);
    CK(hipMalloc(&d_ll, llw * 8)); CK(hipMemset(d_ll, 0, llw * 8));
    CK(hipMalloc(&d_st, (size_t)nb * 8 * 16 * 8)); CK(hipMemset(d_st, 0, (size_t)nb * 8 * 16 * 8));
    std::vector<float> mu(H, 0.f), sd(H, 0.5f);
    RolloutArgs a{}; a.s0[0] = 0.05f; a.s0[1] = -0.1f; a.s0[2] = 2.8f; a.s0[3] = 0.4f; a.lo[0] = -1; a.hi[0] = 1; a.C = 1; a.N = N; a.H = H; a.P = H;
    a.inv_Hp1 = 1.f / (H + 1); a.p_magic = (uint32_t)((0x100000000ull + H - 1) / H); a.identity_interp = 1; a.J = d_J; a.Q_out = d_Q; a.stamps = d_st;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    uint32_t tag = 1;
    for (int rep = 0; rep < 20; ++rep) {
        CK(hipMemcpy(d_mu, mu.data(), H * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_sd, sd.data(), H * 4, hipMemcpyHostToDevice));
        CemFusedLaunch c{its, K, d_ll, tag, 0.01f, 1e8f, 0.5f, d_mu, d_sd, d_u, h_u, d_idx, (uint32_t)(rep + 1), 0.5};
        tag += its;
        CK(hipEventRecord(e0, 0));
        CK(ctk_launch_cem_fused(0, CTK_ENV_CARTPOLE, params, 0.02f, 1, a, d_noise, c, false));
        CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<unsigned long long> st((size_t)nb * 8 * 16);
    CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    printf("N=%d H=%d K=%d blocks=%d its=%d: event time %.2f us (stamped build); err word %u\n", N, H, K, nb, its, ms * 1e3, reinterpret_cast<unsigned*>(h_u)[2]);
    const char* names[8] = {"prepare first 16 steps (+barrier)", "recurrence -> J (thread 0 = wave 0)", "publish J + gather all N costs", "range reduce + radix select",
                            "ties + elite flags", "local moments + publish", "gather all records", "refit from the records     "};
    for (int it = 0; it < its; ++it) {
        printf(" iteration %d (ns, median over workgroups | max)\n", it);
        for (int ph = 0; ph < 8; ++ph) {
            std::vector<double> d;
            for (int b = 0; b < nb; ++b) d.push_back(10.0 * (double)(st[((size_t)b * 8 + it) * 16 + ph + 1] - st[((size_t)b * 8 + it) * 16 + ph]));
            std::sort(d.begin(), d.end());
            printf("   %-40s %8.0f | %8.0f\n", names[ph], d[d.size() / 2], d.back());
        }
        {   // inside the selection: 3 -> 9 (range reduce + first histogram) -> 10 (first scan) -> 11 (second pass) -> 4 (rest)
            const int seq[5] = {3, 9, 10, 11, 4};
            const char* nm[4] = {"range reduce + histogram of pass 0", "scan of pass 0", "pass 1", "remaining passes"};
            for (int i = 0; i < 4; ++i) {
                std::vector<double> d;
                for (int b = 0; b < nb; ++b) d.push_back(10.0 * (double)(long long)(st[((size_t)b * 8 + it) * 16 + seq[i + 1]] - st[((size_t)b * 8 + it) * 16 + seq[i]]));
                std::sort(d.begin(), d.end());
                printf("       %-36s %8.0f\n", nm[i], d[d.size() / 2]);
            }
        }
    }
    unsigned long long mn = ~0ull, mx = 0;
    for (int b = 0; b < nb; ++b) { mn = std::min(mn, st[((size_t)b * 8) * 16]); mx = std::max(mx, st[((size_t)b * 8 + its - 1) * 16 + 8]); }
    printf(" first workgroup's first stamp -> last workgroup's last stamp: %.0f ns\n", 10.0 * (double)(mx - mn));
    return 0;
}
