// Round-trip floor of two ways to trigger work on the GPU and get one word back (diagnostic, not product):
//   (a) launch a kernel that stores the word into pinned host memory; host spins on it   (what ctk_step does)
//   (b) a RESIDENT kernel polls a pinned host mailbox, answers into pinned host memory    ("next")
//   hipcc -O3 --offload-arch=gfx950 tools/diag_mailbox.hip -o tools/diag_mailbox
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void answer_once(volatile unsigned* ack, unsigned v) { if (threadIdx.x == 0) __hip_atomic_store((unsigned*)ack, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

// exits on cmd == 0xFFFFFFFF or after ~2 s without it (bounded: never outlives the process by more than that)
__global__ void resident(const unsigned* cmd, unsigned* ack, int blocks_answering) {
    const unsigned long long t0 = wall_clock64();
    unsigned last = 0;
    while (true) {
        const unsigned c = __hip_atomic_load(cmd, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (c == 0xFFFFFFFFu) break;
        if (c != last) {
            last = c;
            if ((int)blockIdx.x < blocks_answering && threadIdx.x == 0) __hip_atomic_store(ack + blockIdx.x, c, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (wall_clock64() - t0 > 200000000ull) break;
        __builtin_amdgcn_s_sleep(1);
    }
}

int main() {
    unsigned *cmd, *ack, *cmd_d, *ack_d;
    CK(hipHostMalloc((void**)&cmd, 64, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostMalloc((void**)&ack, 256, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void**)&cmd_d, cmd, 0)); CK(hipHostGetDevicePointer((void**)&ack_d, ack, 0));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int iters = 3000;
    auto stats = [&](std::vector<double>& v, const char* name) {
        std::sort(v.begin(), v.end());
        printf("%-58s median %6.2f us  p10 %6.2f  p90 %6.2f\n", name, v[v.size() / 2], v[v.size() / 10], v[v.size() * 9 / 10]);
    };
    {   // (a)
        std::vector<double> v;
        for (int i = 1; i <= iters; ++i) {
            auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(answer_once, dim3(16), dim3(256), 0, st, ack_d, (unsigned)i);
            while (*(volatile unsigned*)ack != (unsigned)i) __builtin_ia32_pause();
            v.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
        stats(v, "(a) launch 16x256 kernel -> host word");
    }
    for (int blocks : {1, 16}) {   // (b)
        *cmd = 0; for (int i = 0; i < 64; ++i) ack[i] = 0;
        hipLaunchKernelGGL(resident, dim3(16), dim3(64), 0, st, cmd_d, ack_d, blocks);
        std::vector<double> v;
        for (int i = 1; i <= iters; ++i) {
            auto t0 = std::chrono::steady_clock::now();
            __atomic_store_n(cmd, (unsigned)i, __ATOMIC_RELEASE);
            bool all = false;
            while (!all) { all = true; for (int b = 0; b < blocks; ++b) all &= (((volatile unsigned*)ack)[b] == (unsigned)i); }
            v.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
        __atomic_store_n(cmd, 0xFFFFFFFFu, __ATOMIC_RELEASE);
        CK(hipStreamSynchronize(st));
        char nm[96]; snprintf(nm, sizeof nm, "(b) resident kernel, mailbox -> %d block(s) answer", blocks);
        stats(v, nm);
    }
    return 0;
}
