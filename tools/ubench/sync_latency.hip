// How long after a kernel's last store does the host know?  (a) hipStreamSynchronize, (b) hipEventSynchronize on an event
// recorded behind the kernel, (c) polling a sequence flag the kernel stores into mapped pinned memory after a system-scope
// fence.  The kernel spins ~20 us first so that the host is already waiting when it ends (the frame-state phases' case).
// build: hipcc -O2 --offload-arch=gfx950 -o sync_latency sync_latency.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void work(volatile unsigned* flag, unsigned seq, long long spin, unsigned long long* t_end) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if (threadIdx.x == 0) {
        *t_end = (unsigned long long)wall_clock64();
        __threadfence_system();
        *flag = seq;
    }
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned* h; CK(hipHostMalloc((void**)&h, 4096, hipHostMallocMapped));
    unsigned* d; CK(hipHostGetDevicePointer((void**)&d, h, 0));
    unsigned long long* d_t; CK(hipMalloc((void**)&d_t, 8));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    h[0] = 0;
    const long long spin = 2000;                       // wall_clock64 ticks at 100 MHz: 20 us
    const int N = 200;
    for (int mode = 0; mode < 4; ++mode) {
        std::vector<double> tot, rec;
        for (int i = 0; i < N + 20; ++i) {
            const unsigned seq = (unsigned)(mode * 1000 + i + 1);
            const double t0 = now();
            hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, st, (volatile unsigned*)d, seq, spin, d_t);
            double t1 = now();
            if (mode == 1) { CK(hipEventRecord(ev, st)); }
            const double t2 = now();
            if (mode == 0) CK(hipStreamSynchronize(st));
            else if (mode == 1) CK(hipEventSynchronize(ev));
            else if (mode == 2) { while (*(volatile unsigned*)h != seq) {} }
            else { while (*(volatile unsigned*)h != seq) { __builtin_ia32_pause(); } }
            const double t3 = now();
            if (mode >= 2) CK(hipStreamSynchronize(st));      // (outside the measurement: keeps the queue short)
            if (i >= 20) { tot.push_back(t3 - t0); rec.push_back(t2 - t1); }
            (void)t1;
        }
        std::sort(tot.begin(), tot.end()); std::sort(rec.begin(), rec.end());
        const char* names[] = {"hipStreamSynchronize", "hipEventRecord + hipEventSynchronize", "poll mapped flag", "poll mapped flag (pause)"};
        printf("%-40s launch -> host knows: median %.1f us, p10 %.1f, p90 %.1f   (kernel body 20 us; event record call %.1f us)\n",
               names[mode], tot[N / 2], tot[N / 10], tot[N * 9 / 10], rec[N / 2]);
    }
    return 0;
}
