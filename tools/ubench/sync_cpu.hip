// What does waiting for the GPU cost in CPU time?  A kernel of ~5 ms, waited for by hipStreamSynchronize, by hipEventSynchronize on a plain event and by
// hipEventSynchronize on an event made with hipEventBlockingSync; and the extra latency of each on a tiny kernel.  (A rank whose launch thread spins burns one
// core for the whole step: eight ranks on a 16-CPU quota cannot afford that.)
// Build: hipcc --offload-arch=gfx950 -O3 -o sync_cpu sync_cpu.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <ctime>
__global__ void k_spin(unsigned long long ticks, int *p) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
    if (p && threadIdx.x == 999) *p = 1;
}
using clk = std::chrono::steady_clock;
static double cpu_ms() { timespec ts; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6; }
// argv[1]: auto | spin | yield | blocking -> hipSetDeviceFlags(hipDeviceSchedule*) before the first other HIP call of the process (r5: the DEVICE flag is what decides how
// this runtime waits -- the event flag and ROC_ACTIVE_WAIT_TIMEOUT are not)
int main(int argc, char **argv) {
    if (argc > 1) {
        const char *m = argv[1];
        const unsigned f = !strcmp(m, "spin") ? hipDeviceScheduleSpin : !strcmp(m, "yield") ? hipDeviceScheduleYield : !strcmp(m, "blocking") ? hipDeviceScheduleBlockingSync : hipDeviceScheduleAuto;
        const hipError_t e = hipSetDeviceFlags(f);
        printf("--- hipSetDeviceFlags(%s) -> %s\n", m, hipGetErrorString(e));
    }
    hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t ev, evb;
    (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&evb, hipEventDisableTiming | hipEventBlockingSync);
    const char *names[3] = {"hipStreamSynchronize", "hipEventSynchronize (plain event)", "hipEventSynchronize (hipEventBlockingSync)"};
    for (int len = 0; len < 2; len++) {
        const unsigned long long ticks = len ? 500000ull : 0ull;      // 100 MHz counter: 5 ms, or nothing
        const int N = len ? 40 : 2000;
        for (int mode = 0; mode < 3; mode++) {
            for (int w = 0; w < 5; w++) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, 0ull, (int *)nullptr); (void)hipStreamSynchronize(st); }
            const auto t0 = clk::now();
            const double c0 = cpu_ms();
            for (int i = 0; i < N; i++) {
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, ticks, (int *)nullptr);
                if (mode == 0) (void)hipStreamSynchronize(st);
                else if (mode == 1) { (void)hipEventRecord(ev, st); (void)hipEventSynchronize(ev); }
                else { (void)hipEventRecord(evb, st); (void)hipEventSynchronize(evb); }
            }
            const double wall = std::chrono::duration<double, std::milli>(clk::now() - t0).count(), cpu = cpu_ms() - c0;
            printf("%-44s kernel of %s: %8.3f ms wall, %8.3f ms CPU per launch + wait\n", names[mode], len ? "5 ms" : "nothing", wall / N, cpu / N);
        }
    }
    return 0;
}
