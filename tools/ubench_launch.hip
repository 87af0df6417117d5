// What one launch + drain costs on the box, by how the host waits: the element-wise rows of the latency table (add,
// multiply_plain: one kernel of a few microseconds) are this number.  Forms: hipStreamSynchronize on the null stream and
// on a created stream, with the default device flags and with hipDeviceScheduleSpin / hipDeviceScheduleYield /
// hipDeviceScheduleBlockingSync (argv[1] = default | spin | yield | block), and a host poll on hipStreamQuery.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_launch.hip -o tools/ubench_launch
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
using clk = std::chrono::steady_clock;

__global__ void touch(unsigned long long *p, unsigned n)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        p[i] += 1;
}

static double median(std::vector<double> v)
{
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main(int argc, char **argv)
{
    const char *mode = argc > 1 ? argv[1] : "default";
    unsigned flags = hipDeviceScheduleAuto;
    if (!strcmp(mode, "spin"))
        flags = hipDeviceScheduleSpin;
    else if (!strcmp(mode, "yield"))
        flags = hipDeviceScheduleYield;
    else if (!strcmp(mode, "block"))
        flags = hipDeviceScheduleBlockingSync;
    if (strcmp(mode, "default"))
        CK(hipSetDeviceFlags(flags));
    const unsigned n = 15 * 32768 * 2; // one N = 32768, 15-limb ciphertext
    unsigned long long *d;
    CK(hipMalloc(&d, (size_t)n * 8));
    CK(hipMemset(d, 0, (size_t)n * 8));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    auto run = [&](const char *name, hipStream_t st, int how) {
        std::vector<double> t;
        for (int it = 0; it < 220; ++it)
        {
            CK(hipStreamSynchronize(st));
            const auto t0 = clk::now();
            touch<<<n / 256, 256, 0, st>>>(d, n);
            if (how == 0)
                CK(hipStreamSynchronize(st));
            else if (how == 1)
            {
                CK(hipEventRecord(ev, st));
                CK(hipEventSynchronize(ev));
            }
            else
                while (hipStreamQuery(st) == hipErrorNotReady)
                    ;
            const double us = std::chrono::duration<double, std::micro>(clk::now() - t0).count();
            if (it >= 20)
                t.push_back(us);
        }
        printf("%-8s %-44s median %6.1f us\n", mode, name, median(t));
    };
    run("null stream, hipStreamSynchronize", nullptr, 0);
    run("created stream, hipStreamSynchronize", s, 0);
    run("created stream, event record + synchronize", s, 1);
    run("created stream, hipStreamQuery poll", s, 2);
    run("null stream, hipStreamQuery poll", nullptr, 2);
    // three launches back to back, one drain (the `add` row of the table is three adds in one timed region)
    {
        std::vector<double> t;
        for (int it = 0; it < 220; ++it)
        {
            CK(hipStreamSynchronize(s));
            const auto t0 = clk::now();
            for (int k = 0; k < 3; ++k)
                touch<<<n / 256, 256, 0, s>>>(d, n);
            CK(hipStreamSynchronize(s));
            const double us = std::chrono::duration<double, std::micro>(clk::now() - t0).count();
            if (it >= 20)
                t.push_back(us);
        }
        printf("%-8s %-44s median %6.1f us\n", mode, "created stream, 3 launches + one drain", median(t));
    }
    CK(hipFree(d));
    return 0;
}
