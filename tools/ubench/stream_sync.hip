// Microbenchmark (round 6): what a cross-stream dependency costs on this chip and runtime -- the price of running two stages of
// the loss side by side on two HIP streams (hipEventRecord on one, hipStreamWaitEvent on the other).
//   hipcc --offload-arch=gfx950 -O2 stream_sync.hip -o stream_sync.bin && ./stream_sync.bin
// Cases (us per repetition, median of 20 after warm-up; "busy" kernels spin for a fixed time on every CU):
//   chain      : N small kernels back to back on ONE stream                       -> launch-to-launch gap
//   pingpong   : N small kernels alternating between TWO streams, each waiting for the previous one through an event
//   forkjoin   : A(main) -> fork -> [B(side) beside C(main)] -> join -> D(main), against the same four kernels on one stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

__global__ __launch_bounds__(256) void k_spin(float *out, long long ticks) {
    const long long t0 = wall_clock64();
    float a = threadIdx.x;
    while (wall_clock64() - t0 < ticks) a = fmaf(a, 1.0001f, 0.5f);
    if (a == 12345.678f) out[blockIdx.x] = a;
}

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
    float *d;
    hipMalloc(&d, 1 << 20);
    hipStream_t s0, s1;
    hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipEvent_t ta, tb, e[64];
    hipEventCreate(&ta); hipEventCreate(&tb);
    for (auto &x : e) hipEventCreateWithFlags(&x, hipEventDisableTiming);
    int clk_khz = 100000;                                   // wall_clock64: 100 MHz
    auto us_ticks = [&](double us) { return (long long)(us * clk_khz / 1000.0); };
    auto timeit = [&](auto body) {
        std::vector<double> v;
        for (int r = 0; r < 25; ++r) {
            hipDeviceSynchronize();
            hipEventRecord(ta, s0);
            body();
            hipEventRecord(tb, s0);
            hipEventSynchronize(tb);
            float ms = 0.f; hipEventElapsedTime(&ms, ta, tb);
            if (r >= 5) v.push_back(1e3 * ms);
        }
        return median(v);
    };
    const int N = 16;
    for (int blocks : {1, 256, 2048}) {
        for (double kus : {2.0, 20.0}) {
            const long long tk = us_ticks(kus);
            const double chain = timeit([&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), 0, s0, d, tk); });
            const double pp = timeit([&] {
                for (int i = 0; i < N; ++i) {
                    hipStream_t s = (i & 1) ? s1 : s0, o = (i & 1) ? s0 : s1;
                    hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), 0, s, d, tk);
                    hipEventRecord(e[i], s);
                    hipStreamWaitEvent(o, e[i], 0);
                }
            });
            printf("blocks %5d  kernel %5.1f us : chain %7.2f us/kernel   pingpong %7.2f us/kernel   => a dependency across streams costs %6.2f us\n",
                   blocks, kus, chain / N, pp / N, (pp - chain) / N);
        }
    }
    // fork / join around kernels that do not fill the chip (256 blocks of 256 threads = one workgroup per CU)
    for (double side_us : {20.0, 60.0}) {
        const long long tA = us_ticks(50), tB = us_ticks(side_us), tC = us_ticks(50), tD = us_ticks(10);
        const double serial = timeit([&] {
            hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s0, d, tA);
            hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s0, d, tB);
            hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s0, d, tC);
            hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s0, d, tD);
        });
        const double fj = timeit([&] {
            hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s0, d, tA);
            hipEventRecord(e[0], s0); hipStreamWaitEvent(s1, e[0], 0);
            hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s1, d, tB);
            hipEventRecord(e[1], s1);
            hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s0, d, tC);
            hipStreamWaitEvent(s0, e[1], 0);
            hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s0, d, tD);
        });
        printf("A 50 -> [B %4.0f beside C 50] -> D 10 us : one stream %7.2f us   fork/join %7.2f us   ideal %7.2f us   => fork + join cost %6.2f us\n",
               side_us, serial, fj, 50 + std::max(side_us, 50.0) + 10, fj - (50 + std::max(side_us, 50.0) + 10));
    }
    // the same captured into a graph and replayed (what a HIP graph makes of the cross-stream edges)
    {
        const long long tA = us_ticks(50), tB = us_ticks(20), tC = us_ticks(50), tD = us_ticks(10);
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal);
        hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s0, d, tA);
        hipEventRecord(e[0], s0); hipStreamWaitEvent(s1, e[0], 0);
        hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s1, d, tB);
        hipEventRecord(e[1], s1);
        hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s0, d, tC);
        hipStreamWaitEvent(s0, e[1], 0);
        hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s0, d, tD);
        hipStreamEndCapture(s0, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        const double gr = timeit([&] { hipGraphLaunch(ge, s0); });
        printf("the fork/join case (B 20) as a replayed HIP graph: %7.2f us (ideal 110)\n", gr);
    }
    return 0;
}
