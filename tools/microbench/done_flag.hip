// How much sooner does the host learn that a kernel's results are in pinned memory from a flag the LAST workgroup writes (after a
// system-scope fence) than from hipStreamSynchronize?  (awpu_hip_process's one-frame call waits ~9 us beyond its kernels' own time.)
// 157 workgroups x 1024 threads, each spins ~20 us, stores 64 floats to pinned memory (system scope), waits for their acknowledgement, counts itself; the last one stores the flag.
//   hipcc --offload-arch=gfx950 -O2 done_flag.hip -o done_flag
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void work(float *out, unsigned long long *counter, volatile unsigned *flag, unsigned seq, long spin_clocks) {
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin_clocks) {}
    // system-scope stores (written through), their acknowledgement, the workgroup's count, the last one's flag: no __threadfence_system()
    // (which writes back and invalidates the whole L2 on this chip)
    if (threadIdx.x < 64) __hip_atomic_store(&out[blockIdx.x * 64 + threadIdx.x], (float) seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && (__hip_atomic_fetch_add(counter, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1) % gridDim.x == 0)
        __hip_atomic_store((unsigned *) flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// the pair the one-frame host call enqueues: a small upload (pinned -> device, 75 KB) and the sweep behind it; `work2` takes its sequence
// number from a pinned mailbox, so that the pair can be replayed as a captured graph with unchanged arguments
__global__ void upload(const float4 *src, float4 *dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
__global__ void work2(float *out, unsigned long long *counter, volatile unsigned *flag, const volatile unsigned *mailbox, long spin_clocks, const float4 *staged) {
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin_clocks) {}
    if (threadIdx.x < 64) __hip_atomic_store(&out[blockIdx.x * 64 + threadIdx.x], 1.0f + staged[threadIdx.x].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && (__hip_atomic_fetch_add(counter, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1) % gridDim.x == 0)
        __hip_atomic_store((unsigned *) flag, *mailbox, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (one PCIe read, by the last workgroup only)
}
static void pair_modes(hipStream_t s, float *out, unsigned long long *counter, unsigned *flag, long spin) {
    const int wgs = 157, reps = 1000, n4 = 64 * 292 / 4;
    float4 *h_in, *d_in;
    unsigned *mailbox;
    hipHostMalloc(&h_in, n4 * sizeof(float4), hipHostMallocDefault);
    hipMalloc(&d_in, n4 * sizeof(float4));
    hipHostMalloc(&mailbox, 64, hipHostMallocDefault);
    for (int i = 0; i < n4; i++) h_in[i] = float4{0, 0, 0, 0};
    hipMemset(counter, 0, 8);
    hipStreamSynchronize(s);
    hipGraph_t graph;
    hipGraphExec_t exec;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    hipLaunchKernelGGL(upload, dim3((n4 + 255) / 256), dim3(256), 0, s, h_in, d_in, n4);
    hipLaunchKernelGGL(work2, dim3(wgs), dim3(1024), 0, s, out, counter, (volatile unsigned *) flag, (const volatile unsigned *) mailbox, spin, d_in);
    hipStreamEndCapture(s, &graph);
    hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    std::vector<double> direct, replay;
    unsigned seq = 100000;
    for (int mode = 0; mode < 2; mode++)
        for (int r = 0; r < reps; r++) {
            seq++;
            const auto t0 = std::chrono::steady_clock::now();
            *mailbox = seq;
            if (mode == 0) {
                hipLaunchKernelGGL(upload, dim3((n4 + 255) / 256), dim3(256), 0, s, h_in, d_in, n4);
                hipLaunchKernelGGL(work2, dim3(wgs), dim3(1024), 0, s, out, counter, (volatile unsigned *) flag, (const volatile unsigned *) mailbox, spin, d_in);
            } else {
                hipGraphLaunch(exec, s);
            }
            while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) __builtin_ia32_pause();
            (mode ? replay : direct).push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
    hipStreamSynchronize(s);
    std::sort(direct.begin(), direct.end());
    std::sort(replay.begin(), replay.end());
    printf("upload + sweep, two direct launches -> flag: median %.2f us (min %.2f)\n", direct[reps / 2], direct[0]);
    printf("upload + sweep, one graph replay    -> flag: median %.2f us (min %.2f)\n", replay[reps / 2], replay[0]);
}
int main() {
    const int wgs = 157, reps = 1000;
    float *out;
    unsigned *flag;
    unsigned long long *counter;
    hipHostMalloc(&out, wgs * 64 * sizeof(float), hipHostMallocDefault);
    hipHostMalloc(&flag, 64, hipHostMallocDefault);
    hipMalloc(&counter, 8);
    hipMemset(counter, 0, 8);
    *flag = 0;
    hipStream_t s;
    hipStreamCreate(&s);
    const long spin = 2000;  // wall_clock64 ticks at 100 MHz: 20 us
    std::vector<double> a, b, c, d;
    int bad = 0;
    for (int mode = 0; mode < 4; mode++)  // 0: stream sync; 1: flag, then a stream sync outside the clock; 2: flag only, calls back to back; 3: flag + hipStreamQuery
        for (int r = 0; r < reps; r++) {
            const unsigned seq = 1 + mode * reps + r;
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(work, dim3(wgs), dim3(1024), 0, s, out, counter, (volatile unsigned *) flag, seq, spin);
            if (mode == 0) {
                hipStreamSynchronize(s);
            } else {
                while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) __builtin_ia32_pause();
                for (int i = 0; i < wgs * 64; i++) bad += out[i] != (float) seq;  // every workgroup's stores must be in already
            }
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            (mode == 0 ? a : mode == 1 ? b : mode == 2 ? c : d).push_back(us);
            if (mode == 1) hipStreamSynchronize(s);
            if (mode == 3) (void) hipStreamQuery(s);
        }
    hipStreamSynchronize(s);
    for (auto *v : {&a, &b, &c, &d}) std::sort(v->begin(), v->end());
    printf("flag only, the next launch straight behind it (the stream never waited for): median %.2f us (min %.2f, max %.2f)\n", c[reps / 2], c[0], c[reps - 1]);
    printf("flag + hipStreamQuery before the next launch: median %.2f us (min %.2f, max %.2f)\n", d[reps / 2], d[0], d[reps - 1]);
    printf("launch -> hipStreamSynchronize returns: median %.2f us (min %.2f)\n", a[reps / 2], a[0]);
    printf("launch -> flag of the last workgroup seen (+ reading all %d results): median %.2f us (min %.2f); results not yet visible: %d\n", wgs * 64, b[reps / 2], b[0], bad);
    pair_modes(s, out, counter, flag, spin);
    return 0;
}
