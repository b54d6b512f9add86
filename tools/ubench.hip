// ubench.hip -- gfx950 micro-benchmarks that size the sweep kernel's inner loop:
// how fast one CU issues v_fmac_f32 / v_pk_fma_f32, how fast it reads LDS with
// ds_read_b32/b64/b128, and what the planned per-(pixel, mic) item
//   { 1 v_add_u32 (address), 2 ds_read_b64, 8 v_fmac_f32 with an SGPR multiplier }
// sustains at 1, 2 and 4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 ubench.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e = (x);                                                           \
        if (e != hipSuccess) {                                                        \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

// ---- 16 independent v_fmac_f32 per iteration, multiplier in an SGPR
__global__ void k_fmac(float *out, int iters, float f) {
    float a[16];
#pragma unroll
    for (int j = 0; j < 16; j++) a[j] = threadIdx.x * 0.001f + j;
    float x = threadIdx.x * 0.5f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 16; j++) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[j]) : "s"(f), "v"(x));
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) s += a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- 8 independent v_pk_fma_f32 per iteration (16 lane-FMAs)
__global__ void k_pkfma(float *out, int iters, float f) {
    f2 a[8];
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = f2{threadIdx.x * 0.001f + j, 1.0f};
    f2 x = f2{threadIdx.x * 0.5f, 2.0f};
    f2 ff = f2{f, f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[j]) : "v"(ff), "v"(x));
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s += a[j].x + a[j].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- LDS reads only.  WIDTH = 4, 8, 16 bytes per lane, conflict-free consecutive lanes
template <int WIDTH>
__global__ void k_lds(float *out, int iters, int skew = 0) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned addr = lane * (WIDTH == 17 ? 8 : WIDTH) + skew;
    float acc = 0;
    for (int it = 0; it < iters; it++) {
        if (WIDTH == 4) {
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; j++) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[j]) : "v"(addr), "n"(j * 256));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 16; j++) asm volatile("" ::"v"(v[j]));
            acc += v[0];
        } else if (WIDTH == 8) {
            f2 v[16];
#pragma unroll
            for (int j = 0; j < 16; j++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[j]) : "v"(addr), "n"(j * 512));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 16; j++) asm volatile("" ::"v"(v[j]));
            acc += v[0].x;
        } else if (WIDTH == 17) {  // 16 bytes per lane as two 8-byte reads 512 bytes apart in ONE instruction
            f4 v[8];
#pragma unroll
            for (int j = 0; j < 8; j++)
                asm volatile("ds_read2st64_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(v[j]) : "v"(addr), "n"(2 * j), "n"(2 * j + 1));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 8; j++) asm volatile("" ::"v"(v[j]));
            acc += v[0].x;
        } else {
            f4 v[16];
#pragma unroll
            for (int j = 0; j < 16; j++) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[j]) : "v"(addr), "n"(j * 1024));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 16; j++) asm volatile("" ::"v"(v[j]));
            acc += v[0].x;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// ---- the planned inner item, software-pipelined one item deep:
//   v_add_u32 addr, s_off, v_lane8 ; ds_read_b64 x, addr ; ds_read_b64 y, addr offset:512
//   8 x v_fmac_f32 acc, s_f|s_g, x|y   (6 accumulators)
// ITEMS per iteration; s_off / f / g arrive as scalars (kernel args, varied per item).
template <bool PK>
__global__ void k_item(float *out, int iters, int off0, float f, float g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i * 1e-6f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned lane8 = lane * 8;
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0, pa = 0, pb = 0;
    f2 x, y, xn, yn;
    unsigned addr;
    int soff = __builtin_amdgcn_readfirstlane(off0);
    asm volatile("v_add_u32 %0, %1, %2" : "=v"(addr) : "s"(soff), "v"(lane8));
    asm volatile("ds_read_b64 %0, %1" : "=v"(x) : "v"(addr));
    asm volatile("ds_read_b64 %0, %1 offset:512" : "=v"(y) : "v"(addr));
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            soff = (soff + 1384) & 0x3ff8;  // scalar ALU: next item's window start (8-byte aligned)
            asm volatile("v_add_u32 %0, %1, %2" : "=v"(addr) : "s"(soff), "v"(lane8));
            asm volatile("ds_read_b64 %0, %1" : "=v"(xn) : "v"(addr));
            asm volatile("ds_read_b64 %0, %1 offset:512" : "=v"(yn) : "v"(addr));
            asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
            if (!PK) {
                asm volatile(
                    "v_fmac_f32 %0, %6, %8\n\tv_fmac_f32 %1, %6, %9\n\t"
                    "v_fmac_f32 %0, %7, %9\n\tv_fmac_f32 %4, %7, %8\n\t"
                    "v_fmac_f32 %2, %6, %10\n\tv_fmac_f32 %3, %6, %11\n\t"
                    "v_fmac_f32 %2, %7, %11\n\tv_fmac_f32 %5, %7, %10"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(pa), "+v"(pb)
                    : "s"(f), "s"(g), "v"(x.x), "v"(x.y), "v"(y.x), "v"(y.y));
            } else {
                // packed form: (a0,a1) += f*(x0,x1); (pa,a0') ... expressed on pairs
                f2 A = f2{a0, a1}, Bq = f2{pa, a2}, Cc = f2{a3, pb};
                f2 ff = f2{f, f}, gg = f2{g, g};
                asm volatile(
                    "v_pk_fma_f32 %0, %3, %5, %0\n\tv_pk_fma_f32 %1, %4, %5, %1\n\t"
                    "v_pk_fma_f32 %2, %3, %6, %2\n\tv_pk_fma_f32 %1, %4, %6, %1"
                    : "+v"(A), "+v"(Bq), "+v"(Cc)
                    : "v"(ff), "v"(gg), "v"(x), "v"(y));
                a0 = A.x; a1 = A.y; pa = Bq.x; a2 = Bq.y; a3 = Cc.x; pb = Cc.y;
            }
            x = xn;
            y = yn;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + pa + pb + x.x + y.x;
}


// ---- design-shaped item in plain C++ (what the sweep kernel's inner loop will look like):
// per item: one address add, FPI x 2 ds_read_b64 (frames at a constant LDS stride), FPI x 4
// v_pk_fma_f32 with the wave-uniform (f, g) as scalar operands.  Table entries come through
// scalar loads from `lut` (off, f, g, pad).  Reports shader cycles per block.
struct Entry { int off; float f, g; int pad; };

template <int FPI>
__global__ __launch_bounds__(256) void k_design(float *out, long long *cyc, const Entry *lut, int n_items, int reps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i * 1e-6f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const char *base = (const char *) lds + lane * 8;
    f2 A[FPI], Q[FPI], C[FPI], R[FPI];
#pragma unroll
    for (int b = 0; b < FPI; b++) A[b] = Q[b] = C[b] = R[b] = f2{0, 0};
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; r++) {
#pragma unroll 4
        for (int i = 0; i < n_items; i++) {
            const Entry e = lut[i];
            const char *p = base + e.off;
            const f2 F = f2{e.f, e.f}, G = f2{e.g, e.g};
#pragma unroll
            for (int b = 0; b < FPI; b++) {
                const f2 x = *(const f2 *) (p + b * 4096);
                const f2 y = *(const f2 *) (p + b * 4096 + 512);
                A[b] = __builtin_elementwise_fma(F, x, A[b]);
                Q[b] = __builtin_elementwise_fma(G, x, Q[b]);
                C[b] = __builtin_elementwise_fma(F, y, C[b]);
                R[b] = __builtin_elementwise_fma(G, y, R[b]);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int b = 0; b < FPI; b++) s += A[b].x + A[b].y + Q[b].x + Q[b].y + C[b].x + C[b].y + R[b].x + R[b].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void k_clock(long long *cyc, long long *rt, int spin) {
    const long long t0 = __builtin_readcyclecounter();
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    float a = threadIdx.x;
    for (int i = 0; i < spin; i++) asm volatile("v_fmac_f32 %0, %0, %0" : "+v"(a));
    const long long t1 = __builtin_readcyclecounter();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
    if (a == 123.456f) cyc[0] = 0;
}

struct Result {
    double ms;
};

template <typename F>
double time_ms(F launch, int reps = 5) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    launch();
    CHECK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < reps; r++) {
        CHECK(hipEventRecord(a));
        launch();
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s  CUs %d  clock %d kHz  LDS/block %zu\n", prop.gcnArchName, cus, prop.clockRate,
           prop.sharedMemPerBlock);
    float *out;
    CHECK(hipMalloc(&out, (size_t) cus * 16 * 1024 * sizeof(float)));
    const double ghz = 2.4;
    CHECK(hipFuncSetAttribute((const void *) k_item<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHECK(hipFuncSetAttribute((const void *) k_item<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));

    for (int wps : {1, 2, 4, 8}) {  // waves per SIMD; block = 256 threads = 1 wave per SIMD
        const int blocks = cus * wps;
        const int iters = 4000;
        {
            double ms = time_ms([&] { hipLaunchKernelGGL(k_fmac, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f); });
            double lane_fma = (double) blocks * 256 * iters * 16;
            printf("fmac      wps %d: %8.3f ms  %.1f lane-FMA/clk/CU @2.4GHz  (%.1f TFLOP/s)\n", wps, ms,
                   lane_fma / (ms * 1e-3) / (ghz * 1e9) / cus, 2 * lane_fma / (ms * 1e-3) / 1e12);
        }
        {
            double ms = time_ms([&] { hipLaunchKernelGGL(k_pkfma, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f); });
            double lane_fma = (double) blocks * 256 * iters * 16;
            printf("pk_fma    wps %d: %8.3f ms  %.1f lane-FMA/clk/CU @2.4GHz  (%.1f TFLOP/s)\n", wps, ms,
                   lane_fma / (ms * 1e-3) / (ghz * 1e9) / cus, 2 * lane_fma / (ms * 1e-3) / 1e12);
        }
        {
            const int it2 = 2000;
            double ms = time_ms([&] { hipLaunchKernelGGL(k_lds<4>, dim3(blocks), dim3(256), 32768, 0, out, it2); });
            double bytes = (double) blocks * 256 * it2 * 16 * 4;
            printf("ds_b32    wps %d: %8.3f ms  %.1f B/clk/CU @2.4GHz\n", wps, ms, bytes / (ms * 1e-3) / (ghz * 1e9) / cus);
            ms = time_ms([&] { hipLaunchKernelGGL(k_lds<8>, dim3(blocks), dim3(256), 32768, 0, out, it2); });
            bytes = (double) blocks * 256 * it2 * 16 * 8;
            printf("ds_b64    wps %d: %8.3f ms  %.1f B/clk/CU @2.4GHz\n", wps, ms, bytes / (ms * 1e-3) / (ghz * 1e9) / cus);
            ms = time_ms([&] { hipLaunchKernelGGL(k_lds<17>, dim3(blocks), dim3(256), 32768, 0, out, it2); });
            bytes = (double) blocks * 256 * it2 * 8 * 16;
            printf("ds_2st64  wps %d: %8.3f ms  %.1f B/clk/CU @2.4GHz\n", wps, ms, bytes / (ms * 1e-3) / (ghz * 1e9) / cus);
            ms = time_ms([&] { hipLaunchKernelGGL(k_lds<16>, dim3(blocks), dim3(256), 32768, 0, out, it2, 8); });
            bytes = (double) blocks * 256 * it2 * 16 * 16;
            printf("ds_b128+8 wps %d: %8.3f ms  %.1f B/clk/CU @2.4GHz  (8-byte aligned only)\n", wps, ms, bytes / (ms * 1e-3) / (ghz * 1e9) / cus);
            if (wps <= 4) {
                ms = time_ms([&] { hipLaunchKernelGGL(k_lds<16>, dim3(blocks), dim3(256), 32768, 0, out, it2); });
                bytes = (double) blocks * 256 * it2 * 16 * 16;
                printf("ds_b128   wps %d: %8.3f ms  %.1f B/clk/CU @2.4GHz\n", wps, ms, bytes / (ms * 1e-3) / (ghz * 1e9) / cus);
            }
        }
        if (wps <= 4) {
            const int it3 = 1000;
            const size_t lds = 32768;
            double ms = time_ms([&] { hipLaunchKernelGGL(k_item<false>, dim3(blocks), dim3(256), lds, 0, out, it3, 64, 0.3f, 0.7f); });
            double items = (double) blocks * 4 * it3 * 8;  // wave-items
            double triples = items * 256;
            printf("item      wps %d: %8.3f ms  %.2f triples/clk/CU @2.4GHz  (%.2f G wave-items/s)\n", wps, ms,
                   triples / (ms * 1e-3) / (ghz * 1e9) / cus, items / (ms * 1e-3) / 1e9);
            ms = time_ms([&] { hipLaunchKernelGGL(k_item<true>, dim3(blocks), dim3(256), lds, 0, out, it3, 64, 0.3f, 0.7f); });
            printf("item(pk)  wps %d: %8.3f ms  %.2f triples/clk/CU @2.4GHz\n", wps, ms,
                   triples / (ms * 1e-3) / (ghz * 1e9) / cus);
        }
    }

    {   // design-shaped kernel, cycle-stamped
        long long *cyc, *rt;
        CHECK(hipMalloc(&cyc, 4096 * sizeof(long long)));
        CHECK(hipMalloc(&rt, 4096 * sizeof(long long)));
        const int n_items = 256;
        std::vector<Entry> h(n_items);
        for (int i = 0; i < n_items; i++) h[i] = Entry{((i * 1384) & 0x3ff8), 0.25f + i * 1e-3f, 0.75f - i * 1e-3f, 0};
        Entry *lut;
        CHECK(hipMalloc(&lut, n_items * sizeof(Entry)));
        CHECK(hipMemcpy(lut, h.data(), n_items * sizeof(Entry), hipMemcpyHostToDevice));
        auto report = [&](const char *name, int fpi, int wps, double ms, int reps) {
            std::vector<long long> c(cus * wps);
            CHECK(hipMemcpy(c.data(), cyc, c.size() * sizeof(long long), hipMemcpyDeviceToHost));
            std::sort(c.begin(), c.end());
            const double med = (double) c[c.size() / 2];
            const double items = (double) n_items * reps;  // per wave
            const double triples_cu = items * fpi * 256 * 4 * wps;  // per CU: 4*wps waves
            printf("%s fpi %d wps %d: %8.3f ms  median %.0f cyc/block  %.1f cyc/item/wave  %.2f triples/cyc/CU (in-kernel clock)  eff clock %.2f GHz\n",
                   name, fpi, wps, ms, med, med / items, triples_cu / med, med / (ms * 1e-3) / 1e9);
        };
        for (int wps : {1, 2, 4}) {
            const int blocks = cus * wps, reps = 40;
            double ms;
            ms = time_ms([&] { hipLaunchKernelGGL(k_design<1>, dim3(blocks), dim3(256), 36864, 0, out, cyc, lut, n_items, reps); });
            report("design", 1, wps, ms, reps);
            ms = time_ms([&] { hipLaunchKernelGGL(k_design<2>, dim3(blocks), dim3(256), 36864, 0, out, cyc, lut, n_items, reps); });
            report("design", 2, wps, ms, reps);
            ms = time_ms([&] { hipLaunchKernelGGL(k_design<4>, dim3(blocks), dim3(256), 36864, 0, out, cyc, lut, n_items, reps); });
            report("design", 4, wps, ms, reps);
        }
        hipLaunchKernelGGL(k_clock, dim3(cus), dim3(256), 0, 0, cyc, rt, 2000000);
        CHECK(hipDeviceSynchronize());
        std::vector<long long> c(cus), r(cus);
        CHECK(hipMemcpy(c.data(), cyc, cus * sizeof(long long), hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(r.data(), rt, cus * sizeof(long long), hipMemcpyDeviceToHost));
        printf("clock: %lld shader cycles in %lld realtime ticks (100 MHz) -> %.3f GHz; %.2f cyc per dependent v_fmac\n", c[0], r[0],
               (double) c[0] / ((double) r[0] / 100e6) / 1e9, (double) c[0] / 2000000);
    }
    CHECK(hipFree(out));
    return 0;
}
