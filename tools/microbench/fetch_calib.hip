// fetch_calib.hip -- what rocprofv3's FETCH_SIZE reports for a KNOWN byte count in the sweep kernels' own access
// patterns (MI355X_MICROARCH.md: "other access widths are uncalibrated: calibrate on a known byte count in your own
// access pattern before trusting an absolute").  Each kernel streams `bytes` bytes of a buffer larger than the
// Infinity Cache exactly once:
//   calib_sload   s_load_dwordx16 (the quad table's stream: one 64-byte scalar request per half trip)
//   calib_dma     global_load_lds_dwordx4 (the sample refill: 16 bytes per lane straight into LDS)
//   calib_vload   global_load_dwordx4 (the guide's reference pattern: reported at 1/2)
// Build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
// Run:   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib      (then TCC_EA0_RDREQ_sum likewise)
// and compare the counter per kernel with the byte count the program prints.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef int i16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

// every wave reads its own contiguous `per_wave` bytes, 64 bytes per scalar load
__global__ __launch_bounds__(64) void calib_sload(const int *buf, size_t per_wave, int *sink) {
    const __attribute__((address_space(4))) i16 *p =
        (const __attribute__((address_space(4))) i16 *) ((const char *) buf + (size_t) blockIdx.x * per_wave);
    int acc = 0;
    for (size_t k = 0; k < per_wave / 64; k++) {
        const i16 v = p[k];
        acc ^= v[0] ^ v[15];
    }
    if (acc == 0x12345678) sink[0] = acc;  // (keeps the loads)
}

// every workgroup (1024 threads) streams its own contiguous region through LDS, 16 KiB per pass
__global__ __launch_bounds__(1024) void calib_dma(const float *buf, size_t per_wg, int *sink) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const char *src = (const char *) buf + (size_t) blockIdx.x * per_wg;
    const int wave = threadIdx.x >> 6;
    for (size_t base = 0; base < per_wg; base += 16384) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (src + base + threadIdx.x * 16),
                                         (__attribute__((address_space(3))) void *) (lds + wave * 256), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (lds[threadIdx.x] == 1.2345e30f) sink[0] = 1;
}

__global__ __launch_bounds__(1024) void calib_vload(const f4 *buf, size_t per_wg, int *sink) {
    const f4 *src = (const f4 *) ((const char *) buf + (size_t) blockIdx.x * per_wg);
    f4 acc = {0, 0, 0, 0};
    for (size_t k = threadIdx.x; k < per_wg / 16; k += 1024) acc += src[k];
    if (acc.x + acc.y + acc.z + acc.w == 1.2345e30f) sink[0] = 1;
}

int main() {
    const size_t bytes = (size_t) 1 << 30;  // 1 GiB: four times the Infinity Cache
    void *buf = nullptr;
    int *sink = nullptr;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 0, bytes));
    CHECK(hipDeviceSynchronize());
    CHECK(hipFuncSetAttribute((const void *) calib_dma, hipFuncAttributeMaxDynamicSharedMemorySize, 16384));
    for (int rep = 0; rep < 2; rep++) {
        const int waves = 16384;
        hipLaunchKernelGGL(calib_sload, dim3(waves), dim3(64), 0, 0, (const int *) buf, bytes / waves, sink);
        CHECK(hipDeviceSynchronize());
        const int wgs = 2048;
        hipLaunchKernelGGL(calib_dma, dim3(wgs), dim3(1024), 16384, 0, (const float *) buf, bytes / wgs, sink);
        CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL(calib_vload, dim3(wgs), dim3(1024), 0, 0, (const f4 *) buf, bytes / wgs, sink);
        CHECK(hipDeviceSynchronize());
    }
    std::printf("each kernel read %zu bytes (%.1f KiB in FETCH_SIZE's unit) once per launch, 2 launches each\n", bytes, bytes / 1024.0);
    return 0;
}
