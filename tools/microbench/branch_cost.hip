// branch_cost.hip -- what a taken branch costs the wave that takes it, and how much of that the other waves of the SIMD hide.
// Build: hipcc --offload-arch=gfx950 -O3 -o branch_cost branch_cost.hip ; run on an MI355X: ./branch_cost
// A loop body of 32 independent v_pk_fma_f32 (16 accumulators, twice) with, per iteration:
//   mode 0  the loop's own backward branch only (taken)
//   mode 1  + one forward s_branch over two s_nop (taken, lands in the same or the next cache line)
//   mode 2  + one s_cbranch_scc1 that is NOT taken
//   mode 3  + one excursion to a cold section behind the loop and back (two taken branches: the shape of the quad
//             blocks' cold paths)
//   mode 4  + two such excursions
//   mode 5  mode 2 with the compare issued four packed FMAs before its branch;  mode 6  eight before
// Printed: cycles per iteration at the in-kernel clock (s_memtime / s_memrealtime), for 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int REPS = 8192;

#define PK(n, x) "v_pk_fma_f32 v[" #n ":" #n "+1], s[40:41], v[" #x ":" #x "+1], v[" #n ":" #n "+1] op_sel_hi:[0,1,1]\n\t"
#define PK8A PK(8, 42) PK(10, 40) PK(12, 46) PK(14, 44) PK(16, 42) PK(18, 40) PK(20, 46) PK(22, 44)
#define PK4B PK(24, 42) PK(26, 40) PK(28, 46) PK(30, 44)
#define PK4C PK(32, 42) PK(34, 40) PK(36, 46) PK(38, 44)
#define PK16 PK8A PK4B PK4C

template <int MODE>
__global__ __launch_bounds__(1024, 4) void branch_kernel(float *out, float seed, unsigned long long *stamps) {
    float r = 0.0f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    asm volatile(
        ".irp n,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,36,37,38,39\n\tv_mov_b32 v\\n, %[s]\n\t.endr\n\t"
        ".irp n,40,41,42,43,44,45,46,47\n\tv_mov_b32 v\\n, 0\n\t.endr\n\t"
        "s_mov_b32 s40, 0x3f800000\n\ts_mov_b32 s41, 0x3f800000\n\t"
        "s_mov_b32 s36, %[n]\n\t"
        "s_mov_b32 s37, 1\n\t"
        ".Lloop_%=:\n\t"
        PK8A
        ".if %c[mode] == 6\n\t"
        "s_cmp_eq_u32 s37, 0\n\t"
        ".endif\n\t"
        PK4B
        ".if %c[mode] == 5\n\t"
        "s_cmp_eq_u32 s37, 0\n\t"
        ".endif\n\t"
        PK4C
        ".if %c[mode] == 5 || %c[mode] == 6\n\t"
        "s_cbranch_scc1 .Lcold_a_%=\n\t"
        ".endif\n\t"
        ".if %c[mode] == 1\n\t"
        "s_branch .Lfwd_%=\n\ts_nop 0\n\ts_nop 0\n\t.Lfwd_%=:\n\t"
        ".elseif %c[mode] == 2\n\t"
        "s_cmp_eq_u32 s37, 0\n\ts_cbranch_scc1 .Lcold_a_%=\n\t"
        ".elseif %c[mode] == 3\n\t"
        "s_cmp_eq_u32 s37, 1\n\ts_cbranch_scc1 .Lcold_a_%=\n\t"
        ".elseif %c[mode] == 4\n\t"
        "s_cmp_eq_u32 s37, 1\n\ts_cbranch_scc1 .Lcold_a_%=\n\t"
        ".endif\n\t"
        ".Lback_a_%=:\n\t"
        PK16
        ".if %c[mode] == 4\n\t"
        "s_cmp_eq_u32 s37, 1\n\ts_cbranch_scc1 .Lcold_b_%=\n\t"
        ".endif\n\t"
        ".Lback_b_%=:\n\t"
        "s_sub_u32 s36, s36, 1\n\t"
        "s_cmp_lg_u32 s36, 0\n\t"
        "s_cbranch_scc1 .Lloop_%=\n\t"
        "s_branch .Ldone_%=\n\t"
        ".rept 64\n\ts_nop 0\n\t.endr\n\t"  // (the cold section sits a few cache lines behind the loop)
        ".Lcold_a_%=:\n\t"
        PK(8, 42) PK(10, 40)
        "s_branch .Lback_a_%=\n\t"
        ".Lcold_b_%=:\n\t"
        PK(12, 46) PK(14, 44)
        "s_branch .Lback_b_%=\n\t"
        ".Ldone_%=:\n\t"
        "v_add_f32 %[r], v8, v10\n\t"
        "v_add_f32 %[r], %[r], v39\n\t"
        : [r] "=v"(r)
        : [s] "v"(seed), [n] "s"(REPS), [mode] "n"(MODE)
        : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27",
          "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47",
          "s36", "s37", "s40", "s41", "scc");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = c1 - c0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
    if (r == 12345.0f) out[threadIdx.x] = r;
}

template <int MODE>
static void run(const char *what, int n_valu, float *d_out, unsigned long long *d_stamps) {
    for (int waves_per_simd : {1, 2, 4}) {
        const int threads = 64 * 4 * waves_per_simd;
        for (int warm = 0; warm < 30; warm++) hipLaunchKernelGGL(branch_kernel<MODE>, dim3(256), dim3(threads), 0, 0, d_out, 0.0f, d_stamps);
        CHECK(hipDeviceSynchronize());
        unsigned long long h_st[512];
        CHECK(hipMemcpy(h_st, d_stamps, sizeof h_st, hipMemcpyDeviceToHost));
        double cyc[256];
        for (int i = 0; i < 256; i++) cyc[i] = (double) h_st[2 * i] / REPS;
        std::sort(cyc, cyc + 256);
        const double ghz = (double) h_st[0] / (double) h_st[1] * 0.1;
        std::printf("%-66s %d waves/SIMD: %7.1f cycles per iteration of a wave (%d VALU: %5.2f SIMD cycles per VALU), clock %.2f GHz\n", what, waves_per_simd,
                    cyc[128], n_valu, cyc[128] / (n_valu * waves_per_simd), ghz);
    }
}

// issue rate of the 64-bit add that would turn two LDS addresses at once (SGPR pair + lane pair): v_lshl_add_u64
template <int VAR>
__global__ __launch_bounds__(1024, 4) void rate64_kernel(float *out, float seed) {
    float r = 0.0f;
    asm volatile(
        ".irp n,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,40,41\n\tv_mov_b32 v\\n, 0\n\t.endr\n\t"
        "s_mov_b32 s40, 8\n\ts_mov_b32 s41, 16\n\ts_mov_b32 s36, %[n]\n\t"
        ".Lloop_%=:\n\t"
        ".rept 4\n\t"
        ".if %c[var] == 0\n\t"
        "v_lshl_add_u64 v[8:9], v[40:41], 0, s[40:41]\n\tv_lshl_add_u64 v[10:11], v[40:41], 0, s[40:41]\n\t"
        "v_lshl_add_u64 v[12:13], v[40:41], 0, s[40:41]\n\tv_lshl_add_u64 v[14:15], v[40:41], 0, s[40:41]\n\t"
        "v_lshl_add_u64 v[16:17], v[40:41], 0, s[40:41]\n\tv_lshl_add_u64 v[18:19], v[40:41], 0, s[40:41]\n\t"
        "v_lshl_add_u64 v[20:21], v[40:41], 0, s[40:41]\n\tv_lshl_add_u64 v[22:23], v[40:41], 0, s[40:41]\n\t"
        ".else\n\t"
        "v_add_u32 v8, s40, v40\n\tv_add_u32 v10, s40, v40\n\tv_add_u32 v12, s40, v40\n\tv_add_u32 v14, s40, v40\n\t"
        "v_add_u32 v16, s40, v40\n\tv_add_u32 v18, s40, v40\n\tv_add_u32 v20, s40, v40\n\tv_add_u32 v22, s40, v40\n\t"
        ".endif\n\t"
        ".endr\n\t"
        "s_sub_u32 s36, s36, 1\n\ts_cmp_lg_u32 s36, 0\n\ts_cbranch_scc1 .Lloop_%=\n\t"
        "v_cvt_f32_u32 %[r], v8\n\t"
        : [r] "=v"(r)
        : [s] "v"(seed), [n] "s"(REPS), [var] "n"(VAR)
        : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v40", "v41", "s36", "s40", "s41", "scc");
    if (r == 12345.0f) out[threadIdx.x] = r;
}

template <int VAR>
static void rate64_one(const char *what, float *d_out) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int warm = 0; warm < 5; warm++) hipLaunchKernelGGL(rate64_kernel<VAR>, dim3(256), dim3(1024), 0, 0, d_out, 0.0f);
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(rate64_kernel<VAR>, dim3(256), dim3(1024), 0, 0, d_out, 0.0f);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    std::printf("%-40s 4 waves/SIMD: %.3f ms, %.2f SIMD cycles per instruction at 2.4 GHz\n", what, ms, ms * 1e-3 * 2.4e9 / ((double) REPS * 32 * 4));
}

static void rate64(float *d_out) {
    rate64_one<0>("v_lshl_add_u64 (VGPR pair + SGPR pair)", d_out);
    rate64_one<1>("v_add_u32 (SGPR + VGPR)", d_out);
}

int main() {
    float *d_out;
    unsigned long long *d_stamps;
    CHECK(hipMalloc(&d_out, 4096));
    CHECK(hipMalloc(&d_stamps, 512 * sizeof(unsigned long long)));
    run<0>("32 v_pk_fma_f32 + the loop branch", 32, d_out, d_stamps);
    run<1>("  + a taken forward s_branch over two s_nop", 32, d_out, d_stamps);
    run<2>("  + an s_cbranch that is not taken", 32, d_out, d_stamps);
    run<3>("  + one excursion to a cold section and back (2 extra VALU)", 34, d_out, d_stamps);
    run<4>("  + two excursions (4 extra VALU)", 36, d_out, d_stamps);
    run<5>("  + an s_cbranch not taken, its s_cmp 4 FMAs earlier", 32, d_out, d_stamps);
    run<6>("  + an s_cbranch not taken, its s_cmp 8 FMAs earlier", 32, d_out, d_stamps);
    rate64(d_out);
    return 0;
}
