// dot2_rate.hip -- BASELINE configs[4] asks for "bf16 vs fp32 accumulator".  gfx950 has no bf16 VALU arithmetic except the
// two-element dot products v_dot2c_f32_bf16 (VOP2: D += a.lo * b.lo + a.hi * b.hi, fp32 accumulate) and v_dot2_f32_bf16
// (VOP3P).  A bf16-SAMPLE sweep would store {X[t], X[t+1]} as one dword and the weights {f, 1 - f} as one SGPR, and spend
// ONE dot2 per output sample and frame where the fp32 sweep spends two FMAs per sample -- packed over two frames, i.e. ONE
// v_pk_fma_f32 per output sample and frame as well.  So a dot2 sweep can only win if a dot2 issues FASTER than a packed FMA.
// This measures it: SIMD cycles per instruction (at the 2.4 GHz the peak is quoted on) for streams of 16 independent
// accumulators, W waves per SIMD on every SIMD of the chip.
// Build: hipcc --offload-arch=gfx950 -O3 -o dot2_rate dot2_rate.hip ; run on an MI355X: ./dot2_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int REPS = 4096;

template <int VAR>
__global__ __launch_bounds__(1024) void rate_kernel(float *out, float seed) {
    float r = 0.0f;
    asm volatile(
        "v_mov_b32 v8, %[s]\n\tv_mov_b32 v9, %[s]\n\tv_mov_b32 v10, %[s]\n\tv_mov_b32 v11, %[s]\n\t"
        "v_mov_b32 v12, %[s]\n\tv_mov_b32 v13, %[s]\n\tv_mov_b32 v14, %[s]\n\tv_mov_b32 v15, %[s]\n\t"
        "v_mov_b32 v16, %[s]\n\tv_mov_b32 v17, %[s]\n\tv_mov_b32 v18, %[s]\n\tv_mov_b32 v19, %[s]\n\t"
        "v_mov_b32 v20, %[s]\n\tv_mov_b32 v21, %[s]\n\tv_mov_b32 v22, %[s]\n\tv_mov_b32 v23, %[s]\n\t"
        "v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\t"
        "v_mov_b32 v44, 0\n\tv_mov_b32 v45, 0\n\tv_mov_b32 v46, 0\n\tv_mov_b32 v47, 0\n\t"
        "s_mov_b32 s40, 0x3f803f80\n\ts_mov_b32 s41, 0x3f800000\n\t"
        "s_mov_b32 s36, %[n]\n\t"
        ".Lloop_%=:\n\t"
        ".rept 4\n\t"
        ".if %c[var] == 0\n\t"  // the fp32 sweep's instruction: packed FMA, scalar weight
        "v_pk_fma_f32 v[8:9], s[40:41], v[42:43], v[8:9] op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 v[10:11], s[40:41], v[40:41], v[10:11] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[12:13], s[40:41], v[46:47], v[12:13] op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 v[14:15], s[40:41], v[44:45], v[14:15] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[16:17], s[40:41], v[42:43], v[16:17] op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 v[18:19], s[40:41], v[40:41], v[18:19] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[20:21], s[40:41], v[46:47], v[20:21] op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 v[22:23], s[40:41], v[44:45], v[22:23] op_sel_hi:[0,1,1]\n\t"
        ".elseif %c[var] == 1\n\t"  // VOP2 dot2, scalar weight pair {f, g} as bf16 x 2, sample pair {X[t], X[t+1]} as bf16 x 2
        "v_dot2c_f32_bf16 v8, s40, v42\n\tv_dot2c_f32_bf16 v9, s40, v43\n\tv_dot2c_f32_bf16 v10, s40, v40\n\tv_dot2c_f32_bf16 v11, s40, v41\n\t"
        "v_dot2c_f32_bf16 v12, s40, v46\n\tv_dot2c_f32_bf16 v13, s40, v47\n\tv_dot2c_f32_bf16 v14, s40, v44\n\tv_dot2c_f32_bf16 v15, s40, v45\n\t"
        ".elseif %c[var] == 2\n\t"  // VOP3P dot2
        "v_dot2_f32_bf16 v8, s40, v42, v8\n\tv_dot2_f32_bf16 v9, s40, v43, v9\n\tv_dot2_f32_bf16 v10, s40, v40, v10\n\tv_dot2_f32_bf16 v11, s40, v41, v11\n\t"
        "v_dot2_f32_bf16 v12, s40, v46, v12\n\tv_dot2_f32_bf16 v13, s40, v47, v13\n\tv_dot2_f32_bf16 v14, s40, v44, v14\n\tv_dot2_f32_bf16 v15, s40, v45, v15\n\t"
        ".elseif %c[var] == 3\n\t"  // plain fp32 FMA (one frame per instruction, like a dot2)
        "v_fma_f32 v8, s40, v42, v8\n\tv_fma_f32 v9, s40, v43, v9\n\tv_fma_f32 v10, s40, v40, v10\n\tv_fma_f32 v11, s40, v41, v11\n\t"
        "v_fma_f32 v12, s40, v46, v12\n\tv_fma_f32 v13, s40, v47, v13\n\tv_fma_f32 v14, s40, v44, v14\n\tv_fma_f32 v15, s40, v45, v15\n\t"
        ".elseif %c[var] == 4\n\t"  // f16 dot2 for comparison
        "v_dot2c_f32_f16 v8, s40, v42\n\tv_dot2c_f32_f16 v9, s40, v43\n\tv_dot2c_f32_f16 v10, s40, v40\n\tv_dot2c_f32_f16 v11, s40, v41\n\t"
        "v_dot2c_f32_f16 v12, s40, v46\n\tv_dot2c_f32_f16 v13, s40, v47\n\tv_dot2c_f32_f16 v14, s40, v44\n\tv_dot2c_f32_f16 v15, s40, v45\n\t"
        ".elseif %c[var] == 5\n\t"  // a bf16 ACCUMULATOR step as the device can do it: fp32 add, convert-and-pack to bf16, unpack (x 2 registers)
        "v_pk_add_f32 v[8:9], v[8:9], v[42:43]\n\tv_cvt_pk_bf16_f32 v10, v8, v9\n\tv_lshlrev_b32 v8, 16, v10\n\tv_and_b32 v9, 0xffff0000, v10\n\t"
        "v_pk_add_f32 v[12:13], v[12:13], v[46:47]\n\tv_cvt_pk_bf16_f32 v14, v12, v13\n\tv_lshlrev_b32 v12, 16, v14\n\tv_and_b32 v13, 0xffff0000, v14\n\t"
        ".endif\n\t"
        ".endr\n\t"
        "s_sub_u32 s36, s36, 1\n\t"
        "s_cmp_lg_u32 s36, 0\n\t"
        "s_cbranch_scc1 .Lloop_%=\n\t"
        "v_add_f32 %[r], v8, v9\n\tv_add_f32 %[r], %[r], v10\n\tv_add_f32 %[r], %[r], v12\n\tv_add_f32 %[r], %[r], v16\n\tv_add_f32 %[r], %[r], v20\n\t"
        : [r] "=&v"(r)
        : [s] "v"(seed), [n] "s"(REPS), [var] "n"(VAR)
        : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v40", "v41", "v42",
          "v43", "v44", "v45", "v46", "v47", "s36", "s40", "s41", "scc");
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int VAR>
static void run(const char *what, int per_rept, int waves_per_simd) {
    int n_cu = 0;
    CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
    float *d = nullptr;
    CHECK(hipMalloc(&d, 4096));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    const int threads = 64 * 4 * waves_per_simd;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(rate_kernel<VAR>, dim3(n_cu), dim3(threads), 0, 0, d, 0.0f);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
    }
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double instr = (double) REPS * 4 * per_rept * waves_per_simd;  // per SIMD
    std::printf("%-64s %d waves/SIMD: %.2f cycles per instruction at 2.4 GHz (%.3f ms)\n", what, waves_per_simd, ms * 1e-3 * 2.4e9 / instr, ms);
    CHECK(hipFree(d));
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_pk_fma_f32 (fp32 sweep: 2 frames per instruction)", 8, w);
        run<1>("v_dot2c_f32_bf16 (bf16 samples + weights, fp32 accumulate)", 8, w);
        run<2>("v_dot2_f32_bf16 (VOP3P form)", 8, w);
        run<3>("v_fma_f32", 8, w);
        run<4>("v_dot2c_f32_f16", 8, w);
        run<5>("bf16 accumulator step: pk_add + cvt_pk_bf16 + 2 unpack (4 instr)", 8, w);
    }
    return 0;
}
