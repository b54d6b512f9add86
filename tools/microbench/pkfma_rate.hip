// pkfma_rate.hip -- what a SIMD sustains on v_pk_fma_f32 streams shaped like the sweep kernels' inner loops.
// Build: hipcc --offload-arch=gfx950 -O3 -o pkfma_rate pkfma_rate.hip ; run on an MI355X: ./pkfma_rate
// Each variant runs REPS x 64 instructions per wave, W waves per SIMD on every SIMD of the chip; the figure printed is
// SIMD cycles per instruction at the 2.4 GHz the peak is quoted on (4.0 = the v_pk_fma_f32 peak).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int REPS = 4096;

// 16 accumulator pairs v[8:39], 16 sample pairs from v40 (X) ; variants pick which registers meet in one instruction
#define BODY16(fmt_fn) fmt_fn(0) fmt_fn(1) fmt_fn(2) fmt_fn(3) fmt_fn(4) fmt_fn(5) fmt_fn(6) fmt_fn(7)

template <int VAR>
__global__ __launch_bounds__(1024) void rate_kernel(float *out, float seed) {
    __shared__ float lds_buf[2048];  // (variant 7 reads it; the values do not matter)
    if (seed == 54321.0f) lds_buf[threadIdx.x] = seed;
    float r = 0.0f;
    // registers are named explicitly: the point is WHICH registers an instruction reads
    asm volatile(
        "v_mov_b32 v8, %[s]\n\tv_mov_b32 v9, %[s]\n\tv_mov_b32 v10, %[s]\n\tv_mov_b32 v11, %[s]\n\t"
        "v_mov_b32 v12, %[s]\n\tv_mov_b32 v13, %[s]\n\tv_mov_b32 v14, %[s]\n\tv_mov_b32 v15, %[s]\n\t"
        "v_mov_b32 v16, %[s]\n\tv_mov_b32 v17, %[s]\n\tv_mov_b32 v18, %[s]\n\tv_mov_b32 v19, %[s]\n\t"
        "v_mov_b32 v20, %[s]\n\tv_mov_b32 v21, %[s]\n\tv_mov_b32 v22, %[s]\n\tv_mov_b32 v23, %[s]\n\t"
        "v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\t"
        "v_mov_b32 v44, 0\n\tv_mov_b32 v45, 0\n\tv_mov_b32 v46, 0\n\tv_mov_b32 v47, 0\n\t"
        "s_mov_b32 s40, 0x3f800000\n\ts_mov_b32 s41, 0x3f800000\n\t"
        "s_mov_b32 s36, %[n]\n\t"
        "s_mov_b32 s37, 0\n\t"
        "v_lshlrev_b32 v56, 3, %[tid]\n\t"
        "v_lshlrev_b32 v57, 4, %[tid]\n\t"
        ".Lloop_%=:\n\t"
        ".rept 8\n\t"
        ".if %c[var] == 0\n\t"  // all VGPR, sample pair in the banks the accumulator is NOT in
        "v_pk_fma_f32 v[8:9], v[40:41], v[42:43], v[8:9]\n\t"
        "v_pk_fma_f32 v[10:11], v[40:41], v[44:45], v[10:11]\n\t"
        "v_pk_fma_f32 v[12:13], v[40:41], v[46:47], v[12:13]\n\t"
        "v_pk_fma_f32 v[14:15], v[40:41], v[44:45], v[14:15]\n\t"
        "v_pk_fma_f32 v[16:17], v[40:41], v[42:43], v[16:17]\n\t"
        "v_pk_fma_f32 v[18:19], v[40:41], v[44:45], v[18:19]\n\t"
        "v_pk_fma_f32 v[20:21], v[40:41], v[46:47], v[20:21]\n\t"
        "v_pk_fma_f32 v[22:23], v[40:41], v[44:45], v[22:23]\n\t"
        ".elseif %c[var] == 1\n\t"  // scalar coefficient, sample pair in the accumulator's banks (v8/v40: both 0 mod 4)
        "v_pk_fma_f32 v[8:9], s[40:41], v[40:41], v[8:9] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[10:11], s[40:41], v[42:43], v[10:11] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[12:13], s[40:41], v[44:45], v[12:13] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[14:15], s[40:41], v[46:47], v[14:15] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[16:17], s[40:41], v[40:41], v[16:17] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[18:19], s[40:41], v[42:43], v[18:19] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[20:21], s[40:41], v[44:45], v[20:21] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[22:23], s[40:41], v[46:47], v[22:23] op_sel_hi:[0,1,1]\n\t"
        ".elseif %c[var] == 2\n\t"  // scalar coefficient, sample pair in the OTHER banks
        "v_pk_fma_f32 v[8:9], s[40:41], v[42:43], v[8:9] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[10:11], s[40:41], v[40:41], v[10:11] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[12:13], s[40:41], v[46:47], v[12:13] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[14:15], s[40:41], v[44:45], v[14:15] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[16:17], s[40:41], v[42:43], v[16:17] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[18:19], s[40:41], v[40:41], v[18:19] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[20:21], s[40:41], v[46:47], v[20:21] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[22:23], s[40:41], v[44:45], v[22:23] op_sel_hi:[0,1,1]\n\t"
        ".elseif %c[var] == 3\n\t"  // the same arithmetic as 16 v_fma_f32
        "v_fma_f32 v8, s40, v42, v8\n\tv_fma_f32 v9, s40, v43, v9\n\t"
        "v_fma_f32 v10, s40, v40, v10\n\tv_fma_f32 v11, s40, v41, v11\n\t"
        "v_fma_f32 v12, s40, v46, v12\n\tv_fma_f32 v13, s40, v47, v13\n\t"
        "v_fma_f32 v14, s40, v44, v14\n\tv_fma_f32 v15, s40, v45, v15\n\t"
        "v_fma_f32 v16, s40, v42, v16\n\tv_fma_f32 v17, s40, v43, v17\n\t"
        "v_fma_f32 v18, s40, v40, v18\n\tv_fma_f32 v19, s40, v41, v19\n\t"
        "v_fma_f32 v20, s40, v46, v20\n\tv_fma_f32 v21, s40, v47, v21\n\t"
        "v_fma_f32 v22, s40, v44, v22\n\tv_fma_f32 v23, s40, v45, v23\n\t"
        ".elseif %c[var] == 4\n\t"  // scalar coefficient picked from the high dword (op_sel), other banks
        "v_pk_fma_f32 v[8:9], s[40:41], v[42:43], v[8:9] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 v[10:11], s[40:41], v[40:41], v[10:11] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 v[12:13], s[40:41], v[46:47], v[12:13] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 v[14:15], s[40:41], v[44:45], v[14:15] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 v[16:17], s[40:41], v[42:43], v[16:17] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 v[18:19], s[40:41], v[40:41], v[18:19] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 v[20:21], s[40:41], v[46:47], v[20:21] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 v[22:23], s[40:41], v[44:45], v[22:23] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        ".elseif %c[var] == 5\n\t"  // v_pk_add_f32, other banks
        "v_pk_add_f32 v[8:9], v[8:9], v[42:43]\n\t"
        "v_pk_add_f32 v[10:11], v[10:11], v[40:41]\n\t"
        "v_pk_add_f32 v[12:13], v[12:13], v[46:47]\n\t"
        "v_pk_add_f32 v[14:15], v[14:15], v[44:45]\n\t"
        "v_pk_add_f32 v[16:17], v[16:17], v[42:43]\n\t"
        "v_pk_add_f32 v[18:19], v[18:19], v[40:41]\n\t"
        "v_pk_add_f32 v[20:21], v[20:21], v[46:47]\n\t"
        "v_pk_add_f32 v[22:23], v[22:23], v[44:45]\n\t"
        ".elseif %c[var] == 6\n\t"  // variant 2 with scalar work between the FMAs, as a sweep block has it (8 FMAs + 4 others)
        "v_pk_fma_f32 v[8:9], s[40:41], v[42:43], v[8:9] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[10:11], s[40:41], v[40:41], v[10:11] op_sel_hi:[0,1,1]\n\t"
        "s_add_u32 s37, s37, 1\n\t"
        "v_pk_fma_f32 v[12:13], s[40:41], v[46:47], v[12:13] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[14:15], s[40:41], v[44:45], v[14:15] op_sel_hi:[0,1,1]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_pk_fma_f32 v[16:17], s[40:41], v[42:43], v[16:17] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[18:19], s[40:41], v[40:41], v[18:19] op_sel_hi:[0,1,1]\n\t"
        "s_cmp_eq_u32 s37, 0\n\t"
        "v_pk_fma_f32 v[20:21], s[40:41], v[46:47], v[20:21] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[22:23], s[40:41], v[44:45], v[22:23] op_sel_hi:[0,1,1]\n\t"
        "s_cselect_b32 s38, 0, s37\n\t"
        ".elseif %c[var] == 7\n\t"  // variant 2 with LDS reads between the FMAs (3 per 8: the FIR8 plane block's ratio)
        "v_pk_fma_f32 v[8:9], s[40:41], v[42:43], v[8:9] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[10:11], s[40:41], v[40:41], v[10:11] op_sel_hi:[0,1,1]\n\t"
        "ds_read_b64 v[48:49], v56\n\t"
        "v_pk_fma_f32 v[12:13], s[40:41], v[46:47], v[12:13] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[14:15], s[40:41], v[44:45], v[14:15] op_sel_hi:[0,1,1]\n\t"
        "ds_read_b64 v[50:51], v56 offset:512\n\t"
        "v_pk_fma_f32 v[16:17], s[40:41], v[42:43], v[16:17] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[18:19], s[40:41], v[40:41], v[18:19] op_sel_hi:[0,1,1]\n\t"
        "ds_read_b64 v[52:53], v56 offset:1024\n\t"
        "v_pk_fma_f32 v[20:21], s[40:41], v[46:47], v[20:21] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[22:23], s[40:41], v[44:45], v[22:23] op_sel_hi:[0,1,1]\n\t"
        "s_waitcnt lgkmcnt(3)\n\t"
        ".elseif %c[var] == 9\n\t"  // the same bytes as variant 7 in one ds_read_b128 (16-byte aligned) + one ds_read_b64
        "v_pk_fma_f32 v[8:9], s[40:41], v[42:43], v[8:9] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[10:11], s[40:41], v[40:41], v[10:11] op_sel_hi:[0,1,1]\n\t"
        "ds_read_b128 v[48:51], v57\n\t"
        "v_pk_fma_f32 v[12:13], s[40:41], v[46:47], v[12:13] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[14:15], s[40:41], v[44:45], v[14:15] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[16:17], s[40:41], v[42:43], v[16:17] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[18:19], s[40:41], v[40:41], v[18:19] op_sel_hi:[0,1,1]\n\t"
        "ds_read_b64 v[52:53], v56 offset:4096\n\t"
        "v_pk_fma_f32 v[20:21], s[40:41], v[46:47], v[20:21] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[22:23], s[40:41], v[44:45], v[22:23] op_sel_hi:[0,1,1]\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        ".elseif %c[var] == 12\n\t"  // variant 9 with the ds_read_b128 only 8-byte aligned (an odd element offset)
        "v_pk_fma_f32 v[8:9], s[40:41], v[42:43], v[8:9] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[10:11], s[40:41], v[40:41], v[10:11] op_sel_hi:[0,1,1]\n\t"
        "ds_read_b128 v[48:51], v57 offset:8\n\t"
        "v_pk_fma_f32 v[12:13], s[40:41], v[46:47], v[12:13] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[14:15], s[40:41], v[44:45], v[14:15] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[16:17], s[40:41], v[42:43], v[16:17] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[18:19], s[40:41], v[40:41], v[18:19] op_sel_hi:[0,1,1]\n\t"
        "ds_read_b64 v[52:53], v56 offset:4096\n\t"
        "v_pk_fma_f32 v[20:21], s[40:41], v[46:47], v[20:21] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[22:23], s[40:41], v[44:45], v[22:23] op_sel_hi:[0,1,1]\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        ".elseif %c[var] == 10\n\t"  // four elements in two ds_read2_b64 (elements 512 bytes apart, as the sweeps' k registers are)
        "v_pk_fma_f32 v[8:9], s[40:41], v[42:43], v[8:9] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[10:11], s[40:41], v[40:41], v[10:11] op_sel_hi:[0,1,1]\n\t"
        "ds_read2_b64 v[48:51], v56 offset1:64\n\t"
        "v_pk_fma_f32 v[12:13], s[40:41], v[46:47], v[12:13] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[14:15], s[40:41], v[44:45], v[14:15] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[16:17], s[40:41], v[42:43], v[16:17] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[18:19], s[40:41], v[40:41], v[18:19] op_sel_hi:[0,1,1]\n\t"
        "ds_read2_b64 v[52:55], v56 offset0:128 offset1:192\n\t"
        "v_pk_fma_f32 v[20:21], s[40:41], v[46:47], v[20:21] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[22:23], s[40:41], v[44:45], v[22:23] op_sel_hi:[0,1,1]\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        ".elseif %c[var] == 11\n\t"  // the same four elements in four ds_read_b64
        "v_pk_fma_f32 v[8:9], s[40:41], v[42:43], v[8:9] op_sel_hi:[0,1,1]\n\t"
        "ds_read_b64 v[48:49], v56\n\t"
        "v_pk_fma_f32 v[10:11], s[40:41], v[40:41], v[10:11] op_sel_hi:[0,1,1]\n\t"
        "ds_read_b64 v[50:51], v56 offset:512\n\t"
        "v_pk_fma_f32 v[12:13], s[40:41], v[46:47], v[12:13] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[14:15], s[40:41], v[44:45], v[14:15] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[16:17], s[40:41], v[42:43], v[16:17] op_sel_hi:[0,1,1]\n\t"
        "ds_read_b64 v[52:53], v56 offset:1024\n\t"
        "v_pk_fma_f32 v[18:19], s[40:41], v[40:41], v[18:19] op_sel_hi:[0,1,1]\n\t"
        "ds_read_b64 v[54:55], v56 offset:1536\n\t"
        "v_pk_fma_f32 v[20:21], s[40:41], v[46:47], v[20:21] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[22:23], s[40:41], v[44:45], v[22:23] op_sel_hi:[0,1,1]\n\t"
        "s_waitcnt lgkmcnt(4)\n\t"
        ".elseif %c[var] == 8\n\t"  // variant 2 with one dependent pair per 8 (the same accumulator twice in a row)
        "v_pk_fma_f32 v[8:9], s[40:41], v[42:43], v[8:9] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[8:9], s[40:41], v[40:41], v[8:9] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[12:13], s[40:41], v[46:47], v[12:13] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[12:13], s[40:41], v[44:45], v[12:13] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[16:17], s[40:41], v[42:43], v[16:17] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[16:17], s[40:41], v[40:41], v[16:17] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[20:21], s[40:41], v[46:47], v[20:21] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[20:21], s[40:41], v[44:45], v[20:21] op_sel_hi:[0,1,1]\n\t"
        ".endif\n\t"
        ".endr\n\t"
        "s_sub_u32 s36, s36, 1\n\t"
        "s_cmp_lg_u32 s36, 0\n\t"
        "s_cbranch_scc1 .Lloop_%=\n\t"
        "v_add_f32 %[r], v8, v10\n\t"
        "v_add_f32 %[r], %[r], v12\n\t"
        "v_add_f32 %[r], %[r], v23\n\t"
        : [r] "=v"(r)
        : [s] "v"(seed), [n] "s"(REPS), [var] "n"(VAR), [tid] "v"(threadIdx.x & 63)
        : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23",
          "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "s36", "s37", "s38",
          "s40", "s41", "scc");
    if (r == 12345.0f) out[threadIdx.x] = r;
}

template <int VAR>
static void run(const char *what, int per_instr_flops, float *d_out) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int waves_per_simd : {1, 2, 4, 8}) {
        const int threads = 64 * 4 * (waves_per_simd > 4 ? 4 : waves_per_simd);  // one workgroup per CU (two for 8 waves)
        const int wgs = waves_per_simd > 4 ? 512 : 256;
        hipLaunchKernelGGL(rate_kernel<VAR>, dim3(wgs), dim3(threads), 0, 0, d_out, 0.0f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(rate_kernel<VAR>, dim3(wgs), dim3(threads), 0, 0, d_out, 0.0f);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        const double instr_per_simd = (double) REPS * 64 * waves_per_simd;
        const double cyc = ms * 1e-3 * 2.4e9 / instr_per_simd;
        const double tflops = instr_per_simd * 1024 * 64 * per_instr_flops / (ms * 1e-3) / 1e12;
        std::printf("%-62s %d waves/SIMD: %6.3f ms  %5.2f cyc/instr  %6.1f TFLOP/s\n", what, waves_per_simd, ms, cyc, tflops);
    }
}

int main() {
    float *d_out;
    CHECK(hipMalloc(&d_out, 4096));
    run<0>("v_pk_fma_f32 all-VGPR, sample/acc in different banks", 4, d_out);
    run<1>("v_pk_fma_f32 SGPR coeff, sample/acc in the SAME banks", 4, d_out);
    run<2>("v_pk_fma_f32 SGPR coeff, sample/acc in different banks", 4, d_out);
    run<4>("v_pk_fma_f32 SGPR coeff (high dword via op_sel), different banks", 4, d_out);
    run<3>("v_fma_f32 SGPR coeff (half the work per instruction)", 2, d_out);
    run<5>("v_pk_add_f32 different banks", 2, d_out);
    std::printf("-- the figures below count the packed FMAs only (8 per group) --\n");
    run<6>("8 v_pk_fma_f32 + 4 scalar instructions (add, waitcnt, cmp, cselect)", 4, d_out);
    run<7>("8 v_pk_fma_f32 + 3 ds_read_b64 + 1 counted waitcnt", 4, d_out);
    run<8>("8 v_pk_fma_f32, each accumulator twice in a row", 4, d_out);
    run<9>("8 v_pk_fma_f32 + ds_read_b128 + ds_read_b64 (the bytes of 3 b64)", 4, d_out);
    run<12>("8 v_pk_fma_f32 + ds_read_b128 at an 8-byte-aligned address + ds_read_b64", 4, d_out);
    run<11>("8 v_pk_fma_f32 + 4 ds_read_b64", 4, d_out);
    run<10>("8 v_pk_fma_f32 + 2 ds_read2_b64 (the same 4 elements)", 4, d_out);
    return 0;
}
