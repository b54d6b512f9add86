// What v_permlane32_swap_b32 / v_permlane16_swap_b32 (gfx950) do to two registers, printed lane by lane, and the joint
// reduction of eight per-lane partial sums built on them (das_fast.hip: wave_sum8) against eight plain wave sums.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/permlane_swap_check.hip -o tools/microbench/permlane_swap_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void swaps(unsigned *o) {
    const unsigned l = threadIdx.x;
    unsigned a = 100 + l, b = 200 + l;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[l] = r[0];
    o[64 + l] = r[1];
    a = 100 + l;
    b = 200 + l;
    auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[128 + l] = q[0];
    o[192 + l] = q[1];
}

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_take(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float fold_halves(float a, float b) {
    // (inline asm, not __builtin_amdgcn_permlane32_swap: hipcc 7.2 folds `r[0] + r[1]` of the builtin's two results into
    // `r[0] + r[0]` -- tools/microbench/permlane_swap_check.hip caught it; the nops cover the VALU-write -> swap-read hazard)
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float fold_rows(float a, float b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float wave_sum8(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7, int lane) {
    const float x = fold_rows(fold_halves(v0, v1), fold_halves(v2, v3));
    const float y = fold_rows(fold_halves(v4, v5), fold_halves(v6, v7));
    const float xs = x + dpp_take<0x128>(x), ys = y + dpp_take<0x128>(y);
    float z = (lane & 8) ? ys : xs;
    z += dpp_take<0xB1>(z);
    z += dpp_take<0x4E>(z);
    z += dpp_take<0x141>(z);
    return z;
}
__device__ __forceinline__ float wave_sum4(float v0, float v1, float v2, float v3) {
    float z = fold_rows(fold_halves(v0, v1), fold_halves(v2, v3));
    z += dpp_take<0xB1>(z);
    z += dpp_take<0x4E>(z);
    z += dpp_take<0x141>(z);
    z += dpp_take<0x140>(z);
    return z;
}
__global__ void sums4(float *o, const float *in) {
    const int l = threadIdx.x;
    o[l] = wave_sum4(in[l], in[64 + l], in[128 + l], in[192 + l]);
}
__global__ void sums(float *o, const float *in) {  // in [8][64] -> o [64]: wave_sum8; o[64 + k]: stages
    const int l = threadIdx.x;
    float v[8];
    for (int k = 0; k < 8; k++) v[k] = in[64 * k + l];
    o[l] = wave_sum8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], l);
    o[64 + l] = fold_halves(v[0], v[1]);
    o[128 + l] = fold_rows(fold_halves(v[0], v[1]), fold_halves(v[2], v[3]));
    const float x = o[128 + l];
    o[192 + l] = x + dpp_take<0x128>(x);
}

int main() {
    {
        float hin[512], hout[256], *din, *dout;
        for (int k = 0; k < 8; k++)
            for (int l = 0; l < 64; l++) hin[64 * k + l] = (float) ((k + 1) * 1000 + l);  // sum of value k = 64000 (k+1) + 2016
        hipMalloc(&din, sizeof(hin));
        hipMalloc(&dout, sizeof(hout));
        hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(sums, dim3(1), dim3(64), 0, 0, dout, din);
        hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost);
        std::printf("wave_sum8 per 8-lane group (expect value k -> 64000 (k+1) + 2016):");
        for (int g = 0; g < 8; g++) std::printf(" g%d: %.0f..%.0f", g, hout[8 * g], hout[8 * g + 7]);
        hipLaunchKernelGGL(sums4, dim3(1), dim3(64), 0, 0, dout, din);
        float h4[64];
        hipMemcpy(h4, dout, sizeof(h4), hipMemcpyDeviceToHost);
        std::printf("\nwave_sum4 per row of 16 lanes (expect rows = values 0, 2, 1, 3):");
        for (int r = 0; r < 4; r++) std::printf(" r%d: %.0f..%.0f", r, h4[16 * r], h4[16 * r + 15]);
        std::printf("\nfold_halves(v0, v1) lanes 0, 31, 32, 63: %.0f %.0f %.0f %.0f (expect 2*1000+0+32=2032.., v1: 4032..)\n", hout[64], hout[95], hout[96], hout[127]);
        std::printf("fold_rows rows 0..3 lane 0 of each: %.0f %.0f %.0f %.0f\n", hout[128], hout[144], hout[160], hout[176]);
        std::printf("x + row_ror8 lanes 0, 8 of row 0: %.0f %.0f\n", hout[192], hout[200]);
    }
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(swaps, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[4] = {"permlane32_swap vdst", "permlane32_swap src ", "permlane16_swap vdst", "permlane16_swap src "};
    for (int k = 0; k < 4; k++) {
        std::printf("%s:", names[k]);
        for (int l = 0; l < 64; l += 8) std::printf(" [%u..%u]", h[64 * k + l], h[64 * k + l + 7]);
        std::printf("\n");
    }
    return 0;
}
