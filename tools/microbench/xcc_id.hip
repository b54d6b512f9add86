// What does s_getreg_b32 hwreg(HW_REG_XCC_ID) return per workgroup, and how do consecutive workgroup ids map to XCDs?
// (das_exact_nd_kernel / das_quad_kernel pick their item queue by it.)   hipcc --offload-arch=gfx950 -O2 xcc_id.hip -o xcc_id
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = x;
}
int main() {
    const int n = 64;
    unsigned *d, h[n];
    hipMalloc(&d, n * sizeof(unsigned));
    hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) printf("%s%08x", i % 8 ? " " : "\n", h[i]);
    printf("\n");
    return 0;
}
