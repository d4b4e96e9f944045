#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *out) {
    unsigned lane = threadIdx.x;
    unsigned a = 1000 + lane, b = 2000 + lane;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[lane] = r[0]; out[64 + lane] = r[1];
    auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[128 + lane] = q[0]; out[192 + lane] = q[1];
}
int main() {
    unsigned *d; hipMalloc(&d, 256 * 4);
    k<<<1, 64>>>(d);
    unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *nm[4] = {"swap32 vdst", "swap32 src ", "swap16 vdst", "swap16 src "};
    for (int r = 0; r < 4; ++r) { printf("%s:", nm[r]); for (int i = 0; i < 64; i += 4) printf(" %u", h[64 * r + i]); printf("\n"); }
    return 0;
}
