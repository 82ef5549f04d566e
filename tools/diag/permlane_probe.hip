// prints what v_permlane32_swap / v_permlane16_swap do to lane ids (probe for attention_mix.hip's transpose4)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* o) {
    const unsigned l = threadIdx.x;
    unsigned a = 1000 + l, b = 2000 + l;
    auto p = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[l] = p[0]; o[64 + l] = p[1];
    auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[128 + l] = q[0]; o[192 + l] = q[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4);
    k<<<1, 64>>>(d);
    unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"swap32 ret[0]", "swap32 ret[1]", "swap16 ret[0]", "swap16 ret[1]"};
    for (int t = 0; t < 4; ++t) { printf("%s:", names[t]); for (int g = 0; g < 4; ++g) printf("  row%d=%u", g, h[t * 64 + g * 16]); printf("\n"); }
    return 0;
}
