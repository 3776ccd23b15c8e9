#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out) {
    const int l = threadIdx.x;
    float v = (float)(l * l + 1);
    const unsigned u = __builtin_bit_cast(unsigned, v);
    unsigned r16[2] = {u, u}, r32[2] = {u, u};
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(r16[0]), "+v"(r16[1]));
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(r32[0]), "+v"(r32[1]));
    out[l] = __builtin_bit_cast(float, r16[0]);
    out[64 + l] = __builtin_bit_cast(float, r16[1]);
    out[128 + l] = __builtin_bit_cast(float, r32[0]);
    out[192 + l] = __builtin_bit_cast(float, r32[1]);
    out[256 + l] = v + __shfl_xor(v, 16);
    out[320 + l] = v + __shfl_xor(v, 32);
}
int main() {
    float* d; hipMalloc(&d, 384 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[384]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad16 = 0, bad32 = 0;
    for (int l = 0; l < 64; ++l) {
        if (h[l] + h[64 + l] != h[256 + l]) ++bad16;
        if (h[128 + l] + h[192 + l] != h[320 + l]) ++bad32;
    }
    printf("bad16=%d bad32=%d\n", bad16, bad32);
    for (int l : {0, 5, 16, 21, 32, 37, 48, 53}) printf("lane %2d: v=%g r16=(%g,%g) r32=(%g,%g) want16=%g want32=%g\n", l, (float)(l*l+1), h[l], h[64+l], h[128+l], h[192+l], h[256+l], h[320+l]);
    return 0;
}
