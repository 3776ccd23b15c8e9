// tr_test.hip -- semantics of ds_read_b64_tr_b16 and the v_mfma_f32_16x16x16_f16 operand layouts on the device.
// LDS holds lds[i] = i (fp16).  Every lane passes address (lane & 15) * 4 + (lane >> 4) * 64 (elements):
// expected (guide): lane l, elem j = lds[(l & 15) + j * 16 + (l >> 4) * 64].
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (_Float16)i;
    __syncthreads();
    const int l = threadIdx.x;
    fp16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
        (__attribute__((address_space(3))) fp16x4*)(lds + (l & 15) * 4 + (l >> 4) * 64));
    h4 v = __builtin_bit_cast(h4, t);
    // MFMA 16x16x16: A[m][k] = (m == 3 && k == 5) ? 1 : 0 ; B[k][n] = 100 k + n  ->  D[3][n] = 500 + n, rest 0
    h4 a, b;
    for (int r = 0; r < 4; ++r) {
        const int kk = 4 * (l >> 4) + r;
        a[r] = ((l & 15) == 3 && kk == 5) ? (_Float16)1 : (_Float16)0;     // A: row = l % 16, k = 4 (l / 16) + r
        b[r] = (_Float16)(100 * kk + (l & 15));                              // B: col = l % 16, k = 4 (l / 16) + r
    }
    f4 d = {0, 0, 0, 0};
    d = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) { out[l * 8 + r] = (float)v[r]; out[l * 8 + 4 + r] = d[r]; }
}
int main() {
    float* o; (void)hipMalloc(&o, 64 * 8 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o);
    float h[512]; (void)hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    int bad_tr = 0, bad_mfma = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) {
            if (h[l * 8 + j] != (float)((l & 15) + j * 16 + (l >> 4) * 64)) ++bad_tr;
            const int m = 4 * (l >> 4) + j, n = l & 15;          // D: row = 4 (l / 16) + j, col = l % 16
            if (h[l * 8 + 4 + j] != (m == 3 ? 500.f + n : 0.f)) ++bad_mfma;
        }
    for (int l : {0, 1, 17, 63}) printf("lane %2d: tr = %g %g %g %g | d = %g %g %g %g\n", l, h[l*8], h[l*8+1], h[l*8+2], h[l*8+3], h[l*8+4], h[l*8+5], h[l*8+6], h[l*8+7]);
    printf("tr_b16 layout mismatches: %d ; mfma 16x16x16 layout mismatches: %d\n", bad_tr, bad_mfma);
    return bad_tr || bad_mfma;
}
