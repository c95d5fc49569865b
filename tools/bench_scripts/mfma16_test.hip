// layout probe for v_mfma_f32_32x32x16_f16 and v_mfma_f32_16x16x32_f16 on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// A [32][16] row-major, B [16][32] row-major (k, j), out [32][32]
__global__ void t32(const float *A, const float *B, float *out)
{
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    h8 a, b;
    for (int t = 0; t < 8; t++) { a[t] = (_Float16)A[i * 16 + 8 * h + t]; b[t] = (_Float16)B[(8 * h + t) * 32 + i]; }
    f32x16 c; for (int r = 0; r < 16; r++) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; r++) { const int row = 8 * (r >> 2) + 4 * h + (r & 3); out[row * 32 + i] = c[r]; }
}
// A [16][32], B [32][16], out [16][16]
__global__ void t16(const float *A, const float *B, float *out)
{
    const int l = threadIdx.x, i = l & 15, q = l >> 4;
    h8 a, b;
    for (int t = 0; t < 8; t++) { a[t] = (_Float16)A[i * 32 + 8 * q + t]; b[t] = (_Float16)B[(8 * q + t) * 16 + i]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) out[(4 * q + r) * 16 + i] = c[r];
}
int main()
{
    float hA[512], hB[512], hO[1024], *dA, *dB, *dO;
    for (int i = 0; i < 512; i++) { hA[i] = (float)((rand() % 17) - 8) / 8.f; hB[i] = (float)((rand() % 13) - 6) / 4.f; }
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dO, 4096);
    hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
    t32<<<1, 64>>>(dA, dB, dO); hipMemcpy(hO, dO, 4096, hipMemcpyDeviceToHost);
    double e = 0; for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { double s = 0; for (int k = 0; k < 16; k++) s += hA[i * 16 + k] * hB[k * 32 + j]; e = fmax(e, fabs(s - hO[i * 32 + j])); }
    printf("32x32x16 max err %g\n", e);
    t16<<<1, 64>>>(dA, dB, dO); hipMemcpy(hO, dO, 1024, hipMemcpyDeviceToHost);
    e = 0; for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0; for (int k = 0; k < 32; k++) s += hA[i * 32 + k] * hB[k * 16 + j]; e = fmax(e, fabs(s - hO[i * 16 + j])); }
    printf("16x16x32 max err %g\n", e);
    return 0;
}
