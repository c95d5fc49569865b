#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
__global__ void bad(const h8 *a, const h8 *b, float *out)
{
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
    float y;
    asm("v_max_f32 %0, %1, 0" : "=v"(y) : "v"(c[0]));       // first reader of the accumulator is inline asm: the hazard the rule forbids
    out[threadIdx.x] = y + c[1];
}
__global__ void good(const h8 *a, const h8 *b, float *out)
{
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
    float y = fmaxf(c[0], 0.f), z;
    asm("v_max_f32 %0, %1, 0" : "=v"(z) : "v"(y));
    out[threadIdx.x] = z + c[1];
}
