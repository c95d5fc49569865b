// What the f16 matrix pipe of an MI355X sustains under the conditions of the split-f16 kernels (256 threads, 2 workgroups per CU, 8 independent
// accumulators, random operands -- the chip clocks to its power budget, so zeros would flatter it): MFMAs alone, + the LDS operand reads of the
// convolution / query loops (8 ds_read_b128 per 24 MFMAs), + the weight-fragment loads (4 x 16 B per lane per 24 MFMAs from an L2-resident table), + both.
// build: hipcc -O3 --offload-arch=gfx950 mfma_roof.hip -o mfma_roof ; run: ./mfma_roof
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int V> __global__ __launch_bounds__(256, 2) void roof(const uint4 *__restrict__ wtab, int wmask, int iters, float *__restrict__ out)
{
    __shared__ uint4 plane[2 * 4 * 192];                    // hi + lo planes of a 32-channel chunk, 192 pixels
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, j = lane & 15;
    for (int i = tid; i < 2 * 4 * 192; i += 256) { unsigned s = (i * 2654435761u + blockIdx.x) | 1u; plane[i] = make_uint4(0x2c002e00u ^ (s & 0x83ff83ffu), 0x2d002b00u ^ ((s >> 3) & 0x83ff83ffu), 0x2a002c80u ^ ((s >> 5) & 0x83ff83ffu), 0x2e402d40u ^ ((s >> 7) & 0x83ff83ffu)); }
    __syncthreads();
    f32x4 acc[8];
    for (int k = 0; k < 8; k++) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint4 w[4], x[8];
    for (int k = 0; k < 4; k++) w[k] = plane[(lane + 64 * k) % 1536];
    for (int k = 0; k < 8; k++) x[k] = plane[(lane * 3 + 64 * k) % 1536];
    const uint4 *wp = wtab + wave * 256 + lane;
    unsigned off = blockIdx.x * 4096u;
    for (int it = 0; it < iters; it++) {
        if (V & 2) {
#pragma unroll
            for (int k = 0; k < 4; k++) w[k] = wp[((off + 1024u * it) & wmask) + 64 * k];
        }
        if (V & 1) {
#pragma unroll
            for (int k = 0; k < 4; k++) { x[k] = plane[q * 192 + ((it + k * 18) % 160) + j]; x[4 + k] = plane[4 * 192 + q * 192 + ((it + k * 18) % 160) + j]; }
        }
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int p = 0; p < 4; p++) acc[4 * n + p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, w[2 * n]), __builtin_bit_cast(h8, x[p]), acc[4 * n + p], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int p = 0; p < 4; p++) acc[4 * n + p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, w[2 * n]), __builtin_bit_cast(h8, x[4 + p]), acc[4 * n + p], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int p = 0; p < 4; p++) acc[4 * n + p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, w[2 * n + 1]), __builtin_bit_cast(h8, x[p]), acc[4 * n + p], 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < 8; k++) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256, 2) void roof32(int iters, float *__restrict__ out)
{
    __shared__ uint4 plane[1536];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 1536; i += 256) { unsigned s = (i * 2654435761u + blockIdx.x) | 1u; plane[i] = make_uint4(0x2c002e00u ^ (s & 0x83ff83ffu), 0x2d002b00u ^ ((s >> 3) & 0x83ff83ffu), 0x2a002c80u ^ ((s >> 5) & 0x83ff83ffu), 0x2e402d40u ^ ((s >> 7) & 0x83ff83ffu)); }
    __syncthreads();
    f32x16 acc[4];
    for (int k = 0; k < 4; k++) for (int r = 0; r < 16; r++) acc[k][r] = 0.f;
    uint4 w[4], x[8];
    for (int k = 0; k < 4; k++) w[k] = plane[(lane + 64 * k) % 1536];
    for (int k = 0; k < 8; k++) x[k] = plane[(lane * 3 + 64 * k) % 1536];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 3; m++)
#pragma unroll
            for (int p = 0; p < 4; p++) acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, w[m]), __builtin_bit_cast(h8, x[(p + 4 * (m == 1)) & 7]), acc[p], 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < 4; k++) for (int r = 0; r < 16; r++) s += acc[k][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}
template <int V> void run(const char *name, const uint4 *wtab, int wmask, float *out)
{
    const int blocks = 256 * 2 * 4, iters = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(roof<V>, dim3(blocks), dim3(256), 0, 0, wtab, wmask, iters, out);
    CK(hipEventRecord(e0));
    for (int k = 0; k < 5; k++) hipLaunchKernelGGL(roof<V>, dim3(blocks), dim3(256), 0, 0, wtab, wmask, iters, out);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double fl = (double)blocks * 4 * iters * 24 * 16384.0;
    printf("%-44s %.3f ms  %.0f TFLOP/s of f16 MFMA = %.3f of 2516.6 (= %.0f TFLOP/s of split-f16 work)\n", name, ms, fl / ms * 1e-9, fl / ms * 1e-9 / 2516.6, fl / ms * 1e-9 / 3);
}
int main()
{
    const size_t wn = 1 << 16;      // 1 MB of weight fragments
    std::vector<unsigned> h(wn * 4);
    srand(1);
    for (auto &v : h) { unsigned r = rand(); v = 0x2c002c00u ^ (r & 0x83ff83ffu); }       // fp16 pairs of magnitude ~0.06, random mantissas and signs
    uint4 *wtab; float *out;
    CK(hipMalloc(&wtab, wn * 16)); CK(hipMemcpy(wtab, h.data(), wn * 16, hipMemcpyHostToDevice)); CK(hipMalloc(&out, (size_t)2048 * 256 * 4));
    {
        const int blocks = 2048, iters = 2000; hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(roof32, dim3(blocks), dim3(256), 0, 0, iters, out);
        CK(hipEventRecord(e0));
        for (int k = 0; k < 5; k++) hipLaunchKernelGGL(roof32, dim3(blocks), dim3(256), 0, 0, iters, out);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        const double fl = (double)blocks * 4 * iters * 12 * 32768.0;
        printf("%-44s %.3f ms  %.0f TFLOP/s of f16 MFMA = %.3f of 2516.6\n", "MFMA only, 32x32x16", ms, fl / ms * 1e-9, fl / ms * 1e-9 / 2516.6);
    }
    run<0>("MFMA only", wtab, (int)wn - 1024, out);
    run<1>("+ 8 ds_read_b128 per 24 MFMAs", wtab, (int)wn - 1024, out);
    run<2>("+ 4 weight-fragment loads per 24 MFMAs", wtab, (int)wn - 1024, out);
    run<3>("+ both", wtab, (int)wn - 1024, out);
    return 0;
}
