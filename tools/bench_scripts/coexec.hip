// How much VALU work hides behind v_mfma_f32_16x16x32_f16 / v_mfma_f32_32x32x16_f16 on an MI355X SIMD -- inside ONE wave (k independent VALU instructions
// placed after every MFMA) and between the TWO waves of a SIMD (one wave issues only MFMAs, its partner only VALU).  The query kernels spend 34 % of
// their SIMD time in the matrix pipe and 28 % in plain VALU with almost no overlap (SQ_VALU_MFMA_COEXEC_CYCLES 6 %): this measures what the hardware
// allows, so that the kernel structure can be chosen for it.
// build: hipcc -O3 --offload-arch=gfx950 coexec.hip -o coexec.bin ; run: ./coexec.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define MF16(acc_) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc_) : "v"(a), "v"(b))
#define MF32(acc_) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc_) : "v"(a), "v"(b))
#define VA(x_) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x_) : "v"(k0), "v"(k1))
// other VALU flavours of the epilogues: packed conversion, mixed-precision fma, bit-field extract
#define VC(x_) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x_) : "v"(k1))
#define VB(x_) asm volatile("v_bfe_i32 %0, %0, 3, 1" : "+v"(x_))

// MODE: 0 = MFMA only, 1 = VALU only (V per group), 2 = per MFMA V independent VALU instructions, 3 = wave halves: waves 0-3 MFMA only, waves 4-7 VALU only (512 threads)
template <int M32, int V, int MODE>
__global__ __launch_bounds__(512, 1) void k(int iters, float *out, unsigned long long *clk)
{
    const int tid = threadIdx.x, wave = tid >> 6;
    h8 a, b;
    for (int t = 0; t < 8; t++) { a[t] = (_Float16)(0.5f + 0.001f * ((tid * 7 + t) & 63)); b[t] = (_Float16)(0.25f + 0.002f * ((tid * 3 + t) & 31)); }
    f32x4 c16[8]; f32x16 c32[4];
    for (int i = 0; i < 8; i++) c16[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) c32[i][r] = 0.f;
    float x[8]; for (int i = 0; i < 8; i++) x[i] = 1.0f + tid * 1e-3f + i;
    const float k0 = 0.999f, k1 = 1e-3f;
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4), do_v = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
    __syncthreads();
    const unsigned long long t0 = clock64();
    if (MODE == 3) {
        if (do_m) {
            for (int it = 0; it < iters; it++) {
                if (M32) { MF32(c32[0]); MF32(c32[1]); MF32(c32[2]); MF32(c32[3]); }
                else { MF16(c16[0]); MF16(c16[1]); MF16(c16[2]); MF16(c16[3]); MF16(c16[4]); MF16(c16[5]); MF16(c16[6]); MF16(c16[7]); }
            }
        } else {
            // V VALU per 16 cycles of the partner's matrix time: 8 groups of V per iteration (M32: 4 groups of 2 V)
            for (int it = 0; it < iters; it++) {
#pragma unroll
                for (int g = 0; g < 8; g++)
#pragma unroll
                    for (int v = 0; v < V; v++) VA(x[(g + v) & 7]);
            }
        }
    } else {
        for (int it = 0; it < iters; it++) {
            if (M32) {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    if (do_m) MF32(c32[g]);
                    if (do_v) {
#pragma unroll
                        for (int v = 0; v < 2 * V; v++) VA(x[(2 * g + v) & 7]);
                    }
                }
            } else {
#pragma unroll
                for (int g = 0; g < 8; g++) {
                    if (do_m) MF16(c16[g]);
                    if (do_v) {
#pragma unroll
                        for (int v = 0; v < V; v++) VA(x[(g + v) & 7]);
                    }
                }
            }
        }
    }
    const unsigned long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 8; i++) s += c16[i][0] + c16[i][1] + c16[i][2] + c16[i][3] + x[i];
    for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) s += c32[i][r];
    out[(size_t)blockIdx.x * blockDim.x + tid] = s;
    if ((tid & 63) == 0) atomicAdd(clk + (wave >= 4), t1 - t0);
}

template <int M32, int V, int MODE>
void run(const char *name, int threads, int wg_per_cu, float *out, unsigned long long *clk)
{
    const int iters = 4000, blocks = 256 * wg_per_cu;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<M32, V, MODE>), dim3(blocks), dim3(threads), 0, 0, 100, out, clk);
    CK(hipMemset(clk, 0, 16));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<M32, V, MODE>), dim3(blocks), dim3(threads), 0, 0, iters, out, clk);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    const int waves = threads / 64;
    const double nw0 = (double)blocks * (MODE == 3 ? 4 : waves), nw1 = (double)blocks * (MODE == 3 ? 4 : 0);
    // per "group" = one 16-cycle matrix slot (a 32x32x16 counts as two)
    const double groups = (double)iters * 8;
    printf("%-64s %2d waves/SIMD  %8.3f ms  clk/group: %6.2f", name, threads * wg_per_cu / 256, ms, h[0] / nw0 / groups);
    if (MODE == 3) printf("  (VALU waves %6.2f)", h[1] / nw1 / groups);
    printf("\n");
}

int main()
{
    float *out; unsigned long long *clk;
    CK(hipMalloc(&out, sizeof(float) * 256 * 4 * 512)); CK(hipMalloc(&clk, 16));
    printf("clk/group = shader clocks (s_memtime) per 16-cycle matrix slot of one wave's stream; a group = 1 MFMA16 (+ V VALU) or half an MFMA32 (+ V VALU)\n");
    printf("--- one wave per SIMD (256 threads, 1 WG / CU)\n");
    run<0, 0, 0>("MFMA 16x16x32 only", 256, 1, out, clk);
    run<0, 1, 2>("MFMA 16x16x32 + 1 VALU each", 256, 1, out, clk);
    run<0, 2, 2>("MFMA 16x16x32 + 2 VALU each", 256, 1, out, clk);
    run<0, 3, 2>("MFMA 16x16x32 + 3 VALU each", 256, 1, out, clk);
    run<0, 4, 2>("MFMA 16x16x32 + 4 VALU each", 256, 1, out, clk);
    run<0, 6, 2>("MFMA 16x16x32 + 6 VALU each", 256, 1, out, clk);
    run<0, 4, 1>("4 VALU per group only", 256, 1, out, clk);
    run<1, 0, 0>("MFMA 32x32x16 only", 256, 1, out, clk);
    run<1, 2, 2>("MFMA 32x32x16 + 4 VALU each (2 per 16-cycle slot)", 256, 1, out, clk);
    run<1, 3, 2>("MFMA 32x32x16 + 6 VALU each (3 per slot)", 256, 1, out, clk);
    run<1, 4, 2>("MFMA 32x32x16 + 8 VALU each (4 per slot)", 256, 1, out, clk);
    printf("--- two waves per SIMD, both the same stream (256 threads, 2 WG / CU)\n");
    run<0, 0, 0>("MFMA 16x16x32 only", 256, 2, out, clk);
    run<0, 1, 2>("MFMA 16x16x32 + 1 VALU each", 256, 2, out, clk);
    run<0, 2, 2>("MFMA 16x16x32 + 2 VALU each", 256, 2, out, clk);
    run<0, 3, 2>("MFMA 16x16x32 + 3 VALU each", 256, 2, out, clk);
    run<0, 4, 2>("MFMA 16x16x32 + 4 VALU each", 256, 2, out, clk);
    run<0, 4, 1>("4 VALU per group only", 256, 2, out, clk);
    run<1, 2, 2>("MFMA 32x32x16 + 4 VALU each (2 per slot)", 256, 2, out, clk);
    run<1, 4, 2>("MFMA 32x32x16 + 8 VALU each (4 per slot)", 256, 2, out, clk);
    printf("--- two waves per SIMD, ROLES split (512 threads, 1 WG / CU): waves 0-3 MFMA only, waves 4-7 VALU only (V per matrix slot)\n");
    run<0, 1, 3>("MFMA16 wave || VALU wave, 1 VALU per slot", 512, 1, out, clk);
    run<0, 2, 3>("MFMA16 wave || VALU wave, 2 VALU per slot", 512, 1, out, clk);
    run<0, 3, 3>("MFMA16 wave || VALU wave, 3 VALU per slot", 512, 1, out, clk);
    run<0, 4, 3>("MFMA16 wave || VALU wave, 4 VALU per slot", 512, 1, out, clk);
    run<1, 2, 3>("MFMA32 wave || VALU wave, 2 VALU per slot", 512, 1, out, clk);
    run<1, 4, 3>("MFMA32 wave || VALU wave, 4 VALU per slot", 512, 1, out, clk);
    return 0;
}
