// Does a co-running small kernel cost the query kernel its DURATION or only what it displaces?  Two query workgroups fill a CU's register file (2 x 240 of the 512
// registers per SIMD lane): a wave of another kernel that needs more than the 32 registers left cannot be placed beside them -- it waits for a query workgroup to retire
// and takes its slot.  spin<N> holds N VGPRs and spins `clocks` shader clocks; coreside.py runs it on a second stream beside back-to-back query launches.
// build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC coreside.hip -o coreside.so
#include <hip/hip_runtime.h>
template <int HI> __global__ __launch_bounds__(256) void spin(long long clocks, float *sink)
{
    const long long t0 = clock64();
    float x = threadIdx.x;
    if (HI == 24) asm volatile("v_mov_b32 v23, 1.0" ::: "v23");
    if (HI == 64) asm volatile("v_mov_b32 v63, 1.0" ::: "v63");
    if (HI == 128) asm volatile("v_mov_b32 v127, 1.0" ::: "v127");
    while (clock64() - t0 < clocks) x = x * 1.0001f + 1.f;
    if (x == 12345.f) sink[0] = x;
}
extern "C" int cs_spin(int vgprs, int blocks, long long clocks, float *sink, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (vgprs <= 24) hipLaunchKernelGGL(spin<24>, dim3(blocks), dim3(256), 0, st, clocks, sink);
    else if (vgprs <= 64) hipLaunchKernelGGL(spin<64>, dim3(blocks), dim3(256), 0, st, clocks, sink);
    else hipLaunchKernelGGL(spin<128>, dim3(blocks), dim3(256), 0, st, clocks, sink);
    return (int)hipGetLastError();
}

// a stream whose kernels may only run on the CUs of `mask` (8 x 32 bits; hipExtStreamCreateWithCUMask): packs the co-running kernels on few CUs
extern "C" void *cs_masked_stream(const unsigned *mask, int words)
{
    hipStream_t st = nullptr;
    if (hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask) != hipSuccess) return nullptr;
    return (void *)st;
}
