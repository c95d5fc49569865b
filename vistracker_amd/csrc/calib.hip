// calib.hip -- BOX CALIBRATION for the bench line: MI355X boxes of the pool differ by +-4 % in what the dominant kernel of the fit
// (query.hip) takes, and a bench line that carries only the kernel's time cannot tell a faster kernel from a faster box.  vt_calibrate runs two fixed
// micro-kernels that exercise the two resources query_kernel<2, MODE_HUMAN> is limited by and returns their rates next to the clock the chip sustained:
//   (1) the f16 matrix pipe under the kernel's own conditions (256 threads, two workgroups per CU, v_mfma_f32_16x16x32_f16 with eight independent
//       accumulators, NON-trivial operands -- the chip clocks to its power budget, zeros would flatter it): TFLOP/s, and shader clocks (s_memtime) against the
//       constant 100 MHz counter (s_memrealtime) -> the sustained shader clock;
//   (2) the vector-memory path from L2 to the registers with the kernel's dominant pattern (lane-linear 16-byte buffer loads, 1 KB per wave and instruction,
//       from an L2-resident table): TB/s delivered.
// Nothing of the reference is replaced here: test / measurement infrastructure behind the C ABI (the reference has no counterpart), ~25 ms per call.
#include "common.h"

typedef _Float16 h8c __attribute__((ext_vector_type(8)));
typedef float f32x4c __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void calib_mfma_kernel(int iters, float *__restrict__ sink, unsigned long long *__restrict__ clk)
{
    __shared__ uint4 plane[1024];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 1024; i += 256) {
        const unsigned s = (i * 2654435761u + blockIdx.x) | 1u;      // halves of magnitude ~0.05 .. 0.1 with mixed signs
        plane[i] = make_uint4(0x2c002e00u ^ (s & 0x83ff83ffu), 0x2d002b00u ^ ((s >> 3) & 0x83ff83ffu), 0x2a002c80u ^ ((s >> 5) & 0x83ff83ffu), 0x2e402d40u ^ ((s >> 7) & 0x83ff83ffu));
    }
    __syncthreads();
    f32x4c acc[8];
    for (int k = 0; k < 8; k++) acc[k] = (f32x4c){0.f, 0.f, 0.f, 0.f};
    uint4 w[4], x[8];
    for (int k = 0; k < 4; k++) w[k] = plane[(lane + 64 * k) & 1023];
    for (int k = 0; k < 8; k++) x[k] = plane[(lane * 3 + 64 * k) & 1023];
    const unsigned long long c0 = clock64(), r0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 3; m++)
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int p = 0; p < 4; p++)
                    acc[4 * n + p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8c, w[(2 * n + (m == 2)) & 3]), __builtin_bit_cast(h8c, x[(p + 4 * (m == 1)) & 7]), acc[4 * n + p], 0, 0, 0);
    }
    const unsigned long long c1 = clock64(), r1 = wall_clock64();
    float s = 0.f;
    for (int k = 0; k < 8; k++) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    sink[(size_t)blockIdx.x * 256 + tid] = s;
    if (tid == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

__global__ __launch_bounds__(256, 2) void calib_gather_kernel(const uint4 *__restrict__ table, unsigned mask, int iters, float *__restrict__ sink)
{
    // every wave streams 1 KB pieces of the table (uint4 per lane, lane-linear) from pseudo-random piece positions: 8 independent loads in flight per lane
    const int tid = threadIdx.x;
    unsigned pos = (blockIdx.x * 2654435761u + (tid >> 6) * 40503u) & mask;
    uint4 a = make_uint4(0u, 0u, 0u, 0u);
    for (int it = 0; it < iters; it++) {
        uint4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { v[k] = table[((pos + 64u * 97u * k) & mask) + (tid & 63)]; }
        pos = (pos * 1664525u + 1013904223u) & mask & ~63u;
#pragma unroll
        for (int k = 0; k < 8; k++) { a.x ^= v[k].x; a.y += v[k].y; a.z ^= v[k].z; a.w += v[k].w; }
    }
    sink[(size_t)blockIdx.x * 256 + tid] = __uint_as_float((a.x ^ a.y ^ a.z ^ a.w) & 0x3fffffffu);
}

// out[0] = f16 MFMA TFLOP/s (dense, 16x16x32), out[1] = sustained shader clock in MHz during (1), out[2] = L2 -> register delivery in TB/s,
// out[3] = milliseconds of (1), out[4] = milliseconds of (2).  `work`: >= vt_calibrate_workspace_bytes() bytes of device memory.  Synchronises the stream.
extern "C" long vt_calibrate_workspace_bytes(void) { return (4L << 20) + 2048L * 256 * 4 + 2048L * 16 + 256; }
extern "C" int vt_calibrate(void *work, double *out, void *stream)
{
    VT_REQUIRE(work && out, "vt_calibrate: null argument");
    hipStream_t st = vt_stream(stream);
    unsigned char *base = static_cast<unsigned char *>(work);
    uint4 *table = reinterpret_cast<uint4 *>(base);                          // 4 MB: L2 resident on every XCD
    float *sink = reinterpret_cast<float *>(base + (4L << 20));
    unsigned long long *clk = reinterpret_cast<unsigned long long *>(base + (4L << 20) + 2048L * 256 * 4);
    VT_HIP(hipMemsetAsync(table, 0x3c, 4L << 20, st));
    hipEvent_t e[4];
    for (auto &x : e) VT_HIP(hipEventCreate(&x));
    const int blocks = 2048, it_m = 6000, it_g = 3000;       // ~10 ms + ~7 ms: long enough for the clock to settle under the load (2.5 ms kernels saw it ramp)
    hipLaunchKernelGGL(calib_mfma_kernel, dim3(blocks), dim3(256), 0, st, 3000, sink, clk);          // warm-up (clocks ramp)
    hipLaunchKernelGGL(calib_gather_kernel, dim3(blocks), dim3(256), 0, st, table, (unsigned)((4u << 20) / 16 - 1) & ~63u, 50, sink);
    VT_HIP(hipEventRecord(e[0], st));
    hipLaunchKernelGGL(calib_mfma_kernel, dim3(blocks), dim3(256), 0, st, it_m, sink, clk);
    VT_HIP(hipEventRecord(e[1], st));
    VT_HIP(hipEventRecord(e[2], st));
    hipLaunchKernelGGL(calib_gather_kernel, dim3(blocks), dim3(256), 0, st, table, (unsigned)((4u << 20) / 16 - 1) & ~63u, it_g, sink);
    VT_HIP(hipEventRecord(e[3], st));
    VT_LAUNCH_CHECK();
    VT_HIP(hipStreamSynchronize(st));
    float ms_m = 0.f, ms_g = 0.f;
    VT_HIP(hipEventElapsedTime(&ms_m, e[0], e[1])); VT_HIP(hipEventElapsedTime(&ms_g, e[2], e[3]));
    for (auto &x : e) (void)hipEventDestroy(x);
    static unsigned long long host_clk[2 * 2048];
    VT_HIP(hipMemcpy(host_clk, clk, sizeof(host_clk), hipMemcpyDeviceToHost));
    double sc = 0, rc = 0;
    for (int i = 0; i < blocks; i++) { sc += (double)host_clk[2 * i]; rc += (double)host_clk[2 * i + 1]; }
    out[0] = (double)blocks * 4 * it_m * 24 * 16384.0 / (ms_m * 1e-3) * 1e-12;
    out[1] = rc > 0 ? sc / rc * 100.0 : 0.0;                                  // s_memrealtime counts at 100 MHz
    out[2] = (double)blocks * 256 * it_g * 8 * 16.0 / (ms_g * 1e-3) * 1e-12;
    out[3] = ms_m; out[4] = ms_g;
    return VT_OK;
}
