// Micro-benchmark: what a 16-byte-per-lane gather costs the L1 (TA/TCP) as a function of how the lanes of one instruction are laid over
// the 128-byte lines.  Every "point" owns one random 128-byte row of a table (L2-resident size by default); a workgroup of 256 threads reads the
// rows of 64 points per iteration (8 x 16 B per row), with three lane mappings:
//   0: lane = (point, 32-byte piece): two instructions, each touching HALF of 16 rows per wave          (the query kernel's tap gathers today)
//   1: lane = (point-of-8, 16-byte piece): each instruction reads 8 FULL rows per wave, two point groups per thread
//   4: lane-linear: a wave reads 1 KB contiguous (the weight-fragment loads)
//   5: lane-linear 1 KB per wave like 4, but delivered to LDS by the asynchronous DMA (global_load_lds, 16 B per lane) instead of to registers
//   2: lane = (point, 64-byte half): four instructions each touching 16 B of ... (the P-row pattern: 4 lanes x 16 B = 64 contiguous bytes per point)
// build: hipcc -O3 --offload-arch=gfx950 gather_patterns.hip -o gather_patterns ; run: ./gather_patterns [table MB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int PAT> __global__ __launch_bounds__(256) void gather(const float4 *__restrict__ tab, const int *__restrict__ rows, int iters, float4 *__restrict__ out)
{
    const int tid = threadIdx.x;
    float4 acc = make_float4(0, 0, 0, 0);
    __shared__ float4 sbuf[8][512];
    for (int it = 0; it < iters; it++) {
        const int *r = rows + ((size_t)blockIdx.x * iters + it) * 64;
        if (PAT == 5) {
            const int lane = tid & 63, w = tid >> 6;
            const float4 *p = tab + ((size_t)r[w * 16] * 8 & ~(size_t)127) + lane;
            float4 *dst = &sbuf[it & 7][w * 128];
            __builtin_amdgcn_global_load_lds(p, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds(p + 64, (__attribute__((address_space(3))) void *)(dst + 64), 16, 0, 0);
            if ((it & 7) == 7) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        } else
        if (PAT == 0) {
            const int pt = tid >> 2, piece = tid & 3;                       // 4 lanes per point, 32 B each = 2 loads
            const float4 *p = tab + (size_t)r[pt] * 8 + piece * 2;
            const float4 a = p[0], b = p[1];
            acc.x += a.x + b.x; acc.y += a.y + b.y; acc.z += a.z + b.z; acc.w += a.w + b.w;
        } else if (PAT == 1) {
            const int pg = tid >> 3, piece = tid & 7;                       // 8 lanes per point, 16 B each; points pg and pg + 32
            const float4 a = tab[(size_t)r[pg] * 8 + piece], b = tab[(size_t)r[pg + 32] * 8 + piece];
            acc.x += a.x + b.x; acc.y += a.y + b.y; acc.z += a.z + b.z; acc.w += a.w + b.w;
        } else if (PAT == 2) {
            // MFMA D-fragment pattern: lane = pt16 + 16 g within a wave; wave w handles ... each lane reads 16 B at g * 16 of a 64-byte half; two halves = 2 loads,
            // waves 0..3 = the 4 point groups of 16
            const int lane = tid & 63, w = tid >> 6, pt = w * 16 + (lane & 15), g = lane >> 4;
            const float4 *p = tab + (size_t)r[pt] * 8 + g;
            const float4 a = p[0], b = p[4];
            acc.x += a.x + b.x; acc.y += a.y + b.y; acc.z += a.z + b.z; acc.w += a.w + b.w;
        } else if (PAT == 4) {
            // weight-fragment pattern: a wave reads 1 KB contiguous (lane-linear 16 B) at the offset of row r[wave * 16] (128-B aligned), twice
            const int lane = tid & 63, w = tid >> 6;
            const float4 *p = tab + ((size_t)r[w * 16] * 8 & ~(size_t)127) + lane;
            const float4 a = p[0], b = p[64];
            acc.x += a.x + b.x; acc.y += a.y + b.y; acc.z += a.z + b.z; acc.w += a.w + b.w;
        } else {
            // like 2, but the four 16-byte groups of a point sit in ADJACENT lanes (lane = 4 pt + g): 4 lanes = 64 contiguous bytes
            const int lane = tid & 63, w = tid >> 6, pt = w * 16 + (lane >> 2), g = lane & 3;
            const float4 *p = tab + (size_t)r[pt] * 8 + g;
            const float4 a = p[0], b = p[4];
            acc.x += a.x + b.x; acc.y += a.y + b.y; acc.z += a.z + b.z; acc.w += a.w + b.w;
        }
    }
    if (PAT == 5) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); acc = sbuf[tid & 7][tid]; }
    out[(size_t)blockIdx.x * 256 + tid] = acc;
}
template <int PAT> float run(const float4 *tab, const int *rows, int blocks, int iters, float4 *out)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(gather<PAT>, dim3(blocks), dim3(256), 0, 0, tab, rows, iters, out);
    CK(hipEventRecord(e0));
    for (int k = 0; k < 5; k++) hipLaunchKernelGGL(gather<PAT>, dim3(blocks), dim3(256), 0, 0, tab, rows, iters, out);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 5;
}
int main(int argc, char **argv)
{
    const size_t mb = argc > 1 ? atoi(argv[1]) : 64; const int local = argc > 2 ? atoi(argv[2]) : 0;
    const size_t nrows = mb * (1 << 20) / 128;
    const int blocks = 256 * 8, iters = 512;
    float4 *tab, *out; int *rows;
    CK(hipMalloc(&tab, nrows * 128)); CK(hipMemset(tab, 0, nrows * 128)); CK(hipMalloc(&out, (size_t)blocks * 256 * 16));
    std::vector<int> h((size_t)blocks * iters * 64);
    srand(1);
    // local = 0: uniformly random rows; local = n: each group of 64 points walks a window of n rows (neighbouring points share texels, like a mesh in an image)
    for (size_t i = 0; i < h.size(); i += 64) {
        const size_t base = ((size_t)rand() * 7919 + rand()) % nrows;
        for (int k = 0; k < 64; k++) h[i + k] = local ? (int)((base + rand() % local) % nrows) : (int)(((size_t)rand() * 7919 + rand()) % nrows);
    }
    CK(hipMalloc(&rows, h.size() * 4)); CK(hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const double bytes = (double)blocks * iters * 64 * 128;
    float t0 = run<0>(tab, rows, blocks, iters, out), t1 = run<1>(tab, rows, blocks, iters, out), t2 = run<2>(tab, rows, blocks, iters, out), t3 = run<3>(tab, rows, blocks, iters, out), t4 = run<4>(tab, rows, blocks, iters, out), t5 = run<5>(tab, rows, blocks, iters, out);
    printf("table %zu MB, window %d rows: half-row x2 (today) %.3f ms = %.2f TB/s | full rows %.3f ms = %.2f TB/s | D-fragment 16 B strided lanes %.3f ms = %.2f TB/s | 64 B adjacent lanes %.3f ms = %.2f TB/s | lane-linear 1 KB per wave %.3f ms = %.2f TB/s | the same through the LDS DMA %.3f ms = %.2f TB/s\n",
           mb, local, t0, bytes / t0 * 1e-9, t1, bytes / t1 * 1e-9, t2, bytes / t2 * 1e-9, t3, bytes / t3 * 1e-9, t4, bytes / t4 * 1e-9, t5, bytes / t5 * 1e-9);
    return 0;
}
