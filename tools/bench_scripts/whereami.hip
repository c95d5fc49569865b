// Where does the dispatcher place the workgroups of a 256-thread kernel that fits two per CU (72 KB LDS), and when do they start?
// per block: XCC id, HW_ID (se / sh / cu / simd / wave slot), start and end shader clock.  build: hipcc -O3 --offload-arch=gfx950 whereami.hip -o whereami.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ __launch_bounds__(256, 2) void k(unsigned long long *rec, int spin)
{
    extern __shared__ float lds[];
    const long long t0 = clock64();
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        rec[4 * blockIdx.x] = hw; rec[4 * blockIdx.x + 1] = xcc; rec[4 * blockIdx.x + 2] = t0;
    }
    lds[threadIdx.x] = (float)t0;
    while (clock64() - t0 < spin) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0) rec[4 * blockIdx.x + 3] = clock64();
}
int main()
{
    const int blocks = 2048;
    unsigned long long *rec; CK(hipMalloc(&rec, blocks * 32));
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
    for (int it = 0; it < 2; it++) { hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 72 * 1024, 0, rec, 100000); CK(hipDeviceSynchronize()); }
    std::vector<unsigned long long> h(blocks * 4); CK(hipMemcpy(h.data(), rec, blocks * 32, hipMemcpyDeviceToHost));
    unsigned long long tmin = ~0ull; for (int b = 0; b < blocks; b++) tmin = h[4 * b + 2] < tmin ? h[4 * b + 2] : tmin;
    printf("block xcc se sh cu simd wave start end (shader clocks after the first start)\n");
    std::map<int, int> percu;
    for (int b = 0; b < blocks; b++) {
        const unsigned hw = (unsigned)h[4 * b], xcc = (unsigned)h[4 * b + 1] & 15;
        const int cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, simd = (hw >> 4) & 3, wave = hw & 15;
        if (b < 1040 && (b < 80 || (b % 8) == 0)) printf("%5d %2u %d %d %2d %d %2d %9llu %9llu\n", b, xcc, se, sh, cu, simd, wave, h[4 * b + 2] - tmin, h[4 * b + 3] - tmin);
        if (b < 512) percu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
    }
    int hist[8] = {0}; for (auto &p : percu) hist[p.second < 7 ? p.second : 7]++;
    printf("first 512 blocks: distinct CUs %zu; CUs with 1 / 2 / 3 / 4+ blocks: %d %d %d %d\n", percu.size(), hist[1], hist[2], hist[3], hist[4] + hist[5] + hist[6] + hist[7]);
    // which block indices share a CU with block b (first 512)?
    std::map<int, std::vector<int>> who;
    for (int b = 0; b < 512; b++) { const unsigned hw = (unsigned)h[4 * b], xcc = (unsigned)h[4 * b + 1] & 15; who[(xcc << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)].push_back(b); }
    int n = 0; for (auto &p : who) { if (n++ < 24) { printf("cu %05x:", p.first); for (int b : p.second) printf(" %d", b); printf("\n"); } }
    return 0;
}
