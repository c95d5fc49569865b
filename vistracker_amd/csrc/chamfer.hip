// chamfer.hip -- ragged contact Chamfer (pytorch3d.loss.chamfer_distance on Pointclouds, K=1 squared-L2 NN both ways,
// point_reduction='mean' over each cloud's true length, batch_reduction='mean' over pairs).  PARITY UNPINNED: pytorch3d
// is not vendored by the reference; semantics restated from its documented defaults.
// One workgroup per (frame, part) pair; the far cloud is staged through LDS in 1024-point chunks (contact sets are
// 1..~1600 points, recon_fit_trivis_full.py:393-457), brute force O(n_x n_y).
#include "common.h"

#define CH_CHUNK 1024
// threads per workgroup = points of the near cloud per sweep over the far cloud: a pair is ONE workgroup, so its sweeps are the launch's critical path
// (the largest contact sets have ~1200 points: three sweeps of 512 instead of five of 256; 1024 threads measure the same)
#ifndef CH_BLK
#define CH_BLK 512
#endif

// Gradients are accumulated WITHOUT atomics so that the 'joint' phase is run-to-run reproducible: the near side's gradient ga[i] is owned
// by the thread of point i; the far side's gb[j] collects the contributions of all points i whose nearest neighbour is j -- after each
// block of 256 points the (nn index, gradient) records go through LDS and thread j adds the records that name j in record order (fixed
// summation order, thread j is the only writer of gb[j]).  Costs one compare per (i, j) pair on top of the distance pass.
__device__ __forceinline__ void chamfer_dir(const float *__restrict__ a, int na, const float *__restrict__ bpts, int nb, float4 *sB4, int *sNN, float *sG,
                                            float gs, float *ga, float *gb, double &acc)
{
    // for every point of a: nearest point of b
    for (int i0 = 0; i0 < na; i0 += CH_BLK) {
        const int i = i0 + threadIdx.x;
        float ax = 0.f, ay = 0.f, az = 0.f;
        if (i < na) { ax = a[3 * i]; ay = a[3 * i + 1]; az = a[3 * i + 2]; }
        float best = INFINITY; int bj = -1;
        for (int c0 = 0; c0 < nb; c0 += CH_CHUNK) {
            const int cn = min(CH_CHUNK, nb - c0);
            __syncthreads();
            // one float4 per far point (one ds_read_b128 per candidate instead of three scalar reads), the chunk padded to a multiple of four with
            // points at infinity (their distance is +inf and never wins): the scan runs four candidates per iteration, in ascending order with a strict
            // compare -- the same winner and the same bits as the scalar loop
            for (int t = threadIdx.x; t < ((cn + 3) & ~3); t += CH_BLK)
                sB4[t] = t < cn ? make_float4(bpts[3 * (c0 + t)], bpts[3 * (c0 + t) + 1], bpts[3 * (c0 + t) + 2], 0.f) : make_float4(INFINITY, INFINITY, INFINITY, 0.f);
            __syncthreads();
            if (i < na) for (int j = 0; j < cn; j += 4) {
                float dd[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const float4 b = sB4[j + u];
                    const float d0 = ax - b.x, d1 = ay - b.y, d2 = az - b.z;
                    dd[u] = d0 * d0 + d1 * d1 + d2 * d2;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) if (dd[u] < best) { best = dd[u]; bj = c0 + j + u; }
            }
        }
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
        const bool live = i < na && bj >= 0;
        if (live) {
            acc += (double)best / (double)na;
            if (ga || gb) { g0 = 2.f * (ax - bpts[3 * bj]) * gs / na; g1 = 2.f * (ay - bpts[3 * bj + 1]) * gs / na; g2 = 2.f * (az - bpts[3 * bj + 2]) * gs / na; }
            if (ga) { ga[3 * i] += g0; ga[3 * i + 1] += g1; ga[3 * i + 2] += g2; }
        }
        if (gb) {
            __syncthreads();
            sNN[threadIdx.x] = live ? bj : -1; sG[3 * threadIdx.x] = g0; sG[3 * threadIdx.x + 1] = g1; sG[3 * threadIdx.x + 2] = g2;
            __syncthreads();
            const int nrec = min(CH_BLK, na - i0);
            for (int j = threadIdx.x; j < nb; j += CH_BLK) {
                float s0 = 0.f, s1 = 0.f, s2 = 0.f; bool any = false;
                for (int t = 0; t < nrec; t++)
                    if (sNN[t] == j) { s0 += sG[3 * t]; s1 += sG[3 * t + 1]; s2 += sG[3 * t + 2]; any = true; }
                if (any) { gb[3 * j] -= s0; gb[3 * j + 1] -= s1; gb[3 * j + 2] -= s2; }
            }
        }
    }
}

__global__ __launch_bounds__(CH_BLK) void chamfer_kernel(const float *__restrict__ x, const int *__restrict__ offx, const float *__restrict__ y,
                                                      const int *__restrict__ offy, int P, float gs, double *term, float *dx, float *dy, const int *skip)
{
    VT_SKIP_RETURN(skip);
    __shared__ float4 sB4[CH_CHUNK];
    __shared__ int sNN[CH_BLK];
    __shared__ float sG[CH_BLK * 3];
    __shared__ double red[CH_BLK / 64];
    const int p = blockIdx.x;
    const int ox = offx[p], nx = offx[p + 1] - ox, oy = offy[p], ny = offy[p + 1] - oy;
    double acc = 0;
    if (nx > 0 && ny > 0) {
        float *gx = dx ? dx + 3 * (size_t)ox : nullptr, *gy = dy ? dy + 3 * (size_t)oy : nullptr;
        chamfer_dir(x + 3 * (size_t)ox, nx, y + 3 * (size_t)oy, ny, sB4, sNN, sG, gs, gx, gy, acc);
        __syncthreads();        // the second direction adds to the same gradient rows from other threads
        chamfer_dir(y + 3 * (size_t)oy, ny, x + 3 * (size_t)ox, nx, sB4, sNN, sG, gs, gy, gx, acc);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && term) {
        double tot = 0;
        for (int w = 0; w < CH_BLK / 64; w++) tot += red[w];
        atomicAdd(term, tot / (double)P);
    }
}

extern "C" int vt_chamfer_ragged(const float *x, const int *offx, const float *y, const int *offy, int P, float gscale, double *term,
                                 float *dx, float *dy, void *stream)
{
    VT_REQUIRE(x && offx && y && offy && P > 0, "vt_chamfer_ragged: bad argument");
    hipLaunchKernelGGL(chamfer_kernel, dim3(P), dim3(CH_BLK), 0, vt_stream(stream), x, offx, y, offy, P, gscale / (float)P, term, dx, dy, vt_skip_flag_of(vt_stream(stream)));
    VT_LAUNCH_CHECK();
    return VT_OK;
}

// ---- the same term with a pair split over several workgroups (vt_chamfer_ragged_ws) ------------------------------------------------------------------
// One workgroup per pair makes the largest pair the launch's critical path (218 us at the bench's contact sets for 20 us of arithmetic).  Here a work
// item is (pair, direction, block of 64 NEAR points, four lanes per point): it finds the nearest far point of its 64 points, adds the near-side gradient (its thread owns the
// row), its share of the term, and leaves (nn index, gradient) records in the workspace; a second launch, item = (pair, side, block of 64 FAR points), lets
// thread j add the records that name j in ascending record order -- still no float atomics, still one writer per gradient row, run-to-run reproducible.
#define CHS_BLK 512
#define CHS_PTS 128         /* near (or far) points per work item: four lanes per point share its scan (the critical path of an item is a serial scan) */
struct ChPlan { int *pre1, *pre2; };      // exclusive prefix sums over the pairs of ceil(nx / 64), ceil(ny / 64): (P + 1) ints each
__global__ __launch_bounds__(1024) void chamfer_plan_kernel(const int *__restrict__ offx, const int *__restrict__ offy, int P, int *pre1, int *pre2)
{
    __shared__ int sc[2][1024];
    const int t = threadIdx.x, per = (P + 1023) / 1024, p0 = t * per, p1 = min(P, p0 + per);
    int a = 0, b = 0;
    for (int p = p0; p < p1; p++) {
        const int nx = offx[p + 1] - offx[p], ny = offy[p + 1] - offy[p];
        const bool live = nx > 0 && ny > 0;
        a += live ? (nx + CHS_PTS - 1) / CHS_PTS : 0; b += live ? (ny + CHS_PTS - 1) / CHS_PTS : 0;
    }
    sc[0][t] = a; sc[1][t] = b;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {            // inclusive scan (Hillis-Steele)
        const int va = t >= o ? sc[0][t - o] : 0, vb = t >= o ? sc[1][t - o] : 0;
        __syncthreads();
        sc[0][t] += va; sc[1][t] += vb;
        __syncthreads();
    }
    int ea = sc[0][t] - a, eb = sc[1][t] - b;      // exclusive prefix of this thread's first pair
    for (int p = p0; p < p1; p++) {
        const int nx = offx[p + 1] - offx[p], ny = offy[p + 1] - offy[p];
        const bool live = nx > 0 && ny > 0;
        pre1[p] = ea; pre2[p] = eb;
        ea += live ? (nx + CHS_PTS - 1) / CHS_PTS : 0; eb += live ? (ny + CHS_PTS - 1) / CHS_PTS : 0;
    }
    if (t == 1023) { pre1[P] = sc[0][1023]; pre2[P] = sc[1][1023]; }
}
// item -> (pair, block): the last pair p with pre[p] <= it (empty pairs have pre[p] == pre[p + 1] and are skipped by the search)
__device__ __forceinline__ int chs_find(const int *__restrict__ pre, int P, int it)
{
    int lo = 0, hi = P;                             // invariant: pre[lo] <= it < pre[hi]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pre[mid] <= it) lo = mid; else hi = mid; }
    return lo;
}
__global__ __launch_bounds__(CHS_BLK) void chamfer_nn_kernel(const float *__restrict__ x, const int *__restrict__ offx, const float *__restrict__ y,
                                                             const int *__restrict__ offy, int P, const int *__restrict__ pre1, const int *__restrict__ pre2, float gs,
                                                             double *term, float *dx, float *dy, int *__restrict__ nn1, float *__restrict__ g1,
                                                             int *__restrict__ nn2, float *__restrict__ g2, double *__restrict__ part, const int *skip,
                                                             const int *__restrict__ idy)
{
    // ``idy`` (vt_chamfer_ragged_idx): row r of the y list is row idy[r] of the array ``y`` points at (the transformed surface samples of the whole batch),
    // and ``dy`` is a compact scratch list this launch WRITES (row r = the near-side gradient of y row r, zero where it has none); the far launch adds
    // row r (minus the far-side records) into the big gradient array.  Without it y / dy are the compact lists themselves, dy is accumulated into.
    VT_SKIP_RETURN(skip);
    __shared__ float4 sB4[CH_CHUNK];
    __shared__ double red[CHS_BLK / 64];
    const int item = blockIdx.x;
    int it = blockIdx.x; const int n1 = pre1[P], n2 = pre2[P];
    const bool d1 = it < n1;
    if (!d1) { it -= n1; if (it >= n2) return; }
    const int *pre = d1 ? pre1 : pre2;
    const int p = chs_find(pre, P, it), blk = it - pre[p];
    const int ox = offx[p], nx = offx[p + 1] - ox, oy = offy[p], ny = offy[p + 1] - oy;
    // direction 1: near = x, far = y (records nn1 / g1 indexed by the GLOBAL x row); direction 2: near = y, far = x
    const int na = d1 ? nx : ny, nb = d1 ? ny : nx, arow = (d1 ? ox : oy);
    // rows of the near (a) and far (bp) cloud: the x list is always compact, the y list goes through idy when given
    const bool ia = idy && !d1, ib = idy && d1;
    const int *ida = idy + oy;          // (only dereferenced when ia / ib)
#define CH_AROW(i_) ((ia ? y + 3 * (size_t)ida[i_] : (d1 ? x + 3 * ((size_t)ox + (i_)) : y + 3 * ((size_t)oy + (i_)))))
#define CH_BROW(j_) ((ib ? y + 3 * (size_t)ida[j_] : (d1 ? y + 3 * ((size_t)oy + (j_)) : x + 3 * ((size_t)ox + (j_)))))
    float *ga = d1 ? dx : dy; const bool want_far = d1 ? dy != nullptr : dx != nullptr;
    int *nn = d1 ? nn1 : nn2; float *gr = d1 ? g1 : g2;
    const int sub = threadIdx.x & 3, i = blk * CHS_PTS + (threadIdx.x >> 2);       // four adjacent lanes scan a quarter of the far cloud each for point i
    float ax = 0.f, ay = 0.f, az = 0.f;
    if (i < na) { const float *a = CH_AROW(i); ax = a[0]; ay = a[1]; az = a[2]; }
    float best = INFINITY; int bj = 0x7fffffff;
    for (int c0 = 0; c0 < nb; c0 += CH_CHUNK) {
        const int cn = min(CH_CHUNK, nb - c0);
        __syncthreads();
        for (int t = threadIdx.x; t < ((cn + 15) & ~15); t += CHS_BLK) {
            float4 v = make_float4(INFINITY, INFINITY, INFINITY, 0.f);
            if (t < cn) { const float *bq = CH_BROW(c0 + t); v = make_float4(bq[0], bq[1], bq[2], 0.f); }
            sB4[t] = v;
        }
        __syncthreads();
        if (i < na) for (int j = 4 * sub; j < cn; j += 16) {
            float dd[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const float4 b = sB4[j + u];
                const float d0 = ax - b.x, d1_ = ay - b.y, d2 = az - b.z;
                dd[u] = d0 * d0 + d1_ * d1_ + d2 * d2;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) if (dd[u] < best) { best = dd[u]; bj = c0 + j + u; }
        }
    }
    // the four partial winners of a point: smallest distance, smallest index among equals = the winner of one ascending scan with a strict compare
#pragma unroll
    for (int o = 1; o < 4; o <<= 1) {
        const float ob = __shfl_xor(best, o, 64); const int oj = __shfl_xor(bj, o, 64);
        if (ob < best || (ob == best && oj < bj)) { best = ob; bj = oj; }
    }
    if (bj == 0x7fffffff) bj = -1;
    double acc = 0;
    const bool own = sub == 0, live = own && i < na && bj >= 0;      // lane 0 of the four writes the point's share, gradient and record
    float g0 = 0.f, g1v = 0.f, g2v = 0.f;
    if (live) {
        acc = (double)best / (double)na;
        if (ga || want_far) { const float *bq = CH_BROW(bj); g0 = 2.f * (ax - bq[0]) * gs / na; g1v = 2.f * (ay - bq[1]) * gs / na; g2v = 2.f * (az - bq[2]) * gs / na; }
        if (ga && !ia) { float *o = ga + 3 * ((size_t)arow + i); o[0] += g0; o[1] += g1v; o[2] += g2v; }
    }
    if (ga && ia && own && i < na) { float *o = ga + 3 * ((size_t)arow + i); o[0] = g0; o[1] = g1v; o[2] = g2v; }      // compact scratch row: written, 0 + g == g
#undef CH_AROW
#undef CH_BROW
    if (want_far && own && i < na) { nn[arow + i] = live ? bj : -1; float *o = gr + 3 * ((size_t)arow + i); o[0] = g0; o[1] = g1v; o[2] = g2v; }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    // the item's share of the term goes to the workspace: the last workgroup of the second launch adds the shares in item order (thousands of fp64
    // atomics on one address were the longest part of this launch, and their order made the last bits of the term vary from run to run)
    if (threadIdx.x == 0) { double tot = 0; for (int w = 0; w < CHS_BLK / 64; w++) tot += red[w]; part[item] = tot; }
}
// far side: item = (pair, side, block of 256 far points); side 1 = the y rows (records of direction 1), side 2 = the x rows (records of direction 2)
__global__ __launch_bounds__(CHS_BLK) void chamfer_far_kernel(const int *__restrict__ offx, const int *__restrict__ offy, int P, const int *__restrict__ pre1,
                                                              const int *__restrict__ pre2, float *dx, float *dy, const int *__restrict__ nn1,
                                                              const float *__restrict__ g1, const int *__restrict__ nn2, const float *__restrict__ g2,
                                                              const double *__restrict__ part, double *term, const int *skip,
                                                              const int *__restrict__ idy, float *__restrict__ dy_base)
{
    VT_SKIP_RETURN(skip);
    __shared__ __attribute__((aligned(16))) int sNN[CH_CHUNK];
    __shared__ float sG[CH_CHUNK * 3];
    if (blockIdx.x == gridDim.x - 1) {              // the reducer: shares of the items of the first launch, fixed order
        __shared__ double rs[CHS_BLK];
        const int nit = pre1[P] + pre2[P];
        double a = 0;
        for (int k = threadIdx.x; k < nit; k += CHS_BLK) a += part[k];
        rs[threadIdx.x] = a;
        __syncthreads();
        for (int o = CHS_BLK / 2; o > 0; o >>= 1) { if (threadIdx.x < o) rs[threadIdx.x] += rs[threadIdx.x + o]; __syncthreads(); }
        if (threadIdx.x == 0 && term) atomicAdd(term, rs[0] / (double)P);
        return;
    }
    // far rows of direction 1 are the y rows: blocks counted by pre2; of direction 2 the x rows: pre1
    int it = blockIdx.x; const int n2 = dy ? pre2[P] : 0, n1 = dx ? pre1[P] : 0;
    const bool s1 = it < n2;                        // side 1: y rows
    if (!s1) { it -= n2; if (it >= n1) return; }
    const int *pre = s1 ? pre2 : pre1;
    const int p = chs_find(pre, P, it), blk = it - pre[p];
    const int ox = offx[p], nx = offx[p + 1] - ox, oy = offy[p], ny = offy[p + 1] - oy;
    const int nfar = s1 ? ny : nx, nrec = s1 ? nx : ny, frow = s1 ? oy : ox, rrow = s1 ? ox : oy;
    const int *nn = s1 ? nn1 : nn2; const float *gr = s1 ? g1 : g2; float *gb = s1 ? dy : dx;
    const int sub = threadIdx.x & 3, j = blk * CHS_PTS + (threadIdx.x >> 2);       // four lanes per far point, a quarter of the records each
    float s0 = 0.f, s1v = 0.f, s2 = 0.f; bool any = false;
    for (int c0 = 0; c0 < nrec; c0 += CH_CHUNK) {
        const int cn = min(CH_CHUNK, nrec - c0);
        __syncthreads();
        for (int t = threadIdx.x; t < ((cn + 15) & ~15); t += CHS_BLK) {
            sNN[t] = t < cn ? nn[rrow + c0 + t] : -1;
            if (t < cn) { sG[3 * t] = gr[3 * ((size_t)rrow + c0 + t)]; sG[3 * t + 1] = gr[3 * ((size_t)rrow + c0 + t) + 1]; sG[3 * t + 2] = gr[3 * ((size_t)rrow + c0 + t) + 2]; }
        }
        __syncthreads();
        // four records per LDS read; a far point is the nearest neighbour of a few near points only, so the adds are rare; every lane adds its records in
        // ascending order, the four partial sums are combined in a fixed order below: one writer per row, no float atomics, reproducible
        if (j < nfar)
            for (int t = 4 * sub; t < cn; t += 16) {
                const int4 n4 = *reinterpret_cast<const int4 *>(sNN + t);
                if ((n4.x == j) | (n4.y == j) | (n4.z == j) | (n4.w == j)) {
                    if (n4.x == j) { s0 += sG[3 * t]; s1v += sG[3 * t + 1]; s2 += sG[3 * t + 2]; }
                    if (n4.y == j) { s0 += sG[3 * t + 3]; s1v += sG[3 * t + 4]; s2 += sG[3 * t + 5]; }
                    if (n4.z == j) { s0 += sG[3 * t + 6]; s1v += sG[3 * t + 7]; s2 += sG[3 * t + 8]; }
                    if (n4.w == j) { s0 += sG[3 * t + 9]; s1v += sG[3 * t + 10]; s2 += sG[3 * t + 11]; }
                    any = true;
                }
            }
    }
    // (lane 0 + lane 1) + (lane 2 + lane 3)
#pragma unroll
    for (int o = 1; o < 4; o <<= 1) {
        s0 += __shfl_xor(s0, o, 64); s1v += __shfl_xor(s1v, o, 64); s2 += __shfl_xor(s2, o, 64);
        any = any | (__shfl_xor((int)any, o, 64) != 0);
    }
    if (s1 && idy) {
        // y rows through the index list: the row's near-side gradient g (compact scratch, written by the first launch) minus its far-side records is ADDED
        // to row idy[r] of the big gradient array: dX + ((0 + g) - s), the additions of "gather, Chamfer on the compact list, index_add_" in their order.
        // A surface point belongs to one (frame, part) pair: one writer per row.
        if (sub == 0 && j < nfar) {
            const float *gq = gb + 3 * ((size_t)frow + j); float *o = dy_base + 3 * (size_t)idy[frow + j];
            float v0 = gq[0], v1 = gq[1], v2 = gq[2];
            if (any) { v0 -= s0; v1 -= s1v; v2 -= s2; }
            o[0] += v0; o[1] += v1; o[2] += v2;
        }
        return;
    }
    if (any && sub == 0 && j < nfar) { float *o = gb + 3 * ((size_t)frow + j); o[0] -= s0; o[1] -= s1v; o[2] -= s2; }
}
extern "C" long vt_chamfer_ws_bytes(long total_x, long total_y, int P)
{
    if (total_x < 0 || total_y < 0 || P <= 0) return 0;
    const long items = total_x / CHS_PTS + total_y / CHS_PTS + 2 * (long)P;        // upper bound of the work items of the first launch
    return (long)sizeof(int) * 2 * ((long)P + 1) + (long)(sizeof(int) + 3 * sizeof(float)) * (total_x + total_y) + 8 + (long)sizeof(double) * items + 64
           + (long)(3 * sizeof(float)) * total_y + 16;        // + the compact near-side gradient list of vt_chamfer_ragged_idx
}
static int chamfer_ws_launch(const float *x, const int *offx, long total_x, const float *y, const int *idy, const int *offy, long total_y, int P, float gscale,
                             double *term, float *dx, float *dy, void *ws, int run_plan, hipStream_t st)
{
    const int *skip = vt_skip_flag_of(st);
    int *pre1 = reinterpret_cast<int *>(ws), *pre2 = pre1 + (P + 1);
    int *nn1 = pre2 + (P + 1), *nn2 = nn1 + total_x;
    float *g1 = reinterpret_cast<float *>(nn2 + total_y), *g2 = g1 + 3 * total_x;
    double *part = reinterpret_cast<double *>((reinterpret_cast<uintptr_t>(g2 + 3 * total_y) + 7) & ~(uintptr_t)7);
    const long m1 = total_x / CHS_PTS + P, m2 = total_y / CHS_PTS + P;      // upper bounds of the item counts (the exact ones are pre1[P], pre2[P] on the device)
    float *dyc = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(part + (m1 + m2)) + 15) & ~(uintptr_t)15);       // compact dy list of the indexed form
    if (run_plan) {
        hipLaunchKernelGGL(chamfer_plan_kernel, dim3(1), dim3(1024), 0, st, offx, offy, P, pre1, pre2);
        VT_LAUNCH_CHECK();
    }
    float *dy_list = idy ? (dy ? dyc : nullptr) : dy;
    hipLaunchKernelGGL(chamfer_nn_kernel, dim3((unsigned)(m1 + m2)), dim3(CHS_BLK), 0, st, x, offx, y, offy, P, pre1, pre2, gscale / (float)P, term, dx, dy_list,
                       nn1, g1, nn2, g2, part, skip, idy);
    VT_LAUNCH_CHECK();
    // far-side gradients (if any are wanted) + one more workgroup that reduces the items' shares of the term
    hipLaunchKernelGGL(chamfer_far_kernel, dim3((unsigned)((dy ? m2 : 0) + (dx ? m1 : 0) + 1)), dim3(CHS_BLK), 0, st, offx, offy, P, pre1, pre2, dx, dy_list, nn1, g1, nn2, g2,
                       part, term, skip, idy, dy);
    VT_LAUNCH_CHECK();
    return VT_OK;
}
extern "C" int vt_chamfer_ragged_ws(const float *x, const int *offx, long total_x, const float *y, const int *offy, long total_y, int P, float gscale,
                                    double *term, float *dx, float *dy, void *ws, void *stream)
{
    VT_REQUIRE(x && offx && y && offy && ws && P > 0 && P <= 1024 * 1024 && total_x >= 0 && total_y >= 0, "vt_chamfer_ragged_ws: bad argument");
    return chamfer_ws_launch(x, offx, total_x, y, nullptr, offy, total_y, P, gscale, term, dx, dy, ws, 1, vt_stream(stream));
}
extern "C" int vt_chamfer_ragged_idx(const float *x, const int *offx, long total_x, const float *y_base, const int *idx_y, const int *offy, long total_y, int P,
                                     float gscale, double *term, float *dy_base, void *ws, int run_plan, void *stream)
{
    VT_REQUIRE(x && offx && y_base && idx_y && offy && ws && P > 0 && P <= 1024 * 1024 && total_x >= 0 && total_y >= 0, "vt_chamfer_ragged_idx: bad argument");
    return chamfer_ws_launch(x, offx, total_x, y_base, idx_y, offy, total_y, P, gscale, term, nullptr, dy_base, ws, run_plan, vt_stream(stream));
}

// ---- evaluation Chamfer (recon/eval/chamfer_distance.py:10-52): per point the Euclidean (NOT squared) distance to the nearest
// point of the other cloud; P equal-sized cloud pairs at once.  grid = (tiles of 256 query points, P); the searched cloud streams
// through LDS as float4 (x, y, z, |p|^2 unused) in 2048-point chunks, every thread keeps its query point in registers.
#define NN_CHUNK 2048
__global__ __launch_bounds__(256) void nn_distance_kernel(const float *__restrict__ q, int nq, const float *__restrict__ s, int ns, float *__restrict__ dist)
{
    __shared__ float sx[NN_CHUNK], sy[NN_CHUNK], sz[NN_CHUNK];
    const int p = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const float *qp = q + (size_t)p * nq * 3, *sp = s + (size_t)p * ns * 3;
    float ax = 0.f, ay = 0.f, az = 0.f;
    if (i < nq) { ax = qp[3 * i]; ay = qp[3 * i + 1]; az = qp[3 * i + 2]; }
    float best = INFINITY;
    for (int c0 = 0; c0 < ns; c0 += NN_CHUNK) {
        const int cn = min(NN_CHUNK, ns - c0);
        __syncthreads();
        for (int t = threadIdx.x; t < cn; t += 256) { sx[t] = sp[3 * (c0 + t)]; sy[t] = sp[3 * (c0 + t) + 1]; sz[t] = sp[3 * (c0 + t) + 2]; }
        __syncthreads();
#pragma unroll 8
        for (int j = 0; j < cn; j++) {
            const float d0 = ax - sx[j], d1 = ay - sy[j], d2 = az - sz[j];
            best = fminf(best, d0 * d0 + d1 * d1 + d2 * d2);
        }
    }
    if (i < nq) dist[(size_t)p * nq + i] = sqrtf(best);
}

extern "C" int vt_nn_distance(const float *query, int nq, const float *search, int ns, int P, float *dist, void *stream)
{
    VT_REQUIRE(query && search && dist && nq > 0 && ns > 0 && P > 0 && P < 65536, "vt_nn_distance: bad argument");
    hipLaunchKernelGGL(nn_distance_kernel, dim3((nq + 255) / 256, P), dim3(256), 0, vt_stream(stream), query, nq, search, ns, dist);
    VT_LAUNCH_CHECK();
    return VT_OK;
}
