// gen.hip -- the bookkeeping of one round of the surface-point generator (recon/gen/generator.py:149-212, Generator.gen_pc_batch): keep the projected samples
// whose clamped distance is below the filter value and that lie in front of the camera, append them (and the predictions at them) to the per-frame output
// buffers, draw the next round's samples around the kept ones.  vistracker_amd/generator.py did this with ~40 torch launches per round (stable argsort of
// the mask, gathers, scatters, where's: 8 % of the generator's GPU time and most of its host time); here it is three launches with the same results:
//   vt_gen_round_compact   mask + STABLE compaction (kept sample indices in sample order, the first cnt[b] entries of torch.argsort(~mask, stable=True)),
//                          kept points appended at the frame's fill position, the positions of the last query at the kept points compacted for the
//                          kept-points head query, new fill levels
//   vt_gen_scatter_heads   predictions (B, C, kmax) at the kept points -> (B, cap + 1, C) buffers at the same fill positions
//   vt_gen_resample        k = floor(u * cnt) -> the k-th kept sample + (threshold / 3) * n, or a restart from the initial grid (+ 0.5 * n) for frames with
//                          fewer than two kept points -- float32 arithmetic in torch's order (u * float(cnt) truncated; scalar * normal; add)
#include "common.h"

// grid = B, block = 256.  order (B, S) int32: only the first cnt[b] entries are written; kept_pre (B, S, 3): ALL rows (kept first, then the rest in sample order).
__global__ __launch_bounds__(256) void gen_round_compact_kernel(const float *__restrict__ surface, const float *__restrict__ df_target, const float *__restrict__ pre,
                                                                const unsigned char *__restrict__ active, int S, float filter_val, float zmin,
                                                                const long long *__restrict__ fill, int cap, int write, float *__restrict__ buf_points,
                                                                int *__restrict__ order, float *__restrict__ kept_pre, long long *__restrict__ cnt,
                                                                long long *__restrict__ fill_out)
{
    __shared__ int wsum[4];
    __shared__ int base;
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const bool act = active[b] != 0;
    const long long f0 = fill[b];
    if (tid == 0) base = 0;
    __syncthreads();
    for (int s0 = 0; s0 < S; s0 += 256) {
        const int i = s0 + tid;
        bool keep = false;
        float sx = 0.f, sy = 0.f, sz = 0.f;
        if (i < S && act) {
            const float *p = surface + ((size_t)b * S + i) * 3;
            sx = p[0]; sy = p[1]; sz = p[2];
            keep = (df_target[(size_t)b * S + i] < filter_val) && (sz > zmin);
        }
        const unsigned long long bal = __ballot(keep);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; w++) woff += wsum[w];
        const int k = base + woff + before;
        if (keep) {
            order[(size_t)b * S + k] = i;
            if (kept_pre) {
                const float *q = pre + ((size_t)b * S + i) * 3; float *o = kept_pre + ((size_t)b * S + k) * 3;
                o[0] = q[0]; o[1] = q[1]; o[2] = q[2];
            }
            const long long dest = f0 + k;
            if (write && dest < cap) { float *o = buf_points + ((size_t)b * (cap + 1) + dest) * 3; o[0] = sx; o[1] = sy; o[2] = sz; }
        }
        __syncthreads();
        if (tid == 0) base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    const int total = base;                 // (every thread read `base` after the last barrier of the loop)
    if (tid == 0) {
        cnt[b] = total;
        const long long f1 = write ? f0 + total : f0;
        fill_out[b] = f1 < cap ? f1 : cap;
    }
    // The kept-points head query evaluates kmax = max over the frames of cnt[b] (rounded up to 64) rows of kept_pre for EVERY frame: the rows behind a frame's own
    // kept points must hold valid positions too -- the split-f16 decoders poison a whole 64-point tile with NaN when one of its points leaves the operand range, and
    // uninitialised memory does (round 5: the first generator call of a process returned NaN heads for a few kept points; later calls found the allocator's old
    // blocks with sane values in them).  They get what torch.argsort(~mask, stable=True) puts there: the NOT-kept samples in sample order.
    if (kept_pre) {
        __syncthreads();
        if (tid == 0) base = 0;
        __syncthreads();
        for (int s0 = 0; s0 < S; s0 += 256) {
            const int i = s0 + tid;
            bool drop = false;
            if (i < S) drop = !(act && (df_target[(size_t)b * S + i] < filter_val) && (surface[((size_t)b * S + i) * 3 + 2] > zmin));
            const unsigned long long bal = __ballot(drop);
            const int before = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) wsum[wave] = __popcll(bal);
            __syncthreads();
            int woff = 0;
            for (int w = 0; w < wave; w++) woff += wsum[w];
            if (drop) {
                const int k = total + base + woff + before;
                const float *q = pre + ((size_t)b * S + i) * 3; float *o = kept_pre + ((size_t)b * S + k) * 3;
                o[0] = q[0]; o[1] = q[1]; o[2] = q[2];
            }
            __syncthreads();
            if (tid == 0) base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256) void gen_scatter_heads_kernel(const float *__restrict__ pred, int C, int kmax, const long long *__restrict__ fill_old,
                                                                const long long *__restrict__ cnt, int cap, float *__restrict__ buf)
{
    const int b = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    if (t >= kmax * C) return;
    const int c = t / kmax, k = t - c * kmax;                      // consecutive threads read consecutive k of one channel
    if (k >= cnt[b]) return;
    const long long dest = fill_old[b] + k;
    if (dest >= cap) return;
    buf[((size_t)b * (cap + 1) + dest) * C + c] = pred[((size_t)b * C + c) * kmax + k];
}

__global__ __launch_bounds__(256) void gen_resample_kernel(const float *__restrict__ samples, const int *__restrict__ order, const long long *__restrict__ cnt,
                                                           const float *__restrict__ init, int S, int S0, const float *__restrict__ u, const float *__restrict__ pert,
                                                           int M, float near_scale, float *__restrict__ out)
{
    const int b = blockIdx.y, m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const long long c = cnt[b];
    const float uu = u[(size_t)b * M + m];
    const float *n = pert + ((size_t)b * M + m) * 3;
    float *o = out + ((size_t)b * M + m) * 3;
    if (c > 1) {
        long long k = (long long)(uu * (float)c);
        if (k > c - 1) k = c - 1;
        const float *p = samples + ((size_t)b * S + order[(size_t)b * S + k]) * 3;
        o[0] = p[0] + near_scale * n[0]; o[1] = p[1] + near_scale * n[1]; o[2] = p[2] + near_scale * n[2];
    } else {
        long long k0 = (long long)(uu * (float)S0);
        if (k0 > S0 - 1) k0 = S0 - 1;
        const float *p = init + ((size_t)b * S0 + k0) * 3;
        o[0] = p[0] + 0.5f * n[0]; o[1] = p[1] + 0.5f * n[1]; o[2] = p[2] + 0.5f * n[2];
    }
}

extern "C" int vt_gen_round_compact(const float *surface, const float *df_target, const float *pre, const unsigned char *active, int B, int S, float filter_val,
                                    float zmin, const long long *fill, int cap, int write, float *buf_points, int *order, float *kept_pre, long long *cnt,
                                    long long *fill_out, void *stream)
{
    VT_REQUIRE(surface && df_target && active && fill && order && cnt && fill_out && B > 0 && S > 0 && cap > 0 && (!write || buf_points) && (!kept_pre || pre),
               "vt_gen_round_compact: bad argument");
    hipLaunchKernelGGL(gen_round_compact_kernel, dim3(B), dim3(256), 0, vt_stream(stream), surface, df_target, pre, active, S, filter_val, zmin, fill, cap, write,
                       buf_points, order, kept_pre, cnt, fill_out);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

extern "C" int vt_gen_scatter_heads(const float *pred, int B, int C, int kmax, const long long *fill_old, const long long *cnt, int cap, float *buf, void *stream)
{
    VT_REQUIRE(pred && fill_old && cnt && buf && B > 0 && C > 0 && kmax > 0 && cap > 0, "vt_gen_scatter_heads: bad argument");
    hipLaunchKernelGGL(gen_scatter_heads_kernel, dim3((kmax * C + 255) / 256, B), dim3(256), 0, vt_stream(stream), pred, C, kmax, fill_old, cnt, cap, buf);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

extern "C" int vt_gen_resample(const float *samples, const int *order, const long long *cnt, const float *init, int B, int S, int S0, const float *u,
                               const float *pert, int M, float near_scale, float *out, void *stream)
{
    VT_REQUIRE(samples && order && cnt && init && u && pert && out && B > 0 && S > 0 && S0 > 0 && M > 0, "vt_gen_resample: bad argument");
    hipLaunchKernelGGL(gen_resample_kernel, dim3((M + 255) / 256, B), dim3(256), 0, vt_stream(stream), samples, order, cnt, init, S, S0, u, pert, M, near_scale, out);
    VT_LAUNCH_CHECK();
    return VT_OK;
}
