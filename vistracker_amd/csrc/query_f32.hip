// query_f32.hip -- STRICT-fp32 route of the SIF-Net point query: the same fused kernel structure as query.hip (projection + 8 bilinear gathers
// + MLP decoders, forward and backward-to-coordinates, fused objectives, surface-projection step) with every GEMM on the f32-INPUT MFMA
// (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate -- the arithmetic of the reference's nn.Conv1d decoders, model/chore.py:113-126).
// 1/5 of the speed of the split-f16 kernels; it exists (a) as the fallback when a decoder activation leaves the range of the split-f16
// operands (|x| >= 1023, DESIGN.md 4.1: the fit loops switch to it automatically instead of failing), (b) to attribute trajectory drift to the
// split arithmetic or to Adam's amplification of round-off (tests/test_gpu_fullsched.py).  Selected per handle: Net_set_precision.
#include "common.h"
#include "query_f32.h"

namespace f32q {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

#define KTOT 612            /* 608 map channels + x,y,z-2.2 + 1 zero pad (internal channel order) */
#define NCHUNK 19
#define FS 36               /* LDS stride of a 32-channel chunk row   [pt][32 + 4]:  36 j mod 64 distinct -> conflict-free ds_read_b64 */
#define HS 132              /* LDS stride of a hidden activation row  [pt][128 + 4]: 132 j mod 64 = 4 j                                  */
#define GS 20               /* LDS stride of an output-gradient row   [pt][16 + 4]                                                       */
#define OUT_DIST 5.0f       /* chore.py:93 */

enum { MODE_FWD = 0, MODE_BWD = 1, MODE_HUMAN = 2, MODE_OBJECT = 3, MODE_PROJECT = 4 };

// Weight fragments.  All GEMMs walk K in "pair steps": lane (q = lane>>4, j = lane&15) owns k = 8 s + 2 q + e, e in {0,1}, so
// its two A values are one ds_read_b64 and its B values for both k's and both N-tiles of the wave are one 16-B global load.
//   wNp  : [K/2][4 waves][16 j][2 e][2 nt]   element = W[k = 2 kp + e][n = (2 wave + nt) 16 + j]        (layers 1-3 fwd and bwd, layer-4 bwd)
//   w1c  : [64 kp][20 chunks][16 j][2 e][2 nt] element = W1(out,in)[u = 2 kp + e][c = chunk 32 + nt 16 + j]   (layer-1 backward)
//   w4p  : [64 kp][16 j][2 e]                element = W4(in,out)[k = 2 kp + e][o = j]
struct HeadW {
    const float *w1p, *w1c, *w1xio, *w1xoi, *b1, *w2p, *w2tp, *b2, *w3p, *w3tp, *b3, *w4p, *w4tp, *b4;
    int kout, id;
};

struct Net {
    float *blob;            // all weights of the 5 heads
    HeadW head[5];
    float cam[5];
};

struct QArgs {
    const float *maps[8];
    int res[8];
    const float *pts, *crop_center, *body_center;
    int B, N;
    float fx, fy, cx, cy, crop;
    HeadW hw[2];
    float *out[2];          // MODE_FWD
    const float *gout[2];   // MODE_BWD
    float *dpts;
    // fused objectives
    const int *labels; const float *occ; float w0, w1; double *terms;
    const int *order;       // optional processing order of the points (slot n handles point order[n])
    int df_idx; float *pts_out, *dft_out;      // MODE_PROJECT (w0 = clamp threshold)
    const int *skip;        // device-side early stop (vt_stream_set_skip_flag) or NULL
};

__device__ __forceinline__ int map_channels(int mi) { return mi == 0 ? 256 : (mi == 1 ? 64 : (mi < 5 ? 32 : 64)); }
__device__ __forceinline__ int map_proj(int mi) { return mi < 2 ? 0 : (mi < 5 ? mi - 1 : mi - 4); }
// chunk i (32 channels) -> map index and channel offset inside the map
__device__ __forceinline__ void chunk_info(int i, int &mi, int &co)
{
    if (i < 8) { mi = 0; co = 32 * i; }
    else if (i < 10) { mi = 1; co = 32 * (i - 8); }
    else if (i < 13) { mi = i - 8; co = 0; }
    else { mi = 5 + (i - 13) / 2; co = 32 * ((i - 13) & 1); }
}

// The four bilinear taps of one 32-channel chunk for two points per thread, held in registers while the loads are in flight.
struct Taps { float4 t[2][4]; float wx1[2], wy1[2]; int inb[2]; float sc; };

// issue the tap loads of chunk (mi, co): thread = (point pp / pp+32, 16-B piece `sub` of the 128-B tap row)
__device__ __forceinline__ void taps_issue(const QArgs &a, int b, int mi, int co, const float *sUV, int tid, Taps &r)
{
    const float *__restrict__ map = a.maps[mi];
    const int R = a.res[mi], C = map_channels(mi), pr = map_proj(mi);
    const int sub = tid & 7, pp = tid >> 3;
    r.sc = 0.5f * (float)(R - 1);
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const int pt = pp + 32 * pass;
        const float u = sUV[(pr * 64 + pt) * 2], v = sUV[(pr * 64 + pt) * 2 + 1];
        // grid_sample, bilinear, align_corners=True, zeros padding (geometry.py:12)
        float ix = (u + 1.0f) * 0.5f * (float)(R - 1), iy = (v + 1.0f) * 0.5f * (float)(R - 1);
        ix = fminf(fmaxf(ix, -2.0f), (float)(R + 1)); iy = fminf(fmaxf(iy, -2.0f), (float)(R + 1));
        const float fxl = floorf(ix), fyl = floorf(iy);
        const int x0 = (int)fxl, y0 = (int)fyl, x1 = x0 + 1, y1 = y0 + 1;
        r.wx1[pass] = ix - fxl; r.wy1[pass] = iy - fyl;
        const bool bx0 = x0 >= 0 && x0 < R, bx1 = x1 >= 0 && x1 < R, by0 = y0 >= 0 && y0 < R, by1 = y1 >= 0 && y1 < R;
        // zeros padding: always load from a clamped (valid) texel -- plain global_load, no divergent branch, no select between a
        // global and a private address -- and fold the in-bounds flag into the interpolation weights (see taps_store_*)
        const int xc0 = min(max(x0, 0), R - 1), xc1 = min(max(x1, 0), R - 1), yc0 = min(max(y0, 0), R - 1), yc1 = min(max(y1, 0), R - 1);
        const size_t rowb = (size_t)b * R;
#ifdef ABL_NOGATHER
        r.t[pass][0] = r.t[pass][1] = r.t[pass][2] = r.t[pass][3] = make_float4(u, v, u, v);
#else
        r.t[pass][0] = *reinterpret_cast<const float4 *>(map + ((rowb + yc0) * R + xc0) * C + co + sub * 4);
        r.t[pass][1] = *reinterpret_cast<const float4 *>(map + ((rowb + yc0) * R + xc1) * C + co + sub * 4);
        r.t[pass][2] = *reinterpret_cast<const float4 *>(map + ((rowb + yc1) * R + xc0) * C + co + sub * 4);
        r.t[pass][3] = *reinterpret_cast<const float4 *>(map + ((rowb + yc1) * R + xc1) * C + co + sub * 4);
#endif
        r.inb[pass] = (bx0 && by0 ? 1 : 0) | (bx1 && by0 ? 2 : 0) | (bx0 && by1 ? 4 : 0) | (bx1 && by1 ? 8 : 0);
    }
}
// blend the taps to features and store [pt][FS] (forward) ...
__device__ __forceinline__ void taps_store_feat(const Taps &r, float *buf, int tid)
{
    const int sub = tid & 7, pp = tid >> 3;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const float wx1 = r.wx1[pass], wy1 = r.wy1[pass], wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
        const int ib = r.inb[pass];
        const float w00 = (ib & 1) ? wx0 * wy0 : 0.f, w10 = (ib & 2) ? wx1 * wy0 : 0.f, w01 = (ib & 4) ? wx0 * wy1 : 0.f, w11 = (ib & 8) ? wx1 * wy1 : 0.f;
        const float4 nw = r.t[pass][0], ne = r.t[pass][1], sw = r.t[pass][2], se = r.t[pass][3];
        *reinterpret_cast<float4 *>(buf + (pp + 32 * pass) * FS + sub * 4) =
            make_float4(nw.x * w00 + ne.x * w10 + sw.x * w01 + se.x * w11, nw.y * w00 + ne.y * w10 + sw.y * w01 + se.y * w11,
                        nw.z * w00 + ne.z * w10 + sw.z * w01 + se.z * w11, nw.w * w00 + ne.w * w10 + sw.w * w01 + se.w * w11);
    }
}
// ... or the tap differences d feat / d u, d feat / d v scaled by (res-1)/2 (backward)
__device__ __forceinline__ void taps_store_grad(const Taps &r, float *bufU, float *bufV, int tid)
{
    const int sub = tid & 7, pp = tid >> 3;
    const float sc = r.sc;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const float wx1 = r.wx1[pass], wy1 = r.wy1[pass], wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
        const int ib = r.inb[pass];
        const float m0 = (ib & 1) ? 1.f : 0.f, m1 = (ib & 2) ? 1.f : 0.f, m2 = (ib & 4) ? 1.f : 0.f, m3 = (ib & 8) ? 1.f : 0.f;
        float4 nw = r.t[pass][0], ne = r.t[pass][1], sw = r.t[pass][2], se = r.t[pass][3];
        nw.x *= m0; nw.y *= m0; nw.z *= m0; nw.w *= m0; ne.x *= m1; ne.y *= m1; ne.z *= m1; ne.w *= m1;
        sw.x *= m2; sw.y *= m2; sw.z *= m2; sw.w *= m2; se.x *= m3; se.y *= m3; se.z *= m3; se.w *= m3;
        *reinterpret_cast<float4 *>(bufU + (pp + 32 * pass) * FS + sub * 4) =
            make_float4(((ne.x - nw.x) * wy0 + (se.x - sw.x) * wy1) * sc, ((ne.y - nw.y) * wy0 + (se.y - sw.y) * wy1) * sc,
                        ((ne.z - nw.z) * wy0 + (se.z - sw.z) * wy1) * sc, ((ne.w - nw.w) * wy0 + (se.w - sw.w) * wy1) * sc);
        *reinterpret_cast<float4 *>(bufV + (pp + 32 * pass) * FS + sub * 4) =
            make_float4(((sw.x - nw.x) * wx0 + (se.x - ne.x) * wx1) * sc, ((sw.y - nw.y) * wx0 + (se.y - ne.y) * wx1) * sc,
                        ((sw.z - nw.z) * wx0 + (se.z - ne.z) * wx1) * sc, ((sw.w - nw.w) * wx0 + (se.w - ne.w) * wx1) * sc);
    }
}

// D-layout accumulators of one wave: acc[mt][nt] covers points mt*16 + (lane>>4)*4 + r, hidden unit (2*wave+nt)*16 + (lane&15)
struct Acc8 { f32x4 v[4][2]; };

__device__ __forceinline__ void acc_zero(Acc8 &c)
{
#pragma unroll
    for (int mt = 0; mt < 4; mt++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++) c.v[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

// bias + ReLU on the D fragments; returns the mask (bit = mt*8 + nt*4 + r) of positive pre-activations
__device__ __forceinline__ unsigned bias_relu(Acc8 &c, const float *__restrict__ bias, int wave, int lane)
{
    unsigned m = 0;
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
        const float bb = bias[(2 * wave + nt) * 16 + (lane & 15)];
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float x = c.v[mt][nt][r] + bb;
                if (x > 0.f) { m |= 1u << (mt * 8 + nt * 4 + r); c.v[mt][nt][r] = x; } else c.v[mt][nt][r] = 0.f;
            }
    }
    return m;
}
__device__ __forceinline__ void apply_mask(Acc8 &c, unsigned m)
{
#pragma unroll
    for (int mt = 0; mt < 4; mt++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int r = 0; r < 4; r++) if (!((m >> (mt * 8 + nt * 4 + r)) & 1u)) c.v[mt][nt][r] = 0.f;
}
// D fragments -> LDS activation buffer [pt][HS]
__device__ __forceinline__ void store_hbuf(const Acc8 &c, float *H, int wave, int lane)
{
#pragma unroll
    for (int mt = 0; mt < 4; mt++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int r = 0; r < 4; r++) H[(mt * 16 + (lane >> 4) * 4 + r) * HS + (2 * wave + nt) * 16 + (lane & 15)] = c.v[mt][nt][r];
}
// one pair step of the wave tile: A pairs of the 4 M-tiles (LDS, row stride `stride`), B float4 = {e0nt0, e0nt1, e1nt0, e1nt1}
__device__ __forceinline__ void pair_step(Acc8 &c, const float *Abase, int stride, int s, int q, int j, const float4 bb)
{
    float2 av[4];
#pragma unroll
    for (int mt = 0; mt < 4; mt++) av[mt] = *reinterpret_cast<const float2 *>(Abase + (mt * 16 + j) * stride + 8 * s + 2 * q);
#pragma unroll
    for (int mt = 0; mt < 4; mt++) { c.v[mt][0] = MFMA16(av[mt].x, bb.x, c.v[mt][0]); c.v[mt][1] = MFMA16(av[mt].x, bb.y, c.v[mt][1]); }
#pragma unroll
    for (int mt = 0; mt < 4; mt++) { c.v[mt][0] = MFMA16(av[mt].y, bb.z, c.v[mt][0]); c.v[mt][1] = MFMA16(av[mt].y, bb.w, c.v[mt][1]); }
}
// out[64 x (32 cols of this wave)] = H[64 x 128] (A, LDS) x W[128 x 128] (B, pair-step fragments from L2), K = 128.
// The weights do not depend on the LDS contents: the first 8 fragments are requested BEFORE the barrier that publishes H
// (wpre), the other 8 while the first MFMAs run.
struct WPre { float4 v[8]; };
__device__ __forceinline__ void wprefetch(WPre &p, const float *__restrict__ Wp, int wave, int lane)
{
    const float4 *__restrict__ w = reinterpret_cast<const float4 *>(Wp) + ((lane >> 4) * 4 + wave) * 16 + (lane & 15);
#pragma unroll
    for (int s = 0; s < 8; s++) p.v[s] = w[(size_t)s * 4 * 64];
}
__device__ __forceinline__ void gemm128(Acc8 &c, const float *H, const float *__restrict__ Wp, const WPre &p, int wave, int lane)
{
    acc_zero(c);
    const int q = lane >> 4, j = lane & 15;
    const float4 *__restrict__ w = reinterpret_cast<const float4 *>(Wp) + (q * 4 + wave) * 16 + j;
    float4 late[8];
#pragma unroll
    for (int s = 0; s < 8; s++) late[s] = w[(size_t)(s + 8) * 4 * 64];
#pragma unroll
    for (int s = 0; s < 8; s++) pair_step(c, H, HS, s, q, j, p.v[s]);
#pragma unroll
    for (int s = 0; s < 8; s++) pair_step(c, H, HS, s + 8, q, j, late[s]);
}

template <int G, int MODE>
__global__ __launch_bounds__(256, (G == 1) ? 3 : 2) void query_kernel(const QArgs a)
{
    VT_SKIP_RETURN(a.skip);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // Region 0 is time-shared: feature-chunk double buffer (layer 1) -> hidden activations per head -> tap-difference double
    // buffers (layer-1 backward, after the d(hidden-1) fragments moved to registers).
    constexpr int R0 = (G * 64 * HS > 2 * 64 * FS + G * 4096) ? G * 64 * HS : 2 * 64 * FS + G * 4096;
    float *Hb = lds;                        // G x [64][HS]
    float *Cb = lds;                        // 2 x [64][FS] feature chunks (fwd)   |   [64][FS] du, [64][FS] dv, weight slab G x 16 KB (bwd)
    float *Go = lds + R0;                   // [64][GS]       output gradient
    float *sPt = Go + 64 * GS;              // [64][3]
    float *sUV = sPt + 64 * 3;              // [4][64][2]
    float *sDf = sUV + 4 * 64 * 2;          // [64] clamped distance (MODE_PROJECT)
    int *sIn = reinterpret_cast<int *>(sDf + 64);  // [64]
    double *sRed = reinterpret_cast<double *>(sIn + 64);     // [8]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
    // XCD-aware block -> (frame, tile) map: the dispatcher places workgroup L on XCD L % 8 (speed only, never correctness), so
    // give every XCD whole frames: all ~108 tiles of a frame then gather from the same few MB of maps through ONE L2.
    int b, tile;
    {
        const int tiles = (a.N + 63) >> 6, L = blockIdx.x;
        if ((a.B & 7) == 0) { const int slot = L >> 3; b = (L & 7) + 8 * (slot / tiles); tile = slot % tiles; }
        else { b = L / tiles; tile = L % tiles; }
    }
    const int n0 = tile * 64;

    // ---- per-point projections (camera.py:52-90, chore_triplane.py:207-251)
    if (tid < 64) {
        const int n = min(n0 + tid, a.N - 1);
        const int pn = a.order ? a.order[n] : n;
        const float *p = a.pts + ((size_t)b * a.N + pn) * 3;
        const float x = p[0], y = p[1], z = p[2];
        float px = a.fx * x / z + a.cx, py = a.fy * y / z + a.cy;
        px = a.crop / 2 + px - a.crop_center[2 * b]; py = a.crop / 2 + py - a.crop_center[2 * b + 1];
        const float nx = 2 * px / a.crop - 1, ny = 2 * py / a.crop - 1;
        sIn[tid] = (pn << 1) | (int)((nx >= -1.0f) && (nx <= 1.0f) && (ny >= -1.0f) && (ny <= 1.0f));     // point index | in-image flag
        const float c0 = x - a.body_center[3 * b], c1 = y - a.body_center[3 * b + 1], c2 = z - a.body_center[3 * b + 2];
        sPt[tid * 3] = x; sPt[tid * 3 + 1] = y; sPt[tid * 3 + 2] = z;
        sUV[(0 * 64 + tid) * 2] = nx;  sUV[(0 * 64 + tid) * 2 + 1] = ny;   // perspective
        sUV[(1 * 64 + tid) * 2] = c2;  sUV[(1 * 64 + tid) * 2 + 1] = c1;   // right
        sUV[(2 * 64 + tid) * 2] = -c0; sUV[(2 * 64 + tid) * 2 + 1] = c1;   // back
        sUV[(3 * 64 + tid) * 2] = c0;  sUV[(3 * 64 + tid) * 2 + 1] = -c2;  // top
    }
    __syncthreads();

    // ---- layer 1: stream the 19 chunks (all heads of the group at once); the tap loads of chunk i+1 are in flight
    //      while the MFMAs of chunk i run (one barrier per chunk thanks to the double buffer)
    Acc8 acc1[G];
#pragma unroll
    for (int g = 0; g < G; g++) acc_zero(acc1[g]);
    Taps tp;
    { int mi, co; chunk_info(0, mi, co); taps_issue(a, b, mi, co, sUV, tid, tp); }
    float4 bw[G][4], bwn[G][4];
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
        for (int s = 0; s < 4; s++) bw[g][s] = reinterpret_cast<const float4 *>(a.hw[g].w1p)[((size_t)(s * 4 + q) * 4 + wave) * 16 + j];
    for (int ci = 0; ci < NCHUNK; ci++) {
        float *buf = Cb + (ci & 1) * 64 * FS;
        taps_store_feat(tp, buf, tid);
        __syncthreads();
        if (ci + 1 < NCHUNK) {
            int mi, co; chunk_info(ci + 1, mi, co); taps_issue(a, b, mi, co, sUV, tid, tp);
#pragma unroll
            for (int g = 0; g < G; g++)
#pragma unroll
                for (int s = 0; s < 4; s++) bwn[g][s] = reinterpret_cast<const float4 *>(a.hw[g].w1p)[((size_t)((ci + 1) * 16 + s * 4 + q) * 4 + wave) * 16 + j];
        }
#ifndef ABL_NOFWDL1
#pragma unroll
        for (int s = 0; s < 4; s++) {
            float2 av[4];
#pragma unroll
            for (int mt = 0; mt < 4; mt++) av[mt] = *reinterpret_cast<const float2 *>(buf + (mt * 16 + j) * FS + 8 * s + 2 * q);
#pragma unroll
            for (int g = 0; g < G; g++) {
#pragma unroll
                for (int mt = 0; mt < 4; mt++) { acc1[g].v[mt][0] = MFMA16(av[mt].x, bw[g][s].x, acc1[g].v[mt][0]); acc1[g].v[mt][1] = MFMA16(av[mt].x, bw[g][s].y, acc1[g].v[mt][1]); }
#pragma unroll
                for (int mt = 0; mt < 4; mt++) { acc1[g].v[mt][0] = MFMA16(av[mt].y, bw[g][s].z, acc1[g].v[mt][0]); acc1[g].v[mt][1] = MFMA16(av[mt].y, bw[g][s].w, acc1[g].v[mt][1]); }
            }
        }
#else
        for (int g = 0; g < G; g++) acc1[g].v[0][0][0] += bw[g][0].x + buf[(j) * FS + q];
#endif
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
            for (int s = 0; s < 4; s++) bw[g][s] = bwn[g][s];
    }
    {   // z_feat = (x, y, z - 2.2): internal channels 608..610 (+ zero pad 611), one plain k-step (k = q)
        float av[4];
#pragma unroll
        for (int mt = 0; mt < 4; mt++) av[mt] = q < 3 ? sPt[(mt * 16 + j) * 3 + q] - (q == 2 ? 2.2f : 0.f) : 0.f;
#pragma unroll
        for (int g = 0; g < G; g++) {
            const float *w = a.hw[g].w1xio + q * 128 + (2 * wave) * 16 + j;
            const float b0 = w[0], b1 = w[16];
#pragma unroll
            for (int mt = 0; mt < 4; mt++) { acc1[g].v[mt][0] = MFMA16(av[mt], b0, acc1[g].v[mt][0]); acc1[g].v[mt][1] = MFMA16(av[mt], b1, acc1[g].v[mt][1]); }
        }
    }
    __syncthreads();        // region 0 changes role: chunk buffers -> hidden activations

    // ---- per head: layers 2..4, objective / upstream gradient, backward to d(hidden-1)
    double loss_acc[2] = {0.0, 0.0};
#pragma unroll
    for (int g = 0; g < G; g++) {
        const HeadW &hw = a.hw[g];
        float *H = Hb + g * 64 * HS;
        Acc8 c;
        WPre wp;
        wprefetch(wp, hw.w2p, wave, lane);
        const unsigned m1 = bias_relu(acc1[g], hw.b1, wave, lane);
        store_hbuf(acc1[g], H, wave, lane);
        __syncthreads();
        gemm128(c, H, hw.w2p, wp, wave, lane);
        wprefetch(wp, hw.w3p, wave, lane);
        const unsigned m2 = bias_relu(c, hw.b2, wave, lane);
        __syncthreads();
        store_hbuf(c, H, wave, lane);
        __syncthreads();
        gemm128(c, H, hw.w3p, wp, wave, lane);
        const unsigned m3 = bias_relu(c, hw.b3, wave, lane);
        __syncthreads();
        store_hbuf(c, H, wave, lane);
        __syncthreads();
        // layer 4: wave w owns the 16 points of M-tile w, N-tile = up to 16 outputs (zero padded)
        f32x4 o4 = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            const float2 *__restrict__ w4 = reinterpret_cast<const float2 *>(hw.w4p) + q * 16 + j;
#pragma unroll
            for (int s = 0; s < 16; s++) {
                const float2 av = *reinterpret_cast<const float2 *>(H + (wave * 16 + j) * HS + 8 * s + 2 * q);
                const float2 bb = w4[s * 64];
                o4 = MFMA16(av.x, bb.x, o4); o4 = MFMA16(av.y, bb.y, o4);
            }
        }
        const float bias4 = hw.b4[j];
        float go[4];    // upstream gradient of output j at points wave*16 + q*4 + r
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int pt = wave * 16 + q * 4 + r, n = n0 + pt;
            const bool valid = n < a.N, live = j < hw.kout;
            const bool inimg = (sIn[pt] & 1) != 0;
            const int pn = sIn[pt] >> 1;
            float val = o4[r] + bias4;
            go[r] = 0.f;
            if (MODE == MODE_FWD) {
                if (hw.id == 0 && !inimg) val = OUT_DIST;                       // df[~in_img] = 5.0 (chore_triplane.py:156-159)
                if (hw.id == 4) val = 1.0f / (1.0f + expf(-val));                // sigmoid on visibility (chore_tri_vis.py:22-27)
                if (valid && live) a.out[g][((size_t)b * hw.kout + j) * a.N + pn] = val;
            } else if (MODE == MODE_BWD) {
                float gg = (valid && live) ? a.gout[g][((size_t)b * hw.kout + j) * a.N + pn] : 0.f;
                if (hw.id == 0 && !inimg) gg = 0.f;
                if (hw.id == 4) { const float s = 1.0f / (1.0f + expf(-val)); gg *= s * (1.0f - s); }
                go[r] = gg;
            } else if (MODE == MODE_HUMAN) {
                if (hw.id == 0) {
                    // df_h = clamp(df[:,0], max=.1).mean()  (recon_fit_base.py:640-647)
                    if (j == 0 && valid) {
                        const float d = inimg ? val : OUT_DIST;
                        loss_acc[0] += (double)fminf(d, 0.1f);
                        if (inimg && d <= 0.1f) go[r] = a.w0 / ((float)a.B * (float)a.N);
                    }
                } else {
                    // part = mean_B sum_N CE(parts, labels)  (recon_fit_behave.py:486): softmax over the 14 logits held by lanes j<14
                    float mx = live ? val : -INFINITY;
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
                    const float e = live ? expf(val - mx) : 0.f;
                    float se = e;
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) se += __shfl_xor(se, o, 64);
                    const int lab = a.labels[pn];
                    if (valid && live) {
                        go[r] = (e / se - (j == lab ? 1.f : 0.f)) * a.w1 / (float)a.B;
                        if (j == lab) loss_acc[1] += (double)(logf(se) - (val - mx));
                    }
                }
            } else if (MODE == MODE_PROJECT) {
                // Generator.approx_surface (recon/gen/generator.py:72-103): target = clamp(df[:, idx], max = threshold); gradient of sum(target)
                if (j == a.df_idx) {
                    const float d = inimg ? val : OUT_DIST;
                    sDf[pt] = fminf(d, a.w0);
                    if (valid && inimg && d <= a.w0) go[r] = 1.0f;
                }
            } else {  // MODE_OBJECT: object = mean_B( mean_N clamp(df[:,1], max=.8) * occ )  (recon_fit_trivis_full.py:155-162)
                if (j == 1 && valid) {
                    const float d = inimg ? val : OUT_DIST, ob = a.occ[b];
                    loss_acc[0] += (double)(fminf(d, 0.8f) * ob);
                    if (inimg && d <= 0.8f) go[r] = a.w0 * ob / ((float)a.B * (float)a.N);
                }
            }
        }
        if (MODE == MODE_FWD) { __syncthreads(); continue; }
        // ---- backward through layer 4: g3 = go[64 x 16] . W4(out,in)[16 x 128]
#pragma unroll
        for (int r = 0; r < 4; r++) Go[(wave * 16 + q * 4 + r) * GS + j] = go[r];
        __syncthreads();
        acc_zero(c);
        {
            const float4 *__restrict__ w = reinterpret_cast<const float4 *>(hw.w4tp) + (q * 4 + wave) * 16 + j;
#pragma unroll
            for (int s = 0; s < 2; s++) pair_step(c, Go, GS, s, q, j, w[s * 4 * 64]);
        }
        wprefetch(wp, hw.w3tp, wave, lane);
        apply_mask(c, m3);
        store_hbuf(c, H, wave, lane);          // H (h3) was last read before the barrier above
        __syncthreads();
        gemm128(c, H, hw.w3tp, wp, wave, lane);    // g2 = g3 . W3(out,in)
        wprefetch(wp, hw.w2tp, wave, lane);
        apply_mask(c, m2);
        __syncthreads();
        store_hbuf(c, H, wave, lane);
        __syncthreads();
        gemm128(c, H, hw.w2tp, wp, wave, lane);    // g1 = g2 . W2(out,in)
        apply_mask(c, m1);
        __syncthreads();
        store_hbuf(c, H, wave, lane);          // H now holds d loss / d (pre-activation 1) of this head
        __syncthreads();
    }

    if (MODE == MODE_HUMAN || MODE == MODE_OBJECT) {
        // block-reduce the loss partials into the fp64 term accumulators
#pragma unroll
        for (int t = 0; t < 2; t++) {
            double s = loss_acc[t];
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (lane == 0) sRed[t * 4 + wave] = s;
        }
        __syncthreads();
        if (tid < 2) {
            double s = sRed[tid * 4] + sRed[tid * 4 + 1] + sRed[tid * 4 + 2] + sRed[tid * 4 + 3];
            if (MODE == MODE_HUMAN) s = tid == 0 ? s / ((double)a.B * a.N) : s / (double)a.B;
            else s = s / ((double)a.B * a.N);
            if (MODE == MODE_HUMAN || tid == 0) atomicAdd(a.terms + tid, s);
        }
    }
    if (MODE == MODE_FWD) return;

    // ---- backward through layer 1 and the gathers: wave w owns the 16 points of M-tile w
    float2 ah[G][16];       // A pair fragments of d(hidden-1): point wave*16 + j, hidden units 8 s + 2 q + {0,1}
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
        for (int s = 0; s < 16; s++) ah[g][s] = *reinterpret_cast<const float2 *>(Hb + g * 64 * HS + (wave * 16 + j) * HS + 8 * s + 2 * q);
    __syncthreads();        // region 0 changes role again: hidden activations -> tap-difference double buffers
    // coordinate-gradient partials of the 4 points (rows) this lane sees, over the channels this lane owns
    float gx[4] = {0.f, 0.f, 0.f, 0.f}, gy[4] = {0.f, 0.f, 0.f, 0.f}, gz[4] = {0.f, 0.f, 0.f, 0.f};
    // Every wave needs the whole 128G x 32 weight slab of a chunk (the waves split the POINTS here): the workgroup stages it once
    // in LDS with the asynchronous global->LDS DMA (16 B per lane, lane-linear destination = the fragment order), no VGPRs.
    float *bu = Cb, *bv = Cb + 64 * FS;                 // tap differences of the current chunk
    float4 *Sl = reinterpret_cast<float4 *>(Cb + 2 * 64 * FS);   // [G][64 kp][16 j] float4 fragments
#define SLAB_DMA(ci_)                                                                                                        \
    _Pragma("unroll") for (int i_ = 0; i_ < 4 * G; i_++) {                                                                    \
        const int idx_ = tid + 256 * i_, g_ = idx_ >> 10, rem_ = idx_ & 1023;                                                \
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const float4 *>(a.hw[g_].w1c) + ((size_t)(rem_ >> 4) * 20 + (ci_)) * 16 + (rem_ & 15), \
                                         (__attribute__((address_space(3))) void *)(Sl + wave * 64 + 256 * i_), 16, 0, 0);    \
    }
    SLAB_DMA(0)
    { int mi, co; chunk_info(0, mi, co); taps_issue(a, b, mi, co, sUV, tid, tp); }
    __syncthreads();
    const float kx = 2.0f / a.crop * a.fx, ky = 2.0f / a.crop * a.fy;
    for (int ci = 0; ci < NCHUNK; ci++) {
        int mi, co; chunk_info(ci, mi, co);
        // d feat[16 pts x 32 ch] = sum_g dh1[g] . W1(out,in)[g][:, chunk];  four independent accumulators
        f32x4 dd[2][2];
#pragma unroll
        for (int x = 0; x < 2; x++)
#pragma unroll
            for (int nt = 0; nt < 2; nt++) dd[x][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifndef ABL_NOB1
#pragma unroll
        for (int s = 0; s < 16; s++) {
#pragma unroll
            for (int g = 0; g < G; g++) {
                const float4 bb = Sl[(g * 64 + 4 * s + q) * 16 + j];
                const int x0 = (G == 2) ? g : 0, x1 = (G == 2) ? g : 1;
                dd[x0][0] = MFMA16(ah[g][s].x, bb.x, dd[x0][0]); dd[x0][1] = MFMA16(ah[g][s].x, bb.y, dd[x0][1]);
                dd[x1][0] = MFMA16(ah[g][s].y, bb.z, dd[x1][0]); dd[x1][1] = MFMA16(ah[g][s].y, bb.w, dd[x1][1]);
            }
        }
#else
        dd[0][0][0] += ah[0][3].x + Sl[tid].x; dd[0][1][0] += ah[0][5].y;
#endif
        const f32x4 d0 = dd[0][0] + dd[1][0], d1 = dd[0][1] + dd[1][1];
        taps_store_grad(tp, bu, bv, tid);
        __syncthreads();                                   // slab(ci) fully consumed, tap differences of chunk ci visible
        if (ci + 1 < NCHUNK) { SLAB_DMA(ci + 1) }
        const int pr = map_proj(mi);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int pt = wave * 16 + q * 4 + r, row = pt * FS;
            const float su = d0[r] * bu[row + j] + d1[r] * bu[row + 16 + j];
            const float sv = d0[r] * bv[row + j] + d1[r] * bv[row + 16 + j];
            // projection Jacobians (camera.py:52-90; chore_triplane.py:220-251)
            if (pr == 0) {
                const float x = sPt[pt * 3], y = sPt[pt * 3 + 1], iz = 1.0f / sPt[pt * 3 + 2];
                gx[r] += su * kx * iz; gy[r] += sv * ky * iz; gz[r] -= (su * kx * x + sv * ky * y) * iz * iz;
            } else if (pr == 1) { gz[r] += su; gy[r] += sv; }     // right (c2, c1)
            else if (pr == 2) { gx[r] -= su; gy[r] += sv; }       // back  (-c0, c1)
            else { gx[r] += su; gz[r] -= sv; }                    // top   (c0, -c2)
        }
        __syncthreads();                                   // slab(ci+1) landed (the barrier drains the DMA); tap buffer free again
        if (ci + 1 < NCHUNK) { int m2i, c2o; chunk_info(ci + 1, m2i, c2o); taps_issue(a, b, m2i, c2o, sUV, tid, tp); }
    }
#undef SLAB_DMA
    // direct xyz features: d feat[608..610] = sum_g dh1 . W1(out,in)[:, 608..611]
    f32x4 dz = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < G; g++) {
        const float *w = a.hw[g].w1xoi + (j & 3);
#pragma unroll
        for (int s = 0; s < 16; s++) {
            const float b0 = j < 4 ? w[(8 * s + 2 * q) * 4] : 0.f, b1 = j < 4 ? w[(8 * s + 2 * q + 1) * 4] : 0.f;
            dz = MFMA16(ah[g][s].x, b0, dz); dz = MFMA16(ah[g][s].y, b1, dz);
        }
    }
    // reduce the channel partials over the 16 lanes that share a row group
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { gx[r] += __shfl_xor(gx[r], o, 64); gy[r] += __shfl_xor(gy[r], o, 64); gz[r] += __shfl_xor(gz[r], o, 64); }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float gy_d = __shfl(dz[r], (lane & 48) + 1, 64), gz_d = __shfl(dz[r], (lane & 48) + 2, 64);
        if (j == 0) {
            const int pt = wave * 16 + q * 4 + r, n = n0 + pt;
            if (n < a.N) {
                const int pn = sIn[pt] >> 1;
                const float g0 = gx[r] + dz[r], g1 = gy[r] + gy_d, g2 = gz[r] + gz_d;
                if (MODE == MODE_PROJECT) {
                    // samples <- samples - normalize(gradient) * target  (F.normalize: g / max(|g|, 1e-12); generator.py:97)
                    const float dft = sDf[pt], sc = dft / fmaxf(sqrtf(g0 * g0 + g1 * g1 + g2 * g2), 1e-12f);
                    float *o = a.pts_out + ((size_t)b * a.N + pn) * 3;
                    o[0] = sPt[pt * 3] - g0 * sc; o[1] = sPt[pt * 3 + 1] - g1 * sc; o[2] = sPt[pt * 3 + 2] - g2 * sc;
                    if (a.dft_out) a.dft_out[(size_t)b * a.N + pn] = dft;
                } else {
                    float *o = a.dpts + ((size_t)b * a.N + pn) * 3;
                    o[0] = g0; o[1] = g1; o[2] = g2;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// handle: weights re-laid out once.  Internal channel order: 608 map channels (im_feat 256, tmpx 64, tri_tmpx 3x32,
// tri_feat 3x64), then x, y, z-2.2, then one zero pad  ->  KTOT = 612.
// ---------------------------------------------------------------------------------------------------
static const int kHeadDims[5] = {2, 9, 14, 3, 1};
static inline int orig_channel(int k) { return k < 256 ? k : (k < 608 ? k + 3 : (k < 611 ? k - 608 + 256 : -1)); }

// [K/2][4 waves][16 j][2 e][2 nt] pair-step fragments of a K x 128 matrix given as get(k, n)
template <typename F>
static void pack_pairs(float *dst, int K, F get)
{
    for (int kp = 0; kp < K / 2; kp++) for (int w = 0; w < 4; w++) for (int j = 0; j < 16; j++) for (int e = 0; e < 2; e++) for (int nt = 0; nt < 2; nt++)
        dst[((((size_t)kp * 4 + w) * 16 + j) * 2 + e) * 2 + nt] = get(2 * kp + e, (2 * w + nt) * 16 + j);
}

int create(Net **out, const float *const *w, const float *const *bvec, const float *cam, void *stream)
{
    VT_REQUIRE(out && w && bvec && cam, "f32q::create: null argument");
    hipStream_t st = vt_stream(stream);
    // per head (floats): w1p 608*128 | w1c 64*20*16*4 | w1xio 4*128 | w1xoi 128*4 | b1 128 | (w2p, w2tp, b2) | (w3p, w3tp, b3) | w4p 128*16 | w4tp 16*128 | b4 16
    const size_t per_head = (size_t)608 * 128 + 64 * 20 * 64 + 512 + 512 + 128 + 2 * (2 * 128 * 128 + 128) + 128 * 16 + 16 * 128 + 16;
    float *host = new float[per_head * 5]();
    Net *h = new Net();
    VT_HIP(hipMalloc(reinterpret_cast<void **>(&h->blob), per_head * 5 * sizeof(float)));
    for (int hd = 0; hd < 5; hd++) {
        float *p = host + per_head * hd; const float *d = h->blob + per_head * hd;
        const int ko = kHeadDims[hd];
        const float *W1 = w[hd * 4];          // (128, 611) reference channel order
        auto w1 = [&](int u, int k) { const int c = orig_channel(k); return c < 0 ? 0.f : W1[(size_t)u * VT_FEAT + c]; };   // internal order
        size_t o = 0;
        HeadW &H = h->head[hd];
        H.kout = ko; H.id = hd;
        H.w1p = d + o; pack_pairs(p + o, 608, [&](int k, int n) { return w1(n, k); }); o += (size_t)608 * 128;
        H.w1c = d + o;
        for (int kp = 0; kp < 64; kp++) for (int c = 0; c < 20; c++) for (int j = 0; j < 16; j++) for (int e = 0; e < 2; e++) for (int nt = 0; nt < 2; nt++) {
            const int k = c * 32 + nt * 16 + j;
            p[o + ((((size_t)kp * 20 + c) * 16 + j) * 2 + e) * 2 + nt] = k < 608 ? w1(2 * kp + e, k) : 0.f;
        }
        o += (size_t)64 * 20 * 64;
        H.w1xio = d + o; for (int k = 0; k < 4; k++) for (int u = 0; u < 128; u++) p[o + k * 128 + u] = w1(u, 608 + k); o += 512;
        H.w1xoi = d + o; for (int u = 0; u < 128; u++) for (int k = 0; k < 4; k++) p[o + u * 4 + k] = w1(u, 608 + k); o += 512;
        H.b1 = d + o; memcpy(p + o, bvec[hd * 4], 128 * sizeof(float)); o += 128;
        for (int l = 1; l <= 2; l++) {
            const float *src = w[hd * 4 + l];      // (out, in)
            const float *fp_ = d + o; pack_pairs(p + o, 128, [&](int k, int n) { return src[n * 128 + k]; }); o += 128 * 128;    // forward: B[k=in][n=out]
            const float *bp_ = d + o; pack_pairs(p + o, 128, [&](int k, int n) { return src[k * 128 + n]; }); o += 128 * 128;    // backward: B[k=out][n=in]
            const float *bb = d + o; memcpy(p + o, bvec[hd * 4 + l], 128 * sizeof(float)); o += 128;
            if (l == 1) { H.w2p = fp_; H.w2tp = bp_; H.b2 = bb; } else { H.w3p = fp_; H.w3tp = bp_; H.b3 = bb; }
        }
        const float *W4 = w[hd * 4 + 3];           // (ko, 128)
        H.w4p = d + o; for (int kp = 0; kp < 64; kp++) for (int j = 0; j < 16; j++) for (int e = 0; e < 2; e++) p[o + ((size_t)kp * 16 + j) * 2 + e] = j < ko ? W4[j * 128 + 2 * kp + e] : 0.f; o += 128 * 16;
        H.w4tp = d + o; pack_pairs(p + o, 16, [&](int k, int n) { return k < ko ? W4[k * 128 + n] : 0.f; }); o += 16 * 128;
        H.b4 = d + o; memcpy(p + o, bvec[hd * 4 + 3], ko * sizeof(float)); o += 16;
    }
    VT_HIP(hipMemcpyAsync(h->blob, host, per_head * 5 * sizeof(float), hipMemcpyHostToDevice, st));
    VT_HIP(hipStreamSynchronize(st));
    delete[] host;
    for (int i = 0; i < 5; i++) h->cam[i] = cam[i];
    *out = h;
    return VT_OK;
}
void destroy(Net *h) { if (!h) return; (void)hipFree(h->blob); delete h; }

static size_t lds_bytes(int G)
{
    const size_t r0 = (size_t)G * 64 * HS > (size_t)2 * 64 * FS + G * 4096 ? (size_t)G * 64 * HS : (size_t)2 * 64 * FS + G * 4096;
    return sizeof(float) * (r0 + 64 * GS + 64 * 3 + 4 * 64 * 2 + 64 + 64) + 8 * sizeof(double);
}

template <int G, int MODE>
static int launch(const QArgs &a, hipStream_t st)
{
    const size_t lds = lds_bytes(G);
    VT_LDS_LIMIT((query_kernel<G, MODE>), lds);
    QArgs b = a; b.skip = vt_skip_flag_of(st);
    hipLaunchKernelGGL((query_kernel<G, MODE>), dim3(((a.N + 63) / 64) * a.B), dim3(256), lds, st, b);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

static int fill_common(QArgs &a, const Net *h, const vt_maps *maps, const float *pts, const float *cc, const float *bc, int B, int N)
{
    VT_REQUIRE(h && maps && pts && cc && bc && B > 0 && N > 0, "vt_query: null argument or empty batch");
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < 8; i++) { VT_REQUIRE(maps->maps[i] && maps->res[i] >= 2, "vt_query: map %d missing", i); a.maps[i] = maps->maps[i]; a.res[i] = maps->res[i]; }
    a.pts = pts; a.crop_center = cc; a.body_center = bc; a.B = B; a.N = N;
    a.fx = h->cam[0]; a.fy = h->cam[1]; a.cx = h->cam[2]; a.cy = h->cam[3]; a.crop = h->cam[4];
    return VT_OK;
}

int forward(const Net *h, const vt_maps *maps, const float *pts, const float *crop_center, const float *body_center,
                                int B, int N, float *df, float *pca, float *parts, float *centers, float *vis, void *stream)
{
    QArgs a; int rc = fill_common(a, h, maps, pts, crop_center, body_center, B, N); if (rc) return rc;
    float *outs[5] = {df, pca, parts, centers, vis};
    int ids[5], n = 0;
    for (int i = 0; i < 5; i++) if (outs[i]) ids[n++] = i;
    VT_REQUIRE(n > 0, "vt_query_forward: no output requested");
    for (int i = 0; i < n; i += 2) {
        const int g = (i + 1 < n) ? 2 : 1;
        for (int k = 0; k < g; k++) { a.hw[k] = h->head[ids[i + k]]; a.out[k] = outs[ids[i + k]]; }
        rc = (g == 2) ? launch<2, MODE_FWD>(a, vt_stream(stream)) : launch<1, MODE_FWD>(a, vt_stream(stream));
        if (rc) return rc;
    }
    return VT_OK;
}

int backward(const Net *h, const vt_maps *maps, const float *pts, const float *crop_center, const float *body_center,
                                 int B, int N, const float *d_df, const float *d_pca, const float *d_parts, const float *d_centers,
                                 const float *d_vis, float *dpts, void *stream)
{
    QArgs a; int rc = fill_common(a, h, maps, pts, crop_center, body_center, B, N); if (rc) return rc;
    VT_REQUIRE(dpts, "vt_query_backward: dpts is null");
    const float *gs[5] = {d_df, d_pca, d_parts, d_centers, d_vis};
    int ids[5], n = 0;
    for (int i = 0; i < 5; i++) if (gs[i]) ids[n++] = i;
    if (n == 0) { VT_HIP(hipMemsetAsync(dpts, 0, sizeof(float) * (size_t)B * N * 3, vt_stream(stream))); return VT_OK; }
    VT_REQUIRE(n <= 2, "vt_query_backward: at most two heads with gradients per call (call again and add for more)");
    a.dpts = dpts;
    for (int k = 0; k < n; k++) { a.hw[k] = h->head[ids[k]]; a.gout[k] = gs[ids[k]]; }
    return n == 2 ? launch<2, MODE_BWD>(a, vt_stream(stream)) : launch<1, MODE_BWD>(a, vt_stream(stream));
}

int human_loss(const Net *h, const vt_maps *maps, const float *pts, const float *crop_center, const float *body_center,
               int B, int N, const int *labels, const int *order, float w_dfh, float w_part, float *dpts, double *terms, void *stream)
{
    QArgs a; int rc = fill_common(a, h, maps, pts, crop_center, body_center, B, N); if (rc) return rc;
    VT_REQUIRE(labels && dpts && terms, "vt_query_human_loss: null argument");
    a.hw[0] = h->head[0]; a.hw[1] = h->head[2]; a.labels = labels; a.order = order; a.w0 = w_dfh; a.w1 = w_part; a.dpts = dpts; a.terms = terms;
    return launch<2, MODE_HUMAN>(a, vt_stream(stream));
}

int object_loss(const Net *h, const vt_maps *maps, const float *pts, const float *crop_center, const float *body_center,
                                    int B, int N, const float *occ, float w_obj, float *dpts, double *terms, void *stream)
{
    QArgs a; int rc = fill_common(a, h, maps, pts, crop_center, body_center, B, N); if (rc) return rc;
    VT_REQUIRE(occ && dpts && terms, "vt_query_object_loss: null argument");
    a.hw[0] = h->head[0]; a.occ = occ; a.w0 = w_obj; a.dpts = dpts; a.terms = terms;
    return launch<1, MODE_OBJECT>(a, vt_stream(stream));
}

int project_step(const Net *h, const vt_maps *maps, const float *pts, const float *crop_center, const float *body_center,
                 int B, int N, int df_idx, float threshold, float *pts_out, float *df_target, void *stream)
{
    QArgs a; int rc = fill_common(a, h, maps, pts, crop_center, body_center, B, N); if (rc) return rc;
    VT_REQUIRE(pts_out && (df_idx == 0 || df_idx == 1), "vt_query_project_step: pts_out is null or df_idx not in {0 (human), 1 (object)}");
    a.hw[0] = h->head[0]; a.df_idx = df_idx; a.w0 = threshold; a.pts_out = pts_out; a.dft_out = df_target;
    return launch<1, MODE_PROJECT>(a, vt_stream(stream));
}

}  // namespace f32q
