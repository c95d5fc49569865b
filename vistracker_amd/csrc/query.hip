// query.hip -- SIF-Net point query for gfx950: projection + 8 bilinear feature gathers + MLP decoders, forward and
// backward-to-coordinates, in ONE kernel per call.
//
// Replaces CHORETriplane.query / CHORETriplaneVisibility.decode / get_preds and the autograd tape behind them
// (model/chore_triplane.py:97-164,207-251; model/chore_tri_vis.py:17-50; model/camera.py:45-90; model/geometry.py:4-14;
// model/chore.py:113-126).  Design (DESIGN.md "point query"):
//   * workgroup = 64 consecutive query points of one frame, 4 wavefronts.
//   * feature maps are channel-last (NHWC): one bilinear tap of a 32-channel chunk is a 128-B burst, 8 lanes x 16 B.
//     Chunks (19 x 32 map channels + the xyz triple) are staged through LDS as the A operand of the layer-1 GEMM;
//     the 611-wide feature vector is never materialised in HBM.
//   * all GEMMs (611->128->128->128->k, and their transposes in the backward) run on v_mfma_f32_16x16x4_f32
//     (exact fp32, no TF32 on gfx950).  Per wave: 4 M-tiles (64 points) x 2 N-tiles (32 hidden units) of accumulators;
//     weights are the B operand, read straight from L2 (0.56 M params per head stay cache resident), both (in,out) and
//     (out,in) copies are kept so that every B fragment is a row-contiguous 64-B segment.
//   * hidden activations move D-layout -> A-layout through one padded LDS buffer per head; ReLU masks stay in
//     registers (1 bit per accumulator element), so the backward needs no activation storage in HBM.
//   * the backward of layer 1 contracts d(features) with the bilinear tap differences chunk by chunk and applies the
//     projection Jacobians in the epilogue: only (B,N,3) gradients are written.
//   * fused objective variants (human: clamp(df_h)-mean + part cross-entropy; object: visibility weighted clamp(df_o))
//     produce the loss terms and the weighted coordinate gradient in the same launch.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

#define KTOT 612            /* 608 map channels + x,y,z-2.2 + 1 zero pad (internal channel order) */
#define NCHUNK 19
#define FS 34               /* LDS stride of a 32-channel chunk row   [pt][32 + 2]  */
#define HS 130              /* LDS stride of a hidden activation row  [pt][128 + 2] */
#define GS 18               /* LDS stride of an output-gradient row   [pt][16 + 2]  */
#define OUT_DIST 5.0f       /* chore.py:93 */

enum { MODE_FWD = 0, MODE_BWD = 1, MODE_HUMAN = 2, MODE_OBJECT = 3 };

struct HeadW {
    const float *w1io, *w1oi, *b1, *w2io, *w2oi, *b2, *w3io, *w3oi, *b3, *w4io, *w4oi, *b4;
    int kout, id;
};

struct vt_sifnet {
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

// stage one 32-channel chunk for the 64 points of the tile: blended features (GRAD = false -> bufA) or the tap
// differences d feat / d u, d feat / d v scaled by (res-1)/2 (GRAD = true -> bufA = d/du, bufB = d/dv)
template <bool GRAD>
__device__ __forceinline__ void gather_chunk(const QArgs &a, int b, int mi, int co, const float *sUV, float *bufA, float *bufB, int tid)
{
    const float *__restrict__ map = a.maps[mi];
    const int R = a.res[mi], C = map_channels(mi), pr = map_proj(mi);
    const int sub = tid & 7, pp = tid >> 3;
    const float sc = 0.5f * (float)(R - 1);
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const int pt = pp + 32 * pass;
        const float u = sUV[(pr * 64 + pt) * 2], v = sUV[(pr * 64 + pt) * 2 + 1];
        // grid_sample, bilinear, align_corners=True, zeros padding (geometry.py:12)
        float ix = (u + 1.0f) * 0.5f * (float)(R - 1), iy = (v + 1.0f) * 0.5f * (float)(R - 1);
        ix = fminf(fmaxf(ix, -2.0f), (float)(R + 1)); iy = fminf(fmaxf(iy, -2.0f), (float)(R + 1));
        const float fxl = floorf(ix), fyl = floorf(iy);
        const int x0 = (int)fxl, y0 = (int)fyl, x1 = x0 + 1, y1 = y0 + 1;
        const float wx1 = ix - fxl, wx0 = (fxl + 1.0f) - ix, wy1 = iy - fyl, wy0 = (fyl + 1.0f) - iy;
        const bool bx0 = x0 >= 0 && x0 < R, bx1 = x1 >= 0 && x1 < R, by0 = y0 >= 0 && y0 < R, by1 = y1 >= 0 && y1 < R;
        const size_t rowb = (size_t)b * R;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 nw = (bx0 && by0) ? *reinterpret_cast<const float4 *>(map + ((rowb + y0) * R + x0) * C + co + sub * 4) : z4;
        const float4 ne = (bx1 && by0) ? *reinterpret_cast<const float4 *>(map + ((rowb + y0) * R + x1) * C + co + sub * 4) : z4;
        const float4 sw = (bx0 && by1) ? *reinterpret_cast<const float4 *>(map + ((rowb + y1) * R + x0) * C + co + sub * 4) : z4;
        const float4 se = (bx1 && by1) ? *reinterpret_cast<const float4 *>(map + ((rowb + y1) * R + x1) * C + co + sub * 4) : z4;
        float2 *dA = reinterpret_cast<float2 *>(bufA + pt * FS + sub * 4);
        if (!GRAD) {
            const float w00 = wx0 * wy0, w10 = wx1 * wy0, w01 = wx0 * wy1, w11 = wx1 * wy1;
            dA[0] = make_float2(nw.x * w00 + ne.x * w10 + sw.x * w01 + se.x * w11, nw.y * w00 + ne.y * w10 + sw.y * w01 + se.y * w11);
            dA[1] = make_float2(nw.z * w00 + ne.z * w10 + sw.z * w01 + se.z * w11, nw.w * w00 + ne.w * w10 + sw.w * w01 + se.w * w11);
        } else {
            float2 *dB = reinterpret_cast<float2 *>(bufB + pt * FS + sub * 4);
            dA[0] = make_float2(((ne.x - nw.x) * wy0 + (se.x - sw.x) * wy1) * sc, ((ne.y - nw.y) * wy0 + (se.y - sw.y) * wy1) * sc);
            dA[1] = make_float2(((ne.z - nw.z) * wy0 + (se.z - sw.z) * wy1) * sc, ((ne.w - nw.w) * wy0 + (se.w - sw.w) * wy1) * sc);
            dB[0] = make_float2(((sw.x - nw.x) * wx0 + (se.x - ne.x) * wx1) * sc, ((sw.y - nw.y) * wx0 + (se.y - ne.y) * wx1) * sc);
            dB[1] = make_float2(((sw.z - nw.z) * wx0 + (se.z - ne.z) * wx1) * sc, ((sw.w - nw.w) * wx0 + (se.w - ne.w) * wx1) * sc);
        }
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
// out[64 x 128 slice of this wave] = H[64 x 128] (A, LDS) x W[128 x 128] (B rows contiguous, global), K = 128
__device__ __forceinline__ void gemm128(Acc8 &c, const float *H, const float *__restrict__ W, int wave, int lane)
{
    acc_zero(c);
    const int q = lane >> 4, j = lane & 15;
#pragma unroll 4
    for (int ks = 0; ks < 32; ks++) {
        const float b0 = W[(ks * 4 + q) * 128 + (2 * wave) * 16 + j], b1 = W[(ks * 4 + q) * 128 + (2 * wave + 1) * 16 + j];
        float av[4];
#pragma unroll
        for (int mt = 0; mt < 4; mt++) av[mt] = H[(mt * 16 + j) * HS + ks * 4 + q];
#pragma unroll
        for (int mt = 0; mt < 4; mt++) { c.v[mt][0] = MFMA16(av[mt], b0, c.v[mt][0]); c.v[mt][1] = MFMA16(av[mt], b1, c.v[mt][1]); }
    }
}

template <int G, int MODE>
__global__ __launch_bounds__(256) void query_kernel(const QArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Hb = lds;                        // G x [64][HS]   hidden activations / d(hidden-1) per head
    float *Fch = Hb + G * 64 * HS;          // [64][FS]       feature chunk (fwd) / d feat / du (bwd)
    float *Gvb = Fch + 64 * FS;             // [64][FS]       d feat / dv (bwd)
    float *Go = Gvb + 64 * FS;              // [64][GS]       output gradient
    float *sPt = Go + 64 * GS;              // [64][3]
    float *sUV = sPt + 64 * 3;              // [4][64][2]
    int *sIn = reinterpret_cast<int *>(sUV + 4 * 64 * 2);  // [64]
    double *sRed = reinterpret_cast<double *>(sIn + 64);     // [8]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
    const int b = blockIdx.y, n0 = blockIdx.x * 64;

    // ---- per-point projections (camera.py:52-90, chore_triplane.py:207-251)
    if (tid < 64) {
        const int n = min(n0 + tid, a.N - 1);
        const float *p = a.pts + ((size_t)b * a.N + n) * 3;
        const float x = p[0], y = p[1], z = p[2];
        float px = a.fx * x / z + a.cx, py = a.fy * y / z + a.cy;
        px = a.crop / 2 + px - a.crop_center[2 * b]; py = a.crop / 2 + py - a.crop_center[2 * b + 1];
        const float nx = 2 * px / a.crop - 1, ny = 2 * py / a.crop - 1;
        sIn[tid] = (nx >= -1.0f) && (nx <= 1.0f) && (ny >= -1.0f) && (ny <= 1.0f);
        const float c0 = x - a.body_center[3 * b], c1 = y - a.body_center[3 * b + 1], c2 = z - a.body_center[3 * b + 2];
        sPt[tid * 3] = x; sPt[tid * 3 + 1] = y; sPt[tid * 3 + 2] = z;
        sUV[(0 * 64 + tid) * 2] = nx;  sUV[(0 * 64 + tid) * 2 + 1] = ny;   // perspective
        sUV[(1 * 64 + tid) * 2] = c2;  sUV[(1 * 64 + tid) * 2 + 1] = c1;   // right
        sUV[(2 * 64 + tid) * 2] = -c0; sUV[(2 * 64 + tid) * 2 + 1] = c1;   // back
        sUV[(3 * 64 + tid) * 2] = c0;  sUV[(3 * 64 + tid) * 2 + 1] = -c2;  // top
    }
    __syncthreads();

    // ---- layer 1: stream the 19 chunks, all heads of the group at once
    Acc8 acc1[G];
#pragma unroll
    for (int g = 0; g < G; g++) acc_zero(acc1[g]);
    for (int ci = 0; ci < NCHUNK; ci++) {
        int mi, co; chunk_info(ci, mi, co);
        // weight fragments of this chunk first (L2 latency overlaps the gather)
        float bw[G][8][2];
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
            for (int ks = 0; ks < 8; ks++) {
                const float *w = a.hw[g].w1io + (size_t)(ci * 32 + ks * 4 + q) * 128 + (2 * wave) * 16 + j;
                bw[g][ks][0] = w[0]; bw[g][ks][1] = w[16];
            }
        gather_chunk<false>(a, b, mi, co, sUV, Fch, nullptr, tid);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 8; ks++) {
            float av[4];
#pragma unroll
            for (int mt = 0; mt < 4; mt++) av[mt] = Fch[(mt * 16 + j) * FS + ks * 4 + q];
#pragma unroll
            for (int g = 0; g < G; g++)
#pragma unroll
                for (int mt = 0; mt < 4; mt++) {
                    acc1[g].v[mt][0] = MFMA16(av[mt], bw[g][ks][0], acc1[g].v[mt][0]);
                    acc1[g].v[mt][1] = MFMA16(av[mt], bw[g][ks][1], acc1[g].v[mt][1]);
                }
        }
        __syncthreads();
    }
    {   // z_feat = (x, y, z - 2.2): internal channels 608..610 (+ zero pad 611)
        float av[4];
#pragma unroll
        for (int mt = 0; mt < 4; mt++) av[mt] = q < 3 ? sPt[(mt * 16 + j) * 3 + q] - (q == 2 ? 2.2f : 0.f) : 0.f;
#pragma unroll
        for (int g = 0; g < G; g++) {
            const float *w = a.hw[g].w1io + (size_t)(608 + q) * 128 + (2 * wave) * 16 + j;
            const float b0 = w[0], b1 = w[16];
#pragma unroll
            for (int mt = 0; mt < 4; mt++) { acc1[g].v[mt][0] = MFMA16(av[mt], b0, acc1[g].v[mt][0]); acc1[g].v[mt][1] = MFMA16(av[mt], b1, acc1[g].v[mt][1]); }
        }
    }

    // ---- per head: layers 2..4, objective / upstream gradient, backward to d(hidden-1)
    double loss_acc[2] = {0.0, 0.0};
#pragma unroll
    for (int g = 0; g < G; g++) {
        const HeadW &hw = a.hw[g];
        float *H = Hb + g * 64 * HS;
        Acc8 c;
        const unsigned m1 = bias_relu(acc1[g], hw.b1, wave, lane);
        store_hbuf(acc1[g], H, wave, lane);
        __syncthreads();
        gemm128(c, H, hw.w2io, wave, lane);
        const unsigned m2 = bias_relu(c, hw.b2, wave, lane);
        __syncthreads();
        store_hbuf(c, H, wave, lane);
        __syncthreads();
        gemm128(c, H, hw.w3io, wave, lane);
        const unsigned m3 = bias_relu(c, hw.b3, wave, lane);
        __syncthreads();
        store_hbuf(c, H, wave, lane);
        __syncthreads();
        // layer 4: wave w owns the 16 points of M-tile w, N-tile = up to 16 outputs (zero padded)
        f32x4 o4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int ks = 0; ks < 32; ks++) o4 = MFMA16(H[(wave * 16 + j) * HS + ks * 4 + q], hw.w4io[(ks * 4 + q) * 16 + j], o4);
        const float bias4 = hw.b4[j];
        float go[4];    // upstream gradient of output j at points wave*16 + q*4 + r
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int pt = wave * 16 + q * 4 + r, n = n0 + pt;
            const bool valid = n < a.N, live = j < hw.kout;
            const bool inimg = sIn[pt] != 0;
            float val = o4[r] + bias4;
            go[r] = 0.f;
            if (MODE == MODE_FWD) {
                if (hw.id == 0 && !inimg) val = OUT_DIST;                       // df[~in_img] = 5.0 (chore_triplane.py:156-159)
                if (hw.id == 4) val = 1.0f / (1.0f + expf(-val));                // sigmoid on visibility (chore_tri_vis.py:22-27)
                if (valid && live) a.out[g][((size_t)b * hw.kout + j) * a.N + n] = val;
            } else if (MODE == MODE_BWD) {
                float gg = (valid && live) ? a.gout[g][((size_t)b * hw.kout + j) * a.N + n] : 0.f;
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
                    const int lab = a.labels[min(n, a.N - 1)];
                    if (valid && live) {
                        go[r] = (e / se - (j == lab ? 1.f : 0.f)) * a.w1 / (float)a.B;
                        if (j == lab) loss_acc[1] += (double)(logf(se) - (val - mx));
                    }
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
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            const float b0 = hw.w4oi[(ks * 4 + q) * 128 + (2 * wave) * 16 + j], b1 = hw.w4oi[(ks * 4 + q) * 128 + (2 * wave + 1) * 16 + j];
#pragma unroll
            for (int mt = 0; mt < 4; mt++) {
                const float av = Go[(mt * 16 + j) * GS + ks * 4 + q];
                c.v[mt][0] = MFMA16(av, b0, c.v[mt][0]); c.v[mt][1] = MFMA16(av, b1, c.v[mt][1]);
            }
        }
        apply_mask(c, m3);
        store_hbuf(c, H, wave, lane);          // H (h3) was last read before the barrier above
        __syncthreads();
        gemm128(c, H, hw.w3oi, wave, lane);    // g2 = g3 . W3(out,in)
        apply_mask(c, m2);
        __syncthreads();
        store_hbuf(c, H, wave, lane);
        __syncthreads();
        gemm128(c, H, hw.w2oi, wave, lane);    // g1 = g2 . W2(out,in)
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
    float ah[G][32];        // A fragments of d(hidden-1): point wave*16 + j, hidden unit ks*4 + q
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
        for (int ks = 0; ks < 32; ks++) ah[g][ks] = Hb[g * 64 * HS + (wave * 16 + j) * HS + ks * 4 + q];
    float du[4][4], dv[4][4];   // [projection][row r]: partial over the channels this lane owns
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
        for (int r = 0; r < 4; r++) { du[p][r] = 0.f; dv[p][r] = 0.f; }
    for (int ci = 0; ci < NCHUNK; ci++) {
        int mi, co; chunk_info(ci, mi, co);
        gather_chunk<true>(a, b, mi, co, sUV, Fch, Gvb, tid);
        f32x4 d0 = (f32x4){0.f, 0.f, 0.f, 0.f}, d1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < G; g++) {
            const float *w = a.hw[g].w1oi + ci * 32 + j;
#pragma unroll 8
            for (int ks = 0; ks < 32; ks++) {
                d0 = MFMA16(ah[g][ks], w[(size_t)(ks * 4 + q) * KTOT], d0);
                d1 = MFMA16(ah[g][ks], w[(size_t)(ks * 4 + q) * KTOT + 16], d1);
            }
        }
        __syncthreads();
        float su[4], sv[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = (wave * 16 + q * 4 + r) * FS;
            su[r] = d0[r] * Fch[row + j] + d1[r] * Fch[row + 16 + j];
            sv[r] = d0[r] * Gvb[row + j] + d1[r] * Gvb[row + 16 + j];
        }
        const int pr = map_proj(mi);
#pragma unroll
        for (int p = 0; p < 4; p++) if (p == pr) {
#pragma unroll
            for (int r = 0; r < 4; r++) { du[p][r] += su[r]; dv[p][r] += sv[r]; }
        }
        __syncthreads();
    }
    // direct xyz features: d feat[608..610] = sum_g dh1 . W1(out,in)[:, 608..611]
    f32x4 dz = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < G; g++) {
        const float *w = a.hw[g].w1oi + 608 + (j & 3);
#pragma unroll 8
        for (int ks = 0; ks < 32; ks++) dz = MFMA16(ah[g][ks], j < 4 ? w[(size_t)(ks * 4 + q) * KTOT] : 0.f, dz);
    }
    // reduce the channel partials over the 16 lanes that share a row group
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { du[p][r] += __shfl_xor(du[p][r], o, 64); dv[p][r] += __shfl_xor(dv[p][r], o, 64); }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float gy_d = __shfl(dz[r], (lane & 48) + 1, 64), gz_d = __shfl(dz[r], (lane & 48) + 2, 64);
        if (j == 0) {
            const int pt = wave * 16 + q * 4 + r, n = n0 + pt;
            if (n < a.N) {
                const float x = sPt[pt * 3], y = sPt[pt * 3 + 1], z = sPt[pt * 3 + 2];
                float gx = dz[r], gy = gy_d, gz = gz_d;
                const float k = 2.0f / a.crop;
                gx += du[0][r] * k * a.fx / z;
                gy += dv[0][r] * k * a.fy / z;
                gz += -du[0][r] * k * a.fx * x / (z * z) - dv[0][r] * k * a.fy * y / (z * z);
                gz += du[1][r]; gy += dv[1][r];          // right (c2, c1)
                gx -= du[2][r]; gy += dv[2][r];          // back  (-c0, c1)
                gx += du[3][r]; gz -= dv[3][r];          // top   (c0, -c2)
                float *o = a.dpts + ((size_t)b * a.N + n) * 3;
                o[0] = gx; o[1] = gy; o[2] = gz;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// MFMA operand-layout self test (A = 16x4, B = 4x16 asymmetric): out = A.B, row-major 16x16
// ---------------------------------------------------------------------------------------------------
__global__ void mfma_selftest_kernel(const float *A, const float *Bm, float *out)
{
    const int lane = threadIdx.x, q = lane >> 4, j = lane & 15;
    f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
    c = MFMA16(A[j * 4 + q], Bm[q * 16 + j], c);
#pragma unroll
    for (int r = 0; r < 4; r++) out[(q * 4 + r) * 16 + j] = c[r];
}
extern "C" int vt_selftest_mfma(const float *A, const float *Bm, float *out, void *stream)
{
    hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0, vt_stream(stream), A, Bm, out);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

// ---------------------------------------------------------------------------------------------------
// handle: weights re-laid out once.  Internal channel order: 608 map channels (im_feat 256, tmpx 64, tri_tmpx 3x32,
// tri_feat 3x64), then x, y, z-2.2, then one zero pad  ->  KTOT = 612.
// ---------------------------------------------------------------------------------------------------
static const int kHeadDims[5] = {2, 9, 14, 3, 1};
static inline int orig_channel(int k) { return k < 256 ? k : (k < 608 ? k + 3 : (k < 611 ? k - 608 + 256 : -1)); }

extern "C" int vt_sifnet_create(vt_sifnet **out, const float *const *w, const float *const *bvec, const float *cam, void *stream)
{
    VT_REQUIRE(out && w && bvec && cam, "vt_sifnet_create: null argument");
    hipStream_t st = vt_stream(stream);
    const size_t per_head = (size_t)KTOT * 128 * 2 + 128 + (128 * 128 * 2 + 128) * 2 + 128 * 16 * 2 + 16;
    float *host = new float[per_head * 5]();
    vt_sifnet *h = new vt_sifnet();
    VT_HIP(hipMalloc(reinterpret_cast<void **>(&h->blob), per_head * 5 * sizeof(float)));
    for (int hd = 0; hd < 5; hd++) {
        float *p = host + per_head * hd; const float *d = h->blob + per_head * hd;
        const int ko = kHeadDims[hd];
        size_t o = 0;
        HeadW &H = h->head[hd];
        H.kout = ko; H.id = hd;
        // layer 1 (128 x 611 in the reference order) -> (in,out) [612][128] and (out,in) [128][612], internal order
        H.w1io = d + o; for (int k = 0; k < KTOT; k++) { const int c = orig_channel(k); for (int u = 0; u < 128; u++) p[o + (size_t)k * 128 + u] = c < 0 ? 0.f : w[hd * 4][(size_t)u * VT_FEAT + c]; } o += (size_t)KTOT * 128;
        H.w1oi = d + o; for (int u = 0; u < 128; u++) for (int k = 0; k < KTOT; k++) { const int c = orig_channel(k); p[o + (size_t)u * KTOT + k] = c < 0 ? 0.f : w[hd * 4][(size_t)u * VT_FEAT + c]; } o += (size_t)KTOT * 128;
        H.b1 = d + o; memcpy(p + o, bvec[hd * 4], 128 * sizeof(float)); o += 128;
        for (int l = 1; l <= 2; l++) {
            const float *src = w[hd * 4 + l];
            const float *io = d + o; for (int i = 0; i < 128; i++) for (int u = 0; u < 128; u++) p[o + i * 128 + u] = src[u * 128 + i]; o += 128 * 128;
            const float *oi = d + o; memcpy(p + o, src, 128 * 128 * sizeof(float)); o += 128 * 128;
            const float *bb = d + o; memcpy(p + o, bvec[hd * 4 + l], 128 * sizeof(float)); o += 128;
            if (l == 1) { H.w2io = io; H.w2oi = oi; H.b2 = bb; } else { H.w3io = io; H.w3oi = oi; H.b3 = bb; }
        }
        H.w4io = d + o; for (int i = 0; i < 128; i++) for (int u = 0; u < ko; u++) p[o + i * 16 + u] = w[hd * 4 + 3][u * 128 + i]; o += 128 * 16;
        H.w4oi = d + o; for (int u = 0; u < ko; u++) for (int i = 0; i < 128; i++) p[o + u * 128 + i] = w[hd * 4 + 3][u * 128 + i]; o += 16 * 128;
        H.b4 = d + o; memcpy(p + o, bvec[hd * 4 + 3], ko * sizeof(float)); o += 16;
    }
    VT_HIP(hipMemcpyAsync(h->blob, host, per_head * 5 * sizeof(float), hipMemcpyHostToDevice, st));
    VT_HIP(hipStreamSynchronize(st));
    delete[] host;
    for (int i = 0; i < 5; i++) h->cam[i] = cam[i];
    *out = h;
    return VT_OK;
}
extern "C" void vt_sifnet_destroy(vt_sifnet *h) { if (!h) return; hipFree(h->blob); delete h; }

static size_t lds_bytes(int G) { return sizeof(float) * ((size_t)G * 64 * HS + 2 * 64 * FS + 64 * GS + 64 * 3 + 4 * 64 * 2 + 64) + 8 * sizeof(double); }

template <int G, int MODE>
static int launch(const QArgs &a, hipStream_t st)
{
    const size_t lds = lds_bytes(G);
    static bool done = false;
    if (!done) { VT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(query_kernel<G, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); done = true; }
    hipLaunchKernelGGL((query_kernel<G, MODE>), dim3((a.N + 63) / 64, a.B), dim3(256), lds, st, a);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

static int fill_common(QArgs &a, const vt_sifnet *h, const vt_maps *maps, const float *pts, const float *cc, const float *bc, int B, int N)
{
    VT_REQUIRE(h && maps && pts && cc && bc && B > 0 && N > 0, "vt_query: null argument or empty batch");
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < 8; i++) { VT_REQUIRE(maps->maps[i] && maps->res[i] >= 2, "vt_query: map %d missing", i); a.maps[i] = maps->maps[i]; a.res[i] = maps->res[i]; }
    a.pts = pts; a.crop_center = cc; a.body_center = bc; a.B = B; a.N = N;
    a.fx = h->cam[0]; a.fy = h->cam[1]; a.cx = h->cam[2]; a.cy = h->cam[3]; a.crop = h->cam[4];
    return VT_OK;
}

extern "C" int vt_query_forward(const vt_sifnet *h, const vt_maps *maps, const float *pts, const float *crop_center, const float *body_center,
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

__global__ void add3_kernel(float *dst, const float *src, long n) { long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) dst[i] += src[i]; }

extern "C" int vt_query_backward(const vt_sifnet *h, const vt_maps *maps, const float *pts, const float *crop_center, const float *body_center,
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

extern "C" int vt_query_human_loss(const vt_sifnet *h, const vt_maps *maps, const float *pts, const float *crop_center, const float *body_center,
                                   int B, int N, const int *labels, float w_dfh, float w_part, float *dpts, double *terms, void *stream)
{
    QArgs a; int rc = fill_common(a, h, maps, pts, crop_center, body_center, B, N); if (rc) return rc;
    VT_REQUIRE(labels && dpts && terms, "vt_query_human_loss: null argument");
    a.hw[0] = h->head[0]; a.hw[1] = h->head[2]; a.labels = labels; a.w0 = w_dfh; a.w1 = w_part; a.dpts = dpts; a.terms = terms;
    return launch<2, MODE_HUMAN>(a, vt_stream(stream));
}

extern "C" int vt_query_object_loss(const vt_sifnet *h, const vt_maps *maps, const float *pts, const float *crop_center, const float *body_center,
                                    int B, int N, const float *occ, float w_obj, float *dpts, double *terms, void *stream)
{
    QArgs a; int rc = fill_common(a, h, maps, pts, crop_center, body_center, B, N); if (rc) return rc;
    VT_REQUIRE(occ && dpts && terms, "vt_query_object_loss: null argument");
    a.hw[0] = h->head[0]; a.occ = occ; a.w0 = w_obj; a.dpts = dpts; a.terms = terms;
    return launch<1, MODE_OBJECT>(a, vt_stream(stream));
}
