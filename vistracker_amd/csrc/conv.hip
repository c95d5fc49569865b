// conv.hip -- 3x3 / stride 1 / pad 1 convolution of the HGFilter encoders (model/HGFilters.py:56-203,
// model/net_util.py:346-396 ConvBlock: three pre-activated 3x3 convolutions, bias-free) as an implicit GEMM on the split-f16 MFMA path of
// the point query: D[cout][pixel] = sum over (tap, cin) W[cout][cin][tap] . X[cin][pixel + tap], fp32 accumulate, operands carried as
// hi = fp16(x), lo = fp16(x - hi) with three MFMAs per product (hi.hi + hi.lo + lo.hi; query.hip).  MIOpen's fp32 convolutions run at
// ~120 TFLOP/s (76 % of the f32-input MFMA peak); this path prices against 839 TFLOP/s.
//   * workgroup = an 8 x 16 pixel tile of one frame x ALL output channels (64 or 128), 4 waves; wave w owns cout 32 w .. +31 (Cout = 128) or
//     16 w .. +15 (Cout = 64) for the 128 pixels = 8 MFMA column tiles (one image row of 16 pixels each): the weight stream, the larger one at
//     1.15 MB per 64 pixels, is amortised over 128 pixels;
//   * per 32-channel chunk the 10 x 18 pixel halo patch is loaded ONCE (NHWC: 128 contiguous bytes per pixel), split and stored as K-block-major
//     planes [4 kb][180 px][8 halves]; the nine taps then read their B fragments from LDS at shifted pixel indices (one conflict-free
//     ds_read_b128 per fragment): 9x fewer global loads and splits than a per-tap gather; double buffered, one barrier per chunk;
//   * weights are packed once per layer in fragment order [chunk][tap][4 waves][NT][hi|lo][64 lanes] and requested one tap ahead;
//   * epilogue: one float4 (4 consecutive output channels of a pixel) per lane and tile, written at a channel offset of a wider NHWC tensor
//     -- the ConvBlock's torch.cat of its three outputs is free.
// Activations are scaled by 2^4 (|x| < 4094; a value beyond the fp16 range poisons the tile's output with NaN -- loud), weights per layer to
// [2^13, 2^14); both scales are powers of two and undone exactly in the epilogue.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define MFMAH(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#define CV_ACT_SCALE 16.0f
#define CV_TH 8
#define CV_TW 16
#define CV_PW (CV_TW + 2)
#define CV_PX ((CV_TH + 2) * CV_PW)         /* 180 halo pixels */
#define CV_SPLIT_MAX 65504.0f

struct vt_conv3x3 {
    uint4 *w;           // [Cin / 32][9][4 waves][NT][hi|lo][64 lanes]
    int cin, cout, nt;
    float inv_scale;    // 1 / (CV_ACT_SCALE * s_W)
};

__device__ __forceinline__ h8 cv_h8(const uint4 v) { return __builtin_bit_cast(h8, v); }
__device__ __forceinline__ void cv_split2(float x0, float x1, unsigned &hi, unsigned &lo, float &rmax)
{
    float r0, r1;
    rmax = fmaxf(fmaxf(rmax, fabsf(x0)), fabsf(x1));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(x0), "v"(x1));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(x1));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(r0), "v"(r1));
}

// sum over the 16 lanes of a DPP row, every lane gets the result
__device__ __forceinline__ float cv_row16_sum(float v)
{
#define CV_DPP(x_, c_) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x_)), (c_), 0xF, 0xF, true))
    v += CV_DPP(v, 0xB1); v += CV_DPP(v, 0x4E); v += CV_DPP(v, 0x141); return v + CV_DPP(v, 0x140);
#undef CV_DPP
}
// K3 = true: the 3 x 3 convolution (halo patch of 10 x 18 pixels, nine taps).  K3 = false: the 1 x 1 convolutions of the encoder (conv_last, l, bl, al,
// the ConvBlock's down-sampling projection: model/HGFilters.py:150-203, net_util.py:364-372) -- plain GEMMs over pixels -- on the same machinery: the
// "patch" is the 8 x 16 tile itself, one tap, an optional per-channel bias, the GroupNorm + ReLU prologue, the residual add and the output
// statistics all shared with the 3 x 3 form (stats_cstride / stats_coff: the channel count and offset of the statistics block when a layer of 256
// output channels runs as two launches of 128).
template <int NT, bool K3, int NCH>
__global__ __launch_bounds__(256, 2) void conv3x3_kernel(const float *__restrict__ in, int in_cstride, int in_coff, int H, int W, int Cin, const uint4 *__restrict__ wpk,
                                                         const float2 *__restrict__ gn_stats, const float *__restrict__ gamma, const float *__restrict__ beta, int groups,
                                                         float *__restrict__ out, int out_cstride, int out_coff, float inv_scale, int cout, double *__restrict__ stats_part,
                                                         const float *__restrict__ res, int res_cstride, int res_coff, float *__restrict__ fin, int fin_cstride, int fin_coff,
                                                         const float *__restrict__ bias, int stats_cstride, int stats_coff,
                                                         double *__restrict__ fstats_part, int fstats_cstride, int fstats_coff)
{
    constexpr int PW = K3 ? CV_PW : CV_TW, PX = K3 ? CV_PX : CV_TH * CV_TW, NLD = K3 ? 6 : 4, NTAP = K3 ? 9 : 1;
    // 1 x 1 form: one tap per chunk makes an iteration nine times shorter than the 3 x 3 one, so the input loads run RING - 1 = 3 chunks ahead of
    // their use (a ring of register sets; the chunk count NCH is a template parameter and the loop fully unrolled, so the ring indices are constants)
    constexpr int RING = K3 ? 1 : 4;
    // two patch buffers of {hi [4 kb][PX], lo [4 kb][PX]} uint4 (sized for the 3 x 3 form; the epilogue's staging tile needs 34 KB of it in both forms)
    __shared__ __attribute__((aligned(16))) uint4 patch[2][2 * 4 * CV_PX];
    __shared__ int sOvf;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
    const int tiles_x = W / CV_TW, tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, b = blockIdx.y;
    const int y0 = ty * CV_TH - (K3 ? 1 : 0), x0 = tx * CV_TW - (K3 ? 1 : 0);            // image position of patch pixel (0, 0)
    const float *__restrict__ inb = in + (size_t)b * H * W * in_cstride + in_coff;
    const int cg = gn_stats ? Cin / groups : 1;
    const int nchunk = K3 ? (Cin >> 5) : NCH;
    if (tid == 0) sOvf = 0;
    float rmax = 0.f;

    f32x4 acc[NT][CV_TH];
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int p = 0; p < CV_TH; p++) acc[nt][p] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // halo patch of one chunk: 180 px x 8 pieces of 16 B = 1440 items, 6 per thread (the last round partly idle)
    float4 ld[RING][NLD];
    unsigned inmask = 0;        // which of this thread's six halo pixels lie inside the image (the same for every chunk)
#pragma unroll
    for (int r = 0; r < NLD; r++) {
        const int it = tid + 256 * r, px = it >> 3, yy = y0 + px / PW, xx = x0 + px % PW;
        inmask |= (unsigned)(it < PX * 8 && yy >= 0 && yy < H && xx >= 0 && xx < W) << r;
    }
    auto issue = [&](int c, int slot) {
#pragma unroll
        for (int r = 0; r < NLD; r++) {
            const int it = tid + 256 * r, px = it >> 3, piece = it & 7;
            const int yy = y0 + px / PW, xx = x0 + px % PW;
            const int yc = min(max(yy, 0), H - 1), xc = min(max(xx, 0), W - 1);
            const float4 v = *reinterpret_cast<const float4 *>(inb + ((size_t)yc * W + xc) * in_cstride + c * 32 + piece * 4);
            ld[slot][r] = v;    // outside pixels are read from a clamped address and zeroed in stage(): the padding pads the RECTIFIED activation
        }
    };
    auto stage = [&](int buf, int c, int slot) {
        uint2 *hi8 = reinterpret_cast<uint2 *>(patch[buf]), *lo8 = reinterpret_cast<uint2 *>(patch[buf] + 4 * PX);
        // fused GroupNorm + ReLU prologue (the pre-activated ConvBlock: GN -> ReLU -> conv): y = max(x * a + s, 0) with a = rstd gamma,
        // s = beta - mean a of this thread's four channels of the chunk (the piece index of a thread is the same in every round)
        float ga[4] = {CV_ACT_SCALE, CV_ACT_SCALE, CV_ACT_SCALE, CV_ACT_SCALE}, gs[4] = {0.f, 0.f, 0.f, 0.f};
        if (gn_stats) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int ch = c * 32 + (tid & 7) * 4 + k;
                const float2 st = gn_stats[b * groups + ch / cg];
                const float a_ = st.y * gamma[ch];
                ga[k] = a_ * CV_ACT_SCALE; gs[k] = (beta[ch] - st.x * a_) * CV_ACT_SCALE;
            }
        }
#pragma unroll
        for (int r = 0; r < NLD; r++) {
            const int it = tid + 256 * r, px = it >> 3, piece = it & 7;
            if (it < PX * 8) {
                uint2 hi, lo;
                const bool inside = (inmask >> r) & 1u;
                float v0 = __builtin_fmaf(ld[slot][r].x, ga[0], gs[0]), v1 = __builtin_fmaf(ld[slot][r].y, ga[1], gs[1]), v2 = __builtin_fmaf(ld[slot][r].z, ga[2], gs[2]), v3 = __builtin_fmaf(ld[slot][r].w, ga[3], gs[3]);
                if (gn_stats) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                if (!inside) v0 = v1 = v2 = v3 = 0.f;
                cv_split2(v0, v1, hi.x, lo.x, rmax);
                cv_split2(v2, v3, hi.y, lo.y, rmax);
                const int idx = (((piece >> 1) * PX + px) << 1) + (piece & 1);
                hi8[idx] = hi; lo8[idx] = lo;
            }
        }
    };
    const unsigned wvo = (unsigned)(wave * NT * 128 + lane);
    // a 32-channel layer runs the 64-channel kernel with its upper rows packed as zeros: the waves that own none of its output channels take part in
    // the staging only (no weight loads, no MFMAs, no stores)
    const bool live = wave * NT * 16 < cout;
    // weight fragments: a ring of three register sets, requested TWO taps ahead of their use (round 3: one tap = 48 MFMAs was shorter than the latency of
    // an L2 hit under load; nine taps per chunk keep the ring aligned across chunks: slot = tap % 3 (3 x 3), chunk % 3 (1 x 1, unrolled))
    uint4 wf[3][NT][2];
#define CV_LOAD_W(slot_, step_)                                                                                      \
    _Pragma("unroll") for (int nt = 0; nt < NT; nt++)                                                                \
        _Pragma("unroll") for (int hl = 0; hl < 2; hl++) wf[slot_][nt][hl] = wpk[(size_t)(step_) * (4 * NT * 128) + wvo + (nt * 2 + hl) * 64];
    auto chunk_mfma = [&](int c) {
        const uint4 *Xhi = patch[c & 1], *Xlo = Xhi + 4 * PX;
        __syncthreads();                                        // patch of chunk c visible; the other buffer's readers are done
#pragma unroll
        for (int t = 0; t < NTAP; t++) {
            const int dy = K3 ? t / 3 : 0, dx = K3 ? t % 3 : 0, step = c * NTAP + t;
            if (live && step + 2 < nchunk * NTAP) { if (K3) { CV_LOAD_W((t + 2) % 3, step + 2) } else { CV_LOAD_W((c + 2) % 3, step + 2) } }
            const int sl = K3 ? (t % 3) : (c % 3);
#pragma unroll
            for (int ph = 0; ph < (live ? 2 : 0); ph++) {       // two halves of four image rows: 8 B fragments live at a time
                h8 xh[4], xl[4];
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const int px = (4 * ph + p + dy) * PW + j + dx;
                    xh[p] = cv_h8(Xhi[q * PX + px]); xl[p] = cv_h8(Xlo[q * PX + px]);
                }
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
#pragma unroll
                    for (int p = 0; p < 4; p++) acc[nt][4 * ph + p] = MFMAH(cv_h8(wf[sl][nt][0]), xh[p], acc[nt][4 * ph + p]);
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
#pragma unroll
                    for (int p = 0; p < 4; p++) acc[nt][4 * ph + p] = MFMAH(cv_h8(wf[sl][nt][0]), xl[p], acc[nt][4 * ph + p]);
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
#pragma unroll
                    for (int p = 0; p < 4; p++) acc[nt][4 * ph + p] = MFMAH(cv_h8(wf[sl][nt][1]), xh[p], acc[nt][4 * ph + p]);
            }
        }
    };
    if (K3) {
        issue(0, 0);
        if (live) { CV_LOAD_W(0, 0) CV_LOAD_W(1, 1) }
        stage(0, 0, 0);
        if (nchunk > 1) issue(1, 0);
        for (int c = 0; c < nchunk; c++) {
            chunk_mfma(c);
            if (c + 1 < nchunk) stage((c + 1) & 1, c + 1, 0);
            if (c + 2 < nchunk) issue(c + 2, 0);
        }
    } else {
#pragma unroll
        for (int k = 0; k < (RING < NCH ? RING : NCH); k++) issue(k, k);
        if (live) { CV_LOAD_W(0, 0) if (NCH > 1) { CV_LOAD_W(1, 1) } }
        stage(0, 0, 0);
        if (RING < NCH) issue(RING, 0);                        // set 0 is free again
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            chunk_mfma(c);
            if (c + 1 < NCH) stage((c + 1) & 1, c + 1, (c + 1) % RING);
            if (c + 1 + RING < NCH) issue(c + 1 + RING, (c + 1) % RING);
        }
    }
#undef CV_LOAD_W
    if (rmax > CV_SPLIT_MAX) sOvf = 1;
    __syncthreads();
    const bool ovf = sOvf != 0;
    // `out` (may be NULL) takes the convolution itself -- what the next convolution of a ConvBlock reads --, `fin` (may be NULL) the block's result
    // for these channels: convolution + residual (model/net_util.py:390-394), so that the concatenation never needs a separate add pass.
    // The D fragments hold 4 channels of 8 x 1 pixels per lane: written from there, the 16 lanes of a row hit 16 different pixels (16 bytes at a
    // stride of the whole channel vector -- the lane pattern that halves the vector-memory rate).  The tile goes through LDS
    // ([pixel][64 channels], one pass per NT) and leaves with 16 adjacent lanes per pixel = 256 contiguous bytes; the residual is read the same way.
    float *__restrict__ ob = out ? out + (size_t)b * H * W * out_cstride + out_coff : nullptr;
    float *__restrict__ fb = fin ? fin + (size_t)b * H * W * fin_cstride + fin_coff : nullptr;
    const float *__restrict__ rb = fin ? res + (size_t)b * H * W * res_cstride + res_coff : nullptr;
    float bv[NT][4];                                            // bias of this lane's four output channels per channel tile (0 without one)
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int r = 0; r < 4; r++) { const int co = (wave * NT + nt) * 16 + 4 * q + r; bv[nt][r] = (bias && co < cout) ? bias[co] : 0.f; }
    constexpr int TS = 68;                                      // floats per staged pixel row: 64 channels + 4 (bank spread)
    float *tile = reinterpret_cast<float *>(patch);            // 128 x 68 floats = 34 KB of the 46 KB patch buffers (all readers passed the barrier above)
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        if (nt) __syncthreads();                                // the readers of the previous pass are done
#pragma unroll
        for (int p = 0; p < CV_TH; p++) {
            float4 v = make_float4(__builtin_fmaf(acc[nt][p][0], inv_scale, bv[nt][0]), __builtin_fmaf(acc[nt][p][1], inv_scale, bv[nt][1]),
                                   __builtin_fmaf(acc[nt][p][2], inv_scale, bv[nt][2]), __builtin_fmaf(acc[nt][p][3], inv_scale, bv[nt][3]));
            if (ovf) v = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
            *reinterpret_cast<float4 *>(tile + (p * CV_TW + j) * TS + wave * 16 + 4 * q) = v;
        }
        __syncthreads();
        float fs[4] = {0.f, 0.f, 0.f, 0.f}, fq[4] = {0.f, 0.f, 0.f, 0.f};       // sum / sum of squares of `fin` over this thread's 8 pixels (4 channels)
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int item = r * 256 + tid, piece = item & 15, px = item >> 4;                  // 128 pixels x 16 pieces of 4 channels
            const int co = ((piece >> 2) * NT + nt) * 16 + (piece & 3) * 4;                     // staged column 4 piece = wave (piece >> 2), row 4 (piece & 3) of its tile
            if (co >= cout) continue;
            const int y = ty * CV_TH + px / CV_TW, x = tx * CV_TW + px % CV_TW;
            const float4 v = *reinterpret_cast<const float4 *>(tile + px * TS + piece * 4);
            if (ob) *reinterpret_cast<float4 *>(ob + ((size_t)y * W + x) * out_cstride + co) = v;
            if (fb) {
                const float4 rr = *reinterpret_cast<const float4 *>(rb + ((size_t)y * W + x) * res_cstride + co);
                const float4 o = make_float4(v.x + rr.x, v.y + rr.y, v.z + rr.z, v.w + rr.w);
                *reinterpret_cast<float4 *>(fb + ((size_t)y * W + x) * fin_cstride + co) = o;
                fs[0] += o.x; fs[1] += o.y; fs[2] += o.z; fs[3] += o.w;
                fq[0] = __builtin_fmaf(o.x, o.x, fq[0]); fq[1] = __builtin_fmaf(o.y, o.y, fq[1]); fq[2] = __builtin_fmaf(o.z, o.z, fq[2]); fq[3] = __builtin_fmaf(o.w, o.w, fq[3]);
            }
        }
        // GroupNorm statistics of the block's RESULT (convolution + residual) for the ConvBlock that reads it next: the piece index of a thread is
        // tid & 15 in every round, so the 4 lanes x 4 waves that share it hold the tile's 128 pixels of its four channels; one block of partials
        // per tile in the layout of vt_groupnorm_stats (the consumer calls vt_groupnorm_finalize instead of a statistics pass over the tensor)
        if (fstats_part) {
            float *xw = tile + 128 * TS;                        // [4 waves][16 pieces][8]: beyond the staged tile (34 KB of the 46 KB patch buffers)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                fs[k] += __shfl_xor(fs[k], 16, 64); fs[k] += __shfl_xor(fs[k], 32, 64);
                fq[k] += __shfl_xor(fq[k], 16, 64); fq[k] += __shfl_xor(fq[k], 32, 64);
            }
            if (lane < 16) {
#pragma unroll
                for (int k = 0; k < 4; k++) { xw[(wave * 16 + lane) * 8 + k] = fs[k]; xw[(wave * 16 + lane) * 8 + 4 + k] = fq[k]; }
            }
            __syncthreads();
            if (tid < 16) {
                const int co = ((tid >> 2) * NT + nt) * 16 + (tid & 3) * 4;
                if (co < cout) {
                    double *pp = fstats_part + (((size_t)blockIdx.x * gridDim.y + b) * fstats_cstride + fstats_coff + co) * 2;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        pp[2 * k] = (double)xw[tid * 8 + k] + (double)xw[(16 + tid) * 8 + k] + (double)xw[(32 + tid) * 8 + k] + (double)xw[(48 + tid) * 8 + k];
                        pp[2 * k + 1] = (double)xw[tid * 8 + 4 + k] + (double)xw[(16 + tid) * 8 + 4 + k] + (double)xw[(32 + tid) * 8 + 4 + k] + (double)xw[(48 + tid) * 8 + 4 + k];
                    }
                }
            }
        }
    }
    // GroupNorm statistics of the OUTPUT for the convolution that follows in a ConvBlock (it normalises exactly these values): per output channel
    // the sum and the sum of squares over this workgroup's 8 x 16 pixels -- 8 rows in the lane, 16 columns across a DPP row -- written as one
    // block of partials [tile][frame][channel][2] (fp64) for vt_groupnorm_finalize.  Saves the statistics pass over the tensor.
    if (stats_part) {
        double *pp = stats_part + (((size_t)blockIdx.x * gridDim.y + b) * stats_cstride + stats_coff) * 2;
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            const int co = (wave * NT + nt) * 16 + 4 * q;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float sm = 0.f, sq = 0.f;
#pragma unroll
                for (int p = 0; p < CV_TH; p++) { const float v = __builtin_fmaf(acc[nt][p][r], inv_scale, bv[nt][r]); sm += v; sq = __builtin_fmaf(v, v, sq); }
                sm = cv_row16_sum(sm); sq = cv_row16_sum(sq);
                if (j == 0 && co + r < cout) { pp[(co + r) * 2] = (double)sm; pp[(co + r) * 2 + 1] = (double)sq; }
            }
        }
    }
}

extern "C" int vt_conv3x3_create(vt_conv3x3 **out, const float *weight, int cout, int cin, void *stream)
{
    VT_REQUIRE(out && weight && (cout == 32 || cout == 64 || cout == 128) && cin >= 32 && cin % 32 == 0, "vt_conv3x3_create: needs Cout in {32, 64, 128} and Cin a multiple of 32");
    const int nt = cout == 128 ? 2 : 1, nchunk = cin / 32;
    float m = 0.f;
    for (size_t i = 0; i < (size_t)cout * cin * 9; i++) m = fmaxf(m, fabsf(weight[i]));
    float sw = 1.0f;
    if (m > 0.f && std::isfinite(m)) { int e; frexpf(m, &e); sw = ldexpf(1.0f, 14 - e); }
    const size_t n16 = (size_t)nchunk * 9 * 4 * nt * 2 * 64;            // uint4 count
    _Float16 *host = new _Float16[n16 * 8];
    // fragment (step = chunk * 9 + tap, wave, nt, hi|lo, lane) halves t: W[cout = (wave nt_count + nt) 16 + (lane & 15)][cin = 32 chunk + 8 (lane >> 4) + t][tap]
    for (int c = 0; c < nchunk; c++) for (int t = 0; t < 9; t++) for (int w = 0; w < 4; w++) for (int n = 0; n < nt; n++) for (int l = 0; l < 64; l++) for (int k = 0; k < 8; k++) {
        const int co = (w * nt + n) * 16 + (l & 15), ci = 32 * c + 8 * (l >> 4) + k;
        const float x = co < cout ? weight[((size_t)co * cin + ci) * 9 + t] * sw : 0.f;          // (Cout, Cin, 3, 3): tap = 3 ky + kx; rows beyond Cout = 32 are zero
        const _Float16 hi = (_Float16)x, lo = (_Float16)(x - (float)hi);
        const size_t base = ((((size_t)(c * 9 + t) * 4 + w) * nt + n) * 2) * 64;
        host[(base + l) * 8 + k] = hi; host[(base + 64 + l) * 8 + k] = lo;
    }
    vt_conv3x3 *h = new vt_conv3x3();
    VT_HIP(hipMalloc(reinterpret_cast<void **>(&h->w), n16 * 16));
    VT_HIP(hipMemcpyAsync(h->w, host, n16 * 16, hipMemcpyHostToDevice, vt_stream(stream)));
    VT_HIP(hipStreamSynchronize(vt_stream(stream)));
    delete[] host;
    h->cin = cin; h->cout = cout; h->nt = nt; h->inv_scale = 1.0f / (CV_ACT_SCALE * sw);
    *out = h;
    return VT_OK;
}
extern "C" void vt_conv3x3_destroy(vt_conv3x3 *h) { if (!h) return; (void)hipFree(h->w); delete h; }

// The general form: GroupNorm + ReLU prologue (gn_stats), the convolution to `out` (or NULL), convolution + residual to `fin` (or NULL; res =
// channels [res_coff, ..) of an NHWC tensor with res_cstride channels), and with stats_ws != NULL the GroupNorm partial sums of the convolution's
// output (one block per 8 x 16 pixel tile, layout of vt_groupnorm_stats' partials) at stats_ws + B * stats_groups doubles: vt_groupnorm_finalize(
// stats_ws, tiles, ...) then yields the {mean, rstd} pairs the next convolution of the ConvBlock consumes, without a statistics pass over the tensor.
extern "C" int vt_conv3x3_forward_block(const vt_conv3x3 *h, const float *in, int in_cstride, int in_coff, const float *gn_stats, const float *gamma,
                                        const float *beta, int groups, int B, int H, int W, float *out, int out_cstride, int out_coff,
                                        const float *res, int res_cstride, int res_coff, float *fin, int fin_cstride, int fin_coff,
                                        double *stats_ws, int stats_groups, void *stream)
{
    return vt_conv3x3_forward_block_stats(h, in, in_cstride, in_coff, gn_stats, gamma, beta, groups, B, H, W, out, out_cstride, out_coff, res, res_cstride, res_coff,
                                          fin, fin_cstride, fin_coff, stats_ws, stats_groups, nullptr, 0, stream);
}
// + the GroupNorm partial sums of `fin` (convolution + residual) for the NEXT ConvBlock, which normalises exactly this tensor: the launches of a block write
// the partials of their channel slices [fin_coff, fin_coff + Cout) into ONE block of fin_cstride channels per tile at fin_stats_ws + B * fin_stats_groups
// doubles; vt_groupnorm_finalize(fin_stats_ws, tiles, B, H * W, fin_cstride, fin_stats_groups, eps) after the last one replaces vt_groupnorm_stats(fin).
extern "C" int vt_conv3x3_forward_block_stats(const vt_conv3x3 *h, const float *in, int in_cstride, int in_coff, const float *gn_stats, const float *gamma,
                                              const float *beta, int groups, int B, int H, int W, float *out, int out_cstride, int out_coff,
                                              const float *res, int res_cstride, int res_coff, float *fin, int fin_cstride, int fin_coff,
                                              double *stats_ws, int stats_groups, double *fin_stats_ws, int fin_stats_groups, void *stream)
{
    VT_REQUIRE(h && in && (out || fin) && B > 0 && H % CV_TH == 0 && W % CV_TW == 0
                   && in_cstride >= in_coff + h->cin && in_cstride % 4 == 0 && in_coff % 4 == 0,
               "vt_conv3x3_forward: needs H %% 8 == 0, W %% 16 == 0 and 16-byte aligned channel slices");
    VT_REQUIRE(!out || (out_cstride >= out_coff + h->cout && out_cstride % 4 == 0 && out_coff % 4 == 0), "vt_conv3x3_forward: bad output slice");
    VT_REQUIRE(!fin || (res && fin_cstride >= fin_coff + h->cout && fin_cstride % 4 == 0 && fin_coff % 4 == 0 && res_cstride >= res_coff + h->cout && res_cstride % 4 == 0 && res_coff % 4 == 0),
               "vt_conv3x3_forward_block: bad residual / result slice");
    VT_REQUIRE(!gn_stats || (gamma && beta && groups > 0 && h->cin % groups == 0), "vt_conv3x3_forward_gn: GroupNorm prologue needs gamma, beta and Cin %% groups == 0");
    const dim3 grid((H / CV_TH) * (W / CV_TW), B);
    const float2 *st2 = reinterpret_cast<const float2 *>(gn_stats);
    VT_REQUIRE(!stats_ws || stats_groups > 0, "vt_conv3x3_forward_block: stats_groups must be positive");
    double *part = stats_ws ? stats_ws + (size_t)B * stats_groups : nullptr;
    VT_REQUIRE(!fin_stats_ws || (fin && fin_stats_groups > 0), "vt_conv3x3_forward_block_stats: the statistics of the result need the result (fin) and a positive group count");
    double *fpart = fin_stats_ws ? fin_stats_ws + (size_t)B * fin_stats_groups : nullptr;
    if (h->nt == 2) hipLaunchKernelGGL((conv3x3_kernel<2, true, 0>), grid, dim3(256), 0, vt_stream(stream), in, in_cstride, in_coff, H, W, h->cin, h->w, st2, gamma, beta, groups, out, out_cstride, out_coff, h->inv_scale, h->cout, part, res, res_cstride, res_coff, fin, fin_cstride, fin_coff, nullptr, h->cout, 0, fpart, fin_cstride, fin_coff);
    else hipLaunchKernelGGL((conv3x3_kernel<1, true, 0>), grid, dim3(256), 0, vt_stream(stream), in, in_cstride, in_coff, H, W, h->cin, h->w, st2, gamma, beta, groups, out, out_cstride, out_coff, h->inv_scale, h->cout, part, res, res_cstride, res_coff, fin, fin_cstride, fin_coff, nullptr, h->cout, 0, fpart, fin_cstride, fin_coff);
    VT_LAUNCH_CHECK();
    return VT_OK;
}
extern "C" int vt_conv3x3_forward_gn_stats(const vt_conv3x3 *h, const float *in, int in_cstride, int in_coff, const float *gn_stats, const float *gamma,
                                           const float *beta, int groups, int B, int H, int W, float *out, int out_cstride, int out_coff, double *stats_ws,
                                           int stats_groups, void *stream)
{
    return vt_conv3x3_forward_block(h, in, in_cstride, in_coff, gn_stats, gamma, beta, groups, B, H, W, out, out_cstride, out_coff, nullptr, 0, 0, nullptr, 0, 0,
                                    stats_ws, stats_groups, stream);
}
extern "C" int vt_conv3x3_tiles(int H, int W) { return (H / CV_TH) * (W / CV_TW); }
extern "C" int vt_conv3x3_forward_gn(const vt_conv3x3 *h, const float *in, int in_cstride, int in_coff, const float *gn_stats, const float *gamma,
                                     const float *beta, int groups, int B, int H, int W, float *out, int out_cstride, int out_coff, void *stream)
{
    return vt_conv3x3_forward_gn_stats(h, in, in_cstride, in_coff, gn_stats, gamma, beta, groups, B, H, W, out, out_cstride, out_coff, nullptr, 0, stream);
}
extern "C" int vt_conv3x3_forward(const vt_conv3x3 *h, const float *in, int B, int H, int W, float *out, int out_cstride, int out_coff, void *stream)
{
    VT_REQUIRE(h, "vt_conv3x3_forward: null handle");
    return vt_conv3x3_forward_gn(h, in, h->cin, 0, nullptr, nullptr, nullptr, 0, B, H, W, out, out_cstride, out_coff, stream);
}


// ---- 1 x 1 convolutions (with bias) on the same kernel ------------------------------------------------------------------------------------------
struct vt_conv1x1 {
    uint4 *w[2];        // per part of <= 128 output channels: [Cin / 32][4 waves][NT][hi|lo][64 lanes]
    float *bias;        // (Cout) or NULL
    int cin, cout, parts, pcout, nt;
    float inv_scale;
};
extern "C" int vt_conv1x1_create(vt_conv1x1 **out, const float *weight, const float *bias, int cout, int cin, void *stream)
{
    VT_REQUIRE(out && weight && (cout == 64 || cout == 128 || cout == 256) && (cin == 32 || cin == 64 || cin == 128 || cin == 256),
               "vt_conv1x1_create: needs Cout in {64, 128, 256} and Cin in {32, 64, 128, 256}");
    vt_conv1x1 *h = new vt_conv1x1();
    h->cin = cin; h->cout = cout; h->parts = cout == 256 ? 2 : 1; h->pcout = cout / h->parts; h->nt = h->pcout == 128 ? 2 : 1; h->bias = nullptr; h->w[0] = h->w[1] = nullptr;
    const int nt = h->nt, nchunk = cin / 32;
    float m = 0.f;
    for (size_t i = 0; i < (size_t)cout * cin; i++) m = fmaxf(m, fabsf(weight[i]));
    float sw = 1.0f;
    if (m > 0.f && std::isfinite(m)) { int e; frexpf(m, &e); sw = ldexpf(1.0f, 14 - e); }
    h->inv_scale = 1.0f / (CV_ACT_SCALE * sw);
    const size_t n16 = (size_t)nchunk * 4 * nt * 2 * 64;
    _Float16 *host = new _Float16[n16 * 8];
    hipStream_t st = vt_stream(stream);
    for (int pt = 0; pt < h->parts; pt++) {
        for (int c = 0; c < nchunk; c++) for (int w = 0; w < 4; w++) for (int n = 0; n < nt; n++) for (int l = 0; l < 64; l++) for (int k = 0; k < 8; k++) {
            const int co = pt * h->pcout + (w * nt + n) * 16 + (l & 15), ci = 32 * c + 8 * (l >> 4) + k;
            const float x = weight[(size_t)co * cin + ci] * sw;
            const _Float16 hi = (_Float16)x, lo = (_Float16)(x - (float)hi);
            const size_t base = ((((size_t)c * 4 + w) * nt + n) * 2) * 64;
            host[(base + l) * 8 + k] = hi; host[(base + 64 + l) * 8 + k] = lo;
        }
        VT_HIP(hipMalloc(reinterpret_cast<void **>(&h->w[pt]), n16 * 16));
        VT_HIP(hipMemcpyAsync(h->w[pt], host, n16 * 16, hipMemcpyHostToDevice, st));
        VT_HIP(hipStreamSynchronize(st));
    }
    delete[] host;
    if (bias) {
        VT_HIP(hipMalloc(reinterpret_cast<void **>(&h->bias), sizeof(float) * cout));
        VT_HIP(hipMemcpyAsync(h->bias, bias, sizeof(float) * cout, hipMemcpyHostToDevice, st));
        VT_HIP(hipStreamSynchronize(st));
    }
    *out = h;
    return VT_OK;
}
extern "C" void vt_conv1x1_destroy(vt_conv1x1 *h) { if (!h) return; (void)hipFree(h->w[0]); if (h->w[1]) (void)hipFree(h->w[1]); if (h->bias) (void)hipFree(h->bias); delete h; }

extern "C" int vt_conv1x1_forward(const vt_conv1x1 *h, const float *in, int in_cstride, int in_coff, const float *gn_stats, const float *gamma, const float *beta, int groups,
                                  int B, int H, int W, float *out, int out_cstride, int out_coff, const float *res, int res_cstride, int res_coff,
                                  double *stats_ws, int stats_groups, void *stream)
{
    VT_REQUIRE(h && in && out && B > 0 && H % CV_TH == 0 && W % CV_TW == 0 && in_cstride >= in_coff + h->cin && in_cstride % 4 == 0 && in_coff % 4 == 0,
               "vt_conv1x1_forward: needs H %% 8 == 0, W %% 16 == 0 and 16-byte aligned channel slices");
    VT_REQUIRE(out_cstride >= out_coff + h->cout && out_cstride % 4 == 0 && out_coff % 4 == 0, "vt_conv1x1_forward: bad output slice");
    VT_REQUIRE(!res || (res_cstride >= res_coff + h->cout && res_cstride % 4 == 0 && res_coff % 4 == 0), "vt_conv1x1_forward: bad residual slice");
    VT_REQUIRE(!gn_stats || (gamma && beta && groups > 0 && h->cin % groups == 0), "vt_conv1x1_forward: GroupNorm prologue needs gamma, beta and Cin %% groups == 0");
    VT_REQUIRE(!stats_ws || stats_groups > 0, "vt_conv1x1_forward: stats_groups must be positive");
    const dim3 grid((H / CV_TH) * (W / CV_TW), B);
    const float2 *st2 = reinterpret_cast<const float2 *>(gn_stats);
    double *part = stats_ws ? stats_ws + (size_t)B * stats_groups : nullptr;
    for (int pt = 0; pt < h->parts; pt++) {
        const int o = pt * h->pcout;
        // with a residual the sum is the layer's only output (`fin`), without one the plain result (`out`): the statistics are those of what is written
        float *o_plain = res ? nullptr : out; float *o_fin = res ? out : nullptr;
        const float *bs = h->bias ? h->bias + o : nullptr;
#define CV_L1(NT_, NCH_) hipLaunchKernelGGL((conv3x3_kernel<NT_, false, NCH_>), grid, dim3(256), 0, vt_stream(stream), in, in_cstride, in_coff, H, W, h->cin, h->w[pt], st2, gamma, beta, groups, \
                                          o_plain, out_cstride, out_coff + o, h->inv_scale, h->pcout, res ? nullptr : part, res, res_cstride, res_coff + o, o_fin, out_cstride, out_coff + o, bs, h->cout, o, \
                                          res ? part : nullptr, h->cout, o)
        const int nch = h->cin / 32;
        if (h->nt == 2) { if (nch == 1) CV_L1(2, 1); else if (nch == 2) CV_L1(2, 2); else if (nch == 4) CV_L1(2, 4); else CV_L1(2, 8); }
        else { if (nch == 1) CV_L1(1, 1); else if (nch == 2) CV_L1(1, 2); else if (nch == 4) CV_L1(1, 4); else CV_L1(1, 8); }
#undef CV_L1
        VT_LAUNCH_CHECK();
    }
    return VT_OK;
}
