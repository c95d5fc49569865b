// query.hip -- SIF-Net point query for gfx950: projection + 8 bilinear feature gathers + MLP decoders, forward and
// backward-to-coordinates, in ONE kernel per call.
//
// Replaces CHORETriplane.query / CHORETriplaneVisibility.decode / get_preds and the autograd tape behind them
// (model/chore_triplane.py:97-164,207-251; model/chore_tri_vis.py:17-50; model/camera.py:45-90; model/geometry.py:4-14;
// model/chore.py:113-126).  Design (DESIGN.md "point query"):
//   * workgroup = 64 consecutive query points of one frame, 4 wavefronts.
//   * feature maps are channel-last (NHWC): one bilinear tap of a 32-channel chunk is a 128-B burst, 8 lanes x 16 B.
//     Chunks (19 x 32 map channels + the xyz triple) are staged through LDS as one operand of the layer-1 GEMM;
//     the 611-wide feature vector is never materialised in HBM.
//   * ARITHMETIC: every GEMM (611->128->128->128->k and the transposes of the backward) runs on v_mfma_f32_16x16x32_f16 with
//     SPLIT operands and fp32 accumulation: a value x (pre-scaled by an exact power of two into the upper fp16 range) is carried
//     as hi = fp16(x), lo = fp16(x - hi), i.e. 22 significand bits, and a product is three MFMAs  hi.hi + hi.lo + lo.hi
//     (the dropped lo.lo term is 2^-22 relative).  That is 16/3 = 5.3x the rate of the f32-input MFMA at an error of
//     ~3 x 2^-22 per product -- two orders of magnitude tighter than the TF32 (10-bit) cuDNN path the reference's nn.Conv1d
//     decoders take on its own GPUs (model/chore.py:113-126).  Weights are split once at handle creation (scale 2^k per matrix,
//     max |w| -> [2^13, 2^14)); activations are scaled by 2^6 (|x| < 1023 representable), upstream gradients are normalised per
//     point to [2^5, 2^6).  All scales are powers of two and are undone exactly in the fp32 epilogues.
//   * orientation: hidden layers are computed transposed, D[n][pt] = W[n][k] . X[k][pt] (weights = A operand, read straight
//     from L2 in fragment order; activations = B operand from LDS), because the D fragment then holds 4 CONSECUTIVE hidden
//     units of one point per lane: the next layer's operand is written with 8-byte LDS stores instead of 2-byte scatters.
//     Per wave: 32 hidden units x 64 points = 2 x 4 tiles of 16x16.
//   * LDS operand planes are K-block major, plane[k / 8][point][8 halves]: a fragment load is one conflict-free ds_read_b128.
//   * ReLU masks stay in registers (1 bit per accumulator element), so the backward needs no activation storage in HBM.
//   * the backward of layer 1 contracts d(features) with the bilinear tap differences chunk by chunk and applies the
//     projection Jacobians in the epilogue: only (B,N,3) gradients are written.
//   * fused objective variants (human: clamp(df_h)-mean + part cross-entropy; object: visibility weighted clamp(df_o))
//     produce the loss terms and the weighted coordinate gradient in the same launch.
#include "common.h"
#include "query_f32.h"
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define MFMAH(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

#define KTOT 612            /* 608 map channels + x,y,z-2.2 + 1 zero pad (internal channel order) */
#define NCHUNK 19
#define NSTEP1 20           /* K32 steps of layer 1: 19 map chunks + the xyz step */
#define TS 36               /* LDS stride (floats) of a tap-difference row [pt][32 + 4] */
#define OUT_DIST 5.0f       /* chore.py:93 */
#define ACT_EXP0 6           /* operand scale U of the forward activations at range level 0: 2^6 (|x| < 1023 representable) */
#define ACT_LEVEL_SHIFT 4    /* every range level divides U by 2^4: level 1 = 2^2 (|x| < 16 368), level 2 = 2^-2 (|x| < 262 016) */
#define ACT_LEVELS 3
#define GO_EXP 5            /* upstream gradients are normalised per point so that the operands of the backward chain stay near [2^5, 2^6) x growth: HeadW::goexp */
#define PROJ_COLS 256       /* columns of the hoisted im_feat projection: hidden-1 pre-activations of head df (0..127) | parts (128..255) */
#define PROJ_C0 8           /* first chunk the layer-1 loops still process when the projection is used (chunks 0..7 = im_feat) */

enum { MODE_FWD = 0, MODE_BWD = 1, MODE_HUMAN = 2, MODE_OBJECT = 3, MODE_PROJECT = 4 };

// Weight fragments (uint4 = 8 halves).  "T" GEMMs: D[row][pt], the weight matrix M[row][k] is the A operand:
//   T-pack : [K/32 steps][4 waves][2 nt][hi|lo][64 lanes]   halves t = M[32 wave + 16 nt + (lane & 15)][32 s + 8 (lane >> 4) + t]
//   w4p    : [4 steps][hi|lo][64 lanes]                     halves t = W4[o = lane & 15][32 s + 8 (lane >> 4) + t]   (B operand of layer 4)
//   w1c    : [20 chunks][4 steps][2 ct][hi|lo][64 lanes]    halves t = W1[u = 32 s + 8 (lane >> 4) + t][c = 32 chunk + 16 ct + (lane & 15)]
// Scales (all exact powers of two).  The operand of layer l is U_l x_l, its weights are stored as s_l W_l, so the accumulators of layer l ARE
// the operands of layer l + 1 with U_(l+1) = s_l U_l: no multiply in the hidden epilogues.  The biases of layers 2, 3 are preloaded into the
// accumulators (U_(l+1) b_l, fp32), the bias of layer 1 rides on the constant-one input channel (internal channel 611) as the weight s_1 b_1.
// s_l lifts max |W_l| into [1, 2) (weight_scale_exp); U_1 is common to all heads and such that U_4 <= 2^ACT_EXP0 at range level 0; a range level
// divides every U_l by 2^ACT_LEVEL_SHIFT.
struct HeadW {
    const uint4 *w1p, *w1c, *w2p, *w2tp, *w3p, *w3tp, *w4p, *w4tp;
    const float *b2, *b3, *b4;      // U_3 b_2 | U_4 b_3 (operand units of the CURRENT range level) | b_4 (plain)
    float cout;             // layer-4 output scale 1 / (U_4 s_4) at the current range level
    float kback;            // backward chain: 1 / (s_1 s_2 s_3 s_4)
    int goexp;              // exponent of the per-point normalisation of the upstream gradient: GO_EXP - log2(s_4 s_3 s_2), so that d(hidden-1)' has the magnitude it had with unit weight scales
    int kout, id;
    int pcol;               // first column of this head in the hoisted im_feat projection (vt_maps.proj), -1: the head is not in it
};

struct vt_sifnet {
    void *blob;             // all weights of the 5 heads
    HeadW head[5];
    float cam[5];
    float *projw;           // [256 im_feat channels][PROJ_COLS] fp32: s_1 W1[u][c] of the heads df | parts (vt_query_build_projection multiplies by U_1)
    const float *bias_dev[5];   // per head: [ACT_LEVELS][512] floats: U_3 b_2 (128) | U_4 b_3 (128) | b_4 (16) | pad
    float cout0[5];         // per head: 1 / (U_4 s_4) at level 0
    int u1_exp;             // U_1 = 2^u1_exp at range level 0 (common to all heads)
    f32q::Net *f32;         // the same decoders packed for the strict-fp32 kernels (query_f32.hip)
    std::atomic<int> precision;     // VT_PRECISION_SPLIT_F16 (default) | VT_PRECISION_FP32: which kernels serve the vt_query_* calls of this handle
};

struct QArgs {
    const float *maps[8];
    int res[8];
    const float *pts, *crop_center, *body_center;
    int B, N;
    float fx, fy, cx, cy, crop;
    float u1, u1inv;        // operand scale of the layer-1 input (features, xyz, the constant one) at the range level of the launch, and its inverse
    HeadW hw[2];
    const float *proj; int pw;      // hoisted layer-1 projection of im_feat (B, res0, res0, pw) or NULL
    float *out[2];          // MODE_FWD
    const float *gout[2];   // MODE_BWD
    float *dpts;
    // fused objectives
    const int *labels; const float *occ; float w0, w1; double *terms;
    const int *order;       // optional processing order of the points: workgroup slot n handles point order[n] (NULL = identity)
    // MODE_HUMAN step form (vt_query_human_step): the gradient is ADDED to what dpts already holds (the keypoint term written by vt_kpts_step), then
    // the vertex acceleration stencil of the batch (fit_SMPLH_30fps.py:196-200 / recon_fit_behave.py:488-495: a_f = 2 v_f - v_(f-1) - v_(f+1)) adds its
    // gradient gs (2 a_f - a_(f-1) - a_(f+1)) and its share of the term -- vt_accel_loss without a launch and without a pass over (B, V, 3)
    int accum; float accel_gs; double *term_accel;
    // surface projection step (MODE_PROJECT): w0 = clamp threshold
    int df_idx; float *pts_out, *dft_out;
    const int *skip;        // device-side early stop of the fit that owns the stream (vt_stream_set_skip_flag) or NULL
    unsigned long long *clk;    // clock probe (vt_query_set_clock_probe: bench.py's solo leg) or NULL: every 1024th workgroup adds its life time in shader clocks
                                // (s_memtime) and in 100 MHz ticks (s_memrealtime) -- the clock the chip SUSTAINED during the launch, i.e. the launch in clocks
};

// chunk i (32 channels) -> map index, channel offset inside the map; map -> channels, projection.  Tables in constant memory: the
// uniform lookups are scalar loads instead of compare-and-branch chains (a taken branch restarts the instruction fetch).
//   chunks 0-7 im_feat (256 ch), 8-9 tmpx (64), 10-12 tri_tmpx right/back/top (32 each), 13-18 tri_feat right/back/top (64 each)
__constant__ int2 kChunk[NCHUNK + 1] = {{0, 0}, {0, 32}, {0, 64}, {0, 96}, {0, 128}, {0, 160}, {0, 192}, {0, 224}, {1, 0}, {1, 32}, {2, 0}, {3, 0}, {4, 0},
                                        {5, 0}, {5, 32}, {6, 0}, {6, 32}, {7, 0}, {7, 32}, {0, 0}};
__constant__ int2 kMap[8] = {{256, 0}, {64, 0}, {32, 1}, {32, 2}, {32, 3}, {64, 1}, {64, 2}, {64, 3}};      // {channels, projection}
__device__ __forceinline__ int map_channels(int mi) { return kMap[mi].x; }
__device__ __forceinline__ int map_proj(int mi) { return kMap[mi].y; }
__device__ __forceinline__ void chunk_info(int i, int &mi, int &co) { const int2 c = kChunk[i]; mi = c.x; co = c.y; }

__device__ __forceinline__ h8 as_h8(const uint4 v) { return __builtin_bit_cast(h8, v); }
// identity the optimiser cannot see through: what is computed from the result is NOT common with what was computed from the argument (no cross-phase CSE, hence
// no value kept alive -- or spilled -- from one phase of the kernel to a far later one just to save its recomputation)
#ifdef NO_LAUNDER      /* A/B builds only */
__device__ __forceinline__ int launder_s(int x) { return x; }
__device__ __forceinline__ int launder_v(int x) { return x; }
#else
__device__ __forceinline__ int launder_s(int x) { x = __builtin_amdgcn_readfirstlane(x); asm("" : "+s"(x)); return x; }
__device__ __forceinline__ int launder_v(int x) { asm("" : "+v"(x)); return x; }
#endif
// All-reduce over the 16 lanes of a DPP row (lanes j = 0..15 of one lane group q) with data-parallel-primitive operands: xor 1, xor 2 inside the quads,
// then row_half_mirror and row_mirror -- four VALU instructions with the cross-lane move folded in, instead of four ds_bpermute round trips through
// the LDS crossbar (address VALU + LDS instruction + wait each), which is what __shfl_xor compiles to.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_max(float v)
{
    v = fmaxf(v, dpp_mov<0xB1>(v)); v = fmaxf(v, dpp_mov<0x4E>(v)); v = fmaxf(v, dpp_mov<0x141>(v)); return fmaxf(v, dpp_mov<0x140>(v));
}
__device__ __forceinline__ float row16_sum(float v)
{
    v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v); return v + dpp_mov<0x140>(v);
}
// x (already scaled) -> hi = fp16(x), lo = fp16(x - hi); x - hi is exact in fp32.  Two elements cost 4 VALU instructions: one packed
// RTN conversion, two mixed-precision FMAs (v_fma_mix_f32: -hi * 1 + x, reading hi straight from its packed half), one packed conversion.
// `rmax` tracks max |x| of everything this thread ever splits (one v_max3_f32 per pair): a value beyond the fp16 range would become inf in
// `hi`, NaN a product later -- and a ReLU (fmaxf(NaN, 0) = 0) would turn that into a FINITE wrong result.  The kernels publish rmax > SPLIT_MAX
// to the workgroup and poison their outputs with NaN, so the failure is loud (and the host layer falls back to the strict-fp32 kernels).
#define SPLIT_MAX 65504.0f
#define OVF_PUBLISH() do { if (rmax > SPLIT_MAX) *sOvf = 1; } while (0)       /* always ahead of the barrier that precedes the readers of *sOvf */
__device__ __forceinline__ void split2(float x0, float x1, unsigned &hi, unsigned &lo, float &rmax)
{
    float r0, r1;
    rmax = fmaxf(fmaxf(rmax, fabsf(x0)), fabsf(x1));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(x0), "v"(x1));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(x1));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(r0), "v"(r1));
}
__device__ __forceinline__ void split4(float x0, float x1, float x2, float x3, uint2 &hi, uint2 &lo, float &rmax)
{
    split2(x0, x1, hi.x, lo.x, rmax);
    split2(x2, x3, hi.y, lo.y, rmax);
}

// Bilinear sampling, split in two so that nothing is recomputed per chunk:
//   TapGeom -- per MAP (8 of them over the 19 chunks): the four texel byte offsets (32-bit, inside the frame's map) and the four tap
//              coefficients of two points per thread (thread = point pp / pp+32, 16-B piece `sub` of a tap row).  Forward (NC = 1):
//              the bilinear weights x U_1 (QArgs::u1); backward (NC = 2): the coefficients of d feat / d u and d feat / d v,
//              d/du = sc ((ne - nw) wy0 + (se - sw) wy1),  d/dv = sc ((sw - nw) wx0 + (se - ne) wx1), sc = (res - 1) / 2.
//              Out-of-bounds taps (zeros padding) get coefficient 0 and are loaded from a clamped, valid texel: plain global_load,
//              no divergent branch, no select between a global and a private address.
//   Taps    -- per CHUNK: the 4 x 2 float4 tap loads, in flight in registers while the previous chunk's MFMAs run
template <int NC> struct TapGeom { unsigned o[2][4]; float c[NC][2][4]; };
struct Taps { float4 t[2][4]; };
// Buffer loads (MUBUF): a 128-bit resource descriptor in SGPRs (uniform base + size) + a 32-bit byte offset per lane + an SGPR offset + a 12-bit
// immediate.  The flat/global form needs a 64-bit address per lane -- one or two VALU instructions per load (v_lshl_add_u64 / v_add_co + v_addc:
// 26 M INT64 + part of the 71 M INT32 of the 462 M VALU wave-instructions of a launch, profiles/r02_experiments.md) -- the buffer form none, and
// a read beyond the size returns zeros instead of faulting.
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4_t;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *base, unsigned bytes)
{
    // base and size are workgroup-uniform at every call site, but hipcc cannot always prove it (pointers picked from the by-value argument
    // struct with a run-time map index) and would wrap EVERY load in a waterfall loop: pin the descriptor to SGPRs
    const unsigned long long p = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo), 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ float4 bload_f4(rsrc_t r, unsigned voff, unsigned soff) { return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)__builtin_amdgcn_readfirstlane(soff), 0)); }
__device__ __forceinline__ uint4 bload_u4(rsrc_t r, unsigned voff, unsigned soff) { return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)__builtin_amdgcn_readfirstlane(soff), 0)); }
#define GATHER_F4(r_, voff_, soff_) bload_f4(r_, voff_, soff_)
#define GATHER_P4(r_, voff_, soff_) bload_f4(r_, voff_, soff_)

template <int NC>
__device__ __forceinline__ void taps_geom(const QArgs &a, int mi, const float *sUV, int tid, TapGeom<NC> &g)
{
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
        const float wx1 = ix - fxl, wy1 = iy - fyl;
        const bool bx0 = x0 >= 0 && x0 < R, bx1 = x1 >= 0 && x1 < R, by0 = y0 >= 0 && y0 < R, by1 = y1 >= 0 && y1 < R;
        const bool i0 = bx0 && by0, i1 = bx1 && by0, i2 = bx0 && by1, i3 = bx1 && by1;
        const int xc0 = min(max(x0, 0), R - 1), xc1 = min(max(x1, 0), R - 1), yc0 = min(max(y0, 0), R - 1), yc1 = min(max(y1, 0), R - 1);
        const unsigned r0 = (unsigned)(yc0 * R) * (unsigned)C + (unsigned)(sub * 4), r1 = (unsigned)(yc1 * R) * (unsigned)C + (unsigned)(sub * 4);
        g.o[pass][0] = (r0 + (unsigned)(xc0 * C)) * 4u; g.o[pass][1] = (r0 + (unsigned)(xc1 * C)) * 4u;
        g.o[pass][2] = (r1 + (unsigned)(xc0 * C)) * 4u; g.o[pass][3] = (r1 + (unsigned)(xc1 * C)) * 4u;
        if (NC == 1) {
            const float wx0 = 1.0f - wx1, wy0s = (1.0f - wy1) * a.u1, wy1s = wy1 * a.u1;
            g.c[0][pass][0] = i0 ? wx0 * wy0s : 0.f; g.c[0][pass][1] = i1 ? wx1 * wy0s : 0.f;
            g.c[0][pass][2] = i2 ? wx0 * wy1s : 0.f; g.c[0][pass][3] = i3 ? wx1 * wy1s : 0.f;
        } else {
            const float sx1 = wx1 * sc, sy1 = wy1 * sc, sx0 = sc - sx1, sy0 = sc - sy1;
            g.c[0][pass][0] = i0 ? -sy0 : 0.f; g.c[0][pass][1] = i1 ? sy0 : 0.f; g.c[0][pass][2] = i2 ? -sy1 : 0.f; g.c[0][pass][3] = i3 ? sy1 : 0.f;
            g.c[NC - 1][pass][0] = i0 ? -sx0 : 0.f; g.c[NC - 1][pass][1] = i1 ? -sx1 : 0.f; g.c[NC - 1][pass][2] = i2 ? sx0 : 0.f; g.c[NC - 1][pass][3] = i3 ? sx1 : 0.f;
        }
    }
}
// Shared geometry: taps_geom above is evaluated by all 256 threads -- 8 threads (the 16-byte pieces of a texel) times 2 passes per point -- for
// every map in BOTH layer-1 loops: ~150 VALU instructions per thread and map, 17 % of the kernel's VALU stream, 8-fold redundant.  Instead ONE
// wave computes the 64 points of a map (lane = point), writes {4 texel byte offsets, 4 + 4 coefficients} to a two-slot LDS ring a map ahead, and
// every thread picks up the entries of its two points (3 + 3 ds_read_b128, 8 integer adds for its piece).  Slot = map index & 1; the maps
// follow each other in index order and every iteration of the loops has a barrier between the ring's writes and reads (see the call sites).
#define GEO_STRIDE 12
template <int NC>
__device__ __forceinline__ void geom_compute(const QArgs &a, int mi, const float *sUV, int pt, float *sGeo)
{
    const int R = a.res[mi], C = map_channels(mi), pr = map_proj(mi);
    const float sc = 0.5f * (float)(R - 1);
    const float u = sUV[(pr * 64 + pt) * 2], v = sUV[(pr * 64 + pt) * 2 + 1];
    // grid_sample, bilinear, align_corners=True, zeros padding (geometry.py:12) -- the arithmetic of taps_geom, piece 0
    float ix = (u + 1.0f) * 0.5f * (float)(R - 1), iy = (v + 1.0f) * 0.5f * (float)(R - 1);
    ix = fminf(fmaxf(ix, -2.0f), (float)(R + 1)); iy = fminf(fmaxf(iy, -2.0f), (float)(R + 1));
    const float fxl = floorf(ix), fyl = floorf(iy);
    const int x0 = (int)fxl, y0 = (int)fyl, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fxl, wy1 = iy - fyl;
    const bool bx0 = x0 >= 0 && x0 < R, bx1 = x1 >= 0 && x1 < R, by0 = y0 >= 0 && y0 < R, by1 = y1 >= 0 && y1 < R;
    const bool i0 = bx0 && by0, i1 = bx1 && by0, i2 = bx0 && by1, i3 = bx1 && by1;
    const int xc0 = min(max(x0, 0), R - 1), xc1 = min(max(x1, 0), R - 1), yc0 = min(max(y0, 0), R - 1), yc1 = min(max(y1, 0), R - 1);
    const unsigned r0 = (unsigned)(yc0 * R) * (unsigned)C, r1 = (unsigned)(yc1 * R) * (unsigned)C;
    uint4 o = make_uint4((r0 + (unsigned)(xc0 * C)) * 4u, (r0 + (unsigned)(xc1 * C)) * 4u, (r1 + (unsigned)(xc0 * C)) * 4u, (r1 + (unsigned)(xc1 * C)) * 4u);
    float4 c0, c1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (NC == 1) {
        const float wx0 = 1.0f - wx1, wy0s = (1.0f - wy1) * a.u1, wy1s = wy1 * a.u1;
        c0 = make_float4(i0 ? wx0 * wy0s : 0.f, i1 ? wx1 * wy0s : 0.f, i2 ? wx0 * wy1s : 0.f, i3 ? wx1 * wy1s : 0.f);
    } else {
        const float sx1 = wx1 * sc, sy1 = wy1 * sc, sx0 = sc - sx1, sy0 = sc - sy1;
        c0 = make_float4(i0 ? -sy0 : 0.f, i1 ? sy0 : 0.f, i2 ? -sy1 : 0.f, i3 ? sy1 : 0.f);
        c1 = make_float4(i0 ? -sx0 : 0.f, i1 ? -sx1 : 0.f, i2 ? sx0 : 0.f, i3 ? sx1 : 0.f);
    }
    float *e = sGeo + (mi & 1) * 64 * GEO_STRIDE + pt * GEO_STRIDE;
    *reinterpret_cast<uint4 *>(e) = o; *reinterpret_cast<float4 *>(e + 4) = c0;
    if (NC == 2) *reinterpret_cast<float4 *>(e + 8) = c1;
}
template <int NC>
__device__ __forceinline__ void geom_fetch(int mi, const float *sGeo, int tid, TapGeom<NC> &g)
{
    const int sub = tid & 7, pp = tid >> 3;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const float *e = sGeo + (mi & 1) * 64 * GEO_STRIDE + (pp + 32 * pass) * GEO_STRIDE;
        const uint4 o = *reinterpret_cast<const uint4 *>(e);
        g.o[pass][0] = o.x + (unsigned)(sub * 16); g.o[pass][1] = o.y + (unsigned)(sub * 16); g.o[pass][2] = o.z + (unsigned)(sub * 16); g.o[pass][3] = o.w + (unsigned)(sub * 16);
#pragma unroll
        for (int n = 0; n < NC; n++) {
            const float4 c = *reinterpret_cast<const float4 *>(e + 4 + 4 * n);
            g.c[n][pass][0] = c.x; g.c[n][pass][1] = c.y; g.c[n][pass][2] = c.z; g.c[n][pass][3] = c.w;
        }
    }
}
// issue the tap loads of chunk (mi, co): uniform base = frame b of map mi + channel offset, per-lane 32-bit byte offsets from the geometry
template <int NC>
__device__ __forceinline__ void taps_issue(const QArgs &a, int b, int mi, int co, const TapGeom<NC> &g, Taps &r)
{
    const int R = a.res[mi], C = map_channels(mi);
    const rsrc_t fb = make_rsrc(a.maps[mi] + (size_t)b * R * R * C, (unsigned)(R * R * C) * 4u);
#pragma unroll
    for (int pass = 0; pass < 2; pass++)
#pragma unroll
        for (int k = 0; k < 4; k++) r.t[pass][k] = GATHER_F4(fb, g.o[pass][k], (unsigned)co * 4u);
}
#define TAPSUM_(c_, w_) __builtin_fmaf(se.c_, (w_)[3], __builtin_fmaf(sw.c_, (w_)[2], __builtin_fmaf(ne.c_, (w_)[1], nw.c_ * (w_)[0])))
// blend the taps to (scaled) features, split, and store into the K-block-major chunk planes [4 kb][64 pt][8 halves] (forward) ...
// Layout of a chunk's operand planes.  K-block major ([4 kb][64 pt][8 halves], PM = false: what the hidden-activation planes use) makes the STORES of this function a
// 4-way bank conflict: ds_write_b64 is serviced in groups of 16 consecutive lanes on 32 banks (MI355X_MICROARCH.md, LDS), a group here is 2 points x 8 pieces, and the
// four K blocks of a point sit 1 KB apart -- the same banks (round 5's counters: SQ_LDS_BANK_CONFLICT = 24 % of the LDS-active cycles; this store is the largest share).
// PM = true (round 6, -DCHUNK_PM=1): POINT major, [64 pt][4 slots of 8 halves], a point's K block q in slot q ^ m(q) ^ ((pt >> 2) & 3), m = {0, 3, 1, 2}:
// a group's 16 lanes now write 128 contiguous bytes (conflict-free), and the fragment reads (ds_read_b128: four groups {0-3, 12-15, 20-27}, ... on 64 banks) stay
// conflict-free because the Latin square puts the 16 lanes of every group on 16 different 16-byte slots of the 256-byte bank row (chunk_slot / k32_step_chunk).
__device__ __forceinline__ int chunk_slot(int pt, int q) { return pt * 4 + (((pt >> 2) & 3) ^ ((0x2130 >> (4 * q)) & 3)); }       // uint4 index inside a plane
// MEASURED (round 6, same box, bit-identical results): PM = true is 0.3 % (two-head kernel) / 1.2 % (one-head kernels) SLOWER than the conflicted layout -- the LDS
// array is 35 % busy in these kernels, the conflicts cost LDS cycles nobody was waiting for, and the swizzled addresses cost a few VALU and a different schedule
// (profiles/r06_query_ab.txt).  The conflict-free form stays as a switch (-DCHUNK_PM=1) with the explanation of the counter; the default is the old layout.
#ifndef CHUNK_PM
#define CHUNK_PM 0
#endif
template <bool PM = false>
__device__ __forceinline__ void taps_store_feat(const Taps &r, const TapGeom<1> &g, uint2 *hi8, uint2 *lo8, int tid, float &rmax)
{
    const int sub = tid & 7, pp = tid >> 3;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const float4 nw = r.t[pass][0], ne = r.t[pass][1], sw = r.t[pass][2], se = r.t[pass][3];
        const float *w = g.c[0][pass];
        uint2 hi, lo;
        split4(TAPSUM_(x, w), TAPSUM_(y, w), TAPSUM_(z, w), TAPSUM_(w, w), hi, lo, rmax);
        const int idx = PM ? (chunk_slot(pp + 32 * pass, sub >> 1) << 1) + (sub & 1) : (((sub >> 1) * 64 + pp + 32 * pass) << 1) + (sub & 1);
        hi8[idx] = hi; lo8[idx] = lo;
    }
}
// ---- hoisted im_feat projection (USEP): layer 1 is linear in the features and the features are bilinear blends of texels, so
//      W1 . (sum_t w_t texel_t) = sum_t w_t (W1 . texel_t): the product P = U_1 s_1 W1[:, im_feat] . texel of every im_feat texel is
//      computed ONCE per batch (vt_query_build_projection; maps and weights do not change over the ~730 Adam steps of a batch) and the
//      256 im_feat channels -- 8 of the 19 chunks of both layer-1 loops -- become a 4-tap blend of P rows (forward, straight into the
//      accumulator fragments) and 4 dot products of P rows with d(hidden-1) (backward).  Same bytes gathered as the im_feat taps.
// Geometry of point pt in the im_feat map (perspective projection, sUV slot 0); texel offsets in floats of a (R, R, pw) array.
__device__ __forceinline__ void proj_geom(const float *sUV, int pt, int R, int pw, unsigned (&o)[4], float (&c0)[4], float (&c1)[4], bool grad)
{
    const float u = sUV[pt * 2], v = sUV[pt * 2 + 1];
    float ix = (u + 1.0f) * 0.5f * (float)(R - 1), iy = (v + 1.0f) * 0.5f * (float)(R - 1);
    ix = fminf(fmaxf(ix, -2.0f), (float)(R + 1)); iy = fminf(fmaxf(iy, -2.0f), (float)(R + 1));
    const float fxl = floorf(ix), fyl = floorf(iy);
    const int x0 = (int)fxl, y0 = (int)fyl, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fxl, wy1 = iy - fyl;
    const bool bx0 = x0 >= 0 && x0 < R, bx1 = x1 >= 0 && x1 < R, by0 = y0 >= 0 && y0 < R, by1 = y1 >= 0 && y1 < R;
    const bool i0 = bx0 && by0, i1 = bx1 && by0, i2 = bx0 && by1, i3 = bx1 && by1;
    const int xc0 = min(max(x0, 0), R - 1), xc1 = min(max(x1, 0), R - 1), yc0 = min(max(y0, 0), R - 1), yc1 = min(max(y1, 0), R - 1);
    o[0] = (unsigned)(yc0 * R + xc0) * (unsigned)pw; o[1] = (unsigned)(yc0 * R + xc1) * (unsigned)pw;
    o[2] = (unsigned)(yc1 * R + xc0) * (unsigned)pw; o[3] = (unsigned)(yc1 * R + xc1) * (unsigned)pw;
    if (!grad) {
        const float wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
        c0[0] = i0 ? wx0 * wy0 : 0.f; c0[1] = i1 ? wx1 * wy0 : 0.f; c0[2] = i2 ? wx0 * wy1 : 0.f; c0[3] = i3 ? wx1 * wy1 : 0.f;
    } else {
        const float sc = 0.5f * (float)(R - 1);
        const float sx1 = wx1 * sc, sy1 = wy1 * sc, sx0 = sc - sx1, sy0 = sc - sy1;
        c0[0] = i0 ? -sy0 : 0.f; c0[1] = i1 ? sy0 : 0.f; c0[2] = i2 ? -sy1 : 0.f; c0[3] = i3 ? sy1 : 0.f;
        c1[0] = i0 ? -sx0 : 0.f; c1[1] = i1 ? -sx1 : 0.f; c1[2] = i2 ? sx0 : 0.f; c1[3] = i3 ? sx1 : 0.f;
    }
}

// D fragments of one wave: acc.v[nt][p] covers hidden unit 32 wave + 16 nt + 4 (lane >> 4) + r, point 16 p + (lane & 15)
struct Acc8 { f32x4 v[2][4]; };

__device__ __forceinline__ void acc_zero(Acc8 &c)
{
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int p = 0; p < 4; p++) c.v[nt][p] = (f32x4){0.f, 0.f, 0.f, 0.f};
}
// (s_setprio around the MFMA clusters, cdna_hip_programming.md T5: measured null in both directions, profiles/r02_experiments.md)
// one K32 step of the wave tile: weights (A) w[nt][hi|lo], activations (B) from the planes; hi.hi, hi.lo, lo.hi
__device__ __forceinline__ void k32_step(Acc8 &c, const uint4 (&w)[2][2], const uint4 *Xhi, const uint4 *Xlo, int kb_base, int lane)
{
    const int q = lane >> 4, j = lane & 15;
    h8 xh[4], xl[4];
#pragma unroll
    for (int p = 0; p < 4; p++) { xh[p] = as_h8(Xhi[(kb_base + q) * 64 + 16 * p + j]); xl[p] = as_h8(Xlo[(kb_base + q) * 64 + 16 * p + j]); }
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int p = 0; p < 4; p++) c.v[nt][p] = MFMAH(as_h8(w[nt][0]), xh[p], c.v[nt][p]);
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int p = 0; p < 4; p++) c.v[nt][p] = MFMAH(as_h8(w[nt][0]), xl[p], c.v[nt][p]);
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int p = 0; p < 4; p++) c.v[nt][p] = MFMAH(as_h8(w[nt][1]), xh[p], c.v[nt][p]);
}
// the same K32 step on a chunk's POINT-major planes (taps_store_feat<true>)
__device__ __forceinline__ void k32_step_chunk(Acc8 &c, const uint4 (&w)[2][2], const uint4 *Xhi, const uint4 *Xlo, int lane)
{
    const int q = lane >> 4, j = lane & 15;
    h8 xh[4], xl[4];
#pragma unroll
    for (int p = 0; p < 4; p++) { xh[p] = as_h8(Xhi[chunk_slot(16 * p + j, q)]); xl[p] = as_h8(Xlo[chunk_slot(16 * p + j, q)]); }
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int p = 0; p < 4; p++) c.v[nt][p] = MFMAH(as_h8(w[nt][0]), xh[p], c.v[nt][p]);
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int p = 0; p < 4; p++) c.v[nt][p] = MFMAH(as_h8(w[nt][0]), xl[p], c.v[nt][p]);
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int p = 0; p < 4; p++) c.v[nt][p] = MFMAH(as_h8(w[nt][1]), xh[p], c.v[nt][p]);
}
// HAZARD RULE of this file: an MFMA accumulator is never read FIRST by inline assembly.  hipcc's hazard recogniser inserts the wait states
// between a matrix instruction and the VALU read of its result only for instructions it knows; an asm statement that consumed c.v[..] directly
// read stale registers (measured: garbage outputs).  The first consumers below are builtins (v_cvt_pk_f16_f32 via the vector conversion,
// v_med3_f32); asm only ever sees their results.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, h2)); }      // v_cvt_pk_f16_f32 (RTN)
// ReLU + split of two pre-activations whose RAW fp16 conversion `hr` is already known: hi = max(hr, 0) (packed, = fp16(max(x, 0)): the conversion is
// monotone), lo = fp16(y - hi) with y = max(x, 0).  8 VALU instructions per pair with the range tracker (y >= 0: no |.|).
__device__ __forceinline__ void relu_split2(float x0, float x1, unsigned hr, unsigned &hi, unsigned &lo, float &rmax)
{
    // max(x, 0) as med3(x, 0, +inf): one instruction (fmaxf is canonicalised first: two per value)
    // max(x, 0) as a SIGNED-INTEGER max of the bit pattern (negative floats -- incl. -0 -- are negative integers; the order of the non-negative ones is the
    // integers'): one v_max_i32 per value.  hipcc turns fmaxf / med3(x, 0, +inf) into canonicalise + max (two instructions), and an inline-asm v_max_f32 may be
    // scheduled AHEAD of the builtin that is meant to read the accumulator first (HAZARD RULE below: garbage -- measured, round 5)
    const float y0 = __int_as_float(max(__float_as_int(x0), 0)), y1 = __int_as_float(max(__float_as_int(x1), 0));
    float r0, r1;
    asm("v_max3_f32 %0, %0, %1, %2" : "+v"(rmax) : "v"(y0), "v"(y1));
    asm("v_pk_max_f16 %0, %1, 0" : "=v"(hi) : "v"(hr));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(y0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(y1));
    lo = cvt_pk(r0, r1);
}
// sign bytes of the four values of a D fragment (raw conversions hr01, hr23) -> bits 8 r + 7 - f of the mask, f = fragment index 0..7: one byte
// permute, one rotate, one bit-field insert per FOUR values (the compare + add-with-carry chain this replaces cost two per value)
template <int F> __device__ __forceinline__ void sign_bits(unsigned &m, unsigned hr01, unsigned hr23)
{
    unsigned t;
    constexpr unsigned K = 0x80808080u >> F;
    asm("v_perm_b32 %0, %1, %2, %3" : "=v"(t) : "v"(hr23), "v"(hr01), "s"(0x07050301u));     // bytes {x0.hi, x1.hi, x2.hi, x3.hi}: sign in bit 7 of each
    if (F) asm("v_alignbit_b32 %0, %1, %1, %2" : "=v"(t) : "v"(t), "n"(F));                  // rotate right by F
    asm("v_bfi_b32 %0, %1, %2, %0" : "+v"(m) : "s"(K), "v"(t));                              // m = (K & t) | (~K & m)
}
#define MASK_BIT(f_, r_) (8 * (r_) + 7 - (f_))
// hidden-layer epilogue, forward: D fragments (bias preloaded, already in the next layer's operand units) -> ReLU -> packed split halves (the plane
// stores follow after the barrier that retires the planes' readers: planes_store); returns the ACTIVITY mask, bit MASK_BIT(f, r) set <=> the
// pre-activation's sign bit is clear
struct Packed8 { uint2 hi[8], lo[8]; };
__device__ __forceinline__ unsigned relu_pack(const Acc8 &c, Packed8 &pk, float &rmax)
{
    unsigned m = 0;
#define RELU_PACK_F(nt_, p_)                                                                                    \
    {                                                                                                           \
        const f32x4 x = c.v[nt_][p_];                                                                           \
        const unsigned h01 = cvt_pk(x[0], x[1]), h23 = cvt_pk(x[2], x[3]);                                      \
        sign_bits<(nt_) * 4 + (p_)>(m, h01, h23);                                                               \
        relu_split2(x[0], x[1], h01, pk.hi[(nt_) * 4 + (p_)].x, pk.lo[(nt_) * 4 + (p_)].x, rmax);               \
        relu_split2(x[2], x[3], h23, pk.hi[(nt_) * 4 + (p_)].y, pk.lo[(nt_) * 4 + (p_)].y, rmax);               \
    }
    RELU_PACK_F(0, 0) RELU_PACK_F(0, 1) RELU_PACK_F(0, 2) RELU_PACK_F(0, 3)
    RELU_PACK_F(1, 0) RELU_PACK_F(1, 1) RELU_PACK_F(1, 2) RELU_PACK_F(1, 3)
#undef RELU_PACK_F
    return ~m;
}
// packed halves of a wave's 32 hidden units x 64 points -> split planes [16 kb][64 pt][8 halves] (8-B units)
__device__ __forceinline__ void planes_store(const Packed8 &pk, uint2 *hi8, uint2 *lo8, int wave, int lane)
{
    const int q = lane >> 4, j = lane & 15;
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int idx = (((4 * wave + 2 * nt + (q >> 1)) * 64 + 16 * p + j) << 1) + (q & 1);
            hi8[idx] = pk.hi[nt * 4 + p]; lo8[idx] = pk.lo[nt * 4 + p];
        }
}
// split without the range tracker (backward chain: a non-finite gradient is loud by itself, nothing downstream rectifies it away)
__device__ __forceinline__ void split2_nr(float x0, float x1, unsigned &hi, unsigned &lo)
{
    float r0, r1;
    hi = cvt_pk(x0, x1);
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(x1));
    lo = cvt_pk(r0, r1);
}
// hidden-layer epilogue, backward: ReLU mask (one signed bit-field extract + and per value; the accumulators already carry the scale of the
// chain, HeadW::kback undoes it at the very end) and split; the stores follow with planes_store
__device__ __forceinline__ void mask_pack(const Acc8 &c, unsigned m, Packed8 &pk)
{
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int p = 0; p < 4; p++) {
            float y[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                unsigned keep;
                asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(keep) : "v"(m), "n"(MASK_BIT(nt * 4 + p, r)));
                y[r] = __uint_as_float(__float_as_uint(c.v[nt][p][r]) & keep);
            }
            split2_nr(y[0], y[1], pk.hi[nt * 4 + p].x, pk.lo[nt * 4 + p].x); split2_nr(y[2], y[3], pk.hi[nt * 4 + p].y, pk.lo[nt * 4 + p].y);
        }
}
// out[32 rows of this wave][64 pts] = M[128 x 128] (A, T-pack fragments from L2) x X[128 x 64 pts] (B, LDS planes), K = 128.
// The weights do not depend on the LDS contents: all 16 fragments are requested BEFORE the barrier that publishes X (wprefetch).
// (hidden-layer weight fragments as buffer loads: measured 0.5 % SLOWER than the saddr global form the compiler already finds here)
struct WPre { uint4 v[4][2][2]; float4 bias[2]; };
__device__ __forceinline__ void wprefetch(WPre &p, const uint4 *__restrict__ Wp, int wave, int lane, const float *__restrict__ bias = nullptr)
{
    // bias (operand units of the layer's OUTPUT, or NULL for the bias-free backward GEMMs): the accumulators start from it
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
        p.bias[nt] = bias ? *reinterpret_cast<const float4 *>(bias + 32 * wave + 16 * nt + 4 * (lane >> 4)) : make_float4(0.f, 0.f, 0.f, 0.f);
    // Wp: [4 K32 steps][4 waves][2 nt][hi|lo][64 lanes] uint4
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int hl = 0; hl < 2; hl++) p.v[s][nt][hl] = Wp[(unsigned)(wave * 256 + lane) + (unsigned)(s * 1024 + (nt * 2 + hl) * 64)];
}
// B fragments of one K32 step for HALF the points of the wave tile (p = 2 half + {0, 1}): 16 registers.  The GEMMs below run a K32 step as two half steps
// and request the fragments of the next K32 step's half as soon as the MFMAs of this step's half have been issued (same registers): the LDS round trip -- ~400
// clocks when the four waves of a workgroup leave a barrier together and all ask for 8 x 1 KB -- runs under the other half's 12 MFMAs instead of in front of all 24.
// Per accumulator the products still arrive as hi.hi, hi.lo, lo.hi of step 0, 1, 2, 3: the bits of k32_step.
struct BHalf { h8 xh[2], xl[2]; };
__device__ __forceinline__ void bhalf_load(BHalf &b, const uint4 *Xhi, const uint4 *Xlo, int kb_base, int half, int lane)
{
    const int q = lane >> 4, j = lane & 15;
#pragma unroll
    for (int pp = 0; pp < 2; pp++) { b.xh[pp] = as_h8(Xhi[(kb_base + q) * 64 + 16 * (2 * half + pp) + j]); b.xl[pp] = as_h8(Xlo[(kb_base + q) * 64 + 16 * (2 * half + pp) + j]); }
}
__device__ __forceinline__ void mfma12(Acc8 &c, const uint4 (&w)[2][2], const BHalf &b, int half)
{
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int pp = 0; pp < 2; pp++) c.v[nt][2 * half + pp] = MFMAH(as_h8(w[nt][0]), b.xh[pp], c.v[nt][2 * half + pp]);
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int pp = 0; pp < 2; pp++) c.v[nt][2 * half + pp] = MFMAH(as_h8(w[nt][0]), b.xl[pp], c.v[nt][2 * half + pp]);
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int pp = 0; pp < 2; pp++) c.v[nt][2 * half + pp] = MFMAH(as_h8(w[nt][1]), b.xh[pp], c.v[nt][2 * half + pp]);
}
// scheduling fence that pins matrix instructions and LDS reads to their side, everything else (VALU, SALU, vector memory, LDS writes) may cross
#define SB_PIN_MFMA_DSREAD() __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x10 | 0x20 | 0x40 | 0x200 | 0x400)
#ifndef GEMM_HALFSTEP
#define GEMM_HALFSTEP 1
#endif
__device__ __forceinline__ void gemm128(Acc8 &c, const uint4 *Xhi, const uint4 *Xlo, const WPre &p, int lane)
{
#if GEMM_HALFSTEP
    BHalf bA, bB;
    bhalf_load(bA, Xhi, Xlo, 0, 0, lane); bhalf_load(bB, Xhi, Xlo, 0, 1, lane);
#endif
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int pp = 0; pp < 4; pp++) c.v[nt][pp] = (f32x4){p.bias[nt].x, p.bias[nt].y, p.bias[nt].z, p.bias[nt].w};
#if GEMM_HALFSTEP
#pragma unroll
    for (int s = 0; s < 4; s++) {
        SB_PIN_MFMA_DSREAD(); mfma12(c, p.v[s], bA, 0); SB_PIN_MFMA_DSREAD();
        if (s < 3) bhalf_load(bA, Xhi, Xlo, 4 * (s + 1), 0, lane);
        SB_PIN_MFMA_DSREAD(); mfma12(c, p.v[s], bB, 1); SB_PIN_MFMA_DSREAD();
        if (s < 3) bhalf_load(bB, Xhi, Xlo, 4 * (s + 1), 1, lane);
    }
    SB_PIN_MFMA_DSREAD();
#else
#pragma unroll
    for (int s = 0; s < 4; s++) k32_step(c, p.v[s], Xhi, Xlo, 4 * s, lane);
#endif
}

// ---- two-head pipeline of the hidden layers (query_kernel<2, MODE_HUMAN>, HPIPE): a wave alone on its SIMD spends the hidden phase as GEMM (96 MFMAs, matrix
// pipe busy, VALU idle) -> epilogue (~200 VALU, matrix pipe idle) -> barrier -> plane stores -> barrier -> ...: a serial chain (52 k of the 145 k clocks of a
// workgroup alone on a CU, profiles/r05_phase_1wg_vs_2wg.txt).  The SIMD issues ~2 independent VALU instructions per v_mfma_f32_16x16x32_f16 for free
// (profiles/r05_coexec.txt), so the two heads leapfrog: while the GEMM of one head runs, the epilogue of the OTHER head's previous GEMM is issued between its MFMAs
// and its split halves go straight to that head's planes (every reader of them is behind the barrier that ended the previous slot): one barrier per GEMM instead
// of two, no packed halves held in registers, the weight fragments of the next GEMM reloaded K32 step by K32 step into the registers the current one just
// finished with.  Same MFMA sequence per accumulator and the same epilogue arithmetic as the head-by-head form: bit-identical results.
#ifndef HPIPE
#define HPIPE 1
#endif
enum { EPI_RELU = 1, EPI_MASK = 2 };
// epilogue of ONE D fragment F = 4 nt + p (four hidden units of one point) + store of its split halves into the planes.  EPI_RELU: `m` collects the raw
// sign bits (the caller inverts it once all eight fragments are through: relu_pack); EPI_MASK: `m` is the activity mask (mask_pack)
template <int EPI, int F>
__device__ __forceinline__ void epi_frag(const Acc8 &c, unsigned &m, uint2 *hi8, uint2 *lo8, int wave, int lane, float &rmax)
{
    constexpr int nt = F >> 2, p = F & 3;
    const int q = lane >> 4, j = lane & 15;
    const f32x4 x = c.v[nt][p];
    uint2 hi, lo;
    if (EPI == EPI_RELU) {
        const unsigned h01 = cvt_pk(x[0], x[1]), h23 = cvt_pk(x[2], x[3]);
        sign_bits<F>(m, h01, h23);
        relu_split2(x[0], x[1], h01, hi.x, lo.x, rmax);
        relu_split2(x[2], x[3], h23, hi.y, lo.y, rmax);
    } else {
        float y[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            unsigned keep;
            asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(keep) : "v"(m), "n"(MASK_BIT(F, r)));
            y[r] = __uint_as_float(__float_as_uint(x[r]) & keep);
        }
        split2_nr(y[0], y[1], hi.x, lo.x); split2_nr(y[2], y[3], hi.y, lo.y);
    }
    const int idx = (((4 * wave + 2 * nt + (q >> 1)) * 64 + 16 * p + j) << 1) + (q & 1);
    hi8[idx] = hi; lo8[idx] = lo;
}
template <int EPI>
__device__ __forceinline__ void epi_all(const Acc8 &c, unsigned &m, uint2 *hi8, uint2 *lo8, int wave, int lane, float &rmax)
{
    epi_frag<EPI, 0>(c, m, hi8, lo8, wave, lane, rmax); epi_frag<EPI, 1>(c, m, hi8, lo8, wave, lane, rmax);
    epi_frag<EPI, 2>(c, m, hi8, lo8, wave, lane, rmax); epi_frag<EPI, 3>(c, m, hi8, lo8, wave, lane, rmax);
    epi_frag<EPI, 4>(c, m, hi8, lo8, wave, lane, rmax); epi_frag<EPI, 5>(c, m, hi8, lo8, wave, lane, rmax);
    epi_frag<EPI, 6>(c, m, hi8, lo8, wave, lane, rmax); epi_frag<EPI, 7>(c, m, hi8, lo8, wave, lane, rmax);
}
// reload the fragments of K32 step s of a T-pack (the registers the GEMM in flight just finished with) with the next GEMM's
__device__ __forceinline__ void wreload(WPre &p, int s, const uint4 *__restrict__ Wp, int wave, int lane)
{
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int hl = 0; hl < 2; hl++) p.v[s][nt][hl] = Wp[(unsigned)(wave * 256 + lane) + (unsigned)(s * 1024 + (nt * 2 + hl) * 64)];
}
template <bool HAS>
__device__ __forceinline__ void bias_load(float4 (&b)[2], const float *__restrict__ bias, int wave, int lane)
{
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
        b[nt] = HAS ? *reinterpret_cast<const float4 *>(bias + 32 * wave + 16 * nt + 4 * (lane >> 4)) : make_float4(0.f, 0.f, 0.f, 0.f);
}
// cg = M x X (gemm128) with the epilogue of `ce` (another head's finished accumulators) issued between its MFMAs: one fragment per half step.  `nextW` (may be
// NULL): T-pack of the GEMM that follows in this wave's stream, `nextBias` its bias (NULL: zeros).  EPI = 0: no epilogue
template <int EPI, int F>
__device__ __forceinline__ void epi_opt(const Acc8 &ce, unsigned &m, uint2 *Ehi8, uint2 *Elo8, int wave, int lane, float &rmax)
{
    if (EPI != 0) epi_frag<(EPI ? EPI : EPI_RELU), F>(ce, m, Ehi8, Elo8, wave, lane, rmax);
}
// (NEXTW / NEXTB are compile-time: a run-time test of the pointers would put branches -- scheduling region borders -- between the half steps)
template <int EPI, bool NEXTW, bool NEXTB, bool KEEPBIAS = false>      // KEEPBIAS (with !NEXTW): the same weights AND bias serve the next GEMM (query128.h: the other half of a tile)
__device__ __forceinline__ void gemm128_epi(Acc8 &cg, const uint4 *Xhi, const uint4 *Xlo, WPre &w, const uint4 *__restrict__ nextW, const float *__restrict__ nextBias,
                                            const Acc8 &ce, unsigned &m, uint2 *Ehi8, uint2 *Elo8, int wave, int lane, float &rmax)
{
    BHalf bA, bB;
    bhalf_load(bA, Xhi, Xlo, 0, 0, lane); bhalf_load(bB, Xhi, Xlo, 0, 1, lane);
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int pp = 0; pp < 4; pp++) cg.v[nt][pp] = (f32x4){w.bias[nt].x, w.bias[nt].y, w.bias[nt].z, w.bias[nt].w};
    float4 nb[2];
    bias_load<NEXTB>(nb, nextBias, wave, lane);
#define GEMM_EPI_STEP(s_)                                                                                       \
    SB_PIN_MFMA_DSREAD();                                                                                       \
    mfma12(cg, w.v[s_], bA, 0);                                                                                 \
    SB_PIN_MFMA_DSREAD();                                                                                       \
    if ((s_) < 3) bhalf_load(bA, Xhi, Xlo, 4 * ((s_) + 1), 0, lane);                                            \
    epi_opt<EPI, 2 * (s_)>(ce, m, Ehi8, Elo8, wave, lane, rmax);                                                \
    SB_PIN_MFMA_DSREAD();                                                                                       \
    mfma12(cg, w.v[s_], bB, 1);                                                                                 \
    SB_PIN_MFMA_DSREAD();                                                                                       \
    if ((s_) < 3) bhalf_load(bB, Xhi, Xlo, 4 * ((s_) + 1), 1, lane);                                            \
    if (NEXTW) wreload(w, (s_), nextW, wave, lane);                                                             \
    epi_opt<EPI, 2 * (s_) + 1>(ce, m, Ehi8, Elo8, wave, lane, rmax);
    GEMM_EPI_STEP(0) GEMM_EPI_STEP(1) GEMM_EPI_STEP(2) GEMM_EPI_STEP(3)
#undef GEMM_EPI_STEP
    SB_PIN_MFMA_DSREAD();
    if (!KEEPBIAS) { w.bias[0] = nb[0]; w.bias[1] = nb[1]; }
}
// (a compile-time 1 MFMA : 2 VALU interleave of the layer-1 loops with sched_group_barrier: measured null, 2.033 vs 2.039 ms -- the ISA interleaves already and
//  those loops wait for their gathers: profiles/r02_experiments.md)
// per-phase shader-clock breakdown (debug builds with -DPHASE_CLK only; tools/bench_scripts/qphase.py)
#ifdef PHASE_CLK
__device__ unsigned long long g_phase[32];
#define PCLK(i_) do { if (tid == 0) { const unsigned long long t_ = clock64(); atomicAdd(&g_phase[i_], t_ - tprev_); tprev_ = t_; } } while (0)
#else
#define PCLK(i_)
#endif
template <int G, int MODE, bool USEP>
__global__ __launch_bounds__(256, (G == 1) ? 3 : 2) void query_kernel(const QArgs a)
{
    constexpr int C0 = USEP ? PROJ_C0 : 0;      // first chunk of the layer-1 loops
    // shared tap geometry (geom_compute / geom_fetch): -1.9 % on the two-head kernel; the one-head kernels (168 VGPRs, three workgroups per CU)
    // spill with it and lose 3 %: they keep the per-thread geometry
    constexpr bool SHGEO = (G == 2);
    VT_SKIP_RETURN(a.skip);
    const bool clk_probe = a.clk != nullptr && (blockIdx.x & 1023u) == 0u;       // (uniform: the two counters live in SGPRs)
    unsigned long long clk_c0 = 0, clk_r0 = 0;
    if (clk_probe) { clk_c0 = clock64(); clk_r0 = wall_clock64(); }
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    // Region 0 is time-shared: feature-chunk double buffer (layer 1) -> hidden-activation planes per head -> tap-difference
    // buffers + weight slab (layer-1 backward, after the d(hidden-1) fragments moved to registers).
    constexpr int R0 = (G * 2048 > 1152 + G * 1024) ? G * 2048 : 1152 + G * 1024;      // uint4 units
    uint4 *Hp = lds;                        // G x {hi [16 kb][64], lo [16 kb][64]}
    uint4 *Go = lds + R0;                   // {hi [2 kb][64], lo [2 kb][64]}  normalised output gradient
    float *sPt = reinterpret_cast<float *>(Go + 256);   // [64][3]
    float *sUV = sPt + 64 * 3;              // [4][64][2]
    float *sInv = sUV + 4 * 64 * 2;         // [G][64] inverse of the per-point gradient normalisation
    float *sDf = sInv + G * 64;             // [64] clamped distance of the point (MODE_PROJECT)
    int *sIn = reinterpret_cast<int *>(sDf + 64);   // [64]
    double *sRed = reinterpret_cast<double *>(sIn + 64);     // [8]
    int *sOvf = reinterpret_cast<int *>(sRed + 8);           // [1] some split operand of this workgroup left the fp16 range
    float *sGeo = reinterpret_cast<float *>(sOvf + 4);       // [2 slots][64 points][GEO_STRIDE] tap geometry of a map, shared by the workgroup
    float rmax = 0.f;

    // (the wave index through readfirstlane: a branch on `wave` is then a SCALAR branch and what is computed behind it from uniform inputs -- map indices,
    //  table addresses, resolutions -- stays in SGPRs instead of being carried, and spilled, in VGPRs: round 6, profiles/r06_kernel_resources.txt)
    const int tid_raw = threadIdx.x, tid = tid_raw, wave = (G == 2) ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6), lane = tid & 63, q = lane >> 4, j = lane & 15;
    // XCD-aware block -> (frame, tile) map: the dispatcher places workgroup L on XCD L % 8 (speed only, never correctness), so
    // give every XCD whole frames: all ~108 tiles of a frame then gather from the same few MB of maps through ONE L2.
    int b, tile;
    {
        const int tiles = (a.N + 63) >> 6, L = blockIdx.x;
        if ((a.B & 7) == 0) { const int slot = L >> 3; b = (L & 7) + 8 * (slot / tiles); tile = slot % tiles; }
        else { b = L / tiles; tile = L % tiles; }
        // the integer division runs on the VALU: without this the frame index -- and every base address and buffer descriptor derived from
        // it -- lives in VGPRs and each use pays v_readfirstlane
        b = __builtin_amdgcn_readfirstlane(b); tile = __builtin_amdgcn_readfirstlane(tile);
    }
    const int n0 = tile * 64;
#ifdef PHASE_CLK
    unsigned long long tprev_ = clock64();
#endif

    // ---- per-point projections (camera.py:52-90, chore_triplane.py:207-251)
    if (tid == 0) *sOvf = 0;
    if (tid < 64) {
        const int n = min(n0 + tid, a.N - 1);
        const int pn = a.order ? a.order[n] : n;        // a locality-preserving order (FitContext: Morton order of the template) keeps the gathers of a tile in few texels
        const float *p = a.pts + ((size_t)b * a.N + pn) * 3;
        const float x = p[0], y = p[1], z = p[2];
        float px = a.fx * x / z + a.cx, py = a.fy * y / z + a.cy;
        px = a.crop / 2 + px - a.crop_center[2 * b]; py = a.crop / 2 + py - a.crop_center[2 * b + 1];
        const float nx = 2 * px / a.crop - 1, ny = 2 * py / a.crop - 1;
        sIn[tid] = (pn << 1) | (int)((nx >= -1.0f) && (nx <= 1.0f) && (ny >= -1.0f) && (ny <= 1.0f));     // point index | in-image flag
        // a non-finite coordinate makes every feature of the point NaN; the ReLUs of the hidden epilogues are integer / packed-f16 maxima that send a NaN with
        // the sign bit set to 0 and the range tracker (v_max3_f32) ignores NaNs, so such a point could come out FINITE and wrong: the whole tile is
        // made loud instead (NaN outputs through *sOvf, like an operand beyond the split range).  Maps are finite by the caller's contract (the encoders' output).
        if (!(fabsf(x) < INFINITY && fabsf(y) < INFINITY && fabsf(z) < INFINITY)) *sOvf = 1;
        const float c0 = x - a.body_center[3 * b], c1 = y - a.body_center[3 * b + 1], c2 = z - a.body_center[3 * b + 2];
        sPt[tid * 3] = x; sPt[tid * 3 + 1] = y; sPt[tid * 3 + 2] = z;
        sUV[(0 * 64 + tid) * 2] = nx;  sUV[(0 * 64 + tid) * 2 + 1] = ny;   // perspective
        sUV[(1 * 64 + tid) * 2] = c2;  sUV[(1 * 64 + tid) * 2 + 1] = c1;   // right
        sUV[(2 * 64 + tid) * 2] = -c0; sUV[(2 * 64 + tid) * 2 + 1] = c1;   // back
        sUV[(3 * 64 + tid) * 2] = c0;  sUV[(3 * 64 + tid) * 2 + 1] = -c2;  // top
    }
    __syncthreads();

    // ---- layer 1: stream the 19 chunks (all heads of the group at once); the tap loads of chunk i+1 are in flight
    //      while the MFMAs of chunk i run (one barrier per chunk thanks to the double buffer); the weight fragments of
    //      chunk i+1 are requested as soon as the MFMAs of chunk i have been issued.
    Acc8 acc1[G];
#pragma unroll
    for (int g = 0; g < G; g++) acc_zero(acc1[g]);
    if (USEP) {
        // The im_feat part of the layer-1 pre-activations: blend of the 4 tap rows of P (fp32).  Gathered with the four lanes of a point ADJACENT
        // (thread = (point, 16-byte column group): 4 lanes read 64 contiguous bytes of a row per instruction) -- in the D-fragment layout the
        // lanes of one instruction sit on 64 different rows (16 points x 4 groups 16 lanes apart) and the texture addresser takes twice as long
        // for the same bytes (tools/bench_scripts/gather_patterns.hip: 9.4 vs 19.7 TB/s; -8 % on this kernel).  The blended rows go through
        // LDS ([point][column] fp32, row stride = columns + 4: both sides near conflict-free) into the accumulator fragments: lane (q, j) holds
        // hidden units 32 wave + 16 nt + 4 q .. +3 of the points 16 p + j.
        constexpr int PS = G * 128 + 4;                         // floats per staged row (G = 2: runs 1 KB into the Go buffer, unused until layer 4)
        float *stage = reinterpret_cast<float *>(lds);
        static_assert(64 * PS * 4 <= 16 * (R0 + 256), "the staged rows must fit region 0 + Go");
        const int R = a.res[0];
        const rsrc_t Pb = make_rsrc(a.proj + (size_t)b * R * R * a.pw, (unsigned)(R * R * a.pw) * 4u);
        {
            const int spt = tid >> 2, seg = tid & 3;
            unsigned o[4]; float w[4], unused[4];
            proj_geom(sUV, spt, R, a.pw, o, w, unused, false);
#pragma unroll
            for (int k = 0; k < 4; k++) o[k] = (o[k] + 4u * seg) * 4u;         // byte offset of this thread's first column group in tap row k
#pragma unroll
            for (int g = 0; g < G; g++) {
                // 32 independent 16-byte loads in flight before the first blend (the phase is latency-bound)
                float4 t[8][4];
                const unsigned pc = (unsigned)a.hw[g].pcol * 4u;
#pragma unroll
                for (int i = 0; i < 8; i++)
#pragma unroll
                    for (int k = 0; k < 4; k++) t[i][k] = GATHER_P4(Pb, o[k] + 64u * i, pc);
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float4 nw = t[i][0], ne = t[i][1], sw = t[i][2], se = t[i][3];
                    *reinterpret_cast<float4 *>(stage + spt * PS + g * 128 + 16 * i + 4 * seg) = make_float4(TAPSUM_(x, w), TAPSUM_(y, w), TAPSUM_(z, w), TAPSUM_(w, w));
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const float4 v = *reinterpret_cast<const float4 *>(stage + (16 * p + j) * PS + g * 128 + 32 * wave + 16 * nt + 4 * q);
                    acc1[g].v[nt][p] = (f32x4){v.x, v.y, v.z, v.w};
                }
        __syncthreads();                        // the chunk double buffer takes region 0 over
    }
    PCLK(0);
    Taps tp;
    TapGeom<1> tg;
    {   // geometry of the first two maps of the loop (waves 0 and 1), then every thread fetches its entries of the first
        int mi, co; chunk_info(C0, mi, co);
        if (SHGEO) {
            if (wave == 0) geom_compute<1>(a, mi, sUV, lane, sGeo);
            if (wave == 1 && mi + 1 < 8) geom_compute<1>(a, mi + 1, sUV, lane, sGeo);
            __syncthreads();
            geom_fetch<1>(mi, sGeo, tid, tg);
        } else
            taps_geom(a, mi, sUV, tid, tg);
        taps_issue(a, b, mi, co, tg, tp);
    }
    uint4 wf[G][2][2];
    const unsigned wvo = (unsigned)(wave * 256 + lane);       // per-lane part of a T-pack fragment index; the rest is uniform / immediate
#define LOAD_W1(step_)                                                                                                       \
    _Pragma("unroll") for (int g = 0; g < G; g++) {                                                                          \
        const rsrc_t wp_ = make_rsrc(a.hw[g].w1p, 0x40000000u);                                                              \
        _Pragma("unroll") for (int nt = 0; nt < 2; nt++)                                                                     \
            _Pragma("unroll") for (int hl = 0; hl < 2; hl++)                                                                 \
                wf[g][nt][hl] = bload_u4(wp_, wvo * 16u + (unsigned)((nt * 2 + hl) * 1024), (unsigned)(step_) * 16384u);     \
    }
    LOAD_W1(C0)
    // software pipeline: the features of chunk ci+1 are blended / split / stored (VALU + LDS stores) in the same barrier interval
    // as the MFMAs of chunk ci, so the two interleave; the taps of chunk ci+2 are requested as soon as the tap registers are free
    taps_store_feat<CHUNK_PM != 0>(tp, tg, reinterpret_cast<uint2 *>(lds + (C0 & 1) * 512), reinterpret_cast<uint2 *>(lds + (C0 & 1) * 512 + 256), tid, rmax);
    { int mi, co; chunk_info(C0 + 1, mi, co); taps_issue(a, b, mi, co, tg, tp); }      // the first map of the loop (im_feat or tmpx) has >= 2 chunks: same geometry
    for (int ci = C0; ci < NCHUNK; ci++) {
        uint4 *buf = lds + (ci & 1) * 512, *nbuf = lds + ((ci + 1) & 1) * 512;      // {hi [4 kb][64], lo [4 kb][64]}
        __syncthreads();                        // chunk ci visible; the other buffer's readers (MFMAs of chunk ci-1) are done
#pragma unroll
#if CHUNK_PM
        for (int g = 0; g < G; g++) k32_step_chunk(acc1[g], wf[g], buf, buf + 256, lane);
#else
        for (int g = 0; g < G; g++) k32_step(acc1[g], wf[g], buf, buf + 256, 0, lane);
#endif
        // vmcnt retires loads IN ORDER: the weight fragments of the next chunk are requested BEFORE the taps of chunk ci+2, so that
        // waiting for them (top of the next iteration) does not also wait for the far slower gather
        if (ci + 1 < NCHUNK) taps_store_feat<CHUNK_PM != 0>(tp, tg, reinterpret_cast<uint2 *>(nbuf), reinterpret_cast<uint2 *>(nbuf + 256), tid, rmax);
        LOAD_W1(ci + 1)
        if (ci + 2 < NCHUNK) {
            int mi, co; chunk_info(ci + 2, mi, co);
            if (co == 0) {
                // a new map (uniform branch): its ring slot was written an iteration or more ago (barrier at the top of the loop); the other slot
                // was last read when the previous map started, so one wave may now fill it with the map after this one
                if (SHGEO) {
                    geom_fetch<1>(mi, sGeo, tid, tg);
                    if (mi + 1 < 8 && wave == (mi & 3)) geom_compute<1>(a, mi + 1, sUV, lane, sGeo);
                } else
                    taps_geom(a, mi, sUV, tid, tg);
            }
            taps_issue(a, b, mi, co, tg, tp);
        }
    }
    {   // z_feat = (x, y, z - 2.2): internal channels 608..610 (K32 step 19, k = 8 q + t: only q == 0, t < 3 are non-zero)
        uint4 xh[4], xl[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            uint2 hi = make_uint2(0u, 0u), lo = make_uint2(0u, 0u);
            if (q == 0) {
                const float *pp = sPt + (16 * p + j) * 3;
                split4(pp[0] * a.u1, pp[1] * a.u1, (pp[2] - 2.2f) * a.u1, a.u1, hi, lo, rmax);       // 4th channel = the constant one (weight s_1 b_1)
            }
            xh[p] = make_uint4(hi.x, hi.y, 0u, 0u); xl[p] = make_uint4(lo.x, lo.y, 0u, 0u);
        }
#pragma unroll
        for (int g = 0; g < G; g++) {
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    acc1[g].v[nt][p] = MFMAH(as_h8(wf[g][nt][0]), as_h8(xh[p]), acc1[g].v[nt][p]);
                    acc1[g].v[nt][p] = MFMAH(as_h8(wf[g][nt][0]), as_h8(xl[p]), acc1[g].v[nt][p]);
                    acc1[g].v[nt][p] = MFMAH(as_h8(wf[g][nt][1]), as_h8(xh[p]), acc1[g].v[nt][p]);
                }
        }
    }
#undef LOAD_W1
    PCLK(1);
    __syncthreads();        // region 0 changes role: chunk buffers -> hidden-activation planes
    constexpr bool PIPE2 = (G == 2) && (MODE == MODE_HUMAN) && HPIPE;
    constexpr int GH = PIPE2 ? 0 : G;        // the head-by-head form below runs for every other instantiation
    // hidden-1 activations of ALL heads go to their planes right away: no head's layer-1 accumulators stay live in registers
    // while another head runs its layers 2..4 and backward.  The layer-1 bias came in through the constant-one channel of the xyz step.
    unsigned m1s[G];
#pragma unroll
    for (int g = 0; g < GH; g++) {
        Packed8 pk;
        m1s[g] = relu_pack(acc1[g], pk, rmax);
        planes_store(pk, reinterpret_cast<uint2 *>(Hp + g * 2048), reinterpret_cast<uint2 *>(Hp + g * 2048 + 1024), wave, lane); OVF_PUBLISH();
    }

    // ---- per head: layers 2..4, objective / upstream gradient, backward to d(hidden-1)
    double loss_acc[2] = {0.0, 0.0};
    if constexpr (PIPE2) {
        // ---- layer 4 (points as rows: wave w owns the 16 points of tile w, columns = up to 16 outputs, zero padded), the objective / upstream gradient of head g and
        //      its per-point normalisation into the gradient buffer Gg ({hi [2 kb][64], lo [2 kb][64]}); returns false for MODE_FWD (nothing to back-propagate)
        // (the global operands of the objective -- layer-4 weight fragments, bias, part labels of the wave's points -- are requested by obj_prefetch, which the
        //  two-head pipeline issues a slot ahead: inside head_objective their latency would sit in front of a serial chain)
        struct ObjPre { uint4 w4[8]; float bias4; int lab[4]; };
        auto obj_prefetch = [&](const int g, ObjPre &o) {
            const HeadW &hw = a.hw[g];
    #pragma unroll
            for (int i = 0; i < 8; i++) o.w4[i] = hw.w4p[i * 64 + lane];
            o.bias4 = hw.b4[j];
    #pragma unroll
            for (int r = 0; r < 4; r++) o.lab[r] = (MODE == MODE_HUMAN && hw.id != 0) ? a.labels[sIn[wave * 16 + q * 4 + r] >> 1] : 0;
        };
        auto head_objective = [&](const int g, const uint4 *Hhi, const uint4 *Hlo, uint4 *Gg, const ObjPre &op, const bool pre) {       // !pre: operands loaded in place
            const HeadW &hw = a.hw[g];
            f32x4 o4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    #pragma unroll
            for (int s = 0; s < 4; s++) {
                const h8 xh = as_h8(Hhi[(4 * s + q) * 64 + 16 * wave + j]), xl = as_h8(Hlo[(4 * s + q) * 64 + 16 * wave + j]);
                const h8 wh = as_h8(pre ? op.w4[s * 2 + 0] : hw.w4p[(s * 2 + 0) * 64 + lane]), wl = as_h8(pre ? op.w4[s * 2 + 1] : hw.w4p[(s * 2 + 1) * 64 + lane]);
                o4 = MFMAH(xh, wh, o4); o4 = MFMAH(xl, wh, o4); o4 = MFMAH(xh, wl, o4);
            }
            const float bias4 = pre ? op.bias4 : hw.b4[j];
            float go[4];    // upstream gradient of output j at points wave*16 + q*4 + r
    #pragma unroll
            for (int r = 0; r < 4; r++) {
                const int pt = wave * 16 + q * 4 + r, n = n0 + pt;
                const bool valid = n < a.N, live = j < hw.kout;
                const bool inimg = (sIn[pt] & 1) != 0;
                const int pn = sIn[pt] >> 1;
                float val = o4[r] * hw.cout + bias4;
                if (*sOvf) val = __builtin_nanf("");        // an operand left the split range: no silent finite garbage
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
                        const float mx = row16_max(live ? val : -INFINITY);
                        const float e = live ? expf(val - mx) : 0.f;
                        const float se = row16_sum(e);
                        const int lab = pre ? op.lab[r] : a.labels[pn];
                        if (valid && live) {
                            go[r] = (e / se - (j == lab ? 1.f : 0.f)) * a.w1 / (float)a.B;
                            if (j == lab) loss_acc[1] += (double)(logf(se) - (val - mx));
                        }
                    }
                } else if (MODE == MODE_PROJECT) {
                    // Generator.approx_surface (recon/gen/generator.py:72-103): target = clamp(df[:, idx], max = threshold), the step
                    // differentiates sum(target): gradient 1 where the prediction is below the threshold and the point is in the image
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
            if (MODE == MODE_FWD) return;
            // ---- normalise the upstream gradient per point (the backward chain is linear in it): go' = go * 2^e with
            //      max_o |go'| in [2^GO_EXP, 2^(GO_EXP+1)); the inverse is applied to d(features) in the layer-1 backward
    #pragma unroll
            for (int r = 0; r < 4; r++) {
                const float m = row16_max(fabsf(go[r]));
                const int eb = (int)((__float_as_uint(m) >> 23) & 255u);
                const int ge = hw.goexp;                            // go' in [2^ge, 2^(ge+1)): the weight scales s_4 s_3 s_2 of the chain are taken out in advance
                const bool ok = eb >= ge + 2 && eb >= 2 && eb < 255 && eb - ge < 254;      // zero / denormal-sized / non-finite gradients pass unscaled
                const float s = ok ? __uint_as_float((unsigned)(254 + ge - eb) << 23) : 1.0f;
                const float inv = ok ? __uint_as_float((unsigned)(eb - ge) << 23) : 1.0f;
                const int pt = wave * 16 + q * 4 + r;
                if (j == 0) sInv[g * 64 + pt] = inv;
                const float x = go[r] * s;
                const _Float16 hi = (_Float16)x, lo = (_Float16)(x - (float)hi);
                _Float16 *gh = reinterpret_cast<_Float16 *>(Gg), *gl = reinterpret_cast<_Float16 *>(Gg + 128);
                gh[((j >> 3) * 64 + pt) * 8 + (j & 7)] = hi; gl[((j >> 3) * 64 + pt) * 8 + (j & 7)] = lo;
            }
        };
        // ---- backward through layer 4: g3[n][pt] = W4[o][n] . go'[o][pt]   (K = 16 outputs, zero padded to one K32 step); w = the four w4tp fragments of the wave
        auto head_g4t = [&](Acc8 &c, const uint4 (&w)[2][2], const uint4 *Gg) {
            acc_zero(c);
            h8 xh[4], xl[4];
    #pragma unroll
            for (int p = 0; p < 4; p++) {
                // K = 16 live rows of the K32 step: lane groups q >= 2 hold zeros (loaded from a valid address and cleared: a select between the buffer and a
                // zero constant makes hipcc pick between two ADDRESSES here -- a generic load through a scratch copy of the constant)
                uint4 vh = Gg[(q & 1) * 64 + 16 * p + j], vl = Gg[128 + (q & 1) * 64 + 16 * p + j];
                if (q >= 2) { vh = make_uint4(0u, 0u, 0u, 0u); vl = make_uint4(0u, 0u, 0u, 0u); }
                xh[p] = as_h8(vh); xl[p] = as_h8(vl);
            }
    #pragma unroll
            for (int nt = 0; nt < 2; nt++)
    #pragma unroll
                for (int p = 0; p < 4; p++) {
                    c.v[nt][p] = MFMAH(as_h8(w[nt][0]), xh[p], c.v[nt][p]);
                    c.v[nt][p] = MFMAH(as_h8(w[nt][0]), xl[p], c.v[nt][p]);
                    c.v[nt][p] = MFMAH(as_h8(w[nt][1]), xh[p], c.v[nt][p]);
                }
        };
        // ---- the two heads leapfrog through layers 2..4 and back (see gemm128_epi): slot k = { GEMM of one head | epilogue of the other head's previous GEMM,
        //      stored straight into that head's planes } + ONE barrier.  P0 / P1 = the planes of head 0 / 1, in place as before.
        uint4 *P0 = Hp, *P1 = Hp + 2048;
        uint2 *P0h = reinterpret_cast<uint2 *>(P0), *P0l = reinterpret_cast<uint2 *>(P0 + 1024), *P1h = reinterpret_cast<uint2 *>(P1), *P1l = reinterpret_cast<uint2 *>(P1 + 1024);
        uint4 *Go0 = Go, *Go1 = reinterpret_cast<uint4 *>(sGeo);      // the tap-geometry ring (6 KB) is idle between the two layer-1 loops: second gradient buffer (4 KB)
        const HeadW &h0 = a.hw[0], &h1 = a.hw[1];
        WPre w;
        Acc8 c0, c1;
        unsigned m1_0 = 0, m1_1 = 0, m2_0 = 0, m2_1 = 0, m3_0 = 0, m3_1 = 0;
        wprefetch(w, h0.w2p, wave, lane, h0.b2);
        epi_all<EPI_RELU>(acc1[0], m1_0, P0h, P0l, wave, lane, rmax); m1_0 = ~m1_0; OVF_PUBLISH();
        __syncthreads();                                                                                        // H1[0] visible
        PCLK(8);
        gemm128_epi<EPI_RELU, true, true>(c0, P0, P0 + 1024, w, h1.w2p, h1.b2, acc1[1], m1_1, P1h, P1l, wave, lane, rmax); m1_1 = ~m1_1; OVF_PUBLISH();      // L2[0] | E1[1]
        __syncthreads();                                                                                        // H1[1] visible; P0 read
        PCLK(9);
        gemm128_epi<EPI_RELU, true, true>(c1, P1, P1 + 1024, w, h0.w3p, h0.b3, c0, m2_0, P0h, P0l, wave, lane, rmax); m2_0 = ~m2_0; OVF_PUBLISH();           // L2[1] | E2[0]
        __syncthreads();                                                                                        // H2[0] visible; P1 read
        PCLK(10);
        gemm128_epi<EPI_RELU, true, true>(c0, P0, P0 + 1024, w, h1.w3p, h1.b3, c1, m2_1, P1h, P1l, wave, lane, rmax); m2_1 = ~m2_1; OVF_PUBLISH();           // L3[0] | E2[1]
        __syncthreads();                                                                                        // H2[1] visible; P0 read
        PCLK(11);
        gemm128_epi<EPI_RELU, true, false>(c1, P1, P1 + 1024, w, h0.w3tp, nullptr, c0, m3_0, P0h, P0l, wave, lane, rmax); m3_0 = ~m3_0; OVF_PUBLISH();       // L3[1] | E3[0]
        __syncthreads();                                                                                        // H3[0] visible; P1 read
        PCLK(12);
        uint4 w4a[2][2], w4b[2][2];
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int hl = 0; hl < 2; hl++) { w4a[nt][hl] = h0.w4tp[(((size_t)wave * 2 + nt) * 2 + hl) * 64 + lane]; w4b[nt][hl] = h1.w4tp[(((size_t)wave * 2 + nt) * 2 + hl) * 64 + lane]; }
        ObjPre op0, op1;
        obj_prefetch(0, op0); obj_prefetch(1, op1);
        epi_all<EPI_RELU>(c1, m3_1, P1h, P1l, wave, lane, rmax); m3_1 = ~m3_1; OVF_PUBLISH();                   // E3[1]
        head_objective(0, P0, P0 + 1024, Go0, op0, true);                                                             // L4[0], objective
        __syncthreads();                                                                                        // H3[1], Go0 visible; P0 read
        PCLK(13);
        head_g4t(c0, w4a, Go0);                                                                                 // L4^T[0]
        head_objective(1, P1, P1 + 1024, Go1, op1, true);                                                             // L4[1], objective
        epi_all<EPI_MASK>(c0, m3_0, P0h, P0l, wave, lane, rmax);                                                // g3[0] -> P0
        __syncthreads();                                                                                        // g3[0], Go1 visible; P1 read
        PCLK(14);
        head_g4t(c1, w4b, Go1);                                                                                 // L4^T[1]
        gemm128_epi<EPI_MASK, true, false>(c0, P0, P0 + 1024, w, h1.w3tp, nullptr, c1, m3_1, P1h, P1l, wave, lane, rmax);       // W3^T[0] | g3[1] -> P1
        __syncthreads();                                                                                        // g3[1] visible; P0 read
        PCLK(15);
        gemm128_epi<EPI_MASK, true, false>(c1, P1, P1 + 1024, w, h0.w2tp, nullptr, c0, m2_0, P0h, P0l, wave, lane, rmax);       // W3^T[1] | g2[0] -> P0
        __syncthreads();                                                                                        // g2[0] visible; P1 read
        PCLK(16);
        gemm128_epi<EPI_MASK, true, false>(c0, P0, P0 + 1024, w, h1.w2tp, nullptr, c1, m2_1, P1h, P1l, wave, lane, rmax);       // W2^T[0] | g2[1] -> P1
        __syncthreads();                                                                                        // g2[1] visible; P0 read
        PCLK(17);
        gemm128_epi<EPI_MASK, false, false>(c1, P1, P1 + 1024, w, nullptr, nullptr, c0, m1_0, P0h, P0l, wave, lane, rmax);       // W2^T[1] | d(hidden-1)[0] -> P0
        __syncthreads();                                                                                        // P1 read
        PCLK(18);
        epi_all<EPI_MASK>(c1, m1_1, P1h, P1l, wave, lane, rmax);                                                // d(hidden-1)[1] -> P1
        __syncthreads();
        PCLK(19);
    }
#pragma unroll
    for (int g = 0; g < GH; g++) {
        const HeadW &hw = a.hw[g];
        uint4 *Hhi = Hp + g * 2048, *Hlo = Hhi + 1024;
        uint2 *Hhi8 = reinterpret_cast<uint2 *>(Hhi), *Hlo8 = reinterpret_cast<uint2 *>(Hlo);
        Acc8 c;
        WPre wp;
        Packed8 pk;
        wprefetch(wp, hw.w2p, wave, lane, hw.b2);
        const unsigned m1 = m1s[g];
        if (g == 0) __syncthreads();           // hidden-1 planes of all heads visible
        gemm128(c, Hhi, Hlo, wp, lane);
        wprefetch(wp, hw.w3p, wave, lane, hw.b3);
        const unsigned m2 = relu_pack(c, pk, rmax);
        __syncthreads();
        planes_store(pk, Hhi8, Hlo8, wave, lane); OVF_PUBLISH();
        __syncthreads();
        gemm128(c, Hhi, Hlo, wp, lane);
        const unsigned m3 = relu_pack(c, pk, rmax);
        __syncthreads();
        planes_store(pk, Hhi8, Hlo8, wave, lane); OVF_PUBLISH();
        __syncthreads();
        // layer 4 (points as rows): wave w owns the 16 points of tile w, columns = up to 16 outputs (zero padded)
        f32x4 o4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const h8 xh = as_h8(Hhi[(4 * s + q) * 64 + 16 * wave + j]), xl = as_h8(Hlo[(4 * s + q) * 64 + 16 * wave + j]);
            const h8 wh = as_h8(hw.w4p[(s * 2 + 0) * 64 + lane]), wl = as_h8(hw.w4p[(s * 2 + 1) * 64 + lane]);
            o4 = MFMAH(xh, wh, o4); o4 = MFMAH(xl, wh, o4); o4 = MFMAH(xh, wl, o4);
        }
        const float bias4 = hw.b4[j];
        float go[4];    // upstream gradient of output j at points wave*16 + q*4 + r
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int pt = wave * 16 + q * 4 + r, n = n0 + pt;
            const bool valid = n < a.N, live = j < hw.kout;
            const bool inimg = (sIn[pt] & 1) != 0;
            const int pn = sIn[pt] >> 1;
            float val = o4[r] * hw.cout + bias4;
            if (*sOvf) val = __builtin_nanf("");        // an operand left the split range: no silent finite garbage
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
                    const float mx = row16_max(live ? val : -INFINITY);
                    const float e = live ? expf(val - mx) : 0.f;
                    const float se = row16_sum(e);
                    const int lab = a.labels[pn];
                    if (valid && live) {
                        go[r] = (e / se - (j == lab ? 1.f : 0.f)) * a.w1 / (float)a.B;
                        if (j == lab) loss_acc[1] += (double)(logf(se) - (val - mx));
                    }
                }
            } else if (MODE == MODE_PROJECT) {
                // Generator.approx_surface (recon/gen/generator.py:72-103): target = clamp(df[:, idx], max = threshold), the step
                // differentiates sum(target): gradient 1 where the prediction is below the threshold and the point is in the image
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
        // ---- normalise the upstream gradient per point (the backward chain is linear in it): go' = go * 2^e with
        //      max_o |go'| in [2^GO_EXP, 2^(GO_EXP+1)); the inverse is applied to d(features) in the layer-1 backward
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float m = row16_max(fabsf(go[r]));
            const int eb = (int)((__float_as_uint(m) >> 23) & 255u);
            const int ge = hw.goexp;                            // go' in [2^ge, 2^(ge+1)): the weight scales s_4 s_3 s_2 of the chain are taken out in advance
            const bool ok = eb >= ge + 2 && eb >= 2 && eb < 255 && eb - ge < 254;      // zero / denormal-sized / non-finite gradients pass unscaled
            const float s = ok ? __uint_as_float((unsigned)(254 + ge - eb) << 23) : 1.0f;
            const float inv = ok ? __uint_as_float((unsigned)(eb - ge) << 23) : 1.0f;
            const int pt = wave * 16 + q * 4 + r;
            if (j == 0) sInv[g * 64 + pt] = inv;
            const float x = go[r] * s;
            const _Float16 hi = (_Float16)x, lo = (_Float16)(x - (float)hi);
            _Float16 *gh = reinterpret_cast<_Float16 *>(Go), *gl = reinterpret_cast<_Float16 *>(Go + 128);
            gh[((j >> 3) * 64 + pt) * 8 + (j & 7)] = hi; gl[((j >> 3) * 64 + pt) * 8 + (j & 7)] = lo;
        }
        // ---- backward through layer 4: g3[n][pt] = W4[o][n] . go'[o][pt]   (K = 16 outputs, zero padded to one K32 step)
        WPre wq;
        {
            uint4 w[2][2];
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int hl = 0; hl < 2; hl++) w[nt][hl] = hw.w4tp[(((size_t)wave * 2 + nt) * 2 + hl) * 64 + lane];
            wprefetch(wq, hw.w3tp, wave, lane);
            __syncthreads();                       // Go visible; H (h3) no longer read
            acc_zero(c);
            h8 xh[4], xl[4];
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const uint4 z = make_uint4(0u, 0u, 0u, 0u);
                xh[p] = as_h8(q < 2 ? Go[q * 64 + 16 * p + j] : z); xl[p] = as_h8(q < 2 ? Go[128 + q * 64 + 16 * p + j] : z);
            }
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    c.v[nt][p] = MFMAH(as_h8(w[nt][0]), xh[p], c.v[nt][p]);
                    c.v[nt][p] = MFMAH(as_h8(w[nt][0]), xl[p], c.v[nt][p]);
                    c.v[nt][p] = MFMAH(as_h8(w[nt][1]), xh[p], c.v[nt][p]);
                }
        }
        mask_pack(c, m3, pk);
        planes_store(pk, Hhi8, Hlo8, wave, lane);
        __syncthreads();
        gemm128(c, Hhi, Hlo, wq, lane);            // g2 = W3^T . g3
        wprefetch(wq, hw.w2tp, wave, lane);
        mask_pack(c, m2, pk);
        __syncthreads();
        planes_store(pk, Hhi8, Hlo8, wave, lane);
        __syncthreads();
        gemm128(c, Hhi, Hlo, wq, lane);            // g1 = W2^T . g2
        mask_pack(c, m1, pk);
        __syncthreads();
        planes_store(pk, Hhi8, Hlo8, wave, lane);   // the planes now hold d loss' / d (pre-activation 1) of this head (times the chain's scale)
        __syncthreads();
    }

    PCLK(2);
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
            if (*sOvf) s = (double)__builtin_nanf("");
            if (MODE == MODE_HUMAN || tid == 0) atomicAdd(a.terms + tid, s);
        }
    }
    if (MODE == MODE_FWD) return;

    // ---- backward through layer 1 and the gathers: wave w owns the 16 points of tile w (one point per lane column j);
    //      d feat[c][pt] = sum_g inv[g][pt] / s_W1[g] * ( W1[g][u][c] . dh1'[g][u][pt] ),  c = the chunk's 32 channels
    // (the lane-derived indices of this half of the kernel are re-derived from a laundered thread id -- see launder_v: the forward half's copies and every LDS
    //  address built on them would otherwise stay live across the hidden layers, where the register file is full)
    const int tid_l1b = launder_v(tid_raw);
    { const int tid = tid_l1b, wave = (G == 2) ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6), lane = tid & 63, q = lane >> 4, j = lane & 15;
    uint4 dh[G][4][2];      // B fragments of d(hidden-1): point 16 wave + j, hidden units 32 s + 8 q + t
    float kscale[G];
#pragma unroll
    for (int g = 0; g < G; g++) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            dh[g][s][0] = Hp[g * 2048 + (4 * s + q) * 64 + 16 * wave + j];
            dh[g][s][1] = Hp[g * 2048 + 1024 + (4 * s + q) * 64 + 16 * wave + j];
        }
        kscale[g] = sInv[g * 64 + 16 * wave + j] * a.hw[g].kback;
    }
    const int mypt = 16 * wave + j;
    const float px_ = sPt[mypt * 3], py_ = sPt[mypt * 3 + 1], iz_ = 1.0f / sPt[mypt * 3 + 2];
    const float kx = 2.0f / a.crop * a.fx, ky = 2.0f / a.crop * a.fy;
    const float j0x = kx * iz_, j0y = ky * iz_, j0zu = -kx * px_ * iz_ * iz_, j0zv = -ky * py_ * iz_ * iz_;
    float gx = 0.f, gy = 0.f, gz = 0.f;     // coordinate-gradient partials of point mypt over the channels this lane sees
    if (USEP) {
        // im_feat part of the coordinate gradient: d/du = sum_t cu_t <d(hidden-1), P row of tap t> (and cv for d/dv).  Same lane layout as the
        // forward blend -- thread = (point, 8-column block group): the four lanes of a point are adjacent and read 128 contiguous bytes of a row
        // per pair of instructions -- with d(hidden-1) taken straight from the operand planes (hi + lo), which are [k block][point] already.
        // Lane 4 j' + seg of wave w works on point 16 w + j': the owner lane (q = 0, j = j') of the SAME wave picks the sums up with one shuffle.
        // (the inputs of this phase's address arithmetic go through launder(): the forward blend computed the same resolution-derived floats, LDS addresses and the
        //  64-bit row base from them a whole kernel ago, and hipcc kept those alive -- in scratch, at 256 / 168 VGPRs -- rather than recompute a handful of
        //  integer instructions: 44 B per lane of spills, 117 MB of scratch writes per launch in round 5)
        const int R = launder_s(a.res[0]), pw_ = launder_s(a.pw), b_ = launder_s(b), tid_ = tid;
        const rsrc_t Pb = make_rsrc(a.proj + (size_t)b_ * R * R * pw_, (unsigned)(R * R * pw_) * 4u);
        const int spt = tid_ >> 2, seg = tid_ & 3;
        unsigned o[4]; float cu[4], cv[4];
        proj_geom(sUV, spt, R, pw_, o, cu, cv, true);
#pragma unroll
        for (int k = 0; k < 4; k++) o[k] = (o[k] + 8u * seg) * 4u;             // byte offset of this thread's first 8-column block in tap row k
        float dot[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < G; g++) {
            float dg[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i2 = 0; i2 < 4; i2 += 2) {
                // two 8-column blocks per round: 16 independent 16-byte loads in flight, then 64 multiply-adds
                float4 pr[2][4][2];
                uint4 xh[2], xl[2];
#pragma unroll
                for (int ii = 0; ii < 2; ii++) {
                    const int kb = 4 * (i2 + ii) + seg;
                    const unsigned pc = (unsigned)a.hw[g].pcol * 4u;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        pr[ii][k][0] = GATHER_P4(Pb, o[k] + 128u * (i2 + ii), pc); pr[ii][k][1] = GATHER_P4(Pb, o[k] + 128u * (i2 + ii) + 16u, pc);
                    }
                    xh[ii] = Hp[g * 2048 + kb * 64 + spt]; xl[ii] = Hp[g * 2048 + 1024 + kb * 64 + spt];
                }
#pragma unroll
                for (int ii = 0; ii < 2; ii++) {
                    const h8 hh = as_h8(xh[ii]), hl = as_h8(xl[ii]);
                    float x[8];
#pragma unroll
                    for (int t = 0; t < 8; t++) x[t] = __builtin_fmaf((float)hh[t], 1.0f, (float)hl[t]);     // hi + lo: one v_fma_mix_f32 per value (both halves extended inside the fma)
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float4 p0 = pr[ii][k][0], p1 = pr[ii][k][1];
                        dg[k] = __builtin_fmaf(x[7], p1.w, __builtin_fmaf(x[6], p1.z, __builtin_fmaf(x[5], p1.y, __builtin_fmaf(x[4], p1.x,
                                __builtin_fmaf(x[3], p0.w, __builtin_fmaf(x[2], p0.z, __builtin_fmaf(x[1], p0.y, __builtin_fmaf(x[0], p0.x, dg[k]))))))));
                    }
                }
            }
            const float ks = sInv[g * 64 + spt] * a.hw[g].kback * a.u1inv;
#pragma unroll
            for (int k = 0; k < 4; k++) dot[k] = __builtin_fmaf(ks, dg[k], dot[k]);
        }
        float su = cu[0] * dot[0] + cu[1] * dot[1] + cu[2] * dot[2] + cu[3] * dot[3];
        float sv = cv[0] * dot[0] + cv[1] * dot[1] + cv[2] * dot[2] + cv[3] * dot[3];
        su += dpp_mov<0xB1>(su); sv += dpp_mov<0xB1>(sv);          // the four column-block lanes of a point are one quad
        su += dpp_mov<0x4E>(su); sv += dpp_mov<0x4E>(sv);
        su = __shfl(su, 4 * j, 64); sv = __shfl(sv, 4 * j, 64);
        if (q == 0) { gx = su * j0x; gy = sv * j0y; gz = __builtin_fmaf(sv, j0zv, su * j0zu); }
    }
    {   // tap geometry (d/du, d/dv coefficient form) of the first two maps of the backward loop into the ring, ahead of the barrier below
        int mi, co; chunk_info(C0, mi, co);
        if (SHGEO && wave == 0) geom_compute<2>(a, mi, sUV, lane, sGeo);
        if (SHGEO && wave == 1 && mi + 1 < 8) geom_compute<2>(a, mi + 1, sUV, lane, sGeo);
    }
    __syncthreads();        // region 0 changes role again: activation planes -> d(feature) rows + weight slab
    // Every wave needs the whole weight slab of a chunk (the waves split the POINTS here): the workgroup stages it once
    // in LDS with the asynchronous global->LDS DMA (16 B per lane, lane-linear destination = the fragment order), no VGPRs.
    //
    // The chunk's contribution to the coordinate gradient is  sum_c d feat[c] (d f_c / d u) = sum_t cu_t <d feat, texel_t>  (and cv for v): FOUR dot
    // products of d feat with the raw tap rows per (point, chunk) -- 4 multiply-adds per channel -- instead of blending the taps to d f / d u and
    // d f / d v first (8 per channel) and contracting afterwards (2 more).  The dot products run in the gather layout (thread = 16-byte piece `sub`
    // of the taps of points pp, pp + 32), so d feat goes through LDS once ([point][32 channels] fp32, half the bytes the two tap-difference
    // buffers took) and the partial sums of a map's chunks stay in registers until the map ends (one 8-lane DPP reduction per map).
    float *sD = reinterpret_cast<float *>(lds);                          // [64 points][TS] d feat of the current chunk
    uint4 *Sl = lds + 1152;                                              // [G][4 s][2 ct][hi|lo][64 lanes]
#define SLAB_DMA(ci_)                                                                                                        \
    _Pragma("unroll") for (int g_ = 0; g_ < G; g_++)                                                                         \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; i_++)                                                                     \
            __builtin_amdgcn_global_load_lds(a.hw[g_].w1c + (size_t)(ci_) * 1024 + 256 * i_ + tid,                          \
                                             (__attribute__((address_space(3))) void *)(Sl + g_ * 1024 + 256 * i_ + wave * 64), 16, 0, 0);
    // SLAB_VIA_REGS (default 1): the slab through registers (8 buffer loads of 16 B per thread, 8 ds_write_b128) instead of the LDS DMA.  Measured
    // (tools/bench_scripts/gather_patterns.hip, pattern 5): the DMA path delivers 16-byte-per-lane loads at HALF the rate of the path to the
    // registers (10.5-11.2 vs 20-21.4 TB/s for the same lane-linear 1 KB per wave from L2) -- and this kernel is bound by that delivery.
#ifndef SLAB_VIA_REGS
#define SLAB_VIA_REGS 1
#endif
    // (the one-head kernels -- 168 VGPRs, three workgroups per CU -- have no room for the eight fragments in flight and keep the DMA)
    constexpr bool SLABR = (G == 2) && SLAB_VIA_REGS;
    uint4 slabr[G][4];
#define SLAB_LOAD(ci_)                                                                                                       \
    _Pragma("unroll") for (int g_ = 0; g_ < G; g_++) {                                                                       \
        const rsrc_t sr_ = make_rsrc(a.hw[g_].w1c, 0x40000000u);                                                             \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; i_++) slabr[g_][i_] = bload_u4(sr_, (unsigned)(256 * i_ + tid) * 16u, (unsigned)(ci_) * 16384u); \
    }
#define SLAB_STORE()                                                                                                         \
    _Pragma("unroll") for (int g_ = 0; g_ < G; g_++)                                                                         \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; i_++) Sl[g_ * 1024 + 256 * i_ + tid] = slabr[g_][i_];
    if (SLABR) { SLAB_LOAD(C0) SLAB_STORE() } else { SLAB_DMA(C0) }
    TapGeom<2> tgb;
    { int mi, co; chunk_info(C0, mi, co); if (SHGEO) geom_fetch<2>(mi, sGeo, tid, tgb); else taps_geom(a, mi, sUV, tid, tgb); taps_issue(a, b, mi, co, tgb, tp); }
    __syncthreads();
    PCLK(3);
    const int gsub = tid & 7, gpp = tid >> 3;                            // gather role: piece gsub of the taps of points gpp, gpp + 32
    float Dt[2][4];                                                      // <d feat, texel_t> of the current MAP, this thread's channels
#pragma unroll
    for (int pass = 0; pass < 2; pass++)
#pragma unroll
        for (int k = 0; k < 4; k++) Dt[pass][k] = 0.f;
    float hx[2] = {0.f, 0.f}, hy[2] = {0.f, 0.f}, hz[2] = {0.f, 0.f};    // coordinate gradient of points gpp, gpp + 32 from the orthographic maps, this thread's channels
    float hp[2][2] = {{0.f, 0.f}, {0.f, 0.f}};                           // (su, sv) of the perspective maps
    for (int ci = C0; ci < NCHUNK; ci++) {
        int mi, co; chunk_info(ci, mi, co);
        f32x4 dd[G][2];
#pragma unroll
        for (int g = 0; g < G; g++) { dd[g][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; dd[g][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int s = 0; s < 4; s++) {
            // all (head, channel tile) accumulators take their hi.hi product first, then hi.lo, then lo.hi: 2 G independent MFMAs
            // sit between two that touch the same accumulator
            h8 wh[G][2], wl[G][2], xh[G], xl[G];
#pragma unroll
            for (int g = 0; g < G; g++) {
                const uint4 *f = Sl + (((g * 4 + s) * 2) * 2) * 64 + lane;
                wh[g][0] = as_h8(f[0]); wl[g][0] = as_h8(f[64]); wh[g][1] = as_h8(f[128]); wl[g][1] = as_h8(f[192]);
                xh[g] = as_h8(dh[g][s][0]); xl[g] = as_h8(dh[g][s][1]);
            }
#pragma unroll
            for (int g = 0; g < G; g++) { dd[g][0] = MFMAH(wh[g][0], xh[g], dd[g][0]); dd[g][1] = MFMAH(wh[g][1], xh[g], dd[g][1]); }
#pragma unroll
            for (int g = 0; g < G; g++) { dd[g][0] = MFMAH(wh[g][0], xl[g], dd[g][0]); dd[g][1] = MFMAH(wh[g][1], xl[g], dd[g][1]); }
#pragma unroll
            for (int g = 0; g < G; g++) { dd[g][0] = MFMAH(wl[g][0], xh[g], dd[g][0]); dd[g][1] = MFMAH(wl[g][1], xh[g], dd[g][1]); }
        }
        // d feat of point mypt, channels 16 ct + 4 q .. +3 (true scale: the per-point normalisation and the chain's weight scales undone)
#pragma unroll
        for (int ct = 0; ct < 2; ct++) {
            float d[4];
#pragma unroll
            for (int r = 0; r < 4; r++) { d[r] = dd[0][ct][r] * kscale[0]; if (G == 2) d[r] = __builtin_fmaf(dd[G - 1][ct][r], kscale[G - 1], d[r]); }
            *reinterpret_cast<float4 *>(sD + mypt * TS + 16 * ct + 4 * q) = make_float4(d[0], d[1], d[2], d[3]);
        }
        __syncthreads();                                   // slab(ci) fully consumed, d feat of chunk ci visible
        // read this thread's d feat FIRST, then start the DMA of the next slab: hipcc orders an LDS read after an LDS-DMA
        // with a full vmcnt(0) wait (they may alias), which put the whole DMA latency in front of the dot products
        float4 d4[2];
#pragma unroll
        for (int pass = 0; pass < 2; pass++) d4[pass] = *reinterpret_cast<const float4 *>(sD + (gpp + 32 * pass) * TS + 4 * gsub);
        if (ci + 1 < NCHUNK) { if (SLABR) { SLAB_LOAD(ci + 1) } else { SLAB_DMA(ci + 1) asm volatile("" ::: "memory"); } }
#pragma unroll
        for (int pass = 0; pass < 2; pass++)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float4 t = tp.t[pass][k];
                Dt[pass][k] = __builtin_fmaf(d4[pass].w, t.w, __builtin_fmaf(d4[pass].z, t.z, __builtin_fmaf(d4[pass].y, t.y, __builtin_fmaf(d4[pass].x, t.x, Dt[pass][k]))));
            }
        int m2i = -1, c2o = 0;
        if (ci + 1 < NCHUNK) chunk_info(ci + 1, m2i, c2o);
        if (c2o == 0) {
            // the map ends with this chunk (uniform branch): apply the tap coefficients to this thread's PARTIAL dot products (everything is linear,
            // the sum over the 8 pieces of a tap row waits until after the loop) and the projection Jacobians (camera.py:52-90;
            // chore_triplane.py:220-251):  right (u, v) = (c2, c1): gz += su, gy += sv;  back (-c0, c1): gx -= su, gy += sv;  top (c0, -c2): gx += su,
            // gz -= sv;  perspective maps collect (su, sv) and take their per-point Jacobian at the end
            const int pr = map_proj(mi);
#pragma unroll
            for (int pass = 0; pass < 2; pass++) {
                float su = tgb.c[0][pass][0] * Dt[pass][0], sv = tgb.c[1][pass][0] * Dt[pass][0];
#pragma unroll
                for (int k = 1; k < 4; k++) { su = __builtin_fmaf(tgb.c[0][pass][k], Dt[pass][k], su); sv = __builtin_fmaf(tgb.c[1][pass][k], Dt[pass][k], sv); }
#pragma unroll
                for (int k = 0; k < 4; k++) Dt[pass][k] = 0.f;
                if (pr == 0) { hp[pass][0] += su; hp[pass][1] += sv; }
                else if (pr == 1) { hz[pass] += su; hy[pass] += sv; }
                else if (pr == 2) { hx[pass] -= su; hy[pass] += sv; }
                else { hx[pass] += su; hz[pass] -= sv; }
            }
        }
        if (ci + 1 < NCHUNK) {
            // taps of chunk ci+1 AFTER the DMA (vmcnt retires in order): the barrier below then waits for the slab only, the gather
            // stays in flight across it
            if (c2o == 0) {         // see the forward loop: fetch this map's slot, one wave refills the other with the next map
                if (SHGEO) {
                    geom_fetch<2>(m2i, sGeo, tid, tgb);
                    if (m2i + 1 < 8 && wave == (m2i & 3)) geom_compute<2>(a, m2i + 1, sUV, lane, sGeo);
                } else
                    taps_geom(a, m2i, sUV, tid, tgb);
            }
            taps_issue(a, b, m2i, c2o, tgb, tp);
            if (SLABR) {
                // the slab pieces (older than the 8 tap loads just issued: the compiler's own vmcnt for them leaves the taps in flight) go to LDS; everybody
                // read the old slab before the barrier in the middle of the iteration
                SLAB_STORE()
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {
                // slab(ci+1) landed, the d feat rows free again.  NOT __syncthreads(): its fence is a vmcnt(0), which would also wait for the
                // 8 tap loads just issued; the DMA pieces are older than those, so "at most 8 outstanding" means the slab is in LDS.
                asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }
#undef SLAB_DMA
#undef SLAB_LOAD
#undef SLAB_STORE
    PCLK(4);
    // the gathered-map part: perspective Jacobian (kx/z, ky/z, -kx x/z^2, -ky y/z^2), sum over the 8 pieces of a tap row (xor 1, xor 2 inside the quad,
    // then the half-row mirror), hand-over from the gather layout (piece 0 of points gpp, gpp + 32) to the owner lanes (q == 0: point mypt)
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const float *pp3 = sPt + (gpp + 32 * pass) * 3;
        const float iz = 1.0f / pp3[2];
        float vx = __builtin_fmaf(hp[pass][0], kx * iz, hx[pass]), vy = __builtin_fmaf(hp[pass][1], ky * iz, hy[pass]);
        float vz = __builtin_fmaf(hp[pass][1], -ky * pp3[1] * iz * iz, __builtin_fmaf(hp[pass][0], -kx * pp3[0] * iz * iz, hz[pass]));
        vx += dpp_mov<0xB1>(vx); vy += dpp_mov<0xB1>(vy); vz += dpp_mov<0xB1>(vz);
        vx += dpp_mov<0x4E>(vx); vy += dpp_mov<0x4E>(vy); vz += dpp_mov<0x4E>(vz);
        vx += dpp_mov<0x141>(vx); vy += dpp_mov<0x141>(vy); vz += dpp_mov<0x141>(vz);
        if (gsub == 0) { float *o = sD + (gpp + 32 * pass) * 4; o[0] = vx; o[1] = vy; o[2] = vz; }
    }
    __syncthreads();
    if (q == 0) { gx += sD[mypt * 4]; gy += sD[mypt * 4 + 1]; gz += sD[mypt * 4 + 2]; }
    {   // direct xyz features: d feat[608..610]: "chunk" 19 of the slab array, rows 0..2 of its first 16-row tile, straight from L2
        f32x4 dz[G];
#pragma unroll
        for (int g = 0; g < G; g++) {
            dz[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const rsrc_t fr = make_rsrc(a.hw[g].w1c + (size_t)NCHUNK * 1024, 1024u * 16u);
                const h8 wh = as_h8(bload_u4(fr, (unsigned)lane * 16u + (unsigned)(s * 4096), 0u)), wl = as_h8(bload_u4(fr, (unsigned)lane * 16u + (unsigned)(s * 4096 + 1024), 0u));
                const h8 xh = as_h8(dh[g][s][0]), xl = as_h8(dh[g][s][1]);
                dz[g] = MFMAH(wh, xh, dz[g]); dz[g] = MFMAH(wh, xl, dz[g]); dz[g] = MFMAH(wl, xh, dz[g]);
            }
        }
        if (q == 0) {
#pragma unroll
            for (int g = 0; g < G; g++) { gx += dz[g][0] * kscale[g]; gy += dz[g][1] * kscale[g]; gz += dz[g][2] * kscale[g]; }
        }
    }
    // (all three parts -- hoisted projection, gathered maps, xyz -- were accumulated by the owner lane q == 0 of the point)
    if (*sOvf) gx = gy = gz = __builtin_nanf("");
    double acc_accel = 0.0;
    if (q == 0) {
        const int n = n0 + mypt;
        if (n < a.N) {
            const int pn = sIn[mypt] >> 1;
            if (MODE == MODE_PROJECT) {
                // samples <- samples - normalize(gradient) * target  (F.normalize: g / max(|g|, 1e-12); generator.py:97)
                const float dft = sDf[mypt], s = dft / fmaxf(sqrtf(gx * gx + gy * gy + gz * gz), 1e-12f);
                float *o = a.pts_out + ((size_t)b * a.N + pn) * 3;
                o[0] = px_ - gx * s; o[1] = py_ - gy * s; o[2] = sPt[mypt * 3 + 2] - gz * s;
                if (a.dft_out) a.dft_out[(size_t)b * a.N + pn] = dft;
            } else if (MODE == MODE_HUMAN && (a.accum || a.term_accel)) {
                float *o = a.dpts + ((size_t)b * a.N + pn) * 3;
                float g[3] = {gx, gy, gz};
                if (a.accum) { g[0] = o[0] + gx; g[1] = o[1] + gy; g[2] = o[2] + gz; }      // (l + q): the same bits as the (q + l) of the separate launches
                if (a.term_accel) {
                    // accel_loss_kernel for the three elements of this vertex in frame b (frames clamped exactly like there)
                    const int B = a.B, f = b;
                    const float *vb = a.pts + (size_t)pn * 3; const size_t fs = (size_t)a.N * 3;
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const float vm2 = vb[(size_t)min(max(f - 2, 0), B - 1) * fs + c], vm1 = vb[(size_t)min(max(f - 1, 0), B - 1) * fs + c], v0 = vb[(size_t)f * fs + c],
                                    vp1 = vb[(size_t)min(max(f + 1, 0), B - 1) * fs + c], vp2 = vb[(size_t)min(max(f + 2, 0), B - 1) * fs + c];
                        const float a_m = (f - 1 >= 1 && f - 1 <= B - 2) ? 2.f * vm1 - vm2 - v0 : 0.f;
                        const float a_0 = (f >= 1 && f <= B - 2) ? 2.f * v0 - vm1 - vp1 : 0.f;
                        const float a_p = (f + 1 >= 1 && f + 1 <= B - 2) ? 2.f * vp1 - v0 - vp2 : 0.f;
                        acc_accel += (double)(1.f * a_0 * a_0);
                        g[c] += a.accel_gs * 1.f * (2.f * a_0 - a_m - a_p);
                    }
                }
                o[0] = g[0]; o[1] = g[1]; o[2] = g[2];
            } else {
                float *o = a.dpts + ((size_t)b * a.N + pn) * 3;
                o[0] = gx; o[1] = gy; o[2] = gz;
            }
        }
    }
    if (MODE == MODE_HUMAN && a.term_accel) {
        // the stencil term: mean over (B - 2) frames x 3 N elements of a_f^2; one fp64 atomic per workgroup
        double s_ = acc_accel;
        for (int o_ = 32; o_ > 0; o_ >>= 1) s_ += __shfl_xor(s_, o_, 64);
        __syncthreads();
        if (lane == 0) sRed[wave] = s_;
        __syncthreads();
        if (tid == 0) atomicAdd(a.term_accel, (sRed[0] + sRed[1] + sRed[2] + sRed[3]) / ((double)(a.B - 2) * (double)a.N * 3.0));
    }
    if (clk_probe && tid_l1b == 0) {
        atomicAdd(a.clk, (unsigned long long)clock64() - clk_c0); atomicAdd(a.clk + 1, (unsigned long long)wall_clock64() - clk_r0); atomicAdd(a.clk + 2, 1ull);
    }
    }       // (scope of the re-derived lane indices)
}


#ifdef VT_EXPERIMENTS      /* measured-negative kernel variants: experiments/ (not in the default library) */
#include "experiments/query128.h"           // the object objective on 128-point tiles, weights once per 128 points (VT_QUERY_OBJECT_TILE=128)
#include "experiments/query_human8.h"      // 512-thread thin waves (vt_query_set_human_kernel(512))
#include "experiments/query_pc.h"          // producer / consumer waves, 128 points per workgroup (vt_query_set_human_kernel(128))
#endif

// ---------------------------------------------------------------------------------------------------
// MFMA operand-layout self test (f32-input 16x16x4, A = 16x4, B = 4x16 asymmetric): out = A.B, row-major 16x16
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
// handle: weights split and re-laid out once.  Internal channel order: 608 map channels (im_feat 256, tmpx 64, tri_tmpx 3x32,
// tri_feat 3x64), then x, y, z-2.2, then one zero pad  ->  KTOT = 612.
// ---------------------------------------------------------------------------------------------------
static const int kHeadDims[5] = {2, 9, 14, 3, 1};
static inline int orig_channel(int k) { return k < 256 ? k : (k < 608 ? k + 3 : (k < 611 ? k - 608 + 256 : -1)); }

// Power-of-two weight scale s_l of a layer: max |s_l W_l| in [1, 2).  A split weight hi + lo carries an absolute error <= max(2^-22 |s w|, 2^-25)
// (fp16 subnormals are honoured by the f16 MFMA: the lo halves of small weights live there), i.e. <= 2^-25 of the matrix maximum for every
// weight below 2^-3 of it and 2^-22 relative above.  The accumulators of layer l are the operands of layer l + 1 (U_(l+1) = s_l U_l, HeadW), so the
// scales of the three hidden matrices decide how far below U_4 = 2^ACT_EXP0 the feature scale U_1 sits.
static int weight_scale_exp(const float *w, size_t n)
{
    float m = 0.f;
    for (size_t i = 0; i < n; i++) m = fmaxf(m, fabsf(w[i]));
    if (!(m > 0.f) || !std::isfinite(m)) return 0;
    int e; frexpf(m, &e);          // m = f * 2^e, f in [0.5, 1)
    return 1 - e;                  // s m in [1, 2)
}
static inline void put_split(_Float16 *hi, _Float16 *lo, float x)
{
    const _Float16 h = (_Float16)x;
    *hi = h; *lo = (_Float16)(x - (float)h);
}
// T-pack: [K/32][4 waves][2 nt][hi|lo][64 lanes][8 halves] of the 128 x K matrix get(row, k) (already scaled)
template <typename F>
static void pack_T(_Float16 *dst, int K, F get)
{
    for (int s = 0; s < K / 32; s++) for (int w = 0; w < 4; w++) for (int nt = 0; nt < 2; nt++) for (int l = 0; l < 64; l++) for (int t = 0; t < 8; t++) {
        const size_t base = ((((size_t)s * 4 + w) * 2 + nt) * 2) * 64;
        put_split(dst + ((base + l) * 8 + t), dst + ((base + 64 + l) * 8 + t), get(32 * w + 16 * nt + (l & 15), 32 * s + 8 * (l >> 4) + t));
    }
}

extern "C" int vt_sifnet_create(vt_sifnet **out, const float *const *w, const float *const *bvec, const float *cam, void *stream)
{
    VT_REQUIRE(out && w && bvec && cam, "vt_sifnet_create: null argument");
    hipStream_t st = vt_stream(stream);
    // per head (halves): w1p 20*4*2*2*64*8 | w1c 20*1024*8 | (w2p, w2tp, w3p, w3tp) 4 x 4*4*2*2*64*8 | w4p 4*2*64*8 | w4tp 4*2*2*64*8 ; then ACT_LEVELS x 512 floats: U_3 b2 | U_4 b3 (128 each) | b4 (16) | pad
    const size_t n_w1p = (size_t)NSTEP1 * 4 * 2 * 2 * 64 * 8, n_w1c = (size_t)(NCHUNK + 1) * 1024 * 8, n_mid = (size_t)4 * 4 * 2 * 2 * 64 * 8,
                 n_w4p = (size_t)4 * 2 * 64 * 8, n_w4t = (size_t)4 * 2 * 2 * 64 * 8;
    const size_t halves = n_w1p + n_w1c + 4 * n_mid + n_w4p + n_w4t, nbias = 512 * ACT_LEVELS;
    const size_t per_head = halves * sizeof(_Float16) + nbias * sizeof(float);     // multiple of 16 bytes
    unsigned char *host = new unsigned char[per_head * 5]();
    vt_sifnet *h = new vt_sifnet();
    VT_HIP(hipMalloc(&h->blob, per_head * 5));
    VT_HIP(hipMalloc(&h->projw, sizeof(float) * 256 * PROJ_COLS));
    float *projw_host = new float[256 * PROJ_COLS]();
    // scales: per head s_1..s_4 from the weights; the feature scale U_1 is shared by the heads of a launch (one set of feature operand planes),
    // chosen so that no head's last hidden operand scale U_4 = s_1 s_2 s_3 U_1 exceeds 2^ACT_EXP0 (|h_3| < 1023 at range level 0, as before)
    int se[5][4], e1 = 64;
    for (int hd = 0; hd < 5; hd++) {
        se[hd][0] = weight_scale_exp(w[hd * 4], (size_t)128 * VT_FEAT); se[hd][1] = weight_scale_exp(w[hd * 4 + 1], 128 * 128);
        se[hd][2] = weight_scale_exp(w[hd * 4 + 2], 128 * 128); se[hd][3] = weight_scale_exp(w[hd * 4 + 3], (size_t)kHeadDims[hd] * 128);
        for (int l = 0; l < 3; l++) se[hd][l] = std::max(se[hd][l], 0);        // hidden layers: never scale DOWN (keeps U_(l+1) >= U_l: headroom only shrinks where weights are small)
        e1 = std::min(e1, ACT_EXP0 - (se[hd][0] + se[hd][1] + se[hd][2]));
    }
    e1 = std::max(e1, -8);
    h->u1_exp = e1;
    for (int hd = 0; hd < 5; hd++) {
        _Float16 *p = reinterpret_cast<_Float16 *>(host + per_head * hd);
        const unsigned char *d = reinterpret_cast<const unsigned char *>(h->blob) + per_head * hd;
        auto dev = [&](size_t off_halves) { return reinterpret_cast<const uint4 *>(d + off_halves * sizeof(_Float16)); };
        const int ko = kHeadDims[hd];
        HeadW &H = h->head[hd];
        H.kout = ko; H.id = hd; H.pcol = hd == 0 ? 0 : (hd == 2 ? 128 : -1);
        const float *W1 = w[hd * 4], *W2 = w[hd * 4 + 1], *W3 = w[hd * 4 + 2], *W4 = w[hd * 4 + 3];   // (out, in), W1 in reference channel order
        const float s1 = ldexpf(1.0f, se[hd][0]), s2 = ldexpf(1.0f, se[hd][1]), s3 = ldexpf(1.0f, se[hd][2]), s4 = ldexpf(1.0f, se[hd][3]);
        const float U1 = ldexpf(1.0f, e1), U2 = s1 * U1, U3 = s2 * U2, U4 = s3 * U3;      // operand scales at range level 0
        H.cout = 1.0f / (U4 * s4); H.kback = 1.0f / (s1 * s2 * s3 * s4);
        H.goexp = std::max(GO_EXP - (se[hd][1] + se[hd][2] + se[hd][3]), -12);
        h->cout0[hd] = H.cout;
        // internal channel 611 is the constant one: its weight is the layer-1 bias (22 significand bits like every other weight)
        auto w1 = [&](int u, int k) { if (k == 611) return bvec[hd * 4][u] * s1; const int c = k < KTOT ? orig_channel(k) : -1; return c < 0 ? 0.f : W1[(size_t)u * VT_FEAT + c] * s1; };   // internal order
        size_t o = 0;
        H.w1p = dev(o); pack_T(p + o, 32 * NSTEP1, [&](int n, int k) { return w1(n, k); }); o += n_w1p;
        H.w1c = dev(o);
        for (int ci = 0; ci <= NCHUNK; ci++) for (int s = 0; s < 4; s++) for (int ct = 0; ct < 2; ct++) for (int l = 0; l < 64; l++) for (int t = 0; t < 8; t++) {
            const size_t base = (size_t)ci * 1024 + ((s * 2 + ct) * 2) * 64;
            put_split(p + o + (base + l) * 8 + t, p + o + (base + 64 + l) * 8 + t, w1(32 * s + 8 * (l >> 4) + t, 32 * ci + 16 * ct + (l & 15)));
        }
        o += n_w1c;
        H.w2p = dev(o); pack_T(p + o, 128, [&](int n, int k) { return W2[n * 128 + k] * s2; }); o += n_mid;
        H.w2tp = dev(o); pack_T(p + o, 128, [&](int n, int k) { return W2[k * 128 + n] * s2; }); o += n_mid;
        H.w3p = dev(o); pack_T(p + o, 128, [&](int n, int k) { return W3[n * 128 + k] * s3; }); o += n_mid;
        H.w3tp = dev(o); pack_T(p + o, 128, [&](int n, int k) { return W3[k * 128 + n] * s3; }); o += n_mid;
        H.w4p = dev(o);
        for (int s = 0; s < 4; s++) for (int l = 0; l < 64; l++) for (int t = 0; t < 8; t++) {
            const int oo = l & 15, k = 32 * s + 8 * (l >> 4) + t;
            put_split(p + o + (((size_t)s * 2) * 64 + l) * 8 + t, p + o + (((size_t)s * 2 + 1) * 64 + l) * 8 + t, oo < ko ? W4[oo * 128 + k] * s4 : 0.f);
        }
        o += n_w4p;
        H.w4tp = dev(o); pack_T(p + o, 32, [&](int n, int k) { return k < ko ? W4[k * 128 + n] * s4 : 0.f; }); o += n_w4t;
        if (H.pcol >= 0)       // im_feat occupies reference channels 0..255 (chore_triplane.py:97-164 feature order)
            for (int c = 0; c < 256; c++) for (int u = 0; u < 128; u++) projw_host[(size_t)c * PROJ_COLS + H.pcol + u] = s1 * W1[(size_t)u * VT_FEAT + c];
        float *bp = reinterpret_cast<float *>(p + o);
        const float *bd = reinterpret_cast<const float *>(d + o * sizeof(_Float16));
        h->bias_dev[hd] = bd;
        for (int lv = 0; lv < ACT_LEVELS; lv++) {
            // hidden biases of layers 2, 3 in the operand units of the layer's output at range level lv (exact: powers of two)
            const float dn = ldexpf(1.0f, -ACT_LEVEL_SHIFT * lv);
            for (int i = 0; i < 128; i++) { bp[lv * 512 + i] = bvec[hd * 4 + 1][i] * (U3 * dn); bp[lv * 512 + 128 + i] = bvec[hd * 4 + 2][i] * (U4 * dn); }
            memcpy(bp + lv * 512 + 256, bvec[hd * 4 + 3], ko * sizeof(float));
        }
        H.b2 = bd; H.b3 = bd + 128; H.b4 = bd + 256;
    }
    VT_HIP(hipMemcpyAsync(h->blob, host, per_head * 5, hipMemcpyHostToDevice, st));
    VT_HIP(hipMemcpyAsync(h->projw, projw_host, sizeof(float) * 256 * PROJ_COLS, hipMemcpyHostToDevice, st));
    VT_HIP(hipStreamSynchronize(st));
    delete[] host; delete[] projw_host;
    for (int i = 0; i < 5; i++) h->cam[i] = cam[i];
    h->f32 = nullptr; h->precision.store(VT_PRECISION_SPLIT_F16);
    { const int rc = f32q::create(&h->f32, w, bvec, cam, stream); if (rc) return rc; }
    *out = h;
    return VT_OK;
}
extern "C" void vt_sifnet_destroy(vt_sifnet *h) { if (!h) return; (void)hipFree(h->blob); (void)hipFree(h->projw); f32q::destroy(h->f32); delete h; }

extern "C" int vt_sifnet_set_precision(vt_sifnet *h, int mode)
{
    VT_REQUIRE(h && (mode == VT_PRECISION_SPLIT_F16 || mode == VT_PRECISION_FP32), "vt_sifnet_set_precision: bad handle or mode");
    h->precision.store(mode, std::memory_order_relaxed);
    return VT_OK;
}
extern "C" int vt_sifnet_get_precision(const vt_sifnet *h) { return h ? h->precision.load(std::memory_order_relaxed) : VT_ERR_ARG; }
// which arithmetic serves a call: the handle's default, overridden per call by the maps (vt_maps::force_fp32 -- per batch, so concurrent fits
// through one handle never see each other's switch)
#define VT_IS_F32(h_, m_) ((h_) && ((h_)->precision.load(std::memory_order_relaxed) == VT_PRECISION_FP32 || ((m_) && (m_)->force_fp32)))
// decoder head `id` at operand-range level `lvl` (vt_maps::act_level): same weights, biases and output scale of that level
static HeadW head_at(const vt_sifnet *h, int id, int lvl)
{
    HeadW H = h->head[id];
    const float *bd = h->bias_dev[id] + 512 * lvl;
    H.b2 = bd; H.b3 = bd + 128; H.b4 = bd + 256;
    H.cout = h->cout0[id] * ldexpf(1.0f, ACT_LEVEL_SHIFT * lvl);
    return H;
}

// ---- hoisted projection: P[m][n] = sum_c im_feat[m][c] Wp[c][n], m over all texels of the batch, c < 256, n < PROJ_COLS; fp32 MFMA
// (16x16x4), one workgroup per 64 texels: the A tile (64 x 256) sits in LDS, wave w owns columns 64 w .. +63 (4 x 4 tiles) and streams
// its slice of Wp (256 KB, L2 resident) from global.  0.2 TFLOP per 96-frame batch, once per batch.
#define PJ_AS 260
typedef float f32x4_ __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void proj_gemm_kernel(const float *__restrict__ A, const float *__restrict__ Wp, float *__restrict__ P, long M, float u1)
{
    extern __shared__ __attribute__((aligned(16))) float sA[];      // [64][PJ_AS]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kq = lane >> 4, i = lane & 15;
    const long m0 = (long)blockIdx.x * 64;
    for (int t = tid; t < 64 * 64; t += 256) {
        const int r = t >> 6, c4 = t & 63;
        const float4 v = (m0 + r < M) ? *reinterpret_cast<const float4 *>(A + (m0 + r) * 256 + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4 *>(sA + r * PJ_AS + 4 * c4) = v;
    }
    __syncthreads();
    f32x4_ acc[4][4];
#pragma unroll
    for (int rt = 0; rt < 4; rt++)
#pragma unroll
        for (int ct = 0; ct < 4; ct++) acc[rt][ct] = (f32x4_){0.f, 0.f, 0.f, 0.f};
    const float *__restrict__ wcol = Wp + 64 * wave + i;
#pragma unroll 4
    for (int ks = 0; ks < 64; ks++) {
        float af[4], bf[4];
#pragma unroll
        for (int ct = 0; ct < 4; ct++) bf[ct] = wcol[(size_t)(4 * ks + kq) * PROJ_COLS + 16 * ct];
#pragma unroll
        for (int rt = 0; rt < 4; rt++) af[rt] = sA[(16 * rt + i) * PJ_AS + 4 * ks + kq];
#pragma unroll
        for (int rt = 0; rt < 4; rt++)
#pragma unroll
            for (int ct = 0; ct < 4; ct++) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[rt], bf[ct], acc[rt][ct], 0, 0, 0);
    }
#pragma unroll
    for (int rt = 0; rt < 4; rt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const long m = m0 + 16 * rt + 4 * kq + r;
            if (m < M) {
#pragma unroll
                for (int ct = 0; ct < 4; ct++) P[m * PROJ_COLS + 64 * wave + 16 * ct + i] = acc[rt][ct][r] * u1;      // exact: a power of two
            }
        }
}

#ifdef PHASE_CLK
extern "C" int vt_phase_clk(unsigned long long *out, int reset)
{
    if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(g_phase));
    if (reset) { unsigned long long z[32] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)); }
    return 0;
}
#endif
extern "C" long vt_query_projection_floats(const vt_maps *maps, int B)
{
    if (!maps || B <= 0 || maps->res[0] < 2) return 0;
    return (long)B * maps->res[0] * maps->res[0] * PROJ_COLS;
}

extern "C" int vt_query_build_projection(const vt_sifnet *h, const vt_maps *maps, int B, float *proj, void *stream)
{
    VT_REQUIRE(h && maps && proj && B > 0 && maps->maps[0] && maps->res[0] >= 2 && maps->act_level >= 0 && maps->act_level < ACT_LEVELS,
               "vt_query_build_projection: bad argument");
    const long M = (long)B * maps->res[0] * maps->res[0];
    const size_t lds = sizeof(float) * 64 * PJ_AS;
    VT_LDS_LIMIT(proj_gemm_kernel, lds);
    hipLaunchKernelGGL(proj_gemm_kernel, dim3((unsigned)((M + 63) / 64)), dim3(256), lds, vt_stream(stream), maps->maps[0], h->projw, proj, M,
                       ldexpf(1.0f, h->u1_exp - ACT_LEVEL_SHIFT * maps->act_level));
    VT_LAUNCH_CHECK();
    return VT_OK;
}

// clock probe of the fused-objective kernels (measurement only; NULL = off, the default): `counters` = 3 device-side 64-bit words {shader clocks, 100 MHz ticks,
// samples}, added to by every 1024th workgroup of every vt_query_* launch while set
static std::atomic<unsigned long long *> g_clk_probe{nullptr};
extern "C" int vt_query_set_clock_probe(unsigned long long *counters) { g_clk_probe.store(counters, std::memory_order_relaxed); return VT_OK; }

static size_t lds_bytes(int G)
{
    const size_t r0 = (size_t)G * 2048 > (size_t)1152 + G * 1024 ? (size_t)G * 2048 : (size_t)1152 + G * 1024;
    return 16 * (r0 + 256) + sizeof(float) * (64 * 3 + 4 * 64 * 2 + G * 64 + 64 + 64) + 8 * sizeof(double) + 16 + (G == 2 ? sizeof(float) * 2 * 64 * GEO_STRIDE : 0);      // + the geometry ring of the two-head kernels
}

template <int G, int MODE, bool USEP>
static int launch_(const QArgs &a, hipStream_t st)
{
#ifndef LDS_PAD
#define LDS_PAD 0       /* experiment builds only: extra dynamic LDS per workgroup, to lower the workgroups per CU */
#endif
    const size_t lds = lds_bytes(G) + LDS_PAD;
    VT_LDS_LIMIT((query_kernel<G, MODE, USEP>), lds);
    QArgs b = a; b.skip = vt_skip_flag_of(st); b.clk = g_clk_probe.load(std::memory_order_relaxed);
    hipLaunchKernelGGL((query_kernel<G, MODE, USEP>), dim3(((a.N + 63) / 64) * a.B), dim3(256), lds, st, b);
    VT_LAUNCH_CHECK();
    return VT_OK;
}
// the hoisted-projection variant is taken when the maps carry a projection and every head of the launch has columns in it
template <int G, int MODE>
static int launch(const QArgs &a, hipStream_t st)
{
    bool usep = a.proj != nullptr && a.pw == PROJ_COLS && (long)a.res[0] * a.res[0] * PROJ_COLS < (1L << 32);       // 32-bit texel offsets
    for (int g = 0; g < G; g++) usep = usep && a.hw[g].pcol >= 0;
    if (MODE == MODE_HUMAN || MODE == MODE_OBJECT || MODE == MODE_PROJECT) { if (usep) return launch_<G, MODE, true>(a, st); }
    return launch_<G, MODE, false>(a, st);
}

static int fill_common(QArgs &a, const vt_sifnet *h, const vt_maps *maps, const float *pts, const float *cc, const float *bc, int B, int N)
{
    VT_REQUIRE(h && maps && pts && cc && bc && B > 0 && N > 0, "vt_query: null argument or empty batch");
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < 8; i++) { VT_REQUIRE(maps->maps[i] && maps->res[i] >= 2, "vt_query: map %d missing", i); a.maps[i] = maps->maps[i]; a.res[i] = maps->res[i]; }
    a.pts = pts; a.crop_center = cc; a.body_center = bc; a.B = B; a.N = N;
    VT_REQUIRE(maps->act_level >= 0 && maps->act_level < ACT_LEVELS, "vt_query: vt_maps::act_level must be in [0, %d)", ACT_LEVELS);
    // the hoisted projection carries the operand scale of the level it was built for
    const bool proj_ok = maps->proj && maps->proj_level == maps->act_level;
    a.proj = proj_ok ? maps->proj : nullptr; a.pw = proj_ok ? maps->proj_cols : 0;
    a.u1 = ldexpf(1.0f, h->u1_exp - ACT_LEVEL_SHIFT * maps->act_level); a.u1inv = 1.0f / a.u1;
    a.fx = h->cam[0]; a.fy = h->cam[1]; a.cx = h->cam[2]; a.cy = h->cam[3]; a.crop = h->cam[4];
    return VT_OK;
}

extern "C" int vt_query_forward(const vt_sifnet *h, const vt_maps *maps, const float *pts, const float *crop_center, const float *body_center,
                                int B, int N, float *df, float *pca, float *parts, float *centers, float *vis, void *stream)
{
    if (VT_IS_F32(h, maps)) return f32q::forward(h->f32, maps, pts, crop_center, body_center, B, N, df, pca, parts, centers, vis, stream);
    QArgs a; int rc = fill_common(a, h, maps, pts, crop_center, body_center, B, N); if (rc) return rc;
    float *outs[5] = {df, pca, parts, centers, vis};
    int ids[5], n = 0;
    for (int i = 0; i < 5; i++) if (outs[i]) ids[n++] = i;
    VT_REQUIRE(n > 0, "vt_query_forward: no output requested");
    for (int i = 0; i < n; i += 2) {
        const int g = (i + 1 < n) ? 2 : 1;
        for (int k = 0; k < g; k++) { a.hw[k] = head_at(h, ids[i + k], maps->act_level); a.out[k] = outs[ids[i + k]]; }
        rc = (g == 2) ? launch<2, MODE_FWD>(a, vt_stream(stream)) : launch<1, MODE_FWD>(a, vt_stream(stream));
        if (rc) return rc;
    }
    return VT_OK;
}

extern "C" int vt_query_backward(const vt_sifnet *h, const vt_maps *maps, const float *pts, const float *crop_center, const float *body_center,
                                 int B, int N, const float *d_df, const float *d_pca, const float *d_parts, const float *d_centers,
                                 const float *d_vis, float *dpts, void *stream)
{
    if (VT_IS_F32(h, maps)) return f32q::backward(h->f32, maps, pts, crop_center, body_center, B, N, d_df, d_pca, d_parts, d_centers, d_vis, dpts, stream);
    QArgs a; int rc = fill_common(a, h, maps, pts, crop_center, body_center, B, N); if (rc) return rc;
    VT_REQUIRE(dpts, "vt_query_backward: dpts is null");
    const float *gs[5] = {d_df, d_pca, d_parts, d_centers, d_vis};
    int ids[5], n = 0;
    for (int i = 0; i < 5; i++) if (gs[i]) ids[n++] = i;
    if (n == 0) { VT_HIP(hipMemsetAsync(dpts, 0, sizeof(float) * (size_t)B * N * 3, vt_stream(stream))); return VT_OK; }
    VT_REQUIRE(n <= 2, "vt_query_backward: at most two heads with gradients per call (call again and add for more)");
    a.dpts = dpts;
    for (int k = 0; k < n; k++) { a.hw[k] = head_at(h, ids[k], maps->act_level); a.gout[k] = gs[ids[k]]; }
    return n == 2 ? launch<2, MODE_BWD>(a, vt_stream(stream)) : launch<1, MODE_BWD>(a, vt_stream(stream));
}

// which kernel serves vt_query_human_loss when the maps carry a projection: 256 (default, the faster one: DESIGN.md 4.1b) or 512 threads per workgroup
// (the environment switch selects a variant only in the experiments build; the product library holds the 256-thread kernel alone and says so instead of silently
//  ignoring the request -- the same rule as vt_query_set_human_kernel below: ADVICE r05)
static std::atomic<int> g_human_kernel_threads{[]() {
    const char *e = getenv("VT_QUERY_HUMAN_KERNEL"); const int v = e ? atoi(e) : 0;
#ifdef VT_EXPERIMENTS
    return (v == 512 || v == 128) ? v : 256;
#else
    if (v == 512 || v == 128) fprintf(stderr, "libvistracker_hip: VT_QUERY_HUMAN_KERNEL=%d ignored -- this library holds the default kernel (256) only (make -C vistracker_amd/csrc experiments)\n", v);
    return 256;
#endif
}()};
extern "C" int vt_query_set_human_kernel(int threads)
{
#ifdef VT_EXPERIMENTS
    VT_REQUIRE(threads == 256 || threads == 512 || threads == 128, "vt_query_set_human_kernel: 256, 512 (threads per 64-point workgroup) or 128 (points per 512-thread producer / consumer workgroup)");
#else
    VT_REQUIRE(threads == 256, "vt_query_set_human_kernel: this library holds the default kernel (256) only; the measured-negative variants 512 / 128 live in csrc/experiments (make experiments)");
#endif
    g_human_kernel_threads.store(threads, std::memory_order_relaxed);
    return VT_OK;
}

extern "C" int vt_query_human_loss(const vt_sifnet *h, const vt_maps *maps, const float *pts, const float *crop_center, const float *body_center,
                                   int B, int N, const int *labels, const int *order, float w_dfh, float w_part, float *dpts, double *terms, void *stream)
{
    if (VT_IS_F32(h, maps)) return f32q::human_loss(h->f32, maps, pts, crop_center, body_center, B, N, labels, order, w_dfh, w_part, dpts, terms, stream);
    QArgs a; int rc = fill_common(a, h, maps, pts, crop_center, body_center, B, N); if (rc) return rc;
    VT_REQUIRE(labels && dpts && terms, "vt_query_human_loss: null argument");
    a.hw[0] = head_at(h, 0, maps->act_level); a.hw[1] = head_at(h, 2, maps->act_level); a.labels = labels; a.order = order; a.w0 = w_dfh; a.w1 = w_part; a.dpts = dpts; a.terms = terms;
    // vt_query_set_human_kernel(512) / VT_QUERY_HUMAN_KERNEL=512 select the 512-thread kernel (one workgroup per CU, deep tap prefetch): an
    // experiment kept for A/B measurements and as an independent cross-check of the 256-thread kernel (measured slower)
    const int variant = g_human_kernel_threads.load(std::memory_order_relaxed);
    const bool use8 = variant == 512;
    const bool usep = a.proj != nullptr && a.pw == PROJ_COLS && (long)a.res[0] * a.res[0] * PROJ_COLS < (1L << 32);
#ifdef VT_EXPERIMENTS
    if (use8 && usep) return launch_human8(a, vt_stream(stream));
    if (variant == 128 && usep) return launch_human_pc(a, vt_stream(stream));
#endif
    return launch<2, MODE_HUMAN>(a, vt_stream(stream));
}

extern "C" int vt_query_human_step(const vt_sifnet *h, const vt_maps *maps, const float *pts, const float *crop_center, const float *body_center,
                                   int B, int N, const int *labels, const int *order, float w_dfh, float w_part, int accumulate, float w_accel,
                                   double *term_accel, float *dpts, double *terms, void *stream)
{
    VT_REQUIRE(!VT_IS_F32(h, maps), "vt_query_human_step: split-f16 route only (run vt_query_human_loss + vt_accel_loss on the strict-fp32 route)");
    QArgs a; int rc = fill_common(a, h, maps, pts, crop_center, body_center, B, N); if (rc) return rc;
    VT_REQUIRE(labels && dpts && terms && (!term_accel || B >= 3), "vt_query_human_step: null argument, or the acceleration term with fewer than 3 frames");
    a.hw[0] = head_at(h, 0, maps->act_level); a.hw[1] = head_at(h, 2, maps->act_level); a.labels = labels; a.order = order; a.w0 = w_dfh; a.w1 = w_part; a.dpts = dpts; a.terms = terms;
    a.accum = accumulate ? 1 : 0; a.term_accel = term_accel;
    a.accel_gs = term_accel ? 2.f * w_accel / ((float)(B - 2) * (float)(N * 3)) : 0.f;      // vt_accel_loss: d/dv of mean(a^2)
    return launch<2, MODE_HUMAN>(a, vt_stream(stream));
}

extern "C" int vt_query_project_step(const vt_sifnet *h, const vt_maps *maps, const float *pts, const float *crop_center, const float *body_center,
                                     int B, int N, int df_idx, float threshold, float *pts_out, float *df_target, void *stream)
{
    if (VT_IS_F32(h, maps)) return f32q::project_step(h->f32, maps, pts, crop_center, body_center, B, N, df_idx, threshold, pts_out, df_target, stream);
    QArgs a; int rc = fill_common(a, h, maps, pts, crop_center, body_center, B, N); if (rc) return rc;
    VT_REQUIRE(pts_out && (df_idx == 0 || df_idx == 1), "vt_query_project_step: pts_out is null or df_idx not in {0 (human), 1 (object)}");
    a.hw[0] = head_at(h, 0, maps->act_level); a.df_idx = df_idx; a.w0 = threshold; a.pts_out = pts_out; a.dft_out = df_target;
    return launch<1, MODE_PROJECT>(a, vt_stream(stream));
}

extern "C" int vt_query_object_loss(const vt_sifnet *h, const vt_maps *maps, const float *pts, const float *crop_center, const float *body_center,
                                    int B, int N, const float *occ, float w_obj, float *dpts, double *terms, void *stream)
{
    if (VT_IS_F32(h, maps)) return f32q::object_loss(h->f32, maps, pts, crop_center, body_center, B, N, occ, w_obj, dpts, terms, stream);
    QArgs a; int rc = fill_common(a, h, maps, pts, crop_center, body_center, B, N); if (rc) return rc;
    VT_REQUIRE(occ && dpts && terms, "vt_query_object_loss: null argument");
    a.hw[0] = head_at(h, 0, maps->act_level); a.occ = occ; a.w0 = w_obj; a.dpts = dpts; a.terms = terms;
#ifdef VT_EXPERIMENTS
    // 128-point tiles (experiments/query128.h: bit-identical, 5 % SLOWER at N = 3000) with VT_QUERY_OBJECT_TILE=128, when the maps carry the hoisted projection
    static const int tile128 = []() { const char *e = getenv("VT_QUERY_OBJECT_TILE"); return e && atoi(e) == 128; }();
    const bool usep = a.proj != nullptr && a.pw == PROJ_COLS && (long)a.res[0] * a.res[0] * PROJ_COLS < (1L << 32) && a.hw[0].pcol >= 0;
    if (tile128 && usep) return launch_object128(a, vt_stream(stream));
#endif
    return launch<1, MODE_OBJECT>(a, vt_stream(stream));
}
