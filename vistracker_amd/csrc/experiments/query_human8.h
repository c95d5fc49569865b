// EXPERIMENT, measured NEGATIVE (2.61 ms against 1.71 ms; profiles/r02_experiments.md): the 512-thread "thin wave" form of the SMPL-stage query kernel.
// Not part of libvistracker_hip.so: included by query.hip only under -DVT_EXPERIMENTS (make -C vistracker_amd/csrc experiments ->
// tools/bench_scripts/_exp/libvistracker_hip_exp.so, selected with VT_LIB_PATH + vt_query_set_human_kernel(512)).  Uses the helpers of query.hip.
#pragma once
// =====================================================================================================================
// 512-thread variant of the fused SMPL-stage objective (vt_query_human_loss with the hoisted projection).
//
// The 256-thread kernel above runs two workgroups per CU at 256 VGPRs: every wave is at once gatherer, blender and MFMA issuer, and the
// registers that hold taps in flight limit the prefetch distance of the layer-1 loops to ONE chunk -- with a measured vector-memory latency
// of ~1.8 k cycles under load that leaves each chunk iteration waiting for its gather (and the compiler's alias-safety vmcnt(0) after the
// weight-slab DMA of the backward loop exposes the whole latency).  Here ONE workgroup of eight "thin" waves owns a CU:
//   * a wave computes 16 hidden units x 64 points (half the accumulators and weight fragments of before) and gathers ONE (point, 16-byte
//     piece) per chunk (16 VGPRs per chunk in flight): W8_PD chunks of taps are in flight in both layer-1 loops at the same 256-VGPR budget;
//   * the two heads advance stage by stage (both layer-2 GEMMs, one barrier pair, both layer-3 GEMMs, ...): half the barriers, twice the
//     independent MFMA work between them; layer 4 / the objectives run head-parallel (waves 0-3 df, waves 4-7 parts);
//   * the layer-1 backward takes its weights straight from L2 as A fragments, prefetched two chunks ahead in registers (wave = (point half,
//     channel tile, head): each fragment is read by two waves instead of four) -- no LDS slab, no DMA, nothing the compiler serialises;
//   * tap-difference rows go through a two-slot LDS ring with one barrier per chunk, the projection dot products are split between the two
//     waves that hold the same d(hidden-1) fragments, and the per-point gradient partials of the four (tile, head) waves are summed in a fixed
//     order (deterministic).
// Same arithmetic as query_kernel<2, MODE_HUMAN, true> (split-f16 MFMA, fp32 accumulate, identical scales); results agree to round-off of the
// different summation order of the coordinate gradient.
// =====================================================================================================================
#define W8_PD 3
#ifndef W8_ABL
#define W8_ABL 0     /* timing experiments only (wrong results): 1 no MFMA in the L1 loops, 2 no tap loads, 4 no weight loads in the loops, 8 no blend/store, 16 no loop barriers */
#endif
struct Acc4 { f32x4 v[4]; };
struct Taps1 { float4 t[4]; };

__device__ __forceinline__ constexpr int w8_map(int ci) { return ci < 8 ? 0 : (ci < 10 ? 1 : (ci < 13 ? ci - 8 : 5 + (ci - 13) / 2)); }
__device__ __forceinline__ constexpr int w8_coff(int ci) { return ci < 8 ? 32 * ci : (ci < 10 ? 32 * (ci - 8) : (ci < 13 ? 0 : 32 * ((ci - 13) & 1))); }
__device__ __forceinline__ constexpr int w8_chan(int mi) { return mi == 0 ? 256 : (mi == 1 || mi >= 5 ? 64 : 32); }
__device__ __forceinline__ constexpr int w8_proj(int mi) { return mi < 2 ? 0 : (mi < 5 ? mi - 1 : mi - 4); }

// texel byte offsets (o) and tap coefficients (c) of one point in a map of resolution R with C channels, for the 16-byte piece `sub`
// (same arithmetic as taps_geom; NC = 1: bilinear weights x U_1, NC = 2: d/du and d/dv coefficients)
template <int NC, bool WANT_O, bool WANT_C>
__device__ __forceinline__ void tap1_geom(int R, int C, float u, float v, int sub, unsigned (&o)[4], float (&c)[NC][4], float u1 = 0.f)
{
    const float sc = 0.5f * (float)(R - 1);
    float ix = (u + 1.0f) * 0.5f * (float)(R - 1), iy = (v + 1.0f) * 0.5f * (float)(R - 1);
    ix = fminf(fmaxf(ix, -2.0f), (float)(R + 1)); iy = fminf(fmaxf(iy, -2.0f), (float)(R + 1));
    const float fxl = floorf(ix), fyl = floorf(iy);
    const int x0 = (int)fxl, y0 = (int)fyl, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fxl, wy1 = iy - fyl;
    if (WANT_O) {
        const int xc0 = min(max(x0, 0), R - 1), xc1 = min(max(x1, 0), R - 1), yc0 = min(max(y0, 0), R - 1), yc1 = min(max(y1, 0), R - 1);
        const unsigned r0 = (unsigned)(yc0 * R) * (unsigned)C + (unsigned)(sub * 4), r1 = (unsigned)(yc1 * R) * (unsigned)C + (unsigned)(sub * 4);
        o[0] = (r0 + (unsigned)(xc0 * C)) * 4u; o[1] = (r0 + (unsigned)(xc1 * C)) * 4u;
        o[2] = (r1 + (unsigned)(xc0 * C)) * 4u; o[3] = (r1 + (unsigned)(xc1 * C)) * 4u;
    }
    if (WANT_C) {
        const bool bx0 = x0 >= 0 && x0 < R, bx1 = x1 >= 0 && x1 < R, by0 = y0 >= 0 && y0 < R, by1 = y1 >= 0 && y1 < R;
        const bool i0 = bx0 && by0, i1 = bx1 && by0, i2 = bx0 && by1, i3 = bx1 && by1;
        if (NC == 1) {
            const float wx0 = 1.0f - wx1, wy0s = (1.0f - wy1) * u1, wy1s = wy1 * u1;
            c[0][0] = i0 ? wx0 * wy0s : 0.f; c[0][1] = i1 ? wx1 * wy0s : 0.f; c[0][2] = i2 ? wx0 * wy1s : 0.f; c[0][3] = i3 ? wx1 * wy1s : 0.f;
        } else {
            const float sx1 = wx1 * sc, sy1 = wy1 * sc, sx0 = sc - sx1, sy0 = sc - sy1;
            c[0][0] = i0 ? -sy0 : 0.f; c[0][1] = i1 ? sy0 : 0.f; c[0][2] = i2 ? -sy1 : 0.f; c[0][3] = i3 ? sy1 : 0.f;
            c[NC - 1][0] = i0 ? -sx0 : 0.f; c[NC - 1][1] = i1 ? -sx1 : 0.f; c[NC - 1][2] = i2 ? sx0 : 0.f; c[NC - 1][3] = i3 ? sx1 : 0.f;
        }
    }
}
// request the four taps of chunk ci (compile-time after unrolling) for this thread's (point, piece)
__device__ __forceinline__ void tap1_issue(const QArgs &a, int b, int ci, const float (&uv)[4][2], int sub, Taps1 &r)
{
    const int mi = w8_map(ci), co = w8_coff(ci), C = w8_chan(mi), pr = w8_proj(mi), R = a.res[mi];
    unsigned o[4]; float unused[1][4];
    tap1_geom<1, true, false>(R, C, uv[pr][0], uv[pr][1], sub, o, unused);
    const char *__restrict__ fb = reinterpret_cast<const char *>(a.maps[mi] + (size_t)b * R * R * C + co);
#pragma unroll
    for (int k = 0; k < 4; k++) r.t[k] = *reinterpret_cast<const float4 *>(fb + o[k]);
}
#define TAP1SUM_(c_, w_) __builtin_fmaf(r.t[3].c_, (w_)[3], __builtin_fmaf(r.t[2].c_, (w_)[2], __builtin_fmaf(r.t[1].c_, (w_)[1], r.t[0].c_ * (w_)[0])))

__device__ __forceinline__ void acc4_zero(Acc4 &c)
{
#pragma unroll
    for (int p = 0; p < 4; p++) c.v[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
}
// one K32 step of a thin wave for G heads at once: the B fragments (activations, LDS planes) are read once and used by every head
template <int G>
__device__ __forceinline__ void k32_step8(Acc4 (&c)[G], const uint4 (&w)[G][2], const uint4 *Xhi, const uint4 *Xlo, int kb_base, int lane)
{
    const int q = lane >> 4, j = lane & 15;
    h8 xh[4], xl[4];
#pragma unroll
    for (int p = 0; p < 4; p++) { xh[p] = as_h8(Xhi[(kb_base + q) * 64 + 16 * p + j]); xl[p] = as_h8(Xlo[(kb_base + q) * 64 + 16 * p + j]); }
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
        for (int p = 0; p < 4; p++) c[g].v[p] = MFMAH(as_h8(w[g][0]), xh[p], c[g].v[p]);
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
        for (int p = 0; p < 4; p++) c[g].v[p] = MFMAH(as_h8(w[g][0]), xl[p], c[g].v[p]);
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
        for (int p = 0; p < 4; p++) c[g].v[p] = MFMAH(as_h8(w[g][1]), xh[p], c[g].v[p]);
}
// thin-wave forms of relu_pack / mask_pack / planes_store (hidden units 16 wave8 + 4 q + r; mask bit MASK_BIT(p, r))
struct Packed4 { uint2 hi[4], lo[4]; };
__device__ __forceinline__ unsigned relu_pack4(const Acc4 &c, Packed4 &pk, float &rmax)
{
    unsigned m = 0;
#define RELU_PACK4_F(p_)                                                                                        \
    {                                                                                                           \
        const f32x4 x = c.v[p_];                                                                                \
        const unsigned h01 = cvt_pk(x[0], x[1]), h23 = cvt_pk(x[2], x[3]);                                      \
        sign_bits<(p_)>(m, h01, h23);                                                                           \
        relu_split2(x[0], x[1], h01, pk.hi[p_].x, pk.lo[p_].x, rmax); relu_split2(x[2], x[3], h23, pk.hi[p_].y, pk.lo[p_].y, rmax); \
    }
    RELU_PACK4_F(0) RELU_PACK4_F(1) RELU_PACK4_F(2) RELU_PACK4_F(3)
#undef RELU_PACK4_F
    return ~m;
}
__device__ __forceinline__ void mask_pack4(const Acc4 &c, unsigned m, Packed4 &pk)
{
#pragma unroll
    for (int p = 0; p < 4; p++) {
        float y[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            unsigned keep;
            asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(keep) : "v"(m), "n"(MASK_BIT(p, r)));
            y[r] = __uint_as_float(__float_as_uint(c.v[p][r]) & keep);
        }
        split2_nr(y[0], y[1], pk.hi[p].x, pk.lo[p].x); split2_nr(y[2], y[3], pk.hi[p].y, pk.lo[p].y);
    }
}
// thin-wave halves -> split planes [16 kb][64 pt][8 halves]: hidden unit 16 wave8 + 4 q + r -> kb = 2 wave8 + (q >> 1), half (q & 1)
__device__ __forceinline__ void planes_store8(const Packed4 &pk, uint2 *hi8, uint2 *lo8, int wave8, int lane)
{
    const int q = lane >> 4, j = lane & 15;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int idx = (((2 * wave8 + (q >> 1)) * 64 + 16 * p + j) << 1) + (q & 1);
        hi8[idx] = pk.hi[p]; lo8[idx] = pk.lo[p];
    }
}
// T-pack fragments of a thin wave: [K/32 steps][8 waves][hi|lo][64 lanes] (the 256-thread layout [4 waves][2 nt] read as wave8 = 2 wave + nt)
struct WPre8 { uint4 v[4][2]; float4 bias; };
__device__ __forceinline__ void wprefetch8(WPre8 &p, const uint4 *__restrict__ Wp, int wave8, int lane, const float *__restrict__ bias = nullptr)
{
    p.bias = bias ? *reinterpret_cast<const float4 *>(bias + 16 * wave8 + 4 * (lane >> 4)) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int hl = 0; hl < 2; hl++) p.v[s][hl] = Wp[(unsigned)(wave8 * 128 + lane) + (unsigned)(s * 1024 + hl * 64)];
}
template <int G>
__device__ __forceinline__ void gemm128x(Acc4 (&c)[G], const uint4 *Hp, const WPre8 (&p)[G], int lane)
{
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
        for (int pp = 0; pp < 4; pp++) c[g].v[pp] = (f32x4){p[g].bias.x, p[g].bias.y, p[g].bias.z, p[g].bias.w};
#pragma unroll
    for (int s = 0; s < 4; s++) {
        // the heads have separate activation planes: one B-fragment read per head
#pragma unroll
        for (int g = 0; g < G; g++) {
            Acc4 (&cg)[1] = reinterpret_cast<Acc4 (&)[1]>(c[g]);
            const uint4 w1[1][2] = {{p[g].v[s][0], p[g].v[s][1]}};
            k32_step8<1>(cg, w1, Hp + g * 2048, Hp + g * 2048 + 1024, 4 * s, lane);
        }
    }
}

__global__ __launch_bounds__(512, 2) void query_human8_kernel(const QArgs a)
{
    VT_SKIP_RETURN(a.skip);
    constexpr int G = 2, C0 = PROJ_C0;
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    // region 0 (G x 2048 uint4 = 64 KB), time-shared: layer-1 chunk slots {hi [4 kb][64], lo [4 kb][64]} x 2 -> hidden-activation planes per head
    // -> tap-difference ring (2 slots of [2][64][TS] floats) of the layer-1 backward
    uint4 *Hp = lds;
    uint4 *Go = lds + G * 2048;                         // [G] x {hi [2 kb][64], lo [2 kb][64]}
    float *sPt = reinterpret_cast<float *>(Go + G * 256);   // [64][3]
    float *sUV = sPt + 64 * 3;                          // [4][64][2]
    float *sInv = sUV + 4 * 64 * 2;                     // [G][64]
    float *sPart = sInv + G * 64;                       // [4][64][3] gradient partials of the four (channel tile, head) waves of a point half
    int *sIn = reinterpret_cast<int *>(sPart + 4 * 64 * 3);  // [64]
    double *sRed = reinterpret_cast<double *>(sIn + 64);     // [8]
    int *sOvf = reinterpret_cast<int *>(sRed + 8);           // [1]
    float rmax = 0.f;

    const int tid = threadIdx.x, wave8 = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
    const int sub = tid & 7, gpt = tid >> 3;            // gather role: 16-byte piece `sub` of the taps of point gpt
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
        const int pn = a.order ? a.order[n] : n;
        const float *p = a.pts + ((size_t)b * a.N + pn) * 3;
        const float x = p[0], y = p[1], z = p[2];
        float px = a.fx * x / z + a.cx, py = a.fy * y / z + a.cy;
        px = a.crop / 2 + px - a.crop_center[2 * b]; py = a.crop / 2 + py - a.crop_center[2 * b + 1];
        const float nx = 2 * px / a.crop - 1, ny = 2 * py / a.crop - 1;
        sIn[tid] = (pn << 1) | (int)((nx >= -1.0f) && (nx <= 1.0f) && (ny >= -1.0f) && (ny <= 1.0f));
        const float c0 = x - a.body_center[3 * b], c1 = y - a.body_center[3 * b + 1], c2 = z - a.body_center[3 * b + 2];
        sPt[tid * 3] = x; sPt[tid * 3 + 1] = y; sPt[tid * 3 + 2] = z;
        sUV[(0 * 64 + tid) * 2] = nx;  sUV[(0 * 64 + tid) * 2 + 1] = ny;
        sUV[(1 * 64 + tid) * 2] = c2;  sUV[(1 * 64 + tid) * 2 + 1] = c1;
        sUV[(2 * 64 + tid) * 2] = -c0; sUV[(2 * 64 + tid) * 2 + 1] = c1;
        sUV[(3 * 64 + tid) * 2] = c0;  sUV[(3 * 64 + tid) * 2 + 1] = -c2;
    }
    __syncthreads();
    float uv[4][2];                                     // the four projections of this thread's gather point, in registers for both layer-1 loops
#pragma unroll
    for (int pr = 0; pr < 4; pr++) { uv[pr][0] = sUV[(pr * 64 + gpt) * 2]; uv[pr][1] = sUV[(pr * 64 + gpt) * 2 + 1]; }

    PCLK(0);
    // ================= layer 1, forward =================
    Taps1 tp[W8_PD];
#pragma unroll
    for (int k = 0; k < W8_PD; k++) tap1_issue(a, b, C0 + k, uv, sub, tp[k]);
    uint4 wf[2][G][2];                                  // weight fragments of two consecutive K32 steps
    const unsigned wvo = (unsigned)(wave8 * 128 + lane);
#define W8_LOAD_W1(slot_, step_)                                                                                     \
    _Pragma("unroll") for (int g = 0; g < G; g++)                                                                    \
        _Pragma("unroll") for (int hl = 0; hl < 2; hl++) wf[slot_][g][hl] = a.hw[g].w1p[(size_t)(step_) * 1024 + wvo + hl * 64];
    W8_LOAD_W1(0, C0)
    W8_LOAD_W1(1, C0 + 1)
    Acc4 acc1[G];
    {   // im_feat part of the pre-activations from the hoisted projection, straight in the D-fragment layout: lane (q, j) holds hidden units
        // 16 wave8 + 4 q .. +3 (one float4 of a P row) of the points 16 p + j; all 32 tap rows are requested before the first blend
        const int R = a.res[0];
        const float *__restrict__ Pb = a.proj + (size_t)b * R * R * a.pw;
        float4 t[4][G][4]; float w[4][4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            unsigned o[4]; float unused[4];
            proj_geom(sUV, 16 * p + j, R, a.pw, o, w[p], unused, false);
#pragma unroll
            for (int g = 0; g < G; g++) {
                const unsigned col = (unsigned)(a.hw[g].pcol + 16 * wave8 + 4 * q);
#pragma unroll
                for (int k = 0; k < 4; k++) t[p][g][k] = *reinterpret_cast<const float4 *>(Pb + o[k] + col);
            }
        }
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int g = 0; g < G; g++) {
                const float4 nw = t[p][g][0], ne = t[p][g][1], sw = t[p][g][2], se = t[p][g][3];
                const float *wq = w[p];
                acc1[g].v[p] = (f32x4){TAPSUM_(x, wq), TAPSUM_(y, wq), TAPSUM_(z, wq), TAPSUM_(w, wq)};
            }
    }
    // blend / split / store the features of chunk ci_ (taps in tp[(ci_ - C0) % W8_PD]) into chunk slot ci_ & 1
#define W8_STORE_FEAT(ci_)                                                                                           \
    {                                                                                                                \
        const int mi_ = w8_map(ci_);                                                                                 \
        unsigned o_[4]; float c_[1][4];                                                                              \
        tap1_geom<1, false, true>(a.res[mi_], w8_chan(mi_), uv[w8_proj(mi_)][0], uv[w8_proj(mi_)][1], sub, o_, c_, a.u1); \
        const Taps1 &r = tp[((ci_) - C0) % W8_PD];                                                                   \
        uint2 hi_, lo_;                                                                                              \
        split4(TAP1SUM_(x, c_[0]), TAP1SUM_(y, c_[0]), TAP1SUM_(z, c_[0]), TAP1SUM_(w, c_[0]), hi_, lo_, rmax);            \
        uint2 *hi8_ = reinterpret_cast<uint2 *>(lds + ((ci_) & 1) * 512), *lo8_ = reinterpret_cast<uint2 *>(lds + ((ci_) & 1) * 512 + 256); \
        const int idx_ = (((sub >> 1) * 64 + gpt) << 1) + (sub & 1);                                                 \
        hi8_[idx_] = hi_; lo8_[idx_] = lo_;                                                                          \
    }
    W8_STORE_FEAT(C0)
    tap1_issue(a, b, C0 + W8_PD, uv, sub, tp[0]);
    PCLK(1);
#pragma unroll
    for (int ci = C0; ci < NCHUNK; ci++) {
        const uint4 *buf = lds + (ci & 1) * 512;
        if (!(W8_ABL & 16)) __syncthreads();            // chunk ci visible; the other slot's readers (MFMAs of chunk ci - 1) are done
        if (!(W8_ABL & 1)) k32_step8<G>(acc1, wf[(ci - C0) & 1], buf, buf + 256, 0, lane);
        if (!(W8_ABL & 8)) if (ci + 1 < NCHUNK) W8_STORE_FEAT(ci + 1)
        if (!(W8_ABL & 4)) if (ci + 2 <= NCHUNK) W8_LOAD_W1((ci - C0) & 1, ci + 2)     // steps C0 + 2 .. 19 (19 = the xyz step)
        if (!(W8_ABL & 2)) if (ci + 1 + W8_PD < NCHUNK) tap1_issue(a, b, ci + 1 + W8_PD, uv, sub, tp[(ci + 1 - C0) % W8_PD]);
    }
    {   // z_feat = (x, y, z - 2.2): internal channels 608..610 (K32 step 19, k = 8 q + t: only q == 0, t < 3 are non-zero)
        constexpr int sl = (NCHUNK - C0) & 1;           // slot that holds step 19
#pragma unroll
        for (int p = 0; p < 4; p++) {
            uint2 hi = make_uint2(0u, 0u), lo = make_uint2(0u, 0u);
            if (q == 0) {
                const float *pp = sPt + (16 * p + j) * 3;
                split4(pp[0] * a.u1, pp[1] * a.u1, (pp[2] - 2.2f) * a.u1, a.u1, hi, lo, rmax);       // 4th channel = the constant one (weight s_1 b_1)
            }
            const h8 xh = as_h8(make_uint4(hi.x, hi.y, 0u, 0u)), xl = as_h8(make_uint4(lo.x, lo.y, 0u, 0u));
#pragma unroll
            for (int g = 0; g < G; g++) {
                acc1[g].v[p] = MFMAH(as_h8(wf[sl][g][0]), xh, acc1[g].v[p]);
                acc1[g].v[p] = MFMAH(as_h8(wf[sl][g][0]), xl, acc1[g].v[p]);
                acc1[g].v[p] = MFMAH(as_h8(wf[sl][g][1]), xh, acc1[g].v[p]);
            }
        }
    }
#undef W8_LOAD_W1
#undef W8_STORE_FEAT
    PCLK(2);
    WPre8 wp[G];
#pragma unroll
    for (int g = 0; g < G; g++) wprefetch8(wp[g], a.hw[g].w2p, wave8, lane, a.hw[g].b2);
    __syncthreads();        // region 0 changes role: chunk slots -> hidden-activation planes
    unsigned m1[G], m2[G], m3[G];
    Packed4 pk[G];
#define W8_PLANES(g_) reinterpret_cast<uint2 *>(Hp + (g_) * 2048), reinterpret_cast<uint2 *>(Hp + (g_) * 2048 + 1024)
#pragma unroll
    for (int g = 0; g < G; g++) {
        m1[g] = relu_pack4(acc1[g], pk[g], rmax);
        planes_store8(pk[g], W8_PLANES(g), wave8, lane); OVF_PUBLISH();
    }
    __syncthreads();

    PCLK(3);
    // ================= layers 2, 3 (both heads per stage) =================
    Acc4 c[G];
    gemm128x<G>(c, Hp, wp, lane);
#pragma unroll
    for (int g = 0; g < G; g++) { wprefetch8(wp[g], a.hw[g].w3p, wave8, lane, a.hw[g].b3); m2[g] = relu_pack4(c[g], pk[g], rmax); }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < G; g++) planes_store8(pk[g], W8_PLANES(g), wave8, lane); OVF_PUBLISH();
    __syncthreads();
    gemm128x<G>(c, Hp, wp, lane);
#pragma unroll
    for (int g = 0; g < G; g++) m3[g] = relu_pack4(c[g], pk[g], rmax);
    __syncthreads();
#pragma unroll
    for (int g = 0; g < G; g++) planes_store8(pk[g], W8_PLANES(g), wave8, lane); OVF_PUBLISH();
    __syncthreads();

    PCLK(4);
    // ================= layer 4 + objective, head-parallel: waves 0-3 head df, waves 4-7 head parts; wave (g4, w4) owns points 16 w4 .. +15 =================
    double loss_acc = 0.0;
    {
        const int g4 = wave8 >> 2, w4 = wave8 & 3;
        const HeadW &hw = a.hw[g4];
        const uint4 *Hhi = Hp + g4 * 2048, *Hlo = Hhi + 1024;
        f32x4 o4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const h8 xh = as_h8(Hhi[(4 * s + q) * 64 + 16 * w4 + j]), xl = as_h8(Hlo[(4 * s + q) * 64 + 16 * w4 + j]);
            const h8 wh = as_h8(hw.w4p[(s * 2 + 0) * 64 + lane]), wl = as_h8(hw.w4p[(s * 2 + 1) * 64 + lane]);
            o4 = MFMAH(xh, wh, o4); o4 = MFMAH(xl, wh, o4); o4 = MFMAH(xh, wl, o4);
        }
        const float bias4 = hw.b4[j];
        float go[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int pt = w4 * 16 + q * 4 + r, n = n0 + pt;
            const bool valid = n < a.N, live = j < hw.kout;
            const bool inimg = (sIn[pt] & 1) != 0;
            const int pn = sIn[pt] >> 1;
            float val = o4[r] * hw.cout + bias4;
            if (*sOvf) val = __builtin_nanf("");
            go[r] = 0.f;
            if (g4 == 0) {
                // df_h = clamp(df[:,0], max=.1).mean()  (recon_fit_base.py:640-647); df[~in_img] = 5 (chore_triplane.py:156-159)
                if (j == 0 && valid) {
                    const float d = inimg ? val : OUT_DIST;
                    loss_acc += (double)fminf(d, 0.1f);
                    if (inimg && d <= 0.1f) go[r] = a.w0 / ((float)a.B * (float)a.N);
                }
            } else {
                // part = mean_B sum_N CE(parts, labels)  (recon_fit_behave.py:486)
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
                    if (j == lab) loss_acc += (double)(logf(se) - (val - mx));
                }
            }
        }
        // per-point normalisation of the upstream gradient (see query_kernel)
        _Float16 *gh = reinterpret_cast<_Float16 *>(Go + g4 * 256), *gl = reinterpret_cast<_Float16 *>(Go + g4 * 256 + 128);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float m = fabsf(go[r]);
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            const int eb = (int)((__float_as_uint(m) >> 23) & 255u);
            const int ge = hw.goexp;
            const bool ok = eb >= ge + 2 && eb >= 2 && eb < 255 && eb - ge < 254;
            const float s = ok ? __uint_as_float((unsigned)(254 + ge - eb) << 23) : 1.0f;
            const float inv = ok ? __uint_as_float((unsigned)(eb - ge) << 23) : 1.0f;
            const int pt = w4 * 16 + q * 4 + r;
            if (j == 0) sInv[g4 * 64 + pt] = inv;
            const float x = go[r] * s;
            const _Float16 hi = (_Float16)x, lo = (_Float16)(x - (float)hi);
            gh[((j >> 3) * 64 + pt) * 8 + (j & 7)] = hi; gl[((j >> 3) * 64 + pt) * 8 + (j & 7)] = lo;
        }
    }
    PCLK(5);
    // ================= backward through layers 4, 3, 2 (both heads per stage) =================
    WPre8 wq[G];
    uint4 w4t[G][2];
#pragma unroll
    for (int g = 0; g < G; g++) {
#pragma unroll
        for (int hl = 0; hl < 2; hl++) w4t[g][hl] = a.hw[g].w4tp[((size_t)wave8 * 2 + hl) * 64 + lane];
        wprefetch8(wq[g], a.hw[g].w3tp, wave8, lane);
    }
    __syncthreads();                                    // Go visible; the planes (h3) are no longer read
#pragma unroll
    for (int g = 0; g < G; g++) {
        acc4_zero(c[g]);
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const uint4 z = make_uint4(0u, 0u, 0u, 0u);
            const h8 xh = as_h8(q < 2 ? Go[g * 256 + q * 64 + 16 * p + j] : z), xl = as_h8(q < 2 ? Go[g * 256 + 128 + q * 64 + 16 * p + j] : z);
            c[g].v[p] = MFMAH(as_h8(w4t[g][0]), xh, c[g].v[p]);
            c[g].v[p] = MFMAH(as_h8(w4t[g][0]), xl, c[g].v[p]);
            c[g].v[p] = MFMAH(as_h8(w4t[g][1]), xh, c[g].v[p]);
        }
        mask_pack4(c[g], m3[g], pk[g]);
        planes_store8(pk[g], W8_PLANES(g), wave8, lane);
    }
    __syncthreads();
    gemm128x<G>(c, Hp, wq, lane);                       // g2 = W3^T . g3
#pragma unroll
    for (int g = 0; g < G; g++) { wprefetch8(wq[g], a.hw[g].w2tp, wave8, lane); mask_pack4(c[g], m2[g], pk[g]); }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < G; g++) planes_store8(pk[g], W8_PLANES(g), wave8, lane);
    __syncthreads();
    gemm128x<G>(c, Hp, wq, lane);                       // g1 = W2^T . g2
#pragma unroll
    for (int g = 0; g < G; g++) mask_pack4(c[g], m1[g], pk[g]);
    __syncthreads();
#pragma unroll
    for (int g = 0; g < G; g++) planes_store8(pk[g], W8_PLANES(g), wave8, lane);
#undef W8_PLANES
    PCLK(6);
    {   // block-reduce the loss partials (waves 0-3 hold the df term, waves 4-7 the part term)
        double s = loss_acc;
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) sRed[wave8] = s;
    }
    __syncthreads();                                    // planes = d loss' / d(pre-activation 1) of both heads; sRed complete
    if (tid < 2) {
        double s = sRed[tid * 4] + sRed[tid * 4 + 1] + sRed[tid * 4 + 2] + sRed[tid * 4 + 3];
        s = tid == 0 ? s / ((double)a.B * a.N) : s / (double)a.B;
        if (*sOvf) s = (double)__builtin_nanf("");
        atomicAdd(a.terms + tid, s);
    }

    // ================= layer 1 + gathers, backward =================
    // wave = (point half ph, channel tile ct, head gb): d feat[16 ct + ..][pts of tiles 2 ph, 2 ph + 1] of head gb per chunk
    const int gb = wave8 & 1, ct = (wave8 >> 1) & 1, ph = wave8 >> 2;
    uint4 dh[2][4][2];      // B fragments of d(hidden-1) of head gb: point 16 (2 ph + t) + j, hidden units 32 s + 8 q + ..
    float kscale[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            dh[t][s][0] = Hp[gb * 2048 + (4 * s + q) * 64 + 16 * (2 * ph + t) + j];
            dh[t][s][1] = Hp[gb * 2048 + 1024 + (4 * s + q) * 64 + 16 * (2 * ph + t) + j];
        }
        kscale[t] = sInv[gb * 64 + 16 * (2 * ph + t) + j] * a.hw[gb].kback;
    }
    float ptx[2], pty[2], piz[2];
#pragma unroll
    for (int t = 0; t < 2; t++) { const int pt = 16 * (2 * ph + t) + j; ptx[t] = sPt[pt * 3]; pty[t] = sPt[pt * 3 + 1]; piz[t] = 1.0f / sPt[pt * 3 + 2]; }
    __syncthreads();        // region 0 changes role again: activation planes -> tap-difference ring
    PCLK(7);
#pragma unroll
    for (int k = 0; k < W8_PD; k++) tap1_issue(a, b, C0 + k, uv, sub, tp[k]);
    uint4 wb[2][4][2];      // A fragments of W1 (chunk, s, hi|lo) for (ct, gb), two chunks
    const uint4 *__restrict__ w1c = a.hw[gb].w1c;
#define W8_LOAD_WB(slot_, ci_)                                                                                       \
    _Pragma("unroll") for (int s = 0; s < 4; s++)                                                                    \
        _Pragma("unroll") for (int hl = 0; hl < 2; hl++) wb[slot_][s][hl] = w1c[(size_t)(ci_) * 1024 + ((s * 2 + ct) * 2 + hl) * 64 + lane];
    W8_LOAD_WB(0, C0)
    W8_LOAD_WB(1, C0 + 1)
    const float kx = 2.0f / a.crop * a.fx, ky = 2.0f / a.crop * a.fy;
    float gx[2] = {0.f, 0.f}, gy[2] = {0.f, 0.f}, gz[2] = {0.f, 0.f};
    {   // im_feat part: d/du = sum_t cu_t <d(hidden-1), P row of tap t>; the two channel-tile waves hold the same fragments and share the work:
        // wave ct takes point tile t = ct.  Two rounds of 16 tap-row loads.
        const int R = a.res[0];
        const float *__restrict__ Pb = a.proj + (size_t)b * R * R * a.pw;
        unsigned o[4]; float cu[4], cv[4];
        proj_geom(sUV, 16 * (2 * ph + ct) + j, R, a.pw, o, cu, cv, true);
        float dg[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < 4; s2 += 2) {
            float4 pr[2][4][2];
#pragma unroll
            for (int ss = 0; ss < 2; ss++) {
                const unsigned col = (unsigned)(a.hw[gb].pcol + 32 * (s2 + ss) + 8 * q);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    pr[ss][k][0] = *reinterpret_cast<const float4 *>(Pb + o[k] + col); pr[ss][k][1] = *reinterpret_cast<const float4 *>(Pb + o[k] + col + 4);
                }
            }
#pragma unroll
            for (int ss = 0; ss < 2; ss++) {
                // dh[ct] with a compile-time index: select between the two tiles' fragments (wave-uniform)
                const uint4 fh = ct ? dh[1][s2 + ss][0] : dh[0][s2 + ss][0], fl = ct ? dh[1][s2 + ss][1] : dh[0][s2 + ss][1];
                const h8 hh = as_h8(fh), hl = as_h8(fl);
                float x[8];
#pragma unroll
                for (int t = 0; t < 8; t++) x[t] = (float)hh[t] + (float)hl[t];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const float4 p0 = pr[ss][k][0], p1 = pr[ss][k][1];
                    dg[k] = __builtin_fmaf(x[7], p1.w, __builtin_fmaf(x[6], p1.z, __builtin_fmaf(x[5], p1.y, __builtin_fmaf(x[4], p1.x,
                            __builtin_fmaf(x[3], p0.w, __builtin_fmaf(x[2], p0.z, __builtin_fmaf(x[1], p0.y, __builtin_fmaf(x[0], p0.x, dg[k]))))))));
                }
            }
        }
        const float ks = (ct ? kscale[1] : kscale[0]) * a.u1inv;
        const float su = ks * (cu[0] * dg[0] + cu[1] * dg[1] + cu[2] * dg[2] + cu[3] * dg[3]);
        const float sv = ks * (cv[0] * dg[0] + cv[1] * dg[1] + cv[2] * dg[2] + cv[3] * dg[3]);
        const float px = ct ? ptx[1] : ptx[0], py = ct ? pty[1] : pty[0], iz = ct ? piz[1] : piz[0];
        const float ax = su * (kx * iz), ay = sv * (ky * iz), az = __builtin_fmaf(sv, -ky * py * iz * iz, su * (-kx * px * iz * iz));
        gx[0] = ct ? 0.f : ax; gy[0] = ct ? 0.f : ay; gz[0] = ct ? 0.f : az;
        gx[1] = ct ? ax : 0.f; gy[1] = ct ? ay : 0.f; gz[1] = ct ? az : 0.f;
    }
    float *ring = reinterpret_cast<float *>(lds);       // slot s: bu = ring + s * 2 * 64 * TS, bv = bu + 64 * TS
    // tap differences d feat / d u, d feat / d v of chunk ci_ -> ring slot ci_ & 1 (this thread: point gpt, channels 4 sub .. +3)
#define W8_STORE_GRAD(ci_)                                                                                           \
    {                                                                                                                \
        const int mi_ = w8_map(ci_);                                                                                 \
        unsigned o_[4]; float c_[2][4];                                                                              \
        tap1_geom<2, false, true>(a.res[mi_], w8_chan(mi_), uv[w8_proj(mi_)][0], uv[w8_proj(mi_)][1], sub, o_, c_);  \
        const Taps1 &r = tp[((ci_) - C0) % W8_PD];                                                                   \
        float *bu_ = ring + ((ci_) & 1) * 2 * 64 * TS, *bv_ = bu_ + 64 * TS;                                         \
        *reinterpret_cast<float4 *>(bu_ + gpt * TS + sub * 4) = make_float4(TAP1SUM_(x, c_[0]), TAP1SUM_(y, c_[0]), TAP1SUM_(z, c_[0]), TAP1SUM_(w, c_[0])); \
        *reinterpret_cast<float4 *>(bv_ + gpt * TS + sub * 4) = make_float4(TAP1SUM_(x, c_[1]), TAP1SUM_(y, c_[1]), TAP1SUM_(z, c_[1]), TAP1SUM_(w, c_[1])); \
    }
    W8_STORE_GRAD(C0)
    tap1_issue(a, b, C0 + W8_PD, uv, sub, tp[0]);
    PCLK(8);
#pragma unroll
    for (int ci = C0; ci < NCHUNK; ci++) {
        const int sl = (ci - C0) & 1;
        if (!(W8_ABL & 16)) __syncthreads();            // tap differences of chunk ci visible; the other slot's readers are done
        f32x4 dd[2];
        dd[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; dd[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < ((W8_ABL & 1) ? 0 : 4); s++) {
            const h8 wh = as_h8(wb[sl][s][0]), wl = as_h8(wb[sl][s][1]);
            dd[0] = MFMAH(wh, as_h8(dh[0][s][0]), dd[0]); dd[1] = MFMAH(wh, as_h8(dh[1][s][0]), dd[1]);
            dd[0] = MFMAH(wh, as_h8(dh[0][s][1]), dd[0]); dd[1] = MFMAH(wh, as_h8(dh[1][s][1]), dd[1]);
            dd[0] = MFMAH(wl, as_h8(dh[0][s][0]), dd[0]); dd[1] = MFMAH(wl, as_h8(dh[1][s][0]), dd[1]);
        }
        const float *bu = ring + (ci & 1) * 2 * 64 * TS, *bv = bu + 64 * TS;
        float4 u4[2], v4[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            u4[t] = *reinterpret_cast<const float4 *>(bu + (16 * (2 * ph + t) + j) * TS + 16 * ct + 4 * q);
            v4[t] = *reinterpret_cast<const float4 *>(bv + (16 * (2 * ph + t) + j) * TS + 16 * ct + 4 * q);
        }
        if (!(W8_ABL & 8)) if (ci + 1 < NCHUNK) W8_STORE_GRAD(ci + 1)
        if (!(W8_ABL & 4)) if (ci + 2 <= NCHUNK) W8_LOAD_WB(sl, ci + 2)    // chunk 19 = the xyz rows
        if (!(W8_ABL & 2)) if (ci + 1 + W8_PD < NCHUNK) tap1_issue(a, b, ci + 1 + W8_PD, uv, sub, tp[(ci + 1 - C0) % W8_PD]);
        // projection Jacobians of the chunk's map (compile-time after unrolling)
        const int pr = w8_proj(w8_map(ci));
#pragma unroll
        for (int t = 0; t < 2; t++) {
            float d[4];
#pragma unroll
            for (int r = 0; r < 4; r++) d[r] = dd[t][r] * kscale[t];
            const float su = __builtin_fmaf(d[3], u4[t].w, __builtin_fmaf(d[2], u4[t].z, __builtin_fmaf(d[1], u4[t].y, d[0] * u4[t].x)));
            const float sv = __builtin_fmaf(d[3], v4[t].w, __builtin_fmaf(d[2], v4[t].z, __builtin_fmaf(d[1], v4[t].y, d[0] * v4[t].x)));
            const float j0x = kx * piz[t], j0y = ky * piz[t], j0zu = -kx * ptx[t] * piz[t] * piz[t], j0zv = -ky * pty[t] * piz[t] * piz[t];
            const float cxu = pr == 0 ? j0x : (pr == 2 ? -1.f : (pr == 3 ? 1.f : 0.f));
            const float cyv = pr == 0 ? j0y : (pr == 3 ? 0.f : 1.f);
            const float czu = pr == 0 ? j0zu : (pr == 1 ? 1.f : 0.f);
            const float czv = pr == 0 ? j0zv : (pr == 3 ? -1.f : 0.f);
            gx[t] = __builtin_fmaf(su, cxu, gx[t]); gy[t] = __builtin_fmaf(sv, cyv, gy[t]); gz[t] = __builtin_fmaf(sv, czv, __builtin_fmaf(su, czu, gz[t]));
        }
    }
#undef W8_STORE_GRAD
#undef W8_LOAD_WB
    PCLK(9);
    if (ct == 0) {   // direct xyz features: rows 0..2 of channel tile 0 of "chunk" 19 (its fragments were the last W8_LOAD_WB)
        constexpr int sl = (NCHUNK - C0) & 1;
#pragma unroll
        for (int t = 0; t < 2; t++) {
            f32x4 dz = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const h8 wh = as_h8(wb[sl][s][0]), wl = as_h8(wb[sl][s][1]);
                dz = MFMAH(wh, as_h8(dh[t][s][0]), dz); dz = MFMAH(wh, as_h8(dh[t][s][1]), dz); dz = MFMAH(wl, as_h8(dh[t][s][0]), dz);
            }
            if (q == 0) { gx[t] += dz[0] * kscale[t]; gy[t] += dz[1] * kscale[t]; gz[t] += dz[2] * kscale[t]; }
        }
    }
    // reduce the channel partials over the 4 lane groups q, then over the four (ct, gb) waves of the point half in a FIXED order
#pragma unroll
    for (int t = 0; t < 2; t++) {
        gx[t] += __shfl_xor(gx[t], 16, 64); gy[t] += __shfl_xor(gy[t], 16, 64); gz[t] += __shfl_xor(gz[t], 16, 64);
        gx[t] += __shfl_xor(gx[t], 32, 64); gy[t] += __shfl_xor(gy[t], 32, 64); gz[t] += __shfl_xor(gz[t], 32, 64);
        if (q == 0) {
            float *sp = sPart + ((wave8 & 3) * 64 + 16 * (2 * ph + t) + j) * 3;
            sp[0] = gx[t]; sp[1] = gy[t]; sp[2] = gz[t];
        }
    }
    __syncthreads();
    if (tid < 64 * 3) {
        const int pt = tid / 3, k = tid - 3 * pt, n = n0 + pt;
        if (n < a.N) {
            float s = ((sPart[(0 * 64 + pt) * 3 + k] + sPart[(1 * 64 + pt) * 3 + k]) + sPart[(2 * 64 + pt) * 3 + k]) + sPart[(3 * 64 + pt) * 3 + k];
            if (*sOvf) s = __builtin_nanf("");
            a.dpts[((size_t)b * a.N + (sIn[pt] >> 1)) * 3 + k] = s;
        }
    }
    PCLK(10);
}
static size_t lds_bytes_human8() { return 16 * (2 * 2048 + 2 * 256) + sizeof(float) * (64 * 3 + 4 * 64 * 2 + 2 * 64 + 4 * 64 * 3 + 64) + 8 * sizeof(double) + 8; }
static int launch_human8(const QArgs &a, hipStream_t st)
{
    const size_t lds = lds_bytes_human8();
    VT_LDS_LIMIT(query_human8_kernel, lds);
    QArgs b8 = a; b8.skip = vt_skip_flag_of(st);
    hipLaunchKernelGGL(query_human8_kernel, dim3(((a.N + 63) / 64) * a.B), dim3(512), lds, st, b8);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

