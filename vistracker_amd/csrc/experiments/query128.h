// query128.h -- the fused OBJECT objective (vt_query_object_loss, MODE_OBJECT with the hoisted projection) on 128-POINT tiles.
//
// The one-head query kernel is bound by the bytes its CU pulls through the vector-memory path (two workgroups of 256 registers deliver what
// three of 168 do, profiles/r05_g1wide_ab.txt), and 39 % of them are decoder weights: every byte of weights is fetched once per WORKGROUP.  Here a workgroup
// owns two 64-point halves of one frame and every weight fragment serves both:
//   * layer-1 loops run over (chunk, half) pairs -- the same iteration as query_kernel<1, ...> (one 64-point feature buffer / d(feature) buffer at a time, one
//     set of taps in flight), with the weight fragments (forward) and the weight slab (backward) fetched once per CHUNK instead of once per iteration;
//   * the hidden layers leapfrog over the two halves exactly like the two heads of query_kernel<2, MODE_HUMAN> (gemm128_epi: GEMM of one half with the other
//     half's epilogue between its MFMAs, one barrier per GEMM) -- and the second GEMM of a pair reuses the weight fragments the first one loaded.
// Per point the arithmetic is query_kernel<1, MODE_OBJECT, true>'s in the same order: bit-identical gradients and bit-identical fitted rows over a whole
// bench run (profiles/r05_object128_bench_ab.txt).  Texture-path bytes per 64 points: 1568 KB -> 1264 KB (-19 %).  256 threads, 2 workgroups per CU (79 KB
// LDS, 250 registers, no spills).
// EXPERIMENT, measured NEGATIVE (round 5): 0.4482 ms against 0.4271 ms at the bench shape (B = 96, N = 3000; part of it is quantisation: 24 tiles of 128 per
// frame = 2304 workgroups on 512 slots = 4.5 rounds, the last tile 44 % empty).  A fifth fewer bytes through the vector-memory path do not make the kernel
// faster: the byte count of that path is not what bounds it.  Built with -DVT_EXPERIMENTS only (VT_QUERY_OBJECT_TILE=128).
#pragma once

static size_t lds_bytes_128() { return 16 * (4096 + 512) + sizeof(float) * (128 * 3 + 2 * 4 * 64 * 2 + 128) + sizeof(int) * 128 + 8 * sizeof(double) + 16; }

template <int MODE>
__global__ __launch_bounds__(256, 2) void query128_kernel(const QArgs a)
{
    static_assert(MODE == MODE_OBJECT, "query128_kernel: the object objective only");
    constexpr int C0 = PROJ_C0, NIT = 2 * (NCHUNK - PROJ_C0);
    VT_SKIP_RETURN(a.skip);
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    // region 0 (64 KB), time-shared: staged projection rows of one half -> feature-chunk double buffer (16 KB) -> activation planes of the two halves ->
    // d(feature) rows of one half + weight slab
    uint4 *Hp = lds;                                        // [2 halves]{hi [16 kb][64], lo [16 kb][64]}
    uint4 *Go = lds + 4096;                                 // [2 halves]{hi [2 kb][64], lo [2 kb][64]}
    float *sPt = reinterpret_cast<float *>(Go + 512);       // [128][3]
    float *sUV = sPt + 128 * 3;                             // [2 halves][4 projections][64][2]
    float *sInv = sUV + 2 * 4 * 64 * 2;                     // [128]
    int *sIn = reinterpret_cast<int *>(sInv + 128);         // [128]
    double *sRed = reinterpret_cast<double *>(sIn + 128);   // [8]
    int *sOvf = reinterpret_cast<int *>(sRed + 8);          // [1]
    float rmax = 0.f;
    const HeadW &hw = a.hw[0];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
    int b, tile;
    {
        const int tiles = (a.N + 127) >> 7, L = blockIdx.x;
        if ((a.B & 7) == 0) { const int slot = L >> 3; b = (L & 7) + 8 * (slot / tiles); tile = slot % tiles; }
        else { b = L / tiles; tile = L % tiles; }
        b = __builtin_amdgcn_readfirstlane(b); tile = __builtin_amdgcn_readfirstlane(tile);
    }
    const int n0 = tile * 128;

    // ---- per-point projections (camera.py:52-90, chore_triplane.py:207-251)
    if (tid == 0) *sOvf = 0;
    if (tid < 128) {
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
        float *uv = sUV + (tid >> 6) * 512; const int lp = tid & 63;
        uv[(0 * 64 + lp) * 2] = nx;  uv[(0 * 64 + lp) * 2 + 1] = ny;   // perspective
        uv[(1 * 64 + lp) * 2] = c2;  uv[(1 * 64 + lp) * 2 + 1] = c1;   // right
        uv[(2 * 64 + lp) * 2] = -c0; uv[(2 * 64 + lp) * 2 + 1] = c1;   // back
        uv[(3 * 64 + lp) * 2] = c0;  uv[(3 * 64 + lp) * 2 + 1] = -c2;  // top
    }
    __syncthreads();

    // ---- layer 1, im_feat part: blend of the four tap rows of the hoisted projection, half by half (query_kernel's USEP block)
    Acc8 acc1[2];
    const int R0_ = a.res[0];
    const rsrc_t Pb = make_rsrc(a.proj + (size_t)b * R0_ * R0_ * a.pw, (unsigned)(R0_ * R0_ * a.pw) * 4u);
    {
        constexpr int PS = 128 + 4;
        float *stage = reinterpret_cast<float *>(lds);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            {
                const int spt = tid >> 2, seg = tid & 3;
                unsigned o[4]; float w[4], unused[4];
                proj_geom(sUV + h * 512, spt, R0_, a.pw, o, w, unused, false);
#pragma unroll
                for (int k = 0; k < 4; k++) o[k] = (o[k] + 4u * seg) * 4u;
                float4 t[8][4];
                const unsigned pc = (unsigned)hw.pcol * 4u;
#pragma unroll
                for (int i = 0; i < 8; i++)
#pragma unroll
                    for (int k = 0; k < 4; k++) t[i][k] = GATHER_P4(Pb, o[k] + 64u * i, pc);
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float4 nw = t[i][0], ne = t[i][1], sw = t[i][2], se = t[i][3];
                    *reinterpret_cast<float4 *>(stage + spt * PS + 16 * i + 4 * seg) = make_float4(TAPSUM_(x, w), TAPSUM_(y, w), TAPSUM_(z, w), TAPSUM_(w, w));
                }
            }
            __syncthreads();
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const float4 v = *reinterpret_cast<const float4 *>(stage + (16 * p + j) * PS + 32 * wave + 16 * nt + 4 * q);
                    acc1[h].v[nt][p] = (f32x4){v.x, v.y, v.z, v.w};
                }
            __syncthreads();
        }
    }

    // ---- layer 1, gathered chunks: iterations over (chunk, half) pairs; the weight fragments of a chunk serve both halves
    Taps tp;
    TapGeom<1> tg[2];
    uint4 wf[2][2];
    const unsigned wvo = (unsigned)(wave * 256 + lane);
#define LOAD_W1(step_)                                                                                                       \
    {                                                                                                                        \
        const rsrc_t wp_ = make_rsrc(hw.w1p, 0x40000000u);                                                                   \
        _Pragma("unroll") for (int nt = 0; nt < 2; nt++)                                                                     \
            _Pragma("unroll") for (int hl = 0; hl < 2; hl++)                                                                 \
                wf[nt][hl] = bload_u4(wp_, wvo * 16u + (unsigned)((nt * 2 + hl) * 1024), (unsigned)(step_) * 16384u);        \
    }
    {
        int mi, co; chunk_info(C0, mi, co);
        taps_geom(a, mi, sUV, tid, tg[0]); taps_geom(a, mi, sUV + 512, tid, tg[1]);
        taps_issue(a, b, mi, co, tg[0], tp);
        LOAD_W1(C0)
        taps_store_feat(tp, tg[0], reinterpret_cast<uint2 *>(lds), reinterpret_cast<uint2 *>(lds + 256), tid, rmax);
        taps_issue(a, b, mi, co, tg[1], tp);                        // pair 1 = (C0, half 1)
    }
    // (H_ = the half = the parity of the iteration IT_: compile-time, so that the per-half accumulators and geometries stay in registers)
#define F1_ITER(H_, IT_)                                                                                                                 \
    {                                                                                                                                    \
        uint4 *buf = lds + (H_) * 512, *nbuf = lds + (1 - (H_)) * 512;                                                                   \
        __syncthreads();                        /* pair IT_ staged; the other buffer's readers (pair IT_ - 1) are done */               \
        k32_step(acc1[H_], wf, buf, buf + 256, 0, lane);                                                                                 \
        if ((IT_) + 1 < NIT) taps_store_feat(tp, tg[1 - (H_)], reinterpret_cast<uint2 *>(nbuf), reinterpret_cast<uint2 *>(nbuf + 256), tid, rmax); \
        if ((H_) == 1) LOAD_W1(ci + 1)          /* after the second half's MFMAs: step ci + 1 (the xyz step after the last chunk) */     \
        if ((IT_) + 2 < NIT) {                                                                                                           \
            int mi, co; chunk_info(ci + 1, mi, co);                                                                                      \
            if (co == 0) taps_geom(a, mi, sUV + (H_) * 512, tid, tg[H_]);     /* half H_ enters a new map (its previous pair was blended an iteration ago) */ \
            taps_issue(a, b, mi, co, tg[H_], tp);                                                                                        \
        }                                                                                                                                \
    }
    for (int it2 = 0; it2 < NIT; it2 += 2) {
        const int ci = C0 + (it2 >> 1);
        F1_ITER(0, it2)
        F1_ITER(1, it2 + 1)
    }
#undef F1_ITER
#pragma unroll
    for (int h = 0; h < 2; h++) {   // z_feat = (x, y, z - 2.2) + the constant one: internal channels 608..611 (K32 step 19)
        uint4 xh[4], xl[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            uint2 hi = make_uint2(0u, 0u), lo = make_uint2(0u, 0u);
            if (q == 0) {
                const float *pp = sPt + (h * 64 + 16 * p + j) * 3;
                split4(pp[0] * a.u1, pp[1] * a.u1, (pp[2] - 2.2f) * a.u1, a.u1, hi, lo, rmax);
            }
            xh[p] = make_uint4(hi.x, hi.y, 0u, 0u); xl[p] = make_uint4(lo.x, lo.y, 0u, 0u);
        }
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int p = 0; p < 4; p++) {
                acc1[h].v[nt][p] = MFMAH(as_h8(wf[nt][0]), as_h8(xh[p]), acc1[h].v[nt][p]);
                acc1[h].v[nt][p] = MFMAH(as_h8(wf[nt][0]), as_h8(xl[p]), acc1[h].v[nt][p]);
                acc1[h].v[nt][p] = MFMAH(as_h8(wf[nt][1]), as_h8(xh[p]), acc1[h].v[nt][p]);
            }
    }
#undef LOAD_W1
    __syncthreads();        // region 0 changes role: chunk buffers -> activation planes

    // ---- layers 2..4 and back: the two halves leapfrog (gemm128_epi), the second GEMM of a pair keeps the first one's weight fragments
    double loss_acc = 0.0;
    uint4 *P0 = Hp, *P1 = Hp + 2048;
    uint2 *P0h = reinterpret_cast<uint2 *>(P0), *P0l = reinterpret_cast<uint2 *>(P0 + 1024), *P1h = reinterpret_cast<uint2 *>(P1), *P1l = reinterpret_cast<uint2 *>(P1 + 1024);
    uint4 *Go0 = Go, *Go1 = Go + 256;
    struct ObjPre { uint4 w4[8]; float bias4; };
    auto half_objective = [&](const int h, const uint4 *Hhi, const uint4 *Hlo, uint4 *Gg, const ObjPre &op) {
        f32x4 o4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const h8 xh = as_h8(Hhi[(4 * s + q) * 64 + 16 * wave + j]), xl = as_h8(Hlo[(4 * s + q) * 64 + 16 * wave + j]);
            const h8 wh = as_h8(op.w4[s * 2 + 0]), wl = as_h8(op.w4[s * 2 + 1]);
            o4 = MFMAH(xh, wh, o4); o4 = MFMAH(xl, wh, o4); o4 = MFMAH(xh, wl, o4);
        }
        float go[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int pt = wave * 16 + q * 4 + r, gp = h * 64 + pt, n = n0 + gp;
            const bool valid = n < a.N;
            const bool inimg = (sIn[gp] & 1) != 0;
            float val = o4[r] * hw.cout + op.bias4;
            if (*sOvf) val = __builtin_nanf("");
            go[r] = 0.f;
            // object = mean_B( mean_N clamp(df[:,1], max=.8) * occ )  (recon_fit_trivis_full.py:155-162)
            if (j == 1 && valid) {
                const float d = inimg ? val : OUT_DIST, ob = a.occ[b];
                loss_acc += (double)(fminf(d, 0.8f) * ob);
                if (inimg && d <= 0.8f) go[r] = a.w0 * ob / ((float)a.B * (float)a.N);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float m = row16_max(fabsf(go[r]));
            const int eb = (int)((__float_as_uint(m) >> 23) & 255u);
            const int ge = hw.goexp;
            const bool ok = eb >= ge + 2 && eb >= 2 && eb < 255 && eb - ge < 254;
            const float s = ok ? __uint_as_float((unsigned)(254 + ge - eb) << 23) : 1.0f;
            const float inv = ok ? __uint_as_float((unsigned)(eb - ge) << 23) : 1.0f;
            const int pt = wave * 16 + q * 4 + r;
            if (j == 0) sInv[h * 64 + pt] = inv;
            const float x = go[r] * s;
            const _Float16 hi = (_Float16)x, lo = (_Float16)(x - (float)hi);
            _Float16 *gh = reinterpret_cast<_Float16 *>(Gg), *gl = reinterpret_cast<_Float16 *>(Gg + 128);
            gh[((j >> 3) * 64 + pt) * 8 + (j & 7)] = hi; gl[((j >> 3) * 64 + pt) * 8 + (j & 7)] = lo;
        }
    };
    auto half_g4t = [&](Acc8 &c, const uint4 (&w)[2][2], const uint4 *Gg) {
        acc_zero(c);
        h8 xh[4], xl[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
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
    {
        WPre w;
        Acc8 c0, c1;
        unsigned m1_0 = 0, m1_1 = 0, m2_0 = 0, m2_1 = 0, m3_0 = 0, m3_1 = 0;
        wprefetch(w, hw.w2p, wave, lane, hw.b2);
        epi_all<EPI_RELU>(acc1[0], m1_0, P0h, P0l, wave, lane, rmax); m1_0 = ~m1_0; OVF_PUBLISH();
        __syncthreads();
        gemm128_epi<EPI_RELU, false, false, true>(c0, P0, P0 + 1024, w, nullptr, nullptr, acc1[1], m1_1, P1h, P1l, wave, lane, rmax); m1_1 = ~m1_1; OVF_PUBLISH();       // L2[0] | E1[1]
        __syncthreads();
        gemm128_epi<EPI_RELU, true, true>(c1, P1, P1 + 1024, w, hw.w3p, hw.b3, c0, m2_0, P0h, P0l, wave, lane, rmax); m2_0 = ~m2_0; OVF_PUBLISH();                       // L2[1] | E2[0]
        __syncthreads();
        gemm128_epi<EPI_RELU, false, false, true>(c0, P0, P0 + 1024, w, nullptr, nullptr, c1, m2_1, P1h, P1l, wave, lane, rmax); m2_1 = ~m2_1; OVF_PUBLISH();            // L3[0] | E2[1]
        __syncthreads();
        gemm128_epi<EPI_RELU, true, false>(c1, P1, P1 + 1024, w, hw.w3tp, nullptr, c0, m3_0, P0h, P0l, wave, lane, rmax); m3_0 = ~m3_0; OVF_PUBLISH();                   // L3[1] | E3[0]
        __syncthreads();
        uint4 w4[2][2];
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int hl = 0; hl < 2; hl++) w4[nt][hl] = hw.w4tp[(((size_t)wave * 2 + nt) * 2 + hl) * 64 + lane];
        ObjPre op;
#pragma unroll
        for (int i = 0; i < 8; i++) op.w4[i] = hw.w4p[i * 64 + lane];
        op.bias4 = hw.b4[j];
        epi_all<EPI_RELU>(c1, m3_1, P1h, P1l, wave, lane, rmax); m3_1 = ~m3_1; OVF_PUBLISH();                                                                          // E3[1]
        half_objective(0, P0, P0 + 1024, Go0, op);
        __syncthreads();
        half_g4t(c0, w4, Go0);
        half_objective(1, P1, P1 + 1024, Go1, op);
        epi_all<EPI_MASK>(c0, m3_0, P0h, P0l, wave, lane, rmax);
        __syncthreads();
        half_g4t(c1, w4, Go1);
        gemm128_epi<EPI_MASK, false, false, true>(c0, P0, P0 + 1024, w, nullptr, nullptr, c1, m3_1, P1h, P1l, wave, lane, rmax);        // W3^T[0] | g3[1]
        __syncthreads();
        gemm128_epi<EPI_MASK, true, false>(c1, P1, P1 + 1024, w, hw.w2tp, nullptr, c0, m2_0, P0h, P0l, wave, lane, rmax);               // W3^T[1] | g2[0]
        __syncthreads();
        gemm128_epi<EPI_MASK, false, false, true>(c0, P0, P0 + 1024, w, nullptr, nullptr, c1, m2_1, P1h, P1l, wave, lane, rmax);        // W2^T[0] | g2[1]
        __syncthreads();
        gemm128_epi<EPI_MASK, false, false>(c1, P1, P1 + 1024, w, nullptr, nullptr, c0, m1_0, P0h, P0l, wave, lane, rmax);              // W2^T[1] | d(hidden-1)[0]
        __syncthreads();
        epi_all<EPI_MASK>(c1, m1_1, P1h, P1l, wave, lane, rmax);
        __syncthreads();
    }
    {   // block-reduce the loss partial into the fp64 term accumulator
        double s = loss_acc;
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) sRed[wave] = s;
        __syncthreads();
        if (tid == 0) {
            double t = (sRed[0] + sRed[1] + sRed[2] + sRed[3]) / ((double)a.B * a.N);
            if (*sOvf) t = (double)__builtin_nanf("");
            atomicAdd(a.terms, t);
        }
    }

    // ---- backward through layer 1 and the gathers
    uint4 dh[2][4][2];
    float kscale[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            dh[h][s][0] = Hp[h * 2048 + (4 * s + q) * 64 + 16 * wave + j];
            dh[h][s][1] = Hp[h * 2048 + 1024 + (4 * s + q) * 64 + 16 * wave + j];
        }
        kscale[h] = sInv[h * 64 + 16 * wave + j] * hw.kback;
    }
    const int mypt = 16 * wave + j;
    const float kx = 2.0f / a.crop * a.fx, ky = 2.0f / a.crop * a.fy;
    float gx[2] = {0.f, 0.f}, gy[2] = {0.f, 0.f}, gz[2] = {0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 2; h++) {
        // im_feat part of the coordinate gradient (query_kernel's USEP block of the backward)
        const float px_ = sPt[(h * 64 + mypt) * 3], py_ = sPt[(h * 64 + mypt) * 3 + 1], iz_ = 1.0f / sPt[(h * 64 + mypt) * 3 + 2];
        const float j0x = kx * iz_, j0y = ky * iz_, j0zu = -kx * px_ * iz_ * iz_, j0zv = -ky * py_ * iz_ * iz_;
        const int spt = tid >> 2, seg = tid & 3;
        unsigned o[4]; float cu[4], cv[4];
        proj_geom(sUV + h * 512, spt, R0_, a.pw, o, cu, cv, true);
#pragma unroll
        for (int k = 0; k < 4; k++) o[k] = (o[k] + 8u * seg) * 4u;
        float dg[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i2 = 0; i2 < 4; i2 += 2) {
            float4 pr[2][4][2];
            uint4 xh[2], xl[2];
#pragma unroll
            for (int ii = 0; ii < 2; ii++) {
                const int kb = 4 * (i2 + ii) + seg;
                const unsigned pc = (unsigned)hw.pcol * 4u;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    pr[ii][k][0] = GATHER_P4(Pb, o[k] + 128u * (i2 + ii), pc); pr[ii][k][1] = GATHER_P4(Pb, o[k] + 128u * (i2 + ii) + 16u, pc);
                }
                xh[ii] = Hp[h * 2048 + kb * 64 + spt]; xl[ii] = Hp[h * 2048 + 1024 + kb * 64 + spt];
            }
#pragma unroll
            for (int ii = 0; ii < 2; ii++) {
                const h8 hh = as_h8(xh[ii]), hl = as_h8(xl[ii]);
                float x[8];
#pragma unroll
                for (int t = 0; t < 8; t++) x[t] = __builtin_fmaf((float)hh[t], 1.0f, (float)hl[t]);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const float4 p0 = pr[ii][k][0], p1 = pr[ii][k][1];
                    dg[k] = __builtin_fmaf(x[7], p1.w, __builtin_fmaf(x[6], p1.z, __builtin_fmaf(x[5], p1.y, __builtin_fmaf(x[4], p1.x,
                            __builtin_fmaf(x[3], p0.w, __builtin_fmaf(x[2], p0.z, __builtin_fmaf(x[1], p0.y, __builtin_fmaf(x[0], p0.x, dg[k]))))))));
                }
            }
        }
        const float ks = sInv[h * 64 + spt] * hw.kback * a.u1inv;
        float dot[4];
#pragma unroll
        for (int k = 0; k < 4; k++) dot[k] = __builtin_fmaf(ks, dg[k], 0.f);
        float su = cu[0] * dot[0] + cu[1] * dot[1] + cu[2] * dot[2] + cu[3] * dot[3];
        float sv = cv[0] * dot[0] + cv[1] * dot[1] + cv[2] * dot[2] + cv[3] * dot[3];
        su += dpp_mov<0xB1>(su); sv += dpp_mov<0xB1>(sv);
        su += dpp_mov<0x4E>(su); sv += dpp_mov<0x4E>(sv);
        su = __shfl(su, 4 * j, 64); sv = __shfl(sv, 4 * j, 64);
        if (q == 0) { gx[h] = su * j0x; gy[h] = sv * j0y; gz[h] = __builtin_fmaf(sv, j0zv, su * j0zu); }
    }
    __syncthreads();        // region 0 changes role again: activation planes -> d(feature) rows + weight slab
    float *sD = reinterpret_cast<float *>(lds);             // [64 points][TS] d feat of the current (chunk, half)
    uint4 *Sl = lds + 1152;                                 // [4 s][2 ct][hi|lo][64 lanes]
    uint4 slabr[4];
#define SLAB_LOAD(ci_)                                                                                                       \
    {                                                                                                                        \
        const rsrc_t sr_ = make_rsrc(hw.w1c, 0x40000000u);                                                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; i_++) slabr[i_] = bload_u4(sr_, (unsigned)(256 * i_ + tid) * 16u, (unsigned)(ci_) * 16384u); \
    }
#define SLAB_STORE() _Pragma("unroll") for (int i_ = 0; i_ < 4; i_++) Sl[256 * i_ + tid] = slabr[i_];
    SLAB_LOAD(C0) SLAB_STORE()
    TapGeom<2> tgb[2];
    { int mi, co; chunk_info(C0, mi, co); taps_geom(a, mi, sUV, tid, tgb[0]); taps_geom(a, mi, sUV + 512, tid, tgb[1]); taps_issue(a, b, mi, co, tgb[0], tp); }
    __syncthreads();
    const int gsub = tid & 7, gpp = tid >> 3;
    float Dt[2][2][4];
    float hx[2][2], hy[2][2], hz[2][2], hp[2][2][2];
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
#pragma unroll
            for (int k = 0; k < 4; k++) Dt[h][pass][k] = 0.f;
            hx[h][pass] = hy[h][pass] = hz[h][pass] = 0.f; hp[h][pass][0] = hp[h][pass][1] = 0.f;
        }
    // one (chunk, half) iteration; H is a compile-time constant so that the per-half accumulators stay in registers
#define B1_ITER(H_, IT_)                                                                                                                      \
    {                                                                                                                                    \
        int mi, co; chunk_info(ci, mi, co);                                                                                              \
        f32x4 dd[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};                                                        \
        _Pragma("unroll") for (int s = 0; s < 4; s++) {                                                                                  \
            const uint4 *f = Sl + ((s * 2) * 2) * 64 + lane;                                                                             \
            const h8 wh0 = as_h8(f[0]), wl0 = as_h8(f[64]), wh1 = as_h8(f[128]), wl1 = as_h8(f[192]);                                    \
            const h8 xh = as_h8(dh[H_][s][0]), xl = as_h8(dh[H_][s][1]);                                                                 \
            dd[0] = MFMAH(wh0, xh, dd[0]); dd[1] = MFMAH(wh1, xh, dd[1]);                                                                \
            dd[0] = MFMAH(wh0, xl, dd[0]); dd[1] = MFMAH(wh1, xl, dd[1]);                                                                \
            dd[0] = MFMAH(wl0, xh, dd[0]); dd[1] = MFMAH(wl1, xh, dd[1]);                                                                \
        }                                                                                                                                \
        _Pragma("unroll") for (int ct = 0; ct < 2; ct++)                                                                                 \
            *reinterpret_cast<float4 *>(sD + mypt * TS + 16 * ct + 4 * q) = make_float4(dd[ct][0] * kscale[H_], dd[ct][1] * kscale[H_], dd[ct][2] * kscale[H_], dd[ct][3] * kscale[H_]); \
        __syncthreads();                                   /* slab(ci) read by this half, d feat visible */                             \
        float4 d4[2];                                                                                                                    \
        _Pragma("unroll") for (int pass = 0; pass < 2; pass++) d4[pass] = *reinterpret_cast<const float4 *>(sD + (gpp + 32 * pass) * TS + 4 * gsub); \
        if ((H_) == 0 && ci + 1 < NCHUNK) SLAB_LOAD(ci + 1)       /* a whole iteration ahead of its store */                            \
        _Pragma("unroll") for (int pass = 0; pass < 2; pass++)                                                                           \
            _Pragma("unroll") for (int k = 0; k < 4; k++) {                                                                              \
                const float4 t = tp.t[pass][k];                                                                                          \
                Dt[H_][pass][k] = __builtin_fmaf(d4[pass].w, t.w, __builtin_fmaf(d4[pass].z, t.z, __builtin_fmaf(d4[pass].y, t.y, __builtin_fmaf(d4[pass].x, t.x, Dt[H_][pass][k])))); \
            }                                                                                                                            \
        int m2i = -1, c2o = 0;                                                                                                           \
        if (ci + 1 < NCHUNK) chunk_info(ci + 1, m2i, c2o);                                                                               \
        if (c2o == 0) {                                    /* the map ends with this chunk */                                            \
            const int pr = map_proj(mi);                                                                                                 \
            _Pragma("unroll") for (int pass = 0; pass < 2; pass++) {                                                                     \
                float su = tgb[H_].c[0][pass][0] * Dt[H_][pass][0], sv = tgb[H_].c[1][pass][0] * Dt[H_][pass][0];                        \
                _Pragma("unroll") for (int k = 1; k < 4; k++) { su = __builtin_fmaf(tgb[H_].c[0][pass][k], Dt[H_][pass][k], su); sv = __builtin_fmaf(tgb[H_].c[1][pass][k], Dt[H_][pass][k], sv); } \
                _Pragma("unroll") for (int k = 0; k < 4; k++) Dt[H_][pass][k] = 0.f;                                                     \
                if (pr == 0) { hp[H_][pass][0] += su; hp[H_][pass][1] += sv; }                                                           \
                else if (pr == 1) { hz[H_][pass] += su; hy[H_][pass] += sv; }                                                            \
                else if (pr == 2) { hx[H_][pass] -= su; hy[H_][pass] += sv; }                                                            \
                else { hx[H_][pass] += su; hz[H_][pass] -= sv; }                                                                         \
            }                                                                                                                            \
        }                                                                                                                                \
        if ((IT_) + 1 < NIT) {                                                                                                           \
            /* the next pair: (ci, half 1) after half 0, (ci + 1, half 0) after half 1; its half enters a new map when its chunk starts one */ \
            const int cin = (H_) == 0 ? ci : ci + 1;                                                                                     \
            int mn, cn; chunk_info(cin, mn, cn);                                                                                         \
            if (cn == 0) taps_geom(a, mn, sUV + (1 - (H_)) * 512, tid, tgb[1 - (H_)]);                                                   \
            taps_issue(a, b, mn, cn, tgb[1 - (H_)], tp);                                                                                 \
            if ((H_) == 1) { SLAB_STORE() }                /* everybody read slab(ci) before the barrier in the middle of this iteration */ \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                           \
            __builtin_amdgcn_s_barrier();                                                                                                \
            asm volatile("" ::: "memory");                                                                                               \
        }                                                                                                                                \
    }
    for (int it2 = 0; it2 < NIT; it2 += 2) {
        const int ci = C0 + (it2 >> 1);
        B1_ITER(0, it2)
        B1_ITER(1, it2 + 1)
    }
#undef B1_ITER
#undef SLAB_LOAD
#undef SLAB_STORE
    // the gathered-map part: perspective Jacobian, sum over the 8 pieces of a tap row, hand-over to the owner lanes -- half by half through sD
#pragma unroll
    for (int h = 0; h < 2; h++) {
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            const float *pp3 = sPt + (h * 64 + gpp + 32 * pass) * 3;
            const float iz = 1.0f / pp3[2];
            float vx = __builtin_fmaf(hp[h][pass][0], kx * iz, hx[h][pass]), vy = __builtin_fmaf(hp[h][pass][1], ky * iz, hy[h][pass]);
            float vz = __builtin_fmaf(hp[h][pass][1], -ky * pp3[1] * iz * iz, __builtin_fmaf(hp[h][pass][0], -kx * pp3[0] * iz * iz, hz[h][pass]));
            vx += dpp_mov<0xB1>(vx); vy += dpp_mov<0xB1>(vy); vz += dpp_mov<0xB1>(vz);
            vx += dpp_mov<0x4E>(vx); vy += dpp_mov<0x4E>(vy); vz += dpp_mov<0x4E>(vz);
            vx += dpp_mov<0x141>(vx); vy += dpp_mov<0x141>(vy); vz += dpp_mov<0x141>(vz);
            if (gsub == 0) { float *o = sD + (gpp + 32 * pass) * 4; o[0] = vx; o[1] = vy; o[2] = vz; }
        }
        __syncthreads();
        if (q == 0) { gx[h] += sD[mypt * 4]; gy[h] += sD[mypt * 4 + 1]; gz[h] += sD[mypt * 4 + 2]; }
    }
    {   // direct xyz features: "chunk" 19 of the slab array, rows 0..2 of its first 16-row tile, straight from L2 (one set of fragments for both halves)
        f32x4 dz[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const rsrc_t fr = make_rsrc(hw.w1c + (size_t)NCHUNK * 1024, 1024u * 16u);
            const h8 wh = as_h8(bload_u4(fr, (unsigned)lane * 16u + (unsigned)(s * 4096), 0u)), wl = as_h8(bload_u4(fr, (unsigned)lane * 16u + (unsigned)(s * 4096 + 1024), 0u));
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const h8 xh = as_h8(dh[h][s][0]), xl = as_h8(dh[h][s][1]);
                dz[h] = MFMAH(wh, xh, dz[h]); dz[h] = MFMAH(wh, xl, dz[h]); dz[h] = MFMAH(wl, xh, dz[h]);
            }
        }
        if (q == 0) {
#pragma unroll
            for (int h = 0; h < 2; h++) { gx[h] += dz[h][0] * kscale[h]; gy[h] += dz[h][1] * kscale[h]; gz[h] += dz[h][2] * kscale[h]; }
        }
    }
    if (q == 0) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            float ox = gx[h], oy = gy[h], oz = gz[h];
            if (*sOvf) ox = oy = oz = __builtin_nanf("");
            const int n = n0 + h * 64 + mypt;
            if (n < a.N) {
                const int pn = sIn[h * 64 + mypt] >> 1;
                float *o = a.dpts + ((size_t)b * a.N + pn) * 3;
                o[0] = ox; o[1] = oy; o[2] = oz;
            }
        }
    }
}

static int launch_object128(const QArgs &a, hipStream_t st)
{
    const size_t lds = lds_bytes_128();
    VT_LDS_LIMIT((query128_kernel<MODE_OBJECT>), lds);
    QArgs b = a; b.skip = vt_skip_flag_of(st);
    hipLaunchKernelGGL((query128_kernel<MODE_OBJECT>), dim3(((a.N + 127) / 128) * a.B), dim3(256), lds, st, b);
    VT_LAUNCH_CHECK();
    return VT_OK;
}
