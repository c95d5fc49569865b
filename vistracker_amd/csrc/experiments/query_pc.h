// query_pc.h -- included by query.hip (same translation unit: it uses that file's helpers).
//
// Producer / consumer variant of the fused SMPL-stage objective (vt_query_human_loss with the hoisted projection): ONE workgroup of eight waves per CU
// works on TWO 64-point tiles of a frame (128 points).
//
// Why: in query_kernel<2, MODE_HUMAN> every wave is gatherer, blender and MFMA issuer in turn, the counters of its three pipes add up
// (texture path 54 % + VALU 35 % + matrix 33 % busy), and half of the bytes through the texture path are weights re-streamed per 64-point workgroup.
// Here, in the two layer-1 loops (half of a workgroup's time, three quarters of its texture-path bytes),
//   * waves 0-3 are CONSUMERS: they only read operand planes / weight slabs from LDS and issue MFMAs -- for BOTH tiles, so a layer-1 weight
//     fragment (forward: A operands from L2; backward: the slab staged in LDS) is fetched once per 128 points instead of once per 64;
//   * waves 4-7 are PRODUCERS: they only gather taps, blend / split / store them (forward) or contract them with d feat (backward) -- for both tiles;
//   * the two roles sit on the same SIMDs pairwise (wave i and wave i + 4), one barrier per chunk, the consumers one chunk behind the producers
//     (forward) / the producers one chunk behind the consumers (backward).
// Everything else -- projections, hoisted-projection blend and dot products, hidden layers, objectives -- is the 256-thread kernel's code run by each
// half of the workgroup on its own tile (waves 0-3 tile 0, waves 4-7 tile 1), with the tile's own LDS region.  Same arithmetic as
// query_kernel<2, MODE_HUMAN, true> (split-f16 MFMA, fp32 accumulate, identical scales and summation order inside a chunk); the coordinate gradient sums
// its three parts in another order (round-off).  Selected with vt_query_set_human_kernel(128) / VT_QUERY_HUMAN_KERNEL=128.
#define PC_R0 4096                        /* region 0 of a tile in uint4 units (64 KB): chunk double buffer -> activation planes of two heads -> d feat rows + weight slab */
#define PC_TILE_FLOATS (64 * 3 + 4 * 64 * 2 + 2 * 64 + 64 + 16 + 4 + 2 * 64 * GEO_STRIDE)
#define PC_TILE_U4 (PC_R0 + 256 + PC_TILE_FLOATS / 4)
static_assert(PC_TILE_FLOATS % 4 == 0, "tile regions are 16-byte multiples");
struct PcTile { uint4 *base, *Go; float *sPt, *sUV, *sInv; int *sIn; double *sRed; int *sOvf; float *sGeo; };
__device__ __forceinline__ PcTile pc_tile(uint4 *lds, int t)
{
    PcTile r;
    r.base = lds + (size_t)t * PC_TILE_U4; r.Go = r.base + PC_R0;
    r.sPt = reinterpret_cast<float *>(r.Go + 256); r.sUV = r.sPt + 64 * 3; r.sInv = r.sUV + 4 * 64 * 2; r.sIn = reinterpret_cast<int *>(r.sInv + 2 * 64);
    r.sRed = reinterpret_cast<double *>(r.sIn + 64); r.sOvf = reinterpret_cast<int *>(r.sRed + 8); r.sGeo = reinterpret_cast<float *>(r.sOvf + 4);
    return r;
}
static size_t lds_bytes_human_pc() { return 16 * (size_t)(2 * PC_TILE_U4) + 4 * 2 * 64 * sizeof(unsigned); }      // + ReLU masks of tile 1's hidden-1 units (4 waves x 2 heads x 64 lanes)

// debug builds with -DPHASE_CLK (tools/bench_scripts/qphase.py pc): shader clocks per phase of a consumer (thread 0: g_phase 0..6) and a producer (thread 256:
// g_phase 8..14) and the clocks each spends waiting at the loop barriers (g_phase 7 / 15)
#ifdef PHASE_CLK
#define PCC(i_) do { if (htid == 0) { const unsigned long long t_ = clock64(); atomicAdd(&g_phase[8 * half + (i_)], t_ - tprev_); tprev_ = t_; } } while (0)
#define PC_LOOP_BARRIER(stmt_) do { const unsigned long long b0_ = clock64(); stmt_; if (htid == 0) atomicAdd(&g_phase[8 * half + 7], clock64() - b0_); } while (0)
// fine timers inside the loops: slot 16 + 8 * half + k accumulates the clocks of section k of an iteration
#define PCF_BEGIN() unsigned long long f0_ = clock64()
#define PCF(k_) do { if (htid == 0) { const unsigned long long t_ = clock64(); atomicAdd(&g_phase[16 + 8 * half + (k_)], t_ - f0_); f0_ = t_; } else f0_ = clock64(); } while (0)
#else
#define PCC(i_)
#define PC_LOOP_BARRIER(stmt_) stmt_
#define PCF_BEGIN()
#define PCF(k_)
#endif
__global__ __launch_bounds__(512, 2) void query_human_pc_kernel(const QArgs a)
{
    VT_SKIP_RETURN(a.skip);
    constexpr int G = 2, C0 = PROJ_C0;
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, q = lane >> 4, j = lane & 15, htid = tid & 255;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6), half = wave8 >> 2, wave = wave8 & 3;       // half: tile of the common phases AND role in the layer-1 loops (0 consumer, 1 producer)
    const PcTile T0 = pc_tile(lds, 0), T1 = pc_tile(lds, 1), Tm = pc_tile(lds, half);
    const PcTile Tt[2] = {T0, T1};
    unsigned *sMask = reinterpret_cast<unsigned *>(lds + 2 * PC_TILE_U4);
    int b, pr;
    {
        const int tiles = (a.N + 63) >> 6, pairs = (tiles + 1) >> 1, L = blockIdx.x;
        if ((a.B & 7) == 0) { const int slot = L >> 3; b = (L & 7) + 8 * (slot / pairs); pr = slot % pairs; }
        else { b = L / pairs; pr = L % pairs; }
        b = __builtin_amdgcn_readfirstlane(b); pr = __builtin_amdgcn_readfirstlane(pr);
    }
#ifdef PHASE_CLK
    unsigned long long tprev_ = clock64();
#endif
    const int n0 = (2 * pr + half) * 64;        // first point slot of this half's tile (a tile past the end of the frame works on clamped points and writes nothing)
    uint4 *Hp = Tm.base, *Go = Tm.Go;
    float *sPt = Tm.sPt, *sUV = Tm.sUV, *sInv = Tm.sInv; int *sIn = Tm.sIn; double *sRed = Tm.sRed; int *sOvf = Tm.sOvf;

    // ---- per-point projections of this half's tile (camera.py:52-90, chore_triplane.py:207-251)
    if (htid == 0) *sOvf = 0;
    if (htid < 64) {
        const int n = min(n0 + htid, a.N - 1);
        const int pn = a.order ? a.order[n] : n;
        const float *p = a.pts + ((size_t)b * a.N + pn) * 3;
        const float x = p[0], y = p[1], z = p[2];
        float px = a.fx * x / z + a.cx, py = a.fy * y / z + a.cy;
        px = a.crop / 2 + px - a.crop_center[2 * b]; py = a.crop / 2 + py - a.crop_center[2 * b + 1];
        const float nx = 2 * px / a.crop - 1, ny = 2 * py / a.crop - 1;
        sIn[htid] = (pn << 1) | (int)((nx >= -1.0f) && (nx <= 1.0f) && (ny >= -1.0f) && (ny <= 1.0f));
        const float c0 = x - a.body_center[3 * b], c1 = y - a.body_center[3 * b + 1], c2 = z - a.body_center[3 * b + 2];
        sPt[htid * 3] = x; sPt[htid * 3 + 1] = y; sPt[htid * 3 + 2] = z;
        sUV[(0 * 64 + htid) * 2] = nx;  sUV[(0 * 64 + htid) * 2 + 1] = ny;
        sUV[(1 * 64 + htid) * 2] = c2;  sUV[(1 * 64 + htid) * 2 + 1] = c1;
        sUV[(2 * 64 + htid) * 2] = -c0; sUV[(2 * 64 + htid) * 2 + 1] = c1;
        sUV[(3 * 64 + htid) * 2] = c0;  sUV[(3 * 64 + htid) * 2 + 1] = -c2;
    }
    __syncthreads();

    // ---- im_feat part of the layer-1 pre-activations: blend of the 4 tap rows of P (query_kernel, USEP), each half for its tile; the blended rows
    //      of BOTH tiles then go into the consumers' accumulator fragments
    constexpr int PS = G * 128 + 4;
    {
        float *stage = reinterpret_cast<float *>(Tm.base);
        const int R = a.res[0];
        const rsrc_t Pb = make_rsrc(a.proj + (size_t)b * R * R * a.pw, (unsigned)(R * R * a.pw) * 4u);
        {
            const int spt = htid >> 2, seg = htid & 3;
            unsigned o[4]; float w[4], unused[4];
            proj_geom(sUV, spt, R, a.pw, o, w, unused, false);
#pragma unroll
            for (int k = 0; k < 4; k++) o[k] = (o[k] + 4u * seg) * 4u;
#pragma unroll
            for (int g = 0; g < G; g++) {
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
    }
    PCC(0);
    // ================= layer 1, forward: producers one chunk ahead of the consumers, one barrier per chunk =================
    // (each role's state is defined and used inside its own branch, barriers included: a value of one role must not be live across the other's code)
    float rmax[2] = {0.f, 0.f};                 // per tile
    unsigned m1v[G] = {0u, 0u};
    if (half == 0) {
        Acc8 acc1[2][G];            // [tile][head]
        {
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const float *st = reinterpret_cast<const float *>(Tt[t].base);
#pragma unroll
                for (int g = 0; g < G; g++)
#pragma unroll
                    for (int nt = 0; nt < 2; nt++)
#pragma unroll
                        for (int p = 0; p < 4; p++) {
                            const float4 v = *reinterpret_cast<const float4 *>(st + (16 * p + j) * PS + g * 128 + 32 * wave + 16 * nt + 4 * q);
                            acc1[t][g].v[nt][p] = (f32x4){v.x, v.y, v.z, v.w};
                        }
            }
        }
        __syncthreads();                        // the chunk double buffers take region 0 over
        // ---- consumers: MFMAs of both tiles on the chunk planes, weight fragments (A operands) once per chunk for both
        uint4 wf[G][2][2];
        const unsigned wvo = (unsigned)(wave * 256 + lane);
#define PC_LOAD_W1(g_, step_)                                                                                                \
    {                                                                                                                        \
        const rsrc_t wp_ = make_rsrc(a.hw[g_].w1p, 0x40000000u);                                                             \
        _Pragma("unroll") for (int nt = 0; nt < 2; nt++)                                                                     \
            _Pragma("unroll") for (int hl = 0; hl < 2; hl++)                                                                 \
                wf[g_][nt][hl] = bload_u4(wp_, wvo * 16u + (unsigned)((nt * 2 + hl) * 1024), (unsigned)(step_) * 16384u);   \
    }
        PC_LOAD_W1(0, C0) PC_LOAD_W1(1, C0)
        for (int ci = C0; ci < NCHUNK; ci++) {
            PC_LOOP_BARRIER(__syncthreads());   // chunk ci of both tiles visible; the other buffers' readers (chunk ci - 1) are done
            // head by head: both tiles take the head's fragments, then the fragments of the next step (19 = the xyz step) are requested into the same
            // registers -- in flight under the other head's MFMAs and the barrier wait
            PCF_BEGIN();
#pragma unroll
            for (int g = 0; g < G; g++) {
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const uint4 *buf = Tt[t].base + (ci & 1) * 512;
                    k32_step(acc1[t][g], wf[g], buf, buf + 256, 0, lane);
                }
                if (g == 0) { PC_LOAD_W1(0, ci + 1) } else { PC_LOAD_W1(1, ci + 1) }
                PCF(g);
            }
        }
        {   // z_feat = (x, y, z - 2.2) + the constant one: K32 step 19
#pragma unroll
            for (int t = 0; t < 2; t++) {
                uint4 xh[4], xl[4];
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    uint2 hi = make_uint2(0u, 0u), lo = make_uint2(0u, 0u);
                    if (q == 0) {
                        const float *pp = Tt[t].sPt + (16 * p + j) * 3;
                        split4(pp[0] * a.u1, pp[1] * a.u1, (pp[2] - 2.2f) * a.u1, a.u1, hi, lo, rmax[t]);
                    }
                    xh[p] = make_uint4(hi.x, hi.y, 0u, 0u); xl[p] = make_uint4(lo.x, lo.y, 0u, 0u);
                }
#pragma unroll
                for (int g = 0; g < G; g++)
#pragma unroll
                    for (int nt = 0; nt < 2; nt++)
#pragma unroll
                        for (int p = 0; p < 4; p++) {
                            acc1[t][g].v[nt][p] = MFMAH(as_h8(wf[g][nt][0]), as_h8(xh[p]), acc1[t][g].v[nt][p]);
                            acc1[t][g].v[nt][p] = MFMAH(as_h8(wf[g][nt][0]), as_h8(xl[p]), acc1[t][g].v[nt][p]);
                            acc1[t][g].v[nt][p] = MFMAH(as_h8(wf[g][nt][1]), as_h8(xh[p]), acc1[t][g].v[nt][p]);
                        }
            }
        }
#undef PC_LOAD_W1
        __syncthreads();        // region 0 of both tiles changes role: chunk buffers -> hidden-activation planes
        // hidden-1 activations of both tiles and heads: packed by the consumers (the accumulators are theirs); the ReLU masks of tile 1 go to its waves through LDS
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int g = 0; g < G; g++) {
                Packed8 pk;
                const unsigned m = relu_pack(acc1[t][g], pk, rmax[t]);
                planes_store(pk, reinterpret_cast<uint2 *>(Tt[t].base + g * 2048), reinterpret_cast<uint2 *>(Tt[t].base + g * 2048 + 1024), wave, lane);
                if (t == 0) m1v[g] = m; else sMask[(wave * G + g) * 64 + lane] = m;
            }
    } else {
        // ---- producers: gather / blend / split / store of both tiles, one chunk ahead
        Taps tp[2]; TapGeom<1> tg[2];
        {
            // tap geometry of the first map of the loop, both tiles, into the rings (producer wave 0)
            int mi, co; chunk_info(C0, mi, co);
            if (wave == 0) { geom_compute<1>(a, mi, T0.sUV, lane, T0.sGeo); geom_compute<1>(a, mi, T1.sUV, lane, T1.sGeo); }
            __syncthreads();                    // (the consumers' "chunk double buffers take region 0 over")
#pragma unroll
            for (int t = 0; t < 2; t++) { geom_fetch<1>(mi, Tt[t].sGeo, htid, tg[t]); taps_issue(a, b, mi, co, tg[t], tp[t]); }
            if (wave == 1 && mi + 1 < 8) { geom_compute<1>(a, mi + 1, T0.sUV, lane, T0.sGeo); geom_compute<1>(a, mi + 1, T1.sUV, lane, T1.sGeo); }
#pragma unroll
            for (int t = 0; t < 2; t++)
                taps_store_feat(tp[t], tg[t], reinterpret_cast<uint2 *>(Tt[t].base + (C0 & 1) * 512), reinterpret_cast<uint2 *>(Tt[t].base + (C0 & 1) * 512 + 256), htid, rmax[t]);
            chunk_info(C0 + 1, mi, co);         // the first map of the loop (tmpx) has two chunks: same geometry
#pragma unroll
            for (int t = 0; t < 2; t++) taps_issue(a, b, mi, co, tg[t], tp[t]);
        }
        for (int ci = C0; ci < NCHUNK; ci++) {
            PC_LOOP_BARRIER(__syncthreads());
            PCF_BEGIN();
            if (ci + 1 < NCHUNK) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PCF(0);        // (timing only: the wait for the taps on its own)
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    uint4 *nbuf = Tt[t].base + ((ci + 1) & 1) * 512;
                    taps_store_feat(tp[t], tg[t], reinterpret_cast<uint2 *>(nbuf), reinterpret_cast<uint2 *>(nbuf + 256), htid, rmax[t]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); PCF(1);
            }
            if (ci + 2 < NCHUNK) {
                int mi, co; chunk_info(ci + 2, mi, co);
                if (co == 0) {
                    // a new map: its ring slot was written an iteration or more ago; the other slot was last read when the previous map started
#pragma unroll
                    for (int t = 0; t < 2; t++) geom_fetch<1>(mi, Tt[t].sGeo, htid, tg[t]);
                    if (mi + 1 < 8 && wave == (mi & 3)) { geom_compute<1>(a, mi + 1, T0.sUV, lane, T0.sGeo); geom_compute<1>(a, mi + 1, T1.sUV, lane, T1.sGeo); }
                }
#pragma unroll
                for (int t = 0; t < 2; t++) taps_issue(a, b, mi, co, tg[t], tp[t]);
                PCF(2);
            }
        }
        __syncthreads();        // (the consumers' region-0 role change)
    }
    PCC(1);
    if (rmax[0] > SPLIT_MAX) *T0.sOvf = 1;
    if (rmax[1] > SPLIT_MAX) *T1.sOvf = 1;
    float rmx = 0.f;            // range tracker of this half's hidden layers (its own tile)
#define PC_OVF() do { if (rmx > SPLIT_MAX) *sOvf = 1; } while (0)

    // ================= per head: layers 2..4, objective, backward to d(hidden-1) -- query_kernel's code, each half on its tile =================
    double loss_acc[2] = {0.0, 0.0};
#pragma unroll
    for (int g = 0; g < G; g++) {
        const HeadW &hw = a.hw[g];
        uint4 *Hhi = Hp + g * 2048, *Hlo = Hhi + 1024;
        uint2 *Hhi8 = reinterpret_cast<uint2 *>(Hhi), *Hlo8 = reinterpret_cast<uint2 *>(Hlo);
        Acc8 c;
        WPre wp;
        Packed8 pk;
        wprefetch(wp, hw.w2p, wave, lane, hw.b2);
        if (g == 0) __syncthreads();           // hidden-1 planes of all heads and tiles visible, sMask too
        const unsigned m1 = half ? sMask[(wave * G + g) * 64 + lane] : m1v[g];
        gemm128(c, Hhi, Hlo, wp, lane);
        wprefetch(wp, hw.w3p, wave, lane, hw.b3);
        const unsigned m2 = relu_pack(c, pk, rmx);
        __syncthreads();
        planes_store(pk, Hhi8, Hlo8, wave, lane); PC_OVF();
        __syncthreads();
        gemm128(c, Hhi, Hlo, wp, lane);
        const unsigned m3 = relu_pack(c, pk, rmx);
        __syncthreads();
        planes_store(pk, Hhi8, Hlo8, wave, lane); PC_OVF();
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
        float go[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int pt = wave * 16 + q * 4 + r, n = n0 + pt;
            const bool valid = n < a.N, live = j < hw.kout;
            const bool inimg = (sIn[pt] & 1) != 0;
            const int pn = sIn[pt] >> 1;
            float val = o4[r] * hw.cout + bias4;
            if (*sOvf) val = __builtin_nanf("");
            go[r] = 0.f;
            if (hw.id == 0) {
                // df_h = clamp(df[:,0], max=.1).mean()  (recon_fit_base.py:640-647); df[~in_img] = 5 (chore_triplane.py:156-159)
                if (j == 0 && valid) {
                    const float d = inimg ? val : OUT_DIST;
                    loss_acc[0] += (double)fminf(d, 0.1f);
                    if (inimg && d <= 0.1f) go[r] = a.w0 / ((float)a.B * (float)a.N);
                }
            } else {
                // part = mean_B sum_N CE(parts, labels)  (recon_fit_behave.py:486)
                const float mx = row16_max(live ? val : -INFINITY);
                const float e = live ? expf(val - mx) : 0.f;
                const float se = row16_sum(e);
                const int lab = a.labels[pn];
                if (valid && live) {
                    go[r] = (e / se - (j == lab ? 1.f : 0.f)) * a.w1 / (float)a.B;
                    if (j == lab) loss_acc[1] += (double)(logf(se) - (val - mx));
                }
            }
        }
        // per-point normalisation of the upstream gradient (see query_kernel)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float m = row16_max(fabsf(go[r]));
            const int eb = (int)((__float_as_uint(m) >> 23) & 255u);
            const int ge = hw.goexp;
            const bool ok = eb >= ge + 2 && eb >= 2 && eb < 255 && eb - ge < 254;
            const float s = ok ? __uint_as_float((unsigned)(254 + ge - eb) << 23) : 1.0f;
            const float inv = ok ? __uint_as_float((unsigned)(eb - ge) << 23) : 1.0f;
            const int pt = wave * 16 + q * 4 + r;
            if (j == 0) sInv[g * 64 + pt] = inv;
            const float x = go[r] * s;
            const _Float16 hi = (_Float16)x, lo = (_Float16)(x - (float)hi);
            _Float16 *gh = reinterpret_cast<_Float16 *>(Go), *gl = reinterpret_cast<_Float16 *>(Go + 128);
            gh[((j >> 3) * 64 + pt) * 8 + (j & 7)] = hi; gl[((j >> 3) * 64 + pt) * 8 + (j & 7)] = lo;
        }
        // backward through layer 4: g3[n][pt] = W4[o][n] . go'[o][pt]
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
        planes_store(pk, Hhi8, Hlo8, wave, lane);   // the planes now hold d loss' / d (pre-activation 1) of this head
        __syncthreads();
    }
#undef PC_OVF
    PCC(2);
    {   // block-reduce the loss partials of this half's tile into the fp64 term accumulators
#pragma unroll
        for (int t = 0; t < 2; t++) {
            double s = loss_acc[t];
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (lane == 0) sRed[t * 4 + wave] = s;
        }
        __syncthreads();
        if (htid < 2) {
            double s = sRed[htid * 4] + sRed[htid * 4 + 1] + sRed[htid * 4 + 2] + sRed[htid * 4 + 3];
            s = htid == 0 ? s / ((double)a.B * a.N) : s / (double)a.B;
            if (*sOvf) s = (double)__builtin_nanf("");
            atomicAdd(a.terms + htid, s);
        }
    }

    // ================= backward through layer 1 and the gathers =================
    const int mypt = 16 * wave + j;             // owner lanes (q == 0) of every wave: point mypt of the half's tile; consumers: the same point index in BOTH tiles
    const float px_ = sPt[mypt * 3], py_ = sPt[mypt * 3 + 1], iz_ = 1.0f / sPt[mypt * 3 + 2];
    const float kx = 2.0f / a.crop * a.fx, ky = 2.0f / a.crop * a.fy;
    const float j0x = kx * iz_, j0y = ky * iz_, j0zu = -kx * px_ * iz_ * iz_, j0zv = -ky * py_ * iz_ * iz_;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    {
        // im_feat part of the coordinate gradient (query_kernel, USEP): each half for its tile
        const int R = a.res[0];
        const rsrc_t Pb = make_rsrc(a.proj + (size_t)b * R * R * a.pw, (unsigned)(R * R * a.pw) * 4u);
        const int spt = htid >> 2, seg = htid & 3;
        unsigned o[4]; float cu[4], cv[4];
        proj_geom(sUV, spt, R, a.pw, o, cu, cv, true);
#pragma unroll
        for (int k = 0; k < 4; k++) o[k] = (o[k] + 8u * seg) * 4u;
        float dot[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < G; g++) {
            float dg[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i2 = 0; i2 < 4; i2 += 2) {
                float4 prw[2][4][2];
                uint4 xh[2], xl[2];
#pragma unroll
                for (int ii = 0; ii < 2; ii++) {
                    const int kb = 4 * (i2 + ii) + seg;
                    const unsigned pc = (unsigned)a.hw[g].pcol * 4u;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        prw[ii][k][0] = GATHER_P4(Pb, o[k] + 128u * (i2 + ii), pc); prw[ii][k][1] = GATHER_P4(Pb, o[k] + 128u * (i2 + ii) + 16u, pc);
                    }
                    xh[ii] = Hp[g * 2048 + kb * 64 + spt]; xl[ii] = Hp[g * 2048 + 1024 + kb * 64 + spt];
                }
#pragma unroll
                for (int ii = 0; ii < 2; ii++) {
                    const h8 hh = as_h8(xh[ii]), hl = as_h8(xl[ii]);
                    float x[8];
#pragma unroll
                    for (int t = 0; t < 8; t++) x[t] = __builtin_fmaf((float)hh[t], 1.0f, (float)hl[t]);
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float4 p0 = prw[ii][k][0], p1 = prw[ii][k][1];
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
        su += dpp_mov<0xB1>(su); sv += dpp_mov<0xB1>(sv);
        su += dpp_mov<0x4E>(su); sv += dpp_mov<0x4E>(sv);
        su = __shfl(su, 4 * j, 64); sv = __shfl(sv, 4 * j, 64);
        if (q == 0) { gx = su * j0x; gy = sv * j0y; gz = __builtin_fmaf(sv, j0zv, su * j0zu); }
    }
    PCC(3);
    // region 0 of tile t now: d feat rows of the tile, two slots of [64 points][TS] floats (uint4 0 .. 1151) | weight slab buffer t, the slab of the chunks with
    // (chunk & 1) == t: [G][4 s][2 ct][hi|lo][64 lanes] (uint4 1152 .. 3199) | hand-over of the xyz part (3200 ..) and of the gathered part (3264 ..) at the end
#define PC_SD(t_, slot_) (reinterpret_cast<float *>(Tt[t_].base) + (slot_) * 64 * TS)
#define PC_SLAB(ci_) ((((ci_) & 1) ? T1.base : T0.base) + 1152)
#define PC_SX(t_) (reinterpret_cast<float *>(Tt[t_].base + 3200))
#define PC_SG(t_) (reinterpret_cast<float *>(Tt[t_].base + 3264))
    // The PRODUCERS issue the slab DMA (asynchronous global -> LDS, 16 B per lane, lane-linear destination = the fragment order) AFTER their LDS reads of an
    // iteration: hipcc orders an LDS access after an LDS-DMA with a full vmcnt(0) (they may alias); at the top of the next iteration that wait is for the
    // taps the dot products need there anyway.  The consumers issue no vector-memory instruction in the loop.
#define PC_SLAB_DMA(ci_)                                                                                                     \
    {                                                                                                                        \
        uint4 *Sl_ = PC_SLAB(ci_);                                                                                           \
        _Pragma("unroll") for (int g_ = 0; g_ < G; g_++)                                                                     \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; i_++)                                                                 \
                __builtin_amdgcn_global_load_lds(a.hw[g_].w1c + (size_t)(ci_) * 1024 + 256 * i_ + htid,                     \
                                                 (__attribute__((address_space(3))) void *)(Sl_ + g_ * 1024 + 256 * i_ + wave * 64), 16, 0, 0); \
    }
    // consumers: B fragments of d(hidden-1) of point mypt in BOTH tiles (128 VGPRs) before the planes give way to the d feat rows / weight slabs
    __builtin_amdgcn_sched_barrier(0);          // (not hoisted above the dot products: their 16 tap rows in flight and these do not fit together)
    if (half == 0) {
        uint4 dh[2][G][4][2];
        float kscale[2][G];
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int g = 0; g < G; g++) {
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    dh[t][g][s][0] = Tt[t].base[g * 2048 + (4 * s + q) * 64 + 16 * wave + j];
                    dh[t][g][s][1] = Tt[t].base[g * 2048 + 1024 + (4 * s + q) * 64 + 16 * wave + j];
                }
                kscale[t][g] = Tt[t].sInv[g * 64 + 16 * wave + j] * a.hw[g].kback;
            }
        __syncthreads();        // region 0 of both tiles changes role again
        // ---- consumers: d feat[32 channels][point mypt] of both tiles per chunk = W1c^T . d(hidden-1), written to the tile's d feat rows; then the xyz part
        for (int ci = C0; ci <= NCHUNK; ci++) {
            PC_LOOP_BARRIER(__syncthreads());   // slab(ci) landed and visible, d feat slot ci & 1 free (its readers finished in iteration ci - 1)
            if (ci < NCHUNK) {
                const uint4 *Sl = PC_SLAB(ci);
                f32x4 dd[2][G][2];
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int g = 0; g < G; g++) { dd[t][g][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; dd[t][g][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    h8 wh[G][2], wl[G][2];
#pragma unroll
                    for (int g = 0; g < G; g++) {
                        const uint4 *f = Sl + (((g * 4 + s) * 2) * 2) * 64 + lane;
                        wh[g][0] = as_h8(f[0]); wl[g][0] = as_h8(f[64]); wh[g][1] = as_h8(f[128]); wl[g][1] = as_h8(f[192]);
                    }
#pragma unroll
                    for (int t = 0; t < 2; t++)
#pragma unroll
                        for (int g = 0; g < G; g++) { dd[t][g][0] = MFMAH(wh[g][0], as_h8(dh[t][g][s][0]), dd[t][g][0]); dd[t][g][1] = MFMAH(wh[g][1], as_h8(dh[t][g][s][0]), dd[t][g][1]); }
#pragma unroll
                    for (int t = 0; t < 2; t++)
#pragma unroll
                        for (int g = 0; g < G; g++) { dd[t][g][0] = MFMAH(wh[g][0], as_h8(dh[t][g][s][1]), dd[t][g][0]); dd[t][g][1] = MFMAH(wh[g][1], as_h8(dh[t][g][s][1]), dd[t][g][1]); }
#pragma unroll
                    for (int t = 0; t < 2; t++)
#pragma unroll
                        for (int g = 0; g < G; g++) { dd[t][g][0] = MFMAH(wl[g][0], as_h8(dh[t][g][s][0]), dd[t][g][0]); dd[t][g][1] = MFMAH(wl[g][1], as_h8(dh[t][g][s][0]), dd[t][g][1]); }
                    __builtin_amdgcn_sched_barrier(0);      // keep the slab fragments of the next K32 step out of this one's registers (128 VGPRs of d(hidden-1) are resident)
                }
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    float *sD = PC_SD(t, ci & 1);
#pragma unroll
                    for (int ct = 0; ct < 2; ct++) {
                        float d[4];
#pragma unroll
                        for (int r = 0; r < 4; r++) d[r] = __builtin_fmaf(dd[t][1][ct][r], kscale[t][1], dd[t][0][ct][r] * kscale[t][0]);
                        *reinterpret_cast<float4 *>(sD + mypt * TS + 16 * ct + 4 * q) = make_float4(d[0], d[1], d[2], d[3]);
                    }
                }
            } else {
                // direct xyz features: d feat[608..610] = rows 0..2 of the first 16-row tile of "chunk" 19, straight from L2 (the producers finish their dot products meanwhile)
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
                    for (int g = 0; g < G; g++) {
                        f32x4 dz = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int s = 0; s < 4; s++) {
                            const rsrc_t fr = make_rsrc(a.hw[g].w1c + (size_t)NCHUNK * 1024, 1024u * 16u);
                            const h8 wh = as_h8(bload_u4(fr, (unsigned)lane * 16u + (unsigned)(s * 4096), 0u)), wl = as_h8(bload_u4(fr, (unsigned)lane * 16u + (unsigned)(s * 4096 + 1024), 0u));
                            dz = MFMAH(wh, as_h8(dh[t][g][s][0]), dz); dz = MFMAH(wh, as_h8(dh[t][g][s][1]), dz); dz = MFMAH(wl, as_h8(dh[t][g][s][0]), dz);
                        }
                        ax += dz[0] * kscale[t][g]; ay += dz[1] * kscale[t][g]; az += dz[2] * kscale[t][g];
                    }
                    if (q == 0) { float *o = PC_SX(t) + mypt * 4; o[0] = ax; o[1] = ay; o[2] = az; }
                }
            }
        }
    } else {
        // ---- producers: one chunk behind: <d feat(ci - 1), raw tap rows> in the gather layout (thread = 16-byte piece gsub of the taps of points gpp, gpp + 32
        //      of both tiles), the tap coefficients and projection Jacobians once per map (query_kernel's layer-1 backward, same arithmetic)
        const int gsub = htid & 7, gpp = htid >> 3;
        {
            int mi, co; chunk_info(C0, mi, co);
            if (wave == 0) { geom_compute<2>(a, mi, T0.sUV, lane, T0.sGeo); geom_compute<2>(a, mi, T1.sUV, lane, T1.sGeo); }
        }
        __syncthreads();        // (the consumers' "region 0 changes role again")
        Taps tp[2]; TapGeom<2> tgb[2];
        float Dt[2][2][4], hx[2][2], hy[2][2], hz[2][2], hp[2][2][2];
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int pass = 0; pass < 2; pass++) {
                hx[t][pass] = hy[t][pass] = hz[t][pass] = hp[t][pass][0] = hp[t][pass][1] = 0.f;
#pragma unroll
                for (int k = 0; k < 4; k++) Dt[t][pass][k] = 0.f;
            }
        {   // what iteration C0 of the consumers needs: slab(C0) landed (first barrier); then its own first step: geometry, slab(C0 + 1), taps(C0)
            PC_SLAB_DMA(C0)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            int mi, co; chunk_info(C0, mi, co);
#pragma unroll
            for (int t = 0; t < 2; t++) geom_fetch<2>(mi, Tt[t].sGeo, htid, tgb[t]);
            if (mi + 1 < 8 && wave == (mi & 3)) { geom_compute<2>(a, mi + 1, T0.sUV, lane, T0.sGeo); geom_compute<2>(a, mi + 1, T1.sUV, lane, T1.sGeo); }
            PC_SLAB_DMA(C0 + 1) asm volatile("" ::: "memory");
#pragma unroll
            for (int t = 0; t < 2; t++) taps_issue(a, b, mi, co, tgb[t], tp[t]);
        }
#pragma nounroll
        for (int ci = C0 + 1; ci <= NCHUNK; ci++) {
            // slab(ci) (the DMA of the last iteration, older than its 16 tap loads) landed, this wave's ring writes done
            PC_LOOP_BARRIER(asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"));
            int mi, co; chunk_info(ci - 1, mi, co);
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const float *sD = PC_SD(t, (ci - 1) & 1);
#pragma unroll
                for (int pass = 0; pass < 2; pass++) {
                    const float4 d4 = *reinterpret_cast<const float4 *>(sD + (gpp + 32 * pass) * TS + 4 * gsub);
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float4 tt = tp[t].t[pass][k];
                        Dt[t][pass][k] = __builtin_fmaf(d4.w, tt.w, __builtin_fmaf(d4.z, tt.z, __builtin_fmaf(d4.y, tt.y, __builtin_fmaf(d4.x, tt.x, Dt[t][pass][k]))));
                    }
                }
            }
            int m2 = -1, c2o = 0;
            if (ci < NCHUNK) chunk_info(ci, m2, c2o);
            if (c2o == 0) {
                // the map of chunk ci - 1 ends with it: tap coefficients on the partial dot products, then the projection Jacobians (query_kernel; camera.py:52-90,
                // chore_triplane.py:220-251) as uniform factors: perspective maps collect (su, sv); right: gz += su, gy += sv; back: gx -= su, gy += sv;
                // top: gx += su, gz -= sv  (x * 1 + y and x * 0 + y are exact: the same sums as the branches they replace)
                const int prj = map_proj(mi);
                const float f_p = prj == 0 ? 1.f : 0.f, f_xu = prj == 2 ? -1.f : (prj == 3 ? 1.f : 0.f), f_yv = (prj == 1 || prj == 2) ? 1.f : 0.f,
                            f_zu = prj == 1 ? 1.f : 0.f, f_zv = prj == 3 ? -1.f : 0.f;
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int pass = 0; pass < 2; pass++) {
                        float su = tgb[t].c[0][pass][0] * Dt[t][pass][0], sv = tgb[t].c[1][pass][0] * Dt[t][pass][0];
#pragma unroll
                        for (int k = 1; k < 4; k++) { su = __builtin_fmaf(tgb[t].c[0][pass][k], Dt[t][pass][k], su); sv = __builtin_fmaf(tgb[t].c[1][pass][k], Dt[t][pass][k], sv); }
#pragma unroll
                        for (int k = 0; k < 4; k++) Dt[t][pass][k] = 0.f;
                        hp[t][pass][0] = __builtin_fmaf(su, f_p, hp[t][pass][0]); hp[t][pass][1] = __builtin_fmaf(sv, f_p, hp[t][pass][1]);
                        hx[t][pass] = __builtin_fmaf(su, f_xu, hx[t][pass]); hy[t][pass] = __builtin_fmaf(sv, f_yv, hy[t][pass]);
                        hz[t][pass] = __builtin_fmaf(sv, f_zv, __builtin_fmaf(su, f_zu, hz[t][pass]));
                    }
            }
            if (ci < NCHUNK) {
                if (c2o == 0) {
#pragma unroll
                    for (int t = 0; t < 2; t++) geom_fetch<2>(m2, Tt[t].sGeo, htid, tgb[t]);
                    if (m2 + 1 < 8 && wave == (m2 & 3)) { geom_compute<2>(a, m2 + 1, T0.sUV, lane, T0.sGeo); geom_compute<2>(a, m2 + 1, T1.sUV, lane, T1.sGeo); }
                }
                if (ci + 1 < NCHUNK) { PC_SLAB_DMA(ci + 1) asm volatile("" ::: "memory"); }
                else {
                    // no slab left to fetch: eight dummy-free iterations would break the "DMA is older than the last 16 loads" count of the wait above;
                    // nothing is outstanding but the taps, and the wait of the last iteration then only retires loads it needs anyway
                }
#pragma unroll
                for (int t = 0; t < 2; t++) taps_issue(a, b, m2, c2o, tgb[t], tp[t]);
            }
        }
        // the gathered-map part of both tiles: perspective Jacobian, sum over the 8 pieces of a tap row, hand-over to the owner lanes
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int pass = 0; pass < 2; pass++) {
                const float *pp3 = Tt[t].sPt + (gpp + 32 * pass) * 3;
                const float iz = 1.0f / pp3[2];
                float vx = __builtin_fmaf(hp[t][pass][0], kx * iz, hx[t][pass]), vy = __builtin_fmaf(hp[t][pass][1], ky * iz, hy[t][pass]);
                float vz = __builtin_fmaf(hp[t][pass][1], -ky * pp3[1] * iz * iz, __builtin_fmaf(hp[t][pass][0], -kx * pp3[0] * iz * iz, hz[t][pass]));
                vx += dpp_mov<0xB1>(vx); vy += dpp_mov<0xB1>(vy); vz += dpp_mov<0xB1>(vz);
                vx += dpp_mov<0x4E>(vx); vy += dpp_mov<0x4E>(vy); vz += dpp_mov<0x4E>(vz);
                vx += dpp_mov<0x141>(vx); vy += dpp_mov<0x141>(vy); vz += dpp_mov<0x141>(vz);
                if (gsub == 0) { float *o = PC_SG(t) + (gpp + 32 * pass) * 4; o[0] = vx; o[1] = vy; o[2] = vz; }
            }
    }
#undef PC_SLAB_DMA
    PCC(4);
    __syncthreads();
    if (q == 0) {
        const float *sg = PC_SG(half) + mypt * 4, *sx = PC_SX(half) + mypt * 4;
        gx += sg[0]; gy += sg[1]; gz += sg[2];
        gx += sx[0]; gy += sx[1]; gz += sx[2];
        if (*sOvf) gx = gy = gz = __builtin_nanf("");
        const int n = n0 + mypt;
        if (n < a.N) {
            const int pn = sIn[mypt] >> 1;
            float *o = a.dpts + ((size_t)b * a.N + pn) * 3;
            o[0] = gx; o[1] = gy; o[2] = gz;
        }
    }
    PCC(5);
#undef PC_SD
#undef PC_SLAB
#undef PC_SX
#undef PC_SG
}
static int launch_human_pc(const QArgs &a, hipStream_t st)
{
    const size_t lds = lds_bytes_human_pc();
    VT_LDS_LIMIT(query_human_pc_kernel, lds);
    QArgs bq = a; bq.skip = vt_skip_flag_of(st);
    const int tiles = (a.N + 63) / 64, pairs = (tiles + 1) / 2;
    hipLaunchKernelGGL(query_human_pc_kernel, dim3(pairs * a.B), dim3(512), lds, st, bq);
    VT_LAUNCH_CHECK();
    return VT_OK;
}
