// sil.hip -- occlusion-aware object silhouette term of SilLossROI (recon/obj_pose_roi.py:77-94,183-207).
//
// PARITY UNPINNED: the reference delegates to neural_renderer (not vendored, not pinned).  Semantics restated from the
// library's published behaviour / Kato et al. 2018 "Neural 3D Mesh Renderer" (DESIGN.md "unpinned"):
//   projection  : x' = x/z, y' = y/z, [u v] = K [x' y' 1], v <- 1 - v, (u,v) <- 2 (. - 1/2)      (orig_size = 1)
//   rasteriser  : faces doubled with reversed winding (fill_back), back faces skipped, pixel (xi,yi) covered when its
//                 centre ((2 xi + 1 - is)/is, (2 yi + 1 - is)/is) is inside the triangle and near < z < far;
//                 nearest face wins; image row r shows yi = is - 1 - r; silhouette = coverage.
//   backward    : Kato's edge-sweep surrogate gradient on the alpha channel.
// MI355X mapping: VALU/latency bound, no MFMA.  Forward = per-face scatter with 64-bit atomicMin depth keys (was: tile binning, the
// faces into an LDS list from precomputed 8-byte pixel boxes, then each pixel tests only that list.  Backward = one wave per
// visible face (faces that own no pixel are skipped).
#include "common.h"

#define SIL_NEAR 0.1f
#define SIL_FAR 100.0f
#define TILE 16
#define MAXLIST 8192

__global__ void sil_project_kernel(const float *__restrict__ verts, const float *__restrict__ K, int NV, float *__restrict__ proj, const int *skip)
{
    VT_SKIP_RETURN(skip);
    const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= NV) return;
    const float *v = verts + ((size_t)b * NV + i) * 3, *k = K + 9 * b;
    const float z = v[2], x_ = v[0] / (z + 1e-9f), y_ = v[1] / (z + 1e-9f);
    const float u = k[0] * x_ + k[1] * y_ + k[2];
    const float w = 1.0f - (k[3] * x_ + k[4] * y_ + k[5]);
    float *o = proj + ((size_t)b * NV + i) * 3;
    o[0] = 2.0f * (u - 0.5f); o[1] = 2.0f * (w - 0.5f); o[2] = z;
}

// Orthographic triplane views of a centred mesh (render/render_triplane_nr.py:110-139, TriplaneNrRenderer.transform_view; the
// renderer is neural_renderer in camera_mode 'look', perspective=False: x, y are used as NDC, z only for near/far).
// "frame" index of the rasteriser = 3 b + view, view 0 right, 1 back, 2 top.
__global__ void sil_triplane_project_kernel(const float *__restrict__ verts, const float *__restrict__ center, int NV, float z_offset,
                                            float *__restrict__ proj)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, bv = blockIdx.y, b = bv / 3, view = bv - 3 * b;
    if (i >= NV) return;
    const float *v = verts + ((size_t)b * NV + i) * 3;
    const float x = v[0] - center[3 * b], y = v[1] - center[3 * b + 1], z = v[2] - center[3 * b + 2];
    float *o = proj + ((size_t)bv * NV + i) * 3;
    if (view == 0) { o[0] = z; o[1] = -y; o[2] = -x + z_offset; }
    else if (view == 1) { o[0] = -x; o[1] = -y; o[2] = -z + z_offset; }
    else { o[0] = x; o[1] = z; o[2] = y + z_offset; }
}

__device__ __forceinline__ void load_face(const float *__restrict__ pv, const int *__restrict__ faces, int NF, int f2, float *fc)
{
    const int f = f2 < NF ? f2 : f2 - NF;
    int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
    if (f2 >= NF) { const int t = i1; i1 = i2; i2 = t; }
    fc[0] = pv[3 * i0]; fc[1] = pv[3 * i0 + 1]; fc[2] = pv[3 * i0 + 2];
    fc[3] = pv[3 * i1]; fc[4] = pv[3 * i1 + 1]; fc[5] = pv[3 * i1 + 2];
    fc[6] = pv[3 * i2]; fc[7] = pv[3 * i2 + 1]; fc[8] = pv[3 * i2 + 2];
}

// workspace layout (floats): proj (B,NV,3) | fc (B,2NF,9) projected face corners | fbox (B,2NF,4 x int16 = 2 floats) pixel bbox,
// x0 > x1 marks culled (back side / off screen) | visible (B,2NF) int | gproj (B,NV,2) fp64 (order-insensitive atomics)
#define SIL_FIX 17179869184.0      /* 2^34: fixed-point unit of the backward accumulators */
struct SilWs { unsigned long long *zbuf, *rowmask, *colmask; float *proj, *fc; int2 *fbox; int *visible; double *gproj; unsigned long long *cnt; unsigned *ticket; int *nvis, *vlist; unsigned *ftick; };
static inline SilWs sil_ws(float *ws, int B, int NV, int NF, int is)
{
    SilWs w;
    w.zbuf = reinterpret_cast<unsigned long long *>(ws);                       // (B,is,is) depth|face keys, internal y-up order
    w.rowmask = w.zbuf + (size_t)B * is * is;                                  // (B,is,is/64) bit xi of row yi: uncovered pixel with d_image < 0
    w.colmask = w.rowmask + (size_t)B * is * (is / 64);                        // (B,is,is/64) bit yi of column xi
    w.proj = reinterpret_cast<float *>(w.colmask + (size_t)B * is * (is / 64));
    w.fc = w.proj + (size_t)B * NV * 3; w.fbox = reinterpret_cast<int2 *>(w.fc + (size_t)B * 2 * NF * 9);
    w.visible = reinterpret_cast<int *>(w.fbox + (size_t)B * 2 * NF); w.gproj = reinterpret_cast<double *>((reinterpret_cast<uintptr_t>(w.visible + (size_t)B * 2 * NF) + 7) & ~(uintptr_t)7);     // 8-byte aligned
    w.cnt = reinterpret_cast<unsigned long long *>(w.gproj + (size_t)B * NV * 2);          // (B) per-frame mask sums, 2^-34 fixed point (vt_sil_step)
    w.ticket = reinterpret_cast<unsigned *>(w.cnt + B);                                      // workgroups of sil_image_kernel that have finished
    w.nvis = reinterpret_cast<int *>(w.ticket + 2);                                          // (B) number of faces that own a pixel (vt_sil_step)
    w.vlist = w.nvis + B;                                                                    // (B,NF) their doubled-face ids, in order of discovery
    w.ftick = reinterpret_cast<unsigned *>(w.vlist + (size_t)B * NF);                        // (B) tiles of the frame that sil_image_kernel has finished
    return w;
}
extern "C" long vt_sil_workspace_floats(int B, int NV, int NF, int size)
{
    return 2L * B * size * size + 4L * B * size * (size / 64) + (long)B * NV * 3 + (long)B * 2 * NF * (9 + 2 + 1) + (long)B * NV * 4 + 16 + 2L * B + 8 + (long)B * (NF + 2);
}

// per (frame, doubled face): corners, back-face test, pixel bounding box -- so that the per-tile culling below streams 8 B per face
__global__ void sil_face_setup_kernel(const float *__restrict__ proj, const int *__restrict__ faces, int NV, int NF, int is,
                                      float *__restrict__ fcbuf, int2 *__restrict__ fbox, int *__restrict__ visible, const int *skip)
{
    VT_SKIP_RETURN(skip);
    const int f2 = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (f2 >= 2 * NF) return;
    float fc[9]; load_face(proj + (size_t)b * NV * 3, faces, NF, f2, fc);
    float *o = fcbuf + ((size_t)b * 2 * NF + f2) * 9;
#pragma unroll
    for (int e = 0; e < 9; e++) o[e] = fc[e];
    visible[(size_t)b * 2 * NF + f2] = 0;
    int x0 = 1, x1 = 0, y0 = 1, y1 = 0;
    if (!((fc[7] - fc[1]) * (fc[3] - fc[0]) < (fc[4] - fc[1]) * (fc[6] - fc[0]))) {     // front side
        const float xmin = fminf(fc[0], fminf(fc[3], fc[6])), xmax = fmaxf(fc[0], fmaxf(fc[3], fc[6]));
        const float ymin = fminf(fc[1], fminf(fc[4], fc[7])), ymax = fmaxf(fc[1], fmaxf(fc[4], fc[7]));
        // pixel centres xp = (2 xi + 1 - is)/is inside [xmin, xmax]  <=>  xi in [ceil((xmin is + is - 1)/2), floor((xmax is + is - 1)/2)];
        // one pixel of slack each side keeps the box conservative under rounding
        const float fx0 = fminf(fmaxf(floorf((xmin * is + is - 1) * 0.5f) - 1.f, -1.f), (float)is), fx1 = fminf(fmaxf(ceilf((xmax * is + is - 1) * 0.5f) + 1.f, -1.f), (float)is);
        const float fy0 = fminf(fmaxf(floorf((ymin * is + is - 1) * 0.5f) - 1.f, -1.f), (float)is), fy1 = fminf(fmaxf(ceilf((ymax * is + is - 1) * 0.5f) + 1.f, -1.f), (float)is);
        x0 = max((int)fx0, 0); x1 = min((int)fx1, is - 1); y0 = max((int)fy0, 0); y1 = min((int)fy1, is - 1);
        if (y0 > y1) { x0 = 1; x1 = 0; }
    }
    fbox[(size_t)b * 2 * NF + f2] = make_int2((x0 & 0xffff) | (x1 << 16), (y0 & 0xffff) | (y1 << 16));
}

// Rasterisation by SCATTER: a group of SIL_G lanes per ORIGINAL face visits only the pixels of the face's box (a 2500-face object covers
// each pixel ~twice, so this is ~35x fewer point-in-triangle tests than testing every pixel against a per-tile face list) and
// resolves visibility with a 64-bit atomicMin on key = (bits of z) << 32 | face id: nearest face wins, ties go to the smaller id.
// Of a face's two orientations (f, f + NF) at most one faces the camera; the record of f + NF is that of f with the last two corners
// swapped (load_face).  A wave is ONE dependent chain record -> box -> pixels -> atomics -> retire, ~4 us of latency at 32 waves per CU
// whatever its contents (measured: with one face per wave the kernel ran at 3.3 waves / ns, the same as the backward kernel; halving the
// number of waves by pairing the orientations changed little) -- so the latency is shared by 64 / SIL_G faces per wave, whose boxes
// (~40 pixels for the fit's object at 256^2) also fill the lanes better than one box per 64 lanes.
#ifndef SIL_G
#define SIL_G 16
#endif
#ifndef SIL_GS
#define SIL_GS SIL_G      /* lanes per face in the scatter kernel (the backward kernel's group is one DPP row: SIL_G) */
#endif
#define SIL_BIG 128     /* boxes above this many pixels are rasterised by the whole wave (64 lanes), not by the face's lane group */
// pixel p of the box (x0, y0, width w) -> image coordinates; p = q w + r without the ~40-instruction integer division: float estimate (p < 2^20 is
// exact in fp32), corrected by at most one
__device__ __forceinline__ void sil_box_pixel(int p, int x0, int y0, int w, float rw, int &xi, int &yi)
{
    int q = (int)((float)p * rw), r = p - q * w;
    if (r >= w) { q++; r -= w; } else if (r < 0) { q--; r += w; }
    xi = x0 + r; yi = y0 + q;
}
// pixel centres in NDC are (2 i + 1 - is) / is: for a power-of-two image size the division is exactly a multiplication by 1 / is (same bits, a tenth
// of the instructions of an IEEE division); other sizes keep the division
__device__ __forceinline__ void sil_ndc(int xi, int yi, int is, bool pow2, float ris, float &xp, float &yp)
{
    const float xn = 2.0f * xi + 1 - is, yn = 2.0f * yi + 1 - is;
    xp = pow2 ? xn * ris : xn / is; yp = pow2 ? yn * ris : yn / is;
}
__device__ __forceinline__ bool sil_inside(const float (&fc)[9], float xp, float yp)
{
    return !(((yp - fc[1]) * (fc[3] - fc[0]) < (xp - fc[0]) * (fc[4] - fc[1])) ||
             ((yp - fc[4]) * (fc[6] - fc[3]) < (xp - fc[3]) * (fc[7] - fc[4])) ||
             ((yp - fc[7]) * (fc[0] - fc[6]) < (xp - fc[6]) * (fc[1] - fc[7])));
}
// depth of a covered pixel (neural_renderer's barycentric formula, division for division) and the visibility vote
// Round 6: the ten IEEE divisions only where they can decide something.  The depth of a covered pixel is used for ONE thing -- which face owns the pixel
// (atomicMin on the depth bits; the owner steers the backward's sweeps) -- and for the near / far test.  Two faces that both cover a pixel centre are either
// separate depth layers of the object (centimetres apart: an approximate depth, relative error ~1e-6, orders them like the exact one) or neighbours meeting in a
// shared edge the centre lies on -- then the centre is on the rim of BOTH, i.e. one of its barycentric weights is ~0 in both.  So: pixels whose smallest weight
// is above SIL_RIM (1/64: >= 0.05 px inside every edge, where the depths of two faces folded over a common edge already differ by >= 1e-5 m) take reciprocal
// multiplies (v_rcp_f32, 1 ulp); pixels on the rim keep neural_renderer's formula division for division, bit for bit the oracle's.  Images and owner maps stay
// identical to the oracle's (tests/test_gpu_parity.py: 96 poses at bench size, every pixel's owner).
#ifndef SIL_IMG_ABL
#define SIL_IMG_ABL 0     /* timing ablations of sil_image_kernel (wrong results): 1 no term / tickets, 2 no flag stores, 4 no gradient clear, 8 no row-mask stores */
#endif
#ifndef SIL_COMPACT
#define SIL_COMPACT 0
#endif
#ifndef SIL_FAST_Z
#define SIL_FAST_Z 1
#endif
#define SIL_RIM 0.015625f
// the visibility vote.  SIL_ZPRETEST: look before the atomic -- keys only ever decrease, so a (possibly stale) value that is already below this face's key means
// the atomic cannot change anything; a stale larger value only costs the atomic it would have cost anyway
#ifndef SIL_ZPRETEST
#define SIL_ZPRETEST 0
#endif
__device__ __forceinline__ void sil_zmin(unsigned long long *p, unsigned long long key)
{
#if SIL_ZPRETEST
    if (*(volatile unsigned long long *)p <= key) return;
#endif
    atomicMin(p, key);
}
__device__ __forceinline__ void sil_vote(const float (&fc)[9], float den, int f2, float xp, float yp, int xi, int yi, int is, unsigned long long *__restrict__ zrow)
{
    const float n0 = (fc[4] - fc[7]) * xp + (fc[6] - fc[3]) * yp + (fc[3] * fc[7] - fc[6] * fc[4]);
    const float n1 = (fc[7] - fc[1]) * xp + (fc[0] - fc[6]) * yp + (fc[6] * fc[1] - fc[0] * fc[7]);
    const float n2 = (fc[1] - fc[4]) * xp + (fc[3] - fc[0]) * yp + (fc[0] * fc[4] - fc[3] * fc[1]);
#if SIL_FAST_Z
    {
        const float rden = __builtin_amdgcn_rcpf(den);
        const float a0 = n0 * rden, a1 = n1 * rden, a2 = n2 * rden;
        if (fminf(a0, fminf(a1, a2)) > SIL_RIM && fmaxf(a0, fmaxf(a1, a2)) < 1.0f - SIL_RIM) {
            // (inside the rim no weight is clamped) zp = 1 / sum_k (w_k / ws) / z_k = ws / sum_k w_k / z_k
            const float ws_ = a0 + a1 + a2;
            const float zp_ = ws_ * __builtin_amdgcn_rcpf(a0 * __builtin_amdgcn_rcpf(fc[2]) + a1 * __builtin_amdgcn_rcpf(fc[5]) + a2 * __builtin_amdgcn_rcpf(fc[8]));
            if (!(zp_ > SIL_NEAR * 1.001f && zp_ < SIL_FAR * 0.999f)) goto exact;            // (at the clipping planes the exact value decides)
            sil_zmin(zrow + (size_t)yi * is + xi, ((unsigned long long)__float_as_uint(zp_) << 32) | (unsigned)f2);
            return;
        }
    }
exact:
#endif
    float w0 = n0 / den;
    float w1 = n1 / den;
    float w2 = n2 / den;
    w0 = fminf(fmaxf(w0, 0.f), 1.f); w1 = fminf(fmaxf(w1, 0.f), 1.f); w2 = fminf(fmaxf(w2, 0.f), 1.f);
    const float ws = w0 + w1 + w2;
    const float zp = 1.0f / (w0 / ws / fc[2] + w1 / ws / fc[5] + w2 / ws / fc[8]);
    if (!(zp > SIL_NEAR && zp < SIL_FAR)) return;
    const unsigned long long key = ((unsigned long long)__float_as_uint(zp) << 32) | (unsigned)f2;     // zp > 0: float order == uint order
    sil_zmin(zrow + (size_t)yi * is + xi, key);
}
// pixels start, start + stride, ... < npx of the box (x0, y0, width w) of face f2 with corners fc (the whole-wave path of large boxes)
__device__ __forceinline__ void sil_scatter_box(const float (&fc)[9], float den, int f2, int x0, int y0, int w, int npx, int start, int stride, int is,
                                                unsigned long long *__restrict__ zrow)
{
    const float rw = 1.0f / (float)w;
    const bool pow2 = (is & (is - 1)) == 0;
    const float ris = 1.0f / (float)is;
    for (int p = start; p < npx; p += stride) {
        int xi, yi; sil_box_pixel(p, x0, y0, w, rw, xi, yi);
        float xp, yp; sil_ndc(xi, yi, is, pow2, ris, xp, yp);
        if (sil_inside(fc, xp, yp)) sil_vote(fc, den, f2, xp, yp, xi, yi, is, zrow);
    }
}
#ifndef SIL_WG_BARRIER
#define SIL_WG_BARRIER 0
#endif
__global__ __launch_bounds__(256) void sil_scatter_kernel(const float *__restrict__ fcbuf, const int2 *__restrict__ fbox, int NF, int is,
                                                          unsigned long long *__restrict__ zbuf, const int *skip)
{
    VT_SKIP_RETURN(skip);
    constexpr int FPW = 64 / SIL_GS;
    __shared__ unsigned short sQ[4][FPW][SIL_BIG];              // per lane group: the box pixels that passed the inside test
    const int lane = threadIdx.x & 63, gl = lane % SIL_GS, grp = lane / SIL_GS, b = blockIdx.y;
    const int f = (blockIdx.x * 4 + (threadIdx.x >> 6)) * FPW + grp;
    unsigned long long *zrow = zbuf + (size_t)b * is * is;
    float fc[9]; float den = 0.f; int f2 = 0, x0 = 0, y0 = 0, w = 1, npx = 0;
    if (f < NF) {
        float fa[9];
#pragma unroll
        for (int e = 0; e < 9; e++) fa[e] = fcbuf[((size_t)b * 2 * NF + f) * 9 + e];
        const int2 bbA = fbox[(size_t)b * 2 * NF + f], bbB = fbox[(size_t)b * 2 * NF + NF + f];
        const bool useA = (bbA.x & 0xffff) <= (bbA.x >> 16);
        const int2 bb = useA ? bbA : bbB;
        f2 = useA ? f : f + NF;
        fc[0] = fa[0]; fc[1] = fa[1]; fc[2] = fa[2];
        fc[3] = useA ? fa[3] : fa[6]; fc[4] = useA ? fa[4] : fa[7]; fc[5] = useA ? fa[5] : fa[8];
        fc[6] = useA ? fa[6] : fa[3]; fc[7] = useA ? fa[7] : fa[4]; fc[8] = useA ? fa[8] : fa[5];
        x0 = bb.x & 0xffff; y0 = bb.y & 0xffff;
        const int x1 = bb.x >> 16, y1 = bb.y >> 16;
        den = fc[0] * (fc[4] - fc[7]) + fc[3] * (fc[7] - fc[1]) + fc[6] * (fc[1] - fc[4]);
        if (x0 <= x1 && den != 0.f) { w = x1 - x0 + 1; npx = w * (y1 - y0 + 1); }
    } else {
#pragma unroll
        for (int e = 0; e < 9; e++) fc[e] = 0.f;
    }
    const bool big = npx > SIL_BIG;
    const float rw = 1.0f / (float)w;
    const bool pow2 = (is & (is - 1)) == 0;
    const float ris = 1.0f / (float)is;
    // small boxes, phase 1: the cheap half of the work (index, NDC, three edge tests) for every pixel of the box; the pixels that are covered are
    // compacted into the group's LDS list.  Half the pixels of a box lie outside its triangle: the expensive half (ten IEEE divisions of the
    // depth formula) then runs on full lanes only.
    unsigned short *qz = sQ[threadIdx.x >> 6][grp];
    int n_in = 0;
    const int nsmall = big ? 0 : npx;
    for (int p0 = 0; p0 < nsmall; p0 += SIL_GS) {
        const int p = p0 + gl;
        bool in = false;
        if (p < nsmall) {
            int xi, yi; sil_box_pixel(p, x0, y0, w, rw, xi, yi);
            float xp, yp; sil_ndc(xi, yi, is, pow2, ris, xp, yp);
            in = sil_inside(fc, xp, yp);
        }
        const unsigned bits = (unsigned)(__ballot(in) >> (SIL_GS * grp)) & (SIL_GS >= 32 ? 0xffffffffu : ((1u << (SIL_GS & 31)) - 1u));
        if (in) qz[n_in + __popc(bits & ((1u << gl) - 1u))] = (unsigned short)p;
        n_in += __popc(bits);
    }
    // the list of a lane group is written and read by lanes of ONE wave: a wave executes its LDS instructions in program order, so only the compiler has to
    // be kept from moving the reads above the writes (round 6: a workgroup barrier here made every wave wait for the largest box among the workgroup's 16 faces)
#if SIL_WG_BARRIER
    __syncthreads();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
    for (int i = gl; i < n_in; i += SIL_GS) {
        const int p = qz[i];
        int xi, yi; sil_box_pixel(p, x0, y0, w, rw, xi, yi);
        float xp, yp; sil_ndc(xi, yi, is, pow2, ris, xp, yp);
        sil_vote(fc, den, f2, xp, yp, xi, yi, is, zrow);
    }
    // large boxes (a close-up mesh, a long sliver): all 64 lanes of the wave take the face, its record broadcast from the group's first lane
    unsigned long long todo = __ballot(big && gl == 0);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1; todo &= todo - 1;
        float fb[9];
#pragma unroll
        for (int e = 0; e < 9; e++) fb[e] = __shfl(fc[e], src, 64);
        sil_scatter_box(fb, __shfl(den, src, 64), __shfl(f2, src, 64), __shfl(x0, src, 64), __shfl(y0, src, 64), __shfl(w, src, 64), __shfl(npx, src, 64), lane, 64, is, zrow);
    }
}

__global__ void sil_resolve_kernel(const unsigned long long *__restrict__ zbuf, int NF, int is, float *__restrict__ image,
                                   int *__restrict__ face_index, int *__restrict__ visible, const int *skip)
{
    VT_SKIP_RETURN(skip);
    const int xi = blockIdx.x * blockDim.x + threadIdx.x, yi = blockIdx.y, b = blockIdx.z;
    if (xi >= is) return;
    const unsigned long long key = zbuf[((size_t)b * is + yi) * is + xi];
    const int f = key == ~0ull ? -1 : (int)(unsigned)(key & 0xffffffffull);
    const size_t o = ((size_t)b * is + (is - 1 - yi)) * is + xi;     // image row 0 = top
    image[o] = f >= 0 ? 1.0f : 0.0f;
    face_index[o] = f;
    if (f >= 0) visible[(size_t)b * 2 * NF + f] = 1;                // benign race: every writer stores 1
}

// bit masks of the pixels that can contribute to an outward sweep: uncovered and d_image < 0 (then (alpha - 1) * g > 0).
// One workgroup per 64 x 64 pixel tile, wave w takes its rows 16 w .. 16 w + 15 with coalesced loads (lane = column, all 32 loads in flight);
// a row's word is the ballot, a column's word collects one bit per row in its lane, the four 16-bit pieces meet in LDS.  Every pixel is read
// once (the first version read it twice, once with a 1 KB lane stride: 52 us).
__global__ __launch_bounds__(256) void sil_sweep_mask_kernel(const int *__restrict__ face_index, const float *__restrict__ d_image, int is,
                                                            unsigned long long *__restrict__ rowmask, unsigned long long *__restrict__ colmask, const int *skip)
{
    VT_SKIP_RETURN(skip);
    __shared__ unsigned sCol[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpl = is / 64;
    const int tile = blockIdx.x, b = blockIdx.y;
    const int tx = tile % wpl, ty = tile / wpl;                 // word column (xi / 64), word row (yi / 64), internal y-up coordinates
    const int xi = tx * 64 + lane;
    int fi[16]; float gd[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int yi = ty * 64 + wave * 16 + r;
        const size_t o = ((size_t)b * is + (is - 1 - yi)) * is + xi;          // pixel (xi, yi) lives at image row is - 1 - yi
        fi[r] = face_index[o]; gd[r] = d_image[o];
    }
    unsigned col = 0;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const bool p = fi[r] < 0 && gd[r] < 0.f;
        const unsigned long long m = __ballot(p);
        if (lane == 0) rowmask[((size_t)b * is + ty * 64 + wave * 16 + r) * wpl + tx] = m;      // row yi, bits xi = 64 tx ..
        col |= (unsigned)p << r;
    }
    sCol[wave][lane] = col;
    __syncthreads();
    if (wave == 0)                                                              // column xi, bits yi = 64 ty ..
        colmask[((size_t)b * is + xi) * wpl + ty] = (unsigned long long)sCol[0][lane] | ((unsigned long long)sCol[1][lane] << 16) |
                                                     ((unsigned long long)sCol[2][lane] << 32) | ((unsigned long long)sCol[3][lane] << 48);
}

#ifndef SIL_BWD_FAST
#define SIL_BWD_FAST 1
#endif
// sum over the SIL_G (= 16: one DPP row) lanes of a face's group, every lane gets the result
__device__ __forceinline__ float group_sum(float v)
{
    static_assert(SIL_G == 16, "group_sum is written for 16-lane groups (one DPP row)");
    auto mv = [](float x, auto ctrl) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, true)); };
    v += mv(v, std::integral_constant<int, 0xB1>{}); v += mv(v, std::integral_constant<int, 0x4E>{});
    v += mv(v, std::integral_constant<int, 0x141>{}); return v + mv(v, std::integral_constant<int, 0x140>{});
}
// Kato et al. edge-sweep surrogate gradient, one wave per (frame, visible doubled face).  A face has six (edge, axis) walks of
// a few positions d0 each; their positions are laid end to end and dealt to the lanes, so all six walks run concurrently (walking
// them one after the other left ~10 of 64 lanes busy and chained six rounds of dependent loads).  The outward sweep of a position
// only visits the pixels flagged in the sweep masks (exactly the pixels whose term is non-zero), the inward sweep is bounded by
// the face itself.  Internal pixel coordinates are y-up: pixel (xi, yi) lives at image row is-1-yi.
__global__ __launch_bounds__(256) void sil_bwd_face_kernel(const float *__restrict__ fcbuf, const int *__restrict__ visible, const int *__restrict__ faces, int NV, int NF, int is,
                                    const int *__restrict__ face_index, const float *__restrict__ d_image, const unsigned long long *__restrict__ rowmask,
                                    const unsigned long long *__restrict__ colmask, float eps, double *__restrict__ gproj, const int *__restrict__ nvis,
                                    const int *__restrict__ vlist, const int *skip)
{
    VT_SKIP_RETURN(skip);
    // (vlist != NULL: vt_sil_step -- group g of frame b takes the g-th listed owner face; the sums are fixed-point atomics, the order of the list does not matter)
    // a group of SIL_G lanes per ORIGINAL face (see sil_scatter_kernel): at most one of its two orientations won pixels.  A visible face is a chain
    // of dependent memory round trips (record -> face index at the edge -> sweep masks -> image gradient), ~8 us at full occupancy whatever the lane
    // count: four faces per wave share that latency (a face has ~40 walk positions: two or three rounds of 16 lanes instead of one of 64).
    const int lane = threadIdx.x & (SIL_G - 1), b = blockIdx.y;
    int f = (blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / SIL_G) + ((threadIdx.x & 63) / SIL_G);
    if (f >= NF) return;
    int visA, visB;
    if (vlist) {
        if (f >= nvis[b]) return;
        const int fl = vlist[(size_t)b * NF + f];
        visA = fl < NF; visB = !visA; f = visA ? fl : fl - NF;
    } else {
        // visibility flags, vertex ids and corners: independent loads, requested before the first branch
        visA = visible[(size_t)b * 2 * NF + f]; visB = visible[(size_t)b * 2 * NF + NF + f];
    }
    int vi[3] = {faces[3 * f], faces[3 * f + 1], faces[3 * f + 2]};
    float PA[3][2];         // corners of orientation f in pixel units
#pragma unroll
    for (int k = 0; k < 3; k++) {
        PA[k][0] = 0.5f * (fcbuf[((size_t)b * 2 * NF + f) * 9 + 3 * k] * is + is - 1);
        PA[k][1] = 0.5f * (fcbuf[((size_t)b * 2 * NF + f) * 9 + 3 * k + 1] * is + is - 1);
    }
    const int vis = visA | visB;
    const int f2 = visA ? f : f + NF;
    const float P[3][2] = {{PA[0][0], PA[0][1]}, {visA ? PA[1][0] : PA[2][0], visA ? PA[1][1] : PA[2][1]}, {visA ? PA[2][0] : PA[1][0], visA ? PA[2][1] : PA[1][1]}};
    if (!vis) return;
    if (f2 >= NF) { const int t = vi[1]; vi[1] = vi[2]; vi[2] = t; }
    const int *fim = face_index + (size_t)b * is * is;
    const float *gal = d_image + (size_t)b * is * is;
    const int wpl = is / 64;
    // walk w = 2 edge + axis covers d0 in [from[w], from[w] + cnt[w]) along the axis; start[] = exclusive prefix of cnt[]
    int from[6], start[7];
    start[0] = 0;
#pragma unroll
    for (int w = 0; w < 6; w++) {
        const int edge = w >> 1, axis = w & 1;
        const float a0 = P[edge][axis], a1 = P[(edge + 1) % 3][axis];
        const int d0_from = (int)fmaxf(ceilf(fminf(a0, a1)), 0.0f), d0_to = (int)fminf(fmaxf(a0, a1), (float)(is - 1));
        from[w] = d0_from; start[w + 1] = start[w] + max(d0_to - d0_from + 1, 0);
    }
    float acc[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    const bool pow2 = (is & (is - 1)) == 0;
    const float ris = 1.0f / (float)is;
#define PIX(d0_, d1_, axis_) ((axis_) == 0 ? (size_t)(is - 1 - (d1_)) * is + (d0_) : (size_t)(is - 1 - (d0_)) * is + (d1_))
    for (int base = 0; base < start[6]; base += SIL_G) {
        const int idx = base + lane;
        if (idx >= start[6]) continue;
        int w = 0, d0 = from[0] + idx;
#pragma unroll
        for (int k = 1; k < 6; k++) if (idx >= start[k]) { w = k; d0 = from[k] + idx - start[k]; }
        const int edge = w >> 1, axis = w & 1;
        // corners of this walk: p[k] = P[(edge + k) % 3] with the walk axis first
        float p[3][2];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float q0 = edge == 0 ? P[k][0] : (edge == 1 ? P[(k + 1) % 3][0] : P[(k + 2) % 3][0]);
            const float q1 = edge == 0 ? P[k][1] : (edge == 1 ? P[(k + 1) % 3][1] : P[(k + 2) % 3][1]);
            p[k][0] = axis == 0 ? q0 : q1; p[k][1] = axis == 0 ? q1 : q0;
        }
        int direction;
        if (axis == 0) direction = (p[0][0] < p[1][0]) ? -1 : 1; else direction = (p[0][0] < p[1][0]) ? 1 : -1;
        const unsigned long long *masks = (axis == 0 ? colmask : rowmask) + (size_t)b * is * wpl;
        float ga = 0.f, gb = 0.f;       // gradient of corner edge (p[0]) and corner edge+1 (p[1]) along the other axis
        do {
            const float d1_cross = (p[1][1] - p[0][1]) / (p[1][0] - p[0][0]) * (d0 - p[0][0]) + p[0][1];
            const int d1_in = (0 < direction) ? (int)floorf(d1_cross) : (int)ceilf(d1_cross);
            const int d1_out = d1_in + direction;
            if (d1_in < 0 || is <= d1_in) break;
            if (d1_out < 0 || is <= d1_out) break;
            const size_t idx_in = PIX(d0, d1_in, axis), idx_out = PIX(d0, d1_out, axis);
            const float alpha_out = fim[idx_out] >= 0 ? 1.f : 0.f;
            // (... * 2.0f / is): for a power-of-two image size the division is exactly a multiplication by 1 / is
            // SIL_BWD_FAST (round 6): the divisions that only SCALE a term -- tA, tB and diff_grad / dist below -- as reciprocal multiplies (v_rcp_f32, 1 ulp: the
            // vertex gradients move by ~2e-7 relative, the tests hold them to 1e-5 of the oracle's); d1_cross and d0_cross2, whose floor / ceil pick the PIXELS of a
            // sweep, keep the IEEE division -- one ulp there moves a sweep boundary (the 1e-4 jumps of section 2 of DESIGN.md)
#if SIL_BWD_FAST
#define SIL_DIV(a_, b_) ((a_) * __builtin_amdgcn_rcpf(b_))
#else
#define SIL_DIV(a_, b_) ((a_) / (b_))
#endif
            const float tA = SIL_DIV(p[1][0] - p[0][0], p[1][0] - d0) * 2.0f, tB = SIL_DIV(p[1][0] - p[0][0], d0 - p[0][0]) * 2.0f;
            const float sA = (p[1][0] != d0) ? (pow2 ? tA * ris : tA / is) : 0.f;   // dist = sA * (d1 - d1_cross) for corner 0
            const float sB = (p[0][0] != d0) ? (pow2 ? tB * ris : tB / is) : 0.f;   // ... for corner 1
            // Round 6: memory-level parallelism inside a position's chain.  The owner of the in-pixel, the sweep-mask words of the whole row / column (is <= 256: four
            // words = 32 contiguous bytes) and, above, the owner of the out-pixel are requested TOGETHER (one round trip instead of two); the two sweeps fetch the
            // next pixel's values while the current one is being added (the adds stay in the same order: bit-identical sums).
            unsigned long long mw[4] = {0ull, 0ull, 0ull, 0ull};
            if (wpl <= 4) {
#pragma unroll
                for (int k = 0; k < 4; k++) if (k < wpl) mw[k] = masks[(size_t)d0 * wpl + k];
            }
            if (fim[idx_in] == f2) {   // sweep outwards from the edge: only flagged pixels have a non-zero term (alpha_in = 1)
                const int d1_limit = (0 < direction) ? is - 1 : 0;
                const int d1_from = max(min(d1_out, d1_limit), 0), d1_to = min(max(d1_out, d1_limit), is - 1);
                for (int wd = d1_from >> 6; wd <= (d1_to >> 6); wd++) {
                    unsigned long long bits = (wpl <= 4) ? (wd == 0 ? mw[0] : wd == 1 ? mw[1] : wd == 2 ? mw[2] : mw[3]) : masks[(size_t)d0 * wpl + wd];
                    const int lo = max(d1_from - wd * 64, 0), hi = min(d1_to - wd * 64, 63);
                    bits &= (~0ull << lo) & (~0ull >> (63 - hi));
                    float nxt = 0.f;
                    if (bits) nxt = gal[PIX(d0, wd * 64 + __ffsll((long long)bits) - 1, axis)];
                    while (bits) {
                        const int d1 = wd * 64 + __ffsll((long long)bits) - 1; bits &= bits - 1;
                        const float diff_grad = -nxt;
                        if (bits) nxt = gal[PIX(d0, wd * 64 + __ffsll((long long)bits) - 1, axis)];
                        if (diff_grad <= 0) continue;
                        if (p[1][0] != d0) { float dist = sA * (d1 - d1_cross); dist = (0 < dist) ? dist + eps : dist - eps; ga -= SIL_DIV(diff_grad, dist); }
                        if (p[0][0] != d0) { float dist = sB * (d1 - d1_cross); dist = (0 < dist) ? dist + eps : dist - eps; gb -= SIL_DIV(diff_grad, dist); }
                    }
                }
            }
            if (alpha_out == 0.f) {     // sweep inwards over this face's own pixels ((1 - alpha_out) * g vanishes otherwise)
                float d0_cross2;
                if ((d0 - p[0][0]) * (d0 - p[2][0]) < 0) d0_cross2 = (p[2][1] - p[0][1]) / (p[2][0] - p[0][0]) * (d0 - p[0][0]) + p[0][1];
                else d0_cross2 = (p[1][1] - p[2][1]) / (p[1][0] - p[2][0]) * (d0 - p[2][0]) + p[2][1];
                const int d1_limit = (0 < direction) ? (int)ceilf(d0_cross2) : (int)floorf(d0_cross2);
                const int d1_from = max(min(d1_in, d1_limit), 0), d1_to = min(max(d1_in, d1_limit), is - 1);
                int nf_ = 0; float ng_ = 0.f;
                if (d1_from <= d1_to) { const size_t ix0 = PIX(d0, d1_from, axis); nf_ = fim[ix0]; ng_ = gal[ix0]; }
                for (int d1 = d1_from; d1 <= d1_to; d1++) {
                    const int own = nf_; const float diff_grad = ng_;
                    if (d1 < d1_to) { const size_t ix1 = PIX(d0, d1 + 1, axis); nf_ = fim[ix1]; ng_ = gal[ix1]; }
                    if (own != f2) continue;
                    if (diff_grad <= 0) continue;
                    if (p[1][0] != d0) { float dist = sA * (d1 - d1_cross); dist = (0 < dist) ? dist + eps : dist - eps; ga -= SIL_DIV(diff_grad, dist); }
                    if (p[0][0] != d0) { float dist = sB * (d1 - d1_cross); dist = (0 < dist) ? dist + eps : dist - eps; gb -= SIL_DIV(diff_grad, dist); }
                }
            }
        } while (false);
        // corner `edge` takes ga, corner (edge + 1) % 3 takes gb, both in the component perpendicular to the walk axis
        const int ka = edge, kb = edge == 2 ? 0 : edge + 1, c = 1 - axis;
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int cc = 0; cc < 2; cc++) acc[k][cc] += (cc == c) ? ((k == ka ? ga : 0.f) + (k == kb ? gb : 0.f)) : 0.f;
    }
#undef PIX
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const float g = group_sum(acc[k][c]);
            // FIXED-POINT accumulation (2^-34 units in a 64-bit integer, |sum| < 5e8): integer addition is associative, so the per-vertex sum over
            // its faces is bit-identical whatever order the waves arrive in.  (fp64 atomics left a ~2^-29 chance per value that two runs round to
            // different floats -- enough, over 300 'sil' steps, to make one run in ten drift from another through Adam's amplification.)
            if (lane == 0 && g != 0.f)
                atomicAdd(reinterpret_cast<unsigned long long *>(gproj) + ((size_t)b * NV + vi[k]) * 2 + c, (unsigned long long)__double2ll_rn((double)g * SIL_FIX));
        }
}

// (cnt != NULL: vt_sil_step -- workgroup (0, 0) also closes the mask term: mean_b(per[b] occ[b]) from sil_image_kernel's per-frame fixed-point sums, added in frame order)
__global__ void sil_unproject_kernel(const float *__restrict__ verts, const float *__restrict__ K, int NV, const double *__restrict__ gproj,
                                     float *__restrict__ dverts, const unsigned long long *__restrict__ cnt, const float *__restrict__ occ, int B, double *term, const int *skip)
{
    VT_SKIP_RETURN(skip);
    if (cnt && term && blockIdx.x == 0 && blockIdx.y == 0) {
        __shared__ double sPer[256];
        double s_ = 0;
        for (int i0 = 0; i0 < B; i0 += (int)blockDim.x) {
            const int i = i0 + (int)threadIdx.x;
            sPer[threadIdx.x] = i < B ? (double)(long long)cnt[i] * (1.0 / SIL_FIX) * (double)occ[i] / (double)B : 0.0;
            __syncthreads();
            if (threadIdx.x == 0) for (int k = 0; k < min((int)blockDim.x, B - i0); k++) s_ += sPer[k];
            __syncthreads();
        }
        if (threadIdx.x == 0) atomicAdd(term, s_);
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= NV) return;
    const float *v = verts + ((size_t)b * NV + i) * 3, *k = K + 9 * b;
    const long long *gq = reinterpret_cast<const long long *>(gproj);
    const float z = v[2] + 1e-9f, gu = (float)((double)gq[((size_t)b * NV + i) * 2] * (1.0 / SIL_FIX)), gv = (float)((double)gq[((size_t)b * NV + i) * 2 + 1] * (1.0 / SIL_FIX));
    const float gx_ = 2.f * gu * k[0] - 2.f * gv * k[3], gy_ = 2.f * gu * k[1] - 2.f * gv * k[4];
    float *o = dverts + ((size_t)b * NV + i) * 3;
    o[0] = gx_ / z; o[1] = gy_ / z; o[2] = -(gx_ * v[0] + gy_ * v[1]) / (z * z);
}

__global__ __launch_bounds__(256) void sil_mask_loss_kernel(const float *__restrict__ image, const float *__restrict__ keep, const float *__restrict__ ref,
                                                            const float *__restrict__ occ, int B, int npx, float gs, double *term,
                                                            float *per_frame, float *__restrict__ d_image, const int *skip)
{
    VT_SKIP_RETURN(skip);
    // grid = (slices, B): a frame's pixels are split over gridDim.x blocks (one block per frame left most of the chip idle);
    // per_frame is only supported with one slice
    __shared__ double red[4];
    const int b = blockIdx.y;
    const float ob = occ[b];
    double acc = 0;
    // 16-byte accesses (npx = size^2 is a multiple of 4096, frames start 16-byte aligned): a handful of wide loads per thread instead of
    // a 64-deep chain of dependent-latency scalar ones
    const int nq = npx >> 2, per_slice = (nq + gridDim.x - 1) / gridDim.x, i_end = min(nq, (int)(blockIdx.x + 1) * per_slice);
    const float4 *im4 = (const float4 *)(image + (size_t)b * npx), *kp4 = (const float4 *)(keep + (size_t)b * npx), *rf4 = (const float4 *)(ref + (size_t)b * npx);
    float4 *di4 = d_image ? (float4 *)(d_image + (size_t)b * npx) : nullptr;
    for (int i = blockIdx.x * per_slice + threadIdx.x; i < i_end; i += 256) {
        const float4 im = im4[i], kp = kp4[i], rf = rf4[i];
        const float d0 = kp.x * im.x - rf.x, d1 = kp.y * im.y - rf.y, d2 = kp.z * im.z - rf.z, d3 = kp.w * im.w - rf.w;
        acc += (double)(d0 * d0) + (double)(d1 * d1) + (double)(d2 * d2) + (double)(d3 * d3);
        if (di4) di4[i] = make_float4(2.f * d0 * kp.x * ob * gs, 2.f * d1 * kp.y * ob * gs, 2.f * d2 * kp.z * ob * gs, 2.f * d3 * kp.w * ob * gs);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double per = red[0] + red[1] + red[2] + red[3];
        if (per_frame) per_frame[b] = (float)per;
        if (term) atomicAdd(term, per * (double)ob / (double)B);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------------------
// vt_sil_step (round 6): the whole silhouette term of ONE Adam step of phase 'sil' -- vt_sil_forward + vt_sil_mask_loss + vt_sil_backward -- in 5 launches instead of
// 10 (project, face set-up, depth-buffer clear, scatter, resolve, mask term, gradient clear, sweep masks, backward walks, un-projection).  The kernels are latency
// chains on a 256-CU chip: what a step pays for is mostly the NUMBER of them.  Same arithmetic, operation for operation: face corners, owner map,
// d_image, sweep masks and vertex gradients are bit-identical to the ten-launch path (tests/test_gpu_parity.py::test_sil_step_equals_the_separate_launches).
//   sil_setup_kernel  = sil_project_kernel (every doubled face projects its own three corners: the same expressions) + sil_face_setup_kernel + the clear of the
//                       frame's depth keys (grid-stride 16-byte stores);
//   sil_image_kernel  = sil_resolve_kernel + sil_mask_loss_kernel + sil_sweep_mask_kernel + the clear of the projected-gradient accumulators, one workgroup per
//                       64 x 64 tile: a pixel's depth key, keep and reference value are read once; owner, d_image and the two sweep-mask words are written.  The mask
//                       term: per-frame sums in 2^-34 fixed point (integer atomics: order-independent), workgroup (0, 0) of sil_unproject_kernel -- the step's
//                       last launch -- adds mean_b(per[b] occ[b]) to *term in frame order (the sums are zeroed by sil_setup_kernel: no state across calls).
__global__ void sil_setup_kernel(const float *__restrict__ verts, const float *__restrict__ K, const int *__restrict__ faces, int NV, int NF, int is,
                                 float *__restrict__ fcbuf, int2 *__restrict__ fbox, int *__restrict__ visible, unsigned long long *__restrict__ zbuf,
                                 unsigned long long *__restrict__ cnt, unsigned *__restrict__ ticket, int *__restrict__ nvis, unsigned *__restrict__ ftick, const int *skip)
{
    VT_SKIP_RETURN(skip);
    const int b = blockIdx.y;
    if (blockIdx.x == 0 && threadIdx.x == 0) { cnt[b] = 0ull; nvis[b] = 0; ftick[b] = 0u; if (b == 0) *ticket = 0u; }      // the mask sums of the step start from zero (no state across calls)
    {   // depth keys of frame b <- ~0 (no face): is * is * 8 bytes over the gridDim.x * 256 threads of the frame
        uint4 *z4 = reinterpret_cast<uint4 *>(zbuf + (size_t)b * is * is);
        const int n4 = is * is / 2;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) z4[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
    }
    const int f2 = blockIdx.x * blockDim.x + threadIdx.x;
    if (f2 >= 2 * NF) return;
    const int f = f2 < NF ? f2 : f2 - NF;
    int vi[3] = {faces[3 * f], faces[3 * f + 1], faces[3 * f + 2]};
    if (f2 >= NF) { const int t = vi[1]; vi[1] = vi[2]; vi[2] = t; }
    const float *k = K + 9 * b;
    float fc[9];
#pragma unroll
    for (int c = 0; c < 3; c++) {       // sil_project_kernel, expression for expression
        const float *v = verts + ((size_t)b * NV + vi[c]) * 3;
        const float z = v[2], x_ = v[0] / (z + 1e-9f), y_ = v[1] / (z + 1e-9f);
        const float u = k[0] * x_ + k[1] * y_ + k[2];
        const float w = 1.0f - (k[3] * x_ + k[4] * y_ + k[5]);
        fc[3 * c] = 2.0f * (u - 0.5f); fc[3 * c + 1] = 2.0f * (w - 0.5f); fc[3 * c + 2] = z;
    }
    float *o = fcbuf + ((size_t)b * 2 * NF + f2) * 9;
#pragma unroll
    for (int e = 0; e < 9; e++) o[e] = fc[e];
    visible[(size_t)b * 2 * NF + f2] = 0;
    int x0 = 1, x1 = 0, y0 = 1, y1 = 0;
    if (!((fc[7] - fc[1]) * (fc[3] - fc[0]) < (fc[4] - fc[1]) * (fc[6] - fc[0]))) {     // front side (sil_face_setup_kernel)
        const float xmin = fminf(fc[0], fminf(fc[3], fc[6])), xmax = fmaxf(fc[0], fmaxf(fc[3], fc[6]));
        const float ymin = fminf(fc[1], fminf(fc[4], fc[7])), ymax = fmaxf(fc[1], fmaxf(fc[4], fc[7]));
        const float fx0 = fminf(fmaxf(floorf((xmin * is + is - 1) * 0.5f) - 1.f, -1.f), (float)is), fx1 = fminf(fmaxf(ceilf((xmax * is + is - 1) * 0.5f) + 1.f, -1.f), (float)is);
        const float fy0 = fminf(fmaxf(floorf((ymin * is + is - 1) * 0.5f) - 1.f, -1.f), (float)is), fy1 = fminf(fmaxf(ceilf((ymax * is + is - 1) * 0.5f) + 1.f, -1.f), (float)is);
        x0 = max((int)fx0, 0); x1 = min((int)fx1, is - 1); y0 = max((int)fy0, 0); y1 = min((int)fy1, is - 1);
        if (y0 > y1) { x0 = 1; x1 = 0; }
    }
    fbox[(size_t)b * 2 * NF + f2] = make_int2((x0 & 0xffff) | (x1 << 16), (y0 & 0xffff) | (y1 << 16));
}

__global__ __launch_bounds__(256) void sil_image_kernel(const unsigned long long *__restrict__ zbuf, int NV, int NF, int is, const float *__restrict__ keep,
                                                        const float *__restrict__ ref, const float *__restrict__ occ, int B, float gs, double *term,
                                                        int *__restrict__ face_index, float *__restrict__ d_image, float *__restrict__ image, int *__restrict__ visible,
                                                        unsigned long long *__restrict__ rowmask, unsigned long long *__restrict__ colmask, double *__restrict__ gproj,
                                                        unsigned long long *__restrict__ cnt, unsigned *__restrict__ ticket, int *__restrict__ nvis,
                                                        int *__restrict__ vlist, unsigned *__restrict__ ftick, const int *skip)
{
    VT_SKIP_RETURN(skip);
    __shared__ unsigned sCol[4][64];
    __shared__ double sSum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpl = is / 64;
    const int tile = blockIdx.x, b = blockIdx.y, tiles = wpl * wpl;
    const int tx = tile % wpl, ty = tile / wpl;                 // internal y-up tile coordinates (as sil_sweep_mask_kernel)
    const int xi = tx * 64 + lane;
    {   // this workgroup's share of the frame's projected-gradient accumulators <- 0 (the memset of vt_sil_backward)
        unsigned long long *g = reinterpret_cast<unsigned long long *>(gproj) + (size_t)b * NV * 2;
        if (!(SIL_IMG_ABL & 4)) for (int i = tile * 256 + threadIdx.x; i < NV * 2; i += tiles * 256) g[i] = 0ull;
    }
    const float ob = occ[b];
    unsigned long long key[16]; float kp[16], rf[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int yi = ty * 64 + wave * 16 + r;
        const size_t o = ((size_t)b * is + (is - 1 - yi)) * is + xi;          // pixel (xi, yi) lives at image row is - 1 - yi
        key[r] = zbuf[((size_t)b * is + yi) * is + xi]; kp[r] = keep[o]; rf[r] = ref[o];
    }
    // The first pixel that finds its owner's flag clear lists the face for the backward walks: the backward kernel then runs over the ~45 % of the faces that own
    // pixels, densely packed.  The flags of the 16 rows are requested together (16 independent loads, not 16 dependent round trips); the faces a workgroup is first
    // to see are collected in LDS and appended to the frame's list with ONE global atomic per workgroup (every face its own atomic on nvis[b]: ~1100 same-address
    // atomics per frame, 90 us of this kernel).
    // MEASURED (round 6): the list costs the image kernel more than it saves the backward kernel -- 137 us with one atomic per pixel's first sight, 97 us with
    // rim-only atomics + LDS aggregation, against 40 us without, for 88 -> 84 us in sil_bwd_face (whose time is the dependent chain of a face, not the number of
    // waves) -> SIL_COMPACT = 0: plain flag stores, the backward kernel tests the flags of all faces.
    constexpr int LCAP = 1024;
    __shared__ int sList[LCAP];
    __shared__ int sN, sBase;
    if (threadIdx.x == 0) sN = 0;
    __syncthreads();
    int fown[16], seen[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        fown[r] = key[r] == ~0ull ? -1 : (int)(unsigned)(key[r] & 0xffffffffull);
        seen[r] = (SIL_COMPACT && fown[r] >= 0) ? visible[(size_t)b * 2 * NF + fown[r]] : 1;
    }
    unsigned col = 0; double acc = 0;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int yi = ty * 64 + wave * 16 + r;
        const size_t o = ((size_t)b * is + (is - 1 - yi)) * is + xi;
        const int f = fown[r];
        const float im = f >= 0 ? 1.0f : 0.0f;
        face_index[o] = f;
        if (image) image[o] = im;
        // (all 1536 workgroups of a launch are resident at once: early on every pixel still reads a clear flag.  Only the pixels at the upper-left rim of a face's
        //  pixels inside this wave's 16 x 64 block -- left neighbour and upper neighbour owned by someone else -- go to the atomic: ~5 per face instead of ~20)
#if SIL_COMPACT
        const int f_left = __shfl_up(f, 1, 64);
        const bool rim = (lane == 0 || f_left != f) && (r == 0 || fown[r > 0 ? r - 1 : 0] != f);
        if (seen[r] == 0 && rim && atomicExch(visible + (size_t)b * 2 * NF + f, 1) == 0) {
            const int k = atomicAdd(&sN, 1);
            if (k < LCAP) sList[k] = f; else vlist[(size_t)b * NF + atomicAdd(nvis + b, 1)] = f;       // (overflow of the tile's list: straight to the frame's)
        }
#else
#if !(SIL_IMG_ABL & 2)
        if (f >= 0) visible[(size_t)b * 2 * NF + f] = 1;                       // benign race: every writer stores 1 (sil_resolve_kernel)
#endif
#endif
        const float d = kp[r] * im - rf[r];                                    // sil_mask_loss_kernel
        acc += (double)(d * d);
        const float gd = 2.f * d * kp[r] * ob * gs;
        d_image[o] = gd;
        const bool p = f < 0 && gd < 0.f;                                      // sil_sweep_mask_kernel
        const unsigned long long m = __ballot(p);
        if (lane == 0 && !(SIL_IMG_ABL & 8)) rowmask[((size_t)b * is + yi) * wpl + tx] = m;
        col |= (unsigned)p << r;
    }
    sCol[wave][lane] = col;
    for (int o_ = 32; o_ > 0; o_ >>= 1) acc += __shfl_xor(acc, o_, 64);
    if (lane == 0) sSum[wave] = acc;
    __syncthreads();
    {   // the tile's newly seen faces -> the frame's list
        const int nl = min(sN, LCAP);
        if (threadIdx.x == 0 && nl > 0) sBase = atomicAdd(nvis + b, nl);
        __syncthreads();
        for (int i = threadIdx.x; i < nl; i += 256) vlist[(size_t)b * NF + sBase + i] = sList[i];
    }
    if (wave == 0)
        colmask[((size_t)b * is + xi) * wpl + ty] = (unsigned long long)sCol[0][lane] | ((unsigned long long)sCol[1][lane] << 16) |
                                                     ((unsigned long long)sCol[2][lane] << 32) | ((unsigned long long)sCol[3][lane] << 48);
    // the tile's share of the frame's mask sum: ONE fixed-point atomic.  (A first form closed the step here -- __threadfence + ticket, the last workgroup adding the
    // frames up: an agent-scope fence on a multi-XCD part writes the XCD's L2 back, and 1536 of them made this kernel 110 us instead of 26; the sums are now added
    // up by the first workgroup of sil_unproject_kernel, the step's last launch, behind the kernel boundary.)
    if (threadIdx.x == 0 && !(SIL_IMG_ABL & 1)) atomicAdd(cnt + b, (unsigned long long)__double2ll_rn((sSum[0] + sSum[1] + sSum[2] + sSum[3]) * SIL_FIX));
}

extern "C" int vt_sil_step(const float *verts, int B, int NV, const int *faces, int NF, const float *K, int size, const float *keep, const float *ref,
                           const float *occ, float gscale, float eps, double *term, int *face_index, float *d_image, float *ws, float *dverts, void *stream)
{
    VT_REQUIRE(verts && faces && K && keep && ref && occ && face_index && d_image && ws && dverts && B > 0 && NV > 0 && NF > 0 && size > 0 && size % 64 == 0 && size < 32768,
               "vt_sil_step: bad argument (size must be a multiple of 64)");
    hipStream_t st = vt_stream(stream); const int *skip = vt_skip_flag_of(st);
    const SilWs w = sil_ws(ws, B, NV, NF, size);
    hipLaunchKernelGGL(sil_setup_kernel, dim3((2 * NF + 255) / 256, B), dim3(256), 0, st, verts, K, faces, NV, NF, size, w.fc, w.fbox, w.visible, w.zbuf, w.cnt, w.ticket, w.nvis, w.ftick, skip);
    VT_LAUNCH_CHECK();
    hipLaunchKernelGGL(sil_scatter_kernel, dim3((NF + 4 * (64 / SIL_GS) - 1) / (4 * (64 / SIL_GS)), B), dim3(256), 0, st, w.fc, w.fbox, NF, size, w.zbuf, skip);
    VT_LAUNCH_CHECK();
    hipLaunchKernelGGL(sil_image_kernel, dim3((size / 64) * (size / 64), B), dim3(256), 0, st, w.zbuf, NV, NF, size, keep, ref, occ, B, gscale / (float)B, term, face_index,
                       d_image, (float *)nullptr, w.visible, w.rowmask, w.colmask, w.gproj, w.cnt, w.ticket, w.nvis, w.vlist, w.ftick, skip);
    VT_LAUNCH_CHECK();
    hipLaunchKernelGGL(sil_bwd_face_kernel, dim3((NF + 4 * (64 / SIL_G) - 1) / (4 * (64 / SIL_G)), B), dim3(256), 0, st, w.fc, w.visible, faces, NV, NF, size, face_index, d_image,
                       w.rowmask, w.colmask, eps, w.gproj, SIL_COMPACT ? w.nvis : (const int *)nullptr, SIL_COMPACT ? w.vlist : (const int *)nullptr, skip);
    VT_LAUNCH_CHECK();
    hipLaunchKernelGGL(sil_unproject_kernel, dim3((NV + 255) / 256, B), dim3(256), 0, st, verts, K, NV, w.gproj, dverts, w.cnt, occ, B, term, skip);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

extern "C" int vt_sil_forward(const float *verts, int B, int NV, const int *faces, int NF, const float *K, int size, float *image,
                              int *face_index, float *ws, void *stream)
{
    VT_REQUIRE(verts && faces && K && image && face_index && ws && B > 0 && NV > 0 && NF > 0 && size > 0 && size % TILE == 0 && size < 32768,
               "vt_sil_forward: bad argument (size must be a multiple of 64)");
    VT_REQUIRE(size % 64 == 0, "vt_sil_forward: size must be a multiple of 64");
    hipStream_t st = vt_stream(stream); const int *skip = vt_skip_flag_of(st);
    const SilWs w = sil_ws(ws, B, NV, NF, size);
    hipLaunchKernelGGL(sil_project_kernel, dim3((NV + 255) / 256, B), dim3(256), 0, st, verts, K, NV, w.proj, skip);
    VT_LAUNCH_CHECK();
    hipLaunchKernelGGL(sil_face_setup_kernel, dim3((2 * NF + 255) / 256, B), dim3(256), 0, st, w.proj, faces, NV, NF, size, w.fc, w.fbox, w.visible, skip);
    VT_LAUNCH_CHECK();
    VT_HIP(hipMemsetAsync(w.zbuf, 0xff, sizeof(unsigned long long) * (size_t)B * size * size, st));
    hipLaunchKernelGGL(sil_scatter_kernel, dim3((NF + 4 * (64 / SIL_GS) - 1) / (4 * (64 / SIL_GS)), B), dim3(256), 0, st, w.fc, w.fbox, NF, size, w.zbuf, skip);
    VT_LAUNCH_CHECK();
    hipLaunchKernelGGL(sil_resolve_kernel, dim3((size + 255) / 256, size, B), dim3(256), 0, st, w.zbuf, NF, size, image, face_index, w.visible, skip);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

extern "C" int vt_triplane_render(const float *verts, const float *center, int B, int NV, const int *faces, int NF, int size, float *masks,
                                  int *face_index, float *ws, void *stream)
{
    VT_REQUIRE(verts && center && faces && masks && face_index && ws && B > 0 && NV > 0 && NF > 0 && size > 0 && size % 64 == 0 && size < 32768,
               "vt_triplane_render: bad argument (size must be a multiple of 64)");
    hipStream_t st = vt_stream(stream);
    const int B3 = 3 * B;
    const SilWs w = sil_ws(ws, B3, NV, NF, size);
    hipLaunchKernelGGL(sil_triplane_project_kernel, dim3((NV + 255) / 256, B3), dim3(256), 0, st, verts, center, NV, 10.0f, w.proj);
    VT_LAUNCH_CHECK();
    hipLaunchKernelGGL(sil_face_setup_kernel, dim3((2 * NF + 255) / 256, B3), dim3(256), 0, st, w.proj, faces, NV, NF, size, w.fc, w.fbox, w.visible, nullptr);
    VT_LAUNCH_CHECK();
    VT_HIP(hipMemsetAsync(w.zbuf, 0xff, sizeof(unsigned long long) * (size_t)B3 * size * size, st));
    hipLaunchKernelGGL(sil_scatter_kernel, dim3((NF + 4 * (64 / SIL_GS) - 1) / (4 * (64 / SIL_GS)), B3), dim3(256), 0, st, w.fc, w.fbox, NF, size, w.zbuf, nullptr);
    VT_LAUNCH_CHECK();
    hipLaunchKernelGGL(sil_resolve_kernel, dim3((size + 255) / 256, size, B3), dim3(256), 0, st, w.zbuf, NF, size, masks, face_index, w.visible, nullptr);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

extern "C" int vt_sil_backward(const float *verts, int B, int NV, const int *faces, int NF, const float *K, int size, const int *face_index,
                               const float *d_image, float eps, float *ws, float *dverts, void *stream)
{
    VT_REQUIRE(verts && faces && K && face_index && d_image && ws && dverts && B > 0, "vt_sil_backward: bad argument");
    hipStream_t st = vt_stream(stream); const int *skip = vt_skip_flag_of(st);
    const SilWs w = sil_ws(ws, B, NV, NF, size);
    VT_HIP(hipMemsetAsync(w.gproj, 0, sizeof(double) * (size_t)B * NV * 2, st));
    hipLaunchKernelGGL(sil_sweep_mask_kernel, dim3((size / 64) * (size / 64), B), dim3(256), 0, st, face_index, d_image, size, w.rowmask, w.colmask, skip);
    VT_LAUNCH_CHECK();
    hipLaunchKernelGGL(sil_bwd_face_kernel, dim3((NF + 4 * (64 / SIL_G) - 1) / (4 * (64 / SIL_G)), B), dim3(256), 0, st, w.fc, w.visible, faces, NV, NF, size, face_index, d_image,
                       w.rowmask, w.colmask, eps, w.gproj, (const int *)nullptr, (const int *)nullptr, skip);
    VT_LAUNCH_CHECK();
    hipLaunchKernelGGL(sil_unproject_kernel, dim3((NV + 255) / 256, B), dim3(256), 0, st, verts, K, NV, w.gproj, dverts, (const unsigned long long *)nullptr, (const float *)nullptr, B, (double *)nullptr, skip);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

extern "C" int vt_sil_mask_loss(const float *image, const float *keep, const float *ref, const float *occ, int B, int size, float gscale,
                                double *term, float *per_frame, float *d_image, void *stream)
{
    VT_REQUIRE(image && keep && ref && occ && B > 0 && size > 0 && size % 2 == 0, "vt_sil_mask_loss: bad argument (size must be even)");
    VT_REQUIRE((((uintptr_t)image | (uintptr_t)keep | (uintptr_t)ref | (uintptr_t)d_image) & 15) == 0, "vt_sil_mask_loss: images must be 16-byte aligned");
    hipLaunchKernelGGL(sil_mask_loss_kernel, dim3(per_frame ? 1 : 16, B), dim3(256), 0, vt_stream(stream), image, keep, ref, occ, B, size * size, gscale / (float)B,
                       term, per_frame, d_image, vt_skip_flag_of(vt_stream(stream)));
    VT_LAUNCH_CHECK();
    return VT_OK;
}

// ---- SilLossROI's per-batch set-up (recon/obj_pose_roi.py:39-75 __init__, 111-181 to_original_bbox / compute_K_roi / cvt_masks; recon/bbox.py:26-48
// make_bbox_square; opt_utils.py:148-153 mask2bbox): object-mask bounding box -> square x (1 + expansion) -> ROIAlign(out, aligned, sampling_ratio = 0)
// crops of the object and the person mask, thresholded at 0.5 -> image_ref, keep mask; the square in full-image pixels -> normalised ROI intrinsics.
// Two launches, no host round trip.  The arithmetic is the host restatement's (vistracker_amd/silhouette.py: masks2bbox, make_bbox_square,
// roi_align_masks, compute_K_roi -- float64 in numpy / torch, the same operations in the same order here; no fused multiply-adds), because the crops are
// thresholded: a last-bit difference in a sample position can move a pixel of the reference mask.
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void sil_setup_box_kernel(const float *__restrict__ mask_o, int H, int W, const float *__restrict__ cc, double expansion, double scale,
                                                            double crop_size, double fxn, double fyn, double cxn, double cyn, double image_width,
                                                            double *__restrict__ boxes, float *__restrict__ K)
{
    __shared__ int red[4][256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float *m = mask_o + (size_t)b * H * W;
    int x0 = W, x1 = -1, y0 = H, y1 = -1;
    for (int i = tid; i < H * W; i += 256)
        // the reference's loader thresholds the uint8 image: (m * 255) truncated > 127 (opt_utils.mask2bbox via obj_pose_roi.py:179; silhouette.masks2bbox) --
        // identical to m > 0.5 for the 256 values k / 255, NOT for soft masks in (0.5, 128 / 255)
        if ((int)(m[i] * 255.0f) > 127) { const int y = i / W, x = i - y * W; x0 = min(x0, x); x1 = max(x1, x); y0 = min(y0, y); y1 = max(y1, y); }
    red[0][tid] = x0; red[1][tid] = x1; red[2][tid] = y0; red[3][tid] = y1;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { red[0][tid] = min(red[0][tid], red[0][tid + s]); red[1][tid] = max(red[1][tid], red[1][tid + s]);
                       red[2][tid] = min(red[2][tid], red[2][tid + s]); red[3][tid] = max(red[3][tid], red[3][tid + s]); }
        __syncthreads();
    }
    if (tid) return;
    double bx0, by0, bx1, by1;
    if (red[1][0] < 0) { bx0 = 50000.0; by0 = 50000.0; bx1 = -100.0; by1 = -100.0; }          // EMPTY_BBOX (opt_utils.py:148-153): no foreground pixel
    else { bx0 = red[0][0]; by0 = red[2][0]; bx1 = red[1][0] + 1; by1 = red[3][0] + 1; }
    const double w = bx1 - bx0, h = by1 - by0;
    const double c0 = bx0 + w / 2, c1 = by0 + h / 2;
    const double s = fmax(w, h) * (1 + expansion);
    const double sx = c0 - s / 2, sy = c1 - s / 2;
    boxes[4 * b + 0] = sx; boxes[4 * b + 1] = sy; boxes[4 * b + 2] = sx + s; boxes[4 * b + 3] = sy + s;
    // to_original_bbox: the square in pixels of the full image (the crop centre arrives as float32, as in the reference's loader)
    const float half = (float)(crop_size / 2.0);
    const double X = sx * scale + (double)(cc[2 * b] - half), Y = sy * scale + (double)(cc[2 * b + 1] - half), S = s * scale;
    float *k = K + 9 * b;
    k[0] = (float)(fxn * image_width / S); k[1] = 0.f; k[2] = (float)((cxn * image_width - X) / S);
    k[3] = 0.f; k[4] = (float)(fyn * image_width / S); k[5] = (float)((cyn * image_width - Y) / S);
    k[6] = 0.f; k[7] = 0.f; k[8] = 1.f;
}

__device__ __forceinline__ void roi_prep(double v, int n, int &lo, int &hi, double &l, bool &dead)
{
    dead = (v < -1.0) || (v > (double)n);
    v = fmax(v, 0.0);
    lo = (int)floor(v);
    const bool top = lo >= n - 1;
    if (top) { lo = n - 1; hi = lo; v = (double)lo; } else hi = lo + 1;
    l = v - (double)lo;
}
__global__ __launch_bounds__(256) void sil_setup_roi_kernel(const float *__restrict__ mask_h, const float *__restrict__ mask_o, int H, int W, const double *__restrict__ boxes,
                                                            int out, float *__restrict__ ref, float *__restrict__ keep)
{
    const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    if (p >= out * out) return;
    const int oy = p / out, ox = p - oy * out;
    const double x1 = boxes[4 * b], y1 = boxes[4 * b + 1], rw = boxes[4 * b + 2] - x1, rh = boxes[4 * b + 3] - y1;
    const int gw = (int)ceil(rw / out), gh = (int)ceil(rh / out);
    float vo = 0.f, vh = 0.f;
    if (gw > 0 && gh > 0) {
        const double bw = rw / out, bh = rh / out;
        const float *mo = mask_o + (size_t)b * H * W, *mh = mask_h + (size_t)b * H * W;
        double so = 0.0, sh = 0.0;
        for (int iy = 0; iy < gh; iy++) {
            const double y = ((double)oy * bh + (y1 - 0.5)) + ((double)iy + 0.5) * bh / (double)gh;
            int ylo, yhi; double ly; bool dy; roi_prep(y, H, ylo, yhi, ly, dy);
            for (int ix = 0; ix < gw; ix++) {
                const double x = ((double)ox * bw + (x1 - 0.5)) + ((double)ix + 0.5) * bw / (double)gw;
                int xlo, xhi; double lx; bool dx; roi_prep(x, W, xlo, xhi, lx, dx);
                const double w00 = (1 - ly) * (1 - lx), w01 = (1 - ly) * lx, w10 = ly * (1 - lx), w11 = ly * lx;
                const double live = (dy ? 0.0 : 1.0) * (dx ? 0.0 : 1.0);
                so += ((((double)mo[ylo * W + xlo] * w00 + (double)mo[ylo * W + xhi] * w01) + (double)mo[yhi * W + xlo] * w10) + (double)mo[yhi * W + xhi] * w11) * live;
                sh += ((((double)mh[ylo * W + xlo] * w00 + (double)mh[ylo * W + xhi] * w01) + (double)mh[yhi * W + xlo] * w10) + (double)mh[yhi * W + xhi] * w11) * live;
            }
        }
        vo = (float)(so / (double)(gh * gw)); vh = (float)(sh / (double)(gh * gw));
    }
    const bool obj = vo >= 0.5f, ps = vh >= 0.5f;
    ref[(size_t)b * out * out + p] = obj ? 1.f : 0.f;                   // image_ref = (object crop > 0)
    keep[(size_t)b * out * out + p] = (ps && !obj) ? 0.f : 1.f;         // cvt_masks: keep foreground and free background, drop person-only pixels
}

extern "C" int vt_sil_setup(const float *mask_h, const float *mask_o, int B, int H, int W, const float *crop_centers, double expansion, int out, double crop_size,
                            double net_size, const double *cam_norm, double image_width, float *image_ref, float *keep_mask, float *K, double *boxes_ws, void *stream)
{
    VT_REQUIRE(mask_h && mask_o && crop_centers && cam_norm && image_ref && keep_mask && K && boxes_ws && B > 0 && H > 0 && W > 0 && out > 0 && net_size > 0,
               "vt_sil_setup: bad argument");
    hipStream_t st = vt_stream(stream);
    hipLaunchKernelGGL(sil_setup_box_kernel, dim3(B), dim3(256), 0, st, mask_o, H, W, crop_centers, expansion, crop_size / net_size, crop_size, cam_norm[0], cam_norm[1],
                       cam_norm[2], cam_norm[3], image_width, boxes_ws, K);
    VT_LAUNCH_CHECK();
    hipLaunchKernelGGL(sil_setup_roi_kernel, dim3((out * out + 255) / 256, B), dim3(256), 0, st, mask_h, mask_o, H, W, boxes_ws, out, image_ref, keep_mask);
    VT_LAUNCH_CHECK();
    return VT_OK;
}
