// smplh.hip -- SMPL-H forward / backward for gfx950.
//
// Replaces SMPL_Layer.forward (lib_smpl/smplpytorch/smplpytorch/pytorch/smpl_layer.py:73-176) and the autograd
// backward the reference gets from ~15k eager ops per call.  Layout decisions (DESIGN.md "SMPL-H"):
//   * blend shapes are ONE matrix: v_posed = [pose_map(459) | betas(10) | 1] . Q with Q = [posedirs ; shapedirs ; template]
//     (472 rows after zero padding).  It is kept in two layouts so that both GEMMs read row-contiguous MFMA B fragments:
//       Q_kcv [472][3][VP]   forward  (M = frames, N = vertices, K = 472)
//       Q_t   [1296 K groups][30 N tiles][64 lanes][4]   backward split-K GEMM (M = frames, N = 472 blend columns, K = vertex
//             coordinates), stored in MFMA fragment order so that one float4 per lane feeds 4 K steps: d[pose_map | betas]
//     both run on v_mfma_f32_16x16x4_f32.  weights -> W_jv [52][VP].  VP = 6912 = 27 * 256 (zero padded).
//   * the joint regressor is folded on the host: J = J_t + J_s . beta (J = Jreg . v_shaped is linear in beta),
//     which removes the 52x6890 reduction from every call.
//   * forward = 2 launches (pose/chain, vertices); backward = 2 launches (vertex tile partials, per-frame
//     reduce + chain VJP).  Reductions are two-stage and deterministic (no float atomics).
#include "common.h"
#include <cstring>

#define V_ VT_SMPL_V
#define J_ VT_SMPL_J
#define NB_ VT_SMPL_NB
#define NP_ VT_SMPL_NP
#define VP_ 6912
#define NVT_ 27 /* vertex tiles of 256 */
#define KQ_ 472 /* 459 pose-map + 10 betas + 1 template + 2 zero rows */
#define NQ_ 480 /* padded row length of Q_t (30 N-tiles) */
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// per-frame workspace layout (floats)
#define WS_R 0
#define WS_J (52 * 9)
#define WS_G (WS_J + 52 * 3)
#define WS_A (WS_G + 52 * 12)
#define WS_FRAME (WS_A + 52 * 12) /* 1872 */
// per (tile, frame) partial layout of the backward
#define PT_DA 0
#define PT_DTR 624
#define PT_N 627
#define KTOT_ (VP_ * 3)              /* 20736 rows of the backward blend-shape GEMM */
#define KG_SLAB 24                   /* 16-row K groups per split-K slab (384 rows) */
#define NKS_ (KTOT_ / 16 / KG_SLAB)  /* 54 slabs */

// parents + the reverse-chain schedule of smplh_bwd_frame_kernel (built once in vt_smplh_create): step t pushes the joints sched[t][0..9] (-1: none) into their
// parents at the same time -- deepest tree level first, inside a level by sibling rank, so that the joints of a step have different parents, a parent receives its
// children in descending index order and every joint is complete before it is pushed; nsteps = 0: no schedule (a tree outside its limits), the serial loop runs
#define SCHED_STEPS 32
#define SCHED_WIDTH 10
struct SmplParents { int p[J_]; int nsteps; short sched[SCHED_STEPS][SCHED_WIDTH]; };

#define SP_K 8                          /* most non-zero skinning weights per vertex the sparse LBS takes (SMPL / SMPL-H: 4) */
struct vt_smplh {
    float *Q_kcv, *Q_t, *W_jv, *W_v64, *J_t, *J_s;
    float *W_sp;                        // [2][SP_K][VP]: joint index (as int bits) | weight of the k-th non-zero of a vertex, ascending joints, zero padded
    int nnz;                            // max non-zeros per vertex if <= SP_K (sparse LBS), else 0 (dense LBS)
    SmplParents par;
};

// ---------------------------------------------------------------------------------------------------
// batch_rodrigues (rodrigues_layer.py:13-52).  NB the norm is of (theta + 1e-8), the division uses theta.
// ---------------------------------------------------------------------------------------------------
struct RodCtx { float n, ax, ay, az, c, s, qn, w, x, y, z; };

__device__ __forceinline__ void rodrigues_fwd(const float *t, float *R, RodCtx *k)
{
    const float xe = t[0] + 1e-8f, ye = t[1] + 1e-8f, ze = t[2] + 1e-8f;
    const float n = sqrtf(xe * xe + ye * ye + ze * ze);
    const float ax = t[0] / n, ay = t[1] / n, az = t[2] / n;
    const float h = n * 0.5f;
    const float c = cosf(h), s = sinf(h);
    const float q0 = c, q1 = s * ax, q2 = s * ay, q3 = s * az;
    const float qn = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
    const float w = q0 / qn, x = q1 / qn, y = q2 / qn, z = q3 / qn;
    const float w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z;
    const float wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
    R[0] = w2 + x2 - y2 - z2; R[1] = 2 * xy - 2 * wz;   R[2] = 2 * wy + 2 * xz;
    R[3] = 2 * wz + 2 * xy;   R[4] = w2 - x2 + y2 - z2; R[5] = 2 * yz - 2 * wx;
    R[6] = 2 * xz - 2 * wy;   R[7] = 2 * wx + 2 * yz;   R[8] = w2 - x2 - y2 + z2;
    if (k) { k->n = n; k->ax = ax; k->ay = ay; k->az = az; k->c = c; k->s = s; k->qn = qn; k->w = w; k->x = x; k->y = y; k->z = z; }
}

__device__ __forceinline__ void rodrigues_bwd(const float *t, const float *dR, float *dt)
{
    float R[9]; RodCtx k; rodrigues_fwd(t, R, &k);
    const float w = k.w, x = k.x, y = k.y, z = k.z;
    const float dw = 2 * w * (dR[0] + dR[4] + dR[8]) + 2 * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
    const float dx = 2 * x * (dR[0] - dR[4] - dR[8]) + 2 * (y * dR[1] + z * dR[2] + y * dR[3] - w * dR[5] + z * dR[6] + w * dR[7]);
    const float dy = 2 * y * (-dR[0] + dR[4] - dR[8]) + 2 * (x * dR[1] + w * dR[2] + x * dR[3] + z * dR[5] - w * dR[6] + z * dR[7]);
    const float dz = 2 * z * (-dR[0] - dR[4] + dR[8]) + 2 * (-w * dR[1] + x * dR[2] + w * dR[3] + y * dR[5] + x * dR[6] + y * dR[7]);
    const float dot = w * dw + x * dx + y * dy + z * dz;
    const float dq0 = (dw - w * dot) / k.qn, dq1 = (dx - x * dot) / k.qn, dq2 = (dy - y * dot) / k.qn, dq3 = (dz - z * dot) / k.qn;
    const float dh = -k.s * dq0 + k.c * (k.ax * dq1 + k.ay * dq2 + k.az * dq3);
    const float dax = k.s * dq1, day = k.s * dq2, daz = k.s * dq3;
    const float dn = 0.5f * dh - (t[0] * dax + t[1] * day + t[2] * daz) / (k.n * k.n);
    dt[0] = dax / k.n + dn * (t[0] + 1e-8f) / k.n;
    dt[1] = day / k.n + dn * (t[1] + 1e-8f) / k.n;
    dt[2] = daz / k.n + dn * (t[2] + 1e-8f) / k.n;
}

__global__ void rodrigues_fwd_kernel(const float *aa, int n, float *R)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float t[3] = {aa[3 * i], aa[3 * i + 1], aa[3 * i + 2]}, r[9];
    rodrigues_fwd(t, r, nullptr);
#pragma unroll
    for (int e = 0; e < 9; e++) R[9 * i + e] = r[e];
}

__global__ void rodrigues_bwd_kernel(const float *aa, int n, const float *dR, float *daa)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float t[3] = {aa[3 * i], aa[3 * i + 1], aa[3 * i + 2]}, g[9], d[3];
#pragma unroll
    for (int e = 0; e < 9; e++) g[e] = dR[9 * i + e];
    rodrigues_bwd(t, g, d);
    daa[3 * i] = d[0]; daa[3 * i + 1] = d[1]; daa[3 * i + 2] = d[2];
}

// ---------------------------------------------------------------------------------------------------
// forward, launch 1: one wave per frame -- rotations, joints, kinematic chain (smpl_layer.py:88-143)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void smplh_pose_kernel(const float *__restrict__ pose, const float *__restrict__ betas,
                                                        const float *__restrict__ trans, const float *__restrict__ J_t,
                                                        const float *__restrict__ J_s, SmplParents par,
                                                        float *__restrict__ ws, float *__restrict__ jtr, const int *skip)
{
    VT_SKIP_RETURN(skip);
    __shared__ float sR[J_ * 9], sJ[J_ * 3], sG[J_ * 12];
    const int b = blockIdx.x, j = threadIdx.x;
    float *w = ws + (size_t)b * WS_FRAME;
    if (j < J_) {
        float t[3] = {pose[(b * J_ + j) * 3], pose[(b * J_ + j) * 3 + 1], pose[(b * J_ + j) * 3 + 2]}, R[9];
        rodrigues_fwd(t, R, nullptr);
#pragma unroll
        for (int e = 0; e < 9; e++) { sR[j * 9 + e] = R[e]; w[WS_R + j * 9 + e] = R[e]; }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float a = J_t[j * 3 + c];
#pragma unroll
            for (int l = 0; l < NB_; l++) a += J_s[(j * 3 + c) * NB_ + l] * betas[b * NB_ + l];
            sJ[j * 3 + c] = a; w[WS_J + j * 3 + c] = a;
        }
    }
    // G_0 = [R_0 | J_0];  G_i = G_parent . [R_i | J_i - J_parent].  Round 6: the chain by tree LEVEL (SMPL-H: 11 levels for 52 joints), lane = joint, instead of
    // one thread walking the 51 joints in turn (15 us of dependent LDS round trips): every joint's twelve values are the same expressions -- bit-identical.
    __shared__ int sPar[J_];
    if (j < J_) sPar[j] = par.p[j];
    __syncthreads();
    int lvl = 0, maxl = 0;
    if (j < J_) { for (int q = j; q != 0; q = sPar[q]) lvl++; }
    maxl = lvl;
    for (int o = 32; o > 0; o >>= 1) maxl = max(maxl, __shfl_xor(maxl, o, 64));
    maxl = __builtin_amdgcn_readfirstlane(maxl);
    if (j == 0) { for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) sG[r * 4 + c] = sR[r * 3 + c]; sG[r * 4 + 3] = sJ[r]; } }
    __syncthreads();
    for (int l = 1; l <= maxl; l++) {
        if (j < J_ && lvl == l) {
            const int i = j, p = sPar[i];
            const float *Gp = sG + 12 * p, *Ri = sR + 9 * i;
            const float d0 = sJ[3 * i] - sJ[3 * p], d1 = sJ[3 * i + 1] - sJ[3 * p + 1], d2 = sJ[3 * i + 2] - sJ[3 * p + 2];
            float *Gi = sG + 12 * i;
            for (int r = 0; r < 3; r++) {
                const float g0 = Gp[r * 4], g1 = Gp[r * 4 + 1], g2 = Gp[r * 4 + 2];
                Gi[r * 4 + 0] = g0 * Ri[0] + g1 * Ri[3] + g2 * Ri[6];
                Gi[r * 4 + 1] = g0 * Ri[1] + g1 * Ri[4] + g2 * Ri[7];
                Gi[r * 4 + 2] = g0 * Ri[2] + g1 * Ri[5] + g2 * Ri[8];
                Gi[r * 4 + 3] = g0 * d0 + g1 * d1 + g2 * d2 + Gp[r * 4 + 3];
            }
        }
        __syncthreads();
    }
    if (j < J_) {
        // A_j = G_j - [0 | G_rot J_j]   (th_results2, smpl_layer.py:133-143);  jtr = G_t + trans (:154,172)
        const float *G = sG + 12 * j;
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const float gt = G[r * 4 + 3];
            w[WS_G + j * 12 + r * 4 + 0] = G[r * 4]; w[WS_G + j * 12 + r * 4 + 1] = G[r * 4 + 1];
            w[WS_G + j * 12 + r * 4 + 2] = G[r * 4 + 2]; w[WS_G + j * 12 + r * 4 + 3] = gt;
            w[WS_A + j * 12 + r * 4 + 0] = G[r * 4]; w[WS_A + j * 12 + r * 4 + 1] = G[r * 4 + 1]; w[WS_A + j * 12 + r * 4 + 2] = G[r * 4 + 2];
            w[WS_A + j * 12 + r * 4 + 3] = gt - (G[r * 4] * sJ[3 * j] + G[r * 4 + 1] * sJ[3 * j + 1] + G[r * 4 + 2] * sJ[3 * j + 2]);
            if (jtr) jtr[(b * J_ + j) * 3 + r] = gt + trans[3 * b + r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// forward, launch 2: blend shapes as an MFMA GEMM + LBS (smpl_layer.py:103-112,145-173).
// Workgroup = 64 vertices x 16 frames; wave w owns vertices 16w..16w+15.  D layout: lane (q = lane>>4, j = lane&15) holds
// v_posed of vertex j for frames 4q..4q+3 and all three coordinates, so skinning continues in-lane (dense 52-joint LBS with
// the frame's A matrices broadcast from LDS).
// ---------------------------------------------------------------------------------------------------
#define FWD_FB 16
#ifndef FWD_PF
#define FWD_PF 6         /* pair steps of Q rows in flight in the blend-shape GEMM of smplh_verts_kernel */
#endif
#define AXS 516   /* LDS stride of an extended pose row [pose_map | betas | 1 | 0 0]: 516 mod 64 = 4 -> conflict-free ds_read_b64 */
#define SAS 628   /* LDS stride of a frame's 52 x 12 A matrices */
#define FWD_R0 (FWD_FB * AXS > 8 * SAS ? FWD_FB * AXS : 8 * SAS)    /* floats of the time-shared region: pose rows, then 8 frames of A matrices */
__global__ __launch_bounds__(256) void smplh_verts_kernel(const float *__restrict__ Q_kcv, const float *__restrict__ W_jv,
                                                          const float *__restrict__ betas, const float *__restrict__ trans,
                                                          const float *__restrict__ ws, int B,
                                                          float *__restrict__ verts, float *__restrict__ v_posed, const float *__restrict__ W_sp, int nnz, const int *skip)
{
    VT_SKIP_RETURN(skip);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *sAx = lds;                       // [16][AXS] extended pose rows (GEMM phase) ...
    float *sA = lds;                        // ... then [8][SAS] A matrices of HALF the frames at a time (skinning phase)
    float *sVp = lds + FWD_R0;              // [16][64][3] v_posed hand-over
    float *sW = sVp + FWD_FB * 64 * 3;      // [52][64] skinning weights of the tile, or [2][SP_K][64] in the sparse form
    float *sTr = sW + (nnz > 0 ? 2 * SP_K * 64 : J_ * 64);     // [16][3]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
    const int b0 = blockIdx.y * FWD_FB, v = blockIdx.x * 64 + wave * 16 + j;
    // extended pose rows [pose_map | betas | 1 | 0 0] of the block's frames: a thread owns a column k and requests it for all FWD_FB frames at once (round 6: the
    // element-per-thread loop was 30 rounds of one dependent load each, with a division and a modulo per element -- a third of the kernel's time)
    for (int k = tid; k < KQ_; k += 256) {
        const int e9 = k % 9;
        const float sub = (k < NP_ && (e9 == 0 || e9 == 4 || e9 == 8)) ? 1.f : 0.f;
        float val[FWD_FB];
#pragma unroll
        for (int f = 0; f < FWD_FB; f++) {
            const int b = min(b0 + f, B - 1);
            if (k < NP_) val[f] = ws[(size_t)b * WS_FRAME + WS_R + 9 + k];
            else if (k < NP_ + NB_) val[f] = betas[b * NB_ + (k - NP_)];
            else val[f] = (k == NP_ + NB_) ? 1.f : 0.f;
        }
#pragma unroll
        for (int f = 0; f < FWD_FB; f++) sAx[f * AXS + k] = val[f] - sub;
    }
    if (nnz > 0) {                          // sparse LBS: sW = [nnz][64] joint indices | [nnz][64] weights of the tile
        for (int i = tid; i < nnz * 64; i += 256) {
            sW[i] = W_sp[(size_t)(i >> 6) * VP_ + blockIdx.x * 64 + (i & 63)];
            sW[SP_K * 64 + i] = W_sp[(size_t)(SP_K + (i >> 6)) * VP_ + blockIdx.x * 64 + (i & 63)];
        }
    } else
        for (int i = tid; i < J_ * 64; i += 256) sW[i] = W_jv[(i >> 6) * VP_ + blockIdx.x * 64 + (i & 63)];
    if (tid < FWD_FB * 3) { const int f = tid / 3, b = min(b0 + f, B - 1); sTr[tid] = trans[b * 3 + tid % 3]; }
    __syncthreads();

    // v_posed[f][v][c] = sum_k Aext[f][k] Q[k][c][v]:  A = pose rows (LDS), B = Q rows (global, 64-B segments), 59 pair steps
    f32x4 acc[3];
#pragma unroll
    for (int c = 0; c < 3; c++) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // Round 6: the Q rows are requested FWD_PF pair steps ahead of their use (a ring of register sets, the loop fully unrolled so that the ring indices are
    // constants).  With the loads of a step issued in front of its own MFMAs (unroll 2) the loop ran one round trip to L2 / the fabric per two steps:
    // ~30 round trips = the kernel's 50 us.  Same MFMAs in the same order: bit-identical.
    constexpr int NS = KQ_ / 8;
    float bq[FWD_PF][2][3];
#define FWD_LOADQ(slot_, s_)                                                                     \
    {                                                                                            \
        const float *qk = Q_kcv + (size_t)(8 * (s_) + 2 * q) * 3 * VP_ + v;                      \
        _Pragma("unroll") for (int e = 0; e < 2; e++)                                            \
            _Pragma("unroll") for (int c = 0; c < 3; c++) bq[slot_][e][c] = qk[(size_t)(e * 3 + c) * VP_]; \
    }
#pragma unroll
    for (int s = 0; s < FWD_PF; s++) FWD_LOADQ(s, s)
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const float2 av = *reinterpret_cast<const float2 *>(sAx + j * AXS + 8 * s + 2 * q);
#pragma unroll
        for (int c = 0; c < 3; c++) acc[c] = MFMA16(av.x, bq[s % FWD_PF][0][c], acc[c]);
#pragma unroll
        for (int c = 0; c < 3; c++) acc[c] = MFMA16(av.y, bq[s % FWD_PF][1][c], acc[c]);
        if (s + FWD_PF < NS) FWD_LOADQ(s % FWD_PF, s + FWD_PF)
        __builtin_amdgcn_sched_barrier(0);      // keeps the request where it is written: the scheduler otherwise sinks it to ~1.5 steps in front of its use
    }
#undef FWD_LOADQ
    // hand v_posed over through LDS so that skinning runs with thread == vertex and a wave-uniform frame: the frame's A
    // matrices are then true LDS broadcasts and nothing is indexed dynamically in registers
    __syncthreads();                        // every wave is done with sAx: the region becomes sA
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) sVp[((4 * q + r) * 64 + wave * 16 + j) * 3 + c] = acc[c][r];
    const int vv = tid & 63, vg = blockIdx.x * 64 + vv;
    // The A matrices of the 16 frames (40 KB) are staged in two halves of 8 (slot 2 w + e <- frame 4 w + 2 half + e): with the sparse weights the
    // workgroup then needs 49.5 KB of LDS and THREE fit a CU -- the 648 workgroups of a B = 96 launch run in one round of 768 slots instead of
    // 1.27 rounds of 512.
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
        if ((r & 1) == 0) {
            if (r) __syncthreads();             // the readers of the first half are done
            for (int i = tid; i < 8 * 624; i += 256) {
                const int sl = i / 624, f = 4 * (sl >> 1) + r + (sl & 1), b = min(b0 + f, B - 1);
                sA[sl * SAS + (i % 624)] = ws[(size_t)b * WS_FRAME + WS_A + (i % 624)];
            }
            __syncthreads();
        }
        const int f = 4 * wave + r, b = b0 + f;
        float T[12];
#pragma unroll
        for (int e = 0; e < 12; e++) T[e] = 0.f;
        const float4 *Af = reinterpret_cast<const float4 *>(sA + (2 * wave + (r & 1)) * SAS);
        // T = sum_j w_j A_j in ascending joint order.  The dense sum spends 52 x 12 multiply-adds per (vertex, frame) on 4 non-zero weights --
        // two thirds of this kernel's arithmetic; the sparse sum adds the same non-zero terms in the same order (a skipped term is w_j A_j = +-0
        // and leaves the partial sum unchanged), so both give identical bits.
        if (nnz > 0) {
            for (int k = 0; k < nnz; k++) {
                const int jj = __float_as_int(sW[k * 64 + vv]);
                const float wj = sW[(SP_K + k) * 64 + vv];
                const float4 a0 = Af[jj * 3], a1 = Af[jj * 3 + 1], a2 = Af[jj * 3 + 2];
                T[0] += wj * a0.x; T[1] += wj * a0.y; T[2] += wj * a0.z; T[3] += wj * a0.w;
                T[4] += wj * a1.x; T[5] += wj * a1.y; T[6] += wj * a1.z; T[7] += wj * a1.w;
                T[8] += wj * a2.x; T[9] += wj * a2.y; T[10] += wj * a2.z; T[11] += wj * a2.w;
            }
        } else {
#pragma unroll 4
            for (int jj = 0; jj < J_; jj++) {
                const float wj = sW[jj * 64 + vv];
                const float4 a0 = Af[jj * 3], a1 = Af[jj * 3 + 1], a2 = Af[jj * 3 + 2];
                T[0] += wj * a0.x; T[1] += wj * a0.y; T[2] += wj * a0.z; T[3] += wj * a0.w;
                T[4] += wj * a1.x; T[5] += wj * a1.y; T[6] += wj * a1.z; T[7] += wj * a1.w;
                T[8] += wj * a2.x; T[9] += wj * a2.y; T[10] += wj * a2.z; T[11] += wj * a2.w;
            }
        }
        if (b < B && vg < V_) {
            const float p0 = sVp[(f * 64 + vv) * 3], p1 = sVp[(f * 64 + vv) * 3 + 1], p2 = sVp[(f * 64 + vv) * 3 + 2];
            const size_t o = ((size_t)b * V_ + vg) * 3;
#pragma unroll
            for (int rr = 0; rr < 3; rr++) verts[o + rr] = T[rr * 4] * p0 + T[rr * 4 + 1] * p1 + T[rr * 4 + 2] * p2 + T[rr * 4 + 3] + sTr[f * 3 + rr];
            v_posed[o] = p0; v_posed[o + 1] = p1; v_posed[o + 2] = p2;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// backward, launch 1: per (256-vertex tile, FB frames): partial sums of dA (52x12) and dtrans (3); d v_posed to global.
//   phase 1 (MFMA, per frame):  T[v][e] = sum_j W[v][j] A_j[e]   M = the wave's 64 vertices, N = 12 (->16), K = 52 joints;
//                               d v_posed[v] = T_rot^T dv[v] -> dvp_g [B][20736] (the A operand of launch 2)
//   phase 2 (MFMA, per frame):  dA[e][j] = sum_v dT[v][e] W[v][j]   M = 12 (->16), N = 52 joints + a ones column (dtrans) (->64), K = 256
//   with dT[v] = dv[v] (x) [v_posed[v]; 1] staged in LDS by thread == vertex.  The weight fragments of both phases are frame
//   independent and live in registers (116 VGPRs) across the FB frames.
// ---------------------------------------------------------------------------------------------------
template <int FB>
__global__ __launch_bounds__(256) void smplh_bwd_tile_kernel(const float *__restrict__ W_v64, const float *__restrict__ ws,
                                                             const float *__restrict__ v_posed, const float *__restrict__ dverts,
                                                             int B, float *__restrict__ part, float *__restrict__ dvp_g, const int *skip)
{
    VT_SKIP_RETURN(skip);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *sdT = lds;                        // [256][13]  dT rows of the frame in flight: dv (x) [v_posed; 1]
    float *sA = sdT + 256 * 13;              // [FB][624]  skinning transforms A_j (3x4) of the block's frames
    const int tid = threadIdx.x, tile = blockIdx.x, b0 = blockIdx.y * FB, v0 = tile * 256, v = v0 + tid;
    const int wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
    for (int i = tid; i < FB * 624; i += 256) { const int f = i / 624, b = min(b0 + f, B - 1); sA[i] = ws[(size_t)b * WS_FRAME + WS_A + (i % 624)]; }
    // Both MFMA phases read the skinning weights of this tile only; they do not depend on the frame, so each lane keeps its
    // fragments in registers for all FB frames:
    //   w1[mt][ks] = W[v0 + 64 wave + 16 mt + j][4 ks + q]   (phase 1, A operand: M = the wave's own 64 vertices, K = joints)
    //   w2[ks]     = W[v0 + 4 ks + q][16 wave + j]           (phase 2, B operand: K = the tile's 256 vertices, N = joints)
    float w1[4][13], w2[64];
#pragma unroll
    for (int mt = 0; mt < 4; mt++)
#pragma unroll
        for (int ks = 0; ks < 13; ks++) w1[mt][ks] = W_v64[(size_t)(v0 + 64 * wave + 16 * mt + j) * 64 + 4 * ks + q];
#pragma unroll
    for (int ks = 0; ks < 64; ks++) w2[ks] = W_v64[(size_t)(v0 + 4 * ks + q) * 64 + wave * 16 + j];
    // (phase 1's K = 52 joints is exactly 13 steps: the all-ones column 52 of W_v64, phase 2's dtrans collector, is never touched)
    __syncthreads();
    // the vertex's gradient and rest position of frame f + 1 are requested before the MFMA phases of frame f (round 6: a rolled loop paid their round trip
    // in front of every frame's LDS staging)
    float dvn[3] = {0.f, 0.f, 0.f}, vpn[3] = {0.f, 0.f, 0.f};
    auto fetch = [&](int b) {
#pragma unroll
        for (int r = 0; r < 3; r++) { dvn[r] = 0.f; vpn[r] = 0.f; }
        if (b < B && v < V_) {
            const size_t o = ((size_t)b * V_ + v) * 3;
            dvn[0] = dverts[o]; dvn[1] = dverts[o + 1]; dvn[2] = dverts[o + 2];
            vpn[0] = v_posed[o]; vpn[1] = v_posed[o + 1]; vpn[2] = v_posed[o + 2];
        }
    };
    fetch(b0);
#pragma unroll 1
    for (int f = 0; f < FB; f++) {
        const int b = b0 + f;
        const float dv[3] = {dvn[0], dvn[1], dvn[2]}, vp[3] = {vpn[0], vpn[1], vpn[2]};
        if (f + 1 < FB) fetch(b + 1);
#pragma unroll
        for (int r = 0; r < 3; r++) {
            sdT[tid * 13 + r * 4 + 0] = dv[r] * vp[0]; sdT[tid * 13 + r * 4 + 1] = dv[r] * vp[1];
            sdT[tid * 13 + r * 4 + 2] = dv[r] * vp[2]; sdT[tid * 13 + r * 4 + 3] = dv[r];
        }
        __syncthreads();
        // phase 1 (MFMA): T[v][e] = sum_j W[v][j] A_j[e]  (e = 4 r + c), then d v_posed[v][c] = sum_r T[v][4r + c] dv[v][r]:
        // lane (q, e) of M-tile mt holds T for vertices 16 mt + 4 q + reg; it scales by dv[.][e >> 2] (column 3 of the dT row)
        // and the three r-lanes of a column c are summed with two in-row shuffles.
        {
            f32x4 t[4];
#pragma unroll
            for (int mt = 0; mt < 4; mt++) t[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 13; ks++) {
                const float bv = j < 12 ? sA[f * 624 + (4 * ks + q) * 12 + j] : 0.f;
#pragma unroll
                for (int mt = 0; mt < 4; mt++) t[mt] = MFMA16(w1[mt][ks], bv, t[mt]);
            }
            const int r4 = (j < 12 ? (j >> 2) : 0) * 4 + 3;
#pragma unroll
            for (int mt = 0; mt < 4; mt++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int vl = 64 * wave + 16 * mt + 4 * q + r;
                    float x = t[mt][r] * sdT[vl * 13 + r4];
                    x += __shfl_down(x, 4, 16) + __shfl_down(x, 8, 16);
                    if (j < 3 && b < B) dvp_g[(size_t)b * KTOT_ + (size_t)(v0 + vl) * 3 + j] = x;   // zero on the padding vertices
                }
            }
        }
        // phase 2 (MFMA): dA[e][jn] = sum_v dT[v][e] W[v][jn]; wave w owns joints 16w..16w+15
        {
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, acc2 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 64; ks += 2) {
                const int va = 4 * ks + q, vb = va + 4;
                const float a0 = j < 12 ? sdT[va * 13 + j] : 0.f, a1 = j < 12 ? sdT[vb * 13 + j] : 0.f;
                acc = MFMA16(a0, w2[ks], acc);
                acc2 = MFMA16(a1, w2[ks + 1], acc2);
            }
            acc += acc2;
            if (b < B) {
                float *dst = part + ((size_t)tile * B + b) * PT_N;
                const int jn = wave * 16 + j;                // joint (column of D); rows of D = e = 4q + r
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int e = 4 * q + r;
                    if (e < 12) {
                        if (jn < J_) dst[PT_DA + jn * 12 + e] = acc[r];
                        else if (jn == J_ && (e & 3) == 3) dst[PT_DTR + (e >> 2)] = acc[r];
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// backward, launch 2: d[pose_map | betas][b][n] = sum_k dvp[b][k] Q[k][n] as one split-K fp32 MFMA GEMM over the whole batch:
// M = frames (96 per block: 6 M-tiles), N = 480 (4 column blocks of 8 N-tiles, 2 per wave), K = 20736 in 54 slabs of 384 rows.
// Within a 16-row K group lane (j,q) takes rows 16g + 4q + s for MFMA step s on BOTH operands (a sum over K does not care
// about the order), so one float4 feeds 4 steps: A from the LDS-staged dvp rows, B from Q packed in exactly that fragment order.
// ---------------------------------------------------------------------------------------------------
#define BL_M 96
#ifndef BL_NW
#define BL_NW 1
#endif
#define BL_AS 132   /* LDS row stride of the staged A chunk (128 k + 4: stride = 4 mod 64 banks) */
__global__ __launch_bounds__(256) void smplh_bwd_blend_kernel(const float *__restrict__ Q_p, const float *__restrict__ dvp_g, int B,
                                                              float *__restrict__ part3, const int *skip)
{
    VT_SKIP_RETURN(skip);
    __shared__ __attribute__((aligned(16))) float sA[BL_M * BL_AS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
    // BL_NW = N-tiles per wave (round 6: 1, was 2): 216 workgroups of 1152 dependent-issue MFMAs per wave left the chip three quarters empty for 18 us of
    // matrix-pipe latency; 432 workgroups of 576 (the staged A chunk is read by twice as many workgroups, from L2)
    const int ks = blockIdx.x, m0 = blockIdx.z * BL_M, nt0 = blockIdx.y * (4 * BL_NW) + wave * BL_NW;
    f32x4 acc[6][BL_NW];
#pragma unroll
    for (int mt = 0; mt < 6; mt++)
#pragma unroll
        for (int n = 0; n < BL_NW; n++) acc[mt][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float4 *Q4 = reinterpret_cast<const float4 *>(Q_p);
#pragma unroll 1
    for (int c = 0; c < KG_SLAB / 8; c++) {
        __syncthreads();
        for (int i = tid; i < BL_M * 32; i += 256) {
            const int row = i >> 5, c4 = i & 31, b = m0 + row;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b < B) v = *reinterpret_cast<const float4 *>(dvp_g + (size_t)b * KTOT_ + (size_t)(ks * KG_SLAB + c * 8) * 16 + c4 * 4);
            *reinterpret_cast<float4 *>(sA + row * BL_AS + c4 * 4) = v;
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 8; g++) {
            const size_t kg = (size_t)ks * KG_SLAB + c * 8 + g;
            float bq[BL_NW][4];
#pragma unroll
            for (int n = 0; n < BL_NW; n++) {
                const float4 fq = Q4[(kg * (NQ_ / 16) + min(nt0 + n, NQ_ / 16 - 1)) * 64 + lane];
                bq[n][0] = fq.x; bq[n][1] = fq.y; bq[n][2] = fq.z; bq[n][3] = fq.w;
            }
#pragma unroll
            for (int mt = 0; mt < 6; mt++) {
                const float4 a4 = *reinterpret_cast<const float4 *>(sA + (mt * 16 + j) * BL_AS + g * 16 + q * 4);
                const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int n = 0; n < BL_NW; n++) acc[mt][n] = MFMA16(av[s], bq[n][s], acc[mt][n]);
            }
        }
    }
#pragma unroll
    for (int mt = 0; mt < 6; mt++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int b = m0 + mt * 16 + 4 * q + r;
            if (b < B) {
                float *dst = part3 + ((size_t)ks * B + b) * NQ_;
#pragma unroll
                for (int n = 0; n < BL_NW; n++)
                    if (nt0 + n < NQ_ / 16) dst[(nt0 + n) * 16 + j] = acc[mt][n][r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// backward, launch 3: per frame -- reduce the 27 tile partials and the 54 K-slab partials, VJP of A, chain, pose map, rodrigues, joints
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void smplh_bwd_frame_kernel(const float *__restrict__ pose, const float *__restrict__ J_s,
                                                              SmplParents par, const float *__restrict__ ws,
                                                              const float *__restrict__ part, const float *__restrict__ part3,
                                                              const float *__restrict__ djtr, int B, float *__restrict__ dpose, float *__restrict__ dbetas,
                                                              float *__restrict__ dtrans, const int *skip)
{
    VT_SKIP_RETURN(skip);
    __shared__ float red[PT_N], red3[NQ_];
    __shared__ float sdG[J_ * 12], sdJ[J_ * 3], sdR[J_ * 9];
    __shared__ float sG[J_ * 12], sRm[J_ * 9], sJr[J_ * 3];     // the frame's G, R, J: the serial chain below must not wait on HBM
    const int b = blockIdx.x, tid = threadIdx.x;
    const float *w = ws + (size_t)b * WS_FRAME;
    for (int i = tid; i < J_ * 12; i += 256) sG[i] = w[WS_G + i];
    for (int i = tid; i < J_ * 9; i += 256) sRm[i] = w[WS_R + i];
    for (int i = tid; i < J_ * 3; i += 256) sJr[i] = w[WS_J + i];
    // all partial loads of an element are independent: issue them in batches of 9 (a rolled loop paid one L2 round trip per partial)
    for (int i = tid; i < PT_N; i += 256) {
        float s = 0.f;
#pragma unroll 9
        for (int t = 0; t < NVT_; t++) s += part[((size_t)t * B + b) * PT_N + i];
        red[i] = s;
    }
    for (int i = tid; i < NQ_; i += 256) {
        float s = 0.f;
#pragma unroll 9
        for (int t = 0; t < NKS_; t++) s += part3[((size_t)t * B + b) * NQ_ + i];
        red3[i] = s;
    }
    __syncthreads();
    if (tid < J_) {
        const int j = tid;
        const float *G = sG + 12 * j; const float *Jr = sJr;
        float dj[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const float dat = red[PT_DA + 12 * j + r * 4 + 3];
#pragma unroll
            for (int c = 0; c < 3; c++) { sdG[12 * j + r * 4 + c] = red[PT_DA + 12 * j + r * 4 + c] - dat * Jr[3 * j + c]; dj[c] -= G[r * 4 + c] * dat; }
            sdG[12 * j + r * 4 + 3] = dat + (djtr ? djtr[(b * J_ + j) * 3 + r] : 0.f);
        }
        sdJ[3 * j] = dj[0]; sdJ[3 * j + 1] = dj[1]; sdJ[3 * j + 2] = dj[2];
#pragma unroll
        for (int e = 0; e < 9; e++) sdR[9 * j + e] = 0.f;
    }
    __syncthreads();
    // reverse kinematic chain: joints strictly in order (a parent collects all its children), but the 24 outputs of one joint --
    // 12 of d G_parent, 9 of d R_j, 3 of d J -- are independent: one thread each, one barrier per joint
    // Round 6: by tree level, deepest first, and inside a level by sibling rank (SmplParents::sched): up to ten joints -- 24 threads each -- per barrier interval,
    // the additions of the serial loop in their order (bit-identical), ~20 intervals instead of 51.
    const int jloc = tid / 24, el = tid - 24 * jloc;          // thread = (which joint of the step, which of its 24 outputs)
    const int nsteps = par.nsteps > 0 ? par.nsteps : J_ - 1;
    for (int step = 0; step < nsteps; step++) {
        const int j = par.nsteps > 0 ? (jloc < SCHED_WIDTH ? (int)par.sched[step][jloc < SCHED_WIDTH ? jloc : 0] : -1) : (jloc == 0 ? J_ - 1 - step : -1);
        if (j >= 1) {
            const int p = par.p[j];
            const float *Gp = sG + 12 * p, *Rj = sRm + 9 * j, *dGj = sdG + 12 * j;
            float upd = 0.f; float *dst = nullptr; float upd2 = 0.f; float *dst2 = nullptr;
            if (el < 12) {
                const int r = el >> 2, c = el & 3;
                dst = sdG + 12 * p + el;
                if (c < 3) upd = dGj[r * 4] * Rj[3 * c] + dGj[r * 4 + 1] * Rj[3 * c + 1] + dGj[r * 4 + 2] * Rj[3 * c + 2] + dGj[r * 4 + 3] * (sJr[3 * j + c] - sJr[3 * p + c]);
                else upd = dGj[r * 4 + 3];
            } else if (el < 21) {
                const int e = el - 12, r = e / 3, c = e - 3 * r;
                dst = sdR + 9 * j + e;
                upd = Gp[r] * dGj[c] + Gp[4 + r] * dGj[4 + c] + Gp[8 + r] * dGj[8 + c];
            } else {
                const int c = el - 21;
                upd = Gp[c] * dGj[3] + Gp[4 + c] * dGj[7] + Gp[8 + c] * dGj[11];
                dst = sdJ + 3 * j + c; dst2 = sdJ + 3 * p + c; upd2 = -upd;
            }
            if (dst) *dst += upd;
            if (dst2) *dst2 += upd2;
        }
        __syncthreads();
    }
    if (tid < 3) { for (int c = 0; c < 3; c++) sdR[3 * tid + c] += sdG[tid * 4 + c]; sdJ[tid] += sdG[tid * 4 + 3]; }
    __syncthreads();
    if (tid < J_) {
        const int j = tid;
        float g[9], t[3] = {pose[(b * J_ + j) * 3], pose[(b * J_ + j) * 3 + 1], pose[(b * J_ + j) * 3 + 2]}, d[3];
#pragma unroll
        for (int e = 0; e < 9; e++) g[e] = sdR[9 * j + e] + (j >= 1 ? red3[(j - 1) * 9 + e] : 0.f);
        rodrigues_bwd(t, g, d);
        dpose[(b * J_ + j) * 3] = d[0]; dpose[(b * J_ + j) * 3 + 1] = d[1]; dpose[(b * J_ + j) * 3 + 2] = d[2];
    } else if (tid >= 64 && tid < 64 + NB_) {
        const int l = tid - 64;
        float s = red3[NP_ + l];
        for (int i = 0; i < J_ * 3; i++) s += J_s[i * NB_ + l] * sdJ[i];
        dbetas[b * NB_ + l] = s;
    } else if (tid >= 128 && tid < 131) {
        const int c = tid - 128;
        float s = red[PT_DTR + c];
        if (djtr) for (int j = 0; j < J_; j++) s += djtr[(b * J_ + j) * 3 + c];
        dtrans[b * 3 + c] = s;
    }
}

// ---------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------
extern "C" int vt_smplh_create(vt_smplh **out, const float *v_template, const float *shapedirs, const float *posedirs,
                               const float *J_regressor, const float *weights, const int *parents, void *stream)
{
    VT_REQUIRE(out && v_template && shapedirs && posedirs && J_regressor && weights && parents, "vt_smplh_create: null argument");
    hipStream_t st = vt_stream(stream);
    vt_smplh *h = new vt_smplh();
    for (int j = 0; j < J_; j++) h->par.p[j] = (j == 0) ? 0 : parents[j];
    for (int j = 1; j < J_; j++) VT_REQUIRE(h->par.p[j] >= 0 && h->par.p[j] < j, "vt_smplh_create: parents[%d]=%d must be in [0,%d)", j, parents[j], j);
    {   // reverse-chain schedule (SmplParents): slot of joint j = (tree level, rank among its siblings by descending index); steps = distinct slots, deepest first
        int lvl[J_], rank[J_], maxl = 0;
        lvl[0] = 0; rank[0] = 0;
        for (int j = 1; j < J_; j++) { lvl[j] = lvl[h->par.p[j]] + 1; maxl = std::max(maxl, lvl[j]); rank[j] = 0; for (int k = j + 1; k < J_; k++) rank[j] += (h->par.p[k] == h->par.p[j]); }
        int n = 0; bool ok = true;
        for (int t = 0; t < SCHED_STEPS; t++) for (int i = 0; i < SCHED_WIDTH; i++) h->par.sched[t][i] = -1;
        for (int l = maxl; l >= 1 && ok; l--)
            for (int r = 0; r < J_ && ok; r++) {
                int cnt = 0;
                for (int j = J_ - 1; j >= 1; j--) if (lvl[j] == l && rank[j] == r) { if (n >= SCHED_STEPS || cnt >= SCHED_WIDTH) { ok = false; break; } h->par.sched[n][cnt++] = (short)j; }
                if (cnt) n++;
            }
        h->par.nsteps = ok ? n : 0;
    }
    float *Q_kcv = new float[(size_t)KQ_ * 3 * VP_](), *Q_t = new float[(size_t)VP_ * 3 * NQ_](), *W_jv = new float[(size_t)J_ * VP_](),
          *J_t = new float[J_ * 3], *J_s = new float[J_ * 3 * NB_];
    for (int v = 0; v < V_; v++) for (int c = 0; c < 3; c++) {
        for (int k = 0; k < NP_ + NB_ + 1; k++) {
            const float val = k < NP_ ? posedirs[((size_t)v * 3 + c) * NP_ + k]
                            : (k < NP_ + NB_ ? shapedirs[((size_t)v * 3 + c) * NB_ + (k - NP_)] : v_template[v * 3 + c]);
            Q_kcv[(size_t)(k * 3 + c) * VP_ + v] = val;
            {   // fragment-ordered Q for the backward GEMM: [k group][N tile][lane = (q, j)][step s] <- row 16g + 4q + s, column 16nt + j
                const size_t kr = (size_t)v * 3 + c, g = kr >> 4, qq = (kr & 15) >> 2, ss = kr & 3;
                Q_t[((g * (NQ_ / 16) + (k >> 4)) * 64 + qq * 16 + (k & 15)) * 4 + ss] = val;   // column 469 (template) is never used
            }
        }
    }
    float *W_v64 = new float[(size_t)VP_ * 64]();
    for (int v = 0; v < V_; v++) { for (int j = 0; j < J_; j++) { W_jv[(size_t)j * VP_ + v] = weights[(size_t)v * J_ + j]; W_v64[(size_t)v * 64 + j] = weights[(size_t)v * J_ + j]; } W_v64[(size_t)v * 64 + J_] = 1.0f; }
    // sparse form of the skinning weights (ascending joints, zero padded) when no vertex has more than SP_K non-zeros
    float *W_sp = new float[(size_t)2 * SP_K * VP_]();
    int nnz = 0;
    for (int v = 0; v < V_; v++) {
        int k = 0;
        for (int j = 0; j < J_; j++) {
            const float w = weights[(size_t)v * J_ + j];
            if (w == 0.0f) continue;
            if (k < SP_K) { const int jj = j; memcpy(&W_sp[(size_t)k * VP_ + v], &jj, 4); W_sp[(size_t)(SP_K + k) * VP_ + v] = w; }
            k++;
        }
        nnz = k > nnz ? k : nnz;
    }
    h->nnz = nnz <= SP_K ? (nnz > 0 ? nnz : 1) : 0;
    for (int j = 0; j < J_; j++) for (int c = 0; c < 3; c++) {
        double a = 0; double sb[NB_] = {0};
        for (int v = 0; v < V_; v++) {
            const double r = J_regressor[(size_t)j * V_ + v];
            if (r == 0.0) continue;
            a += r * v_template[v * 3 + c];
            for (int l = 0; l < NB_; l++) sb[l] += r * shapedirs[((size_t)v * 3 + c) * NB_ + l];
        }
        J_t[j * 3 + c] = (float)a;
        for (int l = 0; l < NB_; l++) J_s[(j * 3 + c) * NB_ + l] = (float)sb[l];
    }
    int rc = VT_OK;
    if ((rc = vt_upload(&h->Q_kcv, Q_kcv, (size_t)KQ_ * 3 * VP_, st)) || (rc = vt_upload(&h->Q_t, Q_t, (size_t)VP_ * 3 * NQ_, st)) ||
        (rc = vt_upload(&h->W_jv, W_jv, (size_t)J_ * VP_, st)) || (rc = vt_upload(&h->W_v64, W_v64, (size_t)VP_ * 64, st)) ||
        (rc = vt_upload(&h->J_t, J_t, (size_t)J_ * 3, st)) || (rc = vt_upload(&h->J_s, J_s, (size_t)J_ * 3 * NB_, st)) ||
        (rc = vt_upload(&h->W_sp, W_sp, (size_t)2 * SP_K * VP_, st))) {
        return rc;
    }
    VT_HIP(hipStreamSynchronize(st));  // host staging buffers are freed below
    delete[] Q_kcv; delete[] Q_t; delete[] W_jv; delete[] W_v64; delete[] J_t; delete[] J_s; delete[] W_sp;
    *out = h;
    return VT_OK;
}

extern "C" void vt_smplh_destroy(vt_smplh *h)
{
    if (!h) return;
    hipFree(h->Q_kcv); hipFree(h->Q_t); hipFree(h->W_jv); hipFree(h->W_v64); hipFree(h->J_t); hipFree(h->J_s); hipFree(h->W_sp);
    delete h;
}

extern "C" long vt_smplh_workspace_floats(int B) { return (long)B * WS_FRAME; }
extern "C" long vt_smplh_bwd_scratch_floats(int B) { return (long)NVT_ * B * PT_N + (long)B * KTOT_ + (long)NKS_ * B * NQ_; }

#define BWD_FB 6      /* frames per workgroup of the backward tile kernel: 27 x 16 = 432 workgroups at B = 96 fill the 512 slots better than 27 x 12 (8 frames): 141 -> 131 us */

extern "C" int vt_smplh_forward(const vt_smplh *h, const float *pose, const float *betas, const float *trans, int B,
                                float *verts, float *jtr, float *v_posed, float *ws, void *stream)
{
    VT_REQUIRE(h && pose && betas && trans && verts && v_posed && ws && B > 0, "vt_smplh_forward: null argument or B <= 0");
    hipStream_t st = vt_stream(stream); const int *skip = vt_skip_flag_of(st);
    hipLaunchKernelGGL(smplh_pose_kernel, dim3(B), dim3(64), 0, st, pose, betas, trans, h->J_t, h->J_s, h->par, ws, jtr, skip);
    VT_LAUNCH_CHECK();
    const size_t lds_f = sizeof(float) * (FWD_R0 + FWD_FB * 64 * 3 + (h->nnz > 0 ? 2 * SP_K * 64 : J_ * 64) + FWD_FB * 3);
    VT_LDS_LIMIT(smplh_verts_kernel, sizeof(float) * (FWD_R0 + FWD_FB * 64 * 3 + J_ * 64 + FWD_FB * 3));     // the limit is set once per device: the dense-weights size
    hipLaunchKernelGGL(smplh_verts_kernel, dim3(VP_ / 64, (B + FWD_FB - 1) / FWD_FB), dim3(256), lds_f, st, h->Q_kcv, h->W_jv, betas, trans, ws, B,
                       verts, v_posed, h->W_sp, h->nnz, skip);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

extern "C" int vt_smplh_backward(const vt_smplh *h, const float *pose, const float *betas, int B, const float *dverts,
                                 const float *djtr, const float *v_posed, const float *ws, float *scratch,
                                 float *dpose, float *dbetas, float *dtrans, void *stream)
{
    (void)betas;
    VT_REQUIRE(h && pose && dverts && v_posed && ws && scratch && dpose && dbetas && dtrans && B > 0, "vt_smplh_backward: null argument or B <= 0");
    hipStream_t st = vt_stream(stream); const int *skip = vt_skip_flag_of(st);
    const size_t lds = sizeof(float) * (256 * 13 + BWD_FB * 624);
    float *dvp_g = scratch + (size_t)NVT_ * B * PT_N, *part3 = dvp_g + (size_t)B * KTOT_;
    VT_LDS_LIMIT(smplh_bwd_tile_kernel<BWD_FB>, lds);
    hipLaunchKernelGGL(smplh_bwd_tile_kernel<BWD_FB>, dim3(NVT_, (B + BWD_FB - 1) / BWD_FB), dim3(256), lds, st, h->W_v64, ws, v_posed, dverts, B, scratch, dvp_g, skip);
    VT_LAUNCH_CHECK();
    hipLaunchKernelGGL(smplh_bwd_blend_kernel, dim3(NKS_, (NQ_ / 16 + 4 * BL_NW - 1) / (4 * BL_NW), (B + BL_M - 1) / BL_M), dim3(256), 0, st, h->Q_t, dvp_g, B, part3, skip);
    VT_LAUNCH_CHECK();
    hipLaunchKernelGGL(smplh_bwd_frame_kernel, dim3(B), dim3(256), 0, st, pose, h->J_s, h->par, ws, scratch, part3, djtr, B, dpose, dbetas, dtrans, skip);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

extern "C" int vt_rodrigues_forward(const float *aa, int n, float *R, void *stream)
{
    VT_REQUIRE(aa && R && n > 0, "vt_rodrigues_forward: bad argument");
    hipLaunchKernelGGL(rodrigues_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, vt_stream(stream), aa, n, R);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

extern "C" int vt_rodrigues_backward(const float *aa, int n, const float *dR, float *daa, void *stream)
{
    VT_REQUIRE(aa && dR && daa && n > 0, "vt_rodrigues_backward: bad argument");
    hipLaunchKernelGGL(rodrigues_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, vt_stream(stream), aa, n, dR, daa);
    VT_LAUNCH_CHECK();
    return VT_OK;
}
