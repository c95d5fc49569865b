// collide.hip -- human/object interpenetration term of the 'joint' phase: replaces
//   RegistrationBase.smpl_obj_collision / compute_collision_loss  (recon/recon_fit_base.py:97-100,736-765; weight 'collide' 3^2 / (1 + decay),
//   recon_fit_trivis_full.py:139,260-264; host-gated OFF in the reference unless the machine is called gpu16 / gpu20: recon_fit_base.py:106).
// The arithmetic lives in the un-vendored package mesh_intersection (torch-mesh-isect: BVH broad phase with max_collisions = 8 + exact
// triangle-triangle test, DistanceFieldPenetrationLoss(sigma = 0.5, point2plane = False)) -> PARITY UNPINNED.  Restated from the published
// definition (Tzionas et al., IJCV 2016, "Capturing hands in action using discriminative salient points and physics simulation", conic distance
// fields; used unchanged by SMPLify-X, Pavlakos et al. CVPR 2019, Sec. 3.4):  for a triangle f with circumcentre o, unit normal n, circumradius r
//     x = n.(v - o),  Phi(v) = |(v - o) - x n| / (r - (r / sigma) x),
//     Upsilon(x) = -x + 1 - sigma                                             x <= -sigma
//                  -(1 - 2 sigma) / (4 sigma^2) x^2 - x / (2 sigma) + (3 - 2 sigma) / 4      -sigma < x < sigma
//                  0                                                          x >= sigma
//     Psi_f(v) = ((1 - Phi(v)) Upsilon(x))^2  if Phi(v) < 1, else 0
//     P = sum over colliding pairs (f_s, f_t)  sum_{v in f_s} |Psi_{f_t}(v) n_s|^2 + sum_{v in f_t} |Psi_{f_s}(v) n_t|^2 ;   loss = mean over frames.
// Design for MI355X: no tree.  The object template has 2 500 faces and touches the body in a small region, so the broad phase is (1) the
// object's bounding box per frame, (2) an ORDERED compaction of the SMPL faces whose box overlaps it (typically a few hundred of 13 776),
// (3) one thread per object face testing the candidates staged through LDS (box test, then Moller's interval test); the first 8 colliding
// candidates in SMPL-face order are kept (max_collisions).  Human-object pairs only: self-collisions of either mesh (which the reference's
// combined-mesh BVH also reports) do not depend on the object pose -- in phase 'joint' only obj_t is optimised -- and are left out of the value.
// The gradient is returned for the object TRANSLATION (the only parameter the term can move): d/dt = sum over object intruder vertices of
// dPsi^2/dv minus the same derivative at SMPL vertices intruding object cones (a translation moves o, not n or r).  Value and gradient are
// accumulated in 64-bit fixed point: bit-reproducible.
#include "common.h"
#include "collide_geom.h"

#define COL_TILE 256
#define COL_FIX 1099511627776.0        /* 2^40 */

// (1) + (2): per frame: object AABB, ordered list of SMPL faces whose AABB overlaps it.  grid = B, block = 256.
__global__ __launch_bounds__(256) void collide_candidates_kernel(const float *__restrict__ sv, int NVs, const int *__restrict__ sf, int NFs,
                                                                 const float *__restrict__ ov, int NVo, int *__restrict__ cand, int *__restrict__ ncand)
{
    __shared__ float red[6][256];
    __shared__ int scan[256];
    __shared__ int base;
    const int b = blockIdx.x, tid = threadIdx.x;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = tid; i < NVo; i += 256)
        for (int k = 0; k < 3; k++) { const float x = ov[((size_t)b * NVo + i) * 3 + k]; lo[k] = fminf(lo[k], x); hi[k] = fmaxf(hi[k], x); }
    for (int k = 0; k < 3; k++) { red[k][tid] = lo[k]; red[3 + k][tid] = hi[k]; }
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) for (int k = 0; k < 3; k++) { red[k][tid] = fminf(red[k][tid], red[k][tid + s]); red[3 + k][tid] = fmaxf(red[3 + k][tid], red[3 + k][tid + s]); }
        __syncthreads();
    }
    for (int k = 0; k < 3; k++) { lo[k] = red[k][0]; hi[k] = red[3 + k][0]; }
    if (tid == 0) base = 0;
    __syncthreads();
    const float *v = sv + (size_t)b * NVs * 3;
    for (int f0 = 0; f0 < NFs; f0 += 256) {
        const int f = f0 + tid;
        int flag = 0;
        if (f < NFs) {
            float tlo[3], thi[3], T[9];
            for (int c = 0; c < 3; c++) for (int k = 0; k < 3; k++) T[3 * c + k] = v[(size_t)sf[3 * f + c] * 3 + k];
            col_tri_box(T, tlo, thi);
            flag = col_box_overlap(tlo, thi, lo, hi);
        }
        scan[tid] = flag;
        __syncthreads();
        for (int s = 1; s < 256; s <<= 1) {            // inclusive Hillis-Steele scan: positions in face order
            const int t = tid >= s ? scan[tid - s] : 0;
            __syncthreads();
            scan[tid] += t;
            __syncthreads();
        }
        if (flag) cand[(size_t)b * NFs + base + scan[tid] - 1] = f;
        __syncthreads();
        if (tid == 255) base += scan[255];
        __syncthreads();
    }
    if (tid == 0) ncand[b] = base;
}

// (3): one thread per object face.  grid = (ceil(NFo / 256), B), block = 256.
__global__ __launch_bounds__(256) void collide_loss_kernel(const float *__restrict__ sv, int NVs, const int *__restrict__ sf, int NFs,
                                                           const float *__restrict__ ov, int NVo, const int *__restrict__ of, int NFo,
                                                           const int *__restrict__ cand, const int *__restrict__ ncand, float sigma, int max_coll,
                                                           long long *__restrict__ acc /* [B][4]: value, d/dt xyz */, int *__restrict__ npairs)
{
    __shared__ float tri[COL_TILE][9];
    const int b = blockIdx.y, fo = blockIdx.x * 256 + threadIdx.x, tid = threadIdx.x;
    const float *vs = sv + (size_t)b * NVs * 3, *vo = ov + (size_t)b * NVo * 3;
    float To[9], olo[3], ohi[3];
    const bool live = fo < NFo;
    if (live) { for (int c = 0; c < 3; c++) for (int k = 0; k < 3; k++) To[3 * c + k] = vo[(size_t)of[3 * fo + c] * 3 + k]; col_tri_box(To, olo, ohi); }
    const int n = ncand[b];
    int found = 0;
    double val = 0.0, g[3] = {0.0, 0.0, 0.0};
    for (int c0 = 0; c0 < n; c0 += COL_TILE) {
        const int cn = min(COL_TILE, n - c0);
        __syncthreads();
        if (tid < cn) { const int f = cand[(size_t)b * NFs + c0 + tid]; for (int c = 0; c < 3; c++) for (int k = 0; k < 3; k++) tri[tid][3 * c + k] = vs[(size_t)sf[3 * f + c] * 3 + k]; }
        __syncthreads();
        if (live) for (int i = 0; i < cn && found < max_coll; i++) {
            float slo[3], shi[3];
            col_tri_box(tri[i], slo, shi);
            if (!col_box_overlap(slo, shi, olo, ohi)) continue;
            if (!col_tri_tri(To, tri[i])) continue;
            found++;
            float gd[3];
            // object vertices intruding the SMPL triangle's cone: they move with t
            val += (double)col_pair_side(tri[i], To, sigma, gd); g[0] += gd[0]; g[1] += gd[1]; g[2] += gd[2];
            // SMPL vertices intruding the object triangle's cone: the cone moves with t -> minus the derivative w.r.t. the vertex
            val += (double)col_pair_side(To, tri[i], sigma, gd); g[0] -= gd[0]; g[1] -= gd[1]; g[2] -= gd[2];
        }
    }
    if (found) {
        atomicAdd(reinterpret_cast<unsigned long long *>(acc) + 4 * b, (unsigned long long)__double2ll_rn(val * COL_FIX));
        for (int k = 0; k < 3; k++) atomicAdd(reinterpret_cast<unsigned long long *>(acc) + 4 * b + 1 + k, (unsigned long long)__double2ll_rn(g[k] * COL_FIX));
        if (npairs) atomicAdd(npairs + b, found);
    }
}

// value: *term += mean_b P_b ;  gradient: dt (B,3) += gscale / B * dP_b/dt
__global__ void collide_finish_kernel(const long long *__restrict__ acc, int B, float gscale, double *term, float *dt)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && dt) for (int k = 0; k < 3; k++) dt[3 * b + k] += (float)((double)acc[4 * b + 1 + k] * (1.0 / COL_FIX) * (double)gscale / (double)B);
    // the value: the per-frame sums stay in 64-bit FIXED POINT across the frames too (integer addition: the same bits whatever the order) and ONE
    // thread converts and adds the total -- an fp64 atomic per frame made the term, which feeds the stop rule, depend on the arrival order
    if (term && b == 0) {
        long long tot = 0;
        for (int f = 0; f < B; f++) tot += acc[4 * f];
        atomicAdd(term, (double)tot * (1.0 / COL_FIX) / (double)B);
    }
}

extern "C" long vt_collision_workspace_bytes(int B, int n_smpl_faces)
{
    return (long)sizeof(int) * ((long)B * n_smpl_faces + 2 * (long)B) + (long)sizeof(long long) * 4 * B + 64;
}

extern "C" int vt_collision_loss(const float *smpl_verts, int n_smpl_verts, const int *smpl_faces, int n_smpl_faces, const float *obj_verts, int n_obj_verts,
                                 const int *obj_faces, int n_obj_faces, int B, float sigma, int max_collisions, float gscale, double *term,
                                 float *d_obj_t, int *pairs_per_frame, void *workspace, void *stream)
{
    VT_REQUIRE(smpl_verts && smpl_faces && obj_verts && obj_faces && workspace && B > 0 && n_smpl_faces > 0 && n_obj_faces > 0 && sigma > 0.f && max_collisions > 0,
               "vt_collision_loss: bad argument");
    hipStream_t st = vt_stream(stream);
    long long *acc = reinterpret_cast<long long *>(workspace);                      // 8-byte aligned first
    int *cand = reinterpret_cast<int *>(acc + 4 * (size_t)B), *ncand = cand + (size_t)B * n_smpl_faces, *npairs = ncand + B;
    VT_HIP(hipMemsetAsync(acc, 0, sizeof(long long) * 4 * B, st));
    VT_HIP(hipMemsetAsync(npairs, 0, sizeof(int) * B, st));
    hipLaunchKernelGGL(collide_candidates_kernel, dim3(B), dim3(256), 0, st, smpl_verts, n_smpl_verts, smpl_faces, n_smpl_faces, obj_verts, n_obj_verts, cand, ncand);
    VT_LAUNCH_CHECK();
    hipLaunchKernelGGL(collide_loss_kernel, dim3((n_obj_faces + 255) / 256, B), dim3(256), 0, st, smpl_verts, n_smpl_verts, smpl_faces, n_smpl_faces, obj_verts,
                       n_obj_verts, obj_faces, n_obj_faces, cand, ncand, sigma, max_collisions, acc, npairs);
    VT_LAUNCH_CHECK();
    hipLaunchKernelGGL(collide_finish_kernel, dim3((B + 63) / 64), dim3(64), 0, st, acc, B, gscale, term, d_obj_t);
    VT_LAUNCH_CHECK();
    if (pairs_per_frame) VT_HIP(hipMemcpyAsync(pairs_per_frame, npairs, sizeof(int) * B, hipMemcpyDeviceToDevice, st));
    return VT_OK;
}
