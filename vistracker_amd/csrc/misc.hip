// misc.hip -- the small ops of the fit step: landmark regressors, priors, SO(3) projection, rigid transform,
// temporal stencils, keypoint terms, Adam, device-side early stop, layout conversion.  All HBM-bound and tiny
// next to the point query; the design goal is "one launch each, no host sync, deterministic gradients".
#include "common.h"

thread_local char vt_err_buf[512] = {0};
extern "C" const char *vt_last_error(void) { return vt_err_buf; }
extern "C" int vt_version(void) { return 1; }

// ---------------------------------------------------------------------------------------------------
// utilities
// ---------------------------------------------------------------------------------------------------
__global__ void fill_kernel(float *p, long n, float v) { long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }
__global__ void fill64_kernel(double *p, long n, double v) { long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }

extern "C" int vt_fill(float *p, long n, float value, void *stream)
{
    VT_REQUIRE(p && n > 0, "vt_fill: bad argument");
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, vt_stream(stream), p, n, value);
    VT_LAUNCH_CHECK();
    return VT_OK;
}
extern "C" int vt_fill_f64(double *p, long n, double value, void *stream)
{
    VT_REQUIRE(p && n > 0, "vt_fill_f64: bad argument");
    hipLaunchKernelGGL(fill64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, vt_stream(stream), p, n, value);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

__global__ __launch_bounds__(256) void sum_to_term_kernel(const float *v, int n, float scale, double *term)
{
    __shared__ double red[4];
    double s = 0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)v[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(term, (red[0] + red[1] + red[2] + red[3]) * (double)scale);
}
extern "C" int vt_sum_to_term(const float *value, int n, float scale, double *term, void *stream)
{
    VT_REQUIRE(value && term && n > 0, "vt_sum_to_term: bad argument");
    hipLaunchKernelGGL(sum_to_term_kernel, dim3(1), dim3(256), 0, vt_stream(stream), value, n, scale, term);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

// block-level fp64 accumulate into a term: every thread contributes `s`
__device__ __forceinline__ void term_add(double s, double *term, double *red /* >= blockDim/64 doubles */)
{
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const int nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0; for (int i = 0; i < nw; i++) t += red[i]; if (term) atomicAdd(term, t); }
}

// NCHW -> NHWC, 32x32 LDS tile transpose per frame: src viewed as [C][HW], dst as [HW][C]
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float *__restrict__ src, int C, int HW, float *__restrict__ dst)
{
    __shared__ float tile[32][33];
    const size_t base = (size_t)blockIdx.z * C * HW;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) { const int c = c0 + i, p = p0 + tx; tile[i][tx] = (c < C && p < HW) ? src[base + (size_t)c * HW + p] : 0.f; }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) { const int p = p0 + i, c = c0 + tx; if (p < HW && c < C) dst[base + (size_t)p * C + c] = tile[tx][i]; }
}
extern "C" int vt_nchw_to_nhwc(const float *src, int B, int C, int H, int W, float *dst, void *stream)
{
    VT_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0, "vt_nchw_to_nhwc: bad argument");
    const int HW = H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((HW + 31) / 32, (C + 31) / 32, B), dim3(256), 0, vt_stream(stream), src, C, HW, dst);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

// ---------------------------------------------------------------------------------------------------
// landmark regressors (body_landmark.py:16-28; torch_functions.py:52-76)
// ---------------------------------------------------------------------------------------------------
struct vt_landmarks {
    int K, V;
    int *indptr, *indices; float *data;      // CSR (K rows)
    int *colptr, *rowidx; float *cdata;      // CSC (V columns) for the VJP
};

__global__ __launch_bounds__(64) void landmarks_fwd_kernel(const int *indptr, const int *indices, const float *data,
                                                           const float *verts, int V, int K, float *out)
{
    const int k = blockIdx.x, b = blockIdx.y;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int e = indptr[k] + threadIdx.x; e < indptr[k + 1]; e += 64) {
        const float w = data[e]; const float *v = verts + ((size_t)b * V + indices[e]) * 3;
        a0 += w * v[0]; a1 += w * v[1]; a2 += w * v[2];
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
    if (threadIdx.x == 0) { float *o = out + ((size_t)b * K + k) * 3; o[0] = a0; o[1] = a1; o[2] = a2; }
}

__global__ void landmarks_bwd_kernel(const int *colptr, const int *rowidx, const float *cdata, const float *dout,
                                     int V, int K, int B, float *dverts, int accumulate)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (v >= V) return;
    const int s = colptr[v], e = colptr[v + 1];
    if (s == e && accumulate) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int i = s; i < e; i++) { const float w = cdata[i]; const float *g = dout + ((size_t)b * K + rowidx[i]) * 3; a0 += w * g[0]; a1 += w * g[1]; a2 += w * g[2]; }
    float *o = dverts + ((size_t)b * V + v) * 3;
    if (accumulate) { o[0] += a0; o[1] += a1; o[2] += a2; } else { o[0] = a0; o[1] = a1; o[2] = a2; }
}

extern "C" int vt_landmarks_create(vt_landmarks **out, const int *indptr, const int *indices, const float *data, int K, int V, void *stream)
{
    VT_REQUIRE(out && indptr && indices && data && K > 0 && V > 0, "vt_landmarks_create: bad argument");
    hipStream_t st = vt_stream(stream);
    const int nnz = indptr[K];
    int *colptr = new int[V + 1](), *rowidx = new int[nnz]; float *cdata = new float[nnz];
    for (int e = 0; e < nnz; e++) { VT_REQUIRE(indices[e] >= 0 && indices[e] < V, "vt_landmarks_create: column index out of range"); colptr[indices[e] + 1]++; }
    for (int v = 0; v < V; v++) colptr[v + 1] += colptr[v];
    int *fillp = new int[V];
    memcpy(fillp, colptr, sizeof(int) * V);
    for (int k = 0; k < K; k++) for (int e = indptr[k]; e < indptr[k + 1]; e++) { const int p = fillp[indices[e]]++; rowidx[p] = k; cdata[p] = data[e]; }
    vt_landmarks *h = new vt_landmarks(); h->K = K; h->V = V;
    int rc;
    if ((rc = vt_upload(&h->indptr, indptr, (size_t)K + 1, st)) || (rc = vt_upload(&h->indices, indices, (size_t)nnz, st)) ||
        (rc = vt_upload(&h->data, data, (size_t)nnz, st)) || (rc = vt_upload(&h->colptr, colptr, (size_t)V + 1, st)) ||
        (rc = vt_upload(&h->rowidx, rowidx, (size_t)nnz, st)) || (rc = vt_upload(&h->cdata, cdata, (size_t)nnz, st))) return rc;
    VT_HIP(hipStreamSynchronize(st));
    delete[] colptr; delete[] rowidx; delete[] cdata; delete[] fillp;
    *out = h;
    return VT_OK;
}
extern "C" void vt_landmarks_destroy(vt_landmarks *h)
{
    if (!h) return;
    hipFree(h->indptr); hipFree(h->indices); hipFree(h->data); hipFree(h->colptr); hipFree(h->rowidx); hipFree(h->cdata);
    delete h;
}
extern "C" int vt_landmarks_forward(const vt_landmarks *h, const float *verts, int B, float *out, void *stream)
{
    VT_REQUIRE(h && verts && out && B > 0, "vt_landmarks_forward: bad argument");
    hipLaunchKernelGGL(landmarks_fwd_kernel, dim3(h->K, B), dim3(64), 0, vt_stream(stream), h->indptr, h->indices, h->data, verts, h->V, h->K, out);
    VT_LAUNCH_CHECK();
    return VT_OK;
}
extern "C" int vt_landmarks_backward(const vt_landmarks *h, const float *dout, int B, float *dverts, int accumulate, void *stream)
{
    VT_REQUIRE(h && dout && dverts && B > 0, "vt_landmarks_backward: bad argument");
    hipLaunchKernelGGL(landmarks_bwd_kernel, dim3((h->V + 255) / 256, B), dim3(256), 0, vt_stream(stream), h->colptr, h->rowidx, h->cdata, dout,
                       h->V, h->K, B, dverts, accumulate);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

// ---------------------------------------------------------------------------------------------------
// Mahalanobis priors (th_smpl_prior.py:30-38; th_hand_prior.py:57-72)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void mahalanobis_kernel(const float *x, int stride, int off, int n, const float *mean,
                                                         const float *prec, float *value, float *dx, float gscale)
{
    __shared__ float d[64], t2[64];
    const int b = blockIdx.x, j = threadIdx.x;
    d[j] = (j < n) ? x[(size_t)b * stride + off + j] - mean[j] : 0.f;
    __syncthreads();
    float a = 0.f;
    if (j < n) for (int i = 0; i < n; i++) a += d[i] * prec[i * n + j];
    t2[j] = a;
    const float val = wave_sum(a * a);
    if (j == 0) value[b] = val;
    __syncthreads();
    if (dx && j < n) {
        float g = 0.f;
        for (int k = 0; k < n; k++) g += t2[k] * prec[j * n + k];
        dx[(size_t)b * stride + off + j] += 2.f * g * gscale;
    }
}
extern "C" int vt_mahalanobis(const float *x, int B, int stride, int off, int n, const float *mean, const float *prec,
                              float *value, float *dx, float gscale, void *stream)
{
    VT_REQUIRE(x && mean && prec && value && B > 0 && n > 0 && n <= 64 && off >= 0 && off + n <= stride, "vt_mahalanobis: bad argument (n must be <= 64)");
    hipLaunchKernelGGL(mahalanobis_kernel, dim3(B), dim3(64), 0, vt_stream(stream), x, stride, off, n, mean, prec, value, dx, gscale);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

// ---------------------------------------------------------------------------------------------------
// SO(3) projection (recon_fit_base.py:179-199): one thread per matrix, one-sided Jacobi SVD in registers.
// VJP in polar form: dM = U D Z V^T, Q = D U^T G V, Z_ij = (Q_ij - Q_ji)/(h_i + h_j), h = (s1, s2, d*s3)
// (the same derivative autograd takes through torch.svd/det, without its 1/(s_i^2 - s_j^2) cancellation).
// ---------------------------------------------------------------------------------------------------
struct Svd3 { float U[9], V[9], s[3], d; };
#define SVD_WS 22    /* floats per frame of the head -> tail hand-over: U, V, s, d */

__device__ __forceinline__ void jacobi_pair(float *A, float *V, const int p, const int q)
{
    float a = 0.f, b = 0.f, g = 0.f;
#pragma unroll
    for (int r = 0; r < 3; r++) { a += A[3 * r + p] * A[3 * r + p]; b += A[3 * r + q] * A[3 * r + q]; g += A[3 * r + p] * A[3 * r + q]; }
    if (fabsf(g) <= 1e-30f) return;
    const float zeta = (b - a) / (2.f * g);
    const float t = (zeta >= 0.f ? 1.f : -1.f) / (fabsf(zeta) + sqrtf(1.f + zeta * zeta));
    const float c = 1.f / sqrtf(1.f + t * t), sn = c * t;
#pragma unroll
    for (int r = 0; r < 3; r++) {
        float x = A[3 * r + p], y = A[3 * r + q]; A[3 * r + p] = c * x - sn * y; A[3 * r + q] = sn * x + c * y;
        x = V[3 * r + p]; y = V[3 * r + q]; V[3 * r + p] = c * x - sn * y; V[3 * r + q] = sn * x + c * y;
    }
}

__device__ __forceinline__ void svd3(const float *M, Svd3 &o)
{
    float A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
#pragma unroll
    for (int e = 0; e < 9; e++) A[e] = M[e];
    for (int sweep = 0; sweep < 8; sweep++) { jacobi_pair(A, V, 0, 1); jacobi_pair(A, V, 0, 2); jacobi_pair(A, V, 1, 2); }
    float sv[3];
#pragma unroll
    for (int c = 0; c < 3; c++) sv[c] = sqrtf(A[c] * A[c] + A[3 + c] * A[3 + c] + A[6 + c] * A[6 + c]);
    // sort columns by singular value, descending (torch.svd order; the last one carries the det sign)
#define SWAPC(i, j)                                                                                          \
    if (sv[j] > sv[i]) {                                                                                     \
        float t_ = sv[i]; sv[i] = sv[j]; sv[j] = t_;                                                         \
        for (int r = 0; r < 3; r++) { t_ = A[3 * r + i]; A[3 * r + i] = A[3 * r + j]; A[3 * r + j] = t_;     \
                                      t_ = V[3 * r + i]; V[3 * r + i] = V[3 * r + j]; V[3 * r + j] = t_; }   \
    }
    SWAPC(0, 1) SWAPC(0, 2) SWAPC(1, 2)
#undef SWAPC
#pragma unroll
    for (int c = 0; c < 3; c++) {
        o.s[c] = sv[c];
        const float inv = sv[c] > 0.f ? 1.f / sv[c] : 0.f;
#pragma unroll
        for (int r = 0; r < 3; r++) { o.U[3 * r + c] = A[3 * r + c] * inv; o.V[3 * r + c] = V[3 * r + c]; }
    }
    // det(U V^T) = det(U) det(V)
    const float *U = o.U, *W = o.V;
    const float dU = U[0] * (U[4] * U[8] - U[5] * U[7]) - U[1] * (U[3] * U[8] - U[5] * U[6]) + U[2] * (U[3] * U[7] - U[4] * U[6]);
    const float dV = W[0] * (W[4] * W[8] - W[5] * W[7]) - W[1] * (W[3] * W[8] - W[5] * W[6]) + W[2] * (W[3] * W[7] - W[4] * W[6]);
    o.d = dU * dV;
}

__global__ void so3_fwd_kernel(const float *M0, const float *noise, int B, float *R)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float M[9]; Svd3 s;
#pragma unroll
    for (int e = 0; e < 9; e++) M[e] = M0[9 * b + e] + (noise ? 1e-4f * noise[9 * b + e] : 0.f);
    svd3(M, s);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) R[9 * b + 3 * r + c] = s.U[3 * r] * s.V[3 * c] + s.U[3 * r + 1] * s.V[3 * c + 1] + s.d * s.U[3 * r + 2] * s.V[3 * c + 2];
}

__global__ void so3_bwd_kernel(const float *M0, const float *noise, int B, const float *dR, float *dM)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float M[9], G[9]; Svd3 s;
#pragma unroll
    for (int e = 0; e < 9; e++) { M[e] = M0[9 * b + e] + (noise ? 1e-4f * noise[9 * b + e] : 0.f); G[e] = dR[9 * b + e]; }
    svd3(M, s);
    const float D[3] = {1.f, 1.f, s.d}, h[3] = {s.s[0], s.s[1], s.d * s.s[2]};
    float UtG[9], Q[9], Z[9], UDZ[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) UtG[3 * r + c] = s.U[r] * G[c] + s.U[3 + r] * G[3 + c] + s.U[6 + r] * G[6 + c];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) Q[3 * r + c] = D[r] * (UtG[3 * r] * s.V[c] + UtG[3 * r + 1] * s.V[3 + c] + UtG[3 * r + 2] * s.V[6 + c]);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) Z[3 * r + c] = (r == c) ? 0.f : (Q[3 * r + c] - Q[3 * c + r]) / (h[r] + h[c]);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) UDZ[3 * r + c] = s.U[3 * r] * D[0] * Z[c] + s.U[3 * r + 1] * D[1] * Z[3 + c] + s.U[3 * r + 2] * D[2] * Z[6 + c];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) dM[9 * b + 3 * r + c] = UDZ[3 * r] * s.V[3 * c] + UDZ[3 * r + 1] * s.V[3 * c + 1] + UDZ[3 * r + 2] * s.V[3 * c + 2];
}

extern "C" int vt_so3_project_forward(const float *M0, const float *noise, int B, float *R, void *stream)
{
    VT_REQUIRE(M0 && R && B > 0, "vt_so3_project_forward: bad argument");
    hipLaunchKernelGGL(so3_fwd_kernel, dim3((B + 63) / 64), dim3(64), 0, vt_stream(stream), M0, noise, B, R);
    VT_LAUNCH_CHECK();
    return VT_OK;
}
extern "C" int vt_so3_project_backward(const float *M0, const float *noise, int B, const float *dR, float *dM, void *stream)
{
    VT_REQUIRE(M0 && dR && dM && B > 0, "vt_so3_project_backward: bad argument");
    hipLaunchKernelGGL(so3_bwd_kernel, dim3((B + 63) / 64), dim3(64), 0, vt_stream(stream), M0, noise, B, dR, dM);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

// ---------------------------------------------------------------------------------------------------
// rigid transform (recon_fit_base.py:455-459)
// ---------------------------------------------------------------------------------------------------
__global__ void rigid_fwd_kernel(const float *X0, int shared, const float *R, const float *t, const float *s, int B, int N, float *X)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (n >= N) return;
    const float *x = X0 + ((shared ? 0 : (size_t)b * N) + n) * 3, *r = R + 9 * b;
    const float x0 = x[0], x1 = x[1], x2 = x[2], sc = s[b];
    float *o = X + ((size_t)b * N + n) * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) o[c] = (x0 * r[c] + x1 * r[3 + c] + x2 * r[6 + c] + t[3 * b + c]) * sc;
}

__global__ __launch_bounds__(256) void rigid_bwd_kernel(const float *X0, int shared, const float *s, int N, const float *dX,
                                                        float *dR, float *dt, int accumulate)
{
    __shared__ float red[4];
    const int b = blockIdx.x;
    float a[12];
#pragma unroll
    for (int e = 0; e < 12; e++) a[e] = 0.f;
    const float sc = s[b];
    for (int n = threadIdx.x; n < N; n += 256) {
        const float *x = X0 + ((shared ? 0 : (size_t)b * N) + n) * 3, *g = dX + ((size_t)b * N + n) * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) { const float gc = g[c] * sc; a[9 + c] += gc; a[c] += x[0] * gc; a[3 + c] += x[1] * gc; a[6 + c] += x[2] * gc; }
    }
#pragma unroll
    for (int e = 0; e < 12; e++) {
        const float v = block_sum<4>(a[e], red);
        if (threadIdx.x == 0) {
            float *dst = (e < 9) ? dR + 9 * b + e : dt + 3 * b + (e - 9);
            *dst = accumulate ? *dst + v : v;
        }
    }
}

extern "C" int vt_rigid_forward(const float *X0, int shared_x0, const float *R, const float *t, const float *s, int B, int N, float *X, void *stream)
{
    VT_REQUIRE(X0 && R && t && s && X && B > 0 && N > 0, "vt_rigid_forward: bad argument");
    hipLaunchKernelGGL(rigid_fwd_kernel, dim3((N + 255) / 256, B), dim3(256), 0, vt_stream(stream), X0, shared_x0, R, t, s, B, N, X);
    VT_LAUNCH_CHECK();
    return VT_OK;
}
extern "C" int vt_rigid_backward(const float *X0, int shared_x0, const float *s, int B, int N, const float *dX, float *dR, float *dt,
                                 int accumulate, void *stream)
{
    VT_REQUIRE(X0 && s && dX && dR && dt && B > 0 && N > 0, "vt_rigid_backward: bad argument");
    hipLaunchKernelGGL(rigid_bwd_kernel, dim3(B), dim3(256), 0, vt_stream(stream), X0, shared_x0, s, N, dX, dR, dt, accumulate);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

// ---------------------------------------------------------------------------------------------------
// temporal stencils over the frames of a batch, v (B, D):  a_b = 2 v_b - v_{b-1} - v_{b+1}  (b = 1 .. B-2).
// thread == one ELEMENT (frame f, column i): it sums the (up to three) stencils that touch v[f][i], so every gradient element is
// written exactly once (no atomics) and the B frames are not walked serially (a thread-per-column walk was latency bound: 59 us).
// ---------------------------------------------------------------------------------------------------
// 1024 threads x at most 256 workgroups (round 6; was 256 x 512): every workgroup ends with ONE fp64 atomic on the term, and same-address atomics are
// performed one after the other at the memory side (~12 ns each: 512 of them were 6 of the kernel's 20 us; with 2048 workgroups the kernel took 33 us,
// with 4096 54 us) -- fewer, larger workgroups: 12.3 us at B = 96, D = 20670, `dv` bit-identical (profiles/r06_accel_loss_ab.txt)
#define ACCEL_T 1024
#define ACCEL_MAX_BLOCKS 256
__global__ __launch_bounds__(ACCEL_T) void accel_loss_kernel(const float *__restrict__ v, int B, int Dcols, int D, const float *__restrict__ elem_w,
                                                         float gs, double *term, float *__restrict__ dv)
{
    // D = row stride (floats per frame), Dcols = columns that take part
    __shared__ double red[ACCEL_T / 64];
    double acc = 0;
    // grid-stride over the elements: at most 512 blocks, so the fp64 atomics on the loss term (one per block, all arriving at the end)
    // do not serialise the tail of the kernel
    for (int t = blockIdx.x * ACCEL_T + threadIdx.x; t < B * Dcols; t += gridDim.x * ACCEL_T) {
        const int f = t / Dcols, i = t - f * Dcols;
        const float w = elem_w ? elem_w[i] : 1.f;
        auto at = [&](int b) { return v[(size_t)min(max(b, 0), B - 1) * D + i]; };
        const float vm2 = at(f - 2), vm1 = at(f - 1), v0 = at(f), vp1 = at(f + 1), vp2 = at(f + 2);
        // stencils centred on f-1, f, f+1 (a stencil exists for centres 1 .. B-2)
        const float a_m = (f - 1 >= 1 && f - 1 <= B - 2) ? 2.f * vm1 - vm2 - v0 : 0.f;
        const float a_0 = (f >= 1 && f <= B - 2) ? 2.f * v0 - vm1 - vp1 : 0.f;
        const float a_p = (f + 1 >= 1 && f + 1 <= B - 2) ? 2.f * vp1 - v0 - vp2 : 0.f;
        acc += (double)(w * a_0 * a_0);
        if (dv) dv[(size_t)f * D + i] += gs * w * (2.f * a_0 - a_m - a_p);
    }
    term_add(acc / ((double)(B - 2) * Dcols), term, red);
}

// d_b = v_b - v_{b-1} (b = 1 .. B-1): loss mean(d^2); thread == one element, as above
__global__ __launch_bounds__(256) void velocity_loss_kernel(const float *__restrict__ v, int B, int D, float gs, double *term, float *__restrict__ dv)
{
    __shared__ double red[4];
    double acc = 0;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < B * D; t += gridDim.x * 256) {
        const int f = t / D, i = t - f * D;
        const float v0 = v[(size_t)f * D + i];
        const float d0 = f >= 1 ? v0 - v[(size_t)(f - 1) * D + i] : 0.f;          // d_f
        const float d1 = f + 1 < B ? v[(size_t)(f + 1) * D + i] - v0 : 0.f;        // d_{f+1}
        acc += (double)(d0 * d0);
        if (dv) dv[(size_t)f * D + i] += gs * (d0 - d1);
    }
    term_add(acc / ((double)(B - 1) * D), term, red);
}

// ---------------------------------------------------------------------------------------------------
// x2 bicubic upsampling (align_corners = True, A = -0.75, clamped taps: torch.nn.functional.interpolate semantics) of an NHWC tensor,
// fused with the skip connection of the hourglass: out = skip + up(low)   (model/HGFilters.py:45-47).
// thread = one output pixel x 4 channels (16-B accesses; the 16 taps of neighbouring threads hit the same cache lines).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cubic_w(float t, float *w)
{
    const float A = -0.75f;
    const float x0 = t + 1.f, x1 = t, x2 = 1.f - t, x3 = 2.f - t;
    w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
    w[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
    w[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
    w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}
__global__ __launch_bounds__(256) void upsample2x_bicubic_add_kernel(const float *__restrict__ low, const float *__restrict__ skip, int B, int h, int w, int C,
                                                                     float *__restrict__ out)
{
    const int C4 = C >> 2, H = 2 * h, W = 2 * w;
    const long t = (long)blockIdx.x * 256 + threadIdx.x, total = (long)B * H * W * C4;
    if (t >= total) return;
    const int c4 = (int)(t % C4); long r = t / C4;
    const int ox = (int)(r % W); r /= W;
    const int oy = (int)(r % H); const int b = (int)(r / H);
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const float fy = sy * oy, fx = sx * ox;
    const int iy = (int)floorf(fy), ix = (int)floorf(fx);
    float wy[4], wx[4];
    cubic_w(fy - iy, wy); cubic_w(fx - ix, wx);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int yy = min(max(iy - 1 + i, 0), h - 1);
        float4 row = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int xx = min(max(ix - 1 + k, 0), w - 1);
            const float4 v = *reinterpret_cast<const float4 *>(low + (((size_t)b * h + yy) * w + xx) * C + 4 * c4);
            row.x += wx[k] * v.x; row.y += wx[k] * v.y; row.z += wx[k] * v.z; row.w += wx[k] * v.w;
        }
        acc.x += wy[i] * row.x; acc.y += wy[i] * row.y; acc.z += wy[i] * row.z; acc.w += wy[i] * row.w;
    }
    const size_t o = (((size_t)b * H + oy) * W + ox) * C + 4 * c4;
    if (skip) { const float4 s = *reinterpret_cast<const float4 *>(skip + o); acc.x += s.x; acc.y += s.y; acc.z += s.z; acc.w += s.w; }
    *reinterpret_cast<float4 *>(out + o) = acc;
}
// ---------------------------------------------------------------------------------------------------
// GroupNorm (+ ReLU) on an NHWC tensor in two passes over x instead of torch's five (moments; normalise + affine; ReLU as a
// separate element-wise kernel): pass 1 accumulates per-(frame, channel) sum / sum of squares in fp64, pass 2 normalises with the
// group statistics and clamps.  The pre-activated blocks of the encoder are GN -> ReLU -> conv (model/net_util.py:374-388).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_stats_kernel(const float *__restrict__ x, int cstride, int HW, int C, int rows_per_block, double *__restrict__ part)
{
    // thread = (pixel row slot, float4 of channels): C/4 float4 per pixel, 256 / (C/4) pixels per sweep
    const int C4 = C >> 2, b = blockIdx.y, c4 = threadIdx.x % C4, slot = threadIdx.x / C4, nslot = 256 / C4;
    const int p0 = blockIdx.x * rows_per_block, p1 = min(HW, p0 + rows_per_block);
    float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
    if (slot < nslot)
        for (int p = p0 + slot; p < p1; p += nslot) {
            const float4 v = *reinterpret_cast<const float4 *>(x + ((size_t)b * HW + p) * cstride + 4 * c4);
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
            q[0] += v.x * v.x; q[1] += v.y * v.y; q[2] += v.z * v.z; q[3] += v.w * v.w;
        }
    // combine the pixel slots of a channel through LDS; one fp64 partial per (block, frame, channel, statistic): no atomics, nothing to zero,
    // and the result does not depend on the order in which the blocks ran
    __shared__ float red[256 * 8];
#pragma unroll
    for (int k = 0; k < 4; k++) { red[threadIdx.x * 8 + k] = s[k]; red[threadIdx.x * 8 + 4 + k] = q[k]; }
    __syncthreads();
    if (threadIdx.x < C4) {
        double ds[4] = {0, 0, 0, 0}, dq[4] = {0, 0, 0, 0};
        for (int sl = 0; sl < nslot; sl++)
#pragma unroll
            for (int k = 0; k < 4; k++) { ds[k] += (double)red[(sl * C4 + threadIdx.x) * 8 + k]; dq[k] += (double)red[(sl * C4 + threadIdx.x) * 8 + 4 + k]; }
        double *o = part + (((size_t)blockIdx.x * gridDim.y + b) * C + 4 * threadIdx.x) * 2;
#pragma unroll
        for (int k = 0; k < 4; k++) { o[2 * k] = ds[k]; o[2 * k + 1] = dq[k]; }
    }
}
// per (frame, group), one wave: mean and 1 / sqrt(var + eps) from the block partials (fp64, fixed order), two floats at stats[b * groups + g]
__global__ __launch_bounds__(64) void gn_finalize_kernel(const double *__restrict__ part, int nblk, int B, int HW, int C, int groups, float eps, float2 *__restrict__ stats)
{
    const int i = blockIdx.x, b = i / groups, g = i - b * groups, cg = C / groups, lane = threadIdx.x;
    double sm = 0, sq = 0;
    for (int e = lane; e < nblk * cg; e += 64) {
        const int blk = e / cg, k = e - blk * cg;
        const double *p = part + (((size_t)blk * B + b) * C + g * cg + k) * 2;
        sm += p[0]; sq += p[1];
    }
    for (int o = 32; o > 0; o >>= 1) { sm += __shfl_xor(sm, o, 64); sq += __shfl_xor(sq, o, 64); }
    if (lane == 0) {
        const double n = (double)HW * cg, mean = sm / n, var = fmax(sq / n - mean * mean, 0.0);
        stats[i] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)eps)));
    }
}
__global__ __launch_bounds__(256) void gn_apply_kernel(const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta,
                                                       const float2 *__restrict__ stats, int B, int HW, int C, int groups, int relu,
                                                       float *__restrict__ y)
{
    const int C4 = C >> 2, cg = C / groups;
    const long t = (long)blockIdx.x * 256 + threadIdx.x, total = (long)B * HW * C4;
    if (t >= total) return;
    const int c4 = (int)(t % C4); const long pix = t / C4; const int b = (int)(pix / HW);
    const float4 v = *reinterpret_cast<const float4 *>(x + pix * C + 4 * c4);
    const float4 ga = *reinterpret_cast<const float4 *>(gamma + 4 * c4), be = *reinterpret_cast<const float4 *>(beta + 4 * c4);
    const float in[4] = {v.x, v.y, v.z, v.w}, gm[4] = {ga.x, ga.y, ga.z, ga.w}, bt[4] = {be.x, be.y, be.z, be.w}; float out[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float2 st = stats[b * groups + (4 * c4 + k) / cg];
        const float o = (in[k] - st.x) * st.y * gm[k] + bt[k];
        out[k] = relu ? fmaxf(o, 0.f) : o;
    }
    *reinterpret_cast<float4 *>(y + pix * C + 4 * c4) = make_float4(out[0], out[1], out[2], out[3]);
}
// Producers that leave the GroupNorm partial sums of their OUTPUT behind (layout of gn_stats_kernel: one block of (B, C) x {sum, sum of squares} per
// workgroup), so that the ConvBlock that reads the tensor next calls vt_groupnorm_finalize instead of a statistics pass over it:
//   OP 0: 2 x 2 average pooling (F.avg_pool2d(x, 2, stride=2): model/HGFilters.py:33, 131-136), x (B, 2h, 2w, C) -> (B, h, w, C)
//   OP 1: skip + bicubic x2 up-sampling of low (upsample2x_bicubic_add_kernel's arithmetic), low (B, h / 2, w / 2, C), skip / out (B, h, w, C)
// grid = (blocks of output pixels, B); thread = (pixel slot, float4 of channels) like gn_stats_kernel.
template <int OP>
__global__ __launch_bounds__(256) void sweep_stats_kernel(const float *__restrict__ a, const float *__restrict__ skip, int h, int w, int C, int rows_per_block,
                                                          float *__restrict__ out, double *__restrict__ part)
{
    // (h, w) = the output size.  OP 0: a block item is an output pixel; OP 1: an output QUAD (2 x 2 pixels = one pixel of `low`): its four pixels read
    // their 4 x 4 taps from one 5 x 5 patch of `low` (the tap bases of neighbouring output pixels differ by at most one), 25 loads instead of 64
    const int C4 = C >> 2, b = blockIdx.y, c4 = threadIdx.x % C4, slot = threadIdx.x / C4, nslot = 256 / C4, HW = h * w;
    const int lh = h >> 1, lw = w >> 1, items = OP == 0 ? HW : lh * lw;
    const int p0 = blockIdx.x * rows_per_block, p1 = min(items, p0 + rows_per_block);
    float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
    auto emit = [&](int oy, int ox, const float4 v) {
        *reinterpret_cast<float4 *>(out + ((size_t)b * HW + (size_t)oy * w + ox) * C + 4 * c4) = v;
        s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
        q[0] += v.x * v.x; q[1] += v.y * v.y; q[2] += v.z * v.z; q[3] += v.w * v.w;
    };
    if (slot < nslot)
        for (int p = p0 + slot; p < p1; p += nslot) {
            if (OP == 0) {
                const int oy = p / w, ox = p - oy * w;
                const float *r0 = a + (((size_t)b * 2 * h + 2 * oy) * 2 * w + 2 * ox) * C + 4 * c4, *r1 = r0 + (size_t)2 * w * C;
                const float4 v00 = *reinterpret_cast<const float4 *>(r0), v01 = *reinterpret_cast<const float4 *>(r0 + C);
                const float4 v10 = *reinterpret_cast<const float4 *>(r1), v11 = *reinterpret_cast<const float4 *>(r1 + C);
                // avg_pool2d: the window summed row by row, divided by 4
                emit(oy, ox, make_float4((v00.x + v01.x + v10.x + v11.x) * 0.25f, (v00.y + v01.y + v10.y + v11.y) * 0.25f, (v00.z + v01.z + v10.z + v11.z) * 0.25f,
                                         (v00.w + v01.w + v10.w + v11.w) * 0.25f));
            } else {
                const int qy = p / lw, qx = p - qy * lw;
                const float sy = h > 1 ? (float)(lh - 1) / (float)(h - 1) : 0.f, sx = w > 1 ? (float)(lw - 1) / (float)(w - 1) : 0.f;
                float fy[2], fx[2]; int iy[2], ix[2];
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    fy[e] = sy * (float)(2 * qy + e); iy[e] = (int)floorf(fy[e]);
                    fx[e] = sx * (float)(2 * qx + e); ix[e] = (int)floorf(fx[e]);
                }
                const int ry = iy[0] - 1, rx = ix[0] - 1;       // patch origin; iy[1] - iy[0], ix[1] - ix[0] are 0 or 1
                float4 pt[5][5];
#pragma unroll
                for (int i = 0; i < 5; i++) {
                    const int yy = min(max(ry + i, 0), lh - 1);
#pragma unroll
                    for (int k = 0; k < 5; k++) {
                        const int xx = min(max(rx + k, 0), lw - 1);
                        pt[i][k] = *reinterpret_cast<const float4 *>(a + (((size_t)b * lh + yy) * lw + xx) * C + 4 * c4);
                    }
                }
                // separable form with FIVE-tap weight rows: the four cubic weights of an output pixel sit at patch columns (rows) d .. d + 3 with d = 0 or 1
                // (its tap base against the patch origin), the fifth weight is zero -- every product with it adds an exact zero, so the sums are the ones of
                // the 4 x 4 form (row sums first, then the column sum, in the same order), without a select per tap and channel and with every row sum
                // computed once for both output rows that use it
                float w5x[2][5], w5y[2][5];
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    float wx[4], wy[4]; cubic_w(fx[e] - ix[e], wx); cubic_w(fy[e] - iy[e], wy);
                    const bool dx = ix[e] != ix[0], dy = iy[e] != iy[0];
                    w5x[e][0] = dx ? 0.f : wx[0]; w5x[e][1] = dx ? wx[0] : wx[1]; w5x[e][2] = dx ? wx[1] : wx[2]; w5x[e][3] = dx ? wx[2] : wx[3]; w5x[e][4] = dx ? wx[3] : 0.f;
                    w5y[e][0] = dy ? 0.f : wy[0]; w5y[e][1] = dy ? wy[0] : wy[1]; w5y[e][2] = dy ? wy[1] : wy[2]; w5y[e][3] = dy ? wy[2] : wy[3]; w5y[e][4] = dy ? wy[3] : 0.f;
                }
                float4 rs[5][2];
#pragma unroll
                for (int i = 0; i < 5; i++)
#pragma unroll
                    for (int ex = 0; ex < 2; ex++) {
                        float4 row = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int k = 0; k < 5; k++) { const float4 t = pt[i][k]; const float wk = w5x[ex][k]; row.x += wk * t.x; row.y += wk * t.y; row.z += wk * t.z; row.w += wk * t.w; }
                        rs[i][ex] = row;
                    }
#pragma unroll
                for (int ey = 0; ey < 2; ey++)
#pragma unroll
                    for (int ex = 0; ex < 2; ex++) {
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int i = 0; i < 5; i++) { const float4 row = rs[i][ex]; const float wi = w5y[ey][i]; v.x += wi * row.x; v.y += wi * row.y; v.z += wi * row.z; v.w += wi * row.w; }
                        const int oy = 2 * qy + ey, ox = 2 * qx + ex;
                        if (skip) { const float4 k4 = *reinterpret_cast<const float4 *>(skip + ((size_t)b * HW + (size_t)oy * w + ox) * C + 4 * c4); v.x += k4.x; v.y += k4.y; v.z += k4.z; v.w += k4.w; }
                        emit(oy, ox, v);
                    }
            }
        }
    if (!part) return;
    __shared__ float red[256 * 8];
#pragma unroll
    for (int k = 0; k < 4; k++) { red[threadIdx.x * 8 + k] = s[k]; red[threadIdx.x * 8 + 4 + k] = q[k]; }
    __syncthreads();
    if (threadIdx.x < C4) {
        double ds[4] = {0, 0, 0, 0}, dq[4] = {0, 0, 0, 0};
        for (int sl = 0; sl < nslot; sl++)
#pragma unroll
            for (int k = 0; k < 4; k++) { ds[k] += (double)red[(sl * C4 + threadIdx.x) * 8 + k]; dq[k] += (double)red[(sl * C4 + threadIdx.x) * 8 + 4 + k]; }
        double *o = part + (((size_t)blockIdx.x * gridDim.y + b) * C + 4 * threadIdx.x) * 2;
#pragma unroll
        for (int k = 0; k < 4; k++) { o[2 * k] = ds[k]; o[2 * k + 1] = dq[k]; }
    }
}
// blocks of output pixels per frame of the sweeping producers: enough workgroups to fill the chip at B = 16 .. 48, few enough partials to finalize
static int sweep_blocks(int HW) { return min(max(HW / 64, 1), 256); }
extern "C" int vt_sweep_blocks(int HW) { return HW > 0 ? sweep_blocks(HW) : 0; }
template <int OP>
static int sweep_launch(const float *a, const float *skip, int B, int h, int w, int C, int groups, float *out, double *stats_ws, hipStream_t st)
{
    const int HW = h * w, nblk = sweep_blocks(HW), items = OP == 0 ? HW : HW / 4, rows = (items + nblk - 1) / nblk;
    double *part = stats_ws ? stats_ws + (size_t)B * groups : nullptr;
    hipLaunchKernelGGL(sweep_stats_kernel<OP>, dim3(nblk, B), dim3(256), 0, st, a, skip, h, w, C, rows, out, part);
    VT_LAUNCH_CHECK();
    return VT_OK;
}
extern "C" int vt_avgpool2x2_stats(const float *x, int B, int H, int W, int C, float *out, double *stats_ws, int stats_groups, void *stream)
{
    VT_REQUIRE(x && out && B > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 4 == 0 && C <= 1024 && (!stats_ws || stats_groups > 0),
               "vt_avgpool2x2_stats: bad argument (even H, W; C a multiple of 4, <= 1024)");
    return sweep_launch<0>(x, nullptr, B, H / 2, W / 2, C, stats_groups, out, stats_ws, vt_stream(stream));
}
extern "C" int vt_upsample2x_bicubic_add_stats(const float *low, const float *skip, int B, int h, int w, int C, float *out, double *stats_ws, int stats_groups,
                                               void *stream)
{
    VT_REQUIRE(low && out && B > 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0 && C <= 1024 && (!stats_ws || stats_groups > 0),
               "vt_upsample2x_bicubic_add_stats: bad argument (C a multiple of 4, <= 1024)");
    return sweep_launch<1>(low, skip, B, 2 * h, 2 * w, C, stats_groups, out, stats_ws, vt_stream(stream));
}
static int gn_blocks(int HW) { return min(max(HW / 256, 1), 128); }
// workspace of vt_groupnorm_nhwc / vt_groupnorm_stats in doubles: (B, groups) x {mean, rstd} as float pairs FIRST, then the block partials
extern "C" long vt_groupnorm_workspace_doubles(int B, int HW, int C, int groups)
{
    if (B <= 0 || HW <= 0 || C <= 0 || groups <= 0) return 0;
    return (long)B * groups + (long)gn_blocks(HW) * B * C * 2;
}
static int gn_statistics(const float *x, int cstride, int B, int HW, int C, int groups, float eps, double *ws, hipStream_t st)
{
    const int nblk = gn_blocks(HW), rows = (HW + nblk - 1) / nblk;
    double *part = ws + (size_t)B * groups;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nblk, B), dim3(256), 0, st, x, cstride, HW, C, rows, part);
    VT_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B * groups), dim3(64), 0, st, part, nblk, B, HW, C, groups, eps, reinterpret_cast<float2 *>(ws));
    VT_LAUNCH_CHECK();
    return VT_OK;
}
extern "C" int vt_groupnorm_nhwc(const float *x, const float *gamma, const float *beta, int B, int HW, int C, int groups, float eps, int relu,
                                 double *ws, float *y, void *stream)
{
    VT_REQUIRE(x && gamma && beta && ws && y && B > 0 && HW > 0 && C > 0 && C % 4 == 0 && C <= 1024 && groups > 0 && C % groups == 0,
               "vt_groupnorm_nhwc: bad argument (C must be a multiple of 4 and of groups, C <= 1024)");
    hipStream_t st = vt_stream(stream);
    if (int e = gn_statistics(x, C, B, HW, C, groups, eps, ws, st)) return e;
    const long total = (long)B * HW * (C / 4);
    hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, gamma, beta, reinterpret_cast<const float2 *>(ws), B, HW, C, groups, relu, y);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

// statistics only, on a channel slice [coff, coff + C) of an NHWC tensor with cstride channels: (B, groups) x {mean, 1 / sqrt(var + eps)} as floats
// at the START of ws (vt_groupnorm_workspace_doubles) -- the GroupNorm + ReLU itself is applied by the consumer (vt_conv3x3_forward_gn stages it into
// its operand planes)
extern "C" int vt_groupnorm_stats(const float *x, int cstride, int coff, int B, int HW, int C, int groups, float eps, double *ws, void *stream)
{
    VT_REQUIRE(x && ws && B > 0 && HW > 0 && C > 0 && C % 4 == 0 && groups > 0 && C % groups == 0 && C <= 1024 && cstride >= coff + C && cstride % 4 == 0 && coff % 4 == 0,
               "vt_groupnorm_stats: bad argument");
    return gn_statistics(x + coff, cstride, B, HW, C, groups, eps, ws, vt_stream(stream));
}

// second half of vt_groupnorm_stats for partial sums that somebody else produced: `part` = ws + B * groups doubles holds nblk x (B, C) x {sum, sum of
// squares} (vt_conv3x3_forward_gn_stats writes one block per output tile); the (B, groups) {mean, rstd} float pairs go to the start of ws
extern "C" int vt_groupnorm_finalize(double *ws, int nblk, int B, int HW, int C, int groups, float eps, void *stream)
{
    VT_REQUIRE(ws && nblk > 0 && B > 0 && HW > 0 && C > 0 && groups > 0 && C % groups == 0, "vt_groupnorm_finalize: bad argument");
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B * groups), dim3(64), 0, vt_stream(stream), ws + (size_t)B * groups, nblk, B, HW, C, groups, eps, reinterpret_cast<float2 *>(ws));
    VT_LAUNCH_CHECK();
    return VT_OK;
}

extern "C" int vt_upsample2x_bicubic_add(const float *low, const float *skip, int B, int h, int w, int C, float *out, void *stream)
{
    VT_REQUIRE(low && out && B > 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0, "vt_upsample2x_bicubic_add: bad argument (C must be a multiple of 4)");
    const long total = (long)B * 4 * h * w * (C / 4);
    hipLaunchKernelGGL(upsample2x_bicubic_add_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, vt_stream(stream), low, skip, B, h, w, C, out);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

extern "C" int vt_accel_loss(const float *v, int B, int D, const float *elem_w, float gscale, double *term, float *dv, void *stream)
{
    VT_REQUIRE(v && B >= 3 && D > 0, "vt_accel_loss: needs B >= 3 (the reference returns NaN for empty stencils)");
    // d/dv of mean(w a^2): 2 w a / cnt per stencil element; a's own coefficient 2 is folded in the kernel
    const float gs = 2.f * gscale / ((float)(B - 2) * (float)D);
    hipLaunchKernelGGL(accel_loss_kernel, dim3(min((B * D + ACCEL_T - 1) / ACCEL_T, ACCEL_MAX_BLOCKS)), dim3(ACCEL_T), 0, vt_stream(stream), v, B, D, D, elem_w, gs, term, dv);
    VT_LAUNCH_CHECK();
    return VT_OK;
}
extern "C" int vt_accel_loss_strided(const float *v, int B, int D, int stride, const float *elem_w, float gscale, double *term, float *dv, void *stream)
{
    VT_REQUIRE(v && B >= 3 && D > 0 && stride >= D, "vt_accel_loss_strided: needs B >= 3 and stride >= D");
    const float gs = 2.f * gscale / ((float)(B - 2) * (float)D);
    hipLaunchKernelGGL(accel_loss_kernel, dim3(min((B * D + ACCEL_T - 1) / ACCEL_T, ACCEL_MAX_BLOCKS)), dim3(ACCEL_T), 0, vt_stream(stream), v, B, D, stride, elem_w, gs, term, dv);
    VT_LAUNCH_CHECK();
    return VT_OK;
}
extern "C" int vt_velocity_loss(const float *v, int B, int D, float gscale, double *term, float *dv, void *stream)
{
    VT_REQUIRE(v && B >= 2 && D > 0, "vt_velocity_loss: needs B >= 2");
    const float gs = 2.f * gscale / ((float)(B - 1) * (float)D);
    hipLaunchKernelGGL(velocity_loss_kernel, dim3(min((B * D + 255) / 256, 512)), dim3(256), 0, vt_stream(stream), v, B, D, gs, term, dv);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

// ---------------------------------------------------------------------------------------------------
// 2D keypoint terms (fit_SMPLH_kpts.py:280-310; recon_fit_base.py:767-802)
// ---------------------------------------------------------------------------------------------------
struct Cam5 { float fx, fy, cx, cy, crop; };

__global__ __launch_bounds__(256) void kpts_loss_kernel(const float *J, const float *kpts, const float *cc, int BK, int K, int mode, Cam5 cam,
                                                        float net_size, float gscale, float inv_cnt, double *term, float *dJ)
{
    __shared__ double red[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    double acc = 0;
    if (i < BK) {
        const int b = i / K;
        const float x = J[3 * i], y = J[3 * i + 1], z = J[3 * i + 2];
        float px = cam.fx * x / z + cam.cx, py = cam.fy * y / z + cam.cy, sc = 1.f;
        if (mode == 1) {
            px = cam.crop / 2 + px - cc[2 * b]; py = cam.crop / 2 + py - cc[2 * b + 1];
            sc = net_size / cam.crop; px *= sc; py *= sc;
        }
        const float ex = px - kpts[3 * i], ey = py - kpts[3 * i + 1], conf = kpts[3 * i + 2];
        acc = (double)((ex * ex + ey * ey) * conf);
        const float gpx = 2.f * ex * conf * gscale * inv_cnt * sc, gpy = 2.f * ey * conf * gscale * inv_cnt * sc;
        dJ[3 * i] = gpx * cam.fx / z; dJ[3 * i + 1] = gpy * cam.fy / z;
        dJ[3 * i + 2] = -gpx * cam.fx * x / (z * z) - gpy * cam.fy * y / (z * z);
    }
    term_add(acc * (double)inv_cnt, term, red);
}
extern "C" int vt_kpts_loss(const float *J, const float *kpts, const float *crop_center, int B, int K, int mode, const float *cam,
                            float net_size, float gscale, double *term, float *dJ, void *stream)
{
    VT_REQUIRE(J && kpts && cam && dJ && B > 0 && K > 0 && (mode == 0 || (mode == 1 && crop_center)), "vt_kpts_loss: bad argument");
    Cam5 c{cam[0], cam[1], cam[2], cam[3], cam[4]};
    const float inv_cnt = 1.f / (mode == 0 ? (float)(B * K * 2) : (float)(B * K));
    hipLaunchKernelGGL(kpts_loss_kernel, dim3((B * K + 255) / 256), dim3(256), 0, vt_stream(stream), J, kpts, crop_center, B * K, K, mode, c,
                       net_size, gscale, inv_cnt, term, dJ);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

__global__ __launch_bounds__(256) void sqdiff_loss_kernel(const float *a, int as, const float *b, int bs, int rows, int cols, float inv_denom,
                                                          float gscale, double *term, float *da)
{
    __shared__ double red[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    double acc = 0;
    if (i < rows * cols) {
        const int r = i / cols, c = i % cols;
        const float d = a[(size_t)r * as + c] - b[(size_t)r * bs + c];
        acc = (double)(d * d);
        if (da) da[(size_t)r * as + c] += 2.f * d * inv_denom * gscale;
    }
    term_add(acc * (double)inv_denom, term, red);
}
extern "C" int vt_sqdiff_loss(const float *a, int a_stride, const float *b, int b_stride, int rows, int cols, float denom, float gscale,
                              double *term, float *da, void *stream)
{
    VT_REQUIRE(a && b && rows > 0 && cols > 0 && denom > 0, "vt_sqdiff_loss: bad argument");
    hipLaunchKernelGGL(sqdiff_loss_kernel, dim3((rows * cols + 255) / 256), dim3(256), 0, vt_stream(stream), a, a_stride, b, b_stride, rows, cols,
                       1.f / denom, gscale, term, da);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

// ---------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam single-tensor path) + device-side early stop
// ---------------------------------------------------------------------------------------------------
__global__ void adam_kernel(float *p, const float *g, float *m, float *v, long n, float step_size, float bc2s, float beta1, float beta2,
                            float eps, const int *stop_flag)
{
    if (stop_flag && *stop_flag) return;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = m[i] * beta1 + (1.f - beta1) * gi;
    const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2s + eps;
    p[i] = p[i] - step_size * (mi / denom);
}
extern "C" int vt_adam_step(float *p, const float *g, float *m, float *v, long n, int step, float lr, float beta1, float beta2, float eps,
                            const int *stop_flag, void *stream)
{
    VT_REQUIRE(p && g && m && v && n > 0 && step >= 1, "vt_adam_step: bad argument");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, vt_stream(stream), p, g, m, v, n, (float)(lr / bc1),
                       (float)sqrt(bc2), beta1, beta2, eps, stop_flag);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

__global__ void adam2d_kernel(float *p, long ps, const float *g, long gs, float *m, float *v, int rows, int cols, float step_size, float bc2s,
                              float beta1, float beta2, float eps, const int *stop_flag)
{
    if (stop_flag && *stop_flag) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int r = i / cols, c = i % cols;
    const float gi = g[(size_t)r * gs + c];
    const float mi = m[i] * beta1 + (1.f - beta1) * gi;
    const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2s + eps;
    p[(size_t)r * ps + c] = p[(size_t)r * ps + c] - step_size * (mi / denom);
}
extern "C" int vt_adam_step_2d(float *p, long p_stride, const float *g, long g_stride, float *m, float *v, int rows, int cols, int step, float lr,
                               float beta1, float beta2, float eps, const int *stop_flag, void *stream)
{
    VT_REQUIRE(p && g && m && v && rows > 0 && cols > 0 && step >= 1 && p_stride >= cols && g_stride >= cols, "vt_adam_step_2d: bad argument");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adam2d_kernel, dim3((rows * cols + 255) / 256), dim3(256), 0, vt_stream(stream), p, p_stride, g, g_stride, m, v, rows, cols,
                       (float)(lr / bc1), (float)sqrt(bc2), beta1, beta2, eps, stop_flag);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

struct TermW { float w[16]; };
__global__ void loss_reduce_kernel(const double *terms, TermW tw, int nterms, float tol, int armed, float *state, int *stop_flag,
                                   float *history, int slot)
{
    if (stop_flag && *stop_flag) { if (history) history[slot] = nanf(""); return; }
    double l = 0;
    for (int k = 0; k < nterms; k++) l += (double)tw.w[k] * terms[k];
    const float loss = (float)l, prev = state[0];
    if (history) history[slot] = loss;
    // reference: (abs(prev_loss - loss) / prev_loss < prev_loss * tol) and <iteration gate>
    if (armed && stop_flag && (fabsf(prev - loss) / prev < prev * tol)) *stop_flag = 1;
    state[0] = loss; state[1] = loss;
}
extern "C" int vt_loss_reduce_and_stop(const double *terms, const float *w, int nterms, float tol, int armed, float *state, int *stop_flag,
                                       float *history, int slot, void *stream)
{
    VT_REQUIRE(terms && w && state && nterms > 0 && nterms <= 16, "vt_loss_reduce_and_stop: bad argument (nterms <= 16)");
    TermW tw; for (int k = 0; k < 16; k++) tw.w[k] = k < nterms ? w[k] : 0.f;
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(1), 0, vt_stream(stream), terms, tw, nterms, tol, armed, state, stop_flag, history, slot);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

// =====================================================================================================================
// Fused heads / tails of an Adam step of the two fit loops (vt_objfit_step / vt_smplfit_step).  A step of the object stage
// used to be ~11 launches of 4-16 us kernels around the one query launch (SO(3) projection, rigid transform, two temporal stencils, rigid VJP,
// SO(3) VJP, one Adam launch per parameter group, loss reduction, term zeroing); the same arithmetic, element for element and in the same
// order, now runs as head -> query -> stencils -> tail.  Per-frame work is done by the workgroup of the frame; what needs every frame (loss
// reduction, stop rule, zeroing the term accumulators for the next step) is done by whichever workgroup finishes LAST (ticket counter), after
// every other workgroup has read the stop flag and stepped its parameters.
// =====================================================================================================================
// One workgroup of 1024 threads per frame (round 6; was ceil(N / 256) workgroups of 256): the projection's SVD is ~10 us of ONE thread, and every workgroup of a
// frame computed it while its other threads waited -- 12-24 x 96 workgroups holding their wave slots for the length of the SVD next to the other batches' query
// launches, for a point transform of microseconds.  Same arithmetic per point.
__global__ __launch_bounds__(1024) void objstep_head_kernel(const float *__restrict__ M0, const float *__restrict__ noise, const float *__restrict__ t,
                                                           const float *__restrict__ s, const float *__restrict__ X0p, int N, float *__restrict__ Xp,
                                                           const float *__restrict__ X0v, int NV, float *__restrict__ Xv, float *__restrict__ Rout,
                                                           double *terms, int nzero, float *__restrict__ svd_ws)
{
    __shared__ float sR[9];
    const int b = blockIdx.y;
    if (threadIdx.x == 0) {
        float M[9]; Svd3 sv;
#pragma unroll
        for (int e = 0; e < 9; e++) M[e] = M0[9 * b + e] + (noise ? 1e-4f * noise[9 * b + e] : 0.f);
        svd3(M, sv);
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) sR[3 * r + c] = sv.U[3 * r] * sv.V[3 * c] + sv.U[3 * r + 1] * sv.V[3 * c + 1] + sv.d * sv.U[3 * r + 2] * sv.V[3 * c + 2];
        if (blockIdx.x == 0) {
#pragma unroll
            for (int e = 0; e < 9; e++) Rout[9 * b + e] = sR[e];
            // the decomposition itself for the step's tail, which needs the SVD of the SAME matrix for the SO(3) VJP (round 6: the one-sided Jacobi SVD --
            // 24 rotations with two divisions and two square roots each, ~10 us of one thread -- was computed twice per step)
            if (svd_ws) {
                float *o = svd_ws + SVD_WS * b;
#pragma unroll
                for (int e = 0; e < 9; e++) { o[e] = sv.U[e]; o[9 + e] = sv.V[e]; }
                o[18] = sv.s[0]; o[19] = sv.s[1]; o[20] = sv.s[2]; o[21] = sv.d;
            }
        }
    }
    if (blockIdx.x == 0 && b == 0 && terms && (int)threadIdx.x < nzero) terms[threadIdx.x] = 0.0;
    __syncthreads();
    const float sc = s[b], t0 = t[3 * b], t1 = t[3 * b + 1], t2 = t[3 * b + 2];
    const float tt[3] = {t0, t1, t2};
    float r[9];
#pragma unroll
    for (int e = 0; e < 9; e++) r[e] = sR[e];
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
        const float *x = X0p + (size_t)n * 3; const float x0 = x[0], x1 = x[1], x2 = x[2];
        float *o = Xp + ((size_t)b * N + n) * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) o[c] = (x0 * r[c] + x1 * r[3 + c] + x2 * r[6 + c] + tt[c]) * sc;
    }
    if (Xv)
        for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < NV; n += gridDim.x * blockDim.x) {
            const float *x = X0v + (size_t)n * 3; const float x0 = x[0], x1 = x[1], x2 = x[2];
            float *o = Xv + ((size_t)b * NV + n) * 3;
#pragma unroll
            for (int c = 0; c < 3; c++) o[c] = (x0 * r[c] + x1 * r[3 + c] + x2 * r[6 + c] + tt[c]) * sc;
        }
}
extern "C" int vt_objstep_head(const float *M0, const float *noise, const float *t, const float *s, int B, const float *X0_points, int N, float *X_points,
                               const float *X0_verts, int NV, float *X_verts, float *R, double *terms, int nzero, float *svd_ws, void *stream)
{
    VT_REQUIRE(M0 && t && s && X0_points && X_points && R && B > 0 && N > 0 && (!X_verts || (X0_verts && NV > 0)) && nzero >= 0 && nzero <= 16, "vt_objstep_head: bad argument");
    hipLaunchKernelGGL(objstep_head_kernel, dim3(1, B), dim3(1024), 0, vt_stream(stream), M0, noise, t, s, X0_points, N, X_points, X0_verts, NV, X_verts,
                       R, terms, nzero, svd_ws);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

// acceleration + velocity stencils of (B, D) in one pass: dv (+)= gs_a (2 a_0 - a_m - a_p), then += gs_v (d_0 - d_1) -- the two updates of
// vt_accel_loss and vt_velocity_loss in their order; init_zero: dv starts from zero (phase 'sil': no query gradient, no fill launch)
__global__ __launch_bounds__(256) void temporal2_kernel(const float *__restrict__ v, int B, int D, float gs_a, double *term_a, float gs_v, double *term_v,
                                                        float *__restrict__ dv, int init_zero)
{
    __shared__ double red[4];
    double acc_a = 0, acc_v = 0;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < B * D; t += gridDim.x * 256) {
        const int f = t / D, i = t - f * D;
        auto at = [&](int b) { return v[(size_t)min(max(b, 0), B - 1) * D + i]; };
        const float vm2 = at(f - 2), vm1 = at(f - 1), v0 = at(f), vp1 = at(f + 1), vp2 = at(f + 2);
        const float a_m = (f - 1 >= 1 && f - 1 <= B - 2) ? 2.f * vm1 - vm2 - v0 : 0.f;
        const float a_0 = (f >= 1 && f <= B - 2) ? 2.f * v0 - vm1 - vp1 : 0.f;
        const float a_p = (f + 1 >= 1 && f + 1 <= B - 2) ? 2.f * vp1 - v0 - vp2 : 0.f;
        acc_a += (double)(1.f * a_0 * a_0);
        float g = init_zero ? 0.f : dv[(size_t)f * D + i];
        g += gs_a * 1.f * (2.f * a_0 - a_m - a_p);
        const float d0 = f >= 1 ? v0 - vm1 : 0.f;
        const float d1 = f + 1 < B ? vp1 - v0 : 0.f;
        acc_v += (double)(d0 * d0);
        g += gs_v * (d0 - d1);
        dv[(size_t)f * D + i] = g;
    }
    term_add(acc_a / ((double)(B - 2) * D), term_a, red);
    term_add(acc_v / ((double)(B - 1) * D), term_v, red);
}
extern "C" int vt_temporal_loss2(const float *v, int B, int D, float gscale_accel, double *term_accel, float gscale_velocity, double *term_velocity, float *dv,
                                 int init_zero, void *stream)
{
    VT_REQUIRE(v && dv && B >= 3 && D > 0, "vt_temporal_loss2: bad argument (B >= 3)");
    // derivative scales as in vt_accel_loss / vt_velocity_loss: d/dv of mean(a^2) resp. mean(d^2)
    const float gs_a = 2.f * gscale_accel / ((float)(B - 2) * (float)D), gs_v = 2.f * gscale_velocity / ((float)(B - 1) * (float)D);
    hipLaunchKernelGGL(temporal2_kernel, dim3(min((B * D + 255) / 256, 512)), dim3(256), 0, vt_stream(stream), v, B, D, gs_a, term_accel, gs_v, term_velocity, dv, init_zero);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

struct AdamSlice { float *p; int pstride; const float *g; int gstride; float *m, *v; int ncols; float step_size; };
struct StepEnd {
    const double *terms_r; double *terms_w; TermW tw; int nterms; float tol; int armed; float *state; int *stop_flag; float *history; int slot;
    int *ticket; int nzero;
};
// Adam on column c of row b of a slice (the arithmetic of adam2d_kernel)
__device__ __forceinline__ void adam_one(const AdamSlice &a, int b, int c, float bc2s, float beta1, float beta2, float eps)
{
    const int i = b * a.ncols + c;
    const float gi = a.g[(size_t)b * a.gstride + c];
    const float mi = a.m[i] * beta1 + (1.f - beta1) * gi;
    const float vi = a.v[i] * beta2 + (1.f - beta2) * gi * gi;
    a.m[i] = mi; a.v[i] = vi;
    const float denom = sqrtf(vi) / bc2s + eps;
    a.p[(size_t)b * a.pstride + c] = a.p[(size_t)b * a.pstride + c] - a.step_size * (mi / denom);
}
#ifndef STEP_END_FENCE
#define STEP_END_FENCE 0
#endif
// the workgroup that takes the last ticket closes the step: weighted loss, history, the reference's stop rule (loss_reduce_kernel), and the term
// accumulators [0, nzero) zeroed for the next step.  Every other workgroup has finished (its writes fenced) by then.
__device__ __forceinline__ void step_end(const StepEnd &e, int nblocks, bool stopped)
{
    __shared__ int last;
    // What the closing workgroup reads of the others are the TERM accumulators only, and those are device-scope atomics (performed at the memory side, dropped from
    // the XCD's L2) read back with agent-scope atomic loads: each wave waits until its own atomics have been performed (s_waitcnt vmcnt(0)) before the workgroup
    // takes its ticket.  No __threadfence(): on a multi-XCD part it writes the XCD's dirty L2 lines back and invalidates the L1 -- ~3.5 us per fencing workgroup,
    // 2-4 x that with all 256 threads fencing (MI355X_MICROARCH.md; measured round 6 in sil_image_kernel: 110 us with a fence per workgroup, 26 us without) -- for
    // plain stores (parameters, Adam moments, history) that nobody reads before the kernel boundary.  -DSTEP_END_FENCE=1 restores the fences.
#if STEP_END_FENCE
    __threadfence();
#else
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
    __syncthreads();
    if (threadIdx.x == 0) last = (atomicAdd(e.ticket, 1) == nblocks - 1);
    __syncthreads();
    if (!last) return;
#if STEP_END_FENCE
    __threadfence();
#endif
    if (threadIdx.x == 0) {
        *e.ticket = 0;
        if (stopped) { if (e.history) e.history[e.slot] = nanf(""); }
        else {
            double l = 0;
            for (int k = 0; k < e.nterms; k++) l += (double)e.tw.w[k] * __hip_atomic_load(e.terms_r + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float loss = (float)l, prev = e.state[0];
            if (e.history) e.history[e.slot] = loss;
            if (e.armed && e.stop_flag && (fabsf(prev - loss) / prev < prev * e.tol)) *e.stop_flag = 1;
            e.state[0] = loss; e.state[1] = loss;
        }
        for (int k = 0; k < e.nzero; k++) e.terms_w[k] = 0.0;
    }
}

// tail of an object-stage step, one workgroup per frame: rigid VJP over the vertex set (phase 'sil') and the surface points, the translation
// regulariser of phase 'sil', the SO(3) VJP, Adam on the frame's rotation parameters (9) and translation (3), then step_end
// vt_objstep_tail_temporal (round 6): the stencils of vt_temporal_loss2 evaluated INSIDE the tail while it reads the points' gradient -- g = dX (or 0: phase 'sil'),
// g += gs_a (2 a_0 - a_m - a_p), g += gs_v (d_0 - d_1): the float additions of temporal2_kernel in their order, so the rigid VJP sees the same bits -- one launch
// (19 us of launch-bound stencil work per object-stage step) less.  mode 0: off (dXp already holds everything), 1: add to dXp, 2: dXp is not read (phase 'sil').
struct TemporalIn { const float *X; float gs_a, gs_v; double *term_a, *term_v; int mode; };
__global__ __launch_bounds__(256) void objstep_tail_kernel(TemporalIn tin, const float *__restrict__ X0v, int NV, const float *__restrict__ dXv, const float *__restrict__ X0p, int N,
                                                           const float *__restrict__ dXp, const float *__restrict__ s, const float *__restrict__ M0,
                                                           const float *__restrict__ noise, const float *__restrict__ tpar, const float *__restrict__ t_init,
                                                           float w_trans, double *term_trans, float *__restrict__ dR, float *__restrict__ dt, float *__restrict__ dM,
                                                           AdamSlice aR, AdamSlice aT, float bc2s, float beta1, float beta2, float eps, StepEnd end,
                                                           const float *__restrict__ svd_ws)
{
    __shared__ float red12[4][12];
    __shared__ double redt[4];
    const int b = blockIdx.x, B = gridDim.x;
    const bool stopped = end.stop_flag && *end.stop_flag;          // read before any workgroup can close the step
    double acc_a = 0, acc_v = 0;
    // phase 'joint' optimises obj_t only (recon_fit_trivis_full.py:343-347: optim.Adam([obj_t], lr=0.002)): the rotation half of the rigid VJP (nine of
    // the twelve sums over the points) and the SO(3) VJP with its second Jacobi SVD feed nothing -- skipped when no rotation slice is optimised
    // (dR / dM are then left untouched; obj_t takes the same three sums in the same order: bit-identical parameters)
    const bool rot = aR.p != nullptr;
    const float sc = s[b];
    float svw[SVD_WS];              // thread 0: the head's SVD of this frame, requested before the sums over the points so that its latency hides behind them
    if (svd_ws && rot && threadIdx.x == 0) {
#pragma unroll
        for (int e = 0; e < SVD_WS; e++) svw[e] = svd_ws[SVD_WS * b + e];
    }
    float tot[12];
#pragma unroll
    for (int e = 0; e < 12; e++) tot[e] = 0.f;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const float *X0 = pass == 0 ? X0v : X0p; const float *dX = pass == 0 ? dXv : dXp; const int n_ = pass == 0 ? NV : N;
        if (!dX) continue;
        float a[12];
#pragma unroll
        for (int e = 0; e < 12; e++) a[e] = 0.f;
        for (int n = threadIdx.x; n < n_; n += 256) {
            const float *x = X0 + (size_t)n * 3; const float *g = dX + ((size_t)b * n_ + n) * 3;
            float gt[3];
            if (pass == 1 && tin.mode) {
                // temporal2_kernel for the three elements of point n in frame b (frames clamped exactly like there)
                const int D = N * 3, f = b;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const int i = n * 3 + c;
                    const float vm2 = tin.X[(size_t)min(max(f - 2, 0), B - 1) * D + i], vm1 = tin.X[(size_t)min(max(f - 1, 0), B - 1) * D + i], v0 = tin.X[(size_t)f * D + i],
                                vp1 = tin.X[(size_t)min(max(f + 1, 0), B - 1) * D + i], vp2 = tin.X[(size_t)min(max(f + 2, 0), B - 1) * D + i];
                    const float a_m = (f - 1 >= 1 && f - 1 <= B - 2) ? 2.f * vm1 - vm2 - v0 : 0.f;
                    const float a_0 = (f >= 1 && f <= B - 2) ? 2.f * v0 - vm1 - vp1 : 0.f;
                    const float a_p = (f + 1 >= 1 && f + 1 <= B - 2) ? 2.f * vp1 - v0 - vp2 : 0.f;
                    acc_a += (double)(1.f * a_0 * a_0);
                    float gg = tin.mode == 2 ? 0.f : g[c];
                    gg += tin.gs_a * 1.f * (2.f * a_0 - a_m - a_p);
                    const float d0 = f >= 1 ? v0 - vm1 : 0.f;
                    const float d1 = f + 1 < B ? vp1 - v0 : 0.f;
                    acc_v += (double)(d0 * d0);
                    gg += tin.gs_v * (d0 - d1);
                    gt[c] = gg;
                }
                g = gt;
            }
            if (rot) {
#pragma unroll
                for (int c = 0; c < 3; c++) { const float gc = g[c] * sc; a[9 + c] += gc; a[c] += x[0] * gc; a[3 + c] += x[1] * gc; a[6 + c] += x[2] * gc; }
            } else {
#pragma unroll
                for (int c = 0; c < 3; c++) a[9 + c] += g[c] * sc;
            }
        }
        // the twelve block sums of rigid_bwd_kernel (wave tree, then the four waves in order: the same additions) with ONE barrier pair instead of twelve
#pragma unroll
        for (int e = 0; e < 12; e++) a[e] = wave_sum(a[e]);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int e = 0; e < 12; e++) red12[threadIdx.x >> 6][e] = a[e];
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 12; e++) {
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < 4; i++) v += red12[i][e];
            tot[e] = (pass == 0 || !dXv) ? v : tot[e] + v;         // rigid_bwd_kernel: first set written, second accumulated
        }
        if (pass == 0 && t_init) {
            // trans = mean_{B,3} (t - t_init)^2  (vt_sqdiff_loss with denom 3 B) adds its gradient to dt BETWEEN the two rigid VJPs in the unfused
            // sequence (vertex set, regulariser, surface points): same order of the three float additions here
            const float inv_denom = 1.f / (float)(3 * B);
#pragma unroll
            for (int c = 0; c < 3; c++) { const float d = tpar[3 * b + c] - t_init[3 * b + c]; tot[9 + c] += 2.f * d * inv_denom * w_trans; }
        }
    }
    if (tin.mode) {
        // the frame's share of the two stencil terms (vt_temporal_loss2 adds per workgroup of 256 elements: the same fp64 atomics, another grouping)
        term_add(acc_a / ((double)(B - 2) * (N * 3)), tin.term_a, redt);
        term_add(acc_v / ((double)(B - 1) * (N * 3)), tin.term_v, redt);
    }
    if (threadIdx.x == 0) {
        float g[12];
#pragma unroll
        for (int e = 0; e < 12; e++) g[e] = tot[e];
        if (t_init) {
            const float inv_denom = 1.f / (float)(3 * B);
            double acc = 0;
#pragma unroll
            for (int c = 0; c < 3; c++) { const float d = tpar[3 * b + c] - t_init[3 * b + c]; acc += (double)(d * d); }
            atomicAdd(term_trans, acc * (double)inv_denom);        // the frame's share of the term
        }
#pragma unroll
        for (int c = 0; c < 3; c++) dt[3 * b + c] = g[9 + c];
        if (rot) {
        // SO(3) VJP (so3_bwd_kernel)
        float M[9], G[9]; Svd3 sv;
#pragma unroll
        for (int e = 0; e < 9; e++) G[e] = g[e];
        if (svd_ws) {               // the step's head decomposed this matrix already (vt_objstep_head with the same workspace): the same numbers
#pragma unroll
            for (int e = 0; e < 9; e++) { sv.U[e] = svw[e]; sv.V[e] = svw[9 + e]; }
            sv.s[0] = svw[18]; sv.s[1] = svw[19]; sv.s[2] = svw[20]; sv.d = svw[21];
        } else {
#pragma unroll
            for (int e = 0; e < 9; e++) M[e] = M0[9 * b + e] + (noise ? 1e-4f * noise[9 * b + e] : 0.f);
            svd3(M, sv);
        }
        const float D[3] = {1.f, 1.f, sv.d}, h[3] = {sv.s[0], sv.s[1], sv.d * sv.s[2]};
        float UtG[9], Q[9], Z[9], UDZ[9];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) UtG[3 * r + c] = sv.U[r] * G[c] + sv.U[3 + r] * G[3 + c] + sv.U[6 + r] * G[6 + c];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) Q[3 * r + c] = D[r] * (UtG[3 * r] * sv.V[c] + UtG[3 * r + 1] * sv.V[3 + c] + UtG[3 * r + 2] * sv.V[6 + c]);
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) Z[3 * r + c] = (r == c) ? 0.f : (Q[3 * r + c] - Q[3 * c + r]) / (h[r] + h[c]);
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) UDZ[3 * r + c] = sv.U[3 * r] * D[0] * Z[c] + sv.U[3 * r + 1] * D[1] * Z[3 + c] + sv.U[3 * r + 2] * D[2] * Z[6 + c];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) { const float v = UDZ[3 * r] * sv.V[3 * c] + UDZ[3 * r + 1] * sv.V[3 * c + 1] + UDZ[3 * r + 2] * sv.V[3 * c + 2]; dM[9 * b + 3 * r + c] = v; }
#pragma unroll
        for (int e = 0; e < 9; e++) dR[9 * b + e] = g[e];
        }
    }
    __syncthreads();
    if (!stopped) {
        // Adam reads the gradients the way adam2d_kernel does: from the gradient tensors (dM, dt) just written by thread 0 of this workgroup
        if (aR.p && threadIdx.x < 9) adam_one(aR, b, threadIdx.x, bc2s, beta1, beta2, eps);
        if (aT.p && threadIdx.x >= 64 && threadIdx.x < 67) adam_one(aT, b, threadIdx.x - 64, bc2s, beta1, beta2, eps);
    }
    step_end(end, B, stopped);
}
static StepEnd make_end(double *terms, const float *w, int nterms, float tol, int armed, float *state, int *stop_flag, float *history, int slot, int *ticket, int nzero)
{
    StepEnd e; e.terms_r = terms; e.terms_w = terms; e.nterms = nterms; e.tol = tol; e.armed = armed; e.state = state; e.stop_flag = stop_flag; e.history = history; e.slot = slot;
    e.ticket = ticket; e.nzero = nzero;
    for (int k = 0; k < 16; k++) e.tw.w[k] = k < nterms ? w[k] : 0.f;
    return e;
}
extern "C" int vt_objstep_tail(const float *X0_verts, int NV, const float *dX_verts, const float *X0_points, int N, const float *dX_points, const float *s, int B,
                               const float *M0, const float *noise, const float *t, const float *t_init, float w_trans, double *term_trans,
                               float *dR, float *dt, float *dM,
                               float *pR, float *mR, float *vR, float lrR, float *pT, float *mT, float *vT, float lrT, int adam_step, float beta1, float beta2, float eps,
                               double *terms, const float *w, int nterms, float tol, int armed, float *state, int *stop_flag, float *history, int slot, int *ticket, int nzero,
                               float *svd_ws, void *stream)
{
    VT_REQUIRE(X0_points && dX_points && s && M0 && t && dR && dt && dM && B > 0 && N > 0 && (!dX_verts || (X0_verts && NV > 0)) && (!t_init || term_trans), "vt_objstep_tail: bad argument");
    VT_REQUIRE(terms && w && state && ticket && nterms > 0 && nterms <= 16 && nzero >= 0 && nzero <= nterms && adam_step >= 1 && (!pR || (mR && vR)) && (!pT || (mT && vT)),
               "vt_objstep_tail: bad optimiser / loss arguments");
    const double bc1 = 1.0 - pow((double)beta1, adam_step), bc2 = 1.0 - pow((double)beta2, adam_step);
    AdamSlice aR = {pR, 9, dM, 9, mR, vR, 9, (float)(lrR / bc1)}, aT = {pT, 3, dt, 3, mT, vT, 3, (float)(lrT / bc1)};
    const TemporalIn tin = {nullptr, 0.f, 0.f, nullptr, nullptr, 0};
    hipLaunchKernelGGL(objstep_tail_kernel, dim3(B), dim3(256), 0, vt_stream(stream), tin, X0_verts, NV, dX_verts, X0_points, N, dX_points, s, M0, noise, t, t_init, w_trans, term_trans,
                       dR, dt, dM, aR, aT, (float)sqrt(bc2), beta1, beta2, eps, make_end(terms, w, nterms, tol, armed, state, stop_flag, history, slot, ticket, nzero), svd_ws);
    VT_LAUNCH_CHECK();
    return VT_OK;
}
extern "C" int vt_objstep_tail_temporal(const float *X_points, float gscale_accel, double *term_accel, float gscale_velocity, double *term_velocity, int init_zero,
                                        const float *X0_verts, int NV, const float *dX_verts, const float *X0_points, int N, const float *dX_points, const float *s, int B,
                                        const float *M0, const float *noise, const float *t, const float *t_init, float w_trans, double *term_trans,
                                        float *dR, float *dt, float *dM,
                                        float *pR, float *mR, float *vR, float lrR, float *pT, float *mT, float *vT, float lrT, int adam_step, float beta1, float beta2, float eps,
                                        double *terms, const float *w, int nterms, float tol, int armed, float *state, int *stop_flag, float *history, int slot, int *ticket, int nzero,
                                        float *svd_ws, void *stream)
{
    VT_REQUIRE(X_points && term_accel && term_velocity && B >= 3, "vt_objstep_tail_temporal: bad argument (B >= 3)");
    VT_REQUIRE(X0_points && dX_points && s && M0 && t && dR && dt && dM && B > 0 && N > 0 && (!dX_verts || (X0_verts && NV > 0)) && (!t_init || term_trans), "vt_objstep_tail_temporal: bad argument");
    VT_REQUIRE(terms && w && state && ticket && nterms > 0 && nterms <= 16 && nzero >= 0 && nzero <= nterms && adam_step >= 1 && (!pR || (mR && vR)) && (!pT || (mT && vT)),
               "vt_objstep_tail_temporal: bad optimiser / loss arguments");
    const double bc1 = 1.0 - pow((double)beta1, adam_step), bc2 = 1.0 - pow((double)beta2, adam_step);
    AdamSlice aR = {pR, 9, dM, 9, mR, vR, 9, (float)(lrR / bc1)}, aT = {pT, 3, dt, 3, mT, vT, 3, (float)(lrT / bc1)};
    const int D = N * 3;
    const TemporalIn tin = {X_points, 2.f * gscale_accel / ((float)(B - 2) * (float)D), 2.f * gscale_velocity / ((float)(B - 1) * (float)D), term_accel, term_velocity, init_zero ? 2 : 1};
    hipLaunchKernelGGL(objstep_tail_kernel, dim3(B), dim3(256), 0, vt_stream(stream), tin, X0_verts, NV, dX_verts, X0_points, N, dX_points, s, M0, noise, t, t_init, w_trans, term_trans,
                       dR, dt, dM, aR, aT, (float)sqrt(bc2), beta1, beta2, eps, make_end(terms, w, nterms, tol, armed, state, stop_flag, history, slot, ticket, nzero), svd_ws);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

// tail of a SMPL-stage step, one 64-thread workgroup per frame: body-pose prior (mahalanobis_kernel, n = 63 at pose[:, 3:66]) with its gradient,
// the pose-initialisation term mean_B sum (pose[:, 3:72] - pose_init)^2 (vt_sqdiff_loss), Adam on up to three column slices, step_end
__global__ __launch_bounds__(64) void smplstep_tail_kernel(float *__restrict__ pose, const float *__restrict__ pose_init, float *__restrict__ dpose,
                                                           const float *__restrict__ mean, const float *__restrict__ prec, float gscale, double *term_prior,
                                                           float w_pinit, double *term_pinit, AdamSlice a0, AdamSlice a1, AdamSlice a2, float bc2s, float beta1,
                                                           float beta2, float eps, StepEnd end)
{
    __shared__ float d[64], t2[64];
    const int b = blockIdx.x, B = gridDim.x, j = threadIdx.x, n = 63, off = 3, stride = 156;
    const bool stopped = end.stop_flag && *end.stop_flag;
    d[j] = (j < n) ? pose[(size_t)b * stride + off + j] - mean[j] : 0.f;
    __syncthreads();
    float a = 0.f;
    if (j < n) for (int i = 0; i < n; i++) a += d[i] * prec[i * n + j];
    t2[j] = a;
    const float val = wave_sum(a * a);
    __syncthreads();
    if (j < n) {
        float g = 0.f;
        for (int k = 0; k < n; k++) g += t2[k] * prec[j * n + k];
        dpose[(size_t)b * stride + off + j] += 2.f * g * gscale;
    }
    // pinit over columns 3 .. 71 (69 of them): thread j takes columns j and j + 64
    double acc = 0;
    const float inv_denom = 1.f / (float)B;
    for (int c = j; c < 69; c += 64) {
        const float dd = pose[(size_t)b * stride + 3 + c] - pose_init[(size_t)b * stride + 3 + c];
        acc += (double)(dd * dd);
        dpose[(size_t)b * stride + 3 + c] += 2.f * dd * inv_denom * w_pinit;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (j == 0) { atomicAdd(term_pinit, acc * (double)inv_denom); atomicAdd(term_prior, (double)val / (double)B); }
    __syncthreads();          // the frame's gradients are complete (same workgroup wrote them)
    if (!stopped) {
        const AdamSlice *sl[3] = {&a0, &a1, &a2};
#pragma unroll
        for (int k = 0; k < 3; k++)
            if (sl[k]->p) for (int c = j; c < sl[k]->ncols; c += 64) adam_one(*sl[k], b, c, bc2s, beta1, beta2, eps);
    }
    step_end(end, B, stopped);
}
extern "C" int vt_smplstep_tail(float *pose, const float *pose_init, float *dpose, int B, const float *prior_mean, const float *prior_prec, float gscale_prior,
                                double *term_prior, float w_pinit, double *term_pinit,
                                float *p0, int ps0, const float *g0, int gs0, float *m0, float *v0, int n0, float lr0,
                                float *p1, int ps1, const float *g1, int gs1, float *m1, float *v1, int n1, float lr1,
                                float *p2, int ps2, const float *g2, int gs2, float *m2, float *v2, int n2, float lr2,
                                int adam_step, float beta1, float beta2, float eps,
                                double *terms, const float *w, int nterms, float tol, int armed, float *state, int *stop_flag, float *history, int slot, int *ticket, int nzero,
                                void *stream)
{
    VT_REQUIRE(pose && pose_init && dpose && prior_mean && prior_prec && term_prior && term_pinit && B > 0 && adam_step >= 1, "vt_smplstep_tail: bad argument");
    VT_REQUIRE(terms && w && state && ticket && nterms > 0 && nterms <= 16 && nzero >= 0 && nzero <= nterms, "vt_smplstep_tail: bad loss arguments");
    const double bc1 = 1.0 - pow((double)beta1, adam_step), bc2 = 1.0 - pow((double)beta2, adam_step);
    AdamSlice a0 = {p0, ps0, g0, gs0, m0, v0, n0, (float)(lr0 / bc1)}, a1 = {p1, ps1, g1, gs1, m1, v1, n1, (float)(lr1 / bc1)}, a2 = {p2, ps2, g2, gs2, m2, v2, n2, (float)(lr2 / bc1)};
    hipLaunchKernelGGL(smplstep_tail_kernel, dim3(B), dim3(64), 0, vt_stream(stream), pose, pose_init, dpose, prior_mean, prior_prec, gscale_prior, term_prior, w_pinit, term_pinit,
                       a0, a1, a2, (float)sqrt(bc2), beta1, beta2, eps, make_end(terms, w, nterms, tol, armed, state, stop_flag, history, slot, ticket, nzero));
    VT_LAUNCH_CHECK();
    return VT_OK;
}


// the keypoint chain of a SMPL-stage step in one launch, one workgroup per frame: body25 joints J = regressor . verts (landmarks_fwd_kernel: wave per
// joint, lane-strided sum, wave tree), the 2-D keypoint term and dJ (kpts_loss_kernel), d verts = regressor^T dJ written -- not accumulated -- for every
// vertex (landmarks_bwd_kernel): the query launch that follows adds its gradient to it (vt_query_human_step)
__global__ __launch_bounds__(1024) void kpts_step_kernel(const int *__restrict__ indptr, const int *__restrict__ indices, const float *__restrict__ data,
                                                        const int *__restrict__ colptr, const int *__restrict__ rowidx, const float *__restrict__ cdata,
                                                        const float *__restrict__ verts, int V, int K, const float *__restrict__ kpts, const float *__restrict__ cc,
                                                        int mode, Cam5 cam, float net_size, float gscale, float inv_cnt, double *term, float *__restrict__ Jout,
                                                        float *__restrict__ dverts, int accumulate)
{
    __shared__ float sJ[64 * 3], sdJ[64 * 3];
    __shared__ double red[16];
    // 16 waves per frame (round 6; was 4): the landmark rows and the vertex columns are chains of dependent gathers, 7 rows / 27 columns deep per thread
    // with 256 threads -- 83 us of latency for microseconds of work; same sums in the same order
    const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    for (int k = wave; k < K; k += nw) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int e = indptr[k] + lane; e < indptr[k + 1]; e += 64) {
            const float w = data[e]; const float *v = verts + ((size_t)b * V + indices[e]) * 3;
            a0 += w * v[0]; a1 += w * v[1]; a2 += w * v[2];
        }
        a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
        if (lane == 0) {
            sJ[3 * k] = a0; sJ[3 * k + 1] = a1; sJ[3 * k + 2] = a2;
            if (Jout) { float *o = Jout + ((size_t)b * K + k) * 3; o[0] = a0; o[1] = a1; o[2] = a2; }
        }
    }
    __syncthreads();
    double acc = 0;
    if ((int)threadIdx.x < K) {
        const int k = threadIdx.x, i = b * K + k;
        const float x = sJ[3 * k], y = sJ[3 * k + 1], z = sJ[3 * k + 2];
        float px = cam.fx * x / z + cam.cx, py = cam.fy * y / z + cam.cy, sc = 1.f;
        if (mode == 1) {
            px = cam.crop / 2 + px - cc[2 * b]; py = cam.crop / 2 + py - cc[2 * b + 1];
            sc = net_size / cam.crop; px *= sc; py *= sc;
        }
        const float ex = px - kpts[3 * i], ey = py - kpts[3 * i + 1], conf = kpts[3 * i + 2];
        acc = (double)((ex * ex + ey * ey) * conf);
        const float gpx = 2.f * ex * conf * gscale * inv_cnt * sc, gpy = 2.f * ey * conf * gscale * inv_cnt * sc;
        sdJ[3 * k] = gpx * cam.fx / z; sdJ[3 * k + 1] = gpy * cam.fy / z;
        sdJ[3 * k + 2] = -gpx * cam.fx * x / (z * z) - gpy * cam.fy * y / (z * z);
    }
    term_add(acc * (double)inv_cnt, term, red);          // (its barriers also publish sdJ)
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
        const int s = colptr[v], e = colptr[v + 1];
        if (s == e && accumulate) continue;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int i = s; i < e; i++) { const float w = cdata[i]; const float *g = sdJ + rowidx[i] * 3; a0 += w * g[0]; a1 += w * g[1]; a2 += w * g[2]; }
        float *o = dverts + ((size_t)b * V + v) * 3;
        if (accumulate) { o[0] += a0; o[1] += a1; o[2] += a2; } else { o[0] = a0; o[1] = a1; o[2] = a2; }
    }
}
extern "C" int vt_kpts_step(const vt_landmarks *h, const float *verts, const float *kpts, const float *crop_center, int B, int mode, const float *cam, float net_size,
                            float gscale, double *term, float *J, float *dverts, int accumulate, void *stream)
{
    VT_REQUIRE(h && verts && kpts && cam && dverts && B > 0 && h->K <= 64 && (mode == 0 || (mode == 1 && crop_center)), "vt_kpts_step: bad argument (at most 64 landmarks)");
    Cam5 c{cam[0], cam[1], cam[2], cam[3], cam[4]};
    const float inv_cnt = 1.f / (mode == 0 ? (float)(B * h->K * 2) : (float)(B * h->K));
    hipLaunchKernelGGL(kpts_step_kernel, dim3(B), dim3(1024), 0, vt_stream(stream), h->indptr, h->indices, h->data, h->colptr, h->rowidx, h->cdata, verts, h->V, h->K,
                       kpts, crop_center, mode, c, net_size, gscale, inv_cnt, term, J, dverts, accumulate);
    VT_LAUNCH_CHECK();
    return VT_OK;
}

// ---- device-side early stop: per-stream skip flag -------------------------------------------------------------------------------------------
// The stop rules of the fits are evaluated on the device (vt_loss_reduce_and_stop / the step tails) and the host looks at the flag once per outer
// iteration of 10 steps, so up to 9 steps are already queued behind the step that stopped the fit.  Adam and the loss history ignore them; with the
// flag registered for the stream the query and SMPL-H kernels of those steps return at their first instruction as well (results unchanged: the
// reference breaks out of its loop at that step, recon_fit_behave.py:447).  The registry is PER HOST THREAD (thread_local), keyed by (device, stream):
// a fit registers its flag from the thread that issues its launches and only that thread's launches see it -- two fits driven by two threads
// through the SAME stream (e.g. both on the default stream) can neither pick up nor delete each other's flag.
#include <vector>
struct SkipEntry { int dev; hipStream_t st; const int *flag; };
static thread_local std::vector<SkipEntry> t_skip_tab;       // (the default stream is the same handle on every device: the device is part of the key)
static int skip_device() { int d = 0; (void)hipGetDevice(&d); return d; }
const int *vt_skip_flag_of(hipStream_t st)
{
    if (t_skip_tab.empty()) return nullptr;
    const int dev = skip_device();
    for (const auto &e : t_skip_tab) if (e.st == st && e.dev == dev) return e.flag;
    return nullptr;
}
extern "C" int vt_stream_set_skip_flag(void *stream, const int *flag)
{
    const hipStream_t st = vt_stream(stream); const int dev = skip_device();
    for (size_t i = 0; i < t_skip_tab.size(); i++)
        if (t_skip_tab[i].st == st && t_skip_tab[i].dev == dev) {
            // a second fit of THIS thread on the stream while the first one's flag is still registered (nested / leaked registration): refuse instead of
            // redirecting the first fit's launches to another flag
            if (flag && t_skip_tab[i].flag != flag) VT_FAIL(VT_ERR_BUSY, "vt_stream_set_skip_flag: another stop flag is already registered for this stream by this thread");
            if (!flag) { t_skip_tab[i] = t_skip_tab.back(); t_skip_tab.pop_back(); }
            return VT_OK;
        }
    if (flag) t_skip_tab.push_back({dev, st, flag});
    return VT_OK;
}
