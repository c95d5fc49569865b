// collide_geom.h -- geometry of the interpenetration term (collide.hip): boxes, Moller's triangle-triangle interval test and the conic
// distance field of Tzionas et al. 2016.  Plain C functions in float so that the device code and the host code agree instruction for instruction.
#pragma once
#ifdef __HIPCC__
#define COL_FN __host__ __device__ static inline
#else
#include <math.h>
#define COL_FN static inline
#endif

COL_FN void col_tri_box(const float *T, float *lo, float *hi)
{
    for (int k = 0; k < 3; k++) { lo[k] = fminf(T[k], fminf(T[3 + k], T[6 + k])); hi[k] = fmaxf(T[k], fmaxf(T[3 + k], T[6 + k])); }
}
COL_FN int col_box_overlap(const float *alo, const float *ahi, const float *blo, const float *bhi)
{
    return alo[0] <= bhi[0] && blo[0] <= ahi[0] && alo[1] <= bhi[1] && blo[1] <= ahi[1] && alo[2] <= bhi[2] && blo[2] <= ahi[2];
}
COL_FN void col_cross(const float *a, const float *b, float *c) { c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0]; }
COL_FN float col_dot(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// interval of triangle (projections p0..2 on the intersection line, signed plane distances d0..2) on the line; vertex `a` is alone on its side
COL_FN void col_interval(float pa, float pb, float pc, float da, float db, float dc, float *t0, float *t1)
{
    *t0 = pa + (pb - pa) * da / (da - db);
    *t1 = pa + (pc - pa) * da / (da - dc);
}
COL_FN int col_intervals(const float *p, const float *d, float *t0, float *t1)
{
    const float d01 = d[0] * d[1], d02 = d[0] * d[2];
    if (d01 > 0.f) col_interval(p[2], p[0], p[1], d[2], d[0], d[1], t0, t1);          // 0, 1 on one side, 2 alone
    else if (d02 > 0.f) col_interval(p[1], p[0], p[2], d[1], d[0], d[2], t0, t1);     // 0, 2 on one side, 1 alone
    else if (d[1] * d[2] > 0.f || d[0] != 0.f) col_interval(p[0], p[1], p[2], d[0], d[1], d[2], t0, t1);
    else if (d[1] != 0.f) col_interval(p[1], p[0], p[2], d[1], d[0], d[2], t0, t1);
    else if (d[2] != 0.f) col_interval(p[2], p[0], p[1], d[2], d[0], d[1], t0, t1);
    else return 0;                                                                       // coplanar: treated as not intersecting
    return 1;
}
// Moller 1997, "A fast triangle-triangle intersection test" (interval overlap on the line of the two planes); coplanar pairs -> 0
COL_FN int col_tri_tri(const float *A, const float *B)
{
    float e1[3], e2[3], n1[3], n2[3], da[3], db[3], D[3];
    for (int k = 0; k < 3; k++) { e1[k] = B[3 + k] - B[k]; e2[k] = B[6 + k] - B[k]; }
    col_cross(e1, e2, n2);
    const float off2 = -col_dot(n2, B);
    for (int i = 0; i < 3; i++) da[i] = col_dot(n2, A + 3 * i) + off2;
    if ((da[0] > 0.f && da[1] > 0.f && da[2] > 0.f) || (da[0] < 0.f && da[1] < 0.f && da[2] < 0.f)) return 0;
    for (int k = 0; k < 3; k++) { e1[k] = A[3 + k] - A[k]; e2[k] = A[6 + k] - A[k]; }
    col_cross(e1, e2, n1);
    const float off1 = -col_dot(n1, A);
    for (int i = 0; i < 3; i++) db[i] = col_dot(n1, B + 3 * i) + off1;
    if ((db[0] > 0.f && db[1] > 0.f && db[2] > 0.f) || (db[0] < 0.f && db[1] < 0.f && db[2] < 0.f)) return 0;
    col_cross(n1, n2, D);
    int ax = 0; float m = fabsf(D[0]);
    if (fabsf(D[1]) > m) { m = fabsf(D[1]); ax = 1; }
    if (fabsf(D[2]) > m) { ax = 2; }
    const float pa[3] = {A[ax], A[3 + ax], A[6 + ax]}, pb[3] = {B[ax], B[3 + ax], B[6 + ax]};
    float a0, a1, b0, b1;
    if (!col_intervals(pa, da, &a0, &a1) || !col_intervals(pb, db, &b0, &b1)) return 0;
    if (a0 > a1) { const float t = a0; a0 = a1; a1 = t; }
    if (b0 > b1) { const float t = b0; b0 = b1; b1 = t; }
    return !(a1 < b0 || b1 < a0);
}

// sum over the three vertices of `intr` of Psi_recv(v)^2; g = d(that sum)/d(common translation of the intruder's vertices)
COL_FN float col_pair_side(const float *recv, const float *intr, float sigma, float *g)
{
    float e1[3], e2[3], n[3], o[3];
    for (int k = 0; k < 3; k++) { e1[k] = recv[3 + k] - recv[k]; e2[k] = recv[6 + k] - recv[k]; }
    col_cross(e1, e2, n);
    const float nn = col_dot(n, n);
    g[0] = g[1] = g[2] = 0.f;
    if (!(nn > 1e-30f)) return 0.f;
    // circumcentre o = a + ((|e2|^2 (e1 x e2)) x e1 + |e1|^2 (e2 x (e1 x e2))) / (2 |e1 x e2|^2), circumradius r = |o - a|
    float c1[3], c2[3];
    col_cross(n, e1, c1); col_cross(e2, n, c2);
    const float l1 = col_dot(e1, e1), l2 = col_dot(e2, e2);
    float rel[3];
    for (int k = 0; k < 3; k++) { rel[k] = (l2 * c1[k] + l1 * c2[k]) / (2.f * nn); o[k] = recv[k] + rel[k]; }
    const float r = sqrtf(col_dot(rel, rel)), inv = 1.0f / sqrtf(nn);
    for (int k = 0; k < 3; k++) n[k] *= inv;
    float total = 0.f;
    for (int i = 0; i < 3; i++) {
        float d[3];
        for (int k = 0; k < 3; k++) d[k] = intr[3 * i + k] - o[k];
        const float x = col_dot(n, d);
        float er[3];
        for (int k = 0; k < 3; k++) er[k] = d[k] - x * n[k];
        const float rho = sqrtf(col_dot(er, er)), Dn = r - (r / sigma) * x;
        if (!(Dn > 0.f)) continue;                       // above the apex of the cone (x >= sigma): Upsilon = 0 there anyway
        const float Phi = rho / Dn;
        if (!(Phi < 1.f)) continue;
        float U, dU;
        if (x <= -sigma) { U = -x + 1.f - sigma; dU = -1.f; }
        else if (x < sigma) { const float a2 = -(1.f - 2.f * sigma) / (4.f * sigma * sigma); U = a2 * x * x - x / (2.f * sigma) + (3.f - 2.f * sigma) / 4.f; dU = 2.f * a2 * x - 1.f / (2.f * sigma); }
        else continue;
        const float u = (1.f - Phi) * U, Psi = u * u;
        total += Psi * Psi;
        // d Psi^2 / dv = 4 u^3 du/dv,  du/dv = -U dPhi/dv + (1 - Phi) dU n,  dPhi/dv = e_r / Dn + rho (r / sigma) n / Dn^2
        const float k4 = 4.f * u * u * u, ir = rho > 0.f ? 1.0f / rho : 0.f;
        for (int k = 0; k < 3; k++) {
            const float dPhi = er[k] * ir / Dn + rho * (r / sigma) * n[k] / (Dn * Dn);
            g[k] += k4 * (-U * dPhi + (1.f - Phi) * dU * n[k]);
        }
    }
    return total;
}
