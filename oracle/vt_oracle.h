/* vt_oracle.h -- CPU oracle of the VisTracker fit hot path. TEST INFRASTRUCTURE ONLY (see vt_oracle.c). */
#ifndef VT_ORACLE_H
#define VT_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int V, J, NB, NP;            /* 6890, 52, 10, 459 */
    const float *v_template;     /* (V,3) */
    const float *shapedirs;      /* (V,3,NB) */
    const float *posedirs;       /* (V,3,NP) */
    const float *J_regressor;    /* (J,V) dense */
    const float *weights;        /* (V,J) */
    const int *parents;          /* (J), parents[0] unused */
} vto_smpl_model;

typedef struct { const float *data; int C, H, W; } vto_map;  /* NCHW, all frames: data + b*C*H*W */

/* heads in order df(2) pca(9) parts(14) centers(3) vis(1); w[h][l] is (out,in) row-major, b[h][l] (out) */
typedef struct { const float *w[5][4]; const float *b[5][4]; } vto_decoders;

void vto_rodrigues(const float *aa, int n, float *R);
void vto_rodrigues_bwd(const float *aa, int n, const float *dR, float *daa);

void vto_smplh_forward(const vto_smpl_model *m, const float *pose, const float *betas, const float *trans,
                       int B, float *verts, float *jtr, float *v_posed_out);
void vto_smplh_backward(const vto_smpl_model *m, const float *pose, const float *betas, const float *trans,
                        int B, const float *dverts, const float *djtr,
                        float *dpose, float *dbetas, float *dtrans);

void vto_landmarks_forward(const int *indptr, const int *indices, const float *data, int K,
                           const float *verts, int B, int V, float *out);
void vto_landmarks_backward(const int *indptr, const int *indices, const float *data, int K,
                            const float *dout, int B, int V, float *dverts);

void vto_mahalanobis(const float *x, int B, int stride, int off, int n, const float *mean, const float *prec,
                     float *value, float *dx, float gscale);

/* cam = {fx_px, fy_px, cx_px, cy_px, crop_size} */
void vto_query_forward(const vto_decoders *d, const vto_map *maps, const float *pts, const float *crop_center,
                       const float *body_center, int B, int N, const float *cam, int head_mask,
                       float *df, float *pca, float *parts, float *centers, float *vis);
void vto_query_backward(const vto_decoders *d, const vto_map *maps, const float *pts, const float *crop_center,
                        const float *body_center, int B, int N, const float *cam,
                        const float *d_df, const float *d_pca, const float *d_parts, const float *d_centers,
                        const float *d_vis, float *dpts);

void vto_so3_project(const float *M, int B, float *R);
void vto_so3_project_bwd(const float *M, int B, const float *dR, float *dM);

void vto_rigid_forward(const float *X0, int shared_x0, const float *R, const float *t, const float *s, int B, int N, float *X);
void vto_rigid_backward(const float *X0, int shared_x0, const float *R, const float *t, const float *s, int B, int N,
                        const float *dX, float *dR, float *dt);

double vto_accel_loss(const float *v, int B, int D, const float *elem_w, float gscale, float *dv);
double vto_velocity_loss(const float *v, int B, int D, float gscale, float *dv);

double vto_chamfer_ragged(const float *x, const int *offx, const float *y, const int *offy, int P,
                          float gscale, float *dx, float *dy);

void vto_sil_forward(const float *verts, int B, int NV, const int *faces, int NF, const float *K, int is, float *image);
void vto_sil_face_index(const float *verts, int B, int NV, const int *faces, int NF, const float *K, int is, int *face_index);
void vto_triplane_render(const float *verts, const float *center, int B, int NV, const int *faces, int NF, int is, float *masks);
void vto_sil_backward(const float *verts, int B, int NV, const int *faces, int NF, const float *K, int is,
                      const float *d_image, float eps, float *dverts);

void vto_adam_step(float *p, const float *g, float *m, float *v, int n, int step, float lr, float beta1, float beta2, float eps);
int vto_num_threads(void);

#ifdef __cplusplus
}
#endif
/* A21 interpenetration term (PARITY UNPINNED: mesh_intersection is un-vendored): returns mean_b P_b; dt (B,3) += gscale / B dP_b/dt; npairs (B) */
double vto_collision_loss(const float *sv, int NVs, const int *sf, int NFs, const float *ov, int NVo, const int *of, int NFo, int B, float sigma,
                          int max_coll, float gscale, float *dt, int *npairs);
#endif
