// query_f32.h -- strict-fp32 route of the point query (query_f32.hip), called by the C ABI entry points of query.hip when a handle's
// precision is VT_PRECISION_FP32.  Same argument meaning as the vt_query_* functions of include/vistracker.h.
#pragma once
#include "../../include/vistracker.h"

namespace f32q {
struct Net;
int create(Net **out, const float *const *w, const float *const *bvec, const float *cam, void *stream);
void destroy(Net *h);
int forward(const Net *h, const vt_maps *maps, const float *pts, const float *crop_center, const float *body_center, int B, int N,
            float *df, float *pca, float *parts, float *centers, float *vis, void *stream);
int backward(const Net *h, const vt_maps *maps, const float *pts, const float *crop_center, const float *body_center, int B, int N,
             const float *d_df, const float *d_pca, const float *d_parts, const float *d_centers, const float *d_vis, float *dpts, void *stream);
int human_loss(const Net *h, const vt_maps *maps, const float *pts, const float *crop_center, const float *body_center, int B, int N,
               const int *labels, const int *order, float w_dfh, float w_part, float *dpts, double *terms, void *stream);
int object_loss(const Net *h, const vt_maps *maps, const float *pts, const float *crop_center, const float *body_center, int B, int N,
                const float *occ, float w_obj, float *dpts, double *terms, void *stream);
int project_step(const Net *h, const vt_maps *maps, const float *pts, const float *crop_center, const float *body_center, int B, int N,
                 int df_idx, float threshold, float *pts_out, float *df_target, void *stream);
}  // namespace f32q
