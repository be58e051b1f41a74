/*
 * gof_oracle.h -- C API of liboracle_gof.so (TEST INFRASTRUCTURE, see gof_oracle.cpp).
 * All pointers are HOST pointers.  GofRasterArgs is the product's POD (include/gof_hip.h).
 */
#ifndef GOF_ORACLE_H_INCLUDED
#define GOF_ORACLE_H_INCLUDED
#include <stddef.h>
#include <stdint.h>
#include "../include/gof_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct GofRefState GofRefState;

const char* gofref_last_error(void);
int gofref_num_threads(void);
float gofref_expf(float x);

/* forward (rasterize_points.cu:36-122 + rasterizer_impl.cu:247-405).  out_color [9,H,W],
 * radii [P] (may be NULL).  *state_out keeps every intermediate for gofref_fetch /
 * gofref_backward; free with gofref_free. */
int gofref_forward(const GofRasterArgs* a, float* out_color, int32_t* radii, GofRefState** state_out);
/* backward (rasterize_points.cu:124-211 + rasterizer_impl.cu:409-526) */
int gofref_backward(const GofRasterArgs* a, const GofRefState* s, const float* dL_dout,
                    float* dL_dmeans2D, float* dL_dcolors, float* dL_dopacity, float* dL_dmeans3D,
                    float* dL_dcov3D, float* dL_dsh, float* dL_dscales, float* dL_drotations, float* dL_dview2gaussian);
/* the per-Gaussian backward stage alone (backward.cu:593-631) on given dL_dview2gaussian / dL_dcolors;
 * outputs must be zero-initialised by the caller */
int gofref_preprocess_backward(const GofRasterArgs* a, const GofRefState* s, const float* dL_dview2gaussian,
                               const float* dL_dcolors, float* dL_dmeans3D, float* dL_dsh, float* dL_dscales, float* dL_drotations);
/* integrate (rasterize_points.cu:234-343 + rasterizer_impl.cu:530-792) */
int gofref_integrate(const GofRasterArgs* a, int32_t PN, const float* points3D,
                     float* out_color, float* out_alpha_integrated, float* out_color_integrated,
                     int32_t* radii, GofRefState** state_out);
int gofref_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present);

void gofref_free(GofRefState* s);
uint32_t gofref_num_rendered(const GofRefState* s);
uint32_t gofref_num_integrated(const GofRefState* s);
/* copy a named intermediate; dst may be NULL to query the element count */
int64_t gofref_fetch(const GofRefState* s, const char* name, void* dst, size_t dst_bytes);

/* marching tetrahedra restatement (utils/tetmesh.py:47-138), see gof_oracle_mtets.inc */
int gofref_mtets(int64_t V, int64_t Tt, const int64_t* tets, const float* vertices, const float* sdf, const float* scales,
                 int64_t* num_edges, int64_t* num_faces,
                 int64_t* edge_ids /*cap*/, float* edge_pos, float* edge_sdf, float* edge_scales, int64_t* faces, int64_t cap_edges, int64_t cap_faces);
#ifdef __cplusplus
}
#endif
#endif
