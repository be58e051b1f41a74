/*
 * gof_knn_hip.h -- C ABI of the 3-nearest-neighbour mean squared distance in libgof_hip.so.
 *
 * Replaces the reference's CUDA extension `simple_knn._C.distCUDA2(points) -> (N,) float`
 * (submodules/simple-knn/spatial.cu:15-26 -> SimpleKNN::knn, simple_knn.cu:185-221), called once per training run to
 * initialise the Gaussian scales (scene/gaussian_model.py:327).  For every point: the mean of the squared Euclidean
 * distances to its 3 nearest OTHER points (by index; coincident points count with distance 0), exact -- the reference
 * prunes with Morton-ordered 1024-point boxes, conservatively, so its result is the exact k-NN too.  Points that have fewer
 * than 3 other points keep FLT_MAX terms, as in the reference (simple_knn.cu:154, 183).
 *
 * Conventions as in gof_hip.h: extern "C", device pointers, caller-owned workspace, asynchronous on `stream`, 0 = ok.
 */
#ifndef GOF_KNN_HIP_H_INCLUDED
#define GOF_KNN_HIP_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

size_t gof_knn_ws_bytes(int64_t num_points);
/* points [N,3] fp32, mean_dists [N] fp32 (fully written).  N < 2^31. */
int gof_knn_mean_dist3(int64_t num_points, const float* points, float* mean_dists, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
