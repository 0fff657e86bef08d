/*
 * gftorf_knn.h -- C ABI of the 3-nearest-neighbour mean squared distance of libgftorf_rast.so.
 *
 * SURVEY section 8(f) row 3.  Replaces the reference's second native extension,
 * `simple_knn._C.distCUDA2` (submodules/simple-knn/ext.cpp:15-17, spatial.cu:14-25,
 * simple_knn.cu:185-220), which scene/gaussian_model.py:194-199 needs to initialise the
 * Gaussian scales: out[i] = mean of the squared distances from point i to its three nearest
 * other points (self excluded by index; fewer than three other points leave FLT_MAX terms in
 * the mean, as in the reference).
 *
 * Device pointers, no torch types; returns 0 on success (gft_last_error()).
 */
#ifndef GFTORF_KNN_H
#define GFTORF_KNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

size_t gft_knn_scratch_bytes(int32_t P);

/* points [P,3] fp32, mean_dist2 [P] fp32 (written in full), scratch of gft_knn_scratch_bytes(P) */
int gft_knn_mean_dist2(void* hip_stream, int32_t P, const float* points, float* mean_dist2, void* scratch);

#ifdef __cplusplus
}
#endif
#endif
