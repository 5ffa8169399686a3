/* r3dgs_reduction.h -- C ABI of the operators either side of the rasterizer hot path:
 * the redundancy-score, codebook (k-means) and nearest-neighbour operators `train.py`'s cull / quantise steps call.
 * Same conventions as r3dgs_rasterizer.h: device pointers, fp32 / int32, contiguous; `void* stream` is a
 * hipStream_t; return >= 0 on success, < 0 with the message in r3dgs_last_error().  No call synchronises the host.
 *
 * Reference (/root/reference/submodules/...) interface each entry point replaces:
 *   r3dgs_min_pixel_size                 Reduced3DGS::calculatePixelSize      diff-gaussian-rasterization/reduced_3dgs.cu:240-264
 *   r3dgs_sphere_ellipsoid_intersection  Reduced3DGS::intersectionTest        reduced_3dgs.cu:205-238
 *   r3dgs_min_redundancy                 Reduced3DGS::assignFinalRedundancyValue  reduced_3dgs.cu:268-287
 *   r3dgs_kmeans                         Reduced3DGS::kmeans                  reduced_3dgs.cu:290-340
 *   r3dgs_knn                            SimpleKNN::knn / knn_index2          simple-knn/simple_knn.h:17-21, simple_knn.cu:179, :468
 *   r3dgs_knn_query                      SimpleKNN::knn_indexQ                simple_knn.cu:523-660
 */
#ifndef R3DGS_REDUCTION_H
#define R3DGS_REDUCTION_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Smallest world-space footprint of one pixel over all cameras whose frustum contains the centre
 * (redundancy_score.cu:46-97).  ONE launch loops over the cameras on the device (the reference launches once per
 * camera and reads each image size back to the host).  w2ndc / w2ndc_inv: [n_cameras,16] in the reference's
 * transposed layout (m[4c+r]); image_height / image_width: int32[n_cameras] on the device.
 * pixel_sizes[P] is written (10000 where no camera sees the centre). */
int r3dgs_min_pixel_size(int P, int n_cameras, const float* w2ndc, const float* w2ndc_inv, const float* means3D,
                         const int* image_height, const int* image_width, float* pixel_sizes, void* stream);

/* For every (Gaussian, candidate neighbour) pair: does the sphere of radius sphere_radius[i] around Gaussian i
 * intersect the neighbour's ellipsoid (redundancy_score.cu:121-159; the rotation used is Gaussian i's own, as in
 * the reference).  neighbours: int32[P,knn]; redundancy[P] (count of intersecting neighbours) and
 * mask[P,knn] (one byte per pair, 0/1) are written. */
int r3dgs_sphere_ellipsoid_intersection(int P, int knn, const float* means3D, const float* scales,
                                        const float* rotations, const int* neighbours, const float* sphere_radius,
                                        int* redundancy, uint8_t* mask, void* stream);

/* min_redundancy[n] = min(P, min over pairs (i, n) with mask set of redundancy[i]) (redundancy_score.cu:6-27). */
int r3dgs_min_redundancy(int P, int knn, const int* redundancy, const int* neighbours, const uint8_t* mask,
                         int* min_redundancy, void* stream);

/* 1-D k-means (Lloyd) with n_centers <= 1024 centres (the reference supports exactly 256): repeat
 * {assign each value to the first nearest centre; centre = mean of its values, 0 if empty} until the summed
 * centre shift < tol or max_iterations, then a final assignment (reduced_3dgs.cu:305-338).  The values are sorted
 * once; an update is one pass over the sorted values (binary search per value, prefix-sum differences per
 * cluster).  The convergence test stays on the device: all updates are enqueued and become no-ops once converged.
 * workspace: r3dgs_kmeans_workspace_bytes(n_values, n_centers) bytes of device scratch (0 is returned, with
 * r3dgs_last_error set, when no GPU is present to size the sort).  ids[n_values], centers_out[n_centers] are
 * written; iterations_run (device int, may be NULL) receives the number of updates.  Values must not be NaN. */
size_t r3dgs_kmeans_workspace_bytes(int n_values, int n_centers);
int r3dgs_kmeans(int n_values, int n_centers, const float* values, const float* centers_in, float tol,
                 int max_iterations, int* ids, float* centers_out, int* iterations_run, char* workspace,
                 void* stream);

/* Exact K nearest neighbours of every point among the other points (squared Euclidean distance, the point itself
 * excluded by index).  dists / indices: [P,K], ascending by (distance, index) -- the reference leaves the slots in
 * the order its box traversal filled them (simple_knn.cu:393-466); unfilled slots (P-1 < K) hold FLT_MAX / -1.
 * mean_dist3 (may be NULL): [P], (d0 + d1 + d2) / 3 of the three nearest = distCUDA2 (simple_knn.cu:143-191);
 * with mean_dist3 set, dists / indices may be NULL and K is ignored.  1 <= K <= r3dgs_knn_max_k().
 * workspace: r3dgs_knn_workspace_bytes(P) bytes of device scratch (0 is returned, with r3dgs_last_error set,
 * when no GPU is present to size the sort). */
int r3dgs_knn_max_k(void);
size_t r3dgs_knn_workspace_bytes(int P);
int r3dgs_knn(int P, int K, const float* points, float* dists, int* indices, float* mean_dist3, char* workspace,
              void* stream);

/* SimpleKNN::knn_indexQ (simple-knn/simple_knn.h:22, simple_knn.cu:523-660; distIndexQ of spatial.cu:43-58): for each of
 * the Q query points points[q_indices[q]], the K nearest (squared Euclidean distance) among the points whose index occurs
 * in n_indices[0..N) -- a set: duplicates count once --, the query point itself excluded by index.  dists / indices:
 * [Q,K], ascending by (distance, index); unfilled slots hold FLT_MAX / -1 (the reference leaves its slots in traversal
 * order and the unfilled ones at FLT_MAX / -1).  A query or candidate index outside [0, P) is ignored (its row stays
 * unfilled).  1 <= K <= 4096.  Exact, O(Q * N): the operator has no caller in the reference tree.
 * workspace: r3dgs_knn_query_workspace_bytes(P) bytes of device scratch. */
size_t r3dgs_knn_query_workspace_bytes(int P);
int r3dgs_knn_query(int P, int K, const float* points, int Q, const int* q_indices, int N, const int* n_indices, float* dists,
                    int* indices, char* workspace, void* stream);

/* Per-view densification statistics of a view-parallel training step (no reference equivalent: the reference is
 * single-GPU; this is what train.py:134 and scene/gaussian_model.py:693-695 accumulate per view), fused into one launch:
 *   grad_norm[i] = radii[i] > 0 ? ||viewspace_grad[i, 0:2]|| : 0      visible[i] = radii[i] > 0 ? 1 : 0
 *   radii_out[i] = radii[i]
 * viewspace_grad: [P,3] (gradient of the means2D dummy).  Outputs may live in a flat exchange buffer. */
int r3dgs_pack_view_stats(int P, const float* viewspace_grad, const int* radii, float* grad_norm, float* visible,
                          int* radii_out, void* stream);

/* Local reduction of the view-parallel exchange: `recv` holds `world` copies (one per rank, in rank order) of this
 * rank's shard of the flat exchange buffer, `shard` 4-byte elements each, the shard starting at element `shard_begin`
 * of the buffer.  Elements of the buffer below `sum_len` are fp32 and are SUMmed, the rest are int32 and MAXed
 * (radii).  out: [shard].  The combination order is the rank order, so all replicas compute identical bits. */
int r3dgs_reduce_shards(int world, long long shard, long long shard_begin, long long sum_len, const float* recv,
                        float* out, void* stream);
/* The same for a buffer with a reduced-precision middle region (round 6, the opt-in bfloat16 transport of the higher-band
 * SH gradients): elements [sum_len, half_end) of the buffer are 4-byte words holding TWO bfloat16 each; every half is
 * widened, summed in fp32 in rank order and rounded once (to nearest even) back to bfloat16.  Elements below sum_len are
 * fp32 SUM, elements from half_end on int32 MAX, as above.  half_end == sum_len makes it r3dgs_reduce_shards. */
int r3dgs_reduce_shards_mixed(int world, long long shard, long long shard_begin, long long sum_len, long long half_end,
                              const float* recv, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* R3DGS_REDUCTION_H */
