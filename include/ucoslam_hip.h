/*
 * ucoslam_hip.h — C ABI of the MI355X (gfx950) tracking hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.
 * Every entry point names the reference interface (file:line in lambdaloop/ucoslam-cv3)
 * whose work it replaces.  The C++ adaptors in include/ucoslam_hip/ mirror the
 * reference class surfaces on top of these calls; INTEGRATION.md shows the
 * reference-side binding.
 *
 * Conventions
 *   - every function returns 0 on success, a negative UH_E* code on failure;
 *     uh_last_error() returns a thread-local message for the last failure.
 *   - "_dev" entry points take DEVICE pointers, enqueue on the context stream
 *     and do not synchronise; the others take HOST pointers, copy in/out and
 *     return when results are in the caller's buffers.
 *   - there is NO CPU fallback: without a usable HIP device every call fails
 *     with UH_ENODEVICE.
 */
#ifndef UCOSLAM_HIP_H
#define UCOSLAM_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UH_OK          0
#define UH_EINVAL     -1   /* bad argument (the reference throws std::runtime_error here) */
#define UH_ENODEVICE  -2   /* no HIP device / HIP runtime failure */
#define UH_ENOTBUILT  -3   /* index/vocabulary not built (reference: search() returns false) */
#define UH_ENOMEM     -4
#define UH_ECAPACITY  -5   /* caller's output buffer too small */

typedef struct uh_ctx uh_ctx;

/* Context = one GPU + one HIP stream (+ scratch owned by the stage objects).
 * stream == NULL creates a private non-blocking stream; otherwise the given
 * hipStream_t is used (e.g. torch.cuda.current_stream().cuda_stream). */
int         uh_ctx_create(int device, void* hip_stream, uh_ctx** out);
void        uh_ctx_destroy(uh_ctx* ctx);
int         uh_ctx_synchronize(uh_ctx* ctx);
void*       uh_ctx_stream(uh_ctx* ctx);
const char* uh_last_error(void);
/* library/ABI version: major*10000 + minor*100 + patch */
int         uh_version(void);

/* ------------------------------------------------------------------------
 * Hamming brute-force kNN  — replaces xflann::Index (Linear) build/search:
 *   3rdparty/xflann/xflann/index.cpp:45 (build), :77-103 (search + sort),
 *   impl/linear.h:68-86 (_knnsearch<Hamming_x64_32bytes>), impl/resultset.h:64-135
 *   (max-heap ResultSet), index.h:119-134 (exchange sort).
 * Results are bit-identical to the reference INCLUDING the heap array order of
 * unsorted rows and the order of equal distances in sorted rows.
 * Unfilled slots (nt < nn): index -1, distance 0   (linear.h:82-85).
 * ------------------------------------------------------------------------ */
typedef struct uh_knn uh_knn;

int  uh_knn_create(uh_ctx* ctx, uh_knn** out);
void uh_knn_destroy(uh_knn* idx);

/* build: train = nt rows of desc_bytes (must be 32 = ORB) bytes, row stride in bytes.
 * Host pointer; the rows are copied to HBM (reference LinearParams(store_data=1)). */
int uh_knn_build(uh_knn* idx, const uint8_t* train, int nt, size_t stride, int desc_bytes);
/* build from rows already resident in HBM (contiguous nt x 32, 16-byte aligned); not copied. */
int uh_knn_build_dev(uh_knn* idx, const uint8_t* d_train, int nt);
/* restrict the scan to train rows [begin,end) (multi-GPU sharding of the train set);
 * indices stay global.  Default = [0,nt). */
int uh_knn_set_shard(uh_knn* idx, int begin, int end);
int uh_knn_size(const uh_knn* idx);

/* search: queries nq x 32 (row stride q_stride bytes), outputs nq x nn int32 row-major.
 * sorted: 0/1 as KnnSearchParams(maxChecks,sorted); max_dist: -1 = kNN, >=0 = radius bound
 * (SearchParams maxDist, resultset.h:66). */
int uh_knn_search(uh_knn* idx, const uint8_t* queries, int nq, size_t q_stride, int nn,
                  int32_t* indices, int32_t* distances, int sorted, int max_dist);
int uh_knn_search_dev(uh_knn* idx, const uint8_t* d_queries, int nq, int nn,
                      int32_t* d_indices, int32_t* d_distances, int sorted, int max_dist);

/* Sharded search (multi-GPU): phase 1 emits, per query, the candidates the local
 * heap ACCEPTED while scanning this shard, in train-index order (a superset of what
 * the global heap accepts).  cand = nq x cap entries, each (dist<<32 | global index);
 * counts = nq int32 (count > cap means overflow: that row must be rescanned).
 * Phase 2 replays the concatenation of all shards' candidate lists (in shard order)
 * through the exact ResultSet semantics.  lists: nshards blocks of [nq x cap],
 * counts: nshards blocks of [nq]. */
int uh_knn_scan_shard_dev(uh_knn* idx, const uint8_t* d_queries, int nq, int nn, int max_dist,
                          uint64_t* d_cand, int32_t* d_counts, int cap);
int uh_knn_replay_dev(uh_knn* idx, const uint8_t* d_queries, int nq, int nn, int sorted, int max_dist,
                      const uint64_t* d_cand_all, const int32_t* d_counts_all, int nshards, int cap,
                      int32_t* d_indices, int32_t* d_distances);

#ifdef __cplusplus
}
#endif
#endif /* UCOSLAM_HIP_H */
