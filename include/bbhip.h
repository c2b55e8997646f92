/*
 * bbhip.h -- C ABI of libbbhip.so, the MI355X (gfx950 / CDNA4) BitBIRCH similarity and
 * insertion engine.  This is the drop-in boundary for the hot path of mqcomplab/bblean:
 * every entry point replaces one binding of the reference's only native module,
 * `bblean._cpp_similarity` (bblean/csrc/similarity.cpp:473-521), or one piece of the
 * per-fingerprint Python loop that calls it (bblean/bitbirch.py:769-787, :848-866).
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch/pybind types.
 *   - every data pointer may be a HOST pointer or a DEVICE (hipMalloc / torch) pointer;
 *     the library detects which (hipPointerGetAttributes).  Host inputs are staged to
 *     HBM, host outputs are copied back before the call returns.  With device pointers
 *     the call only enqueues work on `stream` (NULL = the default stream) unless stated.
 *   - return value: 0 on success, non-zero error code otherwise; bbh_last_error() gives
 *     the message for the calling thread.  (The reference throws std::runtime_error,
 *     similarity.cpp:64-66, :345-351; the Python shim maps codes back to RuntimeError /
 *     ValueError / RuntimeWarning.)
 *   - the library never frees or keeps caller memory; `bbh_tree` handles are created and
 *     destroyed explicitly and are not thread-safe.
 *   - packed fingerprints are uint8 rows, most significant bit first (np.packbits,
 *     bblean/fingerprints.py:46), C-contiguous with the given row stride.
 */
#ifndef BBHIP_H
#define BBHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BBH_OK 0
#define BBH_ERR_INVALID 1     /* bad argument (reference: RuntimeError / ValueError)   */
#define BBH_ERR_HIP 2         /* a HIP runtime call failed                             */
#define BBH_ERR_NO_DEVICE 3   /* no gfx950 device visible                              */
#define BBH_ERR_CAPACITY 4    /* device pools exhausted and could not grow             */
#define BBH_ERR_STATE 5       /* call not valid in the current tree state              */

/* merge criteria, bblean/_merges.py:194-212 */
#define BBH_CRIT_DIAMETER 0
#define BBH_CRIT_RADIUS 1
#define BBH_CRIT_TOL_DIAMETER 2
#define BBH_CRIT_TOL_RADIUS 3
#define BBH_CRIT_TOL_LEGACY 4
#define BBH_CRIT_NEVER 5

const char* bbh_last_error(void);
/* library / device info: writes "gfx950 <name> CUs=.. HBM=.." ; returns device count */
int bbh_device_count(void);
int bbh_device_info(int device, char* buf, size_t buflen);
/* Device blocks released by trees and calls are kept for reuse (hipFree synchronises the device);
 * this returns them to the driver, e.g. before handing the GPU to another allocator.  No
 * reference counterpart: NumPy's allocator plays this role there. */
int bbh_trim_cache(void);

/* A function the library calls when a device allocation fails, before it tries once more: the host side hands over what
 * IT caches on the device (the Python shim registers torch.cuda.empty_cache - a caching tensor allocator keeps freed round
 * tables of gigabytes that the driver then cannot give to a growing tree pool).  NULL removes it. */
int bbh_set_memory_pressure_callback(void (*fn)(void));

/* ---------------------------------------------------------------------------------- */
/* Stateless kernels -- one per pybind11 binding of similarity.cpp:473-521             */
/* ---------------------------------------------------------------------------------- */

/* _popcount_2d (similarity.cpp:99-141) and _popcount_1d (:63-94, n = 1).
 * arr: n rows x nbytes; out: n uint32. */
int bbh_popcount_rows(const uint8_t* arr, int64_t n, int64_t nbytes, int64_t row_stride,
                      uint32_t* out, void* stream);

/* _jt_sim_arr_vec_packed (similarity.cpp:374-377) = jt_sim_packed_precalc_cardinalities
 * (:340-372) + _calc_arr_vec_jt (:304-333):
 *   out_sim[i] = inter_i / max(double(card_i + popcount(vec) - inter_i), 1.0)
 * card: optional precomputed row popcounts (NULL -> computed in the same pass).
 * out_inter / out_union: optional exact integer numerators / denominators (uint32). */
int bbh_jt_arr_vec(const uint8_t* arr, int64_t n, int64_t nbytes, int64_t row_stride,
                   const uint8_t* vec, const uint32_t* card, double* out_sim,
                   uint32_t* out_inter, uint32_t* out_union, void* stream);

/* Batched best match: for each of nq query rows, the FIRST index of the maximum
 * Tanimoto against nc centroid rows (np.argmax at bitbirch.py:320 fused with the
 * similarity call at :317).  out_idx: nq int32; out_inter/out_union: optional nq uint32
 * of the winning pair; out_sims: optional nq x nc float64 full matrix
 * (jt_sim_matrix_packed, similarity.py:239-247, when queries == centroids). */
int bbh_jt_best_match(const uint8_t* queries, int64_t nq, const uint8_t* cents, int64_t nc,
                      int64_t nbytes, int32_t* out_idx, uint32_t* out_inter,
                      uint32_t* out_union, double* out_sims, void* stream);

/* unpack_fingerprints (similarity.cpp:145-214): packed n x nbytes -> n x n_features
 * uint8 of 0/1.  n_features must be a multiple of 8 (same restriction, :163-165). */
int bbh_unpack(const uint8_t* packed, int64_t n, int64_t nbytes, int64_t n_features,
               uint8_t* out, void* stream);

/* pack_fingerprints (fingerprints.py:46-49 = np.packbits(axis=-1), MSB first): n x n_features uint8
 * of 0/non-0 -> n x ceil(n_features/8) packed uint8, the last byte zero-padded. */
int bbh_pack(const uint8_t* unpacked, int64_t n, int64_t n_features, uint8_t* out, void* stream);

/* add_rows (similarity.cpp:381-400): column sums of an n x n_features uint8 array,
 * or, with packed != 0, of the unpacked view of an n x nbytes packed array
 * (jt_isim_packed_u8's inner step, :407-411).  out: n_features uint64. */
int bbh_add_rows(const uint8_t* arr, int64_t n, int64_t n_cols, int packed,
                 int64_t n_features, uint64_t* out, void* stream);

/* centroid_from_sum<uint64_t> (similarity.cpp:216-271; NumPy twin used by the tree,
 * _py_similarity.py:12-42): n_samples <= 1 -> cast; else bit = (2*ls >= n_samples).
 * ls_width: element size of linear_sum in bytes (1, 2, 4 or 8).
 * out: n_features bytes (pack == 0) or ceil(n_features/8) bytes, MSB first. */
int bbh_centroid_from_sum(const void* linear_sum, int32_t ls_width, int64_t n_features,
                          int64_t n_samples, int pack, uint8_t* out, void* stream);

/* jt_isim_from_sum (similarity.cpp:273-301).  *out = NaN and *warn = 1 when
 * n_objects < 2 (the reference raises RuntimeWarning, :275-279).  out is a HOST double. */
int bbh_isim_from_sum(const void* linear_sum, int32_t ls_width, int64_t n_features,
                      int64_t n_objects, double* out, int* warn, void* stream);

/* jt_isim_unpacked_u8 / jt_isim_packed_u8 (similarity.cpp:402-411): add_rows then
 * jt_isim_from_sum with n_objects = n. */
int bbh_isim_rows(const uint8_t* arr, int64_t n, int64_t n_cols, int packed,
                  int64_t n_features, double* out, int* warn, void* stream);

/* The pair loop of metrics.jt_isim_dunn (bblean/metrics.py:186-199) in one call: min over all pairs i < j of
 * 1 - jt_isim_from_sum(sums[i] + sums[j], sizes[i] + sizes[j]); 1.0 when there are fewer than two clusters.
 * sums: k x n_features uint64 column sums (host or device), sizes: k uint64; out: host double. */
int bbh_isim_pair_min_gap(const uint64_t* sums, const uint64_t* sizes, int64_t k, int64_t n_features,
                          double* out, void* stream);

/* jt_most_dissimilar_packed (similarity.cpp:413-471).  idx1/idx2 are HOST int64;
 * sims1/sims2: n float64 (host or device). */
int bbh_most_dissimilar(const uint8_t* Y, int64_t n, int64_t nbytes, int64_t n_features,
                        int64_t* idx1, int64_t* idx2, double* sims1, double* sims2,
                        void* stream);

/* ---------------------------------------------------------------------------------- */
/* Stateful tree engine -- the per-fingerprint loop of BitBirch.fit / _fit_buffers      */
/* (bitbirch.py:769-787, :848-866) with everything it calls: _BFNode.insert_bf_subcluster*/
/* (:305-357), _BFSubcluster.merge_subcluster/update (:488-526), the merge criteria      */
/* (_merges.py), centroid_from_sum (_py_similarity.py:12) and _split_node (:162-211).    */
/* The whole tree (cluster features, centroids, node tables, leaf chain) is resident in  */
/* HBM; one call inserts a whole batch with the reference's sequential semantics.        */
/* ---------------------------------------------------------------------------------- */

typedef struct bbh_tree bbh_tree;

/* BitBirch.__init__ (bitbirch.py:596-643).  tol_table[old_n] is the adaptive tolerance
 * of _merges.py:113 evaluated by the caller (entries >= tol_len are 0); may be NULL. */
int bbh_tree_create(bbh_tree** out, int32_t branching_factor, double threshold,
                    int32_t criterion, double tolerance, const double* tol_table,
                    int64_t tol_len, int32_t n_features, int32_t device);
int bbh_tree_destroy(bbh_tree* t);

/* BitBirch.set_merge (bitbirch.py:674-703).  A branching factor change requires an
 * empty (reset) tree, which is how the reference's callers use it. */
int bbh_tree_set_merge(bbh_tree* t, int32_t criterion, double tolerance,
                       const double* tol_table, int64_t tol_len, double threshold,
                       int32_t branching_factor);

/* BitBirch.reset (bitbirch.py:1078-1090): drops every node and BitFeature. */
int bbh_tree_reset(bbh_tree* t);

/* BitBirch.fit hot loop (bitbirch.py:769-787): insert n packed fingerprints in order.
 * out_leaf[e] = id of the leaf BitFeature that element e was merged into or created
 * (ids are stable until reset); host or device, may be NULL.  Synchronous. */
int bbh_tree_fit_packed(bbh_tree* t, const uint8_t* rows, int64_t n, int64_t row_stride,
                        uint32_t* out_leaf, void* stream);

/* The same for several independent trees (the shards of multiround's first round,
 * multiround.py:401-422) in ONE kernel launch, one workgroup per tree: the trees insert
 * concurrently on different compute units.  All trees must live on one device. */
int bbh_trees_fit_packed(bbh_tree** trees, int32_t n_trees, const uint8_t* const* rows,
                         const int64_t* n, const int64_t* row_stride, uint32_t* const* out_leaf,
                         void* stream);
/* _fit_buffers for several independent trees in one launch (the trees of a merge round,
 * multiround.py:240-264): tree i inserts k[i] buffers of element width width[i] from bufs[i]. */
int bbh_trees_fit_buffers(bbh_tree** trees, int32_t n_trees, const void* const* bufs,
                          const int32_t* width, const int64_t* k, uint32_t* const* out_leaf,
                          void* stream);

/* BitBirch._fit_buffers hot loop (bitbirch.py:848-866): insert k BitFeature buffers
 * [linear_sum(n_features) | n_samples], elements of `width` bytes (1,2,4,8), row-major
 * with (n_features+1) columns.  Synchronous. */
int bbh_tree_fit_buffers(bbh_tree* t, const void* bufs, int32_t width, int64_t k,
                         uint32_t* out_leaf, void* stream);

/* Number of leaf BitFeatures (len(BitBirch._get_leaf_bfs()), bitbirch.py:1216-1222). */
int bbh_tree_leaf_count(bbh_tree* t, int64_t* out);

/* Leaves in leaf-chain order (bitbirch.py:886-893 flattened).  Every output optional
 * (NULL); host or device.  leaf_ids: k uint32; n_samples: k uint64;
 * packed_centroids: k x ceil(F/8) uint8; linear_sums: k x F elements of ls_width bytes
 * (1,2,4,8 -- values must fit; the Python shim asks per dtype group). */
int bbh_tree_export_leaves(bbh_tree* t, uint32_t* leaf_ids, uint64_t* n_samples,
                           uint8_t* packed_centroids, void* linear_sums, int32_t ls_width);

/* Gather selected leaves (by position in chain order) as BitFeature buffer rows
 * [linear_sum | n_samples] of `width` bytes -- the table BitBirch._bf_to_np /
 * multiround's round files hold (bitbirch.py:1292-1308, multiround.py:132-143).
 * positions: m int64 (host); out: m x (F+1) elements, host or device. */
int bbh_tree_gather_buffers(bbh_tree* t, const int64_t* positions, int64_t m,
                            int32_t width, void* out);

/* Packed centroids (np.packbits order, ceil(F/8) bytes each) of the leaves at `positions`
 * (chain order; host int64): BitBirch.get_centroids on a selection (bitbirch.py:895-907).
 * For a leaf BitFeature of one fingerprint this IS its buffer row in packed form, which is
 * how multiround keeps the singleton tail of a uint8 round table (multiround.py:132-143:
 * 2049 bytes per row there) at 256 bytes per row in HBM.  out: m x ceil(F/8), host or device. */
int bbh_tree_gather_centroids(bbh_tree* t, const int64_t* positions, int64_t m, uint8_t* out);

/* counters for tests / profiling: [0] similarity calls, [1] rows compared, [2] merges,
 * [3] appends, [4] splits, [5] nodes, [6] max depth, [7] BitFeature slots used */
int bbh_tree_stats(bbh_tree* t, uint64_t* out8);

/* Which of the three insertion kernels did the work, since the handle was created (tests, bench.py):
 * [0..2] elements inserted by the pipelined / the steady-state / the complete kernel, [3..5] their launches,
 * [6] launches the pipelined kernel ended because the tree had a shape it does not handle, [7] launches that
 * ended because a pool was exhausted (the host grew it and relaunched). */
int bbh_tree_kernel_counts(bbh_tree* t, uint64_t* out8);

/* The level-systolic kernel (one tree over many workgroups: every node has an owner workgroup, elements flow down
 * the levels through one-way rings; the reference's order per node - bitbirch.py:305-357 - by construction; its
 * elements also count as "pipelined" in bbh_tree_kernel_counts): [0] elements it inserted, [1] its launches,
 * [2] relaunches after a root split, [3] workgroups of the last launch, [4] shader cycles its workgroups spent on
 * elements (not waiting), summed, [5] those of workgroup 0 (the root's owner), [6] of the busiest other
 * workgroup (summed over launches), [7] launches it refused (a shape it does not take).
 * Opt-in: BBHIP_SYS=1 uses it wherever the shape allows, =auto where the other kernels are weakest (informative root);
 * unset or 0 (the default): never - its hand-over between workgroups is not dependable yet, DESIGN.md 6s. */
int bbh_tree_sys_counts(bbh_tree* t, uint64_t* out8);

/* What the tree holds in HBM and what its node storage did (tests, tools/config45.py, bench.py):
 * [0] bytes of the node pools (capacity), [1] bytes of their used part, [2] bytes of the uint8 / uint16 / uint32
 * cluster-feature pools (capacity), [3] largest sum of this tree's allocations so far (a pool that is being regrown
 * counts twice), [4] compactions of the node pools, [5] nodes the last one sealed (length rounded up to a block of rows
 * instead of bf + 1 rows: the reference's node is a list that grows, bitbirch.py:264-287), [6] nodes it left at full
 * capacity, [7] sealed nodes moved back to full capacity because an insertion reached them. */
int bbh_tree_memory(bbh_tree* t, uint64_t* out8);

/* Compact the node pools now (the engine does it on its own when pools beyond BBHIP_GC_MIN_MB, default 1024, have to
 * grow): seal != 0 seals every node whose length is what the previous compaction recorded, seal == 0 brings every node
 * back to full capacity.  Results of later calls do not depend on when or whether this ran. */
int bbh_tree_compact(bbh_tree* t, int32_t seal);

/* Per-kernel timing with HIP events on the stream each kernel is launched on.
 * bbh_profile_enable(1) turns it on; bbh_profile_get returns launches and summed
 * milliseconds for the kernel called `name` ("jt_arr_vec", "tree_insert", ...; the tree kernels are recorded as
 * "tree_insert/pipe", "tree_insert/fast" and "tree_insert/complete", "tree_insert" is their sum). */
int bbh_profile_enable(int on);
int bbh_profile_reset(void);
int bbh_profile_get(const char* name, int64_t* launches, double* total_ms);
/* work units (rows / inserted elements) summed over the recorded launches of `name` */
int bbh_profile_units(const char* name, int64_t* units);
/* the longest single recorded launch of `name`: its milliseconds and its work units (the dominant launch of a fit,
 * next to which the short probe / warm-up launches of the same name would only blur an average) */
int bbh_profile_longest(const char* name, double* ms, int64_t* units);

#ifdef __cplusplus
}
#endif
#endif
