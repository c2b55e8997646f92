/*
 * bb_oracle.h -- CPU restatement ("oracle") of the BitBIRCH similarity / insertion hot
 * path of mqcomplab/bblean.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this
 * library.  The product (bblean_amd + libbbhip.so) never links, imports or calls it.
 *
 * Parity status: PINNED.  Every function is checked (tests/test_oracle_golden.py)
 * against fixtures under tests/golden/ that were produced by importing the reference
 * itself (tests/golden/make_golden.py) and against the known-answer values the
 * reference's own tests hold (tests/test_similarity.py, test_refine.py,
 * test_bb_consistency.py, test_multiround.py, test_merges.py of the reference).
 *
 * Each function cites the reference lines it restates, relative to /root/reference/.
 */
#ifndef BB_ORACLE_H
#define BB_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* merge criteria, bblean/_merges.py:194-212 */
enum {
    BBO_CRIT_DIAMETER = 0,      /* _merges.py:56-69   */
    BBO_CRIT_RADIUS = 1,        /* _merges.py:40-53   */
    BBO_CRIT_TOL_DIAMETER = 2,  /* _merges.py:72-116  */
    BBO_CRIT_TOL_RADIUS = 3,    /* _merges.py:119-142 */
    BBO_CRIT_TOL_LEGACY = 4,    /* _merges.py:164-191 */
    BBO_CRIT_NEVER = 5          /* _merges.py:145-161 */
};

/* ---- stateless kernels (bblean/csrc/similarity.cpp) ------------------------------ */

/* _popcount_2d, similarity.cpp:99-141 */
void bbo_popcount_rows(const uint8_t* arr, int64_t n, int64_t nbytes, uint32_t* out);
/* _calc_arr_vec_jt + jt_sim_packed_precalc_cardinalities, similarity.cpp:304-372.
 * card may be NULL (then computed like _jt_sim_arr_vec_packed, :374-377). */
void bbo_jt_arr_vec(const uint8_t* arr, int64_t n, int64_t nbytes, const uint8_t* vec,
                    const uint32_t* card, double* out_sim, uint32_t* out_inter,
                    uint32_t* out_union);
/* unpack_fingerprints, similarity.cpp:145-214 (MSB first; n_features % 8 == 0) */
void bbo_unpack(const uint8_t* packed, int64_t n, int64_t nbytes, int64_t n_features,
                uint8_t* out);
/* np.packbits(axis=-1), fingerprints.py:46-49 */
void bbo_pack(const uint8_t* unpacked, int64_t n, int64_t n_features, uint8_t* out);
/* centroid_from_sum, _py_similarity.py:12-42 and similarity.cpp:216-271.
 * out has n_features bytes if !pack else (n_features+7)/8 bytes. */
void bbo_centroid_from_sum(const uint64_t* ls, int64_t n_features, int64_t n_samples,
                           int pack, uint8_t* out);
/* jt_isim_from_sum, similarity.cpp:273-301; returns NaN when n_objects < 2 */
double bbo_isim_from_sum(const uint64_t* ls, int64_t n_features, int64_t n_objects);
/* add_rows, similarity.cpp:381-400 */
void bbo_add_rows(const uint8_t* arr, int64_t n, int64_t n_features, uint64_t* out);
/* jt_most_dissimilar_packed, similarity.cpp:413-471 */
void bbo_most_dissimilar(const uint8_t* Y, int64_t n, int64_t nbytes, int64_t n_features,
                         int64_t* idx1, int64_t* idx2, double* sims1, double* sims2);
/* jt_isim_radius_compl_from_sum, similarity.py:192-202 */
double bbo_isim_radius_compl_from_sum(const uint64_t* ls, int64_t n_features, int64_t n);
/* one merge decision, _merges.py (whole file). tol_table[old_n] = the reference's
 * max(tolerance*(np.exp(-decay*old_n)-offset),0.0), computed by the caller with numpy
 * (indices >= tol_len mean 0.0). */
int bbo_merge_accept(int crit, double thr, double tolerance, const double* tol_table,
                     int64_t tol_len, const uint64_t* new_ls, int64_t new_n,
                     const uint64_t* old_ls, int64_t old_n, int64_t nom_n,
                     int64_t n_features);

/* ---- stateful tree engine (bblean/bitbirch.py) ------------------------------------- */

typedef struct bbo_tree bbo_tree;

/* BitBirch.__init__, bitbirch.py:596-643 */
bbo_tree* bbo_tree_create(int32_t branching_factor, double threshold, int32_t criterion,
                          double tolerance, const double* tol_table, int64_t tol_len,
                          int32_t n_features);
void bbo_tree_destroy(bbo_tree* t);
/* BitBirch.set_merge, bitbirch.py:674-703 (caller resolves the None-means-keep rules) */
void bbo_tree_set_merge(bbo_tree* t, int32_t criterion, double tolerance,
                        const double* tol_table, int64_t tol_len, double threshold,
                        int32_t branching_factor);
/* BitBirch.reset, bitbirch.py:1078-1090 */
void bbo_tree_reset(bbo_tree* t);
/* BitBirch.fit hot loop, bitbirch.py:769-787.  rows: n x nbytes packed fingerprints.
 * out_leaf[e] = id of the leaf BitFeature element e ended in (merged into or created). */
int bbo_tree_fit_packed(bbo_tree* t, const uint8_t* rows, int64_t n, uint32_t* out_leaf);
/* BitBirch._fit_buffers hot loop, bitbirch.py:848-866.  bufs: k x (n_features+1)
 * elements of `width` bytes (1,2,4,8); last column = n_samples. */
int bbo_tree_fit_buffers(bbo_tree* t, const void* bufs, int32_t width, int64_t k,
                         uint32_t* out_leaf);
/* number of leaf BitFeatures (BitBirch._get_leaf_bfs(sort=False), bitbirch.py:1216) */
int64_t bbo_tree_leaf_count(const bbo_tree* t);
/* leaves in leaf-chain order (bitbirch.py:886-893).  Any output may be NULL.
 * linear_sums: k x n_features uint32. */
void bbo_tree_export_leaves(const bbo_tree* t, uint32_t* leaf_ids, uint64_t* n_samples,
                            uint8_t* packed_centroids, uint32_t* linear_sums);
/* BitFeature buffer rows [linear sum, n_samples] of the leaves at `positions` (indices into that order), `width`
 * (1, 2, 4 or 8) bytes per value; 0 on success. */
int bbo_tree_gather_buffers(const bbo_tree* t, const int64_t* positions, int64_t m, int32_t width, void* out);
/* counters: [0]=similarity calls, [1]=rows compared, [2]=merges, [3]=appends,
 * [4]=splits, [5]=nodes, [6]=max depth seen */
void bbo_tree_stats(const bbo_tree* t, uint64_t* out7);

#ifdef __cplusplus
}
#endif
#endif
