/*
 * bb_oracle.c -- CPU restatement ("oracle") of the BitBIRCH similarity / insertion hot
 * path of mqcomplab/bblean.  TEST INFRASTRUCTURE ONLY -- see bb_oracle.h.
 *
 * Plain C11, scalar 64-bit popcount loops (the shape of similarity.cpp:304-333), one
 * thread.  Written from the behaviour described in SURVEY.md section 8 / Appendix B and
 * the reference lines cited at each function; no reference source is copied.
 *
 * Parity: PINNED against tests/golden/ (generated from the reference, see
 * tests/golden/make_golden.py) by tests/test_oracle_golden.py.
 */
#include "bb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------ */
/* small helpers                                                                        */
/* ------------------------------------------------------------------------------------ */

static inline uint32_t pc64(uint64_t x) { return (uint32_t)__builtin_popcountll(x); }

static inline uint64_t load_u64(const uint8_t* p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;
}

/* popcount of one packed row (similarity.cpp:63-94: u64 fast path or byte loop, same
 * value either way) */
static uint32_t popcount_row(const uint8_t* row, int64_t nbytes) {
    uint32_t c = 0;
    int64_t j = 0;
    for (; j + 8 <= nbytes; j += 8) c += pc64(load_u64(row + j));
    for (; j < nbytes; ++j) c += (uint32_t)__builtin_popcount(row[j]);
    return c;
}

static uint32_t and_popcount_row(const uint8_t* a, const uint8_t* b, int64_t nbytes) {
    uint32_t c = 0;
    int64_t j = 0;
    for (; j + 8 <= nbytes; j += 8) c += pc64(load_u64(a + j) & load_u64(b + j));
    for (; j < nbytes; ++j) c += (uint32_t)__builtin_popcount(a[j] & b[j]);
    return c;
}

/* similarity.cpp:326-331: denominator in uint32, clamp in double, one f64 division */
static inline double jt_from_counts(uint32_t inter, uint32_t card_a, uint32_t card_b) {
    uint32_t denom = card_a + card_b - inter;
    double d = (double)denom;
    if (d < 1.0) d = 1.0;
    return (double)inter / d;
}

/* ------------------------------------------------------------------------------------ */
/* stateless kernels                                                                    */
/* ------------------------------------------------------------------------------------ */

void bbo_popcount_rows(const uint8_t* arr, int64_t n, int64_t nbytes, uint32_t* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = popcount_row(arr + i * nbytes, nbytes);
}

void bbo_jt_arr_vec(const uint8_t* arr, int64_t n, int64_t nbytes, const uint8_t* vec,
                    const uint32_t* card, double* out_sim, uint32_t* out_inter,
                    uint32_t* out_union) {
    uint32_t vec_pc = popcount_row(vec, nbytes);
    for (int64_t i = 0; i < n; ++i) {
        const uint8_t* row = arr + i * nbytes;
        uint32_t c = card ? card[i] : popcount_row(row, nbytes);
        uint32_t inter = and_popcount_row(row, vec, nbytes);
        if (out_sim) out_sim[i] = jt_from_counts(inter, c, vec_pc);
        if (out_inter) out_inter[i] = inter;
        if (out_union) out_union[i] = c + vec_pc - inter;
    }
}

void bbo_unpack(const uint8_t* packed, int64_t n, int64_t nbytes, int64_t n_features,
                uint8_t* out) {
    for (int64_t i = 0; i < n; ++i) {
        const uint8_t* row = packed + i * nbytes;
        uint8_t* o = out + i * n_features;
        for (int64_t j = 0; j < n_features; ++j)
            o[j] = (uint8_t)((row[j >> 3] >> (7 - (j & 7))) & 1u); /* MSB first */
    }
}

void bbo_pack(const uint8_t* unpacked, int64_t n, int64_t n_features, uint8_t* out) {
    int64_t nbytes = (n_features + 7) / 8;
    for (int64_t i = 0; i < n; ++i) {
        const uint8_t* u = unpacked + i * n_features;
        uint8_t* o = out + i * nbytes;
        memset(o, 0, (size_t)nbytes);
        for (int64_t j = 0; j < n_features; ++j)
            if (u[j]) o[j >> 3] |= (uint8_t)(0x80u >> (j & 7));
    }
}

void bbo_centroid_from_sum(const uint64_t* ls, int64_t n_features, int64_t n_samples,
                           int pack, uint8_t* out) {
    /* _py_similarity.py:36-39: n<=1 -> plain cast to uint8; else ls >= n*0.5 */
    int64_t nbytes = (n_features + 7) / 8;
    if (pack) memset(out, 0, (size_t)nbytes);
    double thr = (double)n_samples * 0.5;
    for (int64_t j = 0; j < n_features; ++j) {
        uint8_t bit;
        if (n_samples <= 1)
            bit = (uint8_t)ls[j];
        else
            bit = ((double)ls[j] >= thr) ? 1 : 0;
        if (!pack)
            out[j] = bit;
        else if (bit)
            /* np.packbits treats any non-zero as 1 */
            out[j >> 3] |= (uint8_t)(0x80u >> (j & 7));
    }
}

/* similarity.cpp:297-300, exactly this operation order, IEEE f64 */
static inline double isim_from_moments(uint64_t s1, uint64_t s2, uint64_t n) {
    if (s1 == 0) return 1.0;
    double a = (double)(s2 - s1) / 2.0;
    return a / ((a + (double)(n * s1)) - (double)s2);
}

double bbo_isim_from_sum(const uint64_t* ls, int64_t n_features, int64_t n_objects) {
    if (n_objects < 2) return NAN; /* similarity.cpp:275-279 (RuntimeWarning is host-side) */
    uint64_t s1 = 0, s2 = 0;
    for (int64_t j = 0; j < n_features; ++j) {
        s1 += ls[j];
        s2 += ls[j] * ls[j];
    }
    return isim_from_moments(s1, s2, (uint64_t)n_objects);
}

void bbo_add_rows(const uint8_t* arr, int64_t n, int64_t n_features, uint64_t* out) {
    memset(out, 0, (size_t)n_features * sizeof(uint64_t));
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < n_features; ++j) out[j] += arr[i * n_features + j];
}

static int64_t first_argmin(const double* v, int64_t n) {
    int64_t b = 0;
    for (int64_t i = 1; i < n; ++i)
        if (v[i] < v[b]) b = i;
    return b;
}

void bbo_most_dissimilar(const uint8_t* Y, int64_t n, int64_t nbytes, int64_t n_features,
                         int64_t* idx1, int64_t* idx2, double* sims1, double* sims2) {
    /* similarity.cpp:413-471 / _py_similarity.py:138-178 */
    uint64_t* ls = (uint64_t*)calloc((size_t)n_features, sizeof(uint64_t));
    for (int64_t i = 0; i < n; ++i) {
        const uint8_t* row = Y + i * nbytes;
        for (int64_t j = 0; j < n_features; ++j) ls[j] += (row[j >> 3] >> (7 - (j & 7))) & 1u;
    }
    uint8_t* cen = (uint8_t*)calloc((size_t)nbytes, 1);
    bbo_centroid_from_sum(ls, n_features, n, 1, cen);
    uint32_t* card = (uint32_t*)malloc((size_t)n * sizeof(uint32_t));
    bbo_popcount_rows(Y, n, nbytes, card);
    double* sc = (double*)malloc((size_t)n * sizeof(double));
    bbo_jt_arr_vec(Y, n, nbytes, cen, card, sc, NULL, NULL);
    int64_t f1 = first_argmin(sc, n);
    bbo_jt_arr_vec(Y, n, nbytes, Y + f1 * nbytes, card, sims1, NULL, NULL);
    int64_t f2 = first_argmin(sims1, n);
    bbo_jt_arr_vec(Y, n, nbytes, Y + f2 * nbytes, card, sims2, NULL, NULL);
    *idx1 = f1;
    *idx2 = f2;
    free(ls);
    free(cen);
    free(card);
    free(sc);
}

double bbo_isim_radius_compl_from_sum(const uint64_t* ls, int64_t n_features, int64_t n) {
    /* similarity.py:192-202 */
    uint64_t s1 = 0, s2 = 0, t1 = 0, t2 = 0;
    double thr = (double)n * 0.5;
    for (int64_t j = 0; j < n_features; ++j) {
        uint64_t v = ls[j];
        uint64_t c = (n <= 1) ? (uint64_t)(uint8_t)v : (((double)v >= thr) ? 1u : 0u);
        s1 += v;
        s2 += v * v;
        t1 += v + c;
        t2 += (v + c) * (v + c);
    }
    double jt = (n < 2) ? NAN : isim_from_moments(s1, s2, (uint64_t)n);
    double jt1 = isim_from_moments(t1, t2, (uint64_t)(n + 1));
    return (jt1 * (double)(n + 1) - jt * (double)(n - 1)) / 2;
}

static inline double tol_lookup(const double* tol_table, int64_t tol_len, int64_t old_n) {
    if (tol_table == NULL || old_n < 0 || old_n >= tol_len) return 0.0;
    return tol_table[old_n];
}

int bbo_merge_accept(int crit, double thr, double tolerance, const double* tol_table,
                     int64_t tol_len, const uint64_t* new_ls, int64_t new_n,
                     const uint64_t* old_ls, int64_t old_n, int64_t nom_n,
                     int64_t n_features) {
    switch (crit) {
        case BBO_CRIT_DIAMETER:
            return bbo_isim_from_sum(new_ls, n_features, new_n) >= thr;
        case BBO_CRIT_RADIUS:
            return bbo_isim_radius_compl_from_sum(new_ls, n_features, new_n) >= thr;
        case BBO_CRIT_TOL_DIAMETER: {
            double new_dc = bbo_isim_from_sum(new_ls, n_features, new_n);
            if (new_dc < thr) return 0;
            if (old_n == 1) return 1;
            double old_dc = bbo_isim_from_sum(old_ls, n_features, old_n);
            return new_dc >= old_dc - tol_lookup(tol_table, tol_len, old_n);
        }
        case BBO_CRIT_TOL_RADIUS: {
            double new_rc = bbo_isim_radius_compl_from_sum(new_ls, n_features, new_n);
            if (new_rc < thr) return 0;
            if (old_n == 1) return 1;
            double old_rc = bbo_isim_radius_compl_from_sum(old_ls, n_features, old_n);
            return new_rc >= old_rc - tol_lookup(tol_table, tol_len, old_n);
        }
        case BBO_CRIT_TOL_LEGACY: {
            double new_dc = bbo_isim_from_sum(new_ls, n_features, new_n);
            if (new_dc < thr) return 0;
            if (old_n == 1 || nom_n != 1) return 1;
            double old_dc = bbo_isim_from_sum(old_ls, n_features, old_n);
            return (new_dc * (double)new_n - old_dc * (double)(old_n - 1)) / 2 >=
                   old_dc - tolerance;
        }
        case BBO_CRIT_NEVER:
        default:
            return 0;
    }
}

/* ------------------------------------------------------------------------------------ */
/* tree engine (bblean/bitbirch.py)                                                     */
/* ------------------------------------------------------------------------------------ */

typedef struct Node Node;

/* _BFSubcluster, bitbirch.py:360-526.  The reference keeps linear_sum at the minimum
 * uint width for n_samples (utils.py:25); the arithmetic is exact integer arithmetic at
 * any width, so the oracle keeps uint8 while n <= 255 and uint32 above (n < 2^32). */
typedef struct Sub {
    uint64_t n;
    uint8_t* ls8;
    uint32_t* ls32;
    uint8_t* cent; /* packed centroid, nbytes */
    Node* child;
    uint32_t id;
} Sub;

/* _BFNode, bitbirch.py:214-357 */
struct Node {
    Sub** subs;
    int len;
    uint8_t* cents;  /* (bf+1) x nbytes, rows [0,len) valid (bitbirch.py:264-266) */
    uint32_t* cards; /* popcount of each centroid row (the reference recomputes it in
                        every call, similarity.cpp:374-377; same values) */
    Node* prev_leaf; /* NULL = the dummy leaf is the predecessor */
    Node* next_leaf;
    int is_leaf;
};

struct bbo_tree {
    int bf;
    double thr;
    int crit;
    double tolerance;
    double* tol_table;
    int64_t tol_len;
    int F;
    int nbytes;
    Node* root;
    Node* first_leaf; /* dummy_leaf._next_leaf, bitbirch.py:880-884 */
    uint32_t next_id;
    uint64_t num_fitted;
    /* scratch */
    uint32_t* new_ls; /* F */
    uint8_t* tmp_cent;
    double* s1;
    double* s2;
    double* sc;
    /* result of the last insert */
    uint32_t last_leaf_id;
    int last_merged;
    uint64_t stats[7];
};

static Sub* sub_new(const bbo_tree* t) {
    Sub* s = (Sub*)calloc(1, sizeof(Sub));
    s->ls8 = (uint8_t*)calloc((size_t)t->F, 1);
    s->cent = (uint8_t*)calloc((size_t)t->nbytes, 1);
    s->id = 0xFFFFFFFFu;
    return s;
}

static void sub_free(Sub* s) {
    if (!s) return;
    free(s->ls8);
    free(s->ls32);
    free(s->cent);
    free(s);
}

static inline uint32_t sub_ls(const Sub* s, int j) { return s->ls8 ? s->ls8[j] : s->ls32[j]; }

static void sub_widen(const bbo_tree* t, Sub* s) {
    if (s->ls32) return;
    s->ls32 = (uint32_t*)malloc((size_t)t->F * sizeof(uint32_t));
    for (int j = 0; j < t->F; ++j) s->ls32[j] = s->ls8[j];
    free(s->ls8);
    s->ls8 = NULL;
}

/* centroid_from_sum(pack=True) on a Sub (bitbirch.py:484, :497-499) */
static void sub_recompute_centroid(const bbo_tree* t, Sub* s) {
    memset(s->cent, 0, (size_t)t->nbytes);
    if (s->n <= 1) {
        for (int j = 0; j < t->F; ++j)
            if ((uint8_t)sub_ls(s, j)) s->cent[j >> 3] |= (uint8_t)(0x80u >> (j & 7));
    } else {
        uint64_t n = s->n;
        for (int j = 0; j < t->F; ++j)
            if (2ull * sub_ls(s, j) >= n) s->cent[j >> 3] |= (uint8_t)(0x80u >> (j & 7));
    }
}

/* add_to_n_samples_and_linear_sum without the centroid (bitbirch.py:488-496) */
static void sub_add(const bbo_tree* t, Sub* dst, const Sub* src) {
    uint64_t new_n = dst->n + src->n;
    if (new_n > 255) sub_widen(t, dst);
    if (dst->ls8) {
        for (int j = 0; j < t->F; ++j) dst->ls8[j] = (uint8_t)(dst->ls8[j] + sub_ls(src, j));
    } else if (src->ls8) {
        for (int j = 0; j < t->F; ++j) dst->ls32[j] += src->ls8[j];
    } else {
        for (int j = 0; j < t->F; ++j) dst->ls32[j] += src->ls32[j];
    }
    dst->n = new_n;
}

static Node* node_new(const bbo_tree* t) {
    Node* nd = (Node*)calloc(1, sizeof(Node));
    nd->subs = (Sub**)calloc((size_t)t->bf + 1, sizeof(Sub*));
    nd->cents = (uint8_t*)malloc((size_t)(t->bf + 1) * (size_t)t->nbytes);
    nd->cards = (uint32_t*)calloc((size_t)t->bf + 1, sizeof(uint32_t));
    return nd;
}

static void node_set_row(const bbo_tree* t, Node* nd, int row, Sub* s) {
    nd->subs[row] = s;
    memcpy(nd->cents + (size_t)row * t->nbytes, s->cent, (size_t)t->nbytes);
    nd->cards[row] = popcount_row(s->cent, t->nbytes);
}

/* _BFNode.append_subcluster, bitbirch.py:284-287 */
static void node_append(const bbo_tree* t, Node* nd, Sub* s) {
    node_set_row(t, nd, nd->len, s);
    nd->len++;
}

static void free_subtree(Node* nd) {
    if (!nd) return;
    for (int i = 0; i < nd->len; ++i) {
        if (nd->subs[i]->child) free_subtree(nd->subs[i]->child);
        sub_free(nd->subs[i]);
    }
    free(nd->subs);
    free(nd->cents);
    free(nd->cards);
    free(nd);
}

/* moments of a linear sum held at either width */
static void moments_u32(const uint32_t* ls, int F, uint64_t* s1, uint64_t* s2) {
    uint64_t a = 0, b = 0;
    for (int j = 0; j < F; ++j) {
        uint64_t v = ls[j];
        a += v;
        b += v * v;
    }
    *s1 = a;
    *s2 = b;
}

static void moments_sub(const bbo_tree* t, const Sub* s, uint64_t* s1, uint64_t* s2) {
    uint64_t a = 0, b = 0;
    for (int j = 0; j < t->F; ++j) {
        uint64_t v = sub_ls(s, j);
        a += v;
        b += v * v;
    }
    *s1 = a;
    *s2 = b;
}

/* radius complement on a u32 linear sum (similarity.py:192-202) */
static double radius_compl_u32(const uint32_t* ls, int F, uint64_t n) {
    uint64_t s1 = 0, s2 = 0, t1 = 0, t2 = 0;
    for (int j = 0; j < F; ++j) {
        uint64_t v = ls[j];
        uint64_t c = (n <= 1) ? (uint64_t)(uint8_t)v : ((2 * v >= n) ? 1u : 0u);
        s1 += v;
        s2 += v * v;
        t1 += v + c;
        t2 += (v + c) * (v + c);
    }
    double jt = (n < 2) ? NAN : isim_from_moments(s1, s2, n);
    double jt1 = isim_from_moments(t1, t2, n + 1);
    return (jt1 * (double)(n + 1) - jt * (double)(n - 1)) / 2;
}

static double radius_compl_sub(const bbo_tree* t, const Sub* s) {
    uint32_t* tmp = (uint32_t*)malloc((size_t)t->F * sizeof(uint32_t));
    for (int j = 0; j < t->F; ++j) tmp[j] = sub_ls(s, j);
    double r = radius_compl_u32(tmp, t->F, s->n);
    free(tmp);
    return r;
}

/* merge_accept_fn(threshold, new_ls, new_n, old_ls, nom_ls, old_n, nom_n), _merges.py */
static int tree_accept(const bbo_tree* t, const uint32_t* new_ls, uint64_t new_n,
                       const Sub* old, uint64_t nom_n) {
    uint64_t s1, s2;
    switch (t->crit) {
        case BBO_CRIT_DIAMETER:
            moments_u32(new_ls, t->F, &s1, &s2);
            return isim_from_moments(s1, s2, new_n) >= t->thr;
        case BBO_CRIT_RADIUS:
            return radius_compl_u32(new_ls, t->F, new_n) >= t->thr;
        case BBO_CRIT_TOL_DIAMETER: {
            moments_u32(new_ls, t->F, &s1, &s2);
            double new_dc = isim_from_moments(s1, s2, new_n);
            if (new_dc < t->thr) return 0;
            if (old->n == 1) return 1;
            moments_sub(t, old, &s1, &s2);
            double old_dc = isim_from_moments(s1, s2, old->n);
            return new_dc >= old_dc - tol_lookup(t->tol_table, t->tol_len, (int64_t)old->n);
        }
        case BBO_CRIT_TOL_RADIUS: {
            double new_rc = radius_compl_u32(new_ls, t->F, new_n);
            if (new_rc < t->thr) return 0;
            if (old->n == 1) return 1;
            double old_rc = radius_compl_sub(t, old);
            return new_rc >= old_rc - tol_lookup(t->tol_table, t->tol_len, (int64_t)old->n);
        }
        case BBO_CRIT_TOL_LEGACY: {
            moments_u32(new_ls, t->F, &s1, &s2);
            double new_dc = isim_from_moments(s1, s2, new_n);
            if (new_dc < t->thr) return 0;
            if (old->n == 1 || nom_n != 1) return 1;
            moments_sub(t, old, &s1, &s2);
            double old_dc = isim_from_moments(s1, s2, old->n);
            return (new_dc * (double)new_n - old_dc * (double)(old->n - 1)) / 2 >=
                   old_dc - t->tolerance;
        }
        default:
            return 0;
    }
}

/* _BFSubcluster.merge_subcluster, bitbirch.py:507-526 */
static int sub_try_merge(bbo_tree* t, Sub* T, const Sub* S) {
    uint64_t new_n = T->n + S->n;
    uint32_t* new_ls = t->new_ls;
    for (int j = 0; j < t->F; ++j) new_ls[j] = sub_ls(T, j) + sub_ls(S, j);
    if (!tree_accept(t, new_ls, new_n, T, S->n)) return 0;
    if (new_n > 255) sub_widen(t, T);
    if (T->ls8)
        for (int j = 0; j < t->F; ++j) T->ls8[j] = (uint8_t)new_ls[j];
    else
        memcpy(T->ls32, new_ls, (size_t)t->F * sizeof(uint32_t));
    T->n = new_n;
    sub_recompute_centroid(t, T);
    return 1;
}

/* _split_node, bitbirch.py:162-211 */
static void split_node(bbo_tree* t, Node* node, Sub** outA, Sub** outB) {
    Sub* A = sub_new(t);
    Sub* B = sub_new(t);
    Node* node1 = node_new(t);
    A->child = node1;
    B->child = node;
    if (node->is_leaf) { /* bitbirch.py:182-188: node1 goes immediately before node */
        node1->is_leaf = 1;
        node1->prev_leaf = node->prev_leaf;
        if (node->prev_leaf)
            node->prev_leaf->next_leaf = node1;
        else
            t->first_leaf = node1;
        node1->next_leaf = node;
        node->prev_leaf = node1;
    }
    int m = node->len;
    int64_t f1, f2;
    bbo_most_dissimilar(node->cents, m, t->nbytes, t->F, &f1, &f2, t->s1, t->s2);
    Sub** old = (Sub**)malloc((size_t)m * sizeof(Sub*));
    memcpy(old, node->subs, (size_t)m * sizeof(Sub*));
    node->len = 0;
    for (int i = 0; i < m; ++i) {
        int to1 = (t->s1[i] > t->s2[i]) || (i == (int)f1); /* bitbirch.py:193-200 */
        if (to1) {
            node_append(t, node1, old[i]);
            sub_add(t, A, old[i]);
        } else {
            node_append(t, node, old[i]);
            sub_add(t, B, old[i]);
        }
    }
    /* the reference recomputes the tracking centroid after every update(); only the
     * final value is observable */
    sub_recompute_centroid(t, A);
    sub_recompute_centroid(t, B);
    free(old);
    t->stats[4]++;
    t->stats[5]++;
    *outA = A;
    *outB = B;
}

/* _BFNode.insert_bf_subcluster, bitbirch.py:305-357 */
static int node_insert(bbo_tree* t, Node* node, Sub* S, int depth) {
    if ((uint64_t)depth > t->stats[6]) t->stats[6] = (uint64_t)depth;
    if (node->len == 0) {
        S->id = t->next_id++;
        node_append(t, node, S);
        t->last_leaf_id = S->id;
        t->last_merged = 0;
        t->stats[3]++;
        return 0;
    }
    uint32_t cs = popcount_row(S->cent, t->nbytes);
    int best = 0;
    double best_sim = -1.0;
    for (int i = 0; i < node->len; ++i) {
        uint32_t inter = and_popcount_row(node->cents + (size_t)i * t->nbytes, S->cent, t->nbytes);
        double sim = jt_from_counts(inter, node->cards[i], cs);
        if (sim > best_sim) { /* np.argmax: first maximum, bitbirch.py:320 */
            best_sim = sim;
            best = i;
        }
    }
    t->stats[0]++;
    t->stats[1] += (uint64_t)node->len;
    Sub* T = node->subs[best];
    if (T->child == NULL) {
        if (sub_try_merge(t, T, S)) {
            node_set_row(t, node, best, T);
            t->last_leaf_id = T->id;
            t->last_merged = 1;
            t->stats[2]++;
            return 0;
        }
        S->id = t->next_id++;
        node_append(t, node, S);
        t->last_leaf_id = S->id;
        t->last_merged = 0;
        t->stats[3]++;
        return node->len > t->bf;
    }
    if (node_insert(t, T->child, S, depth + 1)) {
        Sub *A, *B;
        split_node(t, T->child, &A, &B);
        /* update_split_subclusters, bitbirch.py:289-303 */
        node_set_row(t, node, best, A);
        node_append(t, node, B);
        sub_free(T);
        return node->len > t->bf;
    }
    /* tracking subcluster: CF += inserted CF, new centroid (bitbirch.py:352-357) */
    sub_add(t, T, S);
    sub_recompute_centroid(t, T);
    node_set_row(t, node, best, T);
    return 0;
}

static void tree_init_root(bbo_tree* t) { /* _initialize_tree, bitbirch.py:880-884 */
    t->root = node_new(t);
    t->root->is_leaf = 1;
    t->first_leaf = t->root;
    t->stats[5]++;
}

static void tree_insert(bbo_tree* t, Sub* S) {
    if (!t->root) tree_init_root(t);
    if (node_insert(t, t->root, S, 1)) { /* bitbirch.py:778-782 */
        Sub *A, *B;
        split_node(t, t->root, &A, &B);
        Node* nr = node_new(t);
        t->stats[5]++;
        node_append(t, nr, A);
        node_append(t, nr, B);
        t->root = nr;
    }
    if (t->last_merged) sub_free(S);
}

static void set_tol_table(bbo_tree* t, const double* tol_table, int64_t tol_len) {
    free(t->tol_table);
    t->tol_table = NULL;
    t->tol_len = 0;
    if (tol_table && tol_len > 0) {
        t->tol_table = (double*)malloc((size_t)tol_len * sizeof(double));
        memcpy(t->tol_table, tol_table, (size_t)tol_len * sizeof(double));
        t->tol_len = tol_len;
    }
}

static void alloc_scratch(bbo_tree* t) {
    free(t->s1);
    free(t->s2);
    free(t->sc);
    t->s1 = (double*)malloc((size_t)(t->bf + 2) * sizeof(double));
    t->s2 = (double*)malloc((size_t)(t->bf + 2) * sizeof(double));
    t->sc = (double*)malloc((size_t)(t->bf + 2) * sizeof(double));
}

bbo_tree* bbo_tree_create(int32_t branching_factor, double threshold, int32_t criterion,
                          double tolerance, const double* tol_table, int64_t tol_len,
                          int32_t n_features) {
    if (branching_factor < 2 || n_features < 8 || (n_features % 8) != 0) return NULL;
    bbo_tree* t = (bbo_tree*)calloc(1, sizeof(bbo_tree));
    t->bf = branching_factor;
    t->thr = threshold;
    t->crit = criterion;
    t->tolerance = tolerance;
    t->F = n_features;
    t->nbytes = n_features / 8;
    set_tol_table(t, tol_table, tol_len);
    t->new_ls = (uint32_t*)malloc((size_t)t->F * sizeof(uint32_t));
    t->tmp_cent = (uint8_t*)malloc((size_t)t->nbytes);
    alloc_scratch(t);
    return t;
}

void bbo_tree_reset(bbo_tree* t) {
    free_subtree(t->root);
    t->root = NULL;
    t->first_leaf = NULL;
    t->next_id = 0;
    t->num_fitted = 0;
}

void bbo_tree_destroy(bbo_tree* t) {
    if (!t) return;
    bbo_tree_reset(t);
    free(t->tol_table);
    free(t->new_ls);
    free(t->tmp_cent);
    free(t->s1);
    free(t->s2);
    free(t->sc);
    free(t);
}

void bbo_tree_set_merge(bbo_tree* t, int32_t criterion, double tolerance,
                        const double* tol_table, int64_t tol_len, double threshold,
                        int32_t branching_factor) {
    t->crit = criterion;
    t->tolerance = tolerance;
    set_tol_table(t, tol_table, tol_len);
    t->thr = threshold;
    if (branching_factor != t->bf) {
        /* the reference only lets a new branching factor take effect on nodes created
         * afterwards (bitbirch.py:702-703); the host mirrors its call pattern, which
         * always resets before changing it.  The oracle requires an empty tree. */
        if (t->root == NULL) {
            t->bf = branching_factor;
            alloc_scratch(t);
        }
    }
}

int bbo_tree_fit_packed(bbo_tree* t, const uint8_t* rows, int64_t n, uint32_t* out_leaf) {
    for (int64_t e = 0; e < n; ++e) {
        const uint8_t* row = rows + e * t->nbytes;
        Sub* S = sub_new(t); /* bitbirch.py:423-435 */
        S->n = 1;
        for (int j = 0; j < t->F; ++j) S->ls8[j] = (uint8_t)((row[j >> 3] >> (7 - (j & 7))) & 1u);
        memcpy(S->cent, row, (size_t)t->nbytes);
        tree_insert(t, S);
        if (out_leaf) out_leaf[e] = t->last_leaf_id;
        t->num_fitted++;
    }
    return 0;
}

static inline uint64_t load_elem(const void* base, int width, int64_t idx) {
    switch (width) {
        case 1:
            return ((const uint8_t*)base)[idx];
        case 2:
            return ((const uint16_t*)base)[idx];
        case 4:
            return ((const uint32_t*)base)[idx];
        default:
            return ((const uint64_t*)base)[idx];
    }
}

int bbo_tree_fit_buffers(bbo_tree* t, const void* bufs, int32_t width, int64_t k,
                         uint32_t* out_leaf) {
    if (width != 1 && width != 2 && width != 4 && width != 8) return 1;
    int64_t cols = (int64_t)t->F + 1;
    for (int64_t e = 0; e < k; ++e) {
        Sub* S = sub_new(t); /* bitbirch.py:412-421 */
        uint64_t n = load_elem(bufs, width, e * cols + t->F);
        S->n = n;
        if (n > 255) sub_widen(t, S);
        for (int j = 0; j < t->F; ++j) {
            uint64_t v = load_elem(bufs, width, e * cols + j);
            if (S->ls8)
                S->ls8[j] = (uint8_t)v;
            else
                S->ls32[j] = (uint32_t)v;
        }
        sub_recompute_centroid(t, S);
        tree_insert(t, S);
        if (out_leaf) out_leaf[e] = t->last_leaf_id;
        t->num_fitted += n;
    }
    return 0;
}

int64_t bbo_tree_leaf_count(const bbo_tree* t) {
    int64_t c = 0;
    for (const Node* nd = t->first_leaf; nd; nd = nd->next_leaf) c += nd->len;
    return c;
}

void bbo_tree_export_leaves(const bbo_tree* t, uint32_t* leaf_ids, uint64_t* n_samples,
                            uint8_t* packed_centroids, uint32_t* linear_sums) {
    int64_t k = 0;
    for (const Node* nd = t->first_leaf; nd; nd = nd->next_leaf) {
        for (int i = 0; i < nd->len; ++i, ++k) {
            const Sub* s = nd->subs[i];
            if (leaf_ids) leaf_ids[k] = s->id;
            if (n_samples) n_samples[k] = s->n;
            if (packed_centroids)
                memcpy(packed_centroids + (size_t)k * t->nbytes, s->cent, (size_t)t->nbytes);
            if (linear_sums)
                for (int j = 0; j < t->F; ++j) linear_sums[(size_t)k * t->F + j] = sub_ls(s, j);
        }
    }
}

/* BitFeature buffer rows [linear sum (n_features values), n_samples] of the leaves at `positions` (indices into the
 * leaf-chain order of bbo_tree_export_leaves), `width` bytes per value: what _bf_to_np writes (reference
 * bitbirch.py:1224-1290), without exporting every leaf at 4 bytes per feature first. */
int bbo_tree_gather_buffers(const bbo_tree* t, const int64_t* positions, int64_t m, int32_t width, void* out) {
    int64_t k = 0;
    for (const Node* nd = t->first_leaf; nd; nd = nd->next_leaf) k += nd->len;
    const Sub** subs = (const Sub**)malloc((size_t)(k > 0 ? k : 1) * sizeof(*subs));
    if (!subs) return 1;
    k = 0;
    for (const Node* nd = t->first_leaf; nd; nd = nd->next_leaf)
        for (int i = 0; i < nd->len; ++i) subs[k++] = nd->subs[i];
    const size_t cols = (size_t)t->F + 1;
    int rc = 0;
    for (int64_t r = 0; r < m && rc == 0; ++r) {
        const int64_t p = positions[r];
        if (p < 0 || p >= k) { rc = 2; break; }
        const Sub* s = subs[p];
        for (size_t j = 0; j < cols; ++j) {
            const uint64_t v = j < (size_t)t->F ? (uint64_t)sub_ls(s, (int)j) : (uint64_t)s->n;
            const size_t at = (size_t)r * cols + j;
            switch (width) {
                case 1: ((uint8_t*)out)[at] = (uint8_t)v; break;
                case 2: ((uint16_t*)out)[at] = (uint16_t)v; break;
                case 4: ((uint32_t*)out)[at] = (uint32_t)v; break;
                case 8: ((uint64_t*)out)[at] = v; break;
                default: rc = 3; break;
            }
        }
    }
    free(subs);
    return rc;
}

void bbo_tree_stats(const bbo_tree* t, uint64_t* out7) { memcpy(out7, t->stats, sizeof(t->stats)); }
