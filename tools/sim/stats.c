/* Workload statistics for the pipelined insertion design (round 3): an instrumented, iterative restatement of
 * the oracle's node_insert.  Not product code; not part of the test suite.
 *   gcc -O2 -o /tmp/simstats tools/sim/stats.c -lm && /tmp/simstats rows.bin N bf thr */
#include "../../oracle/bb_oracle.c"
#include <stdio.h>

#define MAXLV 64
typedef struct { uint64_t last_leaf_elem; } NodeAux;

static uint64_t g_flip_events[MAXLV], g_flip_bits[MAXLV], g_visits[MAXLV], g_zero_levels, g_levels, g_fullhit, g_single_target;
static uint64_t g_reuse_hist[12]; /* leaf reuse distance: 1,2,<=4,<=8,...  */
static uint64_t g_depth_hist[MAXLV];
static uint64_t g_lvl_from_leaf_flip[8], g_lvl_from_leaf_vis[8], g_lvl_from_leaf_zero[8], g_lvl_from_leaf_n[8];
static uint64_t g_row0[8], g_samerow_prev[8];
#define NC 5
static const int csz[NC] = {2, 4, 8, 16, 32};
static Sub* cache[NC][32]; static uint64_t g_chit[NC], g_cacc;
static uint64_t g_same_target_prev; static Sub* prev_target;

/* a tiny map Node* -> last element index that went to this leaf (open addressing) */
#define HB 22
static Node* hk[1 << HB];
static uint64_t hv[1 << HB];
static uint64_t* hget(Node* n) {
    uint64_t h = ((uint64_t)(uintptr_t)n * 0x9E3779B97F4A7C15ull) >> (64 - HB);
    while (hk[h] && hk[h] != n) h = (h + 1) & ((1u << HB) - 1);
    if (!hk[h]) { hk[h] = n; hv[h] = 0; }
    return &hv[h];
}

static int all_zero(const Node* nd) {
    for (int i = 0; i < nd->len; ++i) if (nd->cards[i]) return 0;
    return 1;
}

static Node* prev_path_node[MAXLV]; static int prev_path_row[MAXLV]; static int prev_D = -1;

int main(int argc, char** argv) {
    const char* path = argv[1];
    int64_t N = atoll(argv[2]);
    int bf = atoi(argv[3]);
    double thr = atof(argv[4]);
    FILE* f = fopen(path, "rb");
    uint8_t* rows = malloc((size_t)N * 256);
    if (fread(rows, 256, N, f) != (size_t)N) return 2;
    fclose(f);
    bbo_tree* t = bbo_tree_create(bf, thr, BBO_CRIT_DIAMETER, 0.0, NULL, 0, 2048);
    uint8_t oldc[256];
    uint64_t splits_leaf = 0;
    for (int64_t e = 0; e < N; ++e) {
        /* pre-pass: walk the path read-only for statistics */
        if (t->root && t->root->len) {
            Node* nd = t->root;
            Node* pn[MAXLV]; int pr[MAXLV]; int D = 0;
            const uint8_t* x = rows + e * 256;
            uint32_t cs = popcount_row(x, 256);
            while (1) {
                int best = 0; double bs = -1.0;
                for (int i = 0; i < nd->len; ++i) {
                    uint32_t inter = and_popcount_row(nd->cents + (size_t)i * 256, x, 256);
                    double s = jt_from_counts(inter, nd->cards[i], cs);
                    if (s > bs) { bs = s; best = i; }
                }
                pn[D] = nd; pr[D] = best;
                if (nd->subs[best]->child == NULL) break;
                nd = nd->subs[best]->child; D++;
            }
            /* D = leaf level */
            g_depth_hist[D]++;
            for (int l = 0; l < D; ++l) {
                int fl = D - l; /* levels above leaf: 1 = leaf parent */
                int z = all_zero(pn[l]);
                g_levels++; g_zero_levels += z;
                if (fl < 8) { g_lvl_from_leaf_vis[fl]++; g_lvl_from_leaf_zero[fl] += z; g_lvl_from_leaf_n[fl] += pn[l]->subs[pr[l]]->n;
                    g_row0[fl] += pr[l] == 0;
                    if (prev_D == D && prev_path_node[l] == pn[l] && prev_path_row[l] == pr[l]) g_samerow_prev[fl]++; }
                /* flips the tracking update would cause */
                Sub* T = pn[l]->subs[pr[l]];
                uint64_t n1 = T->n + 1; int flips = 0;
                for (int j = 0; j < 2048; ++j) {
                    uint32_t v = sub_ls(T, j); int xb = (x[j >> 3] >> (7 - (j & 7))) & 1;
                    int ob = (T->cent[j >> 3] >> (7 - (j & 7))) & 1;
                    int nb = 2ull * (v + xb) >= n1;
                    flips += ob != nb;
                }
                if (flips) { g_flip_events[l]++; g_flip_bits[l] += flips; if (fl < 8) g_lvl_from_leaf_flip[fl]++; }
                g_visits[l]++;
            }
            Node* leaf = pn[D];
            uint64_t* last = hget(leaf);
            uint64_t dist = *last ? (uint64_t)e + 1 - *last : 1u << 30;
            int b = 0; while ((1ull << b) < dist && b < 11) b++;
            g_reuse_hist[b]++;
            *last = (uint64_t)e + 1;
            if (leaf->len == bf) g_fullhit++;
            if (leaf->subs[pr[D]]->n == 1) g_single_target++;
            else {
                Sub* T = leaf->subs[pr[D]];
                g_cacc++;
                for (int c = 0; c < NC; ++c) {
                    int hit = -1;
                    for (int q = 0; q < csz[c]; ++q) if (cache[c][q] == T) hit = q;
                    if (hit >= 0) g_chit[c]++;
                    int from = hit >= 0 ? hit : csz[c] - 1;
                    for (int q = from; q > 0; --q) cache[c][q] = cache[c][q - 1];
                    cache[c][0] = T;
                }
            }
            if (leaf->subs[pr[D]] == prev_target) g_same_target_prev++;
            prev_target = leaf->subs[pr[D]];
            for (int l = 0; l <= D; ++l) { prev_path_node[l] = pn[l]; prev_path_row[l] = pr[l]; }
            prev_D = D;
        }
        bbo_tree_fit_packed(t, rows + e * 256, 1, NULL);
    }
    (void)oldc; (void)splits_leaf;
    uint64_t st[7]; bbo_tree_stats(t, st);
    printf("N %lld bf %d thr %.2f: merges %llu appends %llu splits %llu nodes %llu maxdepth %llu\n", (long long)N, bf, thr,
           (unsigned long long)st[2], (unsigned long long)st[3], (unsigned long long)st[4], (unsigned long long)st[5], (unsigned long long)st[6]);
    printf("levels above leaf per element %.3f, of which all-zero %.3f\n", (double)g_levels / N, (double)g_zero_levels / N);
    printf("full-leaf hits %.4f  singleton targets %.4f\n", (double)g_fullhit / N, (double)g_single_target / N);
    printf("non-singleton targets %.4f; LRU CF cache hit rate:", (double)g_cacc / N); for (int c = 0; c < NC; ++c) printf(" [%d] %.3f", csz[c], (double)g_chit[c] / (g_cacc ? g_cacc : 1)); printf("  same target as prev elem %.4f\n", (double)g_same_target_prev / N);
    printf("leaf depth hist:"); for (int i = 0; i < 12; ++i) printf(" %llu", (unsigned long long)g_depth_hist[i]); printf("\n");
    printf("leaf reuse distance hist (<=1,2,4,8,...,1024,more):"); for (int i = 0; i < 12; ++i) printf(" %.4f", (double)g_reuse_hist[i] / N); printf("\n");
    for (int fl = 1; fl < 8; ++fl) if (g_lvl_from_leaf_vis[fl])
        printf("  level leaf-%d: visits %.3f/elem  all-zero %.3f  flip events %.4f per visit  mean n of chosen row %.1f  row0 %.3f  same (node,row) as prev elem %.3f\n", fl,
               (double)g_lvl_from_leaf_vis[fl] / N, (double)g_lvl_from_leaf_zero[fl] / g_lvl_from_leaf_vis[fl],
               (double)g_lvl_from_leaf_flip[fl] / g_lvl_from_leaf_vis[fl], (double)g_lvl_from_leaf_n[fl] / g_lvl_from_leaf_vis[fl],
               (double)g_row0[fl] / g_lvl_from_leaf_vis[fl], (double)g_samerow_prev[fl] / g_lvl_from_leaf_vis[fl]);
    for (int l = 0; l < 10; ++l) if (g_visits[l])
        printf("  level root+%d: visits %llu flip events %.4f bits/event %.2f\n", l, (unsigned long long)g_visits[l],
               (double)g_flip_events[l] / g_visits[l], g_flip_events[l] ? (double)g_flip_bits[l] / g_flip_events[l] : 0.0);
    return 0;
}
