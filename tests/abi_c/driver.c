/*
 * driver.c -- a plain C99 consumer of include/acmi.h, linked against libacmi.so.
 *
 * What the Zend shim (php_ext/php_ahocorasick_amd.c) does with the ABI, without Zend: the six call sites
 * of the MultiFast API in the reference's glue (src/php_ahocorasick.c:140,484,504,745,812,821; the API
 * itself: src/multifast/ahocorasick.h:73-87) become acmi_* calls.  Built by tests/test_abi_c.py with
 *     gcc -std=c99 -pedantic -Wall -Werror -Iinclude tests/abi_c/driver.c -lacmi
 * so the header is proven to be C (engine.cpp only ever consumed it as C++).
 *
 *   driver layout   print sizeof / offsetof of every ABI struct (compared with the ctypes mirror)
 *   driver cpu      lifecycle, status codes, pattern metadata -- no device needed
 *   driver kat      tests/test1.phpt (KAT-1) through acmi_match on the GPU, findAll true and false
 *   driver comm8    BASELINE configs[3] in miniature, no python anywhere: the 65,536-pattern set (8..64 B, Sigma = 256) on EIGHT
 *                   handles of this process, one thread each (acmi_comm_create_local), every rank matching its contiguous shard,
 *                   then the all-gatherv and a gather to one root -- checked against one handle matching the whole batch
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "acmi.h"

/* C99 has no _Static_assert: a negative array size does the job */
#define STATIC_ASSERT(cond, name) typedef char static_assert_##name[(cond) ? 1 : -1]
STATIC_ASSERT(sizeof(acmi_record) == 16, record_size);
STATIC_ASSERT(offsetof(acmi_record, hay) == 0 && offsetof(acmi_record, pos) == 4 &&
              offsetof(acmi_record, start) == 8 && offsetof(acmi_record, pattern) == 12, record_layout);
STATIC_ASSERT(sizeof(acmi_pattern) == 32, pattern_size);
STATIC_ASSERT(offsetof(acmi_pattern, value) == 0 && offsetof(acmi_pattern, length) == 8 &&
              offsetof(acmi_pattern, id_type) == 12 && offsetof(acmi_pattern, id_number) == 16 &&
              offsetof(acmi_pattern, user) == 24, pattern_layout);
STATIC_ASSERT(offsetof(acmi_tables, table) == 40, tables_first_pointer);
STATIC_ASSERT(sizeof(void *) == 8 && sizeof(size_t) == 8, lp64);

#define CHECK(cond)                                                            \
    do {                                                                       \
        if (!(cond)) {                                                         \
            fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #cond); \
            return 1;                                                          \
        }                                                                      \
    } while (0)

/* tests/test1.phpt:11-25: key => string id, id => numeric id, neither => default */
static const struct { const char *value; uint32_t id_type; int64_t id; } kat1[] = {
    {"alfa", ACMI_ID_STRING, 0},  {"beta", ACMI_ID_STRING, 0},  {"gamma", ACMI_ID_STRING, 0}, {"delta", ACMI_ID_STRING, 0},
    {"zeta", ACMI_ID_NUMBER, 0},  {"omega", ACMI_ID_STRING, 0}, {"lfa", ACMI_ID_DEFAULT, 0},
};
#define KAT1_N (sizeof(kat1) / sizeof(kat1[0]))

static int make_handle(acmi_handle **out)
{
    acmi_pattern p[KAT1_N];
    size_t i, accepted = 0;
    CHECK(acmi_create(-1, out) == ACMI_OK);
    for (i = 0; i < KAT1_N; i++) {
        p[i].value = (const uint8_t *)kat1[i].value;
        p[i].length = (uint32_t)strlen(kat1[i].value);
        p[i].id_type = kat1[i].id_type;
        p[i].id_number = kat1[i].id;
        p[i].user = 1000 + i;
    }
    CHECK(acmi_add_patterns(*out, p, KAT1_N, &accepted) == ACMI_OK);
    CHECK(accepted == KAT1_N);
    return 0;
}

static int run_layout(void)
{
    printf("acmi_record %zu\nacmi_pattern %zu\nacmi_tables %zu\n", sizeof(acmi_record), sizeof(acmi_pattern), sizeof(acmi_tables));
#define OFF(f) printf("acmi_tables.%s %zu\n", #f, offsetof(acmi_tables, f))
    OFF(n_states); OFF(max_pattern_length); OFF(table); OFF(lds_image); OFF(class_of_byte); OFF(own_pattern); OFF(dict_link);
    OFF(out_count); OFF(anchor_q); OFF(anchor_keys); OFF(anchor_table); OFF(anchor_mode); OFF(anchor_bucket_bits); OFF(anchor_bucket_shift);
    OFF(anchor_bitmap); OFF(anchor_choice); OFF(anchor_offsets); OFF(n_flat_outputs); OFF(compact); OFF(dense_nf_end); OFF(n_cold_edges); OFF(cold_rec);
    OFF(cold_edges); OFF(chain_lo); OFF(chain_byte); OFF(chain_run); OFF(out_off); OFF(out_list);
#undef OFF
    return 0;
}

static int run_cpu(void)
{
    acmi_handle *h = NULL;
    acmi_pattern info, extra;
    acmi_tables t;
    double v = 0;
    size_t accepted = 99;
    const uint8_t dup[] = "alfa";

    CHECK(strcmp(acmi_version(), ACMI_VERSION) == 0);
    CHECK(acmi_create(-1, NULL) == ACMI_E_ARG);
    CHECK(acmi_is_valid(NULL) == 0);
    CHECK(acmi_finalize(NULL) == ACMI_E_INVALID);
    CHECK(acmi_destroy(NULL) == ACMI_E_INVALID);
    CHECK(strcmp(acmi_strerror(ACMI_E_INVALID), "Invalid resource.") == 0);    /* php_ahocorasick.c:697 */
    CHECK(strcmp(acmi_strerror(ACMI_E_NOT_INIT), "Not initialized.") == 0);     /* :702 */

    if (make_handle(&h)) return 1;
    CHECK(acmi_is_valid(h) == 1);
    CHECK(acmi_pattern_count(h) == KAT1_N);
    CHECK(acmi_pattern_info(h, 4, &info) == 1);
    CHECK(info.length == 4 && memcmp(info.value, "zeta", 4) == 0 && info.id_type == ACMI_ID_NUMBER && info.user == 1004);
    CHECK(acmi_pattern_info(h, KAT1_N, &info) == ACMI_E_ARG);
    /* duplicate, empty and over-long patterns are dropped silently (ahocorasick.c:101-105,123-124) */
    extra.value = dup; extra.length = 4; extra.id_type = ACMI_ID_DEFAULT; extra.id_number = 0; extra.user = 0;
    CHECK(acmi_add_patterns(h, &extra, 1, &accepted) == ACMI_OK && accepted == 0);
    extra.length = 0;
    CHECK(acmi_add_patterns(h, &extra, 1, &accepted) == ACMI_OK && accepted == 0);
    CHECK(acmi_pattern_info(h, KAT1_N, &info) == 0);   /* known to the handle, not accepted */
    CHECK(acmi_get_stat(h, "finalized", &v) == ACMI_OK && v == 0);
    CHECK(acmi_set_replacement(h, 0, (const uint8_t *)"ALFA", 4) == ACMI_OK);           /* AC_PATTERN_t.rtext, actypes.h:88-94 */
    CHECK(acmi_set_replacement(h, 1000, (const uint8_t *)"x", 1) == ACMI_E_ARG);
    CHECK(acmi_finalize(h) == 1);                       /* php_ahocorasick.c:130-142: 1 the first time ... */
    CHECK(acmi_finalize(h) == 0);                       /* ... 0 after */
    CHECK(acmi_get_stat(h, "finalized", &v) == ACMI_OK && v == 1);
    CHECK(acmi_add_patterns(h, &extra, 1, NULL) == ACMI_E_FINALIZED);   /* :914-917 */
    CHECK(acmi_set_replacement(h, 0, NULL, 0) == ACMI_E_FINALIZED);
    CHECK(acmi_get_stat(h, "nodes", &v) == ACMI_OK && v == 1 + 4 + 4 + 5 + 5 + 4 + 5 + 3);  /* no shared prefixes */
    CHECK(acmi_get_stat(h, "no such stat", &v) == ACMI_E_ARG);
    CHECK(acmi_set_option(h, "kernel", 99) == ACMI_E_ARG);
    CHECK(acmi_debug_tables(h, &t) == ACMI_OK && t.n_states == 32 && t.max_pattern_length == 5 && t.entry_bytes == 2);
    CHECK(acmi_destroy(h) == ACMI_OK);
    CHECK(acmi_is_valid(h) == 0);                       /* tests/test2.phpt:295-298: second use of a closed handle */
    CHECK(acmi_destroy(h) == ACMI_E_INVALID);
    CHECK(acmi_finalize(h) == ACMI_E_INVALID);
    printf("cpu ok\n");
    return 0;
}

static int run_kat(void)
{
    /* tests/test1.phpt:60-119: (pos, start_postion, pattern index) in callback order */
    static const uint32_t expect[5][3] = {{14, 9, 2}, {19, 15, 4}, {24, 19, 5}, {28, 24, 0}, {28, 25, 6}};
    const char *hay = "alFABETA gamma zetaomegaalfa!";
    acmi_handle *h = NULL;
    const acmi_record *rec = NULL;
    size_t n = 0, i;
    int rc;
    if (make_handle(&h)) return 1;
    rc = acmi_match(h, (const uint8_t *)hay, strlen(hay), 1, &rec, &n);      /* lazy finalize, :707 */
    if (rc != ACMI_OK) { fprintf(stderr, "acmi_match: %s (%s)\n", acmi_strerror(rc), acmi_last_error(h)); return 1; }
    CHECK(n == 5);
    for (i = 0; i < n; i++) {
        printf("%u %u %u %u\n", rec[i].hay, rec[i].pos, rec[i].start, rec[i].pattern);
        CHECK(rec[i].hay == 0 && rec[i].pos == expect[i][0] && rec[i].start == expect[i][1] && rec[i].pattern == expect[i][2]);
    }
    CHECK(acmi_match(h, (const uint8_t *)hay, strlen(hay), 0, &rec, &n) == ACMI_OK);   /* findAll=false: first event only */
    CHECK(n == 1 && rec[0].pos == 14 && rec[0].pattern == 2);
    CHECK(acmi_match(h, (const uint8_t *)"", 0, 1, &rec, &n) == ACMI_OK && n == 0);
    CHECK(acmi_destroy(h) == ACMI_OK);
    printf("kat ok\n");
    return 0;
}

/* ---- comm8 ------------------------------------------------------------------------------------------------------ */
#define W8 8
#define NPAT 65536u
#define NHAY 64u
#define HAYLEN 4096u

static uint64_t sm_state;
static uint64_t splitmix(void)
{
    uint64_t z;
    sm_state += 0x9E3779B97F4A7C15ull;
    z = sm_state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

typedef struct {
    int rank;
    acmi_handle *h;
    acmi_comm *comm;
    const uint8_t *blob;
    const uint64_t *off;      /* global offsets, NHAY + 1 */
    const uint64_t *ref_prefix;
    const acmi_record *ref_rec;
    size_t ref_n;
    int failed;
} rank_ctx;

static int rank_body(rank_ctx *c)
{
    const size_t lo = (size_t)NHAY * (size_t)c->rank / W8, hi = (size_t)NHAY * (size_t)(c->rank + 1) / W8, nh = hi - lo;
    uint64_t off[NHAY + 1];
    const uint64_t *prefix = NULL;
    const acmi_record *rec = NULL;
    size_t n = 0, i;
    const void *dp = NULL, *dr = NULL;
    uint64_t g_rec = 0, g_hay = 0;
    double sent = 0, received = 0;
    uint64_t *hp;
    acmi_record *hr;
    int mode;
    for (i = 0; i <= nh; i++) off[i] = c->off[lo + i] - c->off[lo];
    CHECK(acmi_match_batch(c->h, c->blob + c->off[lo], off, nh, 1, &prefix, &rec, &n) == ACMI_OK);
    CHECK(n == (size_t)(c->ref_prefix[hi] - c->ref_prefix[lo]));
    hp = (uint64_t *)malloc((NHAY + 1) * sizeof(uint64_t));
    hr = (acmi_record *)malloc((c->ref_n + 1) * sizeof(acmi_record));
    CHECK(hp && hr);
    for (mode = 0; mode < 2; mode++) {   /* 0: all-gatherv; 1: gather to rank 3 */
        const int root = 3, receiver = mode == 0 || c->rank == root;
        int rc = mode == 0 ? acmi_comm_allgather_result(c->comm, c->h, nh, &dp, &dr, &g_rec, &g_hay)
                           : acmi_comm_gather_result(c->comm, c->h, nh, root, &dp, &dr, &g_rec, &g_hay);
        if (rc != ACMI_OK) { fprintf(stderr, "rank %d: exchange: %s (%s)\n", c->rank, acmi_strerror(rc), acmi_comm_last_error(c->comm)); return 1; }
        CHECK(g_rec == c->ref_n && g_hay == NHAY);
        CHECK(acmi_comm_get_stat(c->comm, "bytes_sent", &sent) == ACMI_OK && acmi_comm_get_stat(c->comm, "bytes_received", &received) == ACMI_OK);
        /* 8 bytes per record on the wire (half a record) + 8 per haystack; a gather's root sends nothing, its senders receive nothing */
        CHECK(sent == ((mode == 1 && c->rank == root) ? 0.0 : (double)(8 * n + 8 * nh)));
        CHECK(received == (receiver ? (double)(8 * (c->ref_n - n) + 8 * (NHAY - nh)) : 0.0));
        if (!receiver) { CHECK(dp == NULL && dr == NULL); continue; }
        CHECK(dp != NULL && acmi_comm_result_to_host(c->comm, hp, hr) == ACMI_OK);
        CHECK(memcmp(hp, c->ref_prefix, (NHAY + 1) * sizeof(uint64_t)) == 0);
        CHECK(memcmp(hr, c->ref_rec, c->ref_n * sizeof(acmi_record)) == 0);
    }
    free(hp);
    free(hr);
    return 0;
}

static void *rank_main(void *p)
{
    rank_ctx *c = (rank_ctx *)p;
    c->failed = rank_body(c);
    return NULL;
}

static int run_comm8(void)
{
    static uint8_t pat_bytes[NPAT * 64u];
    static acmi_pattern pats[NPAT];
    static uint8_t blob[NHAY * HAYLEN];
    uint64_t off[NHAY + 1];
    acmi_handle *h[W8], *whole = NULL;
    acmi_comm *comm[W8];
    rank_ctx ctx[W8];
    pthread_t th[W8];
    const uint64_t *prefix = NULL;
    const acmi_record *rec = NULL;
    uint64_t *ref_prefix;
    acmi_record *ref_rec;
    size_t n = 0, i, k, accepted = 0;
    int r, bad = 0;
    sm_state = 7;
    for (i = 0; i < sizeof(pat_bytes); i++) pat_bytes[i] = (uint8_t)splitmix();
    for (k = 0; k < NPAT; k++) {
        pats[k].value = pat_bytes + 64u * k;
        pats[k].length = 8u + (uint32_t)((k * 2654435761ull) >> 7) % 57u;
        pats[k].id_type = ACMI_ID_NUMBER;
        pats[k].id_number = (int64_t)k;
        pats[k].user = 0;
    }
    for (i = 0; i < sizeof(blob); i++) blob[i] = (uint8_t)splitmix();
    for (k = 0; k <= NHAY; k++) off[k] = (uint64_t)k * HAYLEN;
    for (k = 0; k < NHAY; k++) {   /* two planted patterns per haystack */
        const size_t p1 = (1021u * k + 17u) % NPAT, p2 = (4099u * k + 5u) % NPAT;
        memcpy(blob + k * HAYLEN + (37u * k + 11u) % (HAYLEN / 2u - 64u), pats[p1].value, pats[p1].length);
        memcpy(blob + k * HAYLEN + HAYLEN / 2u + (53u * k + 3u) % (HAYLEN / 2u - 64u), pats[p2].value, pats[p2].length);
    }
    /* one handle over the whole batch: what the ranks' shards must add up to */
    CHECK(acmi_create(-1, &whole) == ACMI_OK && acmi_add_patterns(whole, pats, NPAT, &accepted) == ACMI_OK && accepted == NPAT);
    CHECK(acmi_match_batch(whole, blob, off, NHAY, 1, &prefix, &rec, &n) == ACMI_OK);
    CHECK(n >= 2u * NHAY);
    ref_prefix = (uint64_t *)malloc((NHAY + 1) * sizeof(uint64_t));
    ref_rec = (acmi_record *)malloc((n + 1) * sizeof(acmi_record));
    CHECK(ref_prefix && ref_rec);
    memcpy(ref_prefix, prefix, (NHAY + 1) * sizeof(uint64_t));
    memcpy(ref_rec, rec, n * sizeof(acmi_record));
    for (r = 0; r < W8; r++) {
        CHECK(acmi_create(-1, &h[r]) == ACMI_OK && acmi_add_patterns(h[r], pats, NPAT, &accepted) == ACMI_OK);
        CHECK(acmi_set_option(h[r], "small_path", 0) == ACMI_OK);   /* (the shards' results must stay on the device for the exchange) */
        CHECK(acmi_finalize(h[r]) >= 0);
    }
    CHECK(acmi_comm_create_local(h, W8, comm) == ACMI_OK);
    for (r = 0; r < W8; r++) {
        ctx[r].rank = r; ctx[r].h = h[r]; ctx[r].comm = comm[r]; ctx[r].blob = blob; ctx[r].off = off;
        ctx[r].ref_prefix = ref_prefix; ctx[r].ref_rec = ref_rec; ctx[r].ref_n = n; ctx[r].failed = 1;
        CHECK(pthread_create(&th[r], NULL, rank_main, &ctx[r]) == 0);
    }
    for (r = 0; r < W8; r++) { pthread_join(th[r], NULL); bad |= ctx[r].failed; }
    CHECK(!bad);
    for (r = 0; r < W8; r++) { CHECK(acmi_comm_destroy(comm[r]) == ACMI_OK); CHECK(acmi_destroy(h[r]) == ACMI_OK); }
    CHECK(acmi_destroy(whole) == ACMI_OK);
    free(ref_prefix);
    free(ref_rec);
    printf("comm8 ok: %zu records of %u haystacks gathered on %d ranks\n", n, NHAY, W8);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc == 2 && strcmp(argv[1], "layout") == 0) return run_layout();
    if (argc == 2 && strcmp(argv[1], "cpu") == 0) return run_cpu();
    if (argc == 2 && strcmp(argv[1], "kat") == 0) return run_kat();
    if (argc == 2 && strcmp(argv[1], "comm8") == 0) return run_comm8();
    fprintf(stderr, "usage: %s layout|cpu|kat|comm8\n", argv[0]);
    return 2;
}
