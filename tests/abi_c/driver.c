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
 */
#include <stddef.h>
#include <stdio.h>
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

int main(int argc, char **argv)
{
    if (argc == 2 && strcmp(argv[1], "layout") == 0) return run_layout();
    if (argc == 2 && strcmp(argv[1], "cpu") == 0) return run_cpu();
    if (argc == 2 && strcmp(argv[1], "kat") == 0) return run_kat();
    fprintf(stderr, "usage: %s layout|cpu|kat\n", argv[0]);
    return 2;
}
