/*
 * ref_driver.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Thin driver around the REFERENCE's own MultiFast sources, compiled where they
 * lie under /root/reference/src/multifast by oracle/Makefile into
 * oracle/_ref/libmf_ref.so.  No reference source is copied into this repo and
 * no stand-in header is written: the only PHP dependency of the MultiFast core
 * is acmem.h (it includes php.h solely to redirect AC_MALLOC to emalloc); the
 * Makefile pre-defines that header's own include guard
 * (-DPHP_AHO_CORASICK_ACMEM_H) so it expands to nothing and mpool.h's own
 * built-in fallback (AC_MALLOC = malloc, mpool.h:26-32) applies.
 *
 * The driver mimics what php_ahocorasick.c does around the core:
 *   - patterns are inserted in reverse caller order with copy=1, id.type =
 *     NUMBER, id.u.number = caller index, status ignored
 *     (php_ahocorasick.c:457-486);
 *   - search runs with keep=0 (php_ahocorasick.c:745);
 *   - the callback walks j = 0..size-1 and emits
 *     {hay, position, position - ptext.length, id} (php_ahocorasick.c:555-583),
 *     returning 1 when findAll is false (:588).
 * It exports the same entry points as oracle/ac_oracle.c with a ref_ prefix so
 * tests can run both through one ctypes wrapper.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "ahocorasick.h" /* from /root/reference/src/multifast (-I) */

typedef struct {
    int ret_val;
    uint32_t hay;
    uint32_t *out;
    size_t cap, n;
    uint64_t checksum;
} ref_payload;

static int ref_handler(AC_MATCH_t *m, void *param)
{
    ref_payload *p = (ref_payload *)param;
    for (size_t j = 0; j < m->size; j++) {
        uint32_t pos = (uint32_t)m->position;
        uint32_t start = pos - (uint32_t)m->patterns[j].ptext.length;
        uint32_t idx = (uint32_t)m->patterns[j].id.u.number;
        if (p->out && p->n < p->cap) {
            uint32_t *r = p->out + 4 * p->n;
            r[0] = p->hay; r[1] = pos; r[2] = start; r[3] = idx;
        }
        p->checksum += ((uint64_t)p->hay * 0x9E3779B1u) ^ ((uint64_t)pos * 0x85EBCA77u) ^
                       ((uint64_t)start * 0xC2B2AE3Du) ^ (uint64_t)idx;
        p->n++;
    }
    return p->ret_val == 0 ? 0 : 1;
}

/* replacement texts waiting for their pattern's ac_trie_add (the reference takes rtext with the pattern) */
typedef struct ref_rep { void *trie; uint32_t index; uint8_t *bytes; size_t len; struct ref_rep *next; } ref_rep;
static ref_rep *ref_reps = NULL;

void ref_set_replacement(void *t, uint32_t index, const uint8_t *r, size_t rlen)
{
    if (!r)
        return;
    ref_rep *e = (ref_rep *)malloc(sizeof(ref_rep));
    e->trie = t; e->index = index; e->len = rlen;
    e->bytes = (uint8_t *)malloc(rlen + 1);
    memcpy(e->bytes, r, rlen);
    e->bytes[rlen] = 0;
    e->next = ref_reps;
    ref_reps = e;
}

void *ref_create(void) { return ac_trie_create(); }

int ref_add(void *t, const uint8_t *p, size_t len, uint32_t index)
{
    AC_PATTERN_t patt;
    patt.ptext.astring = (const AC_ALPHABET_t *)p;
    patt.ptext.length = len;
    patt.rtext.astring = NULL;
    patt.rtext.length = 0;
    for (ref_rep *e = ref_reps; e; e = e->next)
        if (e->trie == t && e->index == index) {
            patt.rtext.astring = (const AC_ALPHABET_t *)e->bytes;
            patt.rtext.length = e->len;
            break;
        }
    patt.id.type = AC_PATTID_TYPE_NUMBER;
    patt.id.u.number = (long)index;
    patt.aux = NULL;
    return (int)ac_trie_add((AC_TRIE_t *)t, &patt, 1);
}

void ref_finalize(void *t) { ac_trie_finalize((AC_TRIE_t *)t); }
void ref_release(void *t)
{
    for (ref_rep **pp = &ref_reps; *pp;) {   /* drop this trie's pending replacement texts */
        if ((*pp)->trie == t) { ref_rep *e = *pp; *pp = e->next; free(e->bytes); free(e); }
        else pp = &(*pp)->next;
    }
    ac_trie_release((AC_TRIE_t *)t);
}
size_t ref_pattern_count(const void *t) { return ((const AC_TRIE_t *)t)->patterns_count; }

size_t ref_php_add_patterns(void *t, const uint8_t *blob, const uint64_t *off, size_t n,
                            uint32_t base_index)
{
    size_t accepted = 0;
    for (size_t k = n; k-- > 0;)
        if (ref_add(t, blob + off[k], (size_t)(off[k + 1] - off[k]), base_index + (uint32_t)k) ==
            ACERR_SUCCESS)
            accepted++;
    return accepted;
}

size_t ref_php_match(void *t, const uint8_t *text, size_t len, int find_all, uint32_t hay,
                     uint32_t *out, size_t cap)
{
    AC_TRIE_t *trie = (AC_TRIE_t *)t;
    if (trie->trie_open)
        ac_trie_finalize(trie);
    AC_TEXT_t txt;
    txt.astring = (const AC_ALPHABET_t *)text;
    txt.length = len;
    ref_payload p = { find_all ? 0 : 1, hay, out, cap, 0, 0 };
    ac_trie_search(trie, &txt, 0, ref_handler, &p);
    return p.n;
}

/* one chunk of a stream through the reference's own keep=1 mode */
size_t ref_stream_feed(void *t, const uint8_t *text, size_t len, int first, uint32_t *out, size_t cap)
{
    AC_TRIE_t *trie = (AC_TRIE_t *)t;
    if (trie->trie_open)
        ac_trie_finalize(trie);
    AC_TEXT_t txt;
    txt.astring = (const AC_ALPHABET_t *)text;
    txt.length = len;
    ref_payload p = { 0, 0, out, cap, 0, 0 };
    ac_trie_search(trie, &txt, first ? 0 : 1, ref_handler, &p);
    return p.n;
}

size_t ref_php_match_batch(void *t, const uint8_t *blob, const uint64_t *off, size_t nh,
                           int find_all, uint64_t *counts_prefix, uint32_t *out, size_t cap)
{
    size_t total = 0;
    if (counts_prefix)
        counts_prefix[0] = 0;
    for (size_t h = 0; h < nh; h++) {
        size_t room = total < cap ? cap - total : 0;
        size_t n = ref_php_match(t, blob + off[h], (size_t)(off[h + 1] - off[h]), find_all,
                                 (uint32_t)h, (out && room) ? out + 4 * total : NULL, room);
        total += n;
        if (counts_prefix)
            counts_prefix[h + 1] = total;
    }
    return total;
}

size_t ref_php_scan_checksum(void *t, const uint8_t *blob, const uint64_t *off, size_t nh,
                             uint64_t *checksum)
{
    AC_TRIE_t *trie = (AC_TRIE_t *)t;
    if (trie->trie_open)
        ac_trie_finalize(trie);
    ref_payload p = { 0, 0, NULL, 0, 0, 0 };
    for (size_t h = 0; h < nh; h++) {
        AC_TEXT_t txt;
        txt.astring = (const AC_ALPHABET_t *)(blob + off[h]);
        txt.length = (size_t)(off[h + 1] - off[h]);
        p.hay = (uint32_t)h;
        ac_trie_search(trie, &txt, 0, ref_handler, &p);
    }
    if (checksum)
        *checksum = p.checksum;
    return p.n;
}

/* The same scan `reps` times in one call: what bench.py's all-cores baseline gives each thread (one ctypes call per
 * thread, the loop in C -- a Python loop of short calls measured the interpreter's thread scheduling, not the trie). */
size_t ref_php_scan_checksum_reps(void *t, const uint8_t *blob, const uint64_t *off, size_t nh, size_t reps, uint64_t *checksum)
{
    size_t n = 0;
    for (size_t r = 0; r < reps; r++)
        n = ref_php_scan_checksum(t, blob, off, nh, checksum);
    return n;
}

/* examples/benchmark.php:55-72 with the reference's own C underneath: `samples` x (create, add every pattern in PHP's
 * reverse order, finalize -- lazily, on the first match, php_ahocorasick.c:707 --, one ac_trie_search per haystack,
 * release).  seconds[0] += build (create + add + finalize), seconds[1] += the searches, seconds[2] += release.  Returns
 * the records of the last sample.  (No Zend layer: no zval arrays are built for the records.) */
#include <time.h>
static double ref_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
size_t ref_benchmark_sample(const uint8_t *pat_blob, const uint64_t *pat_off, size_t np, const uint8_t *hay_blob, const uint64_t *hay_off,
                            size_t nh, size_t samples, double *seconds)
{
    size_t n = 0;
    for (size_t s = 0; s < samples; s++) {
        double t0 = ref_now();
        void *t = ref_create();
        ref_php_add_patterns(t, pat_blob, pat_off, np, 0);
        ref_finalize(t);
        double t1 = ref_now();
        uint64_t ck = 0;
        n = ref_php_scan_checksum(t, hay_blob, hay_off, nh, &ck);
        double t2 = ref_now();
        ref_release(t);
        double t3 = ref_now();
        seconds[0] += t1 - t0; seconds[1] += t2 - t1; seconds[2] += t3 - t2;
    }
    return n;
}

/* multifast_replace + multifast_rep_flush(keep=0) over one whole text, the callback's chunks concatenated */
typedef struct { uint8_t *out; size_t cap, n; } ref_rep_out;
static void ref_rep_cb(AC_TEXT_t *txt, void *param)
{
    ref_rep_out *o = (ref_rep_out *)param;
    for (size_t i = 0; i < txt->length; i++) {
        if (o->n < o->cap)
            o->out[o->n] = (uint8_t)txt->astring[i];
        o->n++;
    }
}

size_t ref_replace(void *t, const uint8_t *text, size_t len, int mode, uint8_t *out, size_t cap)
{
    AC_TRIE_t *trie = (AC_TRIE_t *)t;
    if (trie->trie_open)
        return (size_t)-1;
    AC_TEXT_t txt;
    txt.astring = (const AC_ALPHABET_t *)text;
    txt.length = len;
    ref_rep_out o = { out, cap, 0 };
    int rc = multifast_replace(trie, &txt, (MF_REPLACE_MODE_t)mode, ref_rep_cb, &o);
    if (rc < 0)
        return (size_t)(long)rc;
    multifast_rep_flush(trie, 0);
    return o.n;
}

/* Node count = 1 + number of edges, by DFS over the reference's own structs. */
#include "node.h"
static size_t count_nodes(const ACT_NODE_t *n)
{
    size_t c = 1;
    for (size_t i = 0; i < n->outgoing_size; i++)
        c += count_nodes(n->outgoing[i].next);
    return c;
}
size_t ref_node_count(const void *t) { return count_nodes(((const AC_TRIE_t *)t)->root); }
