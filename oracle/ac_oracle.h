/*
 * ac_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's Aho-Corasick hot path (MultiFast 2.0 as
 * bundled by ph4r05/php_aho_corasick) plus the thin slice of the PHP glue that
 * decides WHAT is inserted and WHAT a match record looks like.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this.  The product (php_aho_corasick_amd/) never links, imports or
 * calls anything in oracle/.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this restatement against
 *   (1) the known-answer vectors transcribed from the reference's own
 *       tests/test1..6.phpt (tests/golden/kat.json), and
 *   (2) golden streams produced by the reference's own C sources compiled in
 *       place (oracle/_ref, recipe in oracle/Makefile; generator
 *       tests/golden/make_golden.py; fixtures tests/golden/golden.json).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/).
 */
#ifndef AC_ORACLE_H
#define AC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/multifast/actypes.h:148 */
#define ORC_PATTERN_MAX_LENGTH 1024

/* src/multifast/actypes.h:118-125 (AC_STATUS_t) */
enum orc_status {
    ORC_SUCCESS = 0,
    ORC_DUPLICATE_PATTERN = 1,
    ORC_LONG_PATTERN = 2,
    ORC_ZERO_PATTERN = 3,
    ORC_TRIE_CLOSED = 4
};

typedef struct orc_trie orc_trie;

/* One accepted pattern as seen by the match callback
 * (src/multifast/actypes.h:88-94, reduced to what the PHP handler reads:
 * php_ahocorasick.c:555-583 uses id.u.number / aux / strlen(value)). */
typedef struct orc_pattern {
    const uint8_t *bytes;   /* pattern text (owned by the trie) */
    uint32_t length;        /* ptext.length */
    uint32_t index;         /* caller-order pattern index ("aux" stand-in) */
} orc_pattern;

/* src/multifast/actypes.h:106-113 (AC_MATCH_t) */
typedef struct orc_match {
    const orc_pattern *patterns;
    size_t size;
    size_t position;
} orc_match;

/* src/multifast/actypes.h:138 (AC_MATCH_CALBACK_f): non-zero return stops. */
typedef int (*orc_callback)(const orc_match *m, void *user);

orc_trie *orc_create(void);                                       /* ahocorasick.c:59-77   */
int  orc_add(orc_trie *t, const uint8_t *p, size_t len,
             uint32_t index);                                     /* ahocorasick.c:91-131  */
void orc_finalize(orc_trie *t);                                   /* ahocorasick.c:143-155 */
int  orc_search(orc_trie *t, const uint8_t *text, size_t len,
                orc_callback cb, void *user);                     /* ahocorasick.c:175-241 (keep=0) */
int  orc_search_keep(orc_trie *t, const uint8_t *text, size_t len, int keep,
                     orc_callback cb, void *user);                /* ahocorasick.c:175-241         */
void orc_release(orc_trie *t);                                    /* ahocorasick.c:288-296 */

/* Search and replace (not reachable from the PHP surface: rtext is always NULL there, php_ahocorasick.c:469-470).
 * orc_set_replacement before orc_add / orc_php_add_patterns of that index; r == NULL: not to be replaced. */
void orc_set_replacement(orc_trie *t, uint32_t index, const uint8_t *r, size_t rlen);   /* actypes.h:88-94 rtext, node.c:248-251 */
size_t orc_replace(orc_trie *t, const uint8_t *text, size_t len, int mode,
                   uint8_t *out, size_t cap);                     /* replace.c:469-542 + :552-568 (keep=0) */

size_t orc_node_count(const orc_trie *t);
size_t orc_pattern_count(const orc_trie *t);                      /* AC_TRIE_t.patterns_count */

/* ---- PHP-layer emulation (what php_ahocorasick.c does around the core) ---- */

/* php_ahocorasick.c:389-489: one ahocorasick_init()/add_patterns() call.
 * `blob`/`off` hold n patterns in CALLER order (off has n+1 entries);
 * pattern k gets index base_index+k.  They are inserted in REVERSE caller
 * order (the local list is built by prepending, :410-421, and then walked
 * front to back, :457-486) and ac_trie_add's status is ignored (:484).
 * Returns the number of patterns the trie accepted. */
size_t orc_php_add_patterns(orc_trie *t, const uint8_t *blob,
                            const uint64_t *off, size_t n, uint32_t base_index);

/* php_ahocorasick.c:664-746 + handler :542-589 for one haystack.
 * Appends one record per reported pattern: 4 x u32 {hay, pos, start, index},
 * pos = m->position, start = pos - strlen(value) (:578).  find_all == 0
 * stops after the first EVENT (all of its patterns are still emitted, :555-588).
 * Lazily finalizes (:707).  Returns the number of records the call produced;
 * only the first `cap` are written to out. */
size_t orc_php_match(orc_trie *t, const uint8_t *text, size_t len, int find_all,
                     uint32_t hay, uint32_t *out, size_t cap);

/* One chunk of a stream (MultiFast's keep=1 mode, ahocorasick.c:186-194,220,236-238; not reachable from
 * PHP, which always passes keep=0 at php_ahocorasick.c:745): first != 0 starts a new stream.  Records as
 * orc_php_match with hay = 0 and positions counted from the start of the stream. */
size_t orc_stream_feed(orc_trie *t, const uint8_t *text, size_t len, int first,
                       uint32_t *out, size_t cap);

/* Batch convenience used by goldens and by bench.py's cpu_baseline: runs
 * orc_php_match over nh haystacks (blob + nh+1 offsets), records appended in
 * haystack order, counts_prefix[nh+1] filled.  Returns total records. */
size_t orc_php_match_batch(orc_trie *t, const uint8_t *blob, const uint64_t *off,
                           size_t nh, int find_all, uint64_t *counts_prefix,
                           uint32_t *out, size_t cap);

/* Count + checksum only (callback = count + checksum).
 * checksum = sum over records of (hay*0x9E3779B1 ^ pos*0x85EBCA77 ^ start*0xC2B2AE3D ^ index) mod 2^64. */
size_t orc_php_scan_checksum(orc_trie *t, const uint8_t *blob, const uint64_t *off,
                             size_t nh, uint64_t *checksum);

/* the same `reps` times in one call; examples/benchmark.php:55-72 (samples x build + one search per haystack + release) */
size_t orc_php_scan_checksum_reps(orc_trie *t, const uint8_t *blob, const uint64_t *off, size_t nh, size_t reps, uint64_t *checksum);
size_t orc_benchmark_sample(const uint8_t *pat_blob, const uint64_t *pat_off, size_t np, const uint8_t *hay_blob, const uint64_t *hay_off,
                            size_t nh, size_t samples, double *seconds);

#ifdef __cplusplus
}
#endif
#endif
