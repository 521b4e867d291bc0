/*
 * ac_oracle.c -- TEST INFRASTRUCTURE ONLY (see ac_oracle.h).
 *
 * Plain-C restatement of the algorithm the reference runs on its hot path.
 * It deliberately keeps the reference's algorithmic shape (goto edges found by
 * linear search while building and by signed binary search while scanning,
 * failure links found by the "longest proper suffix that is a trie path"
 * definition, per-node output lists gathered by walking the failure chain) so
 * that it can be read side by side with the cited lines.  Nodes are indices
 * into one growable array instead of malloc'ed structs; that is storage, not
 * behaviour.
 *
 * Parity status: PINNED against tests/golden/kat.json (reference .phpt
 * expectations) and tests/golden/golden.json (streams from the reference's own
 * C sources, oracle/_ref).
 */
#include "ac_oracle.h"

#include <stdlib.h>
#include <string.h>

#define NONE 0xFFFFFFFFu

/* src/multifast/node.h:63-67.  AC_ALPHABET_t is plain `char`
 * (src/multifast/actypes.h:41) which is signed on x86-64: edges sort and
 * binary-search in SIGNED order.  Unobservable in results, kept for fidelity. */
typedef struct {
    int8_t alpha;
    uint32_t next;
} edge_t;

/* src/multifast/node.h:37-58, fields the path touches. */
typedef struct {
    int final;
    uint32_t depth;
    uint32_t failure;           /* NONE for the root (ahocorasick.c:352-353) */
    edge_t *out;
    uint32_t out_n, out_cap;
    orc_pattern *matched;
    uint32_t m_n, m_cap;
} node_t;

struct orc_trie {
    node_t *nodes;
    size_t n, cap;
    int open;                   /* AC_TRIE_t.trie_open */
    size_t patterns_count;
    uint32_t last_node;         /* AC_TRIE_t.last_node     (ahocorasick.h:55) */
    size_t base_position;       /* AC_TRIE_t.base_position (ahocorasick.h:56) */
    /* pattern byte copies (the reference keeps them in its mpool) */
    uint8_t **copies;
    size_t ncopies, copies_cap;
    /* replacement texts (AC_PATTERN_t.rtext, actypes.h:88-94), by caller-order pattern index */
    uint8_t **rep;
    uint32_t *rep_len;
    size_t rep_n;
};

static uint32_t node_new(orc_trie *t)
{
    if (t->n == t->cap) {
        t->cap = t->cap ? t->cap * 2 : 1024;
        t->nodes = (node_t *)realloc(t->nodes, t->cap * sizeof(node_t));
    }
    node_t *nd = &t->nodes[t->n];
    memset(nd, 0, sizeof(*nd));
    nd->failure = NONE;
    return (uint32_t)t->n++;
}

/* src/multifast/node.c:96-108 node_find_next: linear search, used while the
 * trie is still open (edges unsorted). */
static uint32_t find_next(const orc_trie *t, uint32_t n, int8_t alpha)
{
    const node_t *nd = &t->nodes[n];
    for (uint32_t i = 0; i < nd->out_n; i++)
        if (nd->out[i].alpha == alpha)
            return nd->out[i].next;
    return NONE;
}

/* src/multifast/node.c:119-140 node_find_next_bs: binary search over edges
 * sorted by signed alpha. */
static uint32_t find_next_bs(const orc_trie *t, uint32_t n, int8_t alpha)
{
    const node_t *nd = &t->nodes[n];
    int lo = 0, hi = (int)nd->out_n - 1;
    while (lo <= hi) {
        int mid = (lo + hi) >> 1;
        int8_t amid = nd->out[mid].alpha;
        if (alpha > amid)
            lo = mid + 1;
        else if (alpha < amid)
            hi = mid - 1;
        else
            return nd->out[mid].next;
    }
    return NONE;
}

/* src/multifast/node.c:188-203 node_create_next + :268-280 node_add_edge. */
static uint32_t create_next(orc_trie *t, uint32_t n, int8_t alpha)
{
    uint32_t nx = node_new(t);              /* may move t->nodes */
    node_t *nd = &t->nodes[n];
    if (nd->out_n == nd->out_cap) {
        nd->out_cap = nd->out_cap ? nd->out_cap * 2 : 2;
        nd->out = (edge_t *)realloc(nd->out, nd->out_cap * sizeof(edge_t));
    }
    nd->out[nd->out_n].alpha = alpha;
    nd->out[nd->out_n].next = nx;
    nd->out_n++;
    t->nodes[nx].depth = nd->depth + 1;     /* ahocorasick.c:118 */
    return nx;
}

/* src/multifast/node.c:150-175 node_has_pattern: a pattern is "already there"
 * when one of equal length and equal bytes is listed.  (The reference compares
 * its stored copies, which mpool_strndup truncates at NUL, mpool.c:166-181;
 * two equal-length entries on one node are always suffixes of the same path,
 * hence byte-equal, so comparing true copies decides identically.) */
static int has_pattern(const node_t *nd, const orc_pattern *p)
{
    for (uint32_t i = 0; i < nd->m_n; i++) {
        const orc_pattern *q = &nd->matched[i];
        if (q->length != p->length)
            continue;
        if (memcmp(q->bytes, p->bytes, p->length) == 0)
            return 1;
    }
    return 0;
}

/* src/multifast/node.c:205-229 node_accept_pattern (+ :238-261 deep copy when
 * copy=1; the PHP layer always passes copy=1, php_ahocorasick.c:484). */
static void accept_pattern(orc_trie *t, uint32_t n, const orc_pattern *p, int copy)
{
    node_t *nd = &t->nodes[n];
    if (has_pattern(nd, p))
        return;
    if (nd->m_n == nd->m_cap) {
        nd->m_cap = nd->m_cap ? nd->m_cap + 2 : 1;   /* node.c:394-412 */
        nd->matched = (orc_pattern *)realloc(nd->matched, nd->m_cap * sizeof(orc_pattern));
    }
    orc_pattern *dst = &nd->matched[nd->m_n++];
    *dst = *p;
    if (copy) {
        uint8_t *c = (uint8_t *)malloc(p->length ? p->length : 1);
        memcpy(c, p->bytes, p->length);
        if (t->ncopies == t->copies_cap) {
            t->copies_cap = t->copies_cap ? t->copies_cap * 2 : 256;
            t->copies = (uint8_t **)realloc(t->copies, t->copies_cap * sizeof(uint8_t *));
        }
        t->copies[t->ncopies++] = c;
        dst->bytes = c;
    }
}

/* src/multifast/ahocorasick.c:59-77 */
orc_trie *orc_create(void)
{
    orc_trie *t = (orc_trie *)calloc(1, sizeof(*t));
    node_new(t);                /* root: depth 0, failure NONE */
    t->open = 1;
    return t;
}

/* src/multifast/ahocorasick.c:91-131 */
int orc_add(orc_trie *t, const uint8_t *p, size_t len, uint32_t index)
{
    if (!t->open)
        return ORC_TRIE_CLOSED;                     /* :98-99   */
    if (!len)
        return ORC_ZERO_PATTERN;                    /* :101-102 */
    if (len > ORC_PATTERN_MAX_LENGTH)
        return ORC_LONG_PATTERN;                    /* :104-105 */

    uint32_t n = 0;
    for (size_t i = 0; i < len; i++) {              /* :107-121 */
        int8_t alpha = (int8_t)p[i];
        uint32_t nx = find_next(t, n, alpha);
        if (nx == NONE)
            nx = create_next(t, n, alpha);
        n = nx;
    }
    if (t->nodes[n].final)
        return ORC_DUPLICATE_PATTERN;               /* :123-124: first inserted wins */

    t->nodes[n].final = 1;                          /* :126 */
    orc_pattern pat = { p, (uint32_t)len, index };
    accept_pattern(t, n, &pat, 1);                  /* :127 */
    t->patterns_count++;
    return ORC_SUCCESS;
}

/* src/multifast/ahocorasick.c:344-368 ac_trie_set_failure: the failure node
 * is where the LONGEST proper suffix prefix[i..depth) (smallest i >= 1) that
 * is a path from the root ends; the root when there is none. */
static void set_failure(orc_trie *t, uint32_t n, const int8_t *prefix)
{
    if (n == 0)
        return;
    node_t *nd = &t->nodes[n];
    for (uint32_t i = 1; i < nd->depth; i++) {
        uint32_t m = 0;
        for (uint32_t j = i; j < nd->depth && m != NONE; j++)
            m = find_next(t, m, prefix[j]);
        if (m != NONE) {
            nd->failure = m;
            break;
        }
    }
    if (nd->failure == NONE)
        nd->failure = 0;
}

/* src/multifast/ahocorasick.c:381-396 ac_trie_traverse_setfailure (DFS). */
static void traverse_setfailure(orc_trie *t, uint32_t n, int8_t *prefix)
{
    set_failure(t, n, prefix);
    for (uint32_t i = 0; i < t->nodes[n].out_n; i++) {
        prefix[t->nodes[n].depth] = t->nodes[n].out[i].alpha;
        traverse_setfailure(t, t->nodes[n].out[i].next, prefix);
    }
}

static int edge_cmp(const void *l, const void *r)
{
    /* src/multifast/node.c:300-310: alphas are unique per node */
    return ((const edge_t *)l)->alpha >= ((const edge_t *)r)->alpha ? 1 : -1;
}

/* src/multifast/node.c:424-441 node_collect_matches: append the accepted
 * patterns of EVERY node on the failure chain (shallow copies, de-duplicated
 * by node_has_pattern), OR in `final`, then sort the edges (:315-319).  The
 * node's own pattern (if any) is already first, every later one is a shorter
 * suffix => strictly decreasing length. */
static void collect_matches(orc_trie *t, uint32_t n)
{
    uint32_t f = n;
    while ((f = t->nodes[f].failure) != NONE) {
        for (uint32_t i = 0; i < t->nodes[f].m_n; i++) {
            orc_pattern p = t->nodes[f].matched[i];
            accept_pattern(t, n, &p, 0);
        }
        if (t->nodes[f].final)
            t->nodes[n].final = 1;
    }
    if (t->nodes[n].out_n)  /* (the reference calls qsort on a NULL vector of 0 edges too; UBSan objects) */
        qsort(t->nodes[n].out, t->nodes[n].out_n, sizeof(edge_t), edge_cmp);
}

/* src/multifast/ahocorasick.c:407-422 ac_trie_traverse_action, top_down=1. */
static void traverse_collect(orc_trie *t, uint32_t n)
{
    collect_matches(t, n);
    for (uint32_t i = 0; i < t->nodes[n].out_n; i++)
        traverse_collect(t, t->nodes[n].out[i].next);
}

/* src/multifast/ahocorasick.c:143-155 */
void orc_finalize(orc_trie *t)
{
    int8_t prefix[ORC_PATTERN_MAX_LENGTH];
    traverse_setfailure(t, 0, prefix);              /* :149 */
    traverse_collect(t, 0);                         /* :151 */
    t->open = 0;                                    /* :154 */
}

/* src/multifast/ahocorasick.c:175-241, working mode SEARCH.  keep == 0: the call starts at the root
 * with base_position 0 (:191-192, :330-335).  keep != 0: it continues from the node and the stream
 * offset the previous call left (:194, :220, :236-238); a call the callback stopped saves neither. */
int orc_search_keep(orc_trie *t, const uint8_t *text, size_t len, int keep, orc_callback cb, void *user)
{
    if (t->open)
        return -1;                                  /* :183-184 */

    if (!keep) {                                    /* ac_trie_reset :330-335 */
        t->last_node = 0;
        t->base_position = 0;
    }
    size_t position = 0;
    uint32_t current = t->last_node;                /* :194 */
    while (position < len) {                        /* :199 */
        uint32_t next = find_next_bs(t, current, (int8_t)text[position]);
        if (next == NONE) {
            if (t->nodes[current].failure != NONE)  /* not the root: follow failure, */
                current = t->nodes[current].failure;/* position NOT advanced (:203-204) */
            else
                position++;                         /* :205-206 */
        } else {
            current = next;                         /* :210-211 */
            position++;
        }
        /* :214: report only after a goto transition, never after a failure hop */
        if (t->nodes[current].final && next != NONE) {
            orc_match m;
            m.position = position + t->base_position;   /* :220 */
            m.size = t->nodes[current].m_n;
            m.patterns = t->nodes[current].matched;
            if (cb(&m, user))
                return 1;                           /* :225-231 */
        }
    }
    t->last_node = current;                         /* :236 */
    t->base_position += position;                   /* :237 */
    return 0;
}

int orc_search(orc_trie *t, const uint8_t *text, size_t len, orc_callback cb, void *user)
{
    return orc_search_keep(t, text, len, 0, cb, user);
}

/* src/multifast/ahocorasick.c:288-296 */
void orc_release(orc_trie *t)
{
    if (!t)
        return;
    for (size_t i = 0; i < t->n; i++) {
        free(t->nodes[i].out);
        free(t->nodes[i].matched);
    }
    for (size_t i = 0; i < t->ncopies; i++)
        free(t->copies[i]);
    free(t->copies);
    for (size_t i = 0; i < t->rep_n; i++)
        free(t->rep[i]);
    free(t->rep);
    free(t->rep_len);
    free(t->nodes);
    free(t);
}

size_t orc_node_count(const orc_trie *t) { return t->n; }
size_t orc_pattern_count(const orc_trie *t) { return t->patterns_count; }

/* ------------------------- search and replace ----------------------------- */

/* AC_PATTERN_t.rtext as ac_trie_add keeps it with copy=1 (node.c:238-251): the bytes go through
 * mpool_strndup (mpool.c:165-178), i.e. strncpy -- copied up to the first NUL, zero-filled after it --
 * while rtext.length stays the caller's.  NULL means "not to be replaced". */
void orc_set_replacement(orc_trie *t, uint32_t index, const uint8_t *r, size_t rlen)
{
    if (index >= t->rep_n) {
        size_t n = (size_t)index + 1;
        t->rep = (uint8_t **)realloc(t->rep, n * sizeof(uint8_t *));
        t->rep_len = (uint32_t *)realloc(t->rep_len, n * sizeof(uint32_t));
        for (size_t i = t->rep_n; i < n; i++) { t->rep[i] = NULL; t->rep_len[i] = 0; }
        t->rep_n = n;
    }
    free(t->rep[index]);
    t->rep[index] = NULL;
    t->rep_len[index] = 0;
    if (!r)
        return;
    uint8_t *c = (uint8_t *)calloc(rlen + 1, 1);
    strncpy((char *)c, (const char *)r, rlen);
    t->rep[index] = c;
    t->rep_len[index] = (uint32_t)rlen;
}

static const orc_pattern *to_be_replaced(const orc_trie *t, const node_t *nd)
{
    /* node.c:337-362 node_book_replacement: the longest matched pattern that has a replacement */
    const orc_pattern *longest = NULL;
    if (!nd->final)
        return NULL;
    for (uint32_t j = 0; j < nd->m_n; j++) {
        const orc_pattern *p = &nd->matched[j];
        if (p->index < t->rep_n && t->rep[p->index] != NULL) {
            if (!longest || p->length > longest->length)
                longest = p;
        }
    }
    return longest;
}

typedef struct { const orc_pattern *pattern; size_t position; } nominee_t;   /* replace.h:57-61 */

typedef struct {
    nominee_t *noms;
    size_t n, cap;
    size_t curser;                                  /* replace.h:84 */
    uint8_t *out;
    size_t out_n, out_cap;
} repdata_t;

static void rep_append(repdata_t *rd, const uint8_t *p, size_t n)   /* replace.c:279-305, unbuffered */
{
    for (size_t i = 0; i < n; i++) {
        if (rd->out_n < rd->out_cap)
            rd->out[rd->out_n] = p[i];
        rd->out_n++;
    }
}

/* replace.c:222-272 mf_repdata_booknominee */
static void book_nominee(repdata_t *rd, int mode, const orc_pattern *pat, size_t position)
{
    if (!pat)
        return;
    size_t new_start = position - pat->length;
    if (mode == 2) {                                /* MF_REPLACE_MODE_LAZY :236-250 */
        if (new_start < rd->curser)
            return;
        if (rd->n > 0 && new_start < rd->noms[rd->n - 1].position)
            return;
    } else {                                        /* DEFAULT / NORMAL :252-268 */
        while (rd->n > 0) {
            const nominee_t *prev = &rd->noms[rd->n - 1];
            size_t prev_start = prev->position - prev->pattern->length;
            if (new_start <= prev_start)
                rd->n--;                            /* a factor of the new nominee */
            else
                break;
        }
    }
    if (rd->n == rd->cap) {
        rd->cap = rd->cap ? rd->cap * 2 : 64;
        rd->noms = (nominee_t *)realloc(rd->noms, rd->cap * sizeof(nominee_t));
    }
    rd->noms[rd->n].pattern = pat;
    rd->noms[rd->n].position = position;
    rd->n++;
}

/* replace.c:402-455 mf_repdata_do_replace for a text that is wholly in memory (base_position 0: every
 * factor comes from the input, mf_repdata_appendfactor :314-351 first branch; `to < from` appends nothing) */
static void do_replace(const orc_trie *t, repdata_t *rd, const uint8_t *text, size_t to_position)
{
    size_t index = 0;
    for (; index < rd->n; index++) {
        const nominee_t *nom = &rd->noms[index];
        size_t start = nom->position - nom->pattern->length;
        if (to_position <= start)
            break;
        if (start >= rd->curser)
            rep_append(rd, text + rd->curser, start - rd->curser);
        rep_append(rd, t->rep[nom->pattern->index], t->rep_len[nom->pattern->index]);
        rd->curser = nom->position;
    }
    memmove(rd->noms, rd->noms + index, (rd->n - index) * sizeof(nominee_t));
    rd->n -= index;
    if (to_position > rd->curser) {
        rep_append(rd, text + rd->curser, to_position - rd->curser);
        rd->curser = to_position;
    }
}

/* One whole text: multifast_replace (replace.c:469-542) followed by multifast_rep_flush(keep=0) (:552-568).
 * mode 0/1 = DEFAULT/NORMAL, 2 = LAZY (replace.h:34-46).  Writes at most `cap` bytes, returns the length of the
 * result, or (size_t)-2 when no pattern has a replacement (multifast_replace returns -2 and emits nothing, :482-483)
 * and (size_t)-1 on an open trie (:479-480). */
size_t orc_replace(orc_trie *t, const uint8_t *text, size_t len, int mode, uint8_t *out, size_t cap)
{
    if (t->open)
        return (size_t)-1;
    int any = 0;
    for (size_t i = 0; i < t->n && !any; i++)       /* rd->has_replacement, replace.c:95-131 */
        any = to_be_replaced(t, &t->nodes[i]) != NULL;
    if (!any)
        return (size_t)-2;
    repdata_t rd;
    memset(&rd, 0, sizeof(rd));
    rd.out = out;
    rd.out_cap = cap;
    size_t position = 0;
    uint32_t current = 0;
    while (position < len) {                        /* :498-522: the same walk as ac_trie_search */
        uint32_t next = find_next_bs(t, current, (int8_t)text[position]);
        if (next == NONE) {
            if (t->nodes[current].failure != NONE)
                current = t->nodes[current].failure;
            else
                position++;
        } else {
            current = next;
            position++;
        }
        if (t->nodes[current].final && next != NONE)
            book_nominee(&rd, mode, to_be_replaced(t, &t->nodes[current]), position);
    }
    do_replace(t, &rd, text, len - t->nodes[current].depth);   /* :529-532: up to the backlog position */
    do_replace(t, &rd, text, len);                             /* multifast_rep_flush(keep=0) :556-558 */
    free(rd.noms);
    return rd.out_n;
}

/* ------------------------- PHP-layer emulation --------------------------- */

/* src/php_ahocorasick.c:389-489 */
size_t orc_php_add_patterns(orc_trie *t, const uint8_t *blob, const uint64_t *off,
                            size_t n, uint32_t base_index)
{
    size_t accepted = 0;
    for (size_t k = n; k-- > 0;) {                  /* reverse caller order */
        int st = orc_add(t, blob + off[k], (size_t)(off[k + 1] - off[k]),
                         base_index + (uint32_t)k);
        if (st == ORC_SUCCESS)                      /* status ignored by PHP (:484) */
            accepted++;
    }
    return accepted;
}

typedef struct {
    int ret_val;            /* ahocorasick_callback_payload_t.retVal */
    uint32_t hay;
    uint32_t *out;
    size_t cap, n;
    uint64_t checksum;
} php_payload;

/* src/php_ahocorasick.c:542-589 php_ahocorasick_match_handler */
static int php_handler(const orc_match *m, void *param)
{
    php_payload *p = (php_payload *)param;
    for (size_t j = 0; j < m->size; j++) {          /* :555 */
        uint32_t pos = (uint32_t)m->position;       /* "pos" :564 */
        uint32_t start = pos - m->patterns[j].length; /* "start_postion" :578 */
        uint32_t idx = m->patterns[j].index;
        if (p->out && p->n < p->cap) {
            uint32_t *r = p->out + 4 * p->n;
            r[0] = p->hay; r[1] = pos; r[2] = start; r[3] = idx;
        }
        p->checksum += ((uint64_t)p->hay * 0x9E3779B1u) ^ ((uint64_t)pos * 0x85EBCA77u) ^
                       ((uint64_t)start * 0xC2B2AE3Du) ^ (uint64_t)idx;
        p->n++;
    }
    return p->ret_val == 0 ? 0 : 1;                 /* :588 */
}

/* src/php_ahocorasick.c:664-746 */
size_t orc_php_match(orc_trie *t, const uint8_t *text, size_t len, int find_all,
                     uint32_t hay, uint32_t *out, size_t cap)
{
    if (t->open)
        orc_finalize(t);                            /* lazy finalize :707 */
    php_payload p = { find_all ? 0 : 1, hay, out, cap, 0, 0 };   /* :742 */
    orc_search(t, text, len, php_handler, &p);      /* :745, keep=0 */
    return p.n;
}

/* One chunk of a stream: ac_trie_search with keep = !first (ahocorasick.c:175-241); positions count
 * from the start of the stream.  Every event is reported (the PHP layer has no streaming entry). */
size_t orc_stream_feed(orc_trie *t, const uint8_t *text, size_t len, int first, uint32_t *out, size_t cap)
{
    if (t->open)
        orc_finalize(t);
    php_payload p = { 0, 0, out, cap, 0, 0 };
    orc_search_keep(t, text, len, first ? 0 : 1, php_handler, &p);
    return p.n;
}

size_t orc_php_match_batch(orc_trie *t, const uint8_t *blob, const uint64_t *off,
                           size_t nh, int find_all, uint64_t *counts_prefix,
                           uint32_t *out, size_t cap)
{
    size_t total = 0;
    if (counts_prefix)
        counts_prefix[0] = 0;
    for (size_t h = 0; h < nh; h++) {
        size_t room = total < cap ? cap - total : 0;
        size_t n = orc_php_match(t, blob + off[h], (size_t)(off[h + 1] - off[h]), find_all,
                                 (uint32_t)h, (out && room) ? out + 4 * total : NULL, room);
        total += n;
        if (counts_prefix)
            counts_prefix[h + 1] = total;
    }
    return total;
}

size_t orc_php_scan_checksum(orc_trie *t, const uint8_t *blob, const uint64_t *off,
                             size_t nh, uint64_t *checksum)
{
    if (t->open)
        orc_finalize(t);
    php_payload p = { 0, 0, NULL, 0, 0, 0 };
    for (size_t h = 0; h < nh; h++) {
        p.hay = (uint32_t)h;
        orc_search(t, blob + off[h], (size_t)(off[h + 1] - off[h]), php_handler, &p);
    }
    if (checksum)
        *checksum = p.checksum;
    return p.n;
}

/* The same scan `reps` times in one call (bench.py's all-cores baseline: one call per thread, the loop in C). */
size_t orc_php_scan_checksum_reps(orc_trie *t, const uint8_t *blob, const uint64_t *off, size_t nh, size_t reps, uint64_t *checksum)
{
    size_t n = 0;
    for (size_t r = 0; r < reps; r++)
        n = orc_php_scan_checksum(t, blob, off, nh, checksum);
    return n;
}

/* examples/benchmark.php:55-72 over this restatement: `samples` x (create, add in PHP's reverse order, finalize, one search
 * per haystack, release); seconds[0] += build, [1] += searches, [2] += release.  Returns the last sample's records. */
#include <time.h>
static double orc_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
size_t orc_benchmark_sample(const uint8_t *pat_blob, const uint64_t *pat_off, size_t np, const uint8_t *hay_blob, const uint64_t *hay_off,
                            size_t nh, size_t samples, double *seconds)
{
    size_t n = 0;
    for (size_t s = 0; s < samples; s++) {
        double t0 = orc_now();
        orc_trie *t = orc_create();
        orc_php_add_patterns(t, pat_blob, pat_off, np, 0);
        orc_finalize(t);
        double t1 = orc_now();
        uint64_t ck = 0;
        n = orc_php_scan_checksum(t, hay_blob, hay_off, nh, &ck);
        double t2 = orc_now();
        orc_release(t);
        double t3 = orc_now();
        seconds[0] += t1 - t0; seconds[1] += t2 - t1; seconds[2] += t3 - t2;
    }
    return n;
}
