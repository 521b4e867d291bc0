/*
 * php_ahocorasick_amd.c -- Zend shim: the reference's PHP function surface on top of libacmi.so.
 *
 * NOT COMPILED IN THIS REPO'S BUILD: the build image has no php, php-config, phpize or Zend headers
 * (SURVEY.md section 8c), so this file is the reference-side binding a maintainer adds, written
 * against the PHP 7/8 extension API and kept next to the C ABI it binds (include/acmi.h).  Build
 * recipe in php_ext/config.m4; INTEGRATION.md walks through it.  The same surface is exercised in
 * this repo through php_aho_corasick_amd/api.py (a line-for-line behavioural mirror of this file).
 *
 * Function table kept verbatim from the reference (src/php_ahocorasick_arginfo.h:36-44,
 * src/php_ahocorasick.stub.php:12-37):
 *   ahocorasick_init(array): resource|false        ahocorasick_add_patterns(res, array): bool
 *   ahocorasick_finalize(res): bool                ahocorasick_match(string, res, bool=true): array|false
 *   ahocorasick_isValid(res): bool                 ahocorasick_deinit(res): bool
 * plus one addition: ahocorasick_match_batch(array $haystacks, res, bool $findAll=true): array|false
 *
 * What changed underneath: the six MultiFast call sites (php_ahocorasick.c:140,484,504,745,812,821)
 * become acmi_* calls; the per-event callback (php_ahocorasick.c:542-589) becomes a loop over the
 * drained acmi_record array.
 */
#ifdef HAVE_CONFIG_H
#include "config.h"
#endif

#include "php.h"
#include "ext/standard/info.h"
#include "zend_exceptions.h"

#include "acmi.h"

#define PHP_AHOCORASICK_VERSION "0.0.7"            /* php_ahocorasick.h:29 */
#define ACMI_RES_NAME "AhoCorasick search"          /* php_ahocorasick.h:192 */

/* One pattern as the PHP layer holds it: refcounted key / value / aux zvals for the life of the
 * resource (the reference does the same with ZVAL_COPY, php_ahocorasick.c:268,291,296). */
typedef struct acmi_php_pattern {
    zval key, value, aux;       /* IS_UNDEF when absent */
    zend_long key_id;
    uint32_t id_type;           /* ACMI_ID_* */
} acmi_php_pattern;

typedef struct acmi_php_master {
    acmi_handle *h;
    acmi_php_pattern *patterns; /* indexed by acmi_record.pattern (caller order) */
    size_t n_patterns, cap;
    int finalized;
} acmi_php_master;

static int le_acmi_master;
static zend_class_entry *aho_exception_ce;

/* zend_array_is_list() exists from PHP 8.1 on; older versions get the same test spelled out
 * (keys must be exactly 0..n-1 in order). */
static int acmi_array_is_list(HashTable *ht)
{
#if PHP_VERSION_ID >= 80100
    return zend_array_is_list(ht);
#else
    zend_ulong expect = 0, idx;
    zend_string *key;
    ZEND_HASH_FOREACH_KEY(ht, idx, key) {
        if (key != NULL || idx != expect++) return 0;
    } ZEND_HASH_FOREACH_END();
    return 1;
#endif
}

static void master_free(acmi_php_master *m)
{
    size_t i;
    if (!m) return;
    for (i = 0; i < m->n_patterns; i++) {
        zval_ptr_dtor(&m->patterns[i].key);
        zval_ptr_dtor(&m->patterns[i].value);
        zval_ptr_dtor(&m->patterns[i].aux);
    }
    if (m->patterns) efree(m->patterns);
    if (m->h) acmi_destroy(m->h);               /* ac_trie_release: php_ahocorasick.c:504 */
    efree(m);
}

static void master_dtor(zend_resource *rsrc)    /* php_ahocorasick.c:494-512 */
{
    master_free((acmi_php_master *)rsrc->ptr);
    rsrc->ptr = NULL;
}

static const char *type_name(zval *z)           /* php_aho_type_str, php_ahocorasick.c:98-117 */
{
    switch (Z_TYPE_P(z)) {
    case IS_UNDEF: return "undef";   case IS_NULL: return "null";     case IS_FALSE: return "false";
    case IS_TRUE: return "true";     case IS_LONG: return "long";     case IS_DOUBLE: return "double";
    case IS_STRING: return "string"; case IS_ARRAY: return "array";   case IS_OBJECT: return "object";
    case IS_RESOURCE: return "resource"; default: return "reference";
    }
}

/* php_ahocorasick_process_pattern (php_ahocorasick.c:195-336).  0 ok, <0 failed (warned/thrown). */
static int parse_pattern(zend_long pidx, HashTable *sub, acmi_php_pattern *p)
{
    zend_string *key;
    zval *data;
    unsigned all = 0;
    ZVAL_UNDEF(&p->key); ZVAL_UNDEF(&p->value); ZVAL_UNDEF(&p->aux);
    p->key_id = 0; p->id_type = ACMI_ID_DEFAULT;

    ZEND_HASH_FOREACH_STR_KEY_VAL(sub, key, data) {
        unsigned found;
        ZVAL_DEREF(data);
        if (!key) found = 2;                                         /* bare numeric key == value */
        else if (zend_string_equals_literal_ci(key, "key")) found = 1;
        else if (zend_string_equals_literal_ci(key, "value")) found = 2;
        else if (zend_string_equals_literal_ci(key, "ignoreCase")) found = 4;
        else if (zend_string_equals_literal_ci(key, "id")) found = 8;
        else if (zend_string_equals_literal_ci(key, "aux")) found = 0x10;
        else {
            php_error_docref(NULL, E_WARNING, "Invalid structure (unrecognized sub-array key)! Only allowed are: "
                             "{key, id, value, aux, ignoreCase}. Cannot initialize. Pattern index: %ld", (long)pidx);
            return -2;
        }
        all |= found;
        if (found & 8) {
            if (Z_TYPE_P(data) != IS_LONG) {
                zend_throw_exception_ex(aho_exception_ce, 0, "Invalid type of pattern ID given (long required), "
                                        "type: %s, pattern index: %ld", type_name(data), (long)pidx);
                return -5;
            }
            p->key_id = Z_LVAL_P(data);
            p->id_type = ACMI_ID_NUMBER;
        }
        if (found & 0x10) ZVAL_COPY(&p->aux, data);
        if (found & 3) {
            if (Z_TYPE_P(data) != IS_STRING) {
                zend_throw_exception_ex(aho_exception_ce, 0, "Pattern %s has to be a string, type: %s, pattern "
                                        "index: %ld", found == 1 ? "key" : "value", type_name(data), (long)pidx);
                return -5;
            }
            if (found == 1) { ZVAL_COPY(&p->key, data); p->id_type = ACMI_ID_STRING; }
            else ZVAL_COPY(&p->value, data);
        }
    } ZEND_HASH_FOREACH_END();

    if (Z_ISUNDEF(p->value)) {
        php_error_docref(NULL, E_WARNING, "No value was specified for pattern index: %ld", (long)pidx);
        return -2;
    }
    if ((all & 1) && (all & 8)) {
        php_error_docref(NULL, E_WARNING, "Pattern can have either numeric or string identifier, not both! "
                         "Pattern index: %ld", (long)pidx);
        return -3;
    }
    if (all & 4)
        php_error_docref(NULL, E_WARNING, "ignoreCase attribute is deprecated and is ignored. Pattern index: %ld",
                         (long)pidx);
    return 0;
}

/* php_ahocorasick_process_patterns (php_ahocorasick.c:389-489): parse everything first, then hand
 * the batch to the library, which applies reverse insertion + duplicate / empty / long rules. */
static int process_patterns(acmi_php_master *m, HashTable *ht)
{
    uint32_t n = zend_hash_num_elements(ht), k = 0, i;
    acmi_php_pattern *tmp;
    acmi_pattern *abi;
    zval *elem;
    int rc = 0;
    if (n == 0) return 0;
    if (!acmi_array_is_list(ht)) {   /* the reference NULL-derefs here */
        php_error_docref(NULL, E_WARNING, "Invalid pattern structure! Cannot initialize.");
        return -4;
    }
    tmp = ecalloc(n, sizeof(*tmp));
    ZEND_HASH_FOREACH_VAL(ht, elem) {
        ZVAL_DEREF(elem);
        if (Z_TYPE_P(elem) != IS_ARRAY) {
            php_error_docref(NULL, E_WARNING, "Invalid pattern structure! Cannot initialize.");
            rc = -4;
            break;
        }
        rc = parse_pattern((zend_long)k, Z_ARRVAL_P(elem), &tmp[k]);
        k++;
        if (rc != 0) { rc = -1; break; }
    } ZEND_HASH_FOREACH_END();
    if (rc != 0) {
        for (i = 0; i < k; i++) { zval_ptr_dtor(&tmp[i].key); zval_ptr_dtor(&tmp[i].value); zval_ptr_dtor(&tmp[i].aux); }
        efree(tmp);
        return rc;
    }
    abi = ecalloc(n, sizeof(*abi));
    for (i = 0; i < n; i++) {
        abi[i].value = (const uint8_t *)Z_STRVAL(tmp[i].value);
        abi[i].length = (uint32_t)Z_STRLEN(tmp[i].value);
        abi[i].id_type = tmp[i].id_type;
        abi[i].id_number = tmp[i].id_type == ACMI_ID_NUMBER ? tmp[i].key_id : 0;
        abi[i].user = m->n_patterns + i;
    }
    rc = acmi_add_patterns(m->h, abi, n, NULL);         /* ac_trie_add loop: php_ahocorasick.c:484 */
    efree(abi);
    if (rc != ACMI_OK) {                                /* the zvals were ZVAL_COPY'd by parse_pattern: release them */
        for (i = 0; i < n; i++) { zval_ptr_dtor(&tmp[i].key); zval_ptr_dtor(&tmp[i].value); zval_ptr_dtor(&tmp[i].aux); }
        efree(tmp);
        return -6;
    }
    if (m->n_patterns + n > m->cap) {
        m->cap = (m->n_patterns + n) * 2;
        m->patterns = erealloc(m->patterns, m->cap * sizeof(*m->patterns));
    }
    memcpy(m->patterns + m->n_patterns, tmp, n * sizeof(*tmp));
    m->n_patterns += n;
    efree(tmp);
    return 0;
}

static acmi_php_master *fetch_master(zval *zid)
{
    if (Z_TYPE_P(zid) != IS_RESOURCE || Z_RES_TYPE_P(zid) != le_acmi_master) return NULL;
    return (acmi_php_master *)zend_fetch_resource(Z_RES_P(zid), ACMI_RES_NAME, le_acmi_master);
}

/* One record -> one PHP array, keys in the reference's order (php_ahocorasick.c:562-583). */
static void pack_record(acmi_php_master *m, const acmi_record *r, zval *out)
{
    acmi_php_pattern *p = &m->patterns[r->pattern];
    array_init(out);
    add_assoc_long(out, "pos", r->pos);
    if (p->id_type == ACMI_ID_STRING) { Z_TRY_ADDREF(p->key); add_assoc_zval(out, "key", &p->key); }
    else if (p->id_type == ACMI_ID_NUMBER) add_assoc_long(out, "keyIdx", p->key_id);
    if (!Z_ISUNDEF(p->aux)) { Z_TRY_ADDREF(p->aux); add_assoc_zval(out, "aux", &p->aux); }
    add_assoc_long(out, "start_postion", r->start);     /* sic: the reference's spelling, :578 */
    Z_TRY_ADDREF(p->value);
    add_assoc_zval(out, "value", &p->value);
}

static int warn_status(int rc)
{
    if (rc == ACMI_E_INVALID) php_error_docref(NULL, E_WARNING, "Invalid resource.");
    else if (rc == ACMI_E_NOT_INIT) php_error_docref(NULL, E_WARNING, "Not initialized.");
    else if (rc < 0) php_error_docref(NULL, E_WARNING, "ahocorasick (MI355X engine): %s", acmi_strerror(rc));
    return rc;
}

PHP_FUNCTION(ahocorasick_init)                          /* php_ahocorasick.c:798-834 */
{
    zval *arr;
    acmi_php_master *m;
    ZEND_PARSE_PARAMETERS_START(1, 1) Z_PARAM_ARRAY(arr) ZEND_PARSE_PARAMETERS_END();
    m = ecalloc(1, sizeof(*m));
    if (acmi_create(-1, &m->h) != ACMI_OK) { efree(m); RETURN_FALSE; }
    if (process_patterns(m, Z_ARRVAL_P(arr)) != 0) { master_free(m); RETURN_FALSE; }
    RETURN_RES(zend_register_resource(m, le_acmi_master));
}

PHP_FUNCTION(ahocorasick_add_patterns)                  /* php_ahocorasick.c:882-925 */
{
    zval *zid, *arr;
    acmi_php_master *m;
    ZEND_PARSE_PARAMETERS_START(2, 2) Z_PARAM_RESOURCE(zid) Z_PARAM_ARRAY(arr) ZEND_PARSE_PARAMETERS_END();
    m = fetch_master(zid);
    if (!m || !acmi_is_valid(m->h)) {
        php_error_docref(NULL, E_WARNING, "Cannot add a new pattern, not initialized");
        RETURN_FALSE;
    }
    if (m->finalized) {
        php_error_docref(NULL, E_WARNING, "Cannot add a new pattern to finalized search structure");
        RETURN_FALSE;
    }
    RETURN_BOOL(process_patterns(m, Z_ARRVAL_P(arr)) == 0);
}

PHP_FUNCTION(ahocorasick_finalize)                      /* php_ahocorasick.c:845-876 */
{
    zval *zid;
    acmi_php_master *m;
    ZEND_PARSE_PARAMETERS_START(1, 1) Z_PARAM_RESOURCE(zid) ZEND_PARSE_PARAMETERS_END();
    m = fetch_master(zid);
    if (!m) RETURN_FALSE;
    if (acmi_finalize(m->h) == 1) { m->finalized = 1; RETURN_TRUE; }
    RETURN_FALSE;
}

PHP_FUNCTION(ahocorasick_isValid)                       /* php_ahocorasick.c:623-655 */
{
    zval *zid;
    acmi_php_master *m;
    ZEND_PARSE_PARAMETERS_START(1, 1) Z_PARAM_RESOURCE(zid) ZEND_PARSE_PARAMETERS_END();
    m = fetch_master(zid);
    RETURN_BOOL(m && acmi_is_valid(m->h));
}

PHP_FUNCTION(ahocorasick_deinit)                        /* php_ahocorasick.c:754-791 */
{
    zval *zid;
    ZEND_PARSE_PARAMETERS_START(1, 1) Z_PARAM_RESOURCE(zid) ZEND_PARSE_PARAMETERS_END();
    if (!fetch_master(zid)) RETURN_FALSE;
    zend_list_close(Z_RES_P(zid));                      /* -> master_dtor */
    RETURN_TRUE;
}

PHP_FUNCTION(ahocorasick_match)                         /* php_ahocorasick.c:664-746 */
{
    zend_string *text;
    zval *zid;
    zend_bool find_all = 1;
    acmi_php_master *m;
    const acmi_record *recs = NULL;
    size_t n = 0, i;
    ZEND_PARSE_PARAMETERS_START(2, 3)
        Z_PARAM_STR(text) Z_PARAM_RESOURCE(zid) Z_PARAM_OPTIONAL Z_PARAM_BOOL(find_all)
    ZEND_PARSE_PARAMETERS_END();
    m = fetch_master(zid);
    if (!m) { warn_status(ACMI_E_INVALID); RETURN_FALSE; }
    if (warn_status(acmi_match(m->h, (const uint8_t *)ZSTR_VAL(text), ZSTR_LEN(text), find_all ? 1 : 0,
                               &recs, &n)) < 0) RETURN_FALSE;
    m->finalized = 1;                                   /* lazy finalize, :707 */
    array_init_size(return_value, (uint32_t)n);
    for (i = 0; i < n; i++) {                           /* the drain that replaces the callback */
        zval rec;
        pack_record(m, &recs[i], &rec);
        add_next_index_zval(return_value, &rec);
    }
}

/* NEW: one launch for what examples/benchmark.php:65-67 does with a foreach. */
PHP_FUNCTION(ahocorasick_match_batch)
{
    zval *hays, *zid, *elem;
    zend_bool find_all = 1;
    acmi_php_master *m;
    HashTable *ht;
    uint32_t nh, k = 0;
    uint64_t *off;
    const uint64_t *prefix = NULL;
    const acmi_record *recs = NULL;
    size_t n = 0;
    smart_str blob = {0};
    ZEND_PARSE_PARAMETERS_START(2, 3)
        Z_PARAM_ARRAY(hays) Z_PARAM_RESOURCE(zid) Z_PARAM_OPTIONAL Z_PARAM_BOOL(find_all)
    ZEND_PARSE_PARAMETERS_END();
    m = fetch_master(zid);
    if (!m) { warn_status(ACMI_E_INVALID); RETURN_FALSE; }
    ht = Z_ARRVAL_P(hays);
    nh = zend_hash_num_elements(ht);
    off = ecalloc((size_t)nh + 1, sizeof(uint64_t));
    ZEND_HASH_FOREACH_VAL(ht, elem) {
        zend_string *s = zval_get_string(elem);
        smart_str_append(&blob, s);
        off[k + 1] = off[k] + ZSTR_LEN(s);
        zend_string_release(s);
        k++;
    } ZEND_HASH_FOREACH_END();
    if (warn_status(acmi_match_batch(m->h, blob.s ? (const uint8_t *)ZSTR_VAL(blob.s) : (const uint8_t *)"",
                                     off, nh, find_all ? 1 : 0, &prefix, &recs, &n)) < 0) {
        smart_str_free(&blob); efree(off);
        RETURN_FALSE;
    }
    m->finalized = 1;
    array_init_size(return_value, nh);
    for (k = 0; k < nh; k++) {
        zval one;
        uint64_t i;
        array_init(&one);
        for (i = prefix[k]; i < prefix[k + 1]; i++) {
            zval rec;
            pack_record(m, &recs[i], &rec);
            add_next_index_zval(&one, &rec);
        }
        add_next_index_zval(return_value, &one);
    }
    smart_str_free(&blob);
    efree(off);
}

ZEND_BEGIN_ARG_INFO_EX(arginfo_init, 0, 0, 1) ZEND_ARG_TYPE_INFO(0, data, IS_ARRAY, 0) ZEND_END_ARG_INFO()
ZEND_BEGIN_ARG_INFO_EX(arginfo_res, 0, 0, 1) ZEND_ARG_INFO(0, id) ZEND_END_ARG_INFO()
ZEND_BEGIN_ARG_INFO_EX(arginfo_add, 0, 0, 2) ZEND_ARG_INFO(0, id) ZEND_ARG_TYPE_INFO(0, patterns, IS_ARRAY, 0) ZEND_END_ARG_INFO()
ZEND_BEGIN_ARG_INFO_EX(arginfo_match, 0, 0, 2)
    ZEND_ARG_TYPE_INFO(0, needle, IS_STRING, 0) ZEND_ARG_INFO(0, id) ZEND_ARG_TYPE_INFO(0, findAll, _IS_BOOL, 0)
ZEND_END_ARG_INFO()
ZEND_BEGIN_ARG_INFO_EX(arginfo_batch, 0, 0, 2)
    ZEND_ARG_TYPE_INFO(0, haystacks, IS_ARRAY, 0) ZEND_ARG_INFO(0, id) ZEND_ARG_TYPE_INFO(0, findAll, _IS_BOOL, 0)
ZEND_END_ARG_INFO()

static const zend_function_entry ext_functions[] = {    /* php_ahocorasick_arginfo.h:36-44 + 1 */
    PHP_FE(ahocorasick_match, arginfo_match)
    PHP_FE(ahocorasick_init, arginfo_init)
    PHP_FE(ahocorasick_deinit, arginfo_res)
    PHP_FE(ahocorasick_isValid, arginfo_res)
    PHP_FE(ahocorasick_finalize, arginfo_res)
    PHP_FE(ahocorasick_add_patterns, arginfo_add)
    PHP_FE(ahocorasick_match_batch, arginfo_batch)
    PHP_FE_END
};

PHP_MINIT_FUNCTION(ahocorasick)                         /* php_ahocorasick.c:597-609 */
{
    zend_class_entry ce;
    le_acmi_master = zend_register_list_destructors_ex(master_dtor, NULL, ACMI_RES_NAME, module_number);
    INIT_CLASS_ENTRY(ce, "AhoException", NULL);
    aho_exception_ce = zend_register_internal_class_ex(&ce, zend_ce_exception);
    return SUCCESS;
}

PHP_MINFO_FUNCTION(ahocorasick)
{
    php_info_print_table_start();
    php_info_print_table_header(2, "ahocorasick support", "enabled (MI355X engine, libacmi " ACMI_VERSION ")");
    php_info_print_table_end();
}

zend_module_entry ahocorasick_module_entry = {          /* php_ahocorasick.c:75-86 */
    STANDARD_MODULE_HEADER, "ahocorasick", ext_functions, PHP_MINIT(ahocorasick), NULL, NULL, NULL,
    PHP_MINFO(ahocorasick), PHP_AHOCORASICK_VERSION, STANDARD_MODULE_PROPERTIES
};

#ifdef COMPILE_DL_AHOCORASICK
ZEND_GET_MODULE(ahocorasick)
#endif
