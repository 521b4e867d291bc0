/*
 * acmi.h -- C ABI of libacmi.so, the MI355X-native Aho-Corasick batch matcher that
 * replaces the ac_trie_search() hot path of ph4r05/php_aho_corasick.
 *
 * Plain C, plain pointers and sizes, no torch / HIP types in any signature.
 * Each entry point cites the reference interface it replaces (paths relative
 * to the reference tree).  The inner seam being replaced is the MultiFast C API
 * (src/multifast/ahocorasick.h:73-80) as called from exactly six sites in
 * src/php_ahocorasick.c (:140 finalize, :484 add, :504/:821 release, :745
 * search, :812 create).  INTEGRATION.md shows the Zend-side binding.
 *
 * Threading: a handle is single-caller (the reference's AC_TRIE_t carries
 * mutable scan state, ahocorasick.h:49-65, and PHP is single-threaded);
 * different handles may be used from different threads.
 *
 * Ownership: the library copies pattern bytes at acmi_add_patterns()
 * (the reference deep-copies too: node.c:238-261 with copy=1).  Haystack
 * buffers are borrowed for the duration of a call (php_ahocorasick.c:725-727).
 * Result buffers returned by acmi_match()/acmi_match_batch() are owned by the
 * handle and stay valid until the next match call on that handle or
 * acmi_destroy().
 */
#ifndef ACMI_H
#define ACMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 0.3.0: acmi_tables reshuffled in 0.2 (fields inserted mid-struct) -- a consumer checks acmi_version() against the header it
 * was compiled with; _end completes the OLDEST batch; acmi_comm_* exchange is size-proportional (acmi_comm_debug_compact is
 * gone, acmi_comm_create_local / acmi_comm_get_stat / acmi_comm_last_error are new); acmi_replace_batch_device is new. */
#define ACMI_VERSION "0.3.0"

/* ---- status codes (negative = error; never aborts) ------------------------
 * The Zend shim maps them onto the reference's PHP-level behaviour:
 *   ACMI_E_INVALID   -> E_WARNING "Invalid resource." + false   (php_ahocorasick.c:696-699)
 *   ACMI_E_NOT_INIT  -> E_WARNING "Not initialized."  + false   (:701-704)
 *   ACMI_E_FINALIZED -> E_WARNING + false on add after finalize (:914-917)
 *   device / memory errors -> E_WARNING + false (no reference counterpart).   */
#define ACMI_OK            0
#define ACMI_E_INVALID    (-1)   /* NULL or destroyed handle                      */
#define ACMI_E_NOT_INIT   (-2)   /* handle exists but init_ok != 1                */
#define ACMI_E_FINALIZED  (-3)   /* add_patterns after finalize                   */
#define ACMI_E_ARG        (-4)   /* bad argument (NULL pointer, decreasing offsets, haystack >= 4 GiB) */
#define ACMI_E_NOMEM      (-5)   /* host or device allocation failed              */
#define ACMI_E_DEVICE     (-6)   /* a HIP call failed; acmi_last_error() has text */
#define ACMI_E_NO_DEVICE  (-7)   /* no usable gfx950 device: matching is refused, there is NO CPU fallback */
#define ACMI_E_TOO_LARGE  (-8)   /* automaton or result exceeds a hard limit      */
#define ACMI_E_NO_REPLACEMENT (-9) /* acmi_replace_batch: no pattern carries a replacement text (multifast_replace's -2) */
#define ACMI_E_PARTIAL    (-10)  /* "max_result_bytes" is set and the batch's records exceed it: the result holds the first k
                                  * haystacks COMPLETE (their records, the prefix array's first k + 1 words; *n_records is their
                                  * count), k = acmi_get_stat "resume_haystack" >= 1: match the haystacks from k on as the next
                                  * batch.  The reference hands records to its callback one event at a time and holds none
                                  * (php_ahocorasick.c:542-589); a batch API has to bound what it holds itself. */

/* Pattern identifier kinds: src/multifast/actypes.h:57-62 (enum ac_pattid_type). */
#define ACMI_ID_DEFAULT 0
#define ACMI_ID_NUMBER  1
#define ACMI_ID_STRING  2

/* Longest accepted pattern: src/multifast/actypes.h:148 (AC_PATTRN_MAX_LENGTH). */
#define ACMI_PATTERN_MAX_LENGTH 1024

/* One pattern as handed over by the host shim.  Replaces AC_PATTERN_t
 * (src/multifast/actypes.h:88-94) as filled at php_ahocorasick.c:462-484:
 * `value` = ptext, `id_type`/`id_number` = id, `user` = the aux slot that carries
 * the shim's ahocorasick_pattern_t* (holder of the key / value / aux zvals). */
typedef struct acmi_pattern {
    const uint8_t *value;
    uint32_t length;
    uint32_t id_type;
    int64_t id_number;
    uint64_t user;
} acmi_pattern;

/* One match record.  Replaces what php_ahocorasick_match_handler() reads out of
 * AC_MATCH_t for pattern j of an event (php_ahocorasick.c:555-583):
 *   pos     = m->position                      ("pos", end offset, exclusive)
 *   start   = m->position - strlen(value)      ("start_postion")
 *   pattern = dense caller-order pattern index (0-based over ALL patterns ever
 *             passed to acmi_add_patterns, accepted or not) -> key/keyIdx/aux/value
 *   hay     = haystack index inside the batch (0 for acmi_match).
 * Records of one haystack are in the reference's callback order: pos ascending,
 * and within one pos longest pattern first (node.c:424-441). */
typedef struct acmi_record {
    uint32_t hay;
    uint32_t pos;
    uint32_t start;
    uint32_t pattern;
} acmi_record;

typedef struct acmi_handle acmi_handle;

/* ---- lifecycle ---------------------------------------------------------------- */

/* ac_trie_create() (src/multifast/ahocorasick.c:59-77) + the master struct of
 * PHP_FUNCTION(ahocorasick_init) (php_ahocorasick.c:811-816, init_ok=1 at :827).
 * `device` = HIP device ordinal, or -1 for the current device.  Creating a
 * handle does not need a GPU; the device is first touched by a match call. */
int acmi_create(int device, acmi_handle **out);

/* php_ahocorasick_process_patterns() insertion loop (php_ahocorasick.c:455-486)
 * + ac_trie_add() (ahocorasick.c:91-131).  One call == one ahocorasick_init() /
 * ahocorasick_add_patterns() call: the n patterns get indices base..base+n-1 in
 * CALLER order and are inserted in REVERSE caller order; empty, over-long and
 * duplicate patterns are silently dropped (first inserted wins, so within one
 * call the LAST duplicate in caller order wins).  *accepted (optional) receives
 * the number inserted.  Returns ACMI_OK, ACMI_E_FINALIZED, ... */
int acmi_add_patterns(acmi_handle *h, const acmi_pattern *patterns, size_t n, size_t *accepted);

/* php_ahocorasick_finalize() (php_ahocorasick.c:130-142) + ac_trie_finalize()
 * (ahocorasick.c:143-155).  Builds the flat automaton.  Returns 1 the first
 * time, 0 when already finalized (idempotent), <0 on error. */
int acmi_finalize(acmi_handle *h);

/* PHP_FUNCTION(ahocorasick_isValid) (php_ahocorasick.c:623-655): 1 / 0. */
int acmi_is_valid(const acmi_handle *h);

/* PHP_FUNCTION(ahocorasick_deinit) + resource dtor (php_ahocorasick.c:754-791,
 * :494-512) + ac_trie_release() (ahocorasick.c:288-296).  Frees host and device
 * memory.  Returns ACMI_OK, or ACMI_E_INVALID for NULL. */
int acmi_destroy(acmi_handle *h);

/* ---- matching ----------------------------------------------------------------- */

/* PHP_FUNCTION(ahocorasick_match) (php_ahocorasick.c:664-746): one haystack,
 * keep=0 (every call starts at the root, tests/test6.phpt), lazy finalize.
 * find_all == 0 reproduces findAll=false: all records of the FIRST event only
 * (php_ahocorasick.c:555-588).  *records / *n_records: handle-owned result. */
int acmi_match(acmi_handle *h, const uint8_t *text, size_t length, int find_all,
               const acmi_record **records, size_t *n_records);

/* Streaming search: MultiFast's keep=1 mode (ac_trie_search(..., keep=1, ...), ahocorasick.c:186-194,
 * 220, 236-238: the next call continues in the node and at the stream offset the previous one ended).
 * The PHP extension never uses it (keep=0 at php_ahocorasick.c:745); it is what a caller scanning one
 * large input in pieces needs.  A stream carries the last max_pattern_length-1 bytes it was fed; every
 * feed scans [carry | data] from the root and reports the events that END in `data` -- the same events,
 * in the same order, that keep=1 reports, with pos / start counted from the start of the stream (hay = 0).
 * Every event is reported (the findAll=false stop does not save the position in the reference either).
 * Streams of 4 GiB or more are refused (ACMI_E_TOO_LARGE): record positions are 32-bit.
 * *records / *n_records stay valid until the next feed or close of the same stream. */
typedef struct acmi_stream acmi_stream;
int acmi_stream_open(acmi_handle *h, acmi_stream **out);
int acmi_stream_feed(acmi_handle *h, acmi_stream *s, const uint8_t *data, size_t length,
                     const acmi_record **records, size_t *n_records);
uint64_t acmi_stream_position(const acmi_stream *s);   /* bytes fed so far */
void acmi_stream_close(acmi_stream *s);

/* Batch form of acmi_match (NEW: the reference loops in PHP userland,
 * examples/benchmark.php:65-67).  `blob` holds the haystacks back to back,
 * `offsets[n_haystacks+1]` their byte offsets (non-decreasing).  On return
 * (*prefix)[k] .. (*prefix)[k+1] delimit haystack k's records in *records. */
int acmi_match_batch(acmi_handle *h, const uint8_t *blob, const uint64_t *offsets,
                     size_t n_haystacks, int find_all, const uint64_t **prefix,
                     const acmi_record **records, size_t *n_records);

/* Same, with blob and offsets ALREADY RESIDENT in device memory (HBM) and the
 * results left there.  `d_blob` should be 16-byte aligned; `blob_bytes` is the
 * size of the blob (== offsets[n_haystacks]).  *n_records receives the total. */
int acmi_match_batch_device(acmi_handle *h, const void *d_blob, uint64_t blob_bytes,
                            const void *d_offsets, size_t n_haystacks, int find_all,
                            uint64_t *n_records);

/* The same call split in two, for callers that overlap host work (or a collective) with the
 * scan: _begin enqueues the whole pipeline on the handle's stream and returns at once; _end waits
 * for it (the pipeline's only host synchronisation) and reports the record count.  Up to TWO
 * batches may be in flight: a second _begin before the first _end queues its kernels behind the
 * first batch's, so the device does not idle while the host turns around (a third _begin returns
 * ACMI_E_ARG).  _end completes the OLDEST batch in flight; one _end per _begin; a batch's input
 * buffers must stay untouched until its _end.  The result acmi_result_* refer to is that of the
 * last _end; its device buffers stay valid until the second _begin after its own.  A synchronous
 * call (acmi_match*, acmi_match_batch_device) drops batches that were begun and never ended. */
int acmi_match_batch_device_begin(acmi_handle *h, const void *d_blob, uint64_t blob_bytes,
                                  const void *d_offsets, size_t n_haystacks, int find_all);
/* The same for a UNIFORM batch -- n_haystacks haystacks of haystack_bytes each, back to back from d_blob (the shape of
 * examples/benchmark.php's buffers: 256 x 8,192): no offsets array is needed, the library keeps its own, and the work-item table
 * is written in closed form instead of being counted and scanned (19 -> 4 us per 2^19 haystacks) -- and kept: a batch of the
 * shape its slot saw last (same count, length, alignment of d_blob modulo 128) through the anchor filter launches nothing for it
 * (acmi_get_stat "item_tables_kept").  Ended by
 * acmi_match_batch_device_end like any other begun batch; results and lifetimes are the same. */
int acmi_match_batch_device_begin_uniform(acmi_handle *h, const void *d_blob, uint64_t haystack_bytes, size_t n_haystacks, int find_all);
int acmi_match_batch_device_end(acmi_handle *h, uint64_t *n_records);

/* Device pointers of the last result: prefix = uint64[n_haystacks+1],
 * records = acmi_record[n_records].  The memory stays valid until the SECOND next match call on the
 * handle (results alternate between two buffers), so a caller may enqueue the next batch first and ship
 * this result while it runs. */
int acmi_result_device(acmi_handle *h, const void **d_prefix, const void **d_records,
                       uint64_t *n_records);

/* Drain the last device-resident result into caller HOST memory (either may be NULL):
 * the "device match-buffer drain" that replaces the per-event callback
 * (php_ahocorasick.c:542-589). */
int acmi_result_to_host(acmi_handle *h, uint64_t *prefix, acmi_record *records);

/* Copy the last result into caller-owned DEVICE buffers (either may be NULL), e.g. the
 * send buffers of the multi-GPU all-gather.  Ordered on the handle's stream and
 * synchronized before returning. */
int acmi_result_to_device(acmi_handle *h, void *d_prefix_dst, void *d_records_dst);

/* ---- search and replace (MultiFast's multifast_replace; compiled into the reference, config.m4:9, but not exposed
 * by its PHP layer: rtext is always NULL at php_ahocorasick.c:469-470) ---------------------------------------------- */

#define ACMI_REPLACE_NORMAL 1   /* MF_REPLACE_MODE_NORMAL (replace.h:34-46; 0 = DEFAULT behaves the same): a pattern swallows
                                   the factors found inside it; other overlapping patterns are all replaced */
#define ACMI_REPLACE_LAZY   2   /* MF_REPLACE_MODE_LAZY: whichever pattern comes first is replaced, what overlaps it is dropped */

/* AC_PATTERN_t.rtext (actypes.h:88-94) of caller-order pattern `pattern_index`, set between the acmi_add_patterns()
 * call that brought the pattern and finalize.  The text is kept as the reference keeps it (node.c:248-251 through
 * mpool_strndup, mpool.c:165-178): copied up to its first NUL byte, zero-filled from there to `length`.  text == NULL:
 * the pattern is matched but not replaced. */
int acmi_set_replacement(acmi_handle *h, uint32_t pattern_index, const uint8_t *text, uint32_t length);

/* multifast_replace() + multifast_rep_flush(keep=0) (replace.c:469-568) over every haystack of a batch: per event the
 * longest pattern with a replacement text is the nominee (node.c:337-362), nominees are kept or dropped by `mode`
 * (replace.c:222-272), and each output text is the input with the kept nominees replaced left to right (:402-455).
 * *out_blob holds the output texts back to back, (*out_offsets)[n_haystacks + 1] their offsets; both are owned by the
 * handle until its next call.  ACMI_E_NO_REPLACEMENT when no accepted pattern has a replacement text. */
int acmi_replace_batch(acmi_handle *h, const uint8_t *blob, const uint64_t *offsets, size_t n_haystacks, int mode,
                       const uint8_t **out_blob, const uint64_t **out_offsets);

/* The same with input and output RESIDENT in device memory: `d_blob` / `d_offsets` as for acmi_match_batch_device; on return
 * *d_out_blob (device) holds the *out_bytes bytes of output text and *d_out_offsets (device) the uint64[n_haystacks + 1]
 * offsets, both owned by the handle until its next call.  One host wait for the matching, one for the whole replace pass
 * (nominees, a device prefix sum of the output lengths, the copy cut by 32 KiB of OUTPUT per wave: one huge haystack is
 * spread over the chip). */
int acmi_replace_batch_device(acmi_handle *h, const void *d_blob, uint64_t blob_bytes, const void *d_offsets, size_t n_haystacks,
                              int mode, const void **d_out_blob, const void **d_out_offsets, uint64_t *out_bytes);

/* ---- multi-GPU: a batch sharded over the GPUs of one node (NEW; the reference is single-threaded) --------------
 * Haystacks are independent (keep=0, php_ahocorasick.c:745; tests/test6.phpt), so a batch is cut into contiguous
 * haystack ranges, one per rank (one process or thread per GPU, each with its own handle built from the same
 * patterns).  Every rank scans its range with acmi_match_batch_device(); the one exchange step is the all-gatherv
 * that concatenates the per-rank match arrays in rank order -- the reference's global order, because the ranges
 * are contiguous.  The collective is RCCL's ncclAllGather over xGMI, bound at run time from librccl.so (no
 * link-time dependency; a process that already has torch's copy loaded reuses it). */
typedef struct acmi_comm acmi_comm;
typedef struct acmi_comm_id { char internal[128]; } acmi_comm_id;   /* ncclUniqueId */

/* Rank 0 makes an id and hands it to the other ranks out of band (a file, a socket, MPI, torch.distributed). */
int acmi_comm_get_id(acmi_comm_id *id);
/* Joins rank `rank` of `world` (<= 64) on the handle's device; world == 1 needs no id and no RCCL. */
int acmi_comm_create(acmi_handle *h, int rank, int world, const acmi_comm_id *id, acmi_comm **out);
/* The same communicators for ONE process that drives several handles (one per GPU, or several on one GPU) from one thread
 * each: out[r] is rank r's, on handles[r]'s device.  No RCCL: ranks meet in process memory and copy device to device.
 * Every acmi_comm_allgather_result is a rendezvous of all `world` threads.  (Also how the W > 1 logic is tested on a box
 * with one GPU, where RCCL refuses several ranks per device.) */
int acmi_comm_create_local(acmi_handle *const *handles, int world, acmi_comm **out);
int acmi_comm_destroy(acmi_comm *c);
/* After acmi_match_batch_device() on this rank's `n_haystacks_local` haystacks: all-gatherv of the results.  On
 * return every rank holds the GLOBAL arrays in device memory owned by the communicator (valid until its next call):
 * prefix = uint64[*n_haystacks + 1], records = acmi_record[*n_records] with hay counted over all ranks.
 * Collectives: 32 bytes per rank of sizes and status (the call's one host wait), then ONE group of per-rank broadcasts,
 * each exactly its shard's size, received in place -- nothing is padded to the largest shard or staged.  A rank
 * that cannot take part (no result for the count it gave, out of memory) reports it in the size round, which it still
 * joins: every rank then returns an error (ACMI_E_ARG / ACMI_E_NOMEM on the rank at fault, ACMI_E_DEVICE on its peers)
 * instead of waiting inside a collective.  When the global arrays have to grow, a second 32-byte round confirms that every
 * rank could allocate.  EXPERIMENTAL for world > 1 over RCCL until it has run on a multi-GPU node (the build box has one
 * GPU); the process-local transport runs the same code on one. */
int acmi_comm_allgather_result(acmi_comm *c, acmi_handle *h, uint64_t n_haystacks_local, const void **d_prefix,
                               const void **d_records, uint64_t *n_records, uint64_t *n_haystacks);
/* On the wire a record is 8 bytes {pos, pattern}, not 16: its haystack follows from the prefix array (which travels anyway)
 * and its start from the pattern's length (start = pos - length, php_ahocorasick.c:578); the receiver expands what arrives
 * ("bytes_sent" = 8 per record + 8 per haystack of this rank's shard).
 * acmi_comm_gather_result: the same exchange with ONE receiver -- the PHP caller lives on one host, and for dense results
 * (BASELINE configs[4]: gigabytes of records per rank) seven of eight copies of everything are wasted links: every rank sends
 * its shard to `root` (ncclSend / ncclRecv in one group), only `root` gets the global arrays (the others *d_prefix =
 * *d_records = NULL; *n_records / *n_haystacks are the global totals everywhere). */
int acmi_comm_gather_result(acmi_comm *c, acmi_handle *h, uint64_t n_haystacks_local, int root, const void **d_prefix,
                            const void **d_records, uint64_t *n_records, uint64_t *n_haystacks);
/* The gathered arrays of the last exchange copied to host memory (prefix: *n_haystacks + 1 words; records: *n_records): for
 * hosts that do not speak HIP themselves.  Either pointer may be NULL. */
int acmi_comm_result_to_host(acmi_comm *c, uint64_t *prefix, acmi_record *records);
/* "world", "rank", "backend_ranks" (what ncclCommCount reports), "transport" (0 none, 1 RCCL, 2 process-local), and of the
 * last exchange: "rounds" (collectives), "bytes_sent", "bytes_received", "records", "haystacks". */
int acmi_comm_get_stat(const acmi_comm *c, const char *key, double *value);
const char *acmi_comm_last_error(const acmi_comm *c);

/* ---- pattern metadata for the shim's zval pack (php_ahocorasick.c:562-583) ---- */

size_t acmi_pattern_count(const acmi_handle *h);
/* Fills *out for caller-order pattern `index` (value pointer = library copy). */
int acmi_pattern_info(const acmi_handle *h, uint32_t index, acmi_pattern *out);

/* ---- knobs, statistics, diagnostics ---------------------------------------------- */

/* Integer options, set before finalize unless noted:
 *   "lds_bytes"      LDS budget for the hot-state tile (default 160 KiB minus slack)
 *   "slab_events"    per work-item event slots before the overflow path (default: learned -- 16, or 32 with the dense anchor form, quadrupled after a batch that sent more than 1 item in 64 to the drain, up to 256 per item, half the batch's bytes and 2 GiB in all)
 *   "run_cap"        per work-item fix-up runs before the item is rescanned whole (default: learned -- 16, quadrupled up to 64 after a batch whose filter lost to overflowed run lists; the dense anchor form: 256 entries of 4 bytes)
 *   "chunk_bytes"    force the work-item size (0 = automatic)            [any time]
 *   "kernel"         0 = plain global-table walk, 1 = LDS-tile table walk, 2 = anchor q-gram
 *                    filter + exact table walk of the candidates (needs a shortest pattern
 *                    >= 7 B and a collision-free table seed exists, else falls back to 1), 3 = automatic
 *                    (default): 2 when eligible -- 1 for batches of haystacks under 192 B on average --
 *                    and back to 1 for a while if a batch shows the text defeats the filter (the fast
 *                    kernel is probed again after 16, 32, ... batches); from 1 on to 4 if a batch shows
 *                    that its optimistic pass flags most bytes anyway; 4 = exact table walk of whole
 *                    work items only (keyword-like and dense-match text); 6 = the tile walk: the same exact
 *                    walk with a work item spread over the lanes of a wave and the records placed by a running
 *                    sum in ONE pass -- no slabs, no drain (findAll = true, patterns of <= 129 B; what automatic
 *                    mode takes instead of 4 unless the output is denser than a record per two bytes) [any time]
 *   "tile_w", "tile_p"  kernel 6: bytes per lane (16, 32, 64; 128: k_walk_chain only, else taken as 64) and lanes per work item
 *                    (1..64); 0 = automatic [any time]
 *   "chain_walk"     kernel 6 for a pattern set that carries the chain-compressed LDS image of its WHOLE automaton (a few thousand
 *                    patterns: dense rows for the branching states, goto edge + failure pointer -- the reference's node,
 *                    node.h:37-67 -- for the insides of the patterns; "chain_image_bytes", "chain_dense_states"): -1 (default) the
 *                    walk over that image (k_walk_chain + k_chain_emit) unless the automaton is so small that all of its dense
 *                    rows fit the LDS, 1 whenever there is an image, 0 never ("chain_walk_used", "chain_log_reruns").  The
 *                    kernel's own LDS layout adds 4 bytes per branching state to the image and addresses rows by 16 bits: an
 *                    image whose rows end beyond 128 KiB, or that no longer fits 160 KiB with those bytes, is walked by
 *                    k_walk_tiles as if there were none                                                     [any time]
 *   "chain_bytes"    budget of that image, <= 160 KiB - 768 B (the default); 0: none is built             [before finalize]
 *   "reserve_cus"    N (default 0): the scan kernels run as persistent workgroups on all but N compute units, so that the kernels
 *                    of a collective in flight (acmi_comm_*, RCCL) find CUs and LDS while a scan runs          [any time]
 *   "max_result_bytes"  0 (default): the result buffers grow to whatever a batch produces (16 B per record; a^1..a^16 over
 *                    1 GiB: 275 GB -- ACMI_E_NOMEM); else the device-resident batch entries return ACMI_E_PARTIAL with the
 *                    first haystacks whose records fit (see there), never allocating more.  A single haystack whose own
 *                    records exceed the cap fails with ACMI_E_TOO_LARGE.  >= 4096.  A split set (see "split") bounds each
 *                    of its two match arrays and the merged one alike.  acmi_replace_batch(_device) needs every match of
 *                    the batch at once: under a bound its records exceed it fails with ACMI_E_PARTIAL and no output --
 *                    pass smaller batches                                                        [any time]
 *   "part_kernel"    split sets (see "split"): the "kernel" of the SHORT part alone, -1 (default) its own automatic choice; with 6
 *                    and a small short part the long part's records are merged into the result as the short part's are
 *                    written -- no merge pass                                                    [any time]
 *   "tiny_rows"      kernel 2: 1 (default) batches of haystacks of <= 1 KiB on average take the variant whose
 *                    waves pack one-row work items eight to a step, 0 never                [any time]
 *   "short_rows"     kernel 2: the variant that skips the 1-KiB rows a work item lacks: -1 (default)
 *                    unless every haystack is a multiple of 8 KiB on an aligned blob, 0 never, 1 always [any time]
 *   "walk_group"     kernel 4: 16-byte haystack blocks a lane requests back to back.  Always 4 since round 4 (half a
 *                    line, the most that fits 8 waves per SIMD: the variants with 1 and 8 measured 1.2-1.4 x slower
 *                    and were most of the library's compile time); 1, 4, 8 are accepted and ignored  [any time]
 *   "ilp"            streams per lane in the LDS-tile kernel (1 or 2)    [any time]
 *   "group_quads"    16-byte loads per lane per fetch group, 4 or 8 (ILP 1) [any time]
 *   "small_path"     1 (default): host batches of <= 64 KiB in <= 1024 haystacks (every acmi_match of a few
 *                    KiB) take one launch that reads / writes pinned host memory; 0 = always the pipeline
 *   "wave_items"     anchor kernel: items a wave takes per batch, 1..64 (0 = automatic)
 *   "persistent"     1 = one workgroup per CU, waves pull item batches from a counter; 0 = one
 *                    workgroup per 1024 x ILP items; -1 = automatic (default)   [any time]
 *   "compact"        automaton form for the states outside the first trie levels: dense failure-free rows
 *                    (one lookup per byte, 1 KiB per state at 256 byte classes) or 8-byte goto / failure records
 *                    laid out along the patterns (the reference's node form, node.h:37-67, flattened): 1 = compact,
 *                    0 = dense, -1 (default) = compact when the dense table would exceed 64 MiB   [before finalize]
 *   "anchor_dense"   pattern sets whose shortest pattern is 4..6 B (keyword lists): the anchor filter in its dense form -- every
 *                    pattern anchored by its first four bytes, the window tested at every byte offset of the text: -1 (default)
 *                    when at most ~ 160 windows per 8 KiB are expected to hit on text that looks like the patterns, 0 never,
 *                    1 whenever the shortest pattern allows                                         [before finalize]
 *   "anchor_slots2"  dense anchor form: two-slot buckets (2^15 of them) when the keys are at most 3,072 -- one 4-byte LDS read and one
 *                    packed subtract per window: -1 / 1 (default) yes, 0 always four slots                [before finalize]
 *   "dense_hot"      dense anchor form: the scan variant that works out which windows hit in every step, branch-free (+ 10 VALU per
 *                    dword) instead of behind a rarely-taken branch: -1 (default) when a dozen or more hit windows are expected
 *                    per 8 KiB, 0 never, 1 always                                                  [before the first match]
 *   "anchor_choice"  big pattern sets (anchor mode 2): which aligned window of a pattern is its anchor is chosen so that the
 *                    anchors share bits of the LDS bitmap (greedy set cover: 9 % of the bits set instead of 22 % for 65,536
 *                    patterns, as many fewer probes of the fingerprint table): -1 / 1 = yes (default), 0 = always the
 *                    pattern's first aligned window                                               [before finalize]
 *   "dense_bytes"    compact: dense rows are kept for the LDS tile, trie depth <= 1 and further states in BFS order
 *                    up to this many bytes (default 4 MiB)                                        [before finalize]
 *   "verify"         1 (default): a candidate of the anchor filter whose q-gram belongs to exactly one (pattern, offset)
 *                    pair is decided by comparing that pattern's text with the haystack; 0: every candidate is walked
 *                    through the automaton.  Same records either way                              [before the first match]
 *   "device_build"   the level-synchronous part of the build -- failure links (ahocorasick.c:344-368) and per-state outputs
 *                    (node.c:424-441), one kernel launch per trie depth -- on the device: 1 = always (when a gfx950 device
 *                    is there), 0 = never, -1 (default) = from 200,000 trie nodes.  Same tables either way ("built_on_device")
 *                                                                                                  [before finalize]
 *   "keep_host_tables"  1 = keep the host copies of the tables after they were uploaded (acmi_debug_tables after a
 *                    match call); default 0: they are freed by the first match call               [before the first match]
 *   "max_flat_outputs"  per-state output lists are also stored flat ({pattern, length}, longest first) for the
 *                    dense-match drain when they total at most this many entries (default 2^25);
 *                    0 = dictionary links only                                  [before finalize]
 *   "early_stop"     exact walks of single-window anchor candidates end once the state is a non-final state of
 *                    trie depth <= 2 two bytes past the window start: 0 = off, 1 (default) = when some 3-gram
 *                    over the patterns' first-byte alphabet is not a trie path (else it cannot fire on
 *                    in-alphabet text), 2 = always                                        [any time]
 *   "win_table"      the two-byte window table of the whole-item walk (kernel 4; k_walk_win): trie depth <= 2 decided in LDS from the last
 *                    two text bytes, for sets whose shortest pattern is >= 3 B: -1 (default) for compact automata (sets far too
 *                    big for the LDS), 0 never, 1 whenever the shortest pattern allows ("win_table")   [before finalize]
 *   "win_walk"       0: kernel 4 walks the tables (k_walk_items) although the set has the window table; 1 (default)   [any time]
 *   "relist"         1 (default): a batch in which a work item's list of candidate windows overflowed while lists can still grow is
 *                    run again with the lists (and slabs, item size) the scan's own count asks for, instead of walking such items
 *                    whole through the tables; 0 = walk them at once ("relist_reruns")                    [any time]
 *   "wave_runs"      work items with more than three candidate windows of the anchor filter are decided a wave per item, a lane
 *                    per window (k_runs_wave, enqueued while batches report such items): 1 (default) yes, 0 = one lane per
 *                    item whatever it holds.  Same records either way ("busy_items", "max_windows", "serial_items",
 *                    "wave_runs_used")                                                          [any time]
 *   "refill"         drain of work items whose events overflowed their slab (dense-match inputs):
 *                    1 = one thread per item, 2 = one wave per item with whole-row record stores,
 *                    0 = automatic (default): batches scanned by the whole-item walk (items of <= 1 KiB) are drained by the
 *                    same walk writing records, a thread per item, unless the output is dense (a record per two bytes of
 *                    text or more); else 2 unless the longest pattern exceeds 64 B   [any time]
 *   "items_ahead"    1: a batch's work-item table (k_items: it needs the offsets only) is built on a second stream of the handle,
 *                    behind the scan kernel of the batch begun before it, next to that batch's candidate walk and expansion,
 *                    and the batch's pipeline waits for it.  (Measured on the headline batch: no gain -- the small kernels do
 *                    not overlap in practice; kept for experiments.)  The caller vouches that a batch's offsets and blob are COMPLETE
 *                    in device memory when it is begun (not pending on the stream).  0 (default): in stream order   [any time]
 *   "split"          a set that is not eligible for the anchor filter only because of a FEW patterns under 7 B is matched as two
 *                    automata over the same batch -- the long patterns through the filter, the rest through the table walk -- and
 *                    the two match arrays are merged per haystack (position ascending, longer pattern first): -1 (default) when
 *                    the short patterns are at most 256 and a sixteenth of the long ones, 0 never, 1 whenever both kinds exist.
 *                    Same records either way ("split", "split_long_patterns", "split_short_patterns")   [before finalize]
 *   "profile"        1 = bracket the scan kernel with HIP events ("scan_ms"); 2 = also the candidate walk ("walk_ms")
 *                    and the whole pipeline ("total_ms") -- every event is a barrier on the stream   [any time] */
int acmi_set_option(acmi_handle *h, const char *key, int64_t value);

/* Numeric facts about the automaton and the last match call, e.g. "states",
 * "hot_states", "columns", "entry_bytes", "lds_image_bytes", "table_bytes",
 * "max_pattern_length", "items", "overflow_items", "walked_bytes", "fallback_bytes", "kernel_used" (0, 1, 2, 4 as the
 * "kernel" option; 5 = the one-launch small-batch path), "busy_items" / "max_windows" / "serial_items" / "wave_runs_used" /
 * "relist_reruns" / "anchor_chunk_shift" (the filter's hand-over on the last batch: work items with more than a few candidate
 * windows, the longest window list, items k_runs_wave handed to one lane, whether it ran, batches run again for longer lists,
 * the learned shrink of the filter's work items), "win_table" (1: the set has the two-byte window table), "anchor_variant" (the k_scan_anchor template the last batch ran:
 * window dwords in bits 0-2 | 8 big set (LDS bitmap + L2 table) | 16 short-row variant | 32 tiny-item variant | 64 windows
 * at every byte offset | 128 its many-hits variant | 256 two-slot buckets; 0 = another kernel), "tiny_waves" (wave batches that took the tiny-item path), "scan_ms",
 * "total_ms". */
int acmi_get_stat(const acmi_handle *h, const char *key, double *value);

/* Run subsequent launches on this hipStream_t (passed as void*); NULL = the
 * handle's own stream.  ACMI_E_ARG while a begun batch has not been ended. */
int acmi_set_stream(acmi_handle *h, void *hip_stream);

/* Flat host tables of the finalized automaton, for CPU-side inspection by
 * tests (no device needed).  Pointers are handle-owned. */
typedef struct acmi_tables {
    uint32_t n_states;       /* incl. the trap row                             */
    uint32_t n_columns;      /* byte classes per row                           */
    uint32_t entry_bytes;    /* 2 or 4                                         */
    uint32_t first_hot_final;/* F1: states [F1, trap) are hot and final        */
    uint32_t trap;           /* T : states [0, T) are LDS-resident             */
    uint32_t first_cold_final;/* F2: states [F2, n_states) are cold and final  */
    uint32_t class_mode;     /* 0 = table lookup, 1 = 3-bit window (SWAR)      */
    uint32_t swar_shift;
    uint32_t swar_other;
    uint32_t max_pattern_length;
    const void *table;       /* n_states (compact: n_dense_rows) x n_columns transitions */
    const void *lds_image;   /* (trap+1) x n_columns, cold targets -> trap     */
    const uint8_t *class_of_byte; /* 256 entries                               */
    const uint32_t *own_pattern;  /* per state, 0xFFFFFFFF = none              */
    const uint32_t *dict_link;    /* per state, 0xFFFFFFFF = none              */
    const uint32_t *out_count;    /* per state                                 */
    uint32_t anchor_q;            /* anchor filter window in bytes, 0 = not eligible */
    uint32_t anchor_mul;          /* hash multiplier                           */
    uint32_t min_pattern_length;
    uint32_t anchor_keys;
    const uint16_t *anchor_table; /* mode 1: 16384 buckets x 4 fingerprints; mode 2: 2^bucket_bits x 8 */
    uint32_t anchor_mode;         /* 0 none, 1 LDS table, 2 LDS bitmap + global table */
    uint32_t anchor_bucket_bits;  /* mode 2                                    */
    uint32_t anchor_bucket_shift; /* mode 2: bucket = (hash >> shift) & (2^bits - 1) */
    const uint32_t *anchor_bitmap;/* mode 2: 32768 words (2^20 bits)           */
    uint32_t anchor_choice;       /* 1: anchor windows chosen among the aligned windows of a pattern (mode 2) */
    const uint8_t *anchor_offsets;/* [pattern * 4 + a]: offset of the anchor window when the first aligned text dword lies a bytes into the pattern */
    uint32_t n_flat_outputs;      /* entries of out_list (0: dictionary links only) */
    uint32_t compact;             /* 1: `table` holds n_dense_rows rows only, for ids [0, dense_nf_end) and [first_cold_final, dense_f_end) */
    uint32_t dense_nf_end, dense_f_end, n_dense_rows, n_cold_edges;
    const uint32_t *cold_rec;     /* compact: 2 words per state id {x, failure id | code << 24}; code 0 no child, 1: x = child << 8 | byte,
                                     2..254: that many edges at cold_edges[x], 255: cold_edges[x] = count, edges behind it */
    const uint32_t *cold_edges;   /* child << 8 | byte, ascending byte */
    uint32_t chain_lo;            /* compact: ids [chain_lo, first_cold_final) are single-child states numbered along their patterns */
    const uint8_t *chain_byte;    /* per such state: its edge byte (null: none) */
    const uint8_t *chain_run;     /* per such state: how many steps from it on lead to id + 1, again such a state (capped at 255) */
    const uint32_t *out_off;      /* per state: first entry of its flat output list */
    const uint32_t *out_list;     /* 2 words per entry: pattern index, pattern length; longest first */
    uint32_t anchor_slots;        /* slots per bucket of the mode-1 table: 4 (2^14 buckets, hash bits 3..16) or 2 (2^15 buckets, bits 2..16) */
    uint32_t anchor_dense;        /* 1: the filter tests the 4-byte window at EVERY byte offset against the patterns' first four bytes
                                     (shortest pattern 4..6 B); anchor_q = 4, anchor_mode = 1, one key per pattern */
    /* chain-compressed image of the whole automaton for the LDS (null: the set does not fit, or fits there as dense rows):
       rows of the D states (index < chain_dense: n_columns x uint16, target index | 0x8000 when final) at chain_image, 4-byte
       records of the C states at chain_image + chain_rec_off {index of the failure state (a D state) | 2 x class of the edge to
       index + 1 (0xFF: none) << 16 | min(records of that child, 31) << 24 | child is final << 31}, min(records, 255) of the indices
       below chain_cnt_n at chain_image + chain_cnt_off; chain_orig[index] = state id, chain_head = out_head's form per index */
    const uint8_t *chain_image;
    uint32_t chain_bytes, chain_dense, chain_index, chain_rec_off, chain_cnt_off, chain_cnt_n;
    const uint32_t *chain_orig;
    const uint32_t *chain_head;
    /* the two-byte window table of the whole-item walk (kernel 4 on sets whose shortest pattern is >= 3 B; null: none): win_table
       = 65,536 16-bit cells indexed by a << 8 | b, then win_list_bytes / 2 words of child lists -- cell 0: no node (a, b) or no
       child; 0x8000 | c: one child, byte c; 0x4000 | f1 << 7 | f2: two children by 7-bit fingerprints (c ^ c >> 3 ^ c >> 5) & 127;
       0xC000 | u: three or more, their list {count, bytes ...} at byte 2 u of the list area; 0xFFFF: always "maybe".
       win_state[a << 8 | b] = the state after "ab" read from the root.  win_hash: the trie nodes of depth 3 by their three bytes,
       2^win_hash_bits slots of {(key + 1) | bit 7 of the child byte << 31, state | simple << 31 | low 7 bits of the child byte << 24}
       ("simple": one child, not final, failure link of depth <= 2), slot = key * 0x9E3779B1 >> (32 - bits), linear probing, 0 = empty. */
    const uint16_t *win_table;
    const uint32_t *win_state;
    const uint32_t *win_hash;
    uint32_t win_hash_bits, win_list_bytes;
    uint32_t shallow_hot, shallow_cold_end;   /* states of trie depth <= 2 that are not final: ids [0, shallow_hot) and (trap, shallow_cold_end) */
} acmi_tables;
int acmi_debug_tables(const acmi_handle *h, acmi_tables *out);

/* Measurement aid, not on the product path: time a plain coalesced streaming read of `bytes`
 * of device memory (16 B per lane), `repeats` passes.  Gives the HBM read bandwidth a kernel can
 * actually reach on this chip and a known byte count to calibrate FETCH_SIZE against. */
int acmi_debug_stream_read(acmi_handle *h, const void *d_src, uint64_t bytes, int repeats, double *ms_per_pass);

const char *acmi_strerror(int status);
const char *acmi_last_error(const acmi_handle *h);
const char *acmi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ACMI_H */
