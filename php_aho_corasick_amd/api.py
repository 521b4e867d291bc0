"""Host-side mirror of the reference's PHP function surface, on top of the C ABI.

PHP is not installed in this image, so the Zend shim (``php_ext/``) cannot be
compiled or run here.  This module plays the shim's role in Python: same function names,
argument meaning and error behaviour as ``src/php_ahocorasick.c``, with

* PHP arrays            -> Python ``list`` / ``dict`` (insertion-ordered),
* PHP strings           -> ``str`` (encoded as UTF-8) or ``bytes``; offsets are BYTE offsets,
* ``E_WARNING``         -> ``warnings.warn(..., AhoWarning)`` and the same ``False`` return,
* ``AhoException``      -> ``AhoException``,
* the resource          -> an ``AhoResource`` object.

Result records are dicts whose keys appear in the reference's order
(php_ahocorasick.c:562-583): ``pos``, then ``key`` | ``keyIdx`` | nothing, then ``aux`` if the
pattern has one, then ``start_postion`` (sic, the reference's spelling), then ``value``.

All matching goes through libacmi.so's HIP kernels; nothing here scans text on the CPU.
"""
from __future__ import annotations

import ctypes as C
import warnings
from typing import Any

import numpy as np

from . import _native as N

__all__ = ["AhoException", "AhoWarning", "AhoResource", "ahocorasick_init", "ahocorasick_add_patterns",
           "ahocorasick_finalize", "ahocorasick_match", "ahocorasick_match_batch", "ahocorasick_isValid",
           "ahocorasick_deinit"]


class AhoException(Exception):
    """``AhoException extends Exception`` (php_ahocorasick.c:603-604)."""


class AhoWarning(UserWarning):
    """Stands in for PHP's E_WARNING raised through php_error_docref."""


def _warn(msg: str) -> None:
    warnings.warn(msg, AhoWarning, stacklevel=3)


def _php_type(v: Any) -> str:
    """php_aho_type_str (php_ahocorasick.c:98-117)."""
    if v is None:
        return "null"
    if v is True:
        return "true"
    if v is False:
        return "false"
    if isinstance(v, int):
        return "long"
    if isinstance(v, float):
        return "double"
    if isinstance(v, (str, bytes)):
        return "string"
    if isinstance(v, (list, tuple, dict)):
        return "array"
    return "object"


def _to_bytes(v) -> bytes:
    return v.encode("utf-8") if isinstance(v, str) else bytes(v)


class _Pattern:
    """ahocorasick_pattern_t (php_ahocorasick.h:159-174): holds the caller's key/value/aux objects."""
    __slots__ = ("key", "key_id", "key_type", "value", "value_bytes", "aux", "has_aux")

    def __init__(self):
        self.key = None
        self.key_id = 0
        self.key_type = N.ID_DEFAULT
        self.value = None
        self.value_bytes = b""
        self.aux = None
        self.has_aux = False


class AhoResource:
    """The "AhoCorasick search" resource (php_ahocorasick.h:179-192)."""

    def __init__(self, handle):
        self._h = handle
        self._patterns: list[_Pattern] = []
        self._closed = False

    def __repr__(self):
        return f"<AhoCorasick search resource, {len(self._patterns)} patterns, {'closed' if self._closed else 'open'}>"

    def __del__(self):  # resource dtor at request shutdown (php_ahocorasick.c:494-512)
        try:
            if not self._closed and self._h:
                N.lib.acmi_destroy(self._h)
                self._closed = True
        except Exception:
            pass

    # introspection helpers used by tests / bench
    def stat(self, key: str) -> float:
        v = C.c_double(0)
        N.check(N.lib.acmi_get_stat(self._h, key.encode(), C.byref(v)), self._h)
        return v.value

    def set_option(self, key: str, value: int) -> None:
        N.check(N.lib.acmi_set_option(self._h, key.encode(), int(value)), self._h)


def _sub_items(elem):
    """Iterate a pattern sub-array as (string_key_or_None, value), PHP foreach order."""
    if isinstance(elem, dict):
        for k, v in elem.items():
            yield (None if isinstance(k, int) and not isinstance(k, bool) else k), v
    else:  # list / tuple == PHP array with numeric keys
        for v in elem:
            yield None, v


def _process_pattern(pidx: int, elem) -> _Pattern | None:
    """php_ahocorasick_process_pattern (php_ahocorasick.c:195-336).  None == failure (warned)."""
    p = _Pattern()
    # the usual shapes -- {"value": v}, {"id": int, "value": v}, {"key": str, "value": v} with exactly these spellings -- without the
    # general loop below (which they would pass the same way): 2,048 patterns 3.9 -> 2.3 ms of ahocorasick_init
    if type(elem) is dict:
        v = elem.get("value")
        if type(v) is bytes or type(v) is str:
            n = len(elem)
            if n == 2:
                i = elem.get("id")
                if type(i) is int:
                    p.key_id = i
                    p.key_type = N.ID_NUMBER
                    n = 1
                else:
                    k = elem.get("key")
                    if type(k) is str or type(k) is bytes:
                        p.key = k
                        p.key_type = N.ID_STRING
                        n = 1
            if n == 1:
                p.value = v
                p.value_bytes = v if type(v) is bytes else v.encode("utf-8")
                return p
            p = _Pattern()
    all_keys = 0
    for key, data in _sub_items(elem):
        if key is None:
            found = 2  # bare numeric key == value (:233-234)
        else:
            if not isinstance(key, str):
                key = str(key)
            lk = key.lower()  # COMPAT_STR_EQUALS_CI
            if lk == "key":
                found = 1
            elif lk == "value":
                found = 2
            elif lk == "ignorecase":
                found = 4
            elif lk == "id":
                found = 8
            elif lk == "aux":
                found = 0x10
            else:
                _warn("Invalid structure (unrecognized sub-array key)! Only allowed are: {key, id, value, aux, "
                      f"ignoreCase}}. Cannot initialize. Pattern index: {pidx}")
                return None
        all_keys |= found
        if found & 8:
            if isinstance(data, bool) or not isinstance(data, int):
                raise AhoException(f"Invalid type of pattern ID given (long required), type: {_php_type(data)}, "
                                   f"pattern index: {pidx}")
            p.key_id = int(data)
            p.key_type = N.ID_NUMBER
        if found & 0x10:
            p.aux = data
            p.has_aux = True
        if found & 3:
            if not isinstance(data, (str, bytes, bytearray)):
                raise AhoException(f"Pattern {'key' if found == 1 else 'value'} has to be a string, type: "
                                   f"{_php_type(data)}, pattern index: {pidx}")
            if found == 1:
                p.key = data
                p.key_type = N.ID_STRING
            else:
                p.value = data
                p.value_bytes = _to_bytes(data)
    if p.value is None:
        _warn(f"No value was specified for pattern index: {pidx}")
        return None
    if (all_keys & 1) and (all_keys & 8):
        _warn(f"Pattern can have either numeric or string identifier, not both! Pattern index: {pidx}")
        return None
    if all_keys & 4:
        _warn(f"ignoreCase attribute is deprecated and is ignored. Pattern index: {pidx}")
    return p


_PATTERN_DTYPE = np.dtype([("value", "<u8"), ("length", "<u4"), ("id_type", "<u4"), ("id_number", "<i8"), ("user", "<u8")])
assert _PATTERN_DTYPE.itemsize == C.sizeof(N.Pattern)


def _process_patterns(res: AhoResource, data) -> int:
    """php_ahocorasick_process_patterns (php_ahocorasick.c:389-489): 0 on success, <0 on failure."""
    if isinstance(data, dict):
        # The reference NULL-derefs on arrays without key 0 (:411-413,:347); require list-shaped input.
        if list(data.keys()) != list(range(len(data))):
            _warn("Invalid pattern structure! Cannot initialize.")
            return -4
        data = list(data.values())
    parsed: list[_Pattern] = []
    for idx, elem in enumerate(data):
        if not isinstance(elem, (dict, list, tuple)):
            _warn("Invalid pattern structure! Cannot initialize.")
            return -4
        p = _process_pattern(idx, elem)
        if p is None:
            return -1
        parsed.append(p)
    if not parsed:
        return 0
    n = len(parsed)
    # acmi_pattern array filled column-wise (65,536 patterns: 20 ms instead of 0.3 s of per-field ctypes stores): all
    # values back to back in one buffer, `value` = its address + the pattern's offset
    lengths = np.fromiter((len(p.value_bytes) for p in parsed), dtype=np.uint32, count=n)
    blob = np.frombuffer(b"".join(p.value_bytes for p in parsed) + b"\0", dtype=np.uint8)
    offs = np.zeros(n, dtype=np.uint64)
    np.cumsum(lengths[:-1], out=offs[1:])
    rec = np.zeros(n, dtype=_PATTERN_DTYPE)
    rec["value"] = np.uint64(blob.ctypes.data) + offs
    rec["length"] = lengths
    rec["id_type"] = np.fromiter((p.key_type for p in parsed), dtype=np.uint32, count=n)
    rec["id_number"] = np.fromiter((p.key_id if p.key_type == N.ID_NUMBER else 0 for p in parsed), dtype=np.int64, count=n)
    rec["user"] = np.arange(len(res._patterns), len(res._patterns) + n, dtype=np.uint64)
    arr = rec.ctypes.data_as(C.POINTER(N.Pattern))
    N.check(N.lib.acmi_add_patterns(res._h, arr, n, None), res._h)
    res._patterns.extend(parsed)
    return 0


def ahocorasick_init(data):
    """PHP_FUNCTION(ahocorasick_init) (php_ahocorasick.c:798-834): resource, or False on bad patterns."""
    if not isinstance(data, (list, tuple, dict)):
        raise TypeError("ahocorasick_init(): Argument #1 ($data) must be of type array")
    h = N.Handle()
    N.check(N.lib.acmi_create(-1, C.byref(h)))
    res = AhoResource(h)
    try:
        status = _process_patterns(res, data)
    except Exception:
        N.lib.acmi_destroy(h)
        res._closed = True
        raise
    if status != 0:
        N.lib.acmi_destroy(h)
        res._closed = True
        return False
    return res


def _fetch(res) -> AhoResource | None:
    return res if isinstance(res, AhoResource) and not res._closed else None


def ahocorasick_add_patterns(res, patterns) -> bool:
    """PHP_FUNCTION(ahocorasick_add_patterns) (php_ahocorasick.c:882-925)."""
    r = _fetch(res)
    if r is None or not N.lib.acmi_is_valid(r._h):
        _warn("Cannot add a new pattern, not initialized")
        return False
    if r.stat("finalized") > 0:  # php_ahocorasick.c:914-917 (ac_finalized)
        _warn("Cannot add a new pattern to finalized search structure")
        return False
    if not isinstance(patterns, (list, tuple, dict)):
        raise TypeError("ahocorasick_add_patterns(): Argument #2 ($patterns) must be of type array")
    return _process_patterns(r, patterns) == 0


def ahocorasick_finalize(res) -> bool:
    """PHP_FUNCTION(ahocorasick_finalize) (php_ahocorasick.c:845-876): True only the first time."""
    r = _fetch(res)
    if r is None:
        return False
    return N.check(N.lib.acmi_finalize(r._h), r._h) == 1


def ahocorasick_isValid(res) -> bool:
    """PHP_FUNCTION(ahocorasick_isValid) (php_ahocorasick.c:623-655)."""
    r = _fetch(res)
    return bool(r is not None and N.lib.acmi_is_valid(r._h))


def ahocorasick_deinit(res) -> bool:
    """PHP_FUNCTION(ahocorasick_deinit) (php_ahocorasick.c:754-791); a second call returns False."""
    r = _fetch(res)
    if r is None:
        return False
    N.lib.acmi_destroy(r._h)
    r._closed = True
    return True


def _pack(r: AhoResource, rec) -> dict:
    """php_ahocorasick_match_handler's array build (php_ahocorasick.c:562-583), one record."""
    p = r._patterns[rec.pattern]
    out: dict[str, Any] = {"pos": int(rec.pos)}
    if p.key_type == N.ID_STRING:
        out["key"] = p.key
    elif p.key_type == N.ID_NUMBER:
        out["keyIdx"] = p.key_id
    if p.has_aux:
        out["aux"] = p.aux
    out["start_postion"] = int(rec.start)
    out["value"] = p.value
    return out


def ahocorasick_match(needle, res, findAll: bool = True):
    """PHP_FUNCTION(ahocorasick_match) (php_ahocorasick.c:664-746): list of records, or False."""
    r = _fetch(res)
    if r is None:
        _warn("Invalid resource.")
        return False
    if not N.lib.acmi_is_valid(r._h):
        _warn("Not initialized.")
        return False
    text = _to_bytes(needle)
    recs = C.POINTER(N.Record)()
    n = C.c_size_t(0)
    # (the bytes object itself is the buffer: through a numpy view and its .ctypes the call cost 2.5 us more, a tenth of a match)
    N.check(N.lib.acmi_match(r._h, text if text else b"\0", len(text), 1 if findAll else 0, C.byref(recs), C.byref(n)), r._h)
    return [_pack(r, recs[i]) for i in range(n.value)]


def ahocorasick_match_batch(haystacks, res, findAll: bool = True):
    """NEW (not in the reference): one launch for a whole ``foreach`` of ahocorasick_match calls
    (examples/benchmark.php:65-67).  Returns a list of per-haystack record lists, or False."""
    r = _fetch(res)
    if r is None:
        _warn("Invalid resource.")
        return False
    if not N.lib.acmi_is_valid(r._h):
        _warn("Not initialized.")
        return False
    texts = [_to_bytes(t) for t in haystacks]
    off = np.zeros(len(texts) + 1, dtype=np.uint64)
    if texts:
        off[1:] = np.cumsum([len(t) for t in texts], dtype=np.uint64)
    blob = np.frombuffer(b"".join(texts), dtype=np.uint8) if int(off[-1]) else np.zeros(1, np.uint8)
    prefix, records = match_arrays(r, blob, off, findAll)
    out = []
    for k in range(len(texts)):
        lo, hi = int(prefix[k]), int(prefix[k + 1])
        out.append([_pack(r, _Rec(records[i])) for i in range(lo, hi)])
    return out


class _Rec:
    __slots__ = ("hay", "pos", "start", "pattern")

    def __init__(self, row):
        self.hay, self.pos, self.start, self.pattern = (int(x) for x in row)


def match_arrays(res: AhoResource, blob: np.ndarray, off: np.ndarray, findAll: bool = True):
    """Raw batch entry: (prefix uint64[nh+1], records uint32[n,4] = {hay,pos,start,pattern}) from host arrays."""
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    nh = len(off) - 1
    pp = C.POINTER(C.c_uint64)()
    rp = C.POINTER(N.Record)()
    n = C.c_size_t(0)
    N.check(N.lib.acmi_match_batch(res._h, blob.ctypes.data, off.ctypes.data, nh, 1 if findAll else 0,
                                   C.byref(pp), C.byref(rp), C.byref(n)), res._h)
    prefix = np.ctypeslib.as_array(pp, shape=(nh + 1,)).copy()
    if n.value:
        records = np.ctypeslib.as_array(C.cast(rp, C.POINTER(C.c_uint32)), shape=(n.value, 4)).copy()
    else:
        records = np.zeros((0, 4), dtype=np.uint32)
    return prefix, records


def set_replacement(res: AhoResource, index: int, text: bytes | None) -> None:
    """AC_PATTERN_t.rtext of caller-order pattern ``index`` (before finalize); None: matched but not replaced."""
    N.check(N.lib.acmi_set_replacement(res._h, index, text, len(text) if text is not None else 0), res._h)


def replace_arrays(res: AhoResource, blob: np.ndarray, off: np.ndarray, mode: int = 1):
    """multifast_replace + flush over a batch of host haystacks (src/multifast/replace.c:469-568; not part of the PHP
    surface).  Returns (out_blob uint8[], out_off uint64[nh+1]); raises AcmiError(E_NO_REPLACEMENT) when no pattern
    has a replacement text."""
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    nh = len(off) - 1
    bp = C.POINTER(C.c_uint8)()
    op = C.POINTER(C.c_uint64)()
    N.check(N.lib.acmi_replace_batch(res._h, blob.ctypes.data, off.ctypes.data, nh, mode, C.byref(bp), C.byref(op)), res._h)
    out_off = np.ctypeslib.as_array(op, shape=(nh + 1,)).copy()
    total = int(out_off[-1])
    out = np.ctypeslib.as_array(bp, shape=(total,)).copy() if total else np.zeros(0, np.uint8)
    return out, out_off


class AhoStream:
    """Streaming search over one input fed in pieces: MultiFast's keep=1 mode
    (src/multifast/ahocorasick.c:175-241 with keep != 0), which the PHP extension itself never uses
    (keep=0 at src/php_ahocorasick.c:745).  Positions count from the start of the stream; a match that
    straddles two pieces is reported by the feed that delivers its last byte.

        s = AhoStream(res)
        for piece in pieces:
            records = s.feed(piece)        # uint32[n, 4] = {0, pos, start, pattern index}
        s.close()
    """

    def __init__(self, res: AhoResource):
        if not isinstance(res, AhoResource) or not res._h:
            raise ValueError("Invalid resource.")
        self._res = res
        self._s = C.c_void_p()
        N.check(N.lib.acmi_stream_open(res._h, C.byref(self._s)), res._h)

    def feed(self, piece: bytes) -> np.ndarray:
        buf = np.frombuffer(piece, dtype=np.uint8) if len(piece) else np.zeros(1, np.uint8)
        rp = C.POINTER(N.Record)()
        n = C.c_size_t(0)
        N.check(N.lib.acmi_stream_feed(self._res._h, self._s, buf.ctypes.data, len(piece), C.byref(rp), C.byref(n)),
                self._res._h)
        if not n.value:
            return np.zeros((0, 4), dtype=np.uint32)
        return np.ctypeslib.as_array(C.cast(rp, C.POINTER(C.c_uint32)), shape=(n.value, 4)).copy()

    @property
    def position(self) -> int:
        return int(N.lib.acmi_stream_position(self._s))

    def close(self) -> None:
        if self._s:
            N.lib.acmi_stream_close(self._s)
            self._s = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
