"""ctypes front end for the CPU oracle -- TEST INFRASTRUCTURE ONLY.

``Oracle("port")``      -> oracle/liboracle.so   (our restatement, ac_oracle.c)
``Oracle("reference")`` -> oracle/_ref/libmf_ref.so (the reference's own C, built in place)

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PORT_SO = os.path.join(_HERE, "liboracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libmf_ref.so")


def build(quiet: bool = True) -> None:
    """Compile liboracle.so (always) and _ref/libmf_ref.so (only when /root/reference exists)."""
    subprocess.run(["make", "-C", _HERE], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def have_reference() -> bool:
    return os.path.exists(REF_SO)


_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)


def _ptr(a: np.ndarray, typ):
    return a.ctypes.data_as(typ)


class Oracle:
    """One trie.  Mirrors the call sequence of the PHP layer (reverse insertion, keep=0)."""

    def __init__(self, kind: str = "port"):
        if kind == "port":
            if not os.path.exists(PORT_SO):
                build()
            self.lib, self.pfx = C.CDLL(PORT_SO), "orc_"
        elif kind == "reference":
            if not os.path.exists(REF_SO):
                raise FileNotFoundError(REF_SO)
            self.lib, self.pfx = C.CDLL(REF_SO), "ref_"
        else:
            raise ValueError(kind)
        self.kind = kind
        f = self._f
        f("create", C.c_void_p)
        f("add", C.c_int, C.c_void_p, _u8p, C.c_size_t, C.c_uint32)
        f("finalize", None, C.c_void_p)
        f("release", None, C.c_void_p)
        f("node_count", C.c_size_t, C.c_void_p)
        f("pattern_count", C.c_size_t, C.c_void_p)
        f("php_add_patterns", C.c_size_t, C.c_void_p, _u8p, _u64p, C.c_size_t, C.c_uint32)
        f("php_match", C.c_size_t, C.c_void_p, _u8p, C.c_size_t, C.c_int, C.c_uint32, _u32p, C.c_size_t)
        f("php_match_batch", C.c_size_t, C.c_void_p, _u8p, _u64p, C.c_size_t, C.c_int, _u64p, _u32p,
          C.c_size_t)
        f("php_scan_checksum", C.c_size_t, C.c_void_p, _u8p, _u64p, C.c_size_t, _u64p)
        f("php_scan_checksum_reps", C.c_size_t, C.c_void_p, _u8p, _u64p, C.c_size_t, C.c_size_t, _u64p)
        f("benchmark_sample", C.c_size_t, _u8p, _u64p, C.c_size_t, _u8p, _u64p, C.c_size_t, C.c_size_t, C.POINTER(C.c_double))
        f("stream_feed", C.c_size_t, C.c_void_p, _u8p, C.c_size_t, C.c_int, _u32p, C.c_size_t)
        f("set_replacement", None, C.c_void_p, C.c_uint32, C.c_char_p, C.c_size_t)
        f("replace", C.c_size_t, C.c_void_p, _u8p, C.c_size_t, C.c_int, _u8p, C.c_size_t)
        self.t = self._c("create")()
        self.n_added = 0

    def _f(self, name, res, *args):
        fn = getattr(self.lib, self.pfx + name)
        fn.restype, fn.argtypes = res, list(args)

    def _c(self, name):
        return getattr(self.lib, self.pfx + name)

    def close(self):
        if self.t:
            self._c("release")(self.t)
            self.t = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- build -------------------------------------------------------------------------
    def add_patterns(self, patterns) -> int:
        """One init()/add_patterns() call; ``patterns`` is a list of bytes or (blob, off)."""
        if isinstance(patterns, tuple):
            blob, off = patterns
        else:
            from php_aho_corasick_amd.synth import pack
            blob, off = pack([bytes(p) for p in patterns])
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = len(off) - 1
        keep = blob if blob.size else np.zeros(1, np.uint8)
        acc = self._c("php_add_patterns")(self.t, _ptr(keep, _u8p), _ptr(off, _u64p), n, self.n_added)
        self.n_added += n
        return int(acc)

    def finalize(self):
        self._c("finalize")(self.t)

    def node_count(self) -> int:
        return int(self._c("node_count")(self.t))

    def pattern_count(self) -> int:
        return int(self._c("pattern_count")(self.t))

    # -- scan --------------------------------------------------------------------------
    def match(self, text: bytes, find_all: bool = True, hay: int = 0) -> np.ndarray:
        """Records of one haystack as an (n, 4) uint32 array {hay, pos, start, index}."""
        buf = np.frombuffer(text, dtype=np.uint8) if len(text) else np.zeros(1, np.uint8)
        cap = 64
        while True:
            out = np.zeros((cap, 4), dtype=np.uint32)
            n = self._c("php_match")(self.t, _ptr(buf, _u8p), len(text), int(find_all), hay,
                                     _ptr(out, _u32p), cap)
            if n <= cap:
                return out[:n].copy()
            cap = int(n)

    def stream_feed(self, chunk: bytes, first: bool) -> np.ndarray:
        """One chunk through MultiFast's keep=1 mode (ahocorasick.c:186-194,236-238): records (n, 4) uint32
        {0, pos, start, index} with positions counted from the start of the stream."""
        buf = np.frombuffer(chunk, dtype=np.uint8) if len(chunk) else np.zeros(1, np.uint8)
        cap = 64 + 4 * len(chunk)
        while True:
            out = np.zeros((cap, 4), dtype=np.uint32)
            n = self._c("stream_feed")(self.t, _ptr(buf, _u8p), len(chunk), int(first), _ptr(out, _u32p), cap)
            if n <= cap:
                return out[:n].copy()
            # the reference saved its state; a retry would double-count: size generously instead
            raise RuntimeError("stream_feed: output capacity too small")

    def match_batch(self, blob: np.ndarray, off: np.ndarray, find_all: bool = True):
        """(counts_prefix[nh+1] uint64, records (n,4) uint32) over a batch, haystack order."""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        nh = len(off) - 1
        keep = blob if blob.size else np.zeros(1, np.uint8)
        prefix = np.zeros(nh + 1, dtype=np.uint64)
        total = self._c("php_match_batch")(self.t, _ptr(keep, _u8p), _ptr(off, _u64p), nh,
                                           int(find_all), _ptr(prefix, _u64p), None, 0)
        out = np.zeros((max(int(total), 1), 4), dtype=np.uint32)
        self._c("php_match_batch")(self.t, _ptr(keep, _u8p), _ptr(off, _u64p), nh, int(find_all),
                                   _ptr(prefix, _u64p), _ptr(out, _u32p), int(total))
        return prefix, out[:int(total)]

    # -- search and replace (replace.c; not reachable from the PHP surface) --------------------
    def set_replacement(self, index: int, text: bytes | None) -> None:
        """Replacement text of caller-order pattern ``index``; call BEFORE the add_patterns that brings the pattern."""
        self._c("set_replacement")(self.t, index, text, len(text) if text is not None else 0)

    def replace(self, text: bytes, mode: int = 1) -> bytes | None:
        """multifast_replace + multifast_rep_flush(keep=0) over one whole text; None when no pattern has a replacement."""
        self.finalize()
        buf = np.frombuffer(bytes(text) or b"\0", dtype=np.uint8)
        cap = len(text) + 64
        while True:
            out = np.zeros(max(cap, 1), dtype=np.uint8)
            n = self._c("replace")(self.t, _ptr(buf, _u8p), len(text), mode, _ptr(out, _u8p), cap)
            if n >= (1 << 63):   # (size_t)-1 / -2
                return None
            if n <= cap:
                return out[:n].tobytes()
            cap = int(n)

    def scan_checksum(self, blob: np.ndarray, off: np.ndarray):
        """(record count, checksum) without materialising records (cpu_baseline timing leg)."""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        ck = C.c_uint64(0)
        n = self._c("php_scan_checksum")(self.t, _ptr(blob, _u8p), _ptr(off, _u64p), len(off) - 1,
                                         C.byref(ck))
        return int(n), int(ck.value)


    def scan_checksum_reps(self, blob: np.ndarray, off: np.ndarray, reps: int):
        """The same scan `reps` times in ONE call (the loop is in C: what a thread of the all-cores baseline runs)."""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        ck = C.c_uint64(0)
        n = self._c("php_scan_checksum_reps")(self.t, _ptr(blob, _u8p), _ptr(off, _u64p), len(off) - 1, int(reps), C.byref(ck))
        return int(n), int(ck.value)

    def benchmark_sample(self, pat_blob, pat_off, hay_blob, hay_off, samples: int):
        """examples/benchmark.php:55-72 in C: samples x (create + add + finalize, one search per haystack, release).  Returns
        (records of a sample, seconds of build, of the searches, of release) summed over the samples."""
        pb, po = np.ascontiguousarray(pat_blob, dtype=np.uint8), np.ascontiguousarray(pat_off, dtype=np.uint64)
        hb, ho = np.ascontiguousarray(hay_blob, dtype=np.uint8), np.ascontiguousarray(hay_off, dtype=np.uint64)
        sec = (C.c_double * 3)(0.0, 0.0, 0.0)
        n = self._c("benchmark_sample")(_ptr(pb, _u8p), _ptr(po, _u64p), len(po) - 1, _ptr(hb, _u8p), _ptr(ho, _u64p), len(ho) - 1, int(samples), sec)
        return int(n), sec[0], sec[1], sec[2]


def record_checksum(records: np.ndarray) -> int:
    """Same checksum as orc_php_scan_checksum over an (n,4) uint32 record array."""
    if len(records) == 0:
        return 0
    r = records.astype(np.uint64)
    with np.errstate(over="ignore"):
        v = (r[:, 0] * np.uint64(0x9E3779B1)) ^ (r[:, 1] * np.uint64(0x85EBCA77)) ^ \
            (r[:, 2] * np.uint64(0xC2B2AE3D)) ^ r[:, 3]
        return int(np.sum(v, dtype=np.uint64))
