"""Shared test helpers (CPU side)."""
from __future__ import annotations

import ctypes as C
import hashlib
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_json(name):
    with open(os.path.join(GOLDEN, name), encoding="utf-8") as f:
        return json.load(f)


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def pack_record(pattern: dict, pos: int, start: int) -> list:
    """Independent restatement of the record layout (php_ahocorasick.c:562-583) as ordered pairs."""
    out = [["pos", pos]]
    if "key" in pattern:
        out.append(["key", pattern["key"]])
    elif "id" in pattern:
        out.append(["keyIdx", pattern["id"]])
    if "aux" in pattern:
        out.append(["aux", pattern["aux"]])
    out.append(["start_postion", start])
    out.append(["value", pattern["value"]])
    return out


def as_pairs(record: dict) -> list:
    return [[k, v] for k, v in record.items()]


class FlatTables:
    """numpy view of acmi_debug_tables() for CPU-side inspection of the builder's output."""

    def __init__(self, res):
        from php_aho_corasick_amd import _native as N
        t = N.Tables()
        N.check(N.lib.acmi_debug_tables(res._h, C.byref(t)), res._h)
        self.n_states, self.n_columns, self.entry_bytes = t.n_states, t.n_columns, t.entry_bytes
        self.f1, self.trap, self.f2 = t.first_hot_final, t.trap, t.first_cold_final
        self.class_mode, self.swar_shift, self.swar_other = t.class_mode, t.swar_shift, t.swar_other
        self.max_pattern_length = t.max_pattern_length
        dt = np.uint16 if t.entry_bytes == 2 else np.uint32
        self.compact, self.dense_nf_end, self.dense_f_end = t.compact, t.dense_nf_end, t.dense_f_end
        self.n_dense_rows = t.n_dense_rows if t.compact else t.n_states
        n = self.n_dense_rows * t.n_columns
        self.table = np.ctypeslib.as_array(C.cast(t.table, C.POINTER(C.c_uint16 if t.entry_bytes == 2 else C.c_uint32)),
                                           shape=(n,)).astype(np.int64).reshape(self.n_dense_rows, t.n_columns)
        self.cold_rec = self.cold_edges = None
        if t.compact:
            self.cold_rec = np.ctypeslib.as_array(t.cold_rec, shape=(t.n_states, 2)).astype(np.int64)
            self.cold_edges = (np.ctypeslib.as_array(t.cold_edges, shape=(t.n_cold_edges,)).astype(np.int64)
                               if t.n_cold_edges else np.zeros(0, np.int64))
        m = (t.trap + 1) * t.n_columns
        self.lds = np.ctypeslib.as_array(C.cast(t.lds_image, C.POINTER(C.c_uint16 if t.entry_bytes == 2 else C.c_uint32)),
                                         shape=(m,)).astype(np.int64).reshape(t.trap + 1, t.n_columns)
        self.cls = np.ctypeslib.as_array(t.class_of_byte, shape=(256,)).astype(np.int64)
        self.own = np.ctypeslib.as_array(t.own_pattern, shape=(t.n_states,)).astype(np.int64)
        self.dict = np.ctypeslib.as_array(t.dict_link, shape=(t.n_states,)).astype(np.int64)
        self.cnt = np.ctypeslib.as_array(t.out_count, shape=(t.n_states,)).astype(np.int64)
        _ = dt
        self.chain_lo = t.chain_lo
        self.chain_byte = self.chain_run = None
        if t.chain_byte:
            nchain = self.f2 - t.chain_lo
            self.chain_byte = np.ctypeslib.as_array(t.chain_byte, shape=(nchain + 16,)).astype(np.int64)
            self.chain_run = np.ctypeslib.as_array(t.chain_run, shape=(nchain + 16,)).astype(np.int64)
        self.out_off = self.out_list = None
        if t.out_off:
            self.out_off = np.ctypeslib.as_array(t.out_off, shape=(t.n_states,)).astype(np.int64)
            self.out_list = np.ctypeslib.as_array(t.out_list, shape=(max(t.n_flat_outputs, 1), 2)).astype(np.int64)[:t.n_flat_outputs]
        self.chain = None
        if t.chain_image:
            img = np.ctypeslib.as_array(t.chain_image, shape=(t.chain_bytes,)).copy()
            self.chain = dict(n_dense=t.chain_dense, n_index=t.chain_index, bytes=t.chain_bytes, cnt_n=t.chain_cnt_n,
                              rows=img[:t.chain_dense * t.n_columns * 2].view("<u2").astype(np.int64).reshape(t.chain_dense, t.n_columns),
                              recs=img[t.chain_rec_off:t.chain_rec_off + 4 * (t.chain_index - t.chain_dense)].view("<u4").astype(np.int64),
                              cnt=img[t.chain_cnt_off:t.chain_cnt_off + t.chain_cnt_n].astype(np.int64),
                              orig=np.ctypeslib.as_array(t.chain_orig, shape=(t.chain_index,)).astype(np.int64),
                              head=np.ctypeslib.as_array(t.chain_head, shape=(t.chain_index, 2)).astype(np.int64))
        self.win = None
        if t.win_table:
            self.win = dict(cells=np.ctypeslib.as_array(t.win_table, shape=(65536 + t.win_list_bytes // 2,)).astype(np.int64),
                            state=np.ctypeslib.as_array(t.win_state, shape=(65536,)).astype(np.int64),
                            hash=np.ctypeslib.as_array(t.win_hash, shape=(1 << t.win_hash_bits, 2)).astype(np.int64), bits=t.win_hash_bits,
                            shallow_hot=t.shallow_hot, shallow_cold_end=t.shallow_cold_end)
        self.anchor_q, self.anchor_mul, self.min_pattern_length = t.anchor_q, t.anchor_mul, t.min_pattern_length
        self.anchor_keys = t.anchor_keys
        self.anchor_mode, self.anchor_bucket_bits = t.anchor_mode, t.anchor_bucket_bits
        self.anchor_bucket_shift = t.anchor_bucket_shift
        self.anchor_choice = t.anchor_choice
        self.anchor_dense = t.anchor_dense
        self.anchor_slots = t.anchor_slots
        self.anchor_offsets = None   # filled by test code that knows the pattern count
        self._anchor_offsets_ptr = t.anchor_offsets
        self.anchor = self.anchor_bitmap = None
        if t.anchor_mode == 1:
            self.anchor = np.ctypeslib.as_array(t.anchor_table, shape=(16384 * 4 // t.anchor_slots, t.anchor_slots)).copy()
        elif t.anchor_mode == 2:
            self.anchor = np.ctypeslib.as_array(t.anchor_table, shape=(1 << t.anchor_bucket_bits, 8)).copy()
            self.anchor_bitmap = np.ctypeslib.as_array(t.anchor_bitmap, shape=(1 << 15,)).copy()

    def anchor_offset(self, pattern: int, a: int) -> int:
        return int(self._anchor_offsets_ptr[pattern * 4 + a]) if self._anchor_offsets_ptr else a

    def anchor_probe(self, window: bytes) -> bool:
        """The device's window test (k_scan_anchor), restated: hash q bytes, look the fingerprint up."""
        q = self.anchor_q
        words = np.frombuffer(window[:q], dtype="<u4").astype(np.uint64)
        if len(words) == 1:   # a lone dword is hashed raw, one multiply-add (automaton.hpp anchor_hash)
            w = int(words[0])
            h = ((w & 0xFFFFFF) * (self.anchor_mul & 0xFFFFFF) + (w >> 13)) & 0xFFFFFFFF
        else:
            fold = [(int(w) ^ (int(w) >> 11)) & 0xFFFFFFFF for w in words]
            h = fold[0]
            for f in fold[1:]:
                h = ((h & 0xFFFFFF) * (self.anchor_mul & 0xFFFFFF) + f) & 0xFFFFFFFF
        if self.anchor_mode == 2:   # LDS bitmap in front of the 8-slot table read through L2
            if not (int(self.anchor_bitmap[h >> 17]) >> ((h >> 12) & 31)) & 1:
                return False
            return int((h >> 16) | 1) in [int(x) for x in self.anchor[(h >> self.anchor_bucket_shift) & ((1 << self.anchor_bucket_bits) - 1)]]
        if self.anchor_slots == 2:   # 2^15 buckets of two slots: hash bits 2..16
            return int(h >> 16) in [int(x) for x in self.anchor[(h >> 2) & 0x7FFF]]
        return int(h >> 16) in [int(x) for x in self.anchor[(h >> 3) & 0x3FFF]]

    def cold_step(self, s: int, b: int) -> int:
        """kdev.hpp cold_step(): dense row, or goto / failure records until a dense row is reached."""
        c = int(self.cls[b])
        if not self.compact:
            return int(self.table[s, c])
        while True:
            if s < self.dense_nf_end:
                return int(self.table[s, c])
            if self.f2 <= s < self.dense_f_end:
                return int(self.table[self.dense_nf_end + s - self.f2, c])
            x, y = int(self.cold_rec[s, 0]), int(self.cold_rec[s, 1])
            code = y >> 24
            if code == 1:
                if (x & 0xFF) == b:
                    return x >> 8
            elif code >= 2:
                n, at = (code, x) if code != 255 else (int(self.cold_edges[x]), x + 1)
                edges = self.cold_edges[at:at + n]
                assert np.all(np.diff(edges & 0xFF) > 0), "edges must ascend by byte"
                hit = edges[(edges & 0xFF) == b]
                if hit.size:
                    return int(hit[0]) >> 8
            assert (y & 0xFFFFFF) != s
            s = y & 0xFFFFFF

    def walk_win(self, text: bytes, pat_len, hay: int = 0, find_all: bool = True):
        """Scan one haystack as k_walk_win does (csrc/k_win.hip): while the state is of trie depth <= 2 only the last two bytes and the
        window table decide whether the next byte leaves depth 2 downward; a "maybe" is settled by the depth-3 hash; deeper states
        step through dense rows / compact records, and a failure link that leads back to depth <= 2 hands the byte to the table."""
        w = self.win
        cells, lists = w["cells"][:65536], w["cells"][65536:]
        mask = (1 << w["bits"]) - 1

        def shallow(s):
            return s < w["shallow_hot"] or (self.trap < s < w["shallow_cold_end"])

        def maybe(cell, c):
            kind = cell >> 14
            if kind == 2:
                return (cell & 0xFF) == c
            f = (c ^ (c >> 3) ^ (c >> 5)) & 127
            if kind == 1:
                return f == (cell & 127) or f == ((cell >> 7) & 127)
            if kind == 3:
                if cell == 0xFFFF:
                    return True
                u = cell & 0x3FFF
                lb = [(int(lists[(2 * u + i) // 2]) >> (8 * ((2 * u + i) & 1))) & 0xFF for i in range(64)]
                return c in lb[1:1 + lb[0]]
            return False

        out, deep, st, idx = [], False, 0, 0
        for j, c in enumerate(text):
            stepped = False
            if deep:
                # one transition, goto edges and failure links of the records; back at depth <= 2 through a failure link: the table's byte
                s = st
                while True:
                    if not self.compact or s < self.dense_nf_end or self.f2 <= s < self.dense_f_end:
                        st = self.cold_step(s, c)
                        stepped = True
                        break
                    x, y = int(self.cold_rec[s, 0]), int(self.cold_rec[s, 1])
                    code, nxt = y >> 24, None
                    if code == 1 and (x & 0xFF) == c:
                        nxt = x >> 8
                    elif code >= 2:
                        n, at = (code, x) if code != 255 else (int(self.cold_edges[x]), x + 1)
                        hit = [int(e) >> 8 for e in self.cold_edges[at:at + n] if (int(e) & 0xFF) == c]
                        nxt = hit[0] if hit else None
                    if nxt is not None:
                        st, stepped = nxt, True
                        break
                    s = y & 0xFFFFFF
                    if shallow(s):
                        deep = False
                        break
                if stepped:
                    deep = not shallow(st)
            if not stepped and j >= 2 and maybe(int(cells[idx]), c):
                key = (idx << 8 | c) + 1
                at = ((key - 1) * 0x9E3779B1 & 0xFFFFFFFF) >> (32 - w["bits"])
                while int(w["hash"][at, 0]) != 0 and (int(w["hash"][at, 0]) & 0x7FFFFFFF) != key:
                    at = (at + 1) & mask
                if int(w["hash"][at, 0]) != 0:
                    word = int(w["hash"][at, 1])
                    st, deep = (word & 0xFFFFFF if word >> 31 else word), True
                    if word >> 31:   # the entry's child byte restates the node's record
                        cb = ((word >> 24) & 0x7F) | (int(w["hash"][at, 0]) >> 31) << 7
                        assert not self.is_final(st)
                        if self.compact and not (st < self.dense_nf_end):
                            assert int(self.cold_rec[st, 1]) >> 24 == 1 and (int(self.cold_rec[st, 0]) & 0xFF) == cb and shallow(int(self.cold_rec[st, 1]) & 0xFFFFFF)
            idx = ((idx << 8) | c) & 0xFFFF
            if deep and self.is_final(st):
                pos = j + 1
                o, n = int(self.out_off[st]), int(self.cnt[st])
                out += [(hay, pos, pos - int(l), int(p)) for p, l in self.out_list[o:o + n]]
                if not find_all:
                    break
        return out

    def walk_chain(self, text: bytes, pat_len, hay: int = 0, find_all: bool = True):
        """Scan one haystack over the chain-compressed image exactly as k_walk_chain steps through it (automaton.hpp ChainImage): a
        C state's record first -- the edge's class leads to the next index, the parent's record says whether that state is final and
        how many records its event has --, else the row of its failure state (a D state's own row), whose entry carries the final bit."""
        ch = self.chain
        nd, rows, recs_, cnt, orig, head = ch["n_dense"], ch["rows"], ch["recs"], ch["cnt"], ch["orig"], ch["head"]
        out, e = [], 0
        for j, b in enumerate(text):
            c = int(self.cls[b])
            final, n = False, 0
            if e >= nd:
                r = int(recs_[e - nd])
                assert (r & 0xFFFF) < nd, "a C state's failure state must be a D state"
                if ((r >> 16) & 0xFF) == 2 * c:
                    e, final, n = e + 1, bool(r >> 31), (r >> 24) & 31
                    assert e >= nd and e < ch["n_index"], "a C state's child is the next index, a C state"
                else:
                    t = int(rows[r & 0xFFFF, c])
                    e, final = t & 0x7FFF, bool(t & 0x8000)
                    n = (int(cnt[e]) if e < ch["cnt_n"] else 255) if final else 0
            else:
                t = int(rows[e, c])
                e, final = t & 0x7FFF, bool(t & 0x8000)
                n = (int(cnt[e]) if e < ch["cnt_n"] else 255) if final else 0
            s = int(orig[e])
            assert final == self.is_final(s)
            if final:
                if n in (31, 255) or n == 0:
                    n = int(self.cnt[s])
                assert n == self.cnt[s] and (int(head[e, 0]) >> 24) == min(n, 255)
                pos = j + 1
                o = int(self.out_off[s])
                assert (int(head[e, 0]) & 0xFFFFFF, int(head[e, 1])) == (int(self.out_list[o, 0]), int(self.out_list[o, 1]))
                out += [(hay, pos, pos - int(l), int(p)) for p, l in self.out_list[o:o + n]]
                if not find_all:
                    break
        return out

    def is_final(self, s: int) -> bool:
        return (self.f1 <= s < self.trap) or s >= self.f2

    def walk(self, text: bytes, pat_len, hay: int = 0, find_all: bool = True, use_lds: bool = True):
        """Scan one haystack over the flat tables exactly as the device's exact path does."""
        NONE = 0xFFFFFFFF
        recs = []
        s = 0
        lds, cls, trap = self.lds, self.cls, self.trap
        for j, b in enumerate(text):
            c = cls[b]
            w = trap
            if use_lds and s < trap:
                w = int(lds[s, c])
            if w >= trap:
                w = self.cold_step(s, b)  # row not in the LDS tile
            s = w
            if self.is_final(s):
                pos = j + 1
                t = s if self.own[s] != NONE else int(self.dict[s])
                n0 = len(recs)
                while t != NONE:
                    pat = int(self.own[t])
                    recs.append((hay, pos, pos - int(pat_len[pat]), pat))
                    t = int(self.dict[t])
                assert len(recs) - n0 == self.cnt[s]
                if self.out_off is not None:  # the flat list says the same, in the same order
                    o = int(self.out_off[s])
                    flat = [(hay, pos, pos - int(l), int(p)) for p, l in self.out_list[o:o + int(self.cnt[s])]]
                    assert flat == recs[n0:], "flat output list disagrees with the dictionary links"
                if not find_all:
                    break
        return recs
