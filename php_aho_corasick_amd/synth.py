"""Deterministic synthetic pattern / haystack generators.

The reference's own workload generator (examples/benchmark.php:9-31) draws from
PHP's unseeded ``rand()``, so only its SHAPE is reproducible (2048 needles x 16
chars, 256 haystacks x 8192 chars, alphabet ``abcdef``).  Every dataset in this
repo therefore comes from one seeded splitmix64 stream:

* all pattern bytes first (pattern-major, ``NP x PLmax`` draws even when a
  pattern uses fewer), then all haystack bytes (haystack-major);
* alphabet draw ``alphabet[next() % len(alphabet)]``; Sigma=256 draw
  ``next() & 0xFF``;
* variable length ``len_k = Lmin + ((k * 2654435761) >> 7) % (Lmax - Lmin + 1)``;
* plants are applied after generation.

Pure numpy, no device code: the same bytes are fed to the HIP path, to the
oracle and to the compiled reference.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, n: int, start: int = 0) -> np.ndarray:
    """Outputs ``start .. start+n-1`` of the splitmix64 stream seeded with ``seed``."""
    with np.errstate(over="ignore"):
        idx = np.arange(start + 1, start + n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _draw_bytes(seed: int, n: int, start: int, alphabet: bytes | None) -> np.ndarray:
    out = np.empty(n, dtype=np.uint8)
    step = 1 << 22
    alpha = None if alphabet is None else np.frombuffer(alphabet, dtype=np.uint8)
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        r = splitmix64(seed, hi - lo, start + lo)
        if alpha is None:
            out[lo:hi] = (r & np.uint64(0xFF)).astype(np.uint8)
        else:
            out[lo:hi] = alpha[(r % np.uint64(len(alpha))).astype(np.int64)]
    return out


@dataclass
class Dataset:
    """Patterns and haystacks as (blob, offsets) pairs; offsets are uint64, n+1 long."""

    name: str
    pat_blob: np.ndarray
    pat_off: np.ndarray
    hay_blob: np.ndarray
    hay_off: np.ndarray

    @property
    def n_patterns(self) -> int:
        return len(self.pat_off) - 1

    @property
    def n_haystacks(self) -> int:
        return len(self.hay_off) - 1

    def pattern(self, k: int) -> bytes:
        return self.pat_blob[int(self.pat_off[k]):int(self.pat_off[k + 1])].tobytes()

    def haystack(self, h: int) -> bytes:
        return self.hay_blob[int(self.hay_off[h]):int(self.hay_off[h + 1])].tobytes()

    def patterns(self) -> list[bytes]:
        return [self.pattern(k) for k in range(self.n_patterns)]


def pack(items: list[bytes]) -> tuple[np.ndarray, np.ndarray]:
    off = np.zeros(len(items) + 1, dtype=np.uint64)
    if items:
        off[1:] = np.cumsum([len(b) for b in items], dtype=np.uint64)
    blob = np.frombuffer(b"".join(items), dtype=np.uint8).copy() if items else np.zeros(0, np.uint8)
    return blob, off


def var_lengths(n: int, lmin: int, lmax: int) -> np.ndarray:
    k = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return (np.uint64(lmin) + ((k * np.uint64(2654435761)) >> np.uint64(7)) %
                np.uint64(lmax - lmin + 1)).astype(np.int64)


def generate(name: str, seed: int, n_pat: int, pat_lmin: int, pat_lmax: int,
             n_hay: int, hay_len: int, alphabet: bytes | None) -> Dataset:
    """Generic seeded dataset: fixed-length patterns when ``pat_lmin == pat_lmax``."""
    raw = _draw_bytes(seed, n_pat * pat_lmax, 0, alphabet).reshape(n_pat, pat_lmax) if n_pat else \
        np.zeros((0, pat_lmax), np.uint8)
    lens = np.full(n_pat, pat_lmax, dtype=np.int64) if pat_lmin == pat_lmax else \
        var_lengths(n_pat, pat_lmin, pat_lmax)
    pat_off = np.zeros(n_pat + 1, dtype=np.uint64)
    pat_off[1:] = np.cumsum(lens, dtype=np.uint64)
    if pat_lmin == pat_lmax:
        pat_blob = raw.reshape(-1).copy()
    else:
        mask = np.arange(pat_lmax)[None, :] < lens[:, None]
        pat_blob = raw[mask].copy()
    hay_blob = _draw_bytes(seed, n_hay * hay_len, n_pat * pat_lmax, alphabet)
    hay_off = (np.arange(n_hay + 1, dtype=np.uint64) * np.uint64(hay_len))
    return Dataset(name, pat_blob, pat_off, hay_blob, hay_off)


def plant(ds: Dataset, pos_mul: int, pos_add: int, pos_mod: int, pat_mul: int, pat_add: int) -> None:
    """hay[h][(pos_mul*h+pos_add) % pos_mod ...) = pat[(pat_mul*h+pat_add) % NP] for every h."""
    for h in range(ds.n_haystacks):
        p = ds.pattern((pat_mul * h + pat_add) % ds.n_patterns)
        at = int(ds.hay_off[h]) + (pos_mul * h + pos_add) % pos_mod
        ds.hay_blob[at:at + len(p)] = np.frombuffer(p, dtype=np.uint8)


def replacement(k: int, pattern: bytes):
    """Deterministic replacement text of pattern k for the search-and-replace fixtures (None: the pattern is
    matched but not replaced): a third of the patterns keep their text, some are deleted, some reversed."""
    if k % 3 == 2:
        return None
    if k % 7 == 0:
        return b""
    if k % 5 == 1:
        return pattern[::-1] + b"!"
    return b"<" + str(k).encode() + b">"


# ---- the named datasets of SURVEY.md section 8c/8d ---------------------------------------

def g1(n_hay: int = 256) -> Dataset:
    """Config 1/2 shape (examples/benchmark.php:20-31): 2048 x 16 B, n_hay x 8192 B, abcdef, seed 1."""
    return generate("G1", 1, 2048, 16, 16, n_hay, 8192, b"abcdef")


def g2(n_hay: int = 256) -> Dataset:
    """G1 plus one planted pattern per haystack."""
    ds = g1(n_hay)
    ds.name = "G2"
    plant(ds, 31, 5, 8176, 7, 3)
    return ds


def g3(n_hay: int = 4, hay_len: int = 8192, k: int = 16) -> Dataset:
    """Adversarial dense set: patterns a^1..a^k over haystacks a^hay_len (k records per byte)."""
    pat_blob, pat_off = pack([b"a" * (i + 1) for i in range(k)])
    hay_blob = np.full(n_hay * hay_len, ord("a"), dtype=np.uint8)
    hay_off = np.arange(n_hay + 1, dtype=np.uint64) * np.uint64(hay_len)
    return Dataset("G3", pat_blob, pat_off, hay_blob, hay_off)


def g4(n_hay: int = 4, hay_len: int = 8192) -> Dataset:
    """All 30 strings of length 1..4 over {a,b}; random {a,b} haystacks, seed 4."""
    pats = []
    for length in range(1, 5):
        for v in range(1 << length):
            pats.append(bytes(ord("a") + ((v >> (length - 1 - i)) & 1) for i in range(length)))
    pat_blob, pat_off = pack(pats)
    hay_blob = _draw_bytes(4, n_hay * hay_len, 0, b"ab")
    hay_off = np.arange(n_hay + 1, dtype=np.uint64) * np.uint64(hay_len)
    return Dataset("G4", pat_blob, pat_off, hay_blob, hay_off)


def g5() -> Dataset:
    """Duplicate / empty / over-long pattern rules on one tiny haystack."""
    pats = [b"ab", b"ab", b"b", b"", b"x" * 1025, b"xab"]
    pat_blob, pat_off = pack(pats)
    hay_blob, hay_off = pack([b"xabab_xxabb__ab_"])
    return Dataset("G5", pat_blob, pat_off, hay_blob, hay_off)


def g6(n_hay: int = 64, hay_len: int = 4096, n_pat: int = 65536) -> Dataset:
    """Config 3 automaton: 65,536 patterns of 8..64 B over Sigma=256, seed 7, one plant per haystack."""
    ds = generate("G6", 7, n_pat, 8, 64, n_hay, hay_len, None)
    plant(ds, 37, 11, hay_len - 64, 1021, 17)
    return ds


def bench_batch(n_hay: int, hay_len: int = 8192, seed: int = 1, planted: bool = True) -> Dataset:
    """Headline workload: the G1/G2 pattern set over ``n_hay`` haystacks of ``hay_len`` bytes."""
    ds = generate("bench", seed, 2048, 16, 16, n_hay, hay_len, b"abcdef")
    if planted:
        plant(ds, 31, 5, hay_len - 16, 7, 3)
    return ds


def ragged(name: str, seed: int, n_pat: int, pat_lmin: int, pat_lmax: int, n_hay: int,
           hay_lmin: int, hay_lmax: int, alphabet: bytes | None) -> Dataset:
    """Variable-length haystacks (including empty ones) packed back to back, so their
    start offsets are unaligned -- exercises head/tail handling and empty inputs."""
    base = generate(name, seed, n_pat, pat_lmin, pat_lmax, n_hay, hay_lmax, alphabet)
    k = np.arange(n_hay, dtype=np.uint64)
    with np.errstate(over="ignore"):
        lens = (np.uint64(hay_lmin) + ((k * np.uint64(40503) + np.uint64(seed)) * np.uint64(2654435761) >> np.uint64(11)) %
                np.uint64(hay_lmax - hay_lmin + 1)).astype(np.int64)
    full = base.hay_blob.reshape(n_hay, hay_lmax)
    mask = np.arange(hay_lmax)[None, :] < lens[:, None]
    hay_off = np.zeros(n_hay + 1, dtype=np.uint64)
    hay_off[1:] = np.cumsum(lens, dtype=np.uint64)
    return Dataset(name, base.pat_blob, base.pat_off, full[mask].copy(), hay_off)


def r1() -> Dataset:
    """Dense ragged: 24 patterns of 1..5 B over {a,b,NUL,0xFF}; 67 haystacks of 0..300 B."""
    return ragged("R1", 11, 24, 1, 5, 67, 0, 300, b"ab\x00\xff")


def r2() -> Dataset:
    """Binary ragged: 300 patterns of 2..6 B over Sigma=256 (2-byte ones do hit); 129 haystacks of 0..5000 B."""
    return ragged("R2", 12, 300, 2, 6, 129, 0, 5000, None)


def r3() -> Dataset:
    """Long patterns: 40 patterns of 900..1024 B over {x,y}; haystacks of 0..20000 B with plants."""
    ds = ragged("R3", 13, 40, 900, 1024, 9, 3000, 20000, b"xy")
    for h in range(ds.n_haystacks):
        p = ds.pattern((5 * h + 1) % ds.n_patterns)
        lo, hi = int(ds.hay_off[h]), int(ds.hay_off[h + 1])
        at = lo + (977 * h + 13) % max(1, hi - lo - len(p))
        ds.hay_blob[at:at + len(p)] = np.frombuffer(p, dtype=np.uint8)
    return ds


def _planted_ragged(name, seed, n_pat, lmin, lmax, n_hay, hmin, hmax, alphabet, plants_per_hay):
    """Ragged haystacks with several planted patterns each: at the very start, at the very end and at
    pseudo-random places (so matches straddle 16-byte quads, 128-byte lines and work-item chunks)."""
    ds = ragged(name, seed, n_pat, lmin, lmax, n_hay, hmin, hmax, alphabet)
    for h in range(ds.n_haystacks):
        lo, hi = int(ds.hay_off[h]), int(ds.hay_off[h + 1])
        for k in range(plants_per_hay):
            pat = ds.pattern((7 * h + 13 * k + 1) % ds.n_patterns)
            if hi - lo < len(pat):
                continue
            span = hi - lo - len(pat)
            at = 0 if k == 0 else span if k == 1 else ((h * 2654435761 + k * 40503) >> 3) % (span + 1)
            ds.hay_blob[lo + at:lo + at + len(pat)] = np.frombuffer(pat, dtype=np.uint8)
    return ds


def r4() -> Dataset:
    """Anchor filter with 8-byte windows: 150 patterns of 11..30 B over {a,b,c,d}, planted."""
    return _planted_ragged("R4", 14, 150, 11, 30, 61, 0, 6000, b"abcd", 5)


def r5() -> Dataset:
    """Anchor filter with 12-byte windows: 400 patterns of 15..40 B over Sigma=256, planted."""
    return _planted_ragged("R5", 15, 400, 15, 40, 53, 0, 9000, None, 6)


def r6() -> Dataset:
    """Anchor filter with 16-byte windows: 120 patterns of 19..64 B over {0,1}, planted (binary text makes
    accidental anchors common, so the candidate path is exercised hard)."""
    return _planted_ragged("R6", 16, 120, 19, 64, 37, 0, 12000, b"01", 4)


def r7() -> Dataset:
    """Anchor filter with 4-byte windows (LDS table): 300 patterns of 7..12 B over Sigma=256, planted."""
    return _planted_ragged("R7", 17, 300, 7, 12, 47, 0, 7000, None, 6)


def r8() -> Dataset:
    """Anchor filter for big sets (LDS bitmap + fingerprint table through L2): 5,000 patterns of 9..24 B
    over a 16-letter alphabet (20,000 anchor q-grams of 4 bytes), planted."""
    return _planted_ragged("R8", 18, 5000, 9, 24, 41, 0, 10000, b"abcdefghijklmnop", 8)


# ---- targeted datasets for the anchor kernel's item-shape variants (tests/test_gpu_variants.py) -------------------------
# k_scan_anchor has a template per item shape: TINY (items of one 1-KiB row at most, eight to a wave step), SHORT (the rows
# an item lacks are skipped) and BIG (pattern sets of more than 12,288 anchor q-grams: LDS bitmap + L2 table), which also
# combines with TINY.  The haystacks here have exactly those shapes, with plants that BEGIN at a row start, END at a row /
# item end and STRADDLE a row border (rows are the 1-KiB pieces a wave-load covers).

def _regular(name, seed, n_pat, lmin, lmax, n_hay, hay_len, alphabet, plant_at):
    """Regular haystacks; haystack h gets patterns planted at plant_at[(h + j) % len] for j = 0, 1 (skipped when None)."""
    ds = generate(name, seed, n_pat, lmin, lmax, n_hay, hay_len, alphabet)
    for h in range(n_hay):
        for j in range(2):
            at = plant_at[(h + 3 * j) % len(plant_at)]
            if at is None:
                continue
            pat = ds.pattern((11 * h + 5 * j + 2) % ds.n_patterns)
            a = at if at >= 0 else hay_len + at - len(pat) + 1   # negative: the pattern ENDS |at| - 1 bytes before the end
            a = max(0, min(a, hay_len - len(pat)))
            lo = int(ds.hay_off[h])
            ds.hay_blob[lo + a:lo + a + len(pat)] = np.frombuffer(pat, dtype=np.uint8)
    return ds


def t256() -> Dataset:
    """TINY: 2,048 patterns x 16 B over 640 haystacks of 256 B (regular, 16-byte aligned starts)."""
    return _regular("T256", 21, 2048, 16, 16, 640, 256, b"abcdef", [0, -1, 120, None, 233, 7])


def t512() -> Dataset:
    """TINY: the same pattern shape over 400 haystacks of 512 B."""
    return _regular("T512", 22, 2048, 16, 16, 400, 512, b"abcdef", [0, -1, 250, None, 496, 129])


def t1k() -> Dataset:
    """TINY at its limit: 288 haystacks of exactly one row (1,024 B)."""
    return _regular("T1K", 23, 2048, 16, 16, 288, 1024, b"abcdef", [0, -1, 504, 1001, None, 16])


def trag() -> Dataset:
    """TINY on ragged haystacks of 0..1,024 B (unaligned starts: scanned from the aligned address below, `floor`), plants at the
    very start and the very end of every haystack; some items exceed a row once the floor is added (the wave's usual path)."""
    return _planted_ragged("TRAG", 24, 2048, 16, 16, 700, 0, 1024, b"abcdef", 3)


def s3k() -> Dataset:
    """SHORT: 128 haystacks of three rows (3,072 B): plants that straddle a row border, begin at a row start, end at a row end."""
    return _regular("S3K", 25, 2048, 16, 16, 128, 3072, b"abcdef", [1016, 1024, 2032, -1, 0, 2040, 1008, None])


def s2k5() -> Dataset:
    """SHORT, unaligned: 150 haystacks of 2,500 B back to back (starts at every residue mod 16 that is a multiple of 4)."""
    return _regular("S2K5", 26, 2048, 16, 16, 150, 2500, b"abcdef", [0, -1, 1018, 1024, 2040, None, 2484])


def b512() -> Dataset:
    """BIG x TINY: 4,000 patterns of 9..24 B over 16 letters (16,000 anchor q-grams of 4 bytes: LDS bitmap + L2 table) over 360
    haystacks of 512 B."""
    return _regular("B512", 27, 4000, 9, 24, 360, 512, b"abcdefghijklmnop", [0, -1, 250, None, 488, 3])


def brag() -> Dataset:
    """BIG x TINY on ragged haystacks of 0..1,024 B."""
    return _planted_ragged("BRAG", 28, 4000, 9, 24, 500, 0, 1024, b"abcdefghijklmnop", 3)


def b3k() -> Dataset:
    """BIG with items of three rows (its tasks are 4 KiB: the quads an item lacks are skipped)."""
    return _regular("B3K", 29, 4000, 9, 24, 100, 3072, b"abcdefghijklmnop", [1016, 1024, 2030, -1, 0, None])


# ---- pattern sets whose shortest pattern is 4..6 B: the anchor filter's dense form (windows at every byte offset) ----------

def k1() -> Dataset:
    """Keyword list: 600 words of 4..12 letters over text of letters and spaces, ragged haystacks, planted (at the very start,
    the very end and inside): a few candidate windows per KiB, most of them true 4-letter prefixes of longer words."""
    return _planted_ragged("K1", 31, 600, 4, 12, 70, 0, 9000, bytes(range(97, 123)) + b"     ", 6)


def k2() -> Dataset:
    """Binary signatures of 4..9 B over Sigma=256, 2,500 of them; long ragged haystacks."""
    return _planted_ragged("K2", 32, 2500, 4, 9, 40, 0, 20000, None, 8)


def k3() -> Dataset:
    """80 patterns of exactly 5..6 B over 12 letters, nested plants (overlapping candidate windows: clusters), short haystacks
    (the tiny-item path of the dense form)."""
    ds = _planted_ragged("K3", 33, 80, 5, 6, 500, 0, 700, b"abcdefghijkl", 4)
    for h in range(0, ds.n_haystacks, 3):   # two patterns overlapping by two bytes, and one right behind the other
        lo, hi = int(ds.hay_off[h]), int(ds.hay_off[h + 1])
        a, b = ds.pattern((5 * h + 1) % ds.n_patterns), ds.pattern((7 * h + 2) % ds.n_patterns)
        if hi - lo >= 40:
            t = np.frombuffer(a + b[2:] + a + b, dtype=np.uint8)
            ds.hay_blob[lo + 9:lo + 9 + len(t)] = t
    return ds


# ---- a few short patterns among many long ones: the split set (engine.cpp build_parts) ------------------------------------

def _with_short(name, base: Dataset, shorts: list[bytes], every: int) -> Dataset:
    """`base`'s patterns with `shorts` spliced in between (so that caller-order indices interleave), its haystacks with a short
    pattern planted into every `every`-th one -- right behind a long plant where there is room (events of both parts at one
    position and at neighbouring ones)."""
    pats = base.patterns()
    step = max(1, len(pats) // (len(shorts) + 1))
    for i, sp in enumerate(shorts):
        pats.insert((i + 1) * step + i, sp)
    pat_blob, pat_off = pack(pats)
    ds = Dataset(name, pat_blob, pat_off, base.hay_blob.copy(), base.hay_off.copy())
    for h in range(0, ds.n_haystacks, every):
        lo, hi = int(ds.hay_off[h]), int(ds.hay_off[h + 1])
        sp = np.frombuffer(shorts[(h // every) % len(shorts)], dtype=np.uint8)
        if hi - lo >= 64:
            at = lo + (hi - lo) // 3
            ds.hay_blob[at:at + len(sp)] = sp
    return ds


def x1() -> Dataset:
    """2,048 patterns x 16 B over abcdef PLUS eight of 1..6 B, some over the same letters (they match by the dozen), some over
    others, one a suffix of a long pattern's tail, one a duplicate of another short one."""
    base = _planted_ragged("X1", 35, 2048, 16, 16, 90, 0, 5000, b"abcdef", 4)
    tail = base.pattern(7)[-3:]
    return _with_short("X1", base, [b"uvw", b"fedcba", tail, b"xy", b"abcab", b"z", b"uvw", b"ddddd"], 2)


def x2() -> Dataset:
    """5,000 patterns of 9..24 B over 16 letters (the big-set filter) plus three short ones."""
    base = _planted_ragged("X2", 36, 5000, 9, 24, 40, 0, 9000, b"abcdefghijklmnop", 6)
    return _with_short("X2", base, [b"qrs", b"ponm", b"ab"], 1)


TARGETED = {"T256": t256, "T512": t512, "T1K": t1k, "TRAG": trag, "S3K": s3k, "S2K5": s2k5, "B512": b512, "BRAG": brag, "B3K": b3k, "K1": k1, "K2": k2, "K3": k3, "X1": x1, "X2": x2}


# ---- match-density workloads (tests/test_gpu_cliffs.py, bench.py density_sweep) --------------------------------------------
def density_set(kind: str) -> Dataset:
    """Pattern sets of the density matrix with a 1 MiB tile of text without plants: "headline" (2,048 x 16 B over abcdef),
    "words" (2,048 words of 4..12 letters over letters and spaces), "cfg3" (65,536 patterns of 8..64 B, Sigma = 256)."""
    if kind == "headline":
        return generate("D-headline", 1, 2048, 16, 16, 128, 8192, b"abcdef")
    if kind == "words":
        letters = bytes(range(ord("a"), ord("z") + 1))
        text = generate("D-words", 33, 2048, 4, 12, 128, 8192, letters + b"     ")
        pats = generate("D-words", 34, 2048, 4, 12, 1, 16, letters)
        return Dataset("D-words", pats.pat_blob, pats.pat_off, text.hay_blob, text.hay_off)
    if kind == "cfg3":
        return generate("D-cfg3", 7, 65536, 8, 64, 128, 8192, None)
    raise ValueError(kind)


def plant_density(ds: Dataset, per_8k: int, hay_len: int, seed: int = 5) -> Dataset:
    """`ds`'s text tile cut into haystacks of `hay_len` bytes (a divisor or a multiple of 8 KiB) with `per_8k` occurrences of
    patterns of the set planted per 8 KiB of text, one per slot of 8192 / per_8k bytes, none across a haystack border."""
    blob = ds.hay_blob.copy()
    n = len(blob)
    slot = 8192 // per_8k
    lens = np.diff(ds.pat_off.astype(np.int64))
    fits = np.nonzero(lens <= min(slot, hay_len))[0]
    r = np.random.default_rng(seed)
    picks = fits[r.integers(0, len(fits), size=n // slot)]
    offs = r.integers(0, 1 << 30, size=n // slot)
    for k in range(n // slot):
        p = int(picks[k])
        ln = int(lens[p])
        at = k * slot + int(offs[k]) % (slot - ln + 1)
        cell = at // hay_len
        at = min(at, (cell + 1) * hay_len - ln)
        blob[at:at + ln] = ds.pat_blob[int(ds.pat_off[p]):int(ds.pat_off[p]) + ln]
    off = np.arange(n // hay_len + 1, dtype=np.uint64) * np.uint64(hay_len)
    return Dataset(f"{ds.name}-{per_8k}-{hay_len}", ds.pat_blob, ds.pat_off, blob, off)


def prose(ds: Dataset, share: float, seed: int = 11) -> Dataset:
    """Text of `ds` (8 KiB haystacks) of which a share is REPLACED by listed words, a space behind each (share 1: nothing but
    listed words): the word-list-over-prose workload of DESIGN 4.12 (tools/exp_kw_hits.py)."""
    blob = ds.hay_blob.copy()
    words = ds.patterns()
    r = np.random.default_rng(seed)
    for h in range(ds.n_haystacks):
        at, end = int(ds.hay_off[h]), int(ds.hay_off[h + 1])
        while at < end - 16:
            if r.random() < share:
                w = np.frombuffer(words[int(r.integers(len(words)))], dtype=np.uint8)
                blob[at:at + len(w)] = w
                blob[at + len(w)] = 32
                at += len(w) + 1
            else:
                at += 8
    return Dataset(f"{ds.name}-prose{share}", ds.pat_blob, ds.pat_off, blob, ds.hay_off.copy())

NAMED = {"G1": g1, "G2": g2, "G3": g3, "G4": g4, "G5": g5, "G6": g6, "R1": r1, "R2": r2, "R3": r3, "R4": r4, "R5": r5, "R6": r6, "R7": r7, "R8": r8}
