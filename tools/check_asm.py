"""Register / scratch report of the gfx950 code objects, and the gate tests/test_asm.py applies to it.

    python tools/check_asm.py [--all]        table of every kernel: registers, spills, scratch ops in its loops

The build keeps the device assembly of every kernel file (php_aho_corasick_amd/csrc/build/*gfx950*.s, hipcc -save-temps).  For
each kernel this finds its loops (a label that a later conditional branch jumps back to), counts the scratch_load / scratch_store
instructions inside them, and names the HOT loop: the innermost loop with the most instructions.  A kernel listed in HOT below
must have no scratch traffic in its hot loop (round 3's keyword filter reloaded 61 spilled registers per trip of its main loop
and nobody had looked at the compiler's report)."""
from __future__ import annotations

import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "php_aho_corasick_amd", "csrc", "build")

# kernels whose main loop the bench and the legs run: (substring of the demangled name, scratch ops allowed per trip of the hot loop)
HOT = [
    ("k_scan_anchor<3, false, false, false, false, false, false>", 0),   # the headline kernel
    ("k_scan_anchor<1, true, false, false, false, false, false>", 0),    # config 3
    ("k_scan_anchor<1, false, false, false, true, true, true>", 0),      # keyword_set / keyword_words (dense form, HOT, two slots)
    ("k_scan_anchor<1, false, false, false, true, false, true>", 0),
    ("k_scan_lds<unsigned short, 1, 1, 8>", 0),                          # general_path
    # k_walk_tiles marks its step loop in the source (the per-TILE code around it -- piece set-up, scans, look-back, once per
    # 64 x 69 byte steps -- reloads the dozen registers spilled at 64 registers = 8 waves per SIMD, which is what makes it fast)
    ("k_walk_tiles<unsigned short, 1, true, true, false>", 0),                  # short patterns over their own alphabet
    ("k_walk_tiles<unsigned short, 1, true, false, false>", 2),                 # (the wave-queue variants reload a pointer pair or two per
    ("k_walk_tiles<unsigned short, 0, false, false, true>", 4),                # block of sixteen steps, ~ 1,000-4,000 instructions: word lists,
    ("k_walk_tiles<unsigned short, 1, false, false, true>", 4),                # the headline set at many matches per haystack)
    ("k_walk_chain<1, false, true, 2u>", 0),                                              # the whole automaton in LDS: the headline set at many matches per haystack
    ("k_walk_chain<0, false, false, 2u>", 0),                                              # ... word lists over prose
    ("k_walk_chain<1, true, true, 2u>", 0),                                               # small automata, every row in LDS: short patterns over their own alphabet
    ("k_walk_chain<0, true, true, 2u>", 0),
    ("k_walk_items<unsigned short, 0, 4, false, true>", 0),
    ("k_refill_wave<unsigned short, 1024u>", 0),                         # config 5
    ("k_walk_runs<unsigned short, 1, false>", 0),
    ("k_replace_emit", 0),
    ("k_runs_wave<unsigned int, 0, true>", 0),        # config 3 with candidates by the dozen per item (round 6)
    ("k_runs_wave<unsigned short, 1, false>", 0),     # ... the headline set
    ("k_expand_dense", 0),
    ("k_walk_win<unsigned int>", 0),                  # the table walk of big sets over the window table (round 6)
]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def parse(path):
    """-> list of dicts per kernel: name, vgpr, sgpr, vgpr_spill, sgpr_spill, scratch_bytes, loops [(n_instr, n_scratch, depth)]"""
    text = open(path).read().splitlines()
    meta = {}
    cur = None
    for line in text:   # the metadata block at the end of the file
        m = re.match(r"\s+\.name:\s+(\S+)", line)
        if m:
            cur = meta.setdefault(m.group(1), {})
            continue
        m = re.match(r"\s+\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):\s+(\d+)", line)
        if m and cur is not None:
            cur[m.group(1)] = int(m.group(2))
    kernels = []
    start = {}
    for i, line in enumerate(text):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m and m.group(1) in meta:
            start[m.group(1)] = i
    for name, i0 in start.items():
        i1 = i0
        while i1 < len(text) and "s_endpgm" not in text[i1]:
            i1 += 1
        body = text[i0:i1 + 1]
        labels = {}
        instr_at = []   # cumulative instruction count per line
        n = 0
        for k, line in enumerate(body):
            s = line.strip()
            m = re.match(r"^(\.LBB\w+):", s)
            if m:
                labels[m.group(1)] = k
            elif s and not s.startswith(";") and not s.startswith(".") and not s.endswith(":"):
                n += 1
            instr_at.append(n)
        loops = []
        for k, line in enumerate(body):
            m = re.match(r"\s*s_cbranch_\w+\s+(\.LBB\w+)", line)
            if m and m.group(1) in labels and labels[m.group(1)] < k:
                loops.append((labels[m.group(1)], k))
        # per loop: instructions and scratch ops of its OWN body, i.e. not inside a loop nested in it (the streaming loop of a
        # scan kernel is one long body around a few short inner loops); hot = the loop with the longest own body
        is_instr = [bool(l.strip()) and not l.strip().startswith((";", ".")) and not l.strip().endswith(":") for l in body]
        is_scratch = [bool(re.match(r"\s*scratch_(load|store)", l)) for l in body]
        stats = []
        for lo, hi in loops:
            inner = [(a, b) for a, b in loops if (a, b) != (lo, hi) and lo <= a and b <= hi]
            covered = [False] * (hi - lo + 1)
            for a, b in inner:
                for t in range(a - lo, b - lo + 1):
                    covered[t] = True
            own_i = sum(1 for t in range(lo, hi + 1) if is_instr[t] and not covered[t - lo])
            own_s = sum(1 for t in range(lo, hi + 1) if is_scratch[t] and not covered[t - lo])
            stats.append((lo, hi, own_i, own_s))
        hot = max(stats, key=lambda l: l[2]) if stats else None
        # a kernel may mark its step loop itself ("; acmi-hot-begin" ... "; acmi-hot-end" from an asm volatile comment): the marked
        # lines then are its hot loop (k_walk_tiles: the longest loop body is its per-tile set-up, not its steps)
        marks = [t for t, l in enumerate(body) if "acmi-hot-begin" in l]
        if marks:   # (the compiler rotates loops: the smallest loop around the marker is the marked one, whatever the layout)
            around = [(lo, hi) for lo, hi in loops if lo <= marks[0] <= hi]
            if around:
                lo, hi = min(around, key=lambda l: l[1] - l[0])
                hot = (lo, hi, sum(1 for t in range(lo, hi + 1) if is_instr[t]), sum(1 for t in range(lo, hi + 1) if is_scratch[t]))
        kernels.append(dict(name=name, file=os.path.basename(path), loops=stats, hot=hot, **meta[name]))
    return kernels


def collect():
    files = sorted(glob.glob(os.path.join(BUILD, "*gfx950*.s")))
    ks = []
    for f in files:
        ks += parse(f)
    names = demangle([k["name"] for k in ks])
    for k in ks:
        k["pretty"] = re.sub(r"\(acmi::DevAutomaton.*", "", names.get(k["name"], k["name"])).replace("void acmi::", "")
    return ks


def violations(ks):
    bad = []
    for pat, allowed in HOT:
        hits = [k for k in ks if pat in k["pretty"]]
        if not hits:
            bad.append(f"hot kernel not found in the build: {pat}")
        for k in hits:
            if k["hot"] and k["hot"][3] > allowed:
                bad.append(f"{k['pretty']}: {k['hot'][3]} scratch ops per trip of its hot loop ({k['hot'][2]} instructions), {k.get('vgpr_spill_count', 0)} VGPRs spilled")
    return bad


if __name__ == "__main__":
    ks = collect()
    if not ks:
        sys.exit("no assembly under " + BUILD + ": python -m php_aho_corasick_amd.build --force")
    show_all = "--all" in sys.argv
    print(f"{'kernel':92s} {'vgpr':>4s} {'sgpr':>4s} {'vspill':>6s} {'sspill':>6s} {'scratchB':>8s}  hot loop: instr / scratch ops")
    for k in sorted(ks, key=lambda k: k["pretty"]):
        if not show_all and not k.get("vgpr_spill_count") and not (k["hot"] and k["hot"][3]):
            continue
        h = k["hot"]
        print(f"{k['pretty'][:92]:92s} {k.get('vgpr_count', 0):4d} {k.get('sgpr_count', 0):4d} {k.get('vgpr_spill_count', 0):6d} {k.get('sgpr_spill_count', 0):6d} "
              f"{k.get('private_segment_fixed_size', 0):8d}  {h[2] if h else 0:6d} / {h[3] if h else 0}")
    bad = violations(ks)
    print("\n".join(["", "VIOLATIONS:"] + bad) if bad else "\nno hot kernel has scratch traffic in its hot loop")
    sys.exit(1 if bad else 0)
