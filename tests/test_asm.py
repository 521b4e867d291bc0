"""The compiler's register report as a gate (tools/check_asm.py): no kernel the bench runs may have scratch (spill) traffic
inside its main loop.  Round 3's keyword filter reloaded 61 spilled registers per trip of its streaming loop and no check
said so.  CPU only: reads the device assembly the build keeps (php_aho_corasick_amd/csrc/build/*gfx950*.s)."""
import glob
import os
import sys

from helpers import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_asm  # noqa: E402


def _kernels():
    if not glob.glob(os.path.join(check_asm.BUILD, "*gfx950*.s")):
        from php_aho_corasick_amd import build
        build.build(force=True)
    return check_asm.collect()


def test_hot_kernels_have_no_scratch_traffic_in_their_main_loops():
    ks = _kernels()
    assert len(ks) >= 100                       # every kernel family's assembly was found
    bad = check_asm.violations(ks)
    assert not bad, "\n".join(bad)


def test_the_dense_anchor_form_does_not_spill():
    """All twelve DENSE instantiations spilled 4-15 VGPRs with eight quads per lane and task; with four they fit."""
    ks = [k for k in _kernels() if "k_scan_anchor<1, false" in k["pretty"] and k["pretty"].split(", ")[4] == "true"]
    assert len(ks) == 12
    assert all(k.get("vgpr_spill_count", 0) == 0 and k.get("private_segment_fixed_size", 0) == 0 for k in ks), [(k["pretty"], k.get("vgpr_spill_count")) for k in ks]
