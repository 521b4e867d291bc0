"""the Zend shim (php_ext/php_ahocorasick_amd.c) built against libacmi.so and the REFERENCE's own .phpt files run
through it -- wherever a box has what that takes: phpize / php-config / php (php-dev), a checkout of ph4r05/php_aho_corasick
(ACMI_REF_CHECKOUT, default /root/reference: its tests/ are read in place, never copied into this repo) and an MI355X.
None of this repo's build images has PHP, so the test skips there and says why; on a box that has it, a failing .phpt fails
the suite: the drop-in claim is then checked literally (reference: config.m4:1-10, tests/test1.phpt ... test6.phpt,
src/php_ahocorasick.c:664-746)."""
import os
import shutil
import subprocess

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu


def _missing():
    need = [t for t in ("phpize", "php-config", "php") if shutil.which(t) is None]
    ref = os.environ.get("ACMI_REF_CHECKOUT", "/root/reference")
    if need:
        return None, "no PHP toolchain on this box (" + ", ".join(need) + " not found): the Zend shim cannot be compiled here"
    if not os.path.isdir(os.path.join(ref, "tests")):
        return None, f"no reference checkout at {ref} (set ACMI_REF_CHECKOUT): its .phpt files are used in place, not copied"
    return ref, None


def test_zend_shim_builds_and_passes_the_reference_phpt_files():
    ref, why = _missing()
    if why:
        pytest.skip(why)
    out = subprocess.run(["bash", os.path.join(ROOT, "php_ext", "build_and_test.sh"), ref], capture_output=True, text=True, timeout=1800)
    tail = out.stdout[-4000:] + out.stderr[-2000:]
    assert out.returncode == 0, tail
    # run-tests.php's summary: every reference test (test1..test6) and php_ext/tests/match_batch.phpt ran and none failed
    assert "Tests failed    :    0" in out.stdout or "FAILED TEST SUMMARY" not in out.stdout, tail
    passed = [l for l in out.stdout.splitlines() if l.startswith("PASS ")]
    assert len(passed) >= 7, tail


def test_the_skip_reason_names_what_is_missing():
    """(so that a log reader sees WHY f-1 is unchecked on this box, not just an 's')"""
    ref, why = _missing()
    assert (ref is None) == (why is not None)
    if why:
        assert "PHP toolchain" in why or "reference checkout" in why
