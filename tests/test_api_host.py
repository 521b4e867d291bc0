"""The mirrored PHP function surface: argument validation, warnings, exceptions and resource
lifetime as in src/php_ahocorasick.c.  Everything here stops before a scan, so no GPU is needed."""
import warnings

import pytest

from php_aho_corasick_amd import api
from php_aho_corasick_amd.api import AhoException, AhoWarning


def _warns(fn, *a):
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = fn(*a)
    return out, [str(x.message) for x in w if issubclass(x.category, AhoWarning)]


def test_init_returns_resource_and_deinit_twice():
    res = api.ahocorasick_init([{"key": "ab", "value": "alfa"}, {"id": 3, "value": "beta"}, {"value": "x", "aux": [1]}])
    assert isinstance(res, api.AhoResource) and api.ahocorasick_isValid(res) is True
    assert api.ahocorasick_finalize(res) is True          # first call (php_ahocorasick.c:130-142)
    assert api.ahocorasick_finalize(res) is False         # already finalized
    assert api.ahocorasick_deinit(res) is True
    assert api.ahocorasick_isValid(res) is False          # tests/test2.phpt:295-298
    assert api.ahocorasick_deinit(res) is False
    assert api.ahocorasick_finalize(res) is False


def test_bare_numeric_key_is_value_and_keys_are_case_insensitive():
    res = api.ahocorasick_init([["plain"], {0: "zero"}, {"VALUE": "v", "Key": "k"}, {"Id": 9, "vAlUe": "w"}])
    assert res is not False and len(res._patterns) == 4
    assert res._patterns[2].key == "k" and res._patterns[3].key_id == 9
    api.ahocorasick_deinit(res)


@pytest.mark.parametrize("bad,msg", [
    ([{"value": "a"}, "not-an-array"], "Invalid pattern structure! Cannot initialize."),
    ([{"value": "a", "bogus": 1}], "unrecognized sub-array key"),
    ([{"key": "only-key"}], "No value was specified for pattern index: 0"),
    ([{"value": "a"}, {"key": "k", "id": 1, "value": "b"}], "either numeric or string identifier, not both! Pattern index: 1"),
])
def test_init_failures_warn_and_return_false(bad, msg):
    out, w = _warns(api.ahocorasick_init, bad)
    assert out is False and any(msg in m for m in w)


def test_ignorecase_is_deprecated_but_accepted():
    out, w = _warns(api.ahocorasick_init, [{"value": "a", "ignoreCase": True}])
    assert isinstance(out, api.AhoResource) and any("ignoreCase attribute is deprecated" in m for m in w)
    api.ahocorasick_deinit(out)


@pytest.mark.parametrize("bad,msg", [
    ([{"id": "7", "value": "a"}], "Invalid type of pattern ID given (long required), type: string, pattern index: 0"),
    ([{"value": "ok"}, {"value": 5}], "Pattern value has to be a string, type: long, pattern index: 1"),
    ([{"key": ["k"], "value": "a"}], "Pattern key has to be a string, type: array, pattern index: 0"),
    ([{"id": True, "value": "a"}], "type: true"),
])
def test_type_errors_throw_ahoexception(bad, msg):
    with pytest.raises(AhoException) as e:
        api.ahocorasick_init(bad)
    assert msg in str(e.value)


def test_add_patterns_rules():
    res = api.ahocorasick_init([])
    assert api.ahocorasick_add_patterns(res, [{"value": "a"}]) is True
    out, w = _warns(api.ahocorasick_add_patterns, res, [{"key": "x"}])      # bad call adds nothing
    assert out is False and len(res._patterns) == 1
    assert api.ahocorasick_finalize(res) is True
    out, w = _warns(api.ahocorasick_add_patterns, res, [{"value": "b"}])
    assert out is False and any("finalized search structure" in m for m in w)   # php_ahocorasick.c:914-917
    api.ahocorasick_deinit(res)
    out, w = _warns(api.ahocorasick_add_patterns, res, [{"value": "b"}])
    assert out is False and any("not initialized" in m for m in w)             # :909-912


def test_match_on_bad_resource_warns():
    out, w = _warns(api.ahocorasick_match, "text", None)
    assert out is False and w == ["Invalid resource."]                         # php_ahocorasick.c:696-699
    res = api.ahocorasick_init([{"value": "a"}])
    api.ahocorasick_deinit(res)
    out, w = _warns(api.ahocorasick_match, "text", res)
    assert out is False and w == ["Invalid resource."]
    out, w = _warns(api.ahocorasick_match_batch, ["text"], res)
    assert out is False and w == ["Invalid resource."]


def test_non_list_arrays_are_rejected_cleanly():
    """The reference NULL-derefs on arrays without key 0; we refuse them."""
    out, w = _warns(api.ahocorasick_init, {5: {"value": "a"}})
    assert out is False and w
