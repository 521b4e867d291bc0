"""php_aho_corasick_amd -- MI355X-native drop-in for the ac_trie_search() hot path of
ph4r05/php_aho_corasick.

Layout (only what the path needs):
  csrc/        host automaton builder, HIP kernels (gfx950), C ABI  -> libacmi.so
  _native.py   ctypes binding of include/acmi.h
  api.py       mirror of the reference's PHP function surface (ahocorasick_init, ...)
  batch.py     device-resident batch entry + multi-GPU sharding (torch only as plumbing)
  synth.py     seeded synthetic datasets
  build.py     hipcc build driver

The PHP-surface names are importable from the package root; importing them loads
libacmi.so and fails loudly if it has not been built.
"""
from __future__ import annotations

_API = ("AhoException", "AhoWarning", "AhoResource", "ahocorasick_init", "ahocorasick_add_patterns",
        "ahocorasick_finalize", "ahocorasick_match", "ahocorasick_match_batch", "ahocorasick_isValid",
        "ahocorasick_deinit", "match_arrays", "AhoStream")

__version__ = "0.3.0"


def __getattr__(name):
    if name in _API:
        from . import api
        return getattr(api, name)
    raise AttributeError(name)


__all__ = list(_API)
