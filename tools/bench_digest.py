"""One screen of a bench.py line: python tools/bench_digest.py <file with the JSON line>"""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d["roofline"]
print("headline", d["value"], "GB/s", d["ms_per_step"], "ms  frac", r["frac"], "whole step", r["frac_whole_step"], "step-kernels us", r["step_minus_kernels_us"],
      "kernel ms", r["kernel_ms"], "walk", r["walk_kernel_ms"], "ok", d["check_ok"])
print("cpu 1 core", (d.get("cpu_baseline") or {}).get("value"), "all", (d.get("cpu_baseline_all_cores") or {}).get("value"), "cores", (d.get("cpu_baseline_all_cores") or {}).get("cores"))
g = lambda k, f="frac_whole_call": (d.get(k) or {}).get(f)
print("config5_dense", g("config5_dense", "frac_of_hbm_peak"), "general", g("general_path"), "replace", g("replace", "frac_of_hbm_peak"),
      "kw_set", g("keyword_set"), "kw_words", g("keyword_words"), "config3", g("config3"))
for k, v in (d.get("short_patterns") or {}).items():
    if isinstance(v, dict): print("short_patterns", k, v.get("ms_per_call"), v.get("frac_whole_call"), v.get("kernel"))
for s in d.get("short_haystacks") or []:
    print("short_haystacks", s["haystack_bytes"], s["ms_per_call"], s["frac_whole_call"], s.get("uniform_entry"))
for b in d.get("big_dictionary") or []:
    print("dictionary", b["words"], b["ms_per_call"], b["kernel"], b["frac_whole_call"], "floor/measured", b["floor_over_measured"], b["check_ok"])
ds = d.get("density_sweep")
if ds:
    for c in ds["cells"]:
        print("density", c["set"], c["records_per_8KiB"], c["ms_per_call"], c["kernel"], c["frac_whole_call"], c["check_ok"])
    print({k: v for k, v in ds.items() if k != "cells"})
cm = d.get("choice_matrix")
if cm: print("choice_matrix", {k: v for k, v in cm.items() if k != "cells"})
h = d.get("host_entry") or {}
print("host", {k: h.get(k) for k in ("single_match_8KiB_us", "single_match_8KiB_us_c_abi_only", "single_match_8KiB_kernel_us", "literal_256x8KiB_one_call_ms")})
b = d.get("benchmark_php_sample") or {}
print("php sample", {k: (v.get("s_per_sample") if isinstance(v, dict) else v) for k, v in b.items() if k in ("cpu", "per_call", "one_match_batch")})
c3 = d.get("config3") or {}
print("config3 cpu", c3.get("cpu_baseline"))
