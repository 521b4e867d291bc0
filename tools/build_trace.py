"""Phase times of the config-3 automaton build (65,536 patterns, 2.27 M states): ACMI_BUILD_TRACE=1 on stderr.
    python tools/build_trace.py [device_build=0|1] [other build options ...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['ACMI_BUILD_TRACE'] = '1'
from php_aho_corasick_amd import api, synth
opts = dict(a.split("=") for a in sys.argv[1:])
ds = synth.g6(n_hay=16, hay_len=4096)
t0 = time.time()
pats = [{"id": k, "value": ds.pattern(k)} for k in range(ds.n_patterns)]
t1 = time.time()
res = api.ahocorasick_init(pats)
t2 = time.time()
for k, v in opts.items():
    res.set_option(k, int(v))
api.ahocorasick_finalize(res)
t3 = time.time()
print(f"python list {t1-t0:.3f}s  init(add) {t2-t1:.3f}s  finalize {t3-t2:.3f}s", flush=True)
