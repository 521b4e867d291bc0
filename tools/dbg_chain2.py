import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from gpu_util import device_match, resource_for
from oracle.pyoracle import Oracle
from php_aho_corasick_amd import api, synth
base = synth.density_set("headline")
ds = synth.plant_density(base, 64, 8192)
n = 2
ds = synth.Dataset(ds.name, ds.pat_blob, ds.pat_off, ds.hay_blob[:int(ds.hay_off[n])], ds.hay_off[:n + 1])
o = Oracle("port"); o.add_patterns((ds.pat_blob, ds.pat_off))
pe, re_ = o.match_batch(ds.hay_blob, ds.hay_off, True)
res = resource_for(ds, kernel=6, small_path=0)
p, r = device_match(res, ds)
print("chain", res.stat("chain_walk_used"), "got", len(r), "want", len(re_))
print("want pos%128//64 histogram", np.bincount((re_[:, 1] - 1) % 128 // 64), "got", np.bincount((r[:, 1] - 1) % 128 // 64) if len(r) else None)
print("prefix got", p, "want", pe)
print(r[:8].tolist()); print(re_[:8].tolist())
got = set(map(tuple, r.tolist())); want = list(map(tuple, re_.tolist()))
miss = [w for w in want if w not in got]
print("missing", len(miss), "by hay", np.bincount([m[0] for m in miss]))
print("missing pieces (pos-1)//64:", sorted(set((m[1] - 1) // 64 for m in miss if m[0] == 0)))
print("present pieces:", sorted(set((w[1] - 1) // 64 for w in want if w in got and w[0] == 0)))
