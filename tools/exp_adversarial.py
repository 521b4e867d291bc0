import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from gpu_util import host_match, resource_for
from php_aho_corasick_amd import api, synth
ds = synth.g2(256)
frag = np.frombuffer(ds.pattern(5)[:15] * 600, dtype=np.uint8)[:8192]
ds.hay_blob[:] = np.tile(frag, 256)
res = resource_for(ds)
res.set_option("profile", 2)
used = []
for i in range(40):
    p, r = host_match(res, ds)
    used.append((int(res.stat("kernel_used")), round(res.stat("total_ms"), 3), int(res.stat("max_windows")), int(res.stat("chunk_bytes")), int(res.stat("wave_runs_used"))))
print(used)
