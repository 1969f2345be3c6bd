"""Prototype of a placement search: NSETS separately allocated output sets, time each, then greedy swaps of single arrays
between sets.  Reports what a 'best of pool' strategy would achieve in this process."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crt1d_amd import _lib, batched, synth  # noqa: E402

scheme = sys.argv[1] if len(sys.argv) > 1 else "2s"
nsets = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ncol, nb, nz = 10000, 300, 60
d = synth.make_columns(ncol, nb, nz)
cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
base = batched.Plan(scheme, cols, bands)
base(); torch.cuda.synchronize()
keys = list(base.out.keys())
st = torch.cuda.current_stream()
ntime = 0


def timeit(out):
    global ntime
    ntime += 1
    plan = batched.Plan(scheme, cols, bands, out=out, workspace=base.workspace)
    plan(flags=_lib.FLAG_SKIP_PRECOMPUTE); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(4):
        plan(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 4


t00 = time.time()
import random
random.seed(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
pads = []
sets = [base.out]
for i in range(nsets - 1):
    pads.append(torch.empty(random.randrange(1, 150) * 2**21, dtype=torch.uint8, device="cuda"))  # shifts where the next set lands
    sets.append({k: torch.empty_like(v) for k, v in base.out.items()})
times = [timeit(s) for s in sets]
print("sets:", " ".join(f"{t:.3f}" for t in times), flush=True)
bi = min(range(nsets), key=lambda i: times[i])
best, tbest = dict(sets[bi]), times[bi]
for k in keys:
    for j in range(nsets):
        if sets[j][k] is best[k]:
            continue
        cand = dict(best); cand[k] = sets[j][k]
        if len({v.data_ptr() for v in cand.values()}) < len(keys):
            continue
        t = timeit(cand)
        if t < 0.985 * tbest:
            best, tbest = cand, t
print(f"after greedy swaps: {tbest:.3f} ms ({ntime} timings, {time.time() - t00:.2f} s)", flush=True)
# random 4-subsets of the pool for comparison
pool = [s[k] for s in sets for k in keys if s[k].shape == base.out[keys[0]].shape]
res = []
for _ in range(20):
    pick = random.sample(pool, len(keys))
    if any(p.shape != base.out[k].shape for p, k in zip(pick, keys)):
        continue
    res.append(timeit(dict(zip(keys, pick))))
print("random subsets of the pool:", " ".join(f"{t:.3f}" for t in sorted(res)), flush=True)
