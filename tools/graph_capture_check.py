import sys
sys.path.insert(0, "/root/repo")
import torch
from crt1d_amd import batched, synth
for scheme, shape in (("2s", (2000, 300, 60)), ("zq", (500, 300, 100)), ("n79", (300, 107, 60)), ("4s", (64, 38, 60))):
    d = synth.make_columns(*shape, seed=3)
    cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
    plan = batched.Plan(scheme, cols, bands, placement="none")
    plan(); torch.cuda.synchronize()
    ref = {k: v.clone() for k, v in plan.out.items()}
    for v in plan.out.values(): v.zero_()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        plan()  # warm on the side stream
    torch.cuda.synchronize()
    try:
        with torch.cuda.graph(g, stream=s):
            plan()
    except Exception as e:
        print(scheme, "capture failed:", repr(e)[:300]); continue
    for v in plan.out.values(): v.zero_()
    g.replay(); torch.cuda.synchronize()
    ok = all(torch.equal(plan.out[k], ref[k]) for k in ref)
    # timing: 200 replays vs 200 plain calls (small problem -> launch-bound)
    import time
    t0 = time.perf_counter()
    for _ in range(200): g.replay()
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 200
    t0 = time.perf_counter()
    for _ in range(200): plan()
    torch.cuda.synchronize(); tp = (time.perf_counter() - t0) / 200
    print(scheme, shape, "graph replay equal:", ok, f"replay {tg*1e3:.4f} ms/step, plain {tp*1e3:.4f} ms/step")
