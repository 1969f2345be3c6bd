"""Large output sets run the solve kernels at ~0.83 of the peak where 6-9 GB sets run at 0.87-0.88 (store-only probe: 0.87-0.89 on both).
Is it the SET (addresses, placement) or the LAUNCH (number of workgroups, time in flight)?  One 125 000-column set; the kernel launched
on sub-ranges of it (views into the same arrays), and on the whole.

    python tools/subrange_probe.py [scheme]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from crt1d_amd import _lib, batched, synth


def timed(fn, n, st):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    scheme = sys.argv[1] if len(sys.argv) > 1 else "2s"
    nb, nz, big = 300, 60, 125000
    st = torch.cuda.current_stream()
    bps = bench.bytes_per_solve(scheme, nz, 8)
    d = synth.make_columns(big, nb, nz, seed=1234)
    cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
    pb = batched.Plan(scheme, cols, bands)
    pb()
    torch.cuda.synchronize()
    print("classes", {k: v for k, v in pb.placement.items()} if hasattr(pb, "placement") else "")
    ms = timed(lambda: pb(st, flags=_lib.FLAG_SKIP_PRECOMPUTE), 10, st)
    print(f"whole set {big} columns: {ms:.3f} ms  frac {bps * big * nb / (ms * 1e-3) / 8e12:.3f}", flush=True)
    shared = batched.Bands(*[None if t is None else t[0].contiguous() for t in (bands.I_dr0, bands.I_df0, bands.leaf_r, bands.leaf_t, bands.soil_r)])
    ps = batched.Plan(scheme, cols, shared, out=pb.out)
    ps()
    ms = timed(lambda: ps(st, flags=_lib.FLAG_SKIP_PRECOMPUTE), 10, st)
    print(f"whole set, ONE spectrum shared by all columns (no per-column input rows to read): {ms:.3f} ms  frac {bps * big * nb / (ms * 1e-3) / 8e12:.3f}", flush=True)
    del ps
    # (0) the per-column spectra in memory from the set allocator (512 MB physical chunks) instead of torch's
    names5 = ("I_dr0", "I_df0", "leaf_r", "leaf_t", "soil_r")
    bufs = batched.device_buffers([(big, nb)] * 5)
    for t, k in zip(bufs, names5):
        t.copy_(getattr(bands, k))
    pc = batched.Plan(scheme, cols, batched.Bands(*bufs), out=pb.out)
    pc()
    ms = timed(lambda: pc(st, flags=_lib.FLAG_SKIP_PRECOMPUTE), 10, st)
    print(f"whole set, spectra in set-allocator memory: {ms:.3f} ms  frac {bps * big * nb / (ms * 1e-3) / 8e12:.3f}", flush=True)
    del pc, bufs
    # (1) the 1e4-column sub-range again, with the Infinity Cache flushed between launches (a 1 GB read): cold inputs on a small set
    n, c0 = 10000, 60000
    sub = {k: getattr(cols, k)[c0:c0 + n] for k in ("psi", "lai", "g_kind", "g_param", "mla") if getattr(cols, k, None) is not None}
    p = batched.Plan(scheme, batched.Columns(**sub), bands.slice(c0, c0 + n), out={k: v[c0:c0 + n] for k, v in pb.out.items()})
    p()
    junk = torch.empty(1 << 27, dtype=torch.float64, device="cuda").zero_()
    for flush in (False, True):
        ts = []
        for _ in range(12):
            if flush:
                junk.sum()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            p(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)
            e1.record(st)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = sorted(ts)[len(ts) // 2]
        print(f"columns [{c0}, {c0 + n}) single launches, {'1 GB read between launches (inputs evicted)' if flush else 'back to back (inputs stay cached)'}: "
              f"{ms:.3f} ms  frac {bps * n * nb / (ms * 1e-3) / 8e12:.3f}", flush=True)
    del p, junk
    # (2) the whole set as blocks of 16 000 columns, each block's input rows read in one burst (torch reductions) before its solve
    blk = 16000
    plans = []
    for c0 in range(0, big, blk):
        c1 = min(big, c0 + blk)
        sub = {k: getattr(cols, k)[c0:c1] for k in ("psi", "lai", "g_kind", "g_param", "mla") if getattr(cols, k, None) is not None}
        bs = bands.slice(c0, c1)
        pl = batched.Plan(scheme, batched.Columns(**sub), bs, out={k: v[c0:c1] for k, v in pb.out.items()})
        pl()
        rows = [t for t in (bs.I_dr0, bs.I_df0, bs.leaf_r, bs.leaf_t, bs.soil_r) if t is not None]
        plans.append((pl, rows))
    sink = torch.zeros((), dtype=torch.float64, device="cuda")

    import ctypes
    tp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "variants", "libtouch_probe.so")
    touch = ctypes.CDLL(tp).touch_range if os.path.exists(tp) else None
    if touch is not None:
        touch.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]

    def blocked(warm):
        for pl, rows in plans:
            if warm == 1:
                for t in rows:
                    sink.add_(t.sum())
            elif warm == 2:
                for t in rows:
                    touch(t.data_ptr(), t.numel() * 8, sink.data_ptr(), st.cuda_stream)
            pl(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)

    for warm in (0, 1, 2, 0, 1, 2):
        if warm == 2 and touch is None:
            continue
        ms = timed(lambda: blocked(warm), 5, st)
        print(f"whole set as {len(plans)} blocks of {blk}, {('no warm-up', 'inputs read before each block (torch sum)', 'inputs read before each block (plain loads, own kernel)')[warm]}: {ms:.3f} ms  "
              f"frac {bps * big * nb / (ms * 1e-3) / 8e12:.3f}", flush=True)
    # (2b) the read phase of block b + 1 on a SECOND stream while block b is being solved (ordered by events)
    if touch is not None:
        st2 = torch.cuda.Stream()
        evs = [torch.cuda.Event() for _ in plans]

        def warm(i):
            for t in plans[i][1]:
                touch(t.data_ptr(), t.numel() * 8, sink.data_ptr(), st2.cuda_stream)
            evs[i].record(st2)

        def overlapped():
            st2.wait_stream(st)
            warm(0)
            for i, (pl, rows) in enumerate(plans):
                st.wait_event(evs[i])
                if i + 1 < len(plans):
                    warm(i + 1)
                pl(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)

        for _ in range(2):
            ms = timed(overlapped, 5, st)
            print(f"whole set as {len(plans)} blocks of {blk}, inputs of block b+1 read on a second stream while block b is solved: {ms:.3f} ms  "
                  f"frac {bps * big * nb / (ms * 1e-3) / 8e12:.3f}", flush=True)
    # (3) every block solved twice in a row: is the second launch of a block (inputs just read by the first) the fast one?
    first, second = [], []
    for _ in range(3):
        for pl, rows in plans:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record(st)
            pl(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)
            ev[1].record(st)
            pl(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)
            ev[2].record(st)
            torch.cuda.synchronize()
            if rows[0].shape[0] == blk:
                first.append(ev[0].elapsed_time(ev[1]))
                second.append(ev[1].elapsed_time(ev[2]))
    for tag, v in (("first (inputs cold)", first), ("second (inputs just read)", second)):
        ms = sorted(v)[len(v) // 2]
        print(f"blocks of {blk} solved twice in a row, {tag}: {ms:.3f} ms  frac {bps * blk * nb / (ms * 1e-3) / 8e12:.3f}", flush=True)
    del plans
    # (4) which side has to be "recently used": the inputs or the outputs?  Blocks A, B, C of 16 000 columns; launches alternate between
    #     A->A / A->B / A->C (same inputs, 28 GB of outputs in turn) and A->A / B->A / C->A (same outputs, 650 MB of inputs in turn)
    def mk(ci, co):
        sub = {k: getattr(cols, k)[ci:ci + blk] for k in ("psi", "lai", "g_kind", "g_param", "mla") if getattr(cols, k, None) is not None}
        pl = batched.Plan(scheme, batched.Columns(**sub), bands.slice(ci, ci + blk), out={k: v[co:co + blk] for k, v in pb.out.items()})
        pl()
        return pl
    for tag, trio in (("same inputs, outputs alternate over 3 blocks", [mk(0, 0), mk(0, 40000), mk(0, 80000)]),
                      ("same outputs, inputs alternate over 3 blocks", [mk(0, 0), mk(40000, 0), mk(80000, 0)]),
                      ("same inputs and outputs", [mk(0, 0)] * 3)):
        def go():
            for pl in trio:
                pl(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)
        ms = timed(go, 6, st) / 3
        print(f"blocks of {blk}: {tag}: {ms:.3f} ms  frac {bps * blk * nb / (ms * 1e-3) / 8e12:.3f}", flush=True)
        del trio
    for n in (10000, 29000):
        for c0 in sorted({0, (big - n) // 2 // 1000 * 1000, big - n}):
            sub = {k: getattr(cols, k)[c0:c0 + n] for k in ("psi", "lai", "g_kind", "g_param", "mla") if getattr(cols, k, None) is not None}
            cs = batched.Columns(**sub)
            p = batched.Plan(scheme, cs, bands.slice(c0, c0 + n), out={k: v[c0:c0 + n] for k, v in pb.out.items()})
            p()
            ms = timed(lambda: p(st, flags=_lib.FLAG_SKIP_PRECOMPUTE), max(4, 200000 // n), st)
            print(f"columns [{c0}, {c0 + n}): {ms:.3f} ms  frac {bps * n * nb / (ms * 1e-3) / 8e12:.3f}", flush=True)
            del p


if __name__ == "__main__":
    main()
