"""Pairs of a clip are independent: throughput of K engines on K HIP streams (round robin over the pairs) vs one engine on one
stream, 1080p, 2x.  usage: overlap_probe.py [m2m|film|gmfss|ifunet|ifrnet ...] [--k 1,2,3] [--node N]
--node N: the node loop itself (m2m.run_plan: host clip of N frames in, host tensor out, PCIe and host copies included) with a LaneSet
of K lanes, instead of the device-resident pair loop."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

if "--opt" in sys.argv:      # --opt name=v0,v1,...: same-process A/B of a library test option (libvfi_hip_test.so), every K under every value
    ge.load_package()
    from cfi_amd import _lib as _vfi_lib  # noqa: E402

    _vfi_lib.use_test_build()
ge.build()
ge.load_package()
from cfi_amd import synth  # noqa: E402


def make(model):
    if model == "m2m":
        from cfi_amd.m2m import M2MEngine
        sd = synth.m2m_synth_state_dict(1234)
        return lambda: M2MEngine(sd)
    if model == "film":
        from cfi_amd.film import FilmEngine
        sd = synth.film_synth_state_dict(1234)
        return lambda: FilmEngine(sd)
    if model == "gmfss":
        from cfi_amd.gmfss import GMFSSEngine
        sds = synth.gmfss_coherent_state_dicts(3, "union")
        return lambda: GMFSSEngine(sds)
    if model == "ifunet":
        from cfi_amd.ifunet import IFUNetEngine
        sd = synth.ifunet_synth_state_dict(1234)
        return lambda: IFUNetEngine(sd)
    if model == "ifrnet":
        from cfi_amd.ifrnet import IFRNetEngine
        sd = synth.ifrnet_synth_state_dict("L", 1234)
        return lambda: IFRNetEngine(sd, "L")
    raise SystemExit("unknown model " + model)


def step(model, eng, x0, x1, out):
    if model == "film":
        return eng.forward(x0, x1)
    eng.prepare(x0, x1)
    if model == "ifrnet":
        eng.render(1.0, out)
    else:
        eng.render(0.5, out)
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    ks = [1, 2, 3]
    if "--k" in sys.argv:
        ks = [int(v) for v in sys.argv[sys.argv.index("--k") + 1].split(",")]
    args = [a for a in args if not a[0].isdigit()]
    H, W = 1080, 1920
    opt_name, opt_vals = None, [None]
    if "--opt" in sys.argv:
        spec = sys.argv[sys.argv.index("--opt") + 1]
        opt_name, vals = spec.split("=")
        opt_vals = [int(v) for v in vals.split(",")]
        args = [a for a in args if "=" not in a]
    for model, opt_val in [(m, v) for m in (args or ["m2m"]) for v in opt_vals]:
        if opt_name is not None:
            from cfi_amd import _lib
            assert _lib.load().vfi_test_set_option(opt_name.encode(), opt_val) == 0
            print(f"--- {opt_name} = {opt_val}", flush=True)
        fr = synth.texture_frames(2, H, W, seed=5) if model == "gmfss" else synth.smooth_frames(2, H, W, seed=2, shift=4.0)
        x0, x1 = fr[0].cuda().contiguous(), fr[1].cuda().contiguous()
        factory = make(model)
        base = None
        if "--node" in sys.argv and model != "film":
            from cfi_amd.lanes import LaneSet
            from cfi_amd.m2m import run_plan
            from cfi_amd.schedule import generic_output_plan
            n = int(sys.argv[sys.argv.index("--node") + 1])
            clip = fr[torch.arange(n) % 2].contiguous()
            plan, tasks = generic_output_plan(n, 2, None)
            ref = None
            for K in ks:
                lanes = LaneSet(factory, K)
                if model == "ifrnet":
                    lanes.configure(lambda e: setattr(e, "embt", 1.0))
                best = 1e9
                for rep in range(3):
                    t0 = time.perf_counter()
                    got = run_plan(lanes, clip, plan, tasks)
                    best = min(best, time.perf_counter() - t0)
                ref = got if ref is None else ref
                base = base or best
                print(f"{model} node loop, {n} frames 1080p: K = {K}: {best * 1e3:.1f} ms -> {(n - 1) / best:.1f} interpolated frames/s ({base / best:.3f}x), "
                      f"identical to K = {ks[0]}: {torch.equal(ref, got)}", flush=True)
                lanes.close()
                del lanes
                torch.cuda.empty_cache()
            continue
        dma_stop = None
        if "--dma" in sys.argv:      # pinned H2D + D2H of one frame each, back to back on two side streams, while the pairs run
            import threading
            hbuf = [torch.empty(H, W, 3, pin_memory=True) for _ in range(2)]
            dbuf = [torch.empty(H, W, 3, device="cuda") for _ in range(2)]
            s_up, s_dn = torch.cuda.Stream(), torch.cuda.Stream()
            dma_stop = threading.Event()

            def dma():
                n = 0
                while not dma_stop.is_set():
                    with torch.cuda.stream(s_up):
                        dbuf[0].copy_(hbuf[0], non_blocking=True)
                    with torch.cuda.stream(s_dn):
                        hbuf[1].copy_(dbuf[1], non_blocking=True)
                    s_up.synchronize(); s_dn.synchronize()
                    n += 1
                    time.sleep(float(os.environ.get("DMA_GAP_MS", "5")) * 1e-3)
                print(f"   (dma thread: {n} frame copies each way)", flush=True)

            threading.Thread(target=dma, daemon=True).start()
        for K in ks:
            engs = [factory() for _ in range(K)]
            for e in engs:
                if hasattr(e, "lone_pair"):
                    e.lone_pair(K == 1)
            if "--own" in sys.argv:      # the library-made streams the node's lanes run on
                from cfi_amd import _lib
                t_q = time.perf_counter()
                own = _lib.own_streams_apart(torch.device("cuda", 0), K)      # pairwise on different hardware queues (probed)
                print(f"   ({K} streams on different hardware queues found in {(time.perf_counter() - t_q) * 1e3:.1f} ms)", flush=True)
                streams = [o.stream for o in own]
            else:
                streams = [torch.cuda.Stream() for _ in range(K)]
            outs = [torch.empty(H, W, 3, device="cuda") for _ in range(K)]
            res = [None] * K
            for k in range(K):
                with torch.cuda.stream(streams[k]):
                    for _ in range(2):
                        res[k] = step(model, engs[k], x0, x1, outs[k])
            torch.cuda.synchronize()
            n = 12 if model in ("film", "gmfss") else 24
            best = 1e9
            for rep in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(n):
                    k = i % K
                    with torch.cuda.stream(streams[k]):
                        res[k] = step(model, engs[k], x0, x1, outs[k])
                t_issue = time.perf_counter() - t0
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n
                best = min(best, dt)
            if "--stagger" in sys.argv and K > 1:      # lanes started in phase vs a third of a pair apart (host sleeps between the first issues)
                for frac in (0.0, 1.0 / K, 0.0, 1.0 / K, 0.5 / K):
                    nn = 16 * K
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for i in range(nn):
                        k = i % K
                        with torch.cuda.stream(streams[k]):
                            res[k] = step(model, engs[k], x0, x1, outs[k])
                        if i < K - 1 and frac:
                            time.sleep(frac * best * K)      # best = per-pair time of the K-lane loop; one lane's pair takes ~K x that
                    torch.cuda.synchronize()
                    print(f"      start offset {frac:.2f} of a lane's pair: {(time.perf_counter() - t0) / nn * 1e3:.2f} ms per pair", flush=True)
            if "--each" in sys.argv and K > 1:      # every engine of the set alone on its stream, then the pairs of engines together
                import itertools
                for sub in [(k,) for k in range(K)] + list(itertools.combinations(range(K), 2)):
                    nn = 12 * len(sub)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for i in range(nn):
                        k = sub[i % len(sub)]
                        with torch.cuda.stream(streams[k]):
                            res[k] = step(model, engs[k], x0, x1, outs[k])
                    torch.cuda.synchronize()
                    print(f"      engines {sub}: {(time.perf_counter() - t0) / nn * 1e3:.2f} ms per pair", flush=True)
            if "--restream" in sys.argv:      # the same engines on fresh streams, twice
                for trial in range(2):
                    streams2 = [torch.cuda.Stream() for _ in range(K)]
                    for rep in range(2):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        for i in range(n):
                            k = i % K
                            with torch.cuda.stream(streams2[k]):
                                res[k] = step(model, engs[k], x0, x1, outs[k])
                        torch.cuda.synchronize()
                        print(f"      same engines, new streams (set {trial}), run {rep}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per pair", flush=True)
            if "--rewarm" in sys.argv and hasattr(engs[0], "release_workspace"):      # release + re-allocate every workspace, time again
                torch.cuda.synchronize()
                for e in engs:
                    e.release_workspace()
                for rep in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for i in range(n):
                        k = i % K
                        with torch.cuda.stream(streams[k]):
                            res[k] = step(model, engs[k], x0, x1, outs[k])
                    torch.cuda.synchronize()
                    print(f"      after release_workspace, run {rep}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per pair", flush=True)
            same = all(torch.equal(res[0], r) for r in res[1:])
            base = base or best
            print(f"{model}: K = {K}: {best * 1e3:.2f} ms per pair -> {1 / best:.1f} frames/s at 2x ({base / best:.3f}x; host issue {t_issue / n * 1e3:.2f} ms per pair; "
                  f"identical outputs: {same}; device memory {torch.cuda.memory_allocated() / 2**30:.1f} GiB)", flush=True)
            if "--keep" in sys.argv:      # earlier configurations' engines stay alive (their allocations are not recycled)
                kept = globals().setdefault("_kept", [])
                kept.extend(engs)
            else:
                for e in engs:
                    if hasattr(e, "close"):
                        e.close()
            del engs, outs, res
            torch.cuda.empty_cache()
        if dma_stop is not None:
            dma_stop.set()
            time.sleep(0.05)
