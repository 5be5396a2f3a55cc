"""Fixed vs per-image part of the headline phase-2 step: the captured step at several batch sizes, a least-squares line
t(N) = fixed + per_image * N over them, and (with --prof DIR...) the same decomposition per kernel from rocprofv3 --stats
summaries collected at those batch sizes (tools/prof.sh NAME --batch N).

    python tools/scaling.py [--out FILE] [N ...]                       (GPU box) step times
    python tools/scaling.py --prof N=DIR [N=DIR ...] [--out FILE]      (anywhere) per-kernel lines from kernel-stats CSVs
"""
import csv
import glob
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def fit(ns, ts):
    """least squares t = a + b n -> (a, b)"""
    m = len(ns)
    sx, sy = sum(ns), sum(ts)
    sxx, sxy = sum(x * x for x in ns), sum(x * y for x, y in zip(ns, ts))
    b = (m * sxy - sx * sy) / (m * sxx - sx * sx)
    return (sy - b * sx) / m, b


def step_times(ns, reps=400):
    import torch
    import bench
    from cgs_amd import engine
    dev = torch.device("cuda:0")
    out = {}
    for n in ns:
        eng = engine.HourglassEngine(n, device=dev, dropout=0.3, use_graph=True)
        eng.load_state(*bench.g1_weights())
        A, B, Y = bench.synthetic(n, 0, dev)
        eng.phase2_step(A, B, Y)
        snap = eng.snapshot_state()
        best = None
        for _ in range(4):
            for _ in range(reps // 2):
                eng.phase2_step()
            eng.restore_state(snap)
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st = torch.cuda.current_stream()
            ev0.record(st)
            for _ in range(reps):
                eng.phase2_step()
            ev1.record(st)
            torch.cuda.synchronize()
            ms = ev0.elapsed_time(ev1) / reps
            best = ms if best is None else min(best, ms)
            eng.restore_state(snap)
        out[n] = best
        print(f"N = {n:5d}: {best:.4f} ms/step  ({best * 1e3 / n:.4f} us/image)", flush=True)
        del eng
        time.sleep(0.2)
    return out


def kernel_lines(dirs):
    """{kernel: {n: us per step}} from rocprofv3 kernel-stats CSVs, one directory per batch size"""
    per = {}
    for n, d in dirs.items():
        f = sorted(glob.glob(d + "/*/*_kernel_stats.csv") + glob.glob(d + "/*_kernel_stats.csv"))[-1]
        rows = list(csv.DictReader(open(f)))
        once = [int(r["Calls"]) for r in rows if any(k in r["Name"] for k in ("reduce_adam_kernel", "mask_head_kernel"))]
        steps = min(once)
        for r in rows:
            if any(k in r["Name"] for k in ("at::native", "__amd_rocclr", "ncclDevKernel")):
                continue
            name = r["Name"].replace("void ", "").split("(")[0]
            per.setdefault(name, {})[n] = float(r["TotalDurationNs"]) / steps / 1e3
    return per


def main():
    args = sys.argv[1:]
    out_file = None
    if "--out" in args:
        i = args.index("--out")
        out_file = args[i + 1]
        del args[i:i + 2]
    res = {}
    if args and args[0] == "--prof":
        dirs = {int(a.split("=")[0]): a.split("=")[1] for a in args[1:]}
        per = kernel_lines(dirs)
        ns = sorted(dirs)
        tot = {n: 0.0 for n in ns}
        rows = []
        for name, d in sorted(per.items(), key=lambda kv: -kv[1].get(512, 0)):
            if len(d) < len(ns):
                continue
            a, b = fit(ns, [d[n] for n in ns])
            for n in ns:
                tot[n] += d[n]
            rows.append({"kernel": name, "us_per_step": {str(n): round(d[n], 2) for n in ns}, "fixed_us": round(a, 2), "us_per_image": round(b, 5)})
            print(f"{name[:58]:58s} " + " ".join(f"{d[n]:7.1f}" for n in ns) + f"   fixed {a:6.1f} us  + {b * 1e3:7.2f} ns/img")
        a, b = fit(ns, [tot[n] for n in ns])
        print(f"{'sum of the step kernels':58s} " + " ".join(f"{tot[n]:7.1f}" for n in ns) + f"   fixed {a:6.1f} us  + {b * 1e3:7.2f} ns/img")
        res = {"kernels": rows, "sum": {"us_per_step": {str(n): round(tot[n], 2) for n in ns}, "fixed_us": round(a, 2), "us_per_image": round(b, 5)}}
    else:
        ns = [int(a) for a in args] or [128, 256, 384, 512, 768, 1024]
        t = step_times(ns)
        a, b = fit(list(t), list(t.values()))
        print(f"t(N) = {a * 1e3:.1f} us + {b * 1e3:.4f} us/image * N   (least squares over N = {ns})")
        # the part that scales, priced at N = 512
        res = {"ms_per_step": {str(n): round(v, 5) for n, v in t.items()}, "fixed_ms": round(a, 5), "ms_per_image": round(b, 7),
               "fixed_share_at_512": round(a / t[512], 4) if 512 in t else None}
    if out_file:
        with open(out_file, "w") as fp:
            json.dump(res, fp, indent=1)


if __name__ == "__main__":
    main()
