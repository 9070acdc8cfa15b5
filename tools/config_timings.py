"""Wall-clock timings of the non-count collectors at full file size on an MI355X (developer tool).

One synthetic ca13 file (163 M points) resident in HBM; times the buffer collector (stable emit of
31-byte records) and the grid collector (max-density arg-min) for the named ca13 queries, plus K2 on
a rotating 8-file class working set.  Results feed DESIGN.md; they are not the bench metric.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pkg = importlib.import_module("adhoc-queries-pointclouds_amd")
binding = importlib.import_module("adhoc-queries-pointclouds_amd.binding")
specs = importlib.import_module("adhoc-queries-pointclouds_amd.synth_specs")


def timed(fn, rounds=3):
    ts = []
    out = None
    for _ in range(rounds):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=163_000_000)
    ap.add_argument("--file", type=int, default=5)
    args = ap.parse_args()
    n = args.points
    res = {"points": n}
    with pkg.Context(0) as ctx:
        spec = specs.synth_ca13(points_per_file=n)[args.file]
        xyz, cls = ctx.alloc(12 * n), ctx.alloc(n)
        ctx.synth_fill(spec, 0, n, xyz, cls)
        ctx.synchronize()
        cols = binding.make_columns(xyz=xyz, cls=cls, n=n, scale=list(spec.scale), offset=list(spec.offset))
        for q in ("ca13_S", "ca13_L", "ca13_XL"):
            bmin, bmax = specs.box(q)
            lmin, lmax = pkg.box_to_local(bmin, bmax, list(spec.scale), list(spec.offset))
            pred = pkg.Predicate.bounds(lmin, lmax)

            def count():
                c = ctx.count_collector()
                ctx.scan_dev(cols, pred, c)
                m = c.point_count()
                c.free()
                return m

            def buffer():
                c = ctx.buffer_collector()
                ctx.scan_dev(cols, pred, c)
                m = c.point_count()
                c.free()
                return m

            t_c, m = timed(count)
            t_b, mb = timed(buffer)
            assert m == mb
            entry = {"matches": m, "count_ms": t_c * 1e3, "buffer_ms": t_b * 1e3,
                     "buffer_out_GBps": 31 * m / t_b / 1e9 if m else 0.0, "buffer_Mpts_per_s": n / t_b / 1e6}
            for cell in (100.0, 10.0):
                def grid():
                    g = ctx.grid_collector(bmin, bmax, cell)
                    ctx.scan_dev(cols, pred, g)
                    k = g.point_count()
                    g.free()
                    return k
                t_g, cells = timed(grid, rounds=3)
                entry[f"grid_{int(cell)}_ms"] = t_g * 1e3
                entry[f"grid_{int(cell)}_cells"] = cells
                entry[f"grid_{int(cell)}_Mpts_per_s"] = n / t_g / 1e6
            res[q] = entry
            print(q, json.dumps(entry), flush=True)
        pc = pkg.Predicate.classification(6)

        def ccount():
            c = ctx.count_collector()
            ctx.scan_dev(cols, pc, c)
            m = c.point_count()
            c.free()
            return m
        t, m = timed(ccount, rounds=5)
        res["class6"] = {"matches": m, "count_ms": t * 1e3}
        ctx.free(xyz)
        ctx.free(cls)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
