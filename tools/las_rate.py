"""Developer tool: throughput of the AoS (LAS record) count path on device-resident records."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("adhoc-queries-pointclouds_amd")
binding = importlib.import_module("adhoc-queries-pointclouds_amd.binding")
specs = importlib.import_module("adhoc-queries-pointclouds_amd.synth_specs")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60_000_000
dev = torch.device("cuda:0")
ts = torch.cuda.Stream()
torch.cuda.set_stream(ts)
stream = ts.cuda_stream
with pkg.Context(0) as ctx:
    spec = specs.synth_ca13(points_per_file=n)[5]
    xyz = torch.empty(n * 12, dtype=torch.uint8, device=dev)
    cls = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx.synth_fill(spec, 0, n, xyz.data_ptr(), cls.data_ptr(), stream)
    torch.cuda.synchronize()
    bmin, bmax = specs.box("ca13_L")
    lmin, lmax = pkg.box_to_local(bmin, bmax, list(spec.scale), list(spec.offset))
    counter = torch.zeros(2, dtype=torch.int64, device=dev)
    cc = ctx.count_collector(device_counter=counter.data_ptr())
    out = {}
    bpcs = [int(b) for b in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["2"])]
    for rl, bpc in [(rl, b) for rl in (20, 26, 28, 34) for b in bpcs]:
        ctx.set_option("blocks_per_cu", bpc)
        nfiles = 6  # rotate buffers: 6 x n x rl bytes > Infinity Cache
        recs = []
        for f in range(nfiles):
            r = torch.zeros(n, rl, dtype=torch.uint8, device=dev)
            r[:, :12] = xyz.view(n, 12)
            r[:, 15] = cls
            recs.append(r)
        torch.cuda.synchronize()
        res = {}
        for kind in ("bounds", "class"):
            pred = pkg.Predicate.bounds(lmin, lmax) if kind == "bounds" else pkg.Predicate.classification(6)
            times = []
            for it in range(14):
                r = recs[it % nfiles]
                cols = binding.make_columns(xyz=r.data_ptr(), cls=r.data_ptr() + 15, n=n, xyz_stride=rl, cls_stride=rl,
                                            rgb_stride=rl, scale=list(spec.scale), offset=list(spec.offset))
                counter.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ctx.scan_dev(cols, pred, cc, stream)
                e1.record()
                e1.synchronize()
                if it >= 2:
                    times.append(e0.elapsed_time(e1))
            times.sort()
            med = times[len(times) // 2]
            res[kind] = {"ms": med, "GBps": n * rl / med / 1e6, "Mpts_per_s": n / med / 1e3, "count": int(counter[0].item())}
        out[(rl, bpc)] = res
        print(rl, f"blocks/CU={bpc}", json.dumps(res), flush=True)
        del recs
    cc.free()
