"""Turn rocprofv3 output under gpurun_out/ into the committed summaries in profiles/.

    python tools/collect_profiles.py --round r01 --tag v2 \
        --stats gpurun_out/prof/*/*_kernel_stats.csv \
        --fetch gpurun_out/pmc_fetch/*/*_counter_collection.csv \
        --write gpurun_out/pmc_write/*/*_counter_collection.csv --workload cfg2

HBM traffic per scan launch follows /opt/skills/guides/MI355X_MICROARCH.md "HBM":
FETCH_SIZE and WRITE_SIZE come from SEPARATE --pmc passes, are in KiB, and on
gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane)
coalesced streaming read, so   hbm_bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024.
"""
from __future__ import annotations

import argparse
import csv
import glob
import json
import os
import shutil
import statistics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(pattern):
    hits = sorted(glob.glob(pattern))
    if not hits:
        raise SystemExit(f"no file matches {pattern}")
    return hits[-1]


def counter_mean(path, counter):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if "scan_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    if not vals:
        raise SystemExit(f"{path}: no {counter} rows for scan_kernel")
    return statistics.mean(vals), len(vals)


def sq_summary(path, dst):
    """Mean per launch of every SQ counter of the scan kernel (one --pmc pass), as a small CSV."""
    agg, meta = {}, None
    for r in csv.DictReader(open(path)):
        if "scan_kernel" not in r["Kernel_Name"]:
            continue
        agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        meta = meta or (r["Kernel_Name"], r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"], r["SGPR_Count"])
    if not agg:
        return
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Grid_Size", "Workgroup_Size", "VGPR_Count", "SGPR_Count", "Counter_Name",
                    "Mean_Counter_Value_Per_Launch", "Launches", "Share_Of_SQ_WAVE_CYCLES"])
        wc = statistics.mean(agg.get("SQ_WAVE_CYCLES", [0])) or 1.0
        for name in sorted(agg):
            v = statistics.mean(agg[name])
            w.writerow(list(meta) + [name, round(v), len(agg[name]), round(v / wc, 4)])
    print("wrote", dst)


def collect_dir(args, prof):
    d = args.dir
    for w in ("cfg2", "cfg3"):
        st = glob.glob(os.path.join(d, f"stats_{w}", "*", "*_kernel_stats.csv"))
        fe = glob.glob(os.path.join(d, f"fetch_{w}", "*", "*_counter_collection.csv"))
        wr = glob.glob(os.path.join(d, f"write_{w}", "*", "*_counter_collection.csv"))
        sq = glob.glob(os.path.join(d, f"sq_{w}", "*", "*_counter_collection.csv"))
        sub = argparse.Namespace(**vars(args))
        sub.workload, sub.dir = w, None
        sub.stats = st[-1] if st else None
        sub.fetch = fe[-1] if fe and wr else None
        sub.write = wr[-1] if fe and wr else None
        collect_one(sub, prof)
        if sq:
            tag = f"_{args.tag}" if args.tag else ""
            sq_summary(sq[-1], os.path.join(prof, f"{args.round}_sq_scan_kernel{tag}_{w}.csv"))
    sq = glob.glob(os.path.join(d, "sq_short_m2", "*", "*_counter_collection.csv"))
    if sq:
        tag = f"_{args.tag}" if args.tag else ""
        sq_summary(sq[-1], os.path.join(prof, f"{args.round}_sq_scan_kernel{tag}_short_m2.csv"))
    for extra in ("dma_only", "ed64k", "sa2m", "short"):
        st = glob.glob(os.path.join(d, f"stats_{extra}", "*", "*_kernel_stats.csv"))
        if st:
            tag = f"_{args.tag}" if args.tag else ""
            dst = os.path.join(prof, f"{args.round}_kernel_stats{tag}_{extra}.csv")
            shutil.copyfile(st[-1], dst)
            print("wrote", dst)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r01")
    ap.add_argument("--tag", default="")
    ap.add_argument("--stats")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--bytes", type=int, default=4 << 30, help="algorithmic bytes per launch")
    ap.add_argument("--dir", help="a directory written by tools/profile_round.sh (gpurun_out/prof_<tag>): take everything in it")
    args = ap.parse_args()
    prof = os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)
    if args.dir:
        return collect_dir(args, prof)
    collect_one(args, prof)


def collect_one(args, prof):
    tag = f"_{args.tag}" if args.tag else ""
    if args.stats:
        dst = os.path.join(prof, f"{args.round}_kernel_stats{tag}_{args.workload}.csv")
        shutil.copyfile(one(args.stats), dst)
        print("wrote", dst)
    if args.fetch and args.write:
        fpath, wpath = one(args.fetch), one(args.write)
        fetch_kib, nf = counter_mean(fpath, "FETCH_SIZE")
        write_kib, nw = counter_mean(wpath, "WRITE_SIZE")
        hbm = 2.0 * fetch_kib * 1024.0 + write_kib * 1024.0
        rows = [r for p in (fpath, wpath) for r in csv.DictReader(open(p)) if "scan_kernel" in r["Kernel_Name"]]
        dst = os.path.join(prof, f"{args.round}_pmc_scan_kernel{tag}_{args.workload}.csv")
        with open(dst, "w", newline="") as f:
            w = csv.writer(f)
            cols = ["Kernel_Name", "Counter_Name", "Counter_Value", "Grid_Size", "Workgroup_Size", "VGPR_Count",
                    "SGPR_Count"]
            w.writerow(cols)
            for r in rows:
                w.writerow([r[c] for c in cols])
        print("wrote", dst)
        tpath = os.path.join(prof, "traffic.json")
        try:
            traffic = json.load(open(tpath))
        except (OSError, ValueError):
            traffic = {}
        traffic[args.workload] = {
            "hbm_bytes_per_launch": int(round(hbm)),
            "fetch_size_kib_mean": fetch_kib, "write_size_kib_mean": write_kib,
            "launches_averaged": [nf, nw],
            "formula": "2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE halves wide coalesced reads)",
            "algorithmic_bytes_per_launch": args.bytes,
            "ratio_to_algorithmic": hbm / args.bytes,
            "source": os.path.basename(dst), "round": args.round, "tag": args.tag,
        }
        json.dump(traffic, open(tpath, "w"), indent=1, sort_keys=True)
        print("wrote", tpath, traffic[args.workload])


if __name__ == "__main__":
    main()
