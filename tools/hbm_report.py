#!/usr/bin/env python3
"""Per-kernel HBM report: joins the FETCH_SIZE / WRITE_SIZE counter passes with the kernel
durations of a plain --kernel-trace --stats run of the SAME command.

usage: hbm_report.py fetch_counter_collection.csv write_counter_collection.csv kernel_stats.csv STEPS out.md [out.json]

Counters (MI355X_MICROARCH.md, HBM section): both are in KiB; on gfx950 FETCH_SIZE counts 64 B
per 128-B request of wide coalesced reads -> x2.  Each pass is its own `rocprofv3 --pmc` run
(never combined with other trace domains).  GB/s = (2*FETCH + WRITE) bytes / average duration of
that kernel in the un-instrumented stats run."""
import collections, csv, json, sys

PEAK = 8000.0   # GB/s, HBM3E nominal


def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    return k.split("(")[0]


def load(path, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        agg[k][0] += float(r["Counter_Value"])
        agg[k][1] += 1
    return agg


def main():
    fetch, write, stats, steps, out_md = sys.argv[1:6]
    steps = float(steps)
    f, w = load(fetch, "FETCH_SIZE"), load(write, "WRITE_SIZE")
    dur = {}
    for r in csv.DictReader(open(stats)):
        dur[short(r["Name"])] = (float(r["TotalDurationNs"]), int(r["Calls"]))
    rows = []
    for k in f:
        if k not in w or k not in dur:
            continue
        fb = 2.0 * f[k][0] / f[k][1] * 1024.0
        wb = w[k][0] / w[k][1] * 1024.0
        tot_ns, calls = dur[k]
        avg_us = tot_ns / calls / 1e3
        rows.append(dict(kernel=k, launches_per_step=calls / steps, avg_us=avg_us, ms_per_step=tot_ns / steps / 1e6,
                         fetch_bytes_per_launch=fb, write_bytes_per_launch=wb, hbm_bytes_per_launch=fb + wb,
                         gbps=(fb + wb) / (avg_us * 1e-6) / 1e9))
    rows.sort(key=lambda r: -r["ms_per_step"])
    with open(out_md, "w") as o:
        o.write("| kernel | launches/step | avg us | ms/step | HBM MB/launch (2*FETCH+WRITE) | achieved GB/s | of 8 TB/s |\n")
        o.write("|---|---|---|---|---|---|---|\n")
        for r in rows:
            if r["ms_per_step"] < 0.02:
                continue
            o.write(f"| `{r['kernel']}` | {r['launches_per_step']:.0f} | {r['avg_us']:.1f} | {r['ms_per_step']:.3f} | "
                    f"{r['hbm_bytes_per_launch'] / 1e6:.1f} | {r['gbps']:.0f} | {r['gbps'] / PEAK:.2f} |\n")
    if len(sys.argv) > 6:
        json.dump({r["kernel"]: {k: v for k, v in r.items() if k != "kernel"} for r in rows}, open(sys.argv[6], "w"),
                  indent=1, sort_keys=True)
    tot = sum(r["ms_per_step"] for r in rows)
    byt = sum(r["hbm_bytes_per_launch"] * r["launches_per_step"] for r in rows)
    print(f"{len(rows)} kernels, {tot:.2f} ms/step, {byt / 1e9:.2f} GB HBM traffic per step")


if __name__ == "__main__":
    main()
