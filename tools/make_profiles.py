"""Turn the rocprofv3 CSV outputs of one `bench.py` profile session into the files committed under profiles/.

usage: python tools/make_profiles.py <stats_dir> <fetch_dir> <write_dir> <workload> <round-tag>

  stats_dir : output of  rocprofv3 --kernel-trace --stats --output-format csv -d <stats_dir> -- python bench.py ...
  fetch_dir : output of  rocprofv3 --pmc FETCH_SIZE --output-format csv -d <fetch_dir> -- (same command)
  write_dir : output of  rocprofv3 --pmc WRITE_SIZE --output-format csv -d <write_dir> -- (same command)

Writes profiles/<tag>_kernel_stats.csv (verbatim kernel_stats), profiles/pmc_<workload>.json + pmc_latest.json (per-kernel
FETCH_SIZE / WRITE_SIZE KiB per launch, read by bench.py for roofline.traffic) and
profiles/<tag>_pmc_summary.md.  FETCH_SIZE / WRITE_SIZE are in KiB (MI355X_MICROARCH.md, HBM section); on
gfx950 FETCH_SIZE counts wide streaming reads at half their size -- bench.py applies the x2 to the sweep kernel
only (its reads are wide and streaming); the raw figure is stored here.
"""
import csv, glob, json, os, re, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FETCH_X2 = ("k_sweep_lds", "k_em_persist", "k_part_route", "k_part_insert", "k_filter", "k_export")


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True), key=os.path.getmtime)
    if not hits:
        raise SystemExit(f"no *{suffix} under {d}")
    return hits[-1]                        # the newest: gpurun merges a session's files into what earlier sessions left


def short(name):
    """kernel name without namespaces / argument list, template arguments kept"""
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"\s*\[clone.*$", "", name)
    depth, out = 0, []
    for ch in name:                       # strip the (args) part at template depth 0
        if ch == "<": depth += 1
        if ch == ">": depth -= 1
        if ch == "(" and depth == 0: break
        out.append(ch)
    name = "".join(out)
    head, lt, tail = name.partition("<")
    return head.split("::")[-1] + lt + tail


LOOP_KERNELS = ("k_sweep_lds", "k_update")       # enqueued in chunks: the launches past the stop iteration return at once


def counters(d, counter):
    """-> {kernel: [launches, sum, no-op launches]}.  The EM loop's kernels are enqueued in chunks and turn into no-ops once the
    loop has ended (em.hip): a launch whose counter is below 5 % of that kernel's largest is such a no-op and is left out of the
    per-launch average (it would dilute it by the ~20 % of launches that do nothing)."""
    vals = defaultdict(list)
    with open(find(d, "counter_collection.csv")) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            vals[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    per = {}
    for k, v in vals.items():
        noop = 0
        if k.startswith(LOOP_KERNELS) and v:
            cut = 0.05 * max(v)
            noop = sum(1 for x in v if x < cut)
            v = [x for x in v if x >= cut]
        per[k] = [len(v), sum(v), noop]
    return per


def running_launch_us(stats_dir):
    """average duration of the launches of the EM loop's kernels that did run (kernel trace; no-ops are < 30 % of the longest)"""
    out = {}
    try:
        path = find(stats_dir, "kernel_trace.csv")
    except SystemExit:
        return out
    durs = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            if k.startswith(LOOP_KERNELS):
                durs[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    for k, v in durs.items():
        cut = 0.3 * max(v)
        run = [x for x in v if x >= cut]
        out[k] = (len(run), sum(run) / len(run), len(v) - len(run))
    return out


def main():
    stats_dir, fetch_dir, write_dir, workload, tag = sys.argv[1:6]
    prof = os.path.join(ROOT, "profiles")
    ks = find(stats_dir, "kernel_stats.csv")
    shutil.copy(ks, os.path.join(prof, f"{tag}_kernel_stats.csv"))
    stats = {}
    with open(ks) as f:
        for row in csv.DictReader(f):
            stats[short(row["Name"])] = (int(row["Calls"]), float(row["AverageNs"]), float(row["Percentage"]))
    fe, wr = counters(fetch_dir, "FETCH_SIZE"), counters(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, [0, 0])[1] + wr.get(k, [0, 0])[1])):
        if not k.startswith("k_"):
            continue                        # torch ops of the input generator: not part of the path
        n = max(fe.get(k, [0, 0])[0], wr.get(k, [0, 0])[0], 1)
        kernels[k] = {"launches": n, "noop_launches_left_out": max(fe.get(k, [0, 0, 0])[2], wr.get(k, [0, 0, 0])[2]),
                      "fetch_kib_per_launch": fe.get(k, [0, 0.0])[1] / max(fe.get(k, [1, 0])[0], 1),
                      "write_kib_per_launch": wr.get(k, [0, 0.0])[1] / max(wr.get(k, [1, 0])[0], 1),
                      # MI355X_MICROARCH.md, HBM: gfx950 reports a wide (16 B per lane) coalesced streaming read at half its
                      # size.  Which kernels read that way: the sweep (label stream), the route pass (ids staged with
                      # 16-byte loads), the insert pass (one 16-byte granule per lane); the EM update reads 8 B per lane.
                      "fetch_x2": any(k.startswith(p) for p in FETCH_X2)}
    for name in (f"pmc_{workload}.json", "pmc_latest.json"):     # bench.py reads the file of its workload
        json.dump({"workload": workload, "source": f"profiles/{tag}_pmc_summary.md", "kernels": kernels},
                  open(os.path.join(prof, name), "w"), indent=1)
    with open(os.path.join(prof, f"{tag}_pmc_summary.md"), "w") as f:
        f.write(f"# {tag}: per-kernel time and HBM traffic, `bench.py --workload {workload}` on one MI355X\n\n"
                "Three separate rocprofv3 runs of the same command (kernel trace + stats; `--pmc FETCH_SIZE`; "
                "`--pmc WRITE_SIZE`).\nFETCH/WRITE are KiB per launch, raw counter values.  gfx950 reports wide (16 B per lane) "
                "coalesced streaming reads at half their size (MI355X_MICROARCH.md, HBM): kernels marked x2 read that way and "
                "bench.py doubles their FETCH before comparing with byte counts; WRITE_SIZE is taken as reported.\n"
                "`at::native::*` / `compute_cuda_kernel` rows are torch ops of the synthetic input generator "
                "(bench set-up, outside the timed region).\n\n"
                "| kernel | calls | avg us | % time | FETCH KiB/launch | WRITE KiB/launch | FETCH correction |\n|---|---:|---:|---:|---:|---:|---|\n")
        for k, (calls, avg, pct) in sorted(stats.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
            e = kernels.get(k)
            k = k if len(k) <= 72 else k[:69] + "..."
            f.write(f"| `{k}` | {calls} | {avg / 1e3:.2f} | {pct:.2f} | "
                    + (f"{e['fetch_kib_per_launch']:.1f} | {e['write_kib_per_launch']:.1f} | {'x2' if e['fetch_x2'] else 'x1'} |\n" if e else "- | - | |\n"))
        run = running_launch_us(stats_dir)
        if run:
            f.write("\nThe EM loop's kernels are enqueued in chunks; launches past the stop iteration return at once.  The rows above average "
                    "over ALL launches (rocprofv3's own statistics); the launches that did run, from the kernel trace (FETCH / WRITE per launch "
                    "above already leave the no-ops out):\n\n| kernel | launches that ran | avg us | no-op launches |\n|---|---:|---:|---:|\n")
            for k, (n, avg, noop) in sorted(run.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
                f.write(f"| `{k}` | {n} | {avg:.2f} | {noop} |\n")
    print("wrote", tag, "with", len(stats), "kernels;", len(kernels), "with counters")


if __name__ == "__main__":
    main()
