"""Per-PHASE GPU time of the training step from one rocprofv3 run with roctx ranges:
    SLAM_ROCTX=1 rocprofv3 --marker-trace --hip-runtime-trace --kernel-trace --output-format csv -d <dir> -- python bench.py ...
    python tools/phase_table.py <dir> <out.md> [title]
slam_llm_amd/trace.py pushes a range around each phase on the HOST (where the launches are issued); a kernel belongs to the innermost
range that contains the host time of ITS launch call (joined through the correlation id of the HIP runtime trace; kernels whose launch
call is not in the trace fall back to their own start time).  GPU time = sum of kernel durations; steps = number of llm_fwd ranges."""
import collections
import csv
import glob
import sys


def rows(d, pat):
    out = []
    for f in glob.glob(d + "/**/*" + pat, recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


def main():
    d, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else "C3 step by phase"
    marks = [(r["Function"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows(d, "marker_api_trace.csv") if r["Function"].startswith("slam/")]
    launches = {}
    for r in rows(d, "hip_api_trace.csv"):
        if "Launch" in r["Function"]:
            launches[r["Correlation_Id"]] = int(r["Start_Timestamp"])
    kern = rows(d, "kernel_trace.csv")
    if not marks or not kern:
        raise SystemExit(f"no marker ranges ({len(marks)}) or kernels ({len(kern)}) under {d}")
    marks.sort(key=lambda m: m[1])
    starts = [m[1] for m in marks]
    import bisect

    def phase_of(t):
        best = None
        i = bisect.bisect_right(starts, t)
        for name, s, e in marks[max(0, i - 8): i]:      # ranges are shallow: the innermost of the last few that started before t
            if s <= t <= e and (best is None or (e - s) < best[1]):
                best = (name, e - s)
        return best[0][5:] if best else "(outside any range)"

    n_steps = max(1, sum(1 for m in marks if m[0] == "slam/llm_fwd"))
    gpu = collections.defaultdict(float)
    cnt = collections.Counter()
    joined = 0
    for r in kern:
        t = launches.get(r.get("Correlation_Id"))
        if t is not None:
            joined += 1
        else:
            t = int(r["Start_Timestamp"])
        ph = phase_of(t)
        gpu[ph] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        cnt[ph] += 1
    host = collections.defaultdict(float)
    for name, s, e in marks:
        host[name[5:]] += (e - s) / 1e6
    total = sum(gpu.values())
    with open(out, "w") as f:
        f.write(f"# {title}\n\nrocprofv3 --marker-trace --hip-runtime-trace --kernel-trace over the bench command with SLAM_ROCTX=1; {n_steps} steps (warm-up included), "
                f"{len(kern)} kernel dispatches, {joined} of them joined to their launch call by correlation id.  GPU ms = sum of kernel durations; "
                "host ms = time the host spent inside the range issuing launches (the device runs behind).\n\n")
        f.write("| phase | launches / step | GPU ms / step | share of kernel time | host ms / step |\n|---|---|---|---|---|\n")
        for ph, ms in sorted(gpu.items(), key=lambda kv: -kv[1]):
            f.write(f"| {ph} | {cnt[ph] / n_steps:.0f} | {ms / n_steps:.2f} | {ms / total:.3f} | {host.get(ph, 0.0) / n_steps:.2f} |\n")
        f.write(f"| **sum** | {len(kern) / n_steps:.0f} | {total / n_steps:.2f} | 1.000 | |\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
