"""where is the GPU idle inside a step?  usage: python3 tools/step_timeline.py <kernel_trace.csv> [min_gap_us] [launches.txt: the step's launches, one per line]
Takes rocprofv3's kernel trace of a bench run, finds the steps (from one k_index_stage* launch to the next), and for the LAST complete step lists every interval in which no kernel
of the process was running, with the kernel before and after it; then the sums."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))) for r in rows), key=lambda e: e[0])
starts = [i for i, e in enumerate(ev) if e[2].startswith("k_index_stage")]
# the first index-stage launch of each step: launches closer than 1 ms belong together
firsts = [s for k, s in enumerate(starts) if k == 0 or ev[s][0] - ev[starts[k - 1]][0] > 2_000_000]
if len(firsts) < 2:
    sys.exit("fewer than two steps in the trace")
a, b = firsts[-2], firsts[-1]
step = ev[a:b]
t_busy, gaps, cur_end = 0, [], step[0][0]
prev = step[0][2]
for s, e, n in step:
    if s > cur_end:
        gaps.append((s - cur_end, prev, n))
    if e > cur_end:
        t_busy += e - max(s, cur_end)
        cur_end = e; prev = n
span = ev[b][0] - step[0][0]
print("step: %d launches, %.3f ms from its first kernel to the next step's first; kernels running %.3f ms, nothing running %.3f ms" % (len(step), span / 1e6, t_busy / 1e6, (span - t_busy) / 1e6))
print("   of that, behind the step's last kernel (host tail + next step's start): %.3f ms" % ((ev[b][0] - cur_end) / 1e6))
tot = {}
for g, p, n in gaps:
    tot[p] = tot.get(p, 0) + g
print("gaps >= %.0f us (us, after -> before):" % min_gap)
for g, p, n in gaps:
    if g >= min_gap * 1e3:
        print("  %7.1f  %s -> %s" % (g / 1e3, p[:60], n[:60]))
small = sum(g for g, p, n in gaps if g < min_gap * 1e3)
print("gaps below that: %d, %.3f ms together" % (sum(1 for g, _, _ in gaps if g < min_gap * 1e3), small / 1e6))
print("by the kernel in front of the gap (ms):")
for p, g in sorted(tot.items(), key=lambda kv: -kv[1])[:14]:
    print("  %6.3f  %s" % (g / 1e6, p[:80]))
dur = {}
for s, e, n in step:
    d = dur.setdefault(n, [0, 0]); d[0] += e - s; d[1] += 1
print("launches of the step by kernel (ms, count):")
for n, (d, k) in sorted(dur.items(), key=lambda kv: -kv[1][0])[:40]:
    print("  %7.3f %4d  %s" % (d / 1e6, k, n[:80]))
if len(sys.argv) > 3:
    with open(sys.argv[3], "w") as f:
        f.write("# start us (from the step's first kernel), duration us, kernel\n")
        for st_, en_, n in step:
            f.write("%9.1f %8.1f  %s\n" % ((st_ - step[0][0]) / 1e3, (en_ - st_) / 1e3, n[:70]))
