"""usage: python3 tests/tools/round_times.py <kernel_trace.csv> -- per-round kernel durations of the LAST frame in a rocprofv3
--kernel-trace of the wavefront pipeline (which rounds carry the time; gaps between launches)."""
import csv, re, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
def short(n):
    m = re.search(r"(k_wf_[a-z]+|k_pathtrace|k_resolve)", n)
    return m.group(1) if m else n[:24]
names = [short(n) for _, _, n in rows]
gens = [i for i, n in enumerate(names) if n == "k_wf_gen"]
i0 = gens[-1]
i1 = next(i for i in range(i0, len(names)) if names[i] == "k_wf_reduce") + 1
seq, nm = rows[i0:i1], names[i0:i1]
t0 = seq[0][0]
print("frame: %d launches, %.2f ms from first start to last end, %.2f ms of gaps" % (len(seq), (seq[-1][1] - t0) / 1e6, sum(seq[i + 1][0] - seq[i][1] for i in range(len(seq) - 1)) / 1e6))
for kind in sorted(set(nm)):
    d = [(e - s) / 1e3 for (s, e, _), n in zip(seq, nm) if n == kind]
    line = "%-12s x%3d total %7.2f ms" % (kind, len(d), sum(d) / 1e3)
    if len(d) > 4:
        line += "   first 5: " + " ".join("%.0f" % x for x in d[:5]) + "   every 5th after: " + " ".join("%.0f" % x for x in d[5::5]) + " us"
    else:
        line += "   " + " ".join("%.0f" % x for x in d) + " us"
    print(line)
