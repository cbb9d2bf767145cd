#!/usr/bin/env python3
"""The timeline of ONE step of a profile leg: every kernel and copy in start order, its duration and the idle
time of the device before it -- what the launches of a per-table job cost beside its kernels.

  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -o p -- python tools/prof_leg.py --leg c3 --cold --steps 3
  python tools/step_timeline.py DIR [anchor-kernel-substring]  > timeline.txt

The step is delimited by the anchor kernel (default: the first kernel of a cold step, `row_classes`; the last
occurrence but one opens the step that is printed, the last closes it)."""
import csv, glob, os, sys

d = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "row_classes"
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:],
                   "%sx%s" % (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r["Workgroup_Size_X"])))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "?"), r.get("Bytes", "")))
ev.sort()
marks = [i for i, e in enumerate(ev) if anchor in e[2]]
if len(marks) < 2:
    sys.exit("anchor %r seen %d times" % (anchor, len(marks)))
a, b = marks[-2], marks[-1]
t0 = ev[a][0]
busy = idle = 0
prev_end = t0
print("%9s %9s %8s  %-60s %s" % ("start_us", "dur_us", "gap_us", "what", "grid / bytes"))
for s, e, name, extra in ev[a:b]:
    gap = max(0, s - prev_end)
    idle += gap
    busy += max(0, e - max(s, prev_end))
    print("%9.1f %9.1f %8.1f  %-60s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, name, extra))
    prev_end = max(prev_end, e)
print("# step %.1f us: device busy %.1f, idle between launches %.1f, %d launches" % ((ev[b][0] - t0) / 1e3, busy / 1e3, idle / 1e3, b - a))
