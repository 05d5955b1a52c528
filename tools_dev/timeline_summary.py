"""From a rocprofv3 --kernel-trace csv of the two-stream bench: per steady-state step, the time some kernel is running (union), the time
kernels of BOTH queues are running at once, the idle time, and the summed kernel time per queue.
usage: python tools_dev/timeline_summary.py <r_kernel_trace.csv> <out.json>"""
import csv, json, sys, collections

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "mixup_kernel" in r["Kernel_Name"]]      # one per step (the batch's mixup)
lo, hi = marks[len(marks) // 2], marks[-1]                                           # the second half of the run: steady state
steps = len(marks) - 1 - len(marks) // 2
seg = rows[lo:hi]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(rows[hi]["Start_Timestamp"])
ev = []
perq = collections.Counter()
for r in seg:
    s, e = int(r["Start_Timestamp"]), min(int(r["End_Timestamp"]), t1)
    if e <= s:
        continue
    ev.append((s, 1))
    ev.append((e, -1))
    perq[r.get("Queue_Id", "?")] += e - s
ev.sort()
busy = both = 0
depth, prev = 0, t0
for t, d in ev:
    if depth >= 1:
        busy += t - prev
    if depth >= 2:
        both += t - prev
    depth += d
    prev = t
span = t1 - t0
out = {"_how": __doc__, "steps": steps, "ms_per_step_under_rocprof": span / steps / 1e6, "busy_ms_per_step": busy / steps / 1e6,
       "two_or_more_kernels_ms_per_step": both / steps / 1e6, "idle_ms_per_step": (span - busy) / steps / 1e6,
       "kernel_ms_per_step_by_queue": {q: v / steps / 1e6 for q, v in perq.items()}}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in out.items() if k != "_how"}))
