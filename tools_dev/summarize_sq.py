"""Aggregate the SQ / GRBM passes of tools_dev/collect_profiles.sh per kernel symbol.

usage: python tools_dev/summarize_sq.py <dir with mfma/ and waves/ rocprofv3 csv output> <out.json>
Per kernel (mean per dispatch): the raw counters, and
  clock_ghz        = GRBM_GUI_ACTIVE / 8 / duration            (rocprofv3 sums the 8 XCDs; MI355X_MICROARCH.md, DVFS)
  mfma_busy_frac   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)     (1024 = 256 CUs x 4 SIMDs, each with one MFMA pipe;
                     the counter counts pipe-busy cycles: 16 per v_mfma_f32_16x16x32_bf16)
  wait_frac, issue_stall_frac, active_frac = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES
The quotient GRBM_GUI_ACTIVE / 8 / duration reads high on short dispatches (the counter window is wider than the kernel: 4.8 GHz
"clocks" for 40-us kernels in round 2), and mfma_busy_frac is understated by the same factor: for dispatches under 100 us both are
reported as null with "short_dispatch": true, and mfma_busy_frac_at_2p1ghz (pipe-busy cycles over duration x 2.1 GHz x 1024, the clock
the long kernels hold) is given instead.
"""
import collections, csv, glob, json, os, sys

root, out = sys.argv[1], sys.argv[2]


def sym(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    cut = name.find("(")
    return name[:cut] if cut > 0 else name


def load(sub, counters):
    tot = collections.defaultdict(lambda: collections.Counter())
    cnt = collections.defaultdict(lambda: collections.Counter())
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = sym(r["Kernel_Name"])
            c = r["Counter_Name"]
            if c in counters:
                tot[k][c] += float(r["Counter_Value"])
                cnt[k][c] += 1
            if "Start_Timestamp" in r and c == counters[0]:
                dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return {k: {c: tot[k][c] / max(cnt[k][c], 1) for c in counters} | {"dispatches": cnt[k][counters[0]],
            "duration_ns_under_pmc": sum(dur[k]) / max(len(dur[k]), 1)} for k in tot}


mf = load("mfma", ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE"])
wv = load("waves", ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"])
res = {"_how": __doc__}
for k in sorted(set(mf) | set(wv), key=lambda k: -(mf.get(k, {}).get("GRBM_GUI_ACTIVE", 0) * mf.get(k, {}).get("dispatches", 0))):
    e = {}
    if k in mf:
        m = mf[k]
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        e.update(m)
        short = m["duration_ns_under_pmc"] < 100e3
        e["short_dispatch"] = short
        e["clock_ghz"] = None if short or not m["duration_ns_under_pmc"] else cyc / m["duration_ns_under_pmc"]
        e["mfma_busy_frac"] = None if short or not cyc else m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)
        e["mfma_busy_frac_at_2p1ghz"] = (m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["duration_ns_under_pmc"] * 2.1 * 1024.0)
                                         if m["duration_ns_under_pmc"] else None)
    if k in wv:
        w = wv[k]
        e.update({c: w[c] for c in w if c.startswith("SQ_")})
        if w["SQ_WAVE_CYCLES"]:
            e["wait_frac"] = w["SQ_WAIT_ANY"] / w["SQ_WAVE_CYCLES"]
            e["issue_stall_frac"] = w["SQ_WAIT_INST_ANY"] / w["SQ_WAVE_CYCLES"]
            e["active_frac"] = w["SQ_ACTIVE_INST_ANY"] / w["SQ_WAVE_CYCLES"]
    res[k] = e
json.dump(res, open(out, "w"), indent=1)
for k, e in list(res.items())[1:16]:
    print(k[:50], {a: (round(b, 3) if isinstance(b, float) else b) for a, b in e.items() if a in ("mfma_busy_frac", "clock_ghz", "wait_frac", "active_frac", "dispatches")})
