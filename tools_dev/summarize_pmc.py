"""Aggregate the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py` into profiles/pmc_traffic.json.

usage: python tools_dev/summarize_pmc.py <dir with fetch/ and write/ rocprofv3 csv output> <config> <out.json>
Per kernel symbol (template arguments of the ring kernels dropped): mean KiB per dispatch and
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- on gfx950 FETCH_SIZE reports half of a 16-B/lane coalesced stream,
WRITE_SIZE is exact (MI355X_MICROARCH.md, HBM / rocprofv3 section).
"""
import collections, csv, glob, json, os, re, subprocess, sys

root, config, out = sys.argv[1], sys.argv[2], sys.argv[3]
KEYS = ("gemm_nt_kernel<64", "gemm_nt_kernel<128", "gemm_nt256_kernel", "gemm_tn_kernel", "gemm_tn192g_kernel", "gemm_tn192d_kernel", "gemm_tn192_kernel", "attn_fwd_ring_kernel",
        "attn_fwd_kernel", "attn_bwd_dq_kernel", "attn_bwd_dkv_kernel", "ln_fwd_kernel", "ln_bwd_kernel",
        # the student's fused kernels (round 5: VERDICT round 4, item 4a)
        "mlp192_kernel<0", "mlp192_kernel<1", "attn192_fwd_kernel", "attn192_bwd_kernel", "gemm_nt_lnbwd_kernel", "gemm_tn_kernel")


def sym(name):
    for k in KEYS:
        if k in name:
            return k + ">" if k.endswith(("<64", "<128", "<0", "<1")) else k        # any epilogue instantiation of the tile kernel
    return None


raw = collections.defaultdict(dict)
for ctr, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    files = glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True)
    tot, cnt = collections.Counter(), collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != ctr:
                continue
            k = sym(r["Kernel_Name"])
            if k:
                tot[k] += float(r["Counter_Value"])
                cnt[k] += 1
    for k in tot:
        raw[k][ctr] = tot[k] / cnt[k]
        raw[k]["dispatches"] = cnt[k]
res = {"_how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 3 --warmup 2; per-launch mean "
               "over all dispatches of the kernel; bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE reports 1/2 of a "
               "16-B/lane coalesced stream, WRITE_SIZE exact; MI355X_MICROARCH.md section HBM). Memory-side requests include "
               "Infinity-Cache hits.",
       "_source": "tools_dev/collect_profiles.sh -> tools_dev/summarize_pmc.py " + " ".join(sys.argv[1:]),
       "_commit": (subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or
                   os.environ.get("DKD_COMMIT") or None),
       config: {k: (2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024 for k, v in raw.items()},
       "_raw_KiB": raw}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res[config], indent=1))
