"""rocprofv3 --pmc passes + the bench line of the same run -> profiles/rNN/pmc_hbm.json (what bench.py scales into roofline.traffic).

usage: python tools/pmc_to_json.py <bench line .json of a --pmc run> <FETCH_SIZE dir> <WRITE_SIZE dir> <out.json>

Counters (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KB and derive from the L2's memory-side request
counters (TCC_EA0_RDREQ / WRREQ; Infinity-Cache hits included).  The guide's calibration -- FETCH_SIZE shows half the bytes of
a wide (16 B/lane) coalesced streaming read -- does not apply to these kernels' 4- and 8-byte per-lane accesses; the values
are stored as reported and marked uncalibrated.  Two separate passes, as the guide prescribes (the two counters do not fit
one)."""
import csv, glob, json, os, re, sys
from collections import defaultdict

bench_json, fetch_dir, write_dir, out = sys.argv[1:5]
line = json.loads([l for l in open(bench_json) if l.startswith("{")][-1])


def totals(d):
    tot, disp = defaultdict(float), defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace("void ", "").replace("sacamd::", "").strip()
            tot[name] += float(row["Counter_Value"])
            disp[name].add(row["Dispatch_Id"])
    return tot, {k: len(v) for k, v in disp.items()}


fetch, nd = totals(fetch_dir)
write, _ = totals(write_dir)
stage_bytes = {"k_ols": 16, "k_ols_grid": 16, "k_ols_pack": 16, "k_lms": 20}
res = {"source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes) -- bench.py {line['config']['workload']}; "
                 "counters in KB (TCC_EA0 based, Infinity-Cache hits included; 4- and 8-byte per-lane accesses: uncalibrated, see MI355X_MICROARCH.md HBM section)",
       "kernels": {}}
for kname, mb in line["kernel_instances_algorithmic_MB"].items():
    key = kname.split(" (")[0]
    fam = key.split("<")[0]
    if fam not in stage_bytes or key not in fetch:
        continue
    isteps = mb * 1e6 / stage_bytes[fam]
    res["kernels"][key] = {"item_steps": isteps, "fetch_bytes_per_item_step": fetch[key] * 1024 / isteps,
                           "write_bytes_per_item_step": write.get(key, 0.0) * 1024 / isteps,
                           "algorithmic_bytes_per_item_step": stage_bytes[fam], "dispatches": nd[key]}
json.dump(res, open(out, "w"), indent=1)
print("wrote", out, len(res["kernels"]), "kernels")
