"""Turns two rocprofv3 PMC passes over `bench.py --steps 3 --warmup 1` (one with `--pmc FETCH_SIZE`, one with
`--pmc WRITE_SIZE`, each `--output-format csv`) into profiles/pmc_traffic.json: HBM bytes per launch of every
library entry point, bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.  (On gfx950 FETCH_SIZE reports half of wide
coalesced reads; checked on sh_fwd_kernel, whose input size is known.)
usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <workload key> <out.json>"""
import collections
import csv
import json
import sys

ENTRY = {"raster_bwd_kernel": "pr_raster_bwd", "raster_fwd_kernel": "pr_raster_fwd", "raster_fwd_mfma_kernel": "pr_raster_fwd", "sh_fwd_kernel": "pr_sh_fwd",
         "sh_bwd_kernel": "pr_sh_bwd", "project_fwd_kernel": "pr_project_fwd", "project_bwd_kernel": "pr_project_bwd",
         "isect_emit_tight_kernel": "isect_emit_tight_kernel", "isect_count_tight_kernel": "isect_count_tight_kernel"}


def per_launch(path, counter):
    tot, launches = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for frag, name in ENTRY.items():
            if frag in r["Kernel_Name"]:
                tot[name] += float(r["Counter_Value"])
                launches[name].add(r["Dispatch_Id"])
    return {k: tot[k] / len(launches[k]) for k in tot}


fetch, write = per_launch(sys.argv[1], "FETCH_SIZE"), per_launch(sys.argv[2], "WRITE_SIZE")
out = {"workload": sys.argv[3],
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of `bench.py --steps 3 --warmup 1`; "
                 "per-launch average; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE reports half of "
                 "wide coalesced reads; checked on sh_fwd_kernel against its known input size)",
       "raw_kb": {k: {"FETCH_SIZE": fetch.get(k), "WRITE_SIZE": write.get(k)} for k in sorted(set(fetch) | set(write))}}
for k in out["raw_kb"]:
    out[k] = int((2 * (fetch.get(k) or 0) + (write.get(k) or 0)) * 1024)
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k not in ("method", "raw_kb")}))
