"""Per-kernel, per-launch averages of every counter in a set of rocprofv3 `--pmc ... --output-format csv` passes
(one pass per directory, as scripts/gpu_profile_r02.sh writes them) -> one JSON.
usage: pmc_counters_summary.py <out.json> <counter_collection.csv> [...]"""
import collections
import csv
import json
import re
import sys

KERNELS = ["raster_bwd_kernel", "raster_fwd_kernel", "raster_fwd_mfma_kernel", "raster_bwd_geom_mfma_kernel", "sh_fwd_kernel",
           "sh_bwd_kernel", "project_fwd_kernel", "project_bwd_kernel", "isect_emit_tight_kernel",
           "isect_count_tight_kernel"]

out = collections.OrderedDict()
for path in sys.argv[2:]:
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        name = next((k for k in KERNELS if re.search(r"\b%s\b" % k, r["Kernel_Name"])), None)
        if name is None:
            continue
        tot[name][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[name].add(r["Dispatch_Id"])
    for name, counters in tot.items():
        out.setdefault(name, {}).update({c: v / len(launches[name]) for c, v in counters.items()})
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k in ("raster_bwd_kernel", "raster_fwd_kernel"):
    if k in out:
        print(k, json.dumps(out[k]))
