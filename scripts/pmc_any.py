"""Per-launch averages of every counter in rocprofv3 `--pmc ... --output-format csv` passes for kernels whose name
matches a regular expression.  usage: pmc_any.py <regex> <dir-or-csv> [...]"""
import collections, csv, glob, json, os, re, sys
pat = re.compile(sys.argv[1])
out = collections.OrderedDict()
for arg in sys.argv[2:]:
    paths = [arg] if arg.endswith(".csv") else glob.glob(os.path.join(arg, "**", "*counter_collection.csv"), recursive=True)
    for path in paths:
        tot = collections.defaultdict(lambda: collections.defaultdict(float))
        launches = collections.defaultdict(set)
        for r in csv.DictReader(open(path)):
            m = pat.search(r["Kernel_Name"])
            if not m:
                continue
            name = re.sub(r"\(.*", "", r["Kernel_Name"])[-70:]
            tot[name][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[name].add(r["Dispatch_Id"])
        for name, counters in tot.items():
            out.setdefault(name, {}).update({c: v / len(launches[name]) for c, v in counters.items()})
print(json.dumps(out, indent=1))
