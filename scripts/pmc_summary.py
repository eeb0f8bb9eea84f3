import csv, glob, collections, sys
quads = float(sys.argv[1]) if len(sys.argv) > 1 else 4.096e6
for d in sys.argv[2:]:
    fs = glob.glob(f'gpurun_out/{d}/runc/*_counter_collection.csv')
    if not fs: print(d, 'missing'); continue
    rows = list(csv.DictReader(open(sorted(fs)[-1])))
    agg = collections.defaultdict(list)
    for r in rows:
        if "k_sweep" in r["Kernel_Name"]:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in sorted(agg.items()):
        print(f"{d} {k:28s} last={v[-1]:.4g} per_quad={v[-1]/quads:.2f}")
