"""Summarise rocprofv3 counter_collection.csv files: per-launch averages for the render kernel."""
import csv, glob, sys, collections

root, tag = sys.argv[1], sys.argv[2]
print("# rocprofv3 --pmc (separate passes, counters only), per-launch averages of the render kernel; %s" % tag)
print("# SQ_* are summed over the chip; FETCH_SIZE / WRITE_SIZE in KiB as reported (see MI355X_MICROARCH.md for the gfx950 corrections)")
print("pass,kernel,counter,launches,avg_per_launch")
for d in sorted(glob.glob(root + "/pmc*")):
    if not d.rstrip("/").split("/")[-1].startswith("pmc") or d.endswith(".log"):
        continue
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            name = next((n for n in ("render_small_kernel", "render_shaded_kernel", "render_adjoint_kernel", "render_stripe_kernel", "render_kernel") if n in k), None)
            if name is None:
                continue
            k = k[k.index(name):].split("(")[0]
            acc[(k, row["Counter_Name"])][row["Dispatch_Id"]] += float(row["Counter_Value"])
        for (k, c), per in sorted(acc.items()):
            print("%s,\"%s\",%s,%d,%.1f" % (d.split("/")[-1], k, c, len(per), sum(per.values()) / len(per)))
