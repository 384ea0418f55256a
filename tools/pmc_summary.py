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
        dur = collections.defaultdict(dict)  # kernel -> dispatch -> ns (the launch durations of THIS pass: GRBM_GUI_ACTIVE / 8 XCDs / duration = shader clock)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            name = next((n for n in ("render_small_kernel", "render_shaded_cells_kernel", "render_shaded_kernel", "render_adjoint_kernel", "render_stripe_kernel", "render_cells_kernel",
                                     "render_kernel", "evaluate_small_kernel", "evaluate_gradient_kernel", "evaluate_kernel") if n in k), None)
            if name is None:
                continue
            k = k[k.index(name):].split("(")[0]
            acc[(k, row["Counter_Name"])][row["Dispatch_Id"]] += float(row["Counter_Value"])
            if row["Counter_Name"] == "GRBM_GUI_ACTIVE" and row.get("End_Timestamp"):
                dur[k][row["Dispatch_Id"]] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
        for (k, c), per in sorted(acc.items()):
            print("%s,\"%s\",%s,%d,%.1f" % (d.split("/")[-1], k, c, len(per), sum(per.values()) / len(per)))
        for k, per in sorted(dur.items()):
            print("%s,\"%s\",kernel_ms_avg_under_pmc,%d,%.6f" % (d.split("/")[-1], k, len(per), 1e-6 * sum(per.values()) / len(per)))
