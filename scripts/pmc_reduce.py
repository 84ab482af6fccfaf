"""Per-launch means of the PMC counters of the one-system PCG SpMV k_spmv_span<8, true, 0, double, 1, false> from rocprofv3 counter_collection CSVs."""
import csv, glob, json, os, sys
src, dst = sys.argv[1], sys.argv[2]
acc = {}
for f in glob.glob(os.path.join(src, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_spmv_span<8, true, 0, double, 1, false>" not in r["Kernel_Name"]:  # (the one-system PCG SpMV)
            continue
        a = acc.setdefault(r["Counter_Name"], [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
out = {k: v[1] / v[0] for k, v in acc.items()}
out["dispatches_per_counter"] = {k: v[0] for k, v in acc.items()}
alg = 848797852
rd = out.get("TCC_EA0_RDREQ_sum", 0.0) * 128.0   # gfx950: 128-B requests (TCC_EA0_RDREQ_32B = 0)
wr = out.get("WRITE_SIZE", 0.0) * 1024.0           # WRITE_SIZE is in KiB
out.update(hbm_read_bytes_corrected=rd, hbm_write_bytes=wr, algorithmic_bytes=alg,
           traffic_over_algorithmic=(rd + wr) / alg,
           note="rocprofv3 --pmc passes (one counter group per run, --kernel-trace only; scripts/pmc_collect.sh) on "
                "k_spmv_span<8,true,0>, config-3 matrix (100k vertices / 1M edges), per-launch means",
           correction="gfx950: FETCH_SIZE = TCC_EA0_RDREQ x 64 B under-counts the 128-B requests of a wide stream by 2x "
                      "(MI355X_MICROARCH.md, HBM section); read bytes = RDREQ x 128 B (TCC_EA0_RDREQ_32B = 0)")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out))
