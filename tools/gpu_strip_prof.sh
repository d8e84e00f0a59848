#!/bin/bash
# rocprofv3 passes over tools/strip_prof.py: kernel trace + stats, then HBM / L2 counters in passes of their own
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
REPO="$GRAFT_REPO_ROOT"
rm -rf gpurun_out/sprof
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/sprof" -o st -- python3 "$REPO/tools/strip_prof.py" > /dev/null 2> "$REPO/gpurun_out/sprof.err") || { tail -3 gpurun_out/sprof.err; exit 1; }
for f in $(find gpurun_out/sprof -name "*kernel_stats*.csv"); do cp $f gpurun_out/strip_kernel_stats.csv; done
for f in $(find gpurun_out/sprof -name "*kernel_trace*.csv"); do cp $f gpurun_out/strip_kernel_trace.csv; done
pass() { name=$1; shift
  rm -rf gpurun_out/spmc_$name
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$REPO/gpurun_out/spmc_$name" -o $name -- python3 "$REPO/tools/strip_prof.py" > /dev/null 2> "$REPO/gpurun_out/spmc_$name.err") || { tail -3 gpurun_out/spmc_$name.err; return 1; }
}
pass fetch FETCH_SIZE || exit 1
pass write WRITE_SIZE || exit 1
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum || echo "(no TCC counters)"
python3 - <<'PY'
import csv, glob, collections, json
out = {}
tr = glob.glob("gpurun_out/strip_kernel_trace.csv")
dur = collections.defaultdict(list)
if tr:
    for r in csv.DictReader(open(tr[0])):
        dur[r["Kernel_Name"][:48]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for name in ("fetch", "write", "tcc"):
    for f in glob.glob("gpurun_out/spmc_%s/**/*counter_collection.csv" % name, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:48]][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for k, v in agg.items():
            for c, vals in v.items():
                per = collections.defaultdict(float)
                for d, x in vals: per[d] += x
                out.setdefault(k, {})[c] = [per[d] for d in sorted(per)]
for k in out: out[k]["ms"] = dur.get(k, [])
json.dump(out, open("gpurun_out/strip_pmc.json", "w"), indent=1)
for k, v in out.items():
    if "strip" in k or "mfe_kernel" in k or "pf_kernel" in k:
        print(k, {c: [round(x, 1) for x in vals[:12]] for c, vals in v.items()})
PY
