"""Collect the rocprofv3 PMC passes / floor build written by tools/gpu_round4.sh under gpurun_out/ into
gpurun_out/roofline_inputs.json (per-launch averages for the fold kernels of the L=200, R=64 benchmark batch, plus the
calibration kernels that fix the counters' units).  Copy to profiles/roofline_inputs.json once the kernels are final:
bench.py reads it from there."""
import collections, csv, glob, json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")


def counters(prefix, name):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(G, "%s_%s" % (prefix, name), "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


def key_of(kname):
    if "score_fused" in kname: return "fused"
    if "mfe" in kname: return "mfe"
    if "pf_" in kname: return "pf"
    if "eval" in kname: return "eval"
    return None


def grid_workgroups():
    """workgroups per launch of every kernel, from the kernel trace of a PMC pass"""
    out = {}
    for f in glob.glob(os.path.join(G, "pmc_sq1", "**", "*kernel_trace.csv"), recursive=True) + glob.glob(os.path.join(G, "pmc2_sq1", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            try:
                out[r["Kernel_Name"]] = float(r["Grid_Size_X"]) / float(r["Workgroup_Size_X"])
            except Exception:
                pass
    return out


raw, cal = {}, {}
for p in ("fetch", "write", "sq1", "sq2", "grbm"):
    for pre in ("pmc", "pmc2"):                      # pmc2: the same passes with DRNA_FUSED=0 (the two-launch form's kernels)
        for k, d in counters(pre, p).items():
            raw.setdefault(k, {}).update(d)
    for k, d in counters("cal", p).items():
        cal.setdefault(k, {}).update(d)
if not raw:
    sys.exit("no PMC passes under gpurun_out/")
N_CAL = 256 * 4 * 8192          # wave-instructions per calibration kernel
units = {}
for k, d in cal.items():
    if "calib_lds_b64" in k and "SQ_LDS_IDX_ACTIVE" in d:
        units["SQ_LDS_IDX_ACTIVE_per_ds_read_b64"] = d["SQ_LDS_IDX_ACTIVE"] / N_CAL
        units["SQ_ACTIVE_INST_LDS_per_ds_read_b64"] = d.get("SQ_ACTIVE_INST_LDS", 0) / N_CAL
    if "calib_lds_b32" in k and "SQ_LDS_IDX_ACTIVE" in d:
        units["SQ_LDS_IDX_ACTIVE_per_ds_read_b32"] = d["SQ_LDS_IDX_ACTIVE"] / N_CAL
    if "calib_valu" in k and "SQ_ACTIVE_INST_VALU" in d:
        units["SQ_ACTIVE_INST_VALU_per_v_add"] = d["SQ_ACTIVE_INST_VALU"] / N_CAL
        units["SQ_INSTS_VALU_per_v_add"] = d.get("SQ_INSTS_VALU", 0) / N_CAL
floor = {}
try:
    for line in open(os.path.join(G, "floor.txt")):
        m = re.match(r"skip mask\s+(\d+): mfe ([\d.]+) ms\s+pf ([\d.]+) ms", line)
        if m:
            floor[int(m.group(1))] = {"mfe": float(m.group(2)), "pf": float(m.group(3))}
except FileNotFoundError:
    pass
stats = {}
try:
    for r in csv.DictReader(open(os.path.join(G, "rocprofv3_kernel_stats.csv"))):
        if key_of(r["Name"]):
            stats[key_of(r["Name"])] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"])}
except FileNotFoundError:
    pass
head = os.environ.get("DRNA_COMMIT")        # the GPU box's snapshot has no .git: tools/gpu_round4.sh is started with the hash
if not head:
    try:
        head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        sys.exit("roofline_inputs.py: no commit to record (set DRNA_COMMIT: the counters must name the source they were taken from)")
# the guide (MI355X_MICROARCH.md, LDS): a conflict-free ds_read_b64 / ds_read_b32 wave-instruction takes 2 LDS-array cycles;
# the measured counter value per such instruction converts SQ_LDS_IDX_ACTIVE to LDS-array cycles
lds_unit = 2.0 / units["SQ_LDS_IDX_ACTIVE_per_ds_read_b64"] if units.get("SQ_LDS_IDX_ACTIVE_per_ds_read_b64") else 1.0
valu_unit = 4.0 / units["SQ_ACTIVE_INST_VALU_per_v_add"] if units.get("SQ_ACTIVE_INST_VALU_per_v_add") else 4.0
out = {"source": "tools/gpu_round4.sh pmc + floor passes at commit %s; per-launch averages, R=64 x L=200 uniform batch" % head, "commit": head,
       "units": units, "lds_cycles_per_count": lds_unit, "valu_cycles_per_count": valu_unit, "raw": {}, "kernels": {}}
GW = grid_workgroups()
for k, d in raw.items():
    key = key_of(k)
    if key is None:
        continue
    out["raw"][k] = d
    wgs = GW.get(k, 128.0 if "dual" in k else 64.0)      # one workgroup = one CU; two per sequence in the kernels with a helper
    e = {}
    if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
        # MI355X_MICROARCH.md (HBM): both counters are KB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B -> x2
        e["hbm_bytes_per_launch"] = (2.0 * d.get("FETCH_SIZE", 0.0) + d.get("WRITE_SIZE", 0.0)) * 1024.0
    if "SQ_LDS_IDX_ACTIVE" in d:
        e["lds_busy_cycles_per_cu"] = d["SQ_LDS_IDX_ACTIVE"] * lds_unit / wgs
        e["lds_bank_conflict_share"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, d["SQ_LDS_IDX_ACTIVE"])
    if "SQ_ACTIVE_INST_VALU" in d:
        e["valu_busy_cycles_per_simd"] = d["SQ_ACTIVE_INST_VALU"] * valu_unit / (wgs * 4.0)
    if "GRBM_GUI_ACTIVE" in d and key in stats:
        e["clock_hz_grbm"] = d["GRBM_GUI_ACTIVE"] / 8.0 / (stats[key]["avg_ns"] * 1e-9)
    if 15 in floor:
        # the one-launch kernel holds both folds: its floor is the longer of the two roles'
        e["floor_ms"] = floor[15][key] if key in floor[15] else max(floor[15].values()) if key == "fused" else None
        e["full_ms_same_run"] = floor.get(0, {}).get(key) if key != "fused" else (max(floor[0].values()) if 0 in floor else None)
        e["floor_build"] = "-DDRNA_SKIP=15 of the production launch configuration (tools/phase_cost.py): %s" % k
    if key in stats:
        e["rocprof_avg_ms"] = stats[key]["avg_ns"] * 1e-6
    e["clock_hz"] = 2.4e9
    e["kernel"] = k
    e["workgroups"] = wgs
    out["kernels"]["%s_L200_R64" % key] = e
json.dump(out, open(os.path.join(G, "roofline_inputs.json"), "w"), indent=1)
print(json.dumps({"units": units, "kernels": out["kernels"]}, indent=1))
