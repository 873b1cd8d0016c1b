# FETCH_SIZE / WRITE_SIZE calibration for this library's access patterns (tools/microbench/fetch_calib.hip): one --pmc pass per counter, then the factors.
# gpurun -- 'bash tools/fetch_calib.sh'   -> gpurun_out/fetch_calib/calibration.json (copied to profiles/r05_fetch_calibration.json)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=gpurun_out/fetch_calib
mkdir -p $T
[ -x tools/microbench/fetch_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/microbench/fetch_calib.hip -o tools/microbench/fetch_calib
tools/microbench/fetch_calib > $T/known_bytes.json
for counter in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  name=$(echo $counter | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $counter -d $T/p_$name -o r -- tools/microbench/fetch_calib > /dev/null 2> $T/err_$name.txt
  db=$(find $T/p_$name -name '*.db' | head -1)
  if [ -n "$db" ]; then python tools/rocpd_summary.py $db > $T/pmc_$name.txt; else echo "no db for $counter"; tail -3 $T/err_$name.txt; fi
  rm -rf $T/p_$name $T/err_$name.txt
done
python - <<'PY'
import json, re, os
T = "gpurun_out/fetch_calib"
known = json.load(open(T + "/known_bytes.json"))
def counters(path):
    out = {}
    if not os.path.exists(path): return out
    for line in open(path):
        m = re.match(r"(?:void )?(k_calib_\w+)\s+(\S+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)", line)
        if m: out.setdefault(m.group(1), {})[m.group(2)] = float(m.group(5))
    return out
c = {}
for f in os.listdir(T):
    if f.startswith("pmc_"):
        for k, v in counters(T + "/" + f).items(): c.setdefault(k, {}).update(v)
res = {"table_bytes": known["table_bytes"], "lanes": known["lanes"], "kernels": {}}
for k, kb in known["kernels"].items():
    e = dict(kb); e["counters_per_launch"] = c.get(k, {})
    fs = c.get(k, {}).get("FETCH_SIZE"); ws = c.get(k, {}).get("WRITE_SIZE")
    if fs and "store" not in k:
        e["FETCH_SIZE_bytes"] = fs * 1024.0   # rocprofv3 reports KiB
        e["requested_over_FETCH_SIZE"] = kb["requested"] / (fs * 1024.0); e["lines64_over_FETCH_SIZE"] = kb["lines64"] / (fs * 1024.0); e["lines128_over_FETCH_SIZE"] = kb["lines128"] / (fs * 1024.0)
    if ws and "store" in k:
        e["WRITE_SIZE_bytes"] = ws * 1024.0
        e["requested_over_WRITE_SIZE"] = kb["requested"] / (ws * 1024.0); e["lines64_over_WRITE_SIZE"] = kb["lines64"] / (ws * 1024.0)
    res["kernels"][k] = e
json.dump(res, open(T + "/calibration.json", "w"), indent=1)
for k, e in res["kernels"].items():
    print(k, {x: (round(y, 3) if isinstance(y, float) and y < 1e4 else y) for x, y in e.items() if "over" in x or x == "counters_per_launch"})
PY
