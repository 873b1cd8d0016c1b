# how the reference's own stochastic furnace binary fails when it fails (tests/test_cpp_api.py allows ONE repeat for marginal misses): 20 runs, the misses of each
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05t
mkdir -p $T
python - <<'PY' | tee $T/furnace_runs.txt
import os, re, subprocess, sys
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_cpp_api as t
exe = t.REF_EXE
env = dict(os.environ, RT_DATA_DIR=os.path.join(ROOT, "raytracer_amd", "data"))
for run in range(250):
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=900, cwd=os.path.dirname(exe))
    failed = sorted(set(re.findall(r"\[  FAILED  \] (RenderingTest\.\w+)", out.stdout)))
    misses = [(float(d), float(tol)) for d, tol in re.findall(r"is ([0-9.eE+-]+),\s+which exceeds [^\n]*?\n(?:.*\n)*?maxError evaluates to ([0-9.eE+-]+)\.", out.stdout)]
    worst = max((d / tol for d, tol in misses), default=0.0)
    if out.returncode != 0: print("run %3d rc %d failed %s misses %d worst delta/tolerance %.3f" % (run, out.returncode, failed, len(misses), worst), [round(d / tol, 3) for d, tol in misses][:12], flush=True); print(out.stdout[-1200:] if len(misses) == 0 else out.stdout[out.stdout.find("Failure") - 120:][:900], flush=True)
    if out.returncode != 0 and not misses:
        print(out.stdout[-1500:])
PY
