# how the batch lanes overlap in the timed region of the driver's run: rocprofv3 --kernel-trace of BENCH_TIMED_ONLY=1 bench.py + tools/concurrency.py (gpurun -- 'bash tools/prof_concurrency.sh'; profiles/r04_concurrency.txt)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=gpurun_out/probes; mkdir -p $T
BENCH_TIMED_ONLY=1 rocprofv3 --kernel-trace -d $T/prof_conc -o r -- python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline > $T/bench_conc.json 2>$T/bench_conc.err
db=$(find $T/prof_conc -name '*.db' | head -1)
python tools/concurrency.py $db --after-first k_accumulate_home > $T/concurrency_all.txt
python tools/rocpd_summary.py --timeline $db > $T/timeline_conc.txt
python - <<PY > $T/db_schema.txt
import sqlite3,sys
db=sqlite3.connect("$db")
print([r[1] for r in db.execute("pragma table_info(kernels)").fetchall()])
PY
rm -rf $T/prof_conc
cat $T/concurrency_all.txt $T/db_schema.txt; tail -1 $T/bench_conc.json | cut -c1-300; tail -3 $T/bench_conc.err
