cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=gpurun_out/s5; mkdir -p $T
python -m pytest tests -q -m gpu 2>&1 | grep "passed\|failed\|error\|^FAILED\|^ERROR\|assert\|Error" | head -20 > $T/pytest_gpu_packet.log; cat $T/pytest_gpu_packet.log
for n in 8 4; do for e in 0 1; do RTGPU_PACKET=$e BENCH_EMULATE_SHARD=$n python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('shard 1/$n packets $e %8.1f Msamples/s %.3f ms/pass' % (d['value'], d['ms_per_step']))"; done; done
for w in cornell sponza-textured sponza-all; do for e in 0 1; do RTGPU_PACKET=$e python bench.py --workload $w --steps 32 --warmup 4 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$w packets $e %8.1f Msamples/s %.3f ms/pass' % (d['value'], d['ms_per_step']))"; done; done
