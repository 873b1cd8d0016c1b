cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=gpurun_out/s5; mkdir -p $T
python -m pytest tests/test_gpu_parity.py -q -x -m gpu 2>&1 | grep "passed\|failed\|error\|^FAILED\|^ERROR\|assert\|Error" | head -20 > $T/pytest_packet.log; cat $T/pytest_packet.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "^smoke"
bash tools/ab_env.sh "--steps 64 --warmup 5" RTGPU_PACKET=0 RTGPU_PACKET=1 > $T/ab_packet_64.txt; cat $T/ab_packet_64.txt | cut -c1-250
bash tools/ab_env.sh "--steps 20 --warmup 5" RTGPU_PACKET=0 RTGPU_PACKET=1 > $T/ab_packet_20.txt; cat $T/ab_packet_20.txt | cut -c1-250
