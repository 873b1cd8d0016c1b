cd $GRAFT_REPO_ROOT
try() { # name env count selector
  for k in $(seq 1 $3); do
    env $2 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -p no:faulthandler > /tmp/out.txt 2>/tmp/err.txt; rc=$?
    echo "$1 [$2] run $k: rc=$rc $(tail -c 80 /tmp/out.txt | tr '\n' ' ')"
  done
}
try staged A=1 8
