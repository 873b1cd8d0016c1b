# blocks per CU of the persistent trace kernel x batch lanes (the rest of the CU is for the other lanes' kernels)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for cfg in "5 3" "4 3" "4 4" "3 4" "5 4"; do set -- $cfg
  echo -n "blocks $1 lanes $2: "; RTGPU_TRAV_BLOCKS_PER_CU=$1 RTGPU_LANES=$2 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1))"
done; done
