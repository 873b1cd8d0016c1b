#!/bin/bash
cd /root/repo
echo "--- wide sorted"; RTGPU_WIDE=1 RTGPU_WIDE_DIAG=1 RTGPU_WIDE_SORT=1 python tools/wide_diag.py 2>&1 | tail -4
echo "--- wide sorted, 1 lane"; RTGPU_LANES=1 RTGPU_WIDE=1 RTGPU_WIDE_DIAG=1 RTGPU_WIDE_SORT=1 python tools/wide_diag.py 2>&1 | tail -4
