#!/bin/bash
cd /root/repo
python tools/wide_diag.py 2>&1 | tail -1 | cut -c1-400
RTGPU_WIDE_DIAG=1 python tools/wide_diag.py 2>&1 | tail -5 | cut -c1-600
