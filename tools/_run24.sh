#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "overflow or wide" -s 2>&1 | grep -v "^$" | tail -12
