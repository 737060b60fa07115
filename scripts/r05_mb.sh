#!/bin/bash
mkdir -p gpurun_out/r05_mb
MB_PROF=1 timeout 900 python scripts/mb_gelsd.py $1 > gpurun_out/r05_mb/mb_gelsd.txt 2>&1
tail -60 gpurun_out/r05_mb/mb_gelsd.txt
