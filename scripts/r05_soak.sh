#!/bin/bash
# long GPU soaks of the stability settings against the oracle's independent restatement, per solver mode
mkdir -p gpurun_out/profiles_r05
{
  PCT_LSTSQ=gelsd timeout 1200 python scripts/soak_parity.py discrete_s1 4096 8000
  PCT_LSTSQ=gelsd timeout 1200 python scripts/soak_parity.py continuous_s1 4096 3000
  PCT_LSTSQ=gelsd_avx2 timeout 900 python scripts/soak_parity.py discrete_s1 4096 2000
  PCT_LSTSQ=gelsd_avx2 timeout 900 python scripts/soak_parity.py continuous_s1 4096 1000
  PCT_LSTSQ=jacobi timeout 900 python scripts/soak_parity.py discrete_s1 4096 2000
} > gpurun_out/profiles_r05/r05_soak_parity_modes.txt 2>&1
cat gpurun_out/profiles_r05/r05_soak_parity_modes.txt | grep -v amdgpu.ids
