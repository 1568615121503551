#!/bin/bash
# which kernel makes the 2-rank full-size test flaky?  the same test 6 x per library variant
set -u
for v in oldboth oldattn oldtn; do
  echo "== variant $v"
  for i in 1 2 3 4 5 6; do
    VF_HIP_LIB=$PWD/viewformer_amd/variants/libvf_$v.so timeout 200 python -m pytest tests/test_hip_train_full.py -m gpu -q -s -k two_ranks 2>&1 | grep -E "^\{|passed|failed" | sed -E "s/.*'e_sum': ([0-9.e-]+).*'worst_sum': \(([^)]*)\).*/e_sum \1 worst \2/" | cut -c1-150 | tr '\n' ' '; echo
  done
done
