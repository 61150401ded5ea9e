#!/bin/bash
# The network-level GPU tests under every scheduling knob's fallback value (each knob selects a slower path that must stay correct).
cd "${GRAFT_REPO_ROOT:-.}"
for V in "LIDIFF_SINGLE_READ=0" "LIDIFF_LAZY_XT=0" "LIDIFF_ENCODE_AHEAD=0" "LIDIFF_OVERLAP_MAPS=0" "LIDIFF_OVERLAP_MAPS=lazy" "LIDIFF_UP_ORDERED=0" "LIDIFF_CONV_FLAGS=8" "LIDIFF_CONV_FLAGS=16" "LIDIFF_FUSED_BN=0" "LIDIFF_DETERMINISTIC_SCATTER=0" "LIDIFF_MATCHES_AHEAD=0"; do
  echo "== $V"
  env $V timeout 900 python -m pytest tests/test_gpu_network.py -m gpu -q -x -k "not gloo and not reproducible" 2>&1 | tail -1
done
