#!/bin/bash
for b in depthfirst banded supertile; do echo "== $b"; FG_BINNING=$b bash scripts/gpu_dbg.sh 2>&1 | grep "^gc" | awk '{ if ($8 != $12) print }' | head -4; done
