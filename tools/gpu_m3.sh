#!/bin/bash
cd $GRAFT_REPO_ROOT
for m in XDeepFM DeepFM AutoInt; do
for g in "" "--no-graph"; do
/usr/bin/time -f "$m $g wall %e s" python examples/train_ctr.py --model $m --steps 120 $g 2>&1 | grep -E "step  *(0|100|119)|wall|Error|error" | tr '\n' ';'; echo
done; done
