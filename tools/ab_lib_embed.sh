#!/bin/bash
# same-box A/B of two builds of libganrev.so over the cfg5 pipeline (tools/ab_embed.py): A = tools/probe/libganrev_base.so, B = tools/probe/libganrev_new.so
for i in 1 2; do
  for arm in A B; do
    if [ $arm = A ]; then cp tools/probe/libganrev_base.so gan-reverser_amd/ganrev/libganrev.so; else cp tools/probe/libganrev_new.so gan-reverser_amd/ganrev/libganrev.so; fi
    echo "== $arm"; python tools/ab_embed.py 25600 1 2>/dev/null | grep -E "eval_p16=1:|fewin_p16o|quad_po|up2"
  done
done
cp tools/probe/libganrev_new.so gan-reverser_amd/ganrev/libganrev.so
