#!/bin/bash
one() {  # dir tag
  (cd $1 && timeout -k 10 200 python tools/taper_timing.py 100 0.06 nocpu 2>&1 | grep -E "taper objective|batch" | tr '\n' ' ') | sed "s/^/$2: /"; echo
}
one old_r2_tmp/w_bfe8a35 c1
one old_r2_tmp/w_abb836b c2
one old_r2_tmp/w_cb54c59 c3
one old_r2_tmp/w_2077713 c5
one . new
