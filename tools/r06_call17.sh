#!/bin/bash
# round 6, call 17: dK / dV kernel variants (slam_attn_set_fwd_qf 70 lockstep | 71 / 72 SIMD partners one section apart | 73 softmax under the next tile's first products) --
# bit identity + isolated A/B (tools/attn_pp_ab.py); cycle stamps (tools/attn_dkdv_probe.py)
O=gpurun_out/r06_call17; mkdir -p $O
timeout 600 python tools/attn_pp_ab.py > $O/attn_pp_ab.json 2> $O/attn_pp_ab.err; echo "rc $?"; cat $O/attn_pp_ab.err | tail -8
