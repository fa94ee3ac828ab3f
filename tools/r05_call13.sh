#!/bin/bash
# round 5, final GPU call: the driver's suite (-x) + smoke at HEAD, then the round's profile set (tools/profile_r05.sh)
bash tools/r05_verify.sh
bash tools/profile_r05.sh all > gpurun_out/r05_profile.log 2>&1
tail -30 gpurun_out/r05_profile.log | cut -c1-200
