#!/bin/bash
# dev: compile one kernel file of libdkd with the register / scratch report (usage: tools_dev/cc_one.sh attn192_bwd)
cd /root/repo/deltakd_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -c $1.hip -o /tmp/cc_one.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs:|Scratch|VGPRs Spill|warning"
