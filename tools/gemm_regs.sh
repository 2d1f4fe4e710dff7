#!/bin/bash
# VGPRs / SGPR spills / VGPR spills of every bf16 tt_gemm instance (optionally filtered by a regex on the mangled tile args)
cd "$(dirname "$0")/../this_and_that_vdm_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-result -c gemm.hip -o /tmp/gemm_regs.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs:|VGPRs Spill|SGPRs Spill" | paste - - - - |
  sed -E 's/.*gemm_kernelI//; s/EEEvNS_5GemmPE//; s/gemm.hip:[0-9]+:[0-9]+: remark: +//g; s/\[-Rpass-analysis=kernel-resource-usage\]//g' |
  grep "^8bf16" | sort -u | awk '{print $1, "vgpr", $3, "sgpr_spill", $6, "vgpr_spill", $9}' | grep -E "${1:-.}"
