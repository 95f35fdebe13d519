#!/bin/bash
# Register / LDS / scratch use of the kernels of one translation unit whose name matches a pattern (compiler remarks).
# usage: kernel_regs.sh <file.hip> <grep pattern>
cd "$(dirname "$0")/../grapes_amd/csrc" || exit 1
EXTRA=""; [ "$1" = "sampler_kernels.hip" ] && EXTRA="-ffp-contract=off"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $EXTRA -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/_regs.o 2>&1 |
  awk '/Function Name:/ {name=$0} /VGPRs:|SGPRs:|ScratchSize|LDS Size|Occupancy/ {print name " | " $0}' | sed -e 's/.*Function Name: //' -e 's/remark: [^ ]* //' | grep -E "$2" | awk -F'|' '{n=$1; sub(/ +$/,"",n); gsub(/^.*\[-Rpass-analysis=kernel-resource-usage\]/,"",$2); a[n]=a[n] $2 ";"} END {for (k in a) print k ": " a[k]}'
