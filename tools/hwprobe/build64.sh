#!/bin/sh
# builds attn64 and prints per-kernel resource usage + hot-loop scratch statistics
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-honor-nans -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 -Rpass-analysis=kernel-resource-usage attn64.hip -o bin/attn64 -save-temps=obj 2>&1 | grep -E "error|Function Name|VGPRs:|Spill|ScratchSize" | grep -A4 "attn64_kernel" | grep -v AGPR
