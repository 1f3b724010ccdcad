#!/bin/bash
# compile fam_kincar_wave.hip and print the resource usage of every instance
cd /root/repo/ntg_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -c fam_kincar_wave.hip -o fam_kincar_wave.o -I ../../include -Wno-unused-result -Wno-unused-value -Wno-pass-failed -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | grep -E "error|Function Name|VGPRs:|AGPRs|Scratch|Spill|Occupancy" | sed 's/.*remark: //; s/\[-Rpass.*//' | sed 's/Function Name: _ZN4ntgw15sqp_wave_kernelI/-- /; s/EEv7NtgDims.*//'
