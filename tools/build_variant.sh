#!/bin/bash
# usage: tools/build_variant.sh name "-DFLAG ..."   -> exp/lib_<name>.so (tuning builds; RRTMG_LW_HIP_LIB selects one at run time)
name=$1; shift
cd /root/repo/rrtmg_lw_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -mllvm -amdgpu-remove-redundant-endcf=0 $@ driver.hip -o /root/repo/exp/lib_$name.so 2>&1 | grep -E "error" | head -5
echo "built $name $@"
