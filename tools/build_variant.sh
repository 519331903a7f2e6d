#!/bin/bash
# usage: tools/build_variant.sh name "-DFLAG ..."   -> exp/lib_<name>.so + register table
name=$1; shift
d=/tmp/bv_$name; rm -rf $d; mkdir -p $d
cd /root/repo/rrtmg_lw_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC $@ driver.hip -save-temps=obj -o $d/lib.so 2>&1 | grep -E "error|occupancy" | head -5
cp $d/lib.so /root/repo/exp/lib_$name.so
echo "== $name $@"
grep -E "^\s+\.(vgpr_count|name:|vgpr_spill_count)" $d/driver-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - | grep -E "k_sweepILi[0-4]ELi[14]ELb[01]|k_layerILb1ELi[0-3]ELi0" | sed -e 's/_ZN4rrlw//' -e 's/EEEvNS_9DevTablesENS_9Workspace.*E\t/\t/' | awk '{print $2, $4, $6}'
rm -rf $d
