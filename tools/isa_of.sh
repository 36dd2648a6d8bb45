#!/bin/bash
# Disassembly of the device functions of a built liblins_ieskf.so whose mangled name matches a pattern:
#   tools/isa_of.sh <pattern> [lib.so]  ->  /tmp/isa_of/<n>.s per function + a scratch / spill summary per function
pat=$1; lib=${2:-$(dirname $0)/../lins---lidar-inertial-slam_amd/liblins_ieskf.so}
tmp=$(mktemp -d); cp "$lib" $tmp/lib.so; rm -rf /tmp/isa_of; mkdir -p /tmp/isa_of
(cd $tmp && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null 2>&1
 for f in lib.so.*gfx950*; do /opt/rocm/lib/llvm/bin/llvm-objdump -d $f; done > all.s
 awk -v pat="$pat" '/^[0-9a-f]+ <.*>:/{name=$2; on=(name ~ pat); if(on){n++; file=sprintf("/tmp/isa_of/%d.s",n); print name > file}} on{print > file}' all.s)
for f in /tmp/isa_of/*.s; do
  echo "$(head -1 $f | cut -c1-110) : $(grep -c . $f) lines, scratch ld/st $(grep -c scratch_load $f)/$(grep -c scratch_store $f), readlane/writelane $(grep -c v_readlane $f)/$(grep -c v_writelane $f)"
done
rm -rf $tmp
