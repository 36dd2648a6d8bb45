#!/bin/bash
# Register / scratch metadata of the IESKF kernels in a built liblins_ieskf.so and the number of scratch
# instructions of the batch kernel (the "mr" instantiation).  usage: tools/kernel_regs.sh [lib.so]
lib=${1:-$(dirname $0)/../lins---lidar-inertial-slam_amd/liblins_ieskf.so}
tmp=$(mktemp -d)
cp "$lib" $tmp/lib.so
(cd $tmp && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null 2>&1)
for f in $tmp/lib.so.*gfx950; do
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$f" 2>/dev/null | grep -E "^ +\.name:|\.vgpr_count|vgpr_spill|private_segment_fixed|\.sgpr_count:" | paste - - - - - |
    grep -E "ieskf_lds_kernel" | sed -E 's/_ZN4lins[0-9]*//; s/16ieskf_lds_kernelI/ </; s/EEEvNS[^ \t]*//; s/[ \t]+/ /g'
  if grep -q "lds_mr16ieskf_lds_kernelILi512ELi1ELb0ELb0ELb0ELb0E" <(/opt/rocm/lib/llvm/bin/llvm-objdump -d "$f" | grep "^[0-9a-f]* <"); then
    /opt/rocm/lib/llvm/bin/llvm-objdump -d "$f" | awk '/^[0-9a-f]+ <_ZN4lins6lds_mr16ieskf_lds_kernelILi512ELi1ELb0ELb0ELb0ELb0E/{p=1} p{print} /s_endpgm/{if(p){exit}}' > $tmp/mr.s
    echo "batch kernel: $(grep -c scratch_load $tmp/mr.s) scratch loads, $(grep -c scratch_store $tmp/mr.s) scratch stores, $(wc -l < $tmp/mr.s) lines of ISA"
    cp $tmp/mr.s /tmp/mr_last.s
  fi
done
rm -rf $tmp
