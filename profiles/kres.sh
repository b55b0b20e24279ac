#!/bin/bash
# usage: profiles/kres.sh <object.o> : per-kernel VGPR / SGPR / spill / LDS / scratch of the gfx950 code object inside a hipcc object
L=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$L/llvm-objcopy -O binary --only-section=.hip_fatbin "$1" $T/fat.bin
$L/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co 2>/dev/null || \
$L/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hip-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co
$L/llvm-readelf --notes $T/dev.co | awk '/\.name:/{n=$2} /\.vgpr_count:/{v=$2} /\.sgpr_count:/{s=$2} /\.vgpr_spill_count:/{vs=$2} /\.sgpr_spill_count:/{ss=$2} /\.group_segment_fixed_size:/{l=$2} /\.private_segment_fixed_size:/{p=$2} /\.agpr_count:/{a=$2} /\.wavefront_size:/{print n, "vgpr",v,"agpr",a,"sgpr",s,"vspill",vs,"sspill",ss,"lds",l,"scratch",p}' | while read n rest; do echo "$(echo $n | c++filt) $rest"; done
rm -rf $T
