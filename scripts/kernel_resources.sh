#!/bin/bash
# usage: kres.sh file.o pattern  -- prints register / LDS / scratch use of the kernels in an object file
obj=$1; pat=$2
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=/tmp/kres_fat.bin $obj
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=/tmp/kres_fat.bin --output=/tmp/kres.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes /tmp/kres.co | python3 -c "
import sys,re,subprocess
txt=sys.stdin.read()
blocks=txt.split('- .agpr_count:')
pat=re.compile(sys.argv[1])
for b in blocks[1:]:
    name=re.search(r'\.name:\s+(\S+)',b)
    if not name: continue
    n=name.group(1)
    dn=subprocess.run(['c++filt',n],capture_output=True,text=True).stdout.strip()
    if not pat.search(dn): continue
    def g(k):
        m=re.search(r'\.'+k+r':\s+(\d+)',b); return m.group(1) if m else '?'
    agpr=b.strip().split()[0]
    print(dn.split('(')[0][:100], '| vgpr',g('vgpr_count'),'agpr',agpr,'sgpr',g('sgpr_count'),'vspill',g('vgpr_spill_count'),'sspill',g('sgpr_spill_count'),'lds',g('group_segment_fixed_size'),'scratch',g('private_segment_fixed_size'))
" "$pat"
