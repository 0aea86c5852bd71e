#!/bin/bash
# usage: tools/dis_count.sh NAME [extra hipcc flags]  -- disassemble the cbox-only variants and count VALU / packed / mov
name=$1; shift
cd /root/repo/pine_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 --cuda-device-only -S -DPINE_ONLY_CBOX_VARIANT -w "$@" -x hip pine_kernels.hip -o ../../build/dis/$name.s || exit 1
python3 - "$name" <<'PY'
import re,sys
s=open('/root/repo/build/dis/%s.s'%sys.argv[1]).read()
for f in re.split(r'\n(?=_Z[\w]+:\s)', s):
    name=f.split(':',1)[0]
    if 'path_queue_kernel' in name or 'path_trace_kernel' in name:
        c=lambda pat: len(re.findall(pat, f, flags=re.M))
        print(name[10:45], 'valu', c(r'^\s*v_'), 'pk', c(r'^\s*v_pk_'), 'mov', c(r'^\s*v_mov_b32'), 'salu', c(r'^\s*s_'), 'lds', c(r'^\s*ds_'))
for m in re.finditer(r'\.name:\s+(\S*path_queue\S*)\n(.*?)\.wavefront_size', s, flags=re.S):
    print(re.findall(r'\.(?:sgpr_spill_count|vgpr_spill_count|vgpr_count|sgpr_count):\s*\d+', m.group(2)))
PY
