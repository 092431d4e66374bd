"""Instruction-class census of the pipeline kernels' gfx950 code (diagnostic):
   python tools/isa_count.py elem.s 'post_backward_a_vec_kernelILi1E' ...   (elem.s from hipcc -S --cuda-device-only)"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
pats = sys.argv[2:]
starts = [(m.start(), m.group(1)) for m in re.finditer(r'^(_ZN2gr\w+):', s, flags=re.M)]
for i, (pos, name) in enumerate(starts):
    if not any(p in name for p in pats):
        continue
    end = s.find('.Lfunc_end', pos)
    body = s[pos:end]
    ops = re.findall(r'^\s+([vs]_\w+|ds_\w+|global_\w+|buffer_\w+)', body, flags=re.M)
    c = Counter(ops)
    v = sum(n for k, n in c.items() if k.startswith('v_'))
    print(f"{name[:100]}\n   lines {len(ops)} VALU {v} f64 {sum(n for k, n in c.items() if 'f64' in k)} pk {sum(n for k, n in c.items() if k.startswith('v_pk'))} "
          f"exp {c.get('v_exp_f32', 0)} cvt {sum(n for k, n in c.items() if k.startswith('v_cvt'))} cndmask {c.get('v_cndmask_b32', 0)} "
          f"salu {sum(n for k, n in c.items() if k.startswith('s_'))} branches {sum(n for k, n in c.items() if k.startswith('s_cbranch'))} "
          f"gload {sum(n for k, n in c.items() if k.startswith(('global_load', 'buffer_load')))} gstore {sum(n for k, n in c.items() if k.startswith(('global_store', 'buffer_store')))} lds {sum(n for k, n in c.items() if k.startswith('ds_'))}")
    m = re.search(r'\.vgpr_count:\s+(\d+)', s[s.find(name, end):] if False else '')
