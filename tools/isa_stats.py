"""Register / spill / instruction statistics of device functions in the saved ISA (make -C bblean_amd/csrc asm).
    python tools/isa_stats.py <substring of the mangled name> [...]"""
import re, sys
from collections import Counter
s = open('/root/repo/build/asm/bb_tree-hip-amdgcn-amd-amdhsa-gfx950.s').read()
for pat in sys.argv[1:]:
    for m in re.finditer(r'^(\S*' + re.escape(pat) + r'\S*):', s, re.M):
        name = m.group(1)
        i = m.start(); j = s.index('.Lfunc_end', i); body = s[i:j]
        ins = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith((';', '.')) and not l.strip().endswith(':')]
        c = Counter(l.split()[0] for l in ins)
        res = {k: v for k, v in re.findall(r'\.set ' + re.escape(name) + r'\.(num_vgpr|num_agpr|numbered_sgpr|private_seg_size), (\S+)', s)}
        print(name[-70:], res, 'instructions', len(ins))
        print('   ', {k: c.get(k, 0) for k in ['v_writelane_b32', 'v_readlane_b32', 'v_readfirstlane_b32', 's_barrier', 'scratch_store_dword', 'scratch_load_dword', 'scratch_load_dwordx4', 'scratch_store_dwordx4', 'v_accvgpr_read_b32', 'v_accvgpr_write_b32', 's_nop']})
        open('/tmp/' + pat.replace('/', '_') + '.s', 'w').write(body)
