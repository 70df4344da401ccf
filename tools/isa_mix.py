"""Instruction mix of the hottest basic block (most MFMAs / most VALU) of each kernel in a gfx950 .s file."""
import re, collections, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)s_endpgm', txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat not in name:
        continue
    blocks = re.split(r'^\.LBB[0-9_]+:.*$', body, flags=re.M)
    best = max(blocks, key=lambda b: (b.count('v_mfma'), b.count('\n\tv_')))
    c = collections.Counter()
    for line in best.splitlines():
        t = line.strip().split(' ')[0] if line.strip() else ''
        if not t or t[0] in ';.':
            continue
        for k, pre in (('mfma', ('v_mfma',)), ('trans', ('v_exp', 'v_log', 'v_rcp', 'v_rsq', 'v_sqrt', 'v_sin', 'v_cos')),
                       ('accvgpr', ('v_accvgpr',)), ('v_pk', ('v_pk_',)), ('v_mov', ('v_mov',)), ('valu', ('v_',)),
                       ('ds_read', ('ds_read', 'ds_load')), ('ds_write', ('ds_write', 'ds_store')), ('ds_other', ('ds_',)),
                       ('waitcnt', ('s_waitcnt',)), ('s_nop', ('s_nop',)), ('salu', ('s_',)),
                       ('vmem', ('global_', 'buffer_', 'flat_')), ('scratch', ('scratch_',))):
            if t.startswith(pre):
                c[k] += 1
                break
        else:
            c[t] += 1
    print(name[:70], dict(c))
    ops = collections.Counter(l.strip().split(' ')[0] for l in best.splitlines()
                              if l.strip().startswith('v_') and not l.strip().startswith('v_mfma'))
    print('    ', ops.most_common(12))
