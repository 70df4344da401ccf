"""Whole-kernel instruction totals by class of each kernel in a gfx950 .s file (straight-line count, not weighted by trip count)."""
import re, collections, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)s_endpgm', txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat not in name:
        continue
    c = collections.Counter()
    for line in body.splitlines():
        t = line.strip().split(' ')[0] if line.strip() else ''
        if not t or t[0] in ';.' or t.endswith(':'):
            continue
        for k, pre in (('mfma', ('v_mfma',)), ('trans', ('v_exp', 'v_log', 'v_rcp', 'v_rsq', 'v_sqrt', 'v_sin', 'v_cos')),
                       ('accvgpr', ('v_accvgpr',)), ('cvt_pk', ('v_cvt_pk_bf16',)), ('v_pk', ('v_pk_',)), ('valu', ('v_',)),
                       ('ds_read', ('ds_read', 'ds_load')), ('ds_write', ('ds_write', 'ds_store')), ('ds_other', ('ds_',)),
                       ('waitcnt', ('s_waitcnt',)), ('s_nop', ('s_nop',)), ('salu', ('s_',)),
                       ('vmem', ('global_', 'buffer_', 'flat_')), ('scratch', ('scratch_',))):
            if t.startswith(pre):
                c[k] += 1
                break
    tot_issue = sum(v for k, v in c.items())
    print(name[:80], 'total', tot_issue, dict(c))
