import re, sys
from collections import Counter
txt=open(sys.argv[1]).read()
pat=sys.argv[2] if len(sys.argv)>2 else r'_ZN3iak\w+'
labels=[(m.start(), m.group(1)) for m in re.finditer(r'^('+pat+r'):', txt, re.M)]
for pos,name in labels:
    end=txt.index('s_endpgm', pos)
    body=txt[pos:end]
    ins=[]
    for l in body.split('\n'):
        t=l.strip()
        if not l.startswith('\t') or not t or t.startswith(('.',';')): continue
        ins.append(t.split()[0])
    c=Counter(ins)
    v=sum(n for k,n in c.items() if k.startswith('v_'))
    print(name[:70], 'total',len(ins),'valu',v)
    print('   ', c.most_common(16))
