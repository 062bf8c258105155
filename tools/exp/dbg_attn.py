import os, sys, torch
sys.argv=['x','6,9']
__file__ = os.path.abspath('tools/attn_ab.py'); exec(open('tools/attn_ab.py').read().split("# ---- speed")[0])
for v in (6,9):
    out = torch.zeros(N, heads*64, dtype=BF, device="cuda"); run(v,Q,K,V,N,npad,heads,out); torch.cuda.synchronize()
    err=(out.float()-ref).abs(); rows=err.max(dim=1).values
    top=torch.topk(rows,5)
    print(v, [(int(i), round(float(x),4)) for x,i in zip(top.values, top.indices)], 'mean row err', float(rows.mean()))
    # error split: rows 500:520 vs others
    print('  rows500-520 max', float(rows[500:520].max()), 'others max', float(torch.cat([rows[:500],rows[520:]]).max()))
