"""Compares two dumps of tools/run_gardner.py (argv[4] there: an .npz of the first 64 rows) -- e.g. the Gardner mode in pieces
against whole chunks (GSEG=0): symbol counts, differing hard decisions, largest soft-symbol difference per row.
usage: cmp_gardner.py a.npz b.npz"""
import numpy as np, sys
a=np.load(sys.argv[1]); b=np.load(sys.argv[2])
for r in range(0,64,7):
    na,nb=int(a['n_soft'][r]),int(b['n_soft'][r])
    m=min(na,nb)-1
    ha,hb=a['hard'][r,:m],b['hard'][r,:m]
    diff=np.where(ha!=hb)[0]
    sa,sb=a['soft'][r,:m+1],b['soft'][r,:m+1]
    err=np.abs(sa-sb)/np.max(np.abs(sa))
    print(r,'n',na,nb,'hard diff',len(diff),diff[:6],'soft max rel diff',float(err.max()),'at',int(err.argmax()), 'after seam mean', float(err[4300:4400].mean()) if m>4400 else None)
