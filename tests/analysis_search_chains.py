#!/usr/bin/env python3
"""Analysis behind profiles/round3_search_kernel_analysis.txt (sections 1-3): the structure of bwt_match_gap's searches on the
on-target mix, measured with an instrumented COPY of the oracle (oracle/fq_oracle.c is copied to a scratch directory and given a log of
its chains -- an entry popped plus its exact-match children -- per read; the oracle itself is not changed).  Test-side tooling: it
builds and runs oracle code, so it lives under tests/.

    python tests/analysis_search_chains.py [pairs]      # default 20000; prints the three analyses
"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
work = tempfile.mkdtemp(prefix="fq_chains_")
log = os.path.join(work, "chains.txt")

# ---- an oracle with a chain log --------------------------------------------------------------------------------------------------
src = open(os.path.join(ROOT, "oracle", "fq_oracle.c")).read()
src = src.replace("static aln_t *match_gap(", """#include <stdio.h>
static FILE *g_chain_log;
static int g_chain_b = -1, g_chain_len = 0, g_chain_hit = 0, g_prev_match = 0;
static void chain_flush(void) { if (g_chain_b >= 0 && g_chain_log) fprintf(g_chain_log, "%d,%d,%d;", g_chain_b, g_chain_len, g_chain_hit); g_chain_b = -1; g_chain_len = 0; g_chain_hit = 0; }
static aln_t *match_gap(""", 1)
src = src.replace("    ++c->cnt.stack_pops;\n", """    ++c->cnt.stack_pops;
    { if (!g_chain_log) g_chain_log = fopen("%s", "w");
      int sc_ = e.n_mm * o->s_mm + e.n_gapo * o->s_gapo + e.n_gape * o->s_gape;
      if (!g_prev_match) { chain_flush(); g_chain_b = sc_; }
      ++g_chain_len; g_prev_match = 0; }
""" % log, 1)
src = src.replace("        if (kk <= ll) pq_push(q, o, a, i, kk, ll, e.n_mm + is_mm, e.n_gapo, e.n_gape, ST_M, is_mm, e.last_diff);",
                  "        if (kk <= ll) { pq_push(q, o, a, i, kk, ll, e.n_mm + is_mm, e.n_gapo, e.n_gape, ST_M, is_mm, e.last_diff); if (j == 4 && !is_mm) g_prev_match = 1; }", 1)
src = src.replace("      if (kk <= ll) pq_push(q, o, a, i, kk, ll, e.n_mm, e.n_gapo, e.n_gape, ST_M, 0, e.last_diff);",
                  "      if (kk <= ll) { pq_push(q, o, a, i, kk, ll, e.n_mm, e.n_gapo, e.n_gape, ST_M, 0, e.last_diff); g_prev_match = 1; }", 1)
src = src.replace("    if (hit) {\n      int do_add = 1;", "    if (hit) {\n      g_chain_hit = 1;\n      int do_add = 1;", 1)
src = src.replace("  *n_out = n_aln;\n  return aln;\n}", "  chain_flush(); g_prev_match = 0; if (g_chain_log) fprintf(g_chain_log, \"\\n\");\n  *n_out = n_aln;\n  return aln;\n}", 1)
open(os.path.join(work, "fq_oracle_x.c"), "w").write(src)
subprocess.check_call(["cp", os.path.join(ROOT, "oracle", "fq_oracle.h"), work])
subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-fPIC", "-shared", "-ffp-contract=off", "-o", os.path.join(work, "libfq_oracle_x.so"), os.path.join(work, "fq_oracle_x.c"), "-lm", "-lpthread"])
import oracle_binding as ob
ob.LIB_PATH = os.path.join(work, "libfq_oracle_x.so")
from fastquick_amd import api, synth
pre = os.path.join(work, "m10000.FASTQuick.fa")
ref = synth.make_reference(n_markers=10000, n_long=1000, seed=12345)
ref.write_fasta(pre); api.build_index(pre)
rb = synth.make_reads(ref, pairs, on_target=1.0, seed=3000)
oa = ob.OracleAligner(pre)
oa.align(rb.names, rb.seq, rb.qual, rb.lens, None, None, batch=pairs)
print("oracle counters", oa.counters())
oa.close()
ob.lib()._libc.fflush(None) if hasattr(ob.lib(), "_libc") else None

reads = []
for line in open(log):
    line = line.strip()
    reads.append([tuple(map(int, x.split(","))) for x in line.split(";") if x] if line else [])


def needs_round2(r):      # what the round without gap children cannot settle: no hit below s_gapo, or best + s_mm >= s_gapo
    for (b, l, h) in r:
        if b >= 11: return True
        if h: return not (b + 3 < 11)
    return True

print('\n== 1. pops per read, the two classes, whole-wavefront cooperation')
pops=np.array([sum(c[1] for c in r) for r in reads])
nch=np.array([len(r) for r in reads])
print("reads",len(reads),"pops total",pops.sum(),"mean",pops.mean())
srt=np.sort(pops)[::-1]
for q in (0.5,0.9,0.95,0.99,0.999): print("quantile",q,np.quantile(pops,q))
print("max",srt[:10])
# classify round-2: reads that have a chain from bucket>=11 popped or no hit in bucket<8 ... approximate: first hit bucket
r2=np.array([needs_round2(r) for r in reads])
print("round2 reads frac",r2.mean(),"their pops share",pops[r2].sum()/pops.sum(), "mean pops r2", pops[r2].mean(), "mean pops r1", pops[~r2].mean())
p2=pops[r2]
for q in (0.5,0.9,0.99,0.999): print(" r2 quantile",q,np.quantile(p2,q))
# cumulative share of pops by reads above thresholds
for th in (256,512,1024,2048,4096,8192):
    m=p2>th
    print(" r2 reads with pops>%d: %d (%.2f%% of all reads), share of r2 pops %.1f%%"%(th,m.sum(),100*m.sum()/len(reads),100*p2[m].sum()/p2.sum()))
# coop simulation for r2 reads: serial cost = pops + chains (pop trip each chain start); coop cost = per bucket drain: chunks of 64 chains, cost = max len in chunk + 1
def coop_cost(r, W=64):
    cost=0; i=0; n=len(r)
    while i<n:
        b=r[i][0]; j=i
        while j<n and r[j][0]==b: j+=1
        grp=r[i:j]
        k=0
        while k<len(grp):
            chunk=grp[k:k+W]
            # a hit in the chunk cuts the commit at the first hit lane
            cut=len(chunk)
            for t,(bb,l,h) in enumerate(chunk):
                if h: cut=t+1; break
            chunk=chunk[:cut]
            cost+=max(c[1] for c in chunk)+1
            k+=cut
        i=j
    return cost
idx=np.where(r2)[0]
ser=np.array([pops[i]+nch[i] for i in idx]); coop=np.array([coop_cost(reads[i]) for i in idx])
print("r2 serial trips total",ser.sum(),"coop rounds-steps total",coop.sum(), "ratio",ser.sum()/coop.sum())
o=np.argsort(ser)[::-1][:15]
for i in o: print("  read pops",pops[idx[i]],"chains",nch[idx[i]],"serial",ser[i],"coop",coop[i],"speedup %.1f"%(ser[i]/coop[i]))
for th in (512,1024,2048,4096):
    m=ser>th
    print(" long>%d: n=%d serial sum %d coop sum %d max serial %d max coop %d"%(th,m.sum(),ser[m].sum(),coop[m].sum(),ser[m].max() if m.any() else 0,coop[m].max() if m.any() else 0))

print('\n== 2. what predicts a long search')
R=[r for r in reads if needs_round2(r)]
tot=np.array([sum(c[1] for c in r)+len(r) for r in R])
p_lt=np.array([sum(c[1] for c in r if c[0]<11) for r in R])
c_lt=np.array([sum(1 for c in r if c[0]<11) for r in R])
hit_lt=np.array([any(c[2] and c[0]<11 for c in r) for r in R])
print("n r2",len(R),"hit<11 frac",hit_lt.mean())
print("corr tot vs pops<11",np.corrcoef(tot,p_lt)[0,1],"vs chains<11",np.corrcoef(tot,c_lt)[0,1])
for nm,m in (("hit<11",hit_lt),("nohit<11",~hit_lt)):
    print(nm,"n",m.sum(),"tot mean",tot[m].mean(),"max",tot[m].max(),"q99",np.quantile(tot[m],0.99))
# ranking quality: if sorted by predictor desc, where do the top-1% longest land?
for nm,pred in (("pops<11",p_lt),("chains<11",c_lt),("pops<11 if nohit else 0", np.where(hit_lt,0,p_lt))):
    order=np.argsort(-pred,kind='stable')
    rank=np.empty(len(R),int); rank[order]=np.arange(len(R))
    top=np.argsort(-tot)[:max(1,len(R)//50)]
    print(nm,"top2% longest: mean rank frac",rank[top].mean()/len(R),"max rank frac",rank[top].max()/len(R))
# bins
bins=[0,50,100,150,200,300,400,600,1000,5000]
for lo,hi in zip(bins[:-1],bins[1:]):
    m=(p_lt>=lo)&(p_lt<hi)
    if m.any(): print("pops<11 in [%d,%d): n=%d tot mean %.0f max %d min %d"%(lo,hi,m.sum(),tot[m].mean(),tot[m].max(),tot[m].min()))

print('\n== 3. cooperative search by W lanes')
R=[r for r in reads if needs_round2(r)]
def coop_cost(r, W):
    cost=0; lanes_used=0; i=0; n=len(r)
    while i<n:
        b=r[i][0]; j=i
        while j<n and r[j][0]==b: j+=1
        grp=r[i:j]; k=0
        while k<len(grp):
            chunk=grp[k:k+W]; cut=len(chunk)
            for t,(bb,l,h) in enumerate(chunk):
                if h: cut=t+1; break
            chunk=chunk[:cut]
            cost+=max(c[1] for c in chunk)+1
            k+=cut
        i=j
    return cost
ser=np.array([sum(c[1] for c in r)+len(r) for r in R])
for W in (1,2,4,8,16,32,64):
    cc=np.array([coop_cost(r,W) for r in R])
    long=ser>2048
    print("W=%2d: all r2: steps %8d lane-steps %9d (eff %.2f) | reads>2048 serial: max steps %5d, sum steps %8d eff %.2f | max overall %d"%(W,cc.sum(),cc.sum()*W,ser.sum()/(cc.sum()*W),cc[long].max(),cc[long].sum(),ser[long].sum()/(cc[long].sum()*W),cc.max()))
