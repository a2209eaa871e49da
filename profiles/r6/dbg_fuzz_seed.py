import os, sys
import numpy as np
ROOT="/root/repo"
sys.path[:0]=[ROOT, ROOT+"/oracle", ROOT+"/tests"]
import cc_amd, oracle_py as oracle
import torch
cc=cc_amd.load(); L=oracle.L
seed=int(sys.argv[1]) if len(sys.argv)>1 else 131409; Q=int(sys.argv[2]) if len(sys.argv)>2 else 70
rng=np.random.default_rng(seed)
d=L.default_db_cfg()
d.min_elapse=float(rng.uniform(0.8,2.0)); d.max_elapse=d.min_elapse+float(rng.uniform(0.5,1.5))
d.nnk=int(rng.choice([10,30,50,64])); d.max_fine_opt=int(rng.choice([2,5,10]))
qlv=[(1,2,3),(2,3),(2,3,4),(1,2,3)][int(rng.integers(4))]
d.n_q_levels=len(qlv)
for i,v in enumerate(qlv): d.q_levels[i]=v
lb,ub=L.default_thresholds()
if rng.random()<0.5:
    lb.i_ovlp_sum,lb.i_ovlp_max_one,lb.i_in_ang_rng,lb.i_indiv_sim,lb.i_orie_sim=[int(v) for v in rng.integers(2,5,5)]
    lb.correlation=float(rng.uniform(0.1,0.5))
kind=int(rng.integers(3))
world=cc.synth.World(loop_len=float(rng.uniform(24,36)),dense=(kind==1),seed=int(rng.integers(1<<20))) if kind<2 else cc.synth.World(kitti=True,seed=int(rng.integers(1<<20)),block=float(rng.uniform(36,50)),tile=300.0)
n=int(rng.integers(56,84)); full=seed%3==0
x,poses,ts=cc.synth.make_sequence(n,world=world,device="cuda",step=(1.0 if kind<2 else 3.0),**({} if full else dict(beams=16,azim=450)))
P=x.shape[1]; offs=np.arange(n+1,dtype=np.int64)*P
seeds=rng.choice(1<<20,n,replace=False).astype(np.int32)
os.environ["CC_KNN_MODE"]="2" if seed%2 else "0"
print("kind",kind,"n",n,"full",full,"nnk",d.nnk,"qlv",qlv,"lb",lb.i_ovlp_sum,lb.i_ovlp_max_one,lb.i_in_ang_rng,lb.i_indiv_sim,lb.i_orie_sim,lb.correlation)
ctx=cc.Context(0,None,max_batch=512)
desc=ctx.ingest(x.reshape(-1,4),offs)
db=cc.Database(ctx,cfg=d,capacity=n)
db.add_scans(desc,ts,seeds)
res,knn,cnt=db.query(desc,np.arange(n,dtype=np.int32),lb=lb,ub=ub,want_knn=True,allow_flagged=True)
dn=cc.desc_to_numpy(desc)
ores,_,odesc=oracle.run_sequence(x.cpu().numpy().reshape(-1,4),offs,ts,seeds,dcfg=d,lb=lb,ub=ub,want_desc=True)
for f in ("n_knn_hits","cand_aft_check1","cand_aft_check2","cand_aft_check3","n_cand_pose","n_cand_tidy","n_res"):
    print(f,"gpu",res[f][Q],"oracle",ores[f][Q])
# hints from the gpu's knn hits of query Q
hits=[]
NQ,NP=knn.shape[1],knn.shape[2]
for ll in range(NQ):
    for sq in range(NP):
        for j in range(cnt[Q,ll,sq]):
            h=knn[Q,ll,sq,j]
            hits.append((int(h["gidx"]),int(h["level"]) if "level" in h.dtype.names else 0,int(h["seq"]),sq,ll))
print("hits",len(hits), knn.dtype)
hints=np.zeros(len(hits),L.hint_dt)
for k,(g,lv,ss,st,ll) in enumerate(hits):
    hints[k]=(g,lv,ss,st,0)
gr,gsc=db.check_hints(desc[Q],hints,lb=lb,ub=ub,max_fine_opt=d.max_fine_opt)
gidxs=sorted(set(h[0] for h in hits))
cm={g:i for i,g in enumerate(gidxs)}
tgt=oracle.Scan.from_desc(odesc[Q],int_id=Q)
cands=[oracle.Scan.from_desc(odesc[g],int_id=g) for g in gidxs]
oh=np.array([(cm[g],lv,ss,st) for (g,lv,ss,st,ll) in hits],np.int32)
orr,osc=oracle.check_hints(tgt,cands,oh,sim=d.cont_sim,lb=lb,ub=ub,max_fine_opt=d.max_fine_opt)
names=("i_ovlp_sum","i_ovlp_max_one","i_in_ang_rng","i_indiv_sim","i_orie_sim","passed")
G=np.stack([gsc[f] for f in names],1)
print("gpu passes",int(G[:,5].sum()),"oracle passes",int(osc[:,5].sum()))
diff=np.where((G!=osc).any(1))[0]
print("hints differing:",len(diff))
for k in diff[:10]:
    print(k,hits[k],"gpu",G[k],"oracle",osc[k])

print("result gpu n_cand_pose/tidy", gr["n_cand_pose"], gr["n_cand_tidy"], "oracle", orr["n_cand_pose"], orr["n_cand_tidy"], "corr", gr["correlation"], orr["correlation"], "cand", gr["cand_gidx"], gidxs[int(orr["cand_gidx"])] if orr["n_res"] else -1)
if gr["n_cand_tidy"] != orr["n_cand_tidy"]:
    # which candidate scan: drop one candidate's hints at a time
    for g in gidxs:
        keep=[k for k,h in enumerate(hits) if h[0]!=g]
        g1,_=db.check_hints(desc[Q],hints[keep],lb=lb,ub=ub,max_fine_opt=d.max_fine_opt)
        o1,_=oracle.check_hints(tgt,cands,oh[keep],sim=d.cont_sim,lb=lb,ub=ub,max_fine_opt=d.max_fine_opt)
        if g1["n_cand_tidy"]==o1["n_cand_tidy"]:
            print("without candidate scan",g,"both agree:",g1["n_cand_tidy"])
            only=[k for k,h in enumerate(hits) if h[0]==g]
            g2,s2=db.check_hints(desc[Q],hints[only],lb=lb,ub=ub,max_fine_opt=d.max_fine_opt)
            o2,so2=oracle.check_hints(tgt,cands,oh[only],sim=d.cont_sim,lb=lb,ub=ub,max_fine_opt=d.max_fine_opt)
            print(" alone: gpu tidy",g2["n_cand_tidy"],"pose",g2["n_cand_pose"],"corr",g2["correlation"],"tf",g2["tf"]," oracle tidy",o2["n_cand_tidy"],"pose",o2["n_cand_pose"],"corr",o2["correlation"],"tf",o2["tf"], "lb.corr", lb.correlation)
            dp=db.debug_passes()
            print(" gpu passes of this candidate:", [(int(p["hint"]), int(p["n_pairs"]), p["tf"].tolist()) for p in dp])
if len(sys.argv) > 3:   # subsets of one candidate's hints: python dbg_fuzz_seed.py <seed> <scan> <candidate scan>
    g = int(sys.argv[3])
    only = [k for k, h in enumerate(hits) if h[0] == g]
    g2, s2 = db.check_hints(desc[Q], hints[only], lb=lb, ub=ub, max_fine_opt=d.max_fine_opt)
    ps = [k for k in range(len(only)) if s2["passed"][k]]
    print("passing hints of the candidate:", ps, [hits[only[k]] for k in ps])
    import itertools
    for r in range(1, len(ps) + 1):
        for sub in itertools.combinations(ps, r):
            idx = [only[k] for k in sub]
            ga, _ = db.check_hints(desc[Q], hints[idx], lb=lb, ub=ub, max_fine_opt=d.max_fine_opt)
            dp = db.debug_passes()
            oa, _ = oracle.check_hints(tgt, cands, oh[idx], sim=d.cont_sim, lb=lb, ub=ub, max_fine_opt=d.max_fine_opt)
            print(sub, "gpu tidy", ga["n_cand_tidy"], "corr %.6f" % ga["correlation"], "| oracle tidy", oa["n_cand_tidy"], "corr %.6f" % oa["correlation"], "| gpu T:", [[round(v, 4) for v in p["tf"]] + [int(p["n_pairs"])] for p in dp])
if len(sys.argv) > 4:   # the constellation of one passing hint of that candidate, its umeyama on the host: ... <candidate scan> <k-th hint of the candidate>
    g = int(sys.argv[3]); kk = int(sys.argv[4])
    only = [k for k, h in enumerate(hits) if h[0] == g]
    ga, sa = db.check_hints(desc[Q], hints[[only[kk]]], lb=lb, ub=ub, max_fine_opt=d.max_fine_opt)
    dp = db.debug_passes()
    print("scores", sa, "passes", len(dp))
    p = dp[0]
    pairs = []
    for w in range(7):
        m = int(p["pairs"][w])
        while m:
            b = (m & -m).bit_length() - 1
            m &= m - 1
            bit = w * 64 + b
            pairs.append((bit // 100 + 1, (bit % 100) // 10, bit % 10))
    print("pairs (level, seq_src, seq_tgt):", pairs, "n_pairs", int(p["n_pairs"]))
    S = np.array([dn[g]["cont"][l][s]["pos_mean"] for (l, s, t) in pairs], np.float64)
    T = np.array([dn[Q]["cont"][l][t]["pos_mean"] for (l, s, t) in pairs], np.float64)
    sm, dm = S.mean(0), T.mean(0)
    A, B = S - sm, T - dm
    s00, s01, s10, s11 = (B[:, 0] * A[:, 0]).mean(), (B[:, 0] * A[:, 1]).mean(), (B[:, 1] * A[:, 0]).mean(), (B[:, 1] * A[:, 1]).mean()
    ang = np.arctan2(s10 - s01, s00 + s11)
    R = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
    print("host umeyama: angle", ang, "t", dm - R @ sm, " gpu record tf", p["tf"])
    print("src pts", S.tolist()); print("tgt pts", T.tolist())
