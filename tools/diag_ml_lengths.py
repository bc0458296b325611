"""Diagnostic: two rounds of ML branch lengths on the white-box fixtures vs the reference, worst deviations per model."""
import sys; sys.path.insert(0,"tests"); sys.path.insert(0,".")
import numpy as np, golden_util as G
from test_gpu_ml_lengths import _setup
from veryfasttree_amd import backend
for name in ["wb_nt_f32","wb_nt_f32_gappy","wb_aa_f32"]:
    d=G.load(name); n_seqs, root = int(d["nSeqs"]), int(d["nj.root"])
    for model in (["lg"] if "_aa_" in name else ["jc","gtr"]):
        ops=_setup(d,model,8)
        gaps = int((d["leaf.codes"] == G.NOCODE).sum()) if model == "jc" else -1
        bl,ll,ev=backend.ml_lengths(ops,n_seqs,d["nj.parent"][:root+1],d["nj.child"][:root+1],root,d["nj.branchlength"][:root+1],rounds=2,n_leaf_gaps=gaps)
        w=d[model+".opt2.branchlength"][:root]; g=bl[:root]
        rel=np.abs(g-w)/np.maximum(np.abs(w),1e-3)
        idx=np.argsort(-rel)[:5]
        print(name,model,"ll",ll,[float(d["%s.opt%d.treeloglk"%(model,r)]) for r in (1,2)],"evals",ev, "exact frac",(g==w).mean())
        print("  worst", [(int(i),float(g[i]),float(w[i])) for i in idx])
        ops.close()
