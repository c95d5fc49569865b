import numpy as np
a=np.load("gpurun_out/q_new.npz"); b=np.load("gpurun_out/q_old.npz")
for k in a.files:
    d=np.abs(a[k]-b[k]); s=np.abs(b[k]).max()
    print(k, "max diff/scale", d.max()/s, "scale", s)
    if d.ndim==3:
        e=d.max(-1)/s; bad=np.argwhere(e>1e-5); print("  bad points", len(bad), bad[:12].tolist())
d=np.abs(a["o_dp"]-a["d_dp"]).max(-1)/np.abs(a["d_dp"]).max(); print("new hoisted vs new direct: bad", int((d>1e-5).sum()), np.argwhere(d>1e-5)[:10].tolist())
d=np.abs(b["o_dp"]-b["d_dp"]).max(-1)/np.abs(b["d_dp"]).max(); print("old hoisted vs old direct: bad", int((d>1e-5).sum()))
