import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from oracle import oracle as O
from vistracker_amd import ops, synthetic as syn
model = syn.smplh_model(0); rng = np.random.default_rng(0); B,N=2,100
pose = rng.normal(0, 0.2, (B, 156)).astype(np.float32); betas = rng.normal(0, 1, (B, 10)).astype(np.float32)
trans = (rng.normal(0, 0.1, (B, 3)) + [0, 0, 2.2]).astype(np.float32)
mo = O.SmplModel(model); v_o,_,_ = mo.forward(pose,betas,trans)
dec = syn.sifnet_decoders(3); mp = syn.feature_maps(B, 4, res_scale=1 / 8)
net = ops.SifNetHandle(dec); maps = ops.FeatureMaps.from_nchw(mp)
pts_np = v_o[:, :N].copy()
pts = torch.tensor(pts_np, device="cuda", requires_grad=True)
cc = torch.tensor([[1018.952, 779.486]] * B, device="cuda"); bc = torch.tensor(trans, device="cuda")
no = O.SifNet(dec, mp)
for mask,name in ((1,"df"),(4,"parts"),(5,"both")):
    pts.grad=None
    outs = ops.sifnet_query(net, maps, pts, cc, bc, head_mask=mask)
    loss = sum(o.sum() for o in outs if o.numel()>0); loss.backward()
    df_o, _, parts_o, _, _ = no.query(pts_np, cc.cpu().numpy(), trans)
    d_o = no.query_bwd(pts_np, cc.cpu().numpy(), trans, d_df=np.ones_like(df_o) if mask&1 else None, d_parts=np.ones_like(parts_o) if mask&4 else None)
    err = np.abs(pts.grad.cpu().numpy()-d_o).max(-1)/np.abs(d_o).max()
    print(name, "max", err.max(), "n>1e-3:", (err>1e-3).sum(), "median", np.median(err), np.argwhere(err>1e-3).tolist())
