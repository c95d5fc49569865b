import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from oracle import oracle as O
from vistracker_amd import ops, synthetic as syn
model = syn.smplh_model(0); rng = np.random.default_rng(0); B,N=2,100
pose = rng.normal(0, 0.2, (B, 156)).astype(np.float32); betas = rng.normal(0, 1, (B, 10)).astype(np.float32)
trans = (rng.normal(0, 0.1, (B, 3)) + [0, 0, 2.2]).astype(np.float32)
h = ops.SmplhHandle(model)
p, b_, t = (torch.tensor(x, device="cuda", requires_grad=True) for x in (pose, betas, trans))
verts, jtr, _ = ops.smplh_forward(h, p, b_, t)
dec = syn.sifnet_decoders(3); mp = syn.feature_maps(B, 4, res_scale=1 / 8)
net = ops.SifNetHandle(dec); maps = ops.FeatureMaps.from_nchw(mp)
pts = verts.detach()[:, :N].contiguous().requires_grad_(True)
cc = torch.tensor([[1018.952, 779.486]] * B, device="cuda"); bc = torch.tensor(trans, device="cuda")
df, _, parts, _, _ = ops.sifnet_query(net, maps, pts, cc, bc, head_mask=0b00101)
(df.sum() + parts.sum()).backward()
no = O.SifNet(dec, mp)
pts_np = pts.detach().cpu().numpy()
df_o, _, parts_o, _, _ = no.query(pts_np, cc.cpu().numpy(), trans)
d_o = no.query_bwd(pts_np, cc.cpu().numpy(), trans, d_df=np.ones_like(df_o), d_parts=np.ones_like(parts_o))
err = np.abs(pts.grad.cpu().numpy()-d_o).max(-1)/np.abs(d_o).max()
print("max", err.max(), "n>1e-3:", (err>1e-3).sum(), np.argwhere(err>1e-3).tolist())
i=np.argwhere(err>1e-3)
for (b,n) in i[:3]:
    print(b,n,pts_np[b,n], pts.grad.cpu().numpy()[b,n], d_o[b,n], "parts diff", np.abs(parts.detach().cpu().numpy()[b,:,n]-parts_o[b,:,n]).max())
