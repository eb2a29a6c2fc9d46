import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from doda_amd import ops, spconv
from oracle import oracle as orc
from tests.util import surface_voxels
dev = torch.device("cuda:0")
shape=[25,20,23]; batch=2
idx = surface_voxels(48, 1500, batch, shape)
oi, pairs, pn, oshape = orc.indice_pairs_conv(idx, batch, shape, 2,2,0,1)
data = spconv.ops.build_down2(torch.from_numpy(idx).to(dev), batch, shape, 2,2,0,1)
rng=np.random.default_rng(0)
for (cin,cout) in [(16,32),(32,48)]:
    w=(rng.standard_normal((2,2,2,cin,cout))*0.3).astype(np.float32)
    x=rng.standard_normal((idx.shape[0],cin)).astype(np.float32)
    gmid=rng.standard_normal((oi.shape[0],cout)).astype(np.float32)
    ref_dx, ref_dw = orc.indice_conv_backward(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(gmid).double(), pairs, pn, False, False)
    wt=torch.from_numpy(w).to(dev).reshape(8,cin,cout)
    for rep in range(3):
        dx = ops.spconv_gather(torch.from_numpy(gmid).to(dev), wt, data.tbl_rev, idx.shape[0], 1, cin)
        torch.cuda.synchronize()
        e=(dx.cpu().double()-ref_dx).abs().max()/ref_dx.abs().max()
        dw = ops.spconv_wgrad(torch.from_numpy(x).to(dev), torch.from_numpy(gmid).to(dev), data.tbl, oi.shape[0])
        torch.cuda.synchronize()
        e2=(dw.cpu().double().reshape(ref_dw.shape)-ref_dw).abs().max()/ref_dw.abs().max()
        dx2 = ops.spconv_gather(torch.from_numpy(gmid).to(dev), wt, data.tbl_rev, idx.shape[0], 1, cin)
        e3=(dx2.cpu().double()-ref_dx).abs().max()/ref_dx.abs().max()
        print(cin,cout,"dgrad err %.2e  wgrad err %.2e  dgrad-again %.2e" % (e,e2,e3), "rows with err:", int(((dx.cpu().double()-ref_dx).abs().max(1).values>1e-3).sum()))
