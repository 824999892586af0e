"""Localise a wrong att_rows result: Wo = I, bo = 0, no residual -> y = LN(concat_h O_h); error per head / per 32-row block."""
import ctypes, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cotr_amd import _lib
from tests import gpu_helpers as G

lib = _lib.load_library()
d = G.dev()
g = torch.Generator().manual_seed(0)
for qp in (False, True):
    for nb, nq in ((1, 64), (2, 128)):
        R = nb * nq
        kvw = torch.randn(nb * 512, 768, generator=g)
        k, v = kvw[:, 256:512], kvw[:, 512:768]
        wo, bo = torch.eye(256), torch.zeros(256)
        lw, lb = torch.ones(256), torch.zeros(256)
        x2 = torch.randn(R, 256, generator=g)
        wq, bq = torch.randn(256, 256, generator=g) / 16, torch.randn(256, generator=g) * 0.1
        qw = torch.randn(R, 768, generator=g) * 0.5
        q = (F.linear(x2.double(), wq.double(), bq.double()) * 32 ** -0.5).float() if qp else qw[:, :256]
        qh = q.double().view(nb, nq, 8, 32).permute(0, 2, 1, 3)
        kh = k.double().view(nb, 512, 8, 32).permute(0, 2, 1, 3)
        vh = v.double().view(nb, 512, 8, 32).permute(0, 2, 1, 3)
        o = (torch.softmax(qh @ kh.transpose(-1, -2), dim=-1) @ vh).permute(0, 2, 1, 3).reshape(R, 256)
        ref = F.layer_norm(o, (256,))
        kvd = kvw.to(d)
        kp, vp = ctypes.c_void_p(kvd.data_ptr() + 1024), ctypes.c_void_p(kvd.data_ptr() + 2048)
        t = [wo.to(d), bo.to(d), lw.to(d), lb.to(d), x2.to(d), wq.to(d), bq.to(d), qw.to(d)]
        y = torch.empty(R, 256, device=d)
        if qp:
            rc = lib.cotr_op_att_rows(None, 0, None, G.P(t[4]), G.P(t[5]), G.P(t[6]), 32 ** -0.5, kp, vp, 768, G.P(t[0]), G.P(t[1]), None, G.P(t[2]), G.P(t[3]), G.P(y), nb, nq, G.sptr())
        else:
            rc = lib.cotr_op_att_rows(G.P(t[7]), 768, None, None, None, None, 0.0, kp, vp, 768, G.P(t[0]), G.P(t[1]), None, G.P(t[2]), G.P(t[3]), G.P(y), nb, nq, G.sptr())
        err = (y.cpu().double() - ref).abs()
        print(f'qp={qp} nb={nb} nq={nq} rc={rc} max err {float(err.max()):.3e}')
        eh = err.view(R // 32, 32, 8, 32).amax(dim=(1, 3))
        print('  max error per [32-row block][head]:')
        for rb in range(R // 32):
            print('   ', ' '.join(f'{float(e):8.1e}' for e in eh[rb]))
