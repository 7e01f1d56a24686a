import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from oracle import csa_oracle as orc
from csn_amd import _lib
from csn_amd.minkowski_attention import MultiHeadAttention
rng = np.random.default_rng(29)
H, C = 4, 256
d = C // H
p = orc.make_params(rng, H, d_model=C, d_k=d, d_v=d)
lens = [(7, 301), (45, 70), (1301, 37), (512, 500), (100, 1)]
qs = [torch.from_numpy(rng.standard_normal((n, C)).astype(np.float32)) for n, _ in lens]
ks = [torch.from_numpy(rng.standard_normal((m, C)).astype(np.float32)) for _, m in lens]
vs = [torch.from_numpy(rng.standard_normal((m, C)).astype(np.float32)) for _, m in lens]
gs = [torch.from_numpy(rng.standard_normal((n, C)).astype(np.float32)) for n, _ in lens]
for mode in (0, 1):
    _lib.check(_lib.lib().csn_set_math_mode(mode))
    res = []
    for varlen in (True, False):
        m = MultiHeadAttention(H, C, d, d)
        m.load_state_dict({k[len("attention."):]: v for k, v in p.items() if k.startswith("attention.")}, strict=False)
        m = m.cuda().eval()
        qd, kd, vd = ([t.cuda().requires_grad_(True) for t in ts] for ts in (qs, ks, vs))
        outs = m.forward_varlen(qd, kd, vd) if varlen else [m(q[None], k[None], v[None])[0][0] for q, k, v in zip(qd, kd, vd)]
        sum((o * g.cuda()).sum() for o, g in zip(outs, gs)).backward()
        res.append(([[t.grad.cpu() for t in ts] for ts in (qd, kd, vd)], {n: q.grad.cpu() for n, q in m.named_parameters()}))
    (g1, w1), (g0, w0) = res
    for i in range(len(lens)):
        print(mode, lens[i], " ".join("%s %.1e/%.1e" % (nm, (g1[j][i] - g0[j][i]).abs().max().item(), g0[j][i].abs().max().item()) for j, nm in enumerate(("dq", "dk", "dv"))))
    print(mode, "weights", " ".join("%s %.1e" % (n, ((w1[n] - w0[n]).abs().max() / w0[n].abs().max()).item()) for n in w0))
