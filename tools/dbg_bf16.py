import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vi1_oracle as O
from tests.golden_util import load
from variational_mmt_amd.engine import Dims, Engine
name = sys.argv[1] if len(sys.argv) > 1 else "small_fixed"
c, p, bt, z, (B, S, T) = load(name)
for dtype in ("f32", "bf16"):
    e = Engine(Dims(c.vs, c.vt, c.emb, c.hid, c.z, c.img, c.layers, c.brnn, 0.0), dtype=dtype)
    e.load_state_dict(p); e.set_image_table(bt["table"])
    ws = e.forward(bt["src"], bt["src_len"], bt["tgt"], bt["indices"], training=True, eps=bt["eps"])
    e.loss_backward(ws, normalization=B); torch.cuda.synchronize()
    img = bt["table"][bt["indices"]]
    r, Lo, g = O.step_grads(p, c, bt["src"], bt["src_len"], bt["tgt"], img, bt["eps"])
    st = e.read_stats(ws)
    print(dtype, "elbo", st["elbo"], float(Lo["elbo"]), "kl", st["td_kl_before"], float(Lo["kl_before"]))
    for k in g:
        got, want = e.grads[k].cpu().double(), g[k].double()
        print("  %-55s max|g| %.3e  maxerr/max %.3e  relL2 %.3e" % (k, want.abs().max(), (got-want).abs().max()/want.abs().max(), (got-want).norm()/want.norm()))
