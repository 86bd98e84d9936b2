"""Run-to-run spread of 24 tiny training steps (the shapes of tests/test_gpu_row_adam.py): five dense engines against a first one, conditional and
fixed prior, f32 and bf16 -- the largest parameter difference and where it sits.   python tests/run_to_run_matrix.py   (not a test: a measurement the tolerance of tests/test_gpu_row_adam.py rests on)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vi1_oracle as O
from variational_mmt_amd.engine import Dims, Engine
d_ = lambda x, y: (x - y).abs().max().item()
for cond in (True, False):
    c = O.Cfg(vs=97, vt=89, emb=24, hid=32, z=8, layers=1, brnn=True, conditional=cond)
    p = O.init_params(c, seed=2)
    def run(rows, dtype, **kw):
        e = Engine(Dims(c.vs, c.vt, c.emb, c.hid, c.z, c.img, c.layers, c.brnn, 0.0, conditional=cond), dtype=dtype, device="cuda", seed=1)
        e.row_adam = rows; e.lazy_roll = 5
        e.load_state_dict(p)
        for step in range(24):
            bt = O.synth_batch(c, 6, 5 + step % 3, 6 + step % 2, n_img=12, seed=50 + step, fixed_len=False)
            e.set_image_table(bt["table"])
            ws = e.forward(bt["src"], bt["src_len"], bt["tgt"], bt["indices"], training=True, eps=bt["eps"], **(dict(tgt_len=bt["tgt_len"]) if cond else {}))
            e.loss_backward(ws, normalization=6)
            e.optim_step(lr=0.01 if step < 6 else 0.004, max_grad_norm=5.0 if step % 4 else 0.5)
        torch.cuda.synchronize()
        return e.flat_p[:e.n_opt].clone(), e
    for dtype in ("f32", "bf16"):
        b, eb = run(False, dtype)
        out = []
        for r in range(5):
            x, e = run(False, dtype)
            w = sorted(((float((e.params[n] - eb.params[n]).abs().max()), n) for n in e.params if e.offsets[n][0] < e.n_opt), reverse=True)[0]
            out.append("%.1e %s" % (d_(x, b), w[1][-28:]))
        print("cond" if cond else "fixed", dtype, "|", " | ".join(out))
