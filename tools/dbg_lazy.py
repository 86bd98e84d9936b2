import sys, torch
sys.path.insert(0, ".")
from oracle import vi1_oracle as O
from variational_mmt_amd.engine import Dims, Engine
c = O.Cfg(vs=30000, vt=30000, emb=500, hid=512, z=256, img=2048, layers=1, brnn=True)
p = O.init_params(c, seed=0)
Bg = 256
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
bts = [O.synth_batch(c, Bg, 20, 21, n_img=1000, seed=7 + i, fixed_len=False) for i in range(NS)]
def run(rows, dtype="f32"):
    e = Engine(Dims(c.vs, c.vt, c.emb, c.hid, c.z, c.img, c.layers, c.brnn, 0.0), dtype=dtype, device="cuda:0")
    e.persistent_lstm = False
    e.row_adam = rows
    e.load_state_dict(p)
    e.set_image_table(bts[0]["table"])
    gs = []
    for bt in bts:
        ws = e.forward(bt["src"], bt["src_len"], bt["tgt"], bt["indices"], training=True, eps=bt["eps"])
        e.loss_backward(ws, normalization=Bg)
        if "sync" in sys.argv:
            torch.cuda.synchronize()
            gs.append(e.flat_g.clone())
        e.optim_step(lr=0.002, max_grad_norm=0.5)
    torch.cuda.synchronize()
    return e, gs
(a, ga), (b, gb) = run(True), run(False)
(a2, _), (b2, _) = run(True), run(False)
nn = a.n_opt
print("lazy vs lazy m_rel %.3e   dense vs dense %.3e   lazy vs dense %.3e" % tuple(float((x.flat_m[:nn] - y.flat_m[:nn]).norm() / y.flat_m[:nn].norm()) for x, y in ((a2, a), (b2, b), (a, b))))
n = a.n_opt
name = "encoder.embeddings.make_embedding.emb_luts.0.weight"
o, shp = a.offsets[name]
k = shp[0] * shp[1]
touched = [torch.zeros(shp[0], dtype=torch.bool) for _ in bts]
for i, bt in enumerate(bts):
    touched[i][bt["src"].reshape(-1)] = True
for i in range(NS if "sync" in sys.argv else 0):
    gl, gd = ga[i][o:o + k].view(*shp).cpu(), gb[i][o:o + k].view(*shp).cpu()
    t = touched[i]
    print("step", i + 1, "gradient rows: |lazy - dense| over touched rows", float((gl[t] - gd[t]).norm()), "of", float(gd[t].norm()),
          "; dense rows outside the batch nonzero:", int((gd[~t] != 0).any(dim=1).sum()), "; lazy stale rows outside:", int((gl[~t] != 0).any(dim=1).sum()))
ml, md = a.flat_m[o:o + k].view(*shp).cpu(), b.flat_m[o:o + k].view(*shp).cpu()
dr = (ml - md).norm(dim=1)
print("rows with |dm| > 1e-9:", int((dr > 1e-9).sum()), "of", shp[0])
if NS == 2:
    for tag, sel in (("both", touched[0] & touched[1]), ("step1 only", touched[0] & ~touched[1]), ("step2 only", ~touched[0] & touched[1]), ("never", ~touched[0] & ~touched[1])):
        print("  %-11s rows %5d  |dm| %.3e  |m| %.3e   rows off: %d" % (tag, int(sel.sum()), float(dr[sel].norm()), float(md[sel].norm()), int((dr[sel] > 1e-9).sum())))
bad = torch.nonzero(dr > 1e-9).reshape(-1)[:8].tolist()
print("first bad rows", bad, "pad row 1 touched:", [bool(t[1]) for t in touched])
for r in bad[:4]:
    print("   row", r, "touched", [bool(t[r]) for t in touched], "m lazy", ml[r, :3].tolist(), "dense", md[r, :3].tolist(), "last", int(a.row_tables[0]["last"][r]))
