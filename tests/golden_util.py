"""Loading of tests/golden/*.npz (made by oracle/make_golden.py from the real reference)."""
import os

import numpy as np
import torch

from oracle import vi1_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# cfg1_shape: BASELINE.json config 1 at its real shape (batch 40, V 30 000, biLSTM 512, z 256, emb 500) -- reference-generated
# script_shape: the run scripts as written (batch 40, V 30 000, 2-layer uni-directional LSTM 500, z 500, emb 500) -- reference-generated
CASES = ["tiny_uni_l2", "tiny_bi_l1", "tiny_bi_l2", "small_fixed", "cfg1_shape", "script_shape"]
COND_CASES = ["cond_bi_l1", "cond_uni_l2"]          # --conditional prior
GREEDY_CASES = ["greedy_bi_l1", "greedy_cond_uni_l2"]   # step-wise decoding, beam size 1
BEAM_CASES = ["beam_bi_l1", "beam_cond_uni_l2", "beam_bi_l2_alpha", "beam_bi_l1_k2"]   # beam search through the reference's own translator


def load(name, dtype=torch.float32):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    vs, vt, emb, hid, zd, img, layers, brnn, B, S, T = [int(x) for x in z["cfg"][:11]]
    cond = bool(z["cfg"][11]) if len(z["cfg"]) > 11 else False
    c = O.Cfg(vs=vs, vt=vt, emb=emb, hid=hid, z=zd, img=img, layers=layers, brnn=bool(brnn), conditional=cond)
    p = {}
    for k, shp in O.param_shapes(c).items():
        if "p0_" + k in z.files:
            p[k] = torch.from_numpy(z["p0_" + k]).to(dtype)
        elif "inf_net_image.scale" in k and int(np.prod(shp)) <= O.BIG:
            p[k] = torch.zeros(*shp, dtype=dtype)       # dead branch (H6), dropped from the fixture
        else:
            p[k] = O.formula_param(k, shp, dtype)
        assert tuple(p[k].shape) == tuple(shp), k
    bt = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    for k in ("table", "eps"):
        if k in bt:                                     # decoding fixtures carry only the source side
            bt[k] = bt[k].to(dtype)
    return c, p, bt, z, (B, S, T)
