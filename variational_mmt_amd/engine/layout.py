"""Sizes, padding rules and the 2-D device buffer of the VI_Model1 engine (see the package docstring)."""
import collections
import ctypes as C
import math
from os import environ as _os_env

import torch

from .. import _lib as L

PAD = 1  # '<blank>' (onmt/io/DatasetBase.py:7-11)


def _ru(x, m):
    return (x + m - 1) // m * m


class Dims(object):
    def __init__(self, vs, vt, emb=500, hid=500, z=500, img=2048, layers=2, brnn=False, dropout=0.0, conditional=False):
        self.vs, self.vt, self.emb, self.hid, self.z, self.img = vs, vt, emb, hid, z, img
        self.layers, self.brnn, self.dropout = layers, bool(brnn), float(dropout)
        self.conditional = bool(conditional)       # --conditional prior (ModelConstructor.py:435-460)
        self.ht = hid // 2                          # encoder_tgt is always bidirectional (ModelConstructor.py:456-457)
        self.qin = 2 * hid + img if conditional else hid
        assert not conditional or hid % 2 == 0
        self.dirs = 2 if brnn else 1
        assert hid % self.dirs == 0
        self.hd = hid // self.dirs
        assert hid <= 1024, "attention kernel limit (H <= 1024)"
        # COMPUTE layout of the hidden size.  The run scripts train -rnn_size 500 --z_latent_dim 500, 2-layer uni-directional
        # (run_translated_m30k_only.sh:46-57, opts.py:14-16,54,67-69); the MFMA LSTM / attention kernels tile H in 32s and the persistent
        # recurrences serve H in {64, 128, 256, 512}.  So hidden vectors are computed `hp` wide (500 -> 512, gate g of a 4H vector at
        # g * hp) with zeros in the padding: shadows are packed gate block by gate block, pre-activations / h / c / every gradient
        # are exactly zero in padded lanes (sigmoid(0) * tanh(0)), and gradients are stored back through the block map of
        # vmmt_gemm_args.c_row_blk.  The arena, the state dict, checkpoints, Adam and the all-reduce keep the reference shapes.
        # (a bidirectional ENCODER with an odd per-direction size keeps the general kernels.)  The conditional model's encoder_tgt is
        # always bidirectional with hid / 2 units per direction (250 -> 256): its output is laid out [fwd | pad | bwd | pad], 2 * htp
        # wide, and feeds the posterior network's input [h_x : hp | h_y : 2 htp | v : img] (qin_p columns).
        self.pad = (not self.brnn) and hid % 32 != 0
        self.hp = _ru(hid, 32) if self.pad else hid
        self.hdp = self.hp // self.dirs
        self.htp = _ru(self.ht, 32) if self.pad else self.ht
        self.qin_p = (self.hp + 2 * self.htp + img) if self.conditional else self.hp
        self.zp = _ru(z, 128)                       # tiled latent size of the fused q(z|x) kernel (Z_valid = z)

    def param_shapes(self):
        """name -> shape, in ARENA order (reverse of backward completion is not needed: order == completion)."""
        d = self
        s = []
        s += [("generator.0.weight", (d.vt, d.hid)), ("generator.0.bias", (d.vt,))]
        s += [("decoder.attn.linear_out.weight", (d.hid, 2 * d.hid)), ("decoder.attn.linear_in.weight", (d.hid, d.hid))]
        for l in reversed(range(d.layers)):
            i = d.emb + d.z if l == 0 else d.hid
            s += [("decoder.rnn.weight_ih_l%d" % l, (4 * d.hid, i)), ("decoder.rnn.weight_hh_l%d" % l, (4 * d.hid, d.hid)),
                  ("decoder.rnn.bias_ih_l%d" % l, (4 * d.hid,)), ("decoder.rnn.bias_hh_l%d" % l, (4 * d.hid,))]
        s += [("decoder.embeddings.make_embedding.emb_luts.0.weight", (d.vt, d.emb))]
        for l in reversed(range(d.layers)):
            i = d.emb if l == 0 else d.hid
            for suf in ([""] + (["_reverse"] if d.brnn else [])):
                s += [("encoder.rnn.weight_ih_l%d%s" % (l, suf), (4 * d.hd, i)),
                      ("encoder.rnn.weight_hh_l%d%s" % (l, suf), (4 * d.hd, d.hd)),
                      ("encoder.rnn.bias_ih_l%d%s" % (l, suf), (4 * d.hd,)),
                      ("encoder.rnn.bias_hh_l%d%s" % (l, suf), (4 * d.hd,))]
        s += [("encoder.embeddings.make_embedding.emb_luts.0.weight", (d.vs, d.emb))]
        if d.conditional:
            for br in ("location", "scale"):       # p(z|x)
                s += [("gen_net_global.%s.fc2.weight" % br, (d.z, d.z)), ("gen_net_global.%s.fc2.bias" % br, (d.z,)),
                      ("gen_net_global.%s.fc1.weight" % br, (d.z, d.hid)), ("gen_net_global.%s.fc1.bias" % br, (d.z,))]
            for l in reversed(range(d.layers)):    # encoder_tgt (shares the decoder's embedding table)
                i = d.emb if l == 0 else d.hid
                for suf in ("", "_reverse"):
                    s += [("encoder_tgt.rnn.weight_ih_l%d%s" % (l, suf), (4 * d.ht, i)),
                          ("encoder_tgt.rnn.weight_hh_l%d%s" % (l, suf), (4 * d.ht, d.ht)),
                          ("encoder_tgt.rnn.bias_ih_l%d%s" % (l, suf), (4 * d.ht,)),
                          ("encoder_tgt.rnn.bias_hh_l%d%s" % (l, suf), (4 * d.ht,))]
        s += [("inf_net_image.location.fc2.weight", (d.img, d.img)), ("inf_net_image.location.fc2.bias", (d.img,)),
              ("inf_net_image.location.fc1.weight", (d.img, d.z)), ("inf_net_image.location.fc1.bias", (d.img,)),
              ("inf_net_image.gate_affine_transform.weight", (1, d.z)), ("inf_net_image.gate_affine_transform.bias", (1,))]
        for br in ("location", "scale"):
            s += [("inf_net_global.%s.fc2.weight" % br, (d.z, d.z)), ("inf_net_global.%s.fc2.bias" % br, (d.z,)),
                  ("inf_net_global.%s.fc1.weight" % br, (d.z, d.qin)), ("inf_net_global.%s.fc1.bias" % br, (d.z,))]
        nograd = [("inf_net_image.scale.fc1.weight", (d.img, d.z)), ("inf_net_image.scale.fc1.bias", (d.img,)),
                  ("inf_net_image.scale.fc2.weight", (d.img, d.img)), ("inf_net_image.scale.fc2.bias", (d.img,))]
        return s, nograd


KPAD = 64          # GEMM reduction slab (elements)
SEG_ALIGN = 512    # arena segments start on multiples of this many elements (8 ranks x 64-element units)


class Buf(object):
    """2-D device buffer [rows][ld].  Rows and (unless `ld` is given) columns are zero-padded to whole 64-element GEMM
    slabs plus one spare slab of rows, so that a GEMM may round its reduction length K up to a multiple of 64 whichever
    way the buffer is traversed (K-contiguous or K-strided, also from a row / column offset): the padding contributes
    exact zeros.  Nothing ever writes the padding."""

    def __init__(self, rows, cols, dtype, device, ld=None, fill=None, storage=None):
        esz = torch.empty((), dtype=dtype).element_size()
        self.ld = ld if ld is not None else _ru(max(cols, 1), KPAD)
        self.rows, self.cols, self.esz = rows, cols, esz
        prow = _ru(max(rows, 1), KPAD) + KPAD
        if storage is not None:
            # a view of storage shared between workspaces (Engine.shared_storage): it holds FINITE leftovers of other shapes
            # instead of zeros; only for buffers whose every reduction partner is zero-padded itself (see Workspace.GT)
            self.t = storage[:prow * self.ld].view(prow, self.ld)
        else:
            self.t = torch.zeros(prow, self.ld, dtype=dtype, device=device)
        if fill is not None:
            self.t[:rows, :cols].fill_(fill)

    @staticmethod
    def elems(rows, cols):
        return (_ru(max(rows, 1), KPAD) + KPAD) * _ru(max(cols, 1), KPAD)

    def p(self, r=0, c=0):
        return self.t.data_ptr() + (r * self.ld + c) * self.esz

    def view(self):
        return self.t[:self.rows, :self.cols]
