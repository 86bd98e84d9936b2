"""Host-side mirror of the reference's NMTVIModel (onmt/Models.py:737-1011) for the fixed-prior vi-model1: same call
signature and return contract, but forward() launches the HIP kernels of libvmmt.so through variational_mmt_amd.engine
instead of building a torch autograd graph.  Sub-modules exist so that state_dict() carries the reference's parameter
names; their tensors are views into the engine's flat fp32 arena."""
import torch
import torch.nn as nn

from ..engine import Dims, Engine
from .modules.Dists import Normal


class _Holder(nn.Module):
    """parameter container whose attribute path reproduces a reference state-dict key"""

    def forward(self, *a, **k):
        raise RuntimeError("sub-modules of the MI355X VI_Model1 are parameter holders; call the model itself")


def _attach(root, dotted, param):
    mod = root
    parts = dotted.split(".")
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, _Holder())
        mod = mod._modules[p]
    mod.register_parameter(parts[-1], param)


class NMTVIModel(nn.Module):
    """encoder + variational decoder + inference networks of VI_Model1 (global image features): fixed prior, or with
    `conditional` the conditional prior p(z|x) + q(z|x,y,v) + encoder_tgt (Models.py:883-914)."""

    def __init__(self, dims, dtype="bf16", device="cuda", param_init=0.1, seed=0, conditional=False,
                 multimodal_model_type="vi-model1", image_loss_type="logprob"):
        super(NMTVIModel, self).__init__()
        assert bool(conditional) == bool(dims.conditional)
        self.conditional = bool(conditional)
        self.multimodal_model_type = multimodal_model_type
        self.image_loss_type = image_loss_type
        self.model_type = "text"
        if not conditional:
            self.gen_net_global = None
        self.dims = dims
        self.engine = Engine(dims, dtype=dtype, device=device, seed=seed, param_init=param_init)
        for name in self.engine.names_grad + self.engine.names_nograd:
            p = nn.Parameter(self.engine.params[name], requires_grad=True)
            p._vmmt_engine = self.engine
            p._vmmt_name = name
            if name in self.engine.grads:
                p.grad = self.engine.grads[name]
            _attach(self, name, p)

    # nn.Module plumbing: parameters live in the engine's arena on the GPU and must not be moved or re-created
    def _apply(self, fn, recurse=True):
        return self

    def cuda(self, device=None):
        return self

    def cpu(self):
        raise RuntimeError("the MI355X VI_Model1 has no CPU path")

    def zero_grad(self, set_to_none=False):
        pass          # the gradient arena is zeroed by the training forward plan (side stream)

    def state_dict(self, *args, **kwargs):
        self.engine.wait_background()       # (the decoder-side half of the last update runs on, or is held back for, the side stream)
        return super(NMTVIModel, self).state_dict(*args, **kwargs)

    def named_parameters(self, *args, **kwargs):
        self.engine.wait_background()       # (parameters() goes through here: whoever asks for the tensors gets the updated arena)
        return super(NMTVIModel, self).named_parameters(*args, **kwargs)

    def load_state_dict(self, state_dict, strict=True):
        # (the copies below go into the arena through the aliasing nn.Parameters: behind the side stream's half of the last update, and
        #  with every lazily updated embedding row brought up to date first -- a row loaded now must not be replayed for steps it was behind)
        self.engine.wait_background()
        out = super(NMTVIModel, self).load_state_dict(state_dict, strict=strict)
        self.engine.shadows_dirty = True
        return out

    def set_image_tables(self, train=None, valid=None):
        """image-feature arrays -> HBM-resident tables.  Each may be a numpy array [N, D] (the reference's contract,
        train_mm_vi_model1.py:468-469), a tensor already on the device (features.load_image_table), or the path of an
        HDF5 file whose `/global_feats` node is streamed into HBM."""
        self._tables = getattr(self, "_tables", {})
        for k, v in (("train", train), ("valid", valid)):
            if isinstance(v, str):
                from ..features import load_image_table
                v = load_image_table(v, "global_feats", device=self.engine.dev)
            if v is not None:
                self._tables[k] = torch.as_tensor(v).to(device=self.engine.dev, dtype=torch.float32).contiguous()

    def forward(self, src, tgt, lengths, tgt_lengths, img_feats, img_vecs=None, dec_state=None, padding_token=None,
                img_indices=None, img_table=None, eps=None, masks=None, n_tgt_tokens=None):
        """Same positional contract as the reference (Models.py:850).  `img_feats` may be a [B, D] tensor (reference
        behaviour) or None when `img_indices` + `img_table` select rows of an HBM-resident table."""
        if src.dim() == 3:
            src = src[:, :, 0]
        if tgt.dim() == 3:
            tgt = tgt[:, :, 0]
        e = self.engine
        B = src.shape[1]
        if img_indices is None:
            table = img_feats.to(device=e.dev, dtype=torch.float32).contiguous()
            img_indices = torch.arange(B, device=e.dev)
        else:
            table = img_table
        ws = e.forward(src, lengths, tgt, img_indices, training=self.training, eps=eps, masks=masks, table=table,
                       tgt_len=tgt_lengths if self.conditional else None, n_tgt_tokens=n_tgt_tokens)
        S, Tp, H = src.shape[0], tgt.shape[0] - 1, self.dims.hid
        ob = ws.O if (self.training and self.dims.dropout > 0) else ws.AH
        out = ob.t.as_strided((Tp, B, H), (B * ob.ld, ob.ld, 1))
        attns = {
            "std": ws.probs.view(ws.Tp, B, ws.S)[:Tp, :, :S],       # the workspace may be bucketed to a larger (T', S)
            "p_global_image_features": [Normal(ws.mu_v.view(), None)],
            "ground_truth_global_image_features": [ws.img.view()],
            "z_latent": [Normal(ws.mu.view(), ws.sigma.view())],
            # fixed prior: standard normal (Models.py:936-939), implicit in the KL kernel; conditional: p(z|x) = gen_net_global
            "p_latent": [Normal(ws.mu_p.view(), ws.sigma_p.view()) if self.conditional else Normal(None, None)],
            "z0_sample": [ws.z32.view()],
            "zz": [None], "logdet": [None],
            "_ws": ws,
        }
        return out, attns, None
