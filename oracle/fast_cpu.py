"""TEST INFRASTRUCTURE -- the CPU baseline leg of bench.py.  The restatement in vi1_oracle.py walks the LSTM time steps in
Python; the reference itself runs nn.LSTM (one fused ATen call per layer over a packed sequence: onmt/Models.py:124-129,
140-147; onmt/VI_Model1.py:106,149-152).  `lstm_layer` below plugs torch's fused LSTM into the oracle's forward through the
hook `vi1_oracle.forward(..., lstm_layer=)`, so that the reported CPU baseline runs at the speed of the reference's own CPU
path.  tests/test_oracle_fast_cpu.py checks it against the plain restatement (forward, loss, every gradient) to 1e-6."""
import torch
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

from . import vi1_oracle as O


def lstm_layer(x, mask, lengths, dirs, h0s, c0s):
    """same contract as vi1_oracle._lstm_layer; both directions in ONE fused call"""
    bidir = len(dirs) == 2
    flat = []
    for (w_ih, w_hh, b_ih, b_hh, _rev) in dirs:
        flat += [w_ih, w_hh, b_ih, b_hh]
    hx = (torch.stack(list(h0s)), torch.stack(list(c0s)))
    S = x.shape[0]
    if lengths is not None:
        pk = pack_padded_sequence(x, lengths.cpu(), enforce_sorted=True)
        out, h, c = torch._VF.lstm(pk.data, pk.batch_sizes, hx, flat, True, 1, 0.0, False, bidir)
        out, _ = pad_packed_sequence(torch.nn.utils.rnn.PackedSequence(out, pk.batch_sizes), total_length=S)
    else:
        out, h, c = torch._VF.lstm(x, hx, flat, True, 1, 0.0, False, bidir, False)
    return out, list(h), list(c)


def step_grads(*a, **k):
    return O.step_grads(*a, lstm_layer=lstm_layer, **k)


def clip_and_adam(p, grads, state, lr=0.002, max_grad_norm=5.0, b1=0.9, b2=0.999, eps=1e-9):
    """onmt/Optim.py:94-96 the way the reference's torch executes it: `clip_grad_norm` (one fp32 norm per tensor, scaled in place) then
    torch.optim.Adam's in-place update (exp_avg.mul_().add_(), exp_avg_sq.mul_().addcmul_(), p.addcdiv_()).  Same arithmetic as
    vi1_oracle.clip_and_adam (checked in tests/test_oracle_fast_cpu.py), without its out-of-place 60 M-element temporaries and fp64 norm:
    this is the optimiser leg of the timed CPU baseline.  Updates `p` IN PLACE and returns (p, total norm)."""
    import math
    tot = math.sqrt(sum(float(torch.linalg.vector_norm(g)) ** 2 for g in grads.values()))
    coef = max_grad_norm / (tot + 1e-6)
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    m, v = state.setdefault("m", {}), state.setdefault("v", {})
    bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
    with torch.no_grad():
        for k, g in grads.items():
            if coef < 1:
                g = g.mul_(coef)
            if k not in m:
                m[k], v[k] = torch.zeros_like(p[k]), torch.zeros_like(p[k])
            m[k].mul_(b1).add_(g, alpha=1 - b1)
            v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (v[k].sqrt() / math.sqrt(bc2)).add_(eps)
            p[k].addcdiv_(m[k], denom, value=-lr / bc1)
    return p, tot
