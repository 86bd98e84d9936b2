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
