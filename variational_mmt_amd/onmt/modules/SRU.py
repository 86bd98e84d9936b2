"""`onmt.modules.SRU` as far as the drivers' flag parsing needs it: the reference's `opts.py:2` imports `CheckSRU` (an argparse action
on `-rnn_type`, onmt/modules/SRU.py:15-23) at import time.  The SRU recurrence itself is outside the VI_Model1 hot path: this build
runs LSTMs only, so asking for SRU is refused where the reference would go looking for cupy / pynvrtc."""
import argparse


def check_sru_requirement(abort=False):
    """the reference probes for its CUDA-only dependencies here (SRU.py:29-60); on this path SRU is simply unavailable"""
    if abort:
        raise AssertionError("-rnn_type SRU is not available in variational_mmt_amd (MI355X build: LSTM only)")
    return False


class CheckSRU(argparse.Action):
    def __call__(self, parser, namespace, values, option_string=None):
        if values == "SRU":
            check_sru_requirement(abort=True)
        setattr(namespace, self.dest, values)
