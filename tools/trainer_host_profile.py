"""where the host time of a step through onmt.TrainerMultimodal goes (GPU box): cProfile over one epoch of `bench.py --through-trainer`'s
second epoch.  usage: python tools/trainer_host_profile.py [n_lines] [config] [batch]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

a = type("A", (), dict(config=sys.argv[2] if len(sys.argv) > 2 else "2", batch=int(sys.argv[3]) if len(sys.argv) > 3 else 256, dtype="bf16",
                       dropout=0.5, conditional=False))()
pr = cProfile.Profile()
real = bench.time.perf_counter
state = {"n": 0}


def hook():
    # profile only the SECOND epoch: through_trainer calls perf_counter at the start of each epoch
    state["n"] += 1
    if state["n"] == 4:
        pr.enable()
    return real()


bench.time.perf_counter = hook
bench.through_trainer(a, torch.device("cuda", 0), 0, 1)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 45)
