"""Data parallelism for the VI_Model1 step: one process per GPU, gradients of the flat fp32 arena summed with RCCL
(torch.distributed backend "nccl" on ROCm) over xGMI.  The reference has no multi-GPU path at all (it exits when
len(gpuid) > 1, train_mm_vi_model1.py:73-75); the semantics reproduced are those of the single-process reference run on
the CONCATENATED global batch B_g = sum_r B_r:

    loss / B_g  with  loss = sum NLL + image term + (1/B_g) sum_b KL_b        (VILoss.py:460,478; Loss.py:129)

so every rank back-propagates with normalization = B_g and KL batch size = B_g and the gradients are SUMMED (not
averaged) -- see SURVEY.md section 8e.  The arena is ordered by backward completion (generator first), and it is reduced
in a few large buckets (xGMI is per-link bound: fewer, larger collectives)."""
import torch


class GradSync(object):
    def __init__(self, engine=None, flat=None, bucket_elems=16 * 1024 * 1024):
        self.engine = engine
        self._flat = flat
        self.bucket_elems = bucket_elems
        self.dist = None
        self.world = 1
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                self.dist = dist
                self.world = dist.get_world_size()
                if engine is not None:
                    engine.dp = self
        except Exception:
            pass

    @property
    def flat(self):
        return self._flat if self._flat is not None else self.engine.flat_g

    def global_batch(self, local_value):
        """sum of a per-rank integer (batch size / token count) over the ranks"""
        if self.world == 1:
            return local_value
        t = torch.tensor([float(local_value)], device=self.flat.device)
        self.dist.all_reduce(t)
        return float(t.item())

    def buckets(self):
        n = self.flat.numel()
        return [(o, min(n, o + self.bucket_elems)) for o in range(0, n, self.bucket_elems)]

    def all_reduce(self):
        """Engine path: the backward plan already issued one asynchronous all-reduce per arena segment (inference
        networks, generator, attention+decoder, encoder) right behind the kernels that produced it, so that the
        collectives overlap the rest of backward; here the compute stream only waits for them.  A bare flat tensor
        (no engine) is reduced bucket by bucket."""
        if self.world == 1:
            return
        if self.engine is not None:
            self.engine.finish_allreduce()
            return
        for o, e in self.buckets():
            self.dist.all_reduce(self.flat[o:e])
