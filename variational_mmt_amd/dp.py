"""Data parallelism for the VI_Model1 step: one process per GPU, gradients of the flat fp32 arena summed with RCCL
(torch.distributed backend "nccl" on ROCm) over xGMI.  The reference has no multi-GPU path at all (it exits when
len(gpuid) > 1, train_mm_vi_model1.py:73-75); the semantics reproduced are those of the single-process reference run on
the CONCATENATED global batch B_g = sum_r B_r:

    loss / B_g  with  loss = sum NLL + image term + max(m * (1/B_g) sum_b KL_b, margin)        (VILoss.py:460-478; Loss.py:129)

so every rank back-propagates with normalization = B_g and KL batch size = B_g and the gradients are SUMMED (not
averaged) -- see SURVEY.md section 8e.  What crosses ranks per step:

  * the gradient arena, reduced segment by segment from inside the backward plan (engine/backward.py: ALLREDUCE entries) so that the
    collectives overlap the rest of backward (xGMI is per-link bound: few, large collectives);
  * ONE float, the KL sum, before the latent backward (free bits compares the GLOBAL batch-mean KL with the margin);
  * nothing else: the global batch size / token count come from the loader (`onmt.io.OrderedIterator(dp_rank=, dp_world=)`
    cuts every global minibatch into the ranks' shares on the host), the gradient norm is a deterministic reduction of the
    (bit-identical) reduced gradients on every rank, and eps / dropout streams are offset by the rank."""
import torch


class GradSync(object):
    def __init__(self, engine=None, flat=None, bucket_elems=16 * 1024 * 1024, sharded=None):
        """sharded (default: on, VMMT_DP_SHARDED=0 switches it off): reduce-scatter the gradients, run clip + Adam on this rank's
        1 / world of every arena segment, all-gather the parameters (Engine._optim_step_sharded) -- the same bytes on the wire as the
        all-reduce it replaces (an all-reduce IS a reduce-scatter followed by an all-gather), but the optimiser's 28 B/param of HBM
        traffic shrinks world-fold and the all-gather half moves behind Adam, where the decoder-side half of it overlaps the next
        step's encoder.  sharded=False: all-reduce + replicated Adam."""
        import os
        self.engine = engine
        self._flat = flat
        self.bucket_elems = bucket_elems
        self.dist = None
        self.world = 1
        self.rank = 0
        self.sharded = (os.environ.get("VMMT_DP_SHARDED", "1") == "1") if sharded is None else bool(sharded)
        self._native = {}
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                self.dist = dist
                self.world = dist.get_world_size()
                self.rank = dist.get_rank()
                if engine is not None and engine.dp is None:
                    engine.drop_workspaces()           # (plans built for one process carry the row-wise gradient bookkeeping of the embedding tables: engine._build_row_tables)
                    engine.dp = self
                    # the seed is shared (identical initial parameters); the noise streams must not be: each replica draws
                    # its own eps ~ N(0, I) and dropout masks (counter-based RNG: disjoint counter ranges per rank)
                    engine.rng_counter += self.rank * (1 << 40)
        except Exception:
            pass

    @property
    def flat(self):
        return self._flat if self._flat is not None else self.engine.flat_g

    def global_sizes(self, *local_values):
        """FALLBACK for loaders that do not shard by rank themselves: sums of per-rank integers (batch size, token count)
        over the ranks -- ONE blocking all-reduce + device-to-host copy for all of them.  The sharding iterator
        (`onmt.io.OrderedIterator(dp_rank=, dp_world=)`) knows the global figures on the host and never calls this."""
        if self.world == 1:
            return tuple(float(v) for v in local_values)
        t = torch.tensor([float(v) for v in local_values], device=self.flat.device, dtype=torch.float64)
        self.dist.all_reduce(t)
        return tuple(t.tolist())

    def global_batch(self, local_value):
        return self.global_sizes(local_value)[0]

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

    # ---- sharded optimiser: ownership and the two collectives -----------------------------------------------------------
    def shard(self, lo, hi):
        """this rank's share [a, b) of the arena segment [lo, hi): the segment cut into `world` equal pieces of whole 64-element
        units (engine.SEG_ALIGN makes hi - lo a multiple of 64 * 8 for every segment but possibly the last one, whose tail piece
        may be shorter or empty)"""
        n = hi - lo
        per = -(-n // self.world)
        per = -(-per // 64) * 64
        a = min(hi, lo + self.rank * per)
        return a, min(hi, a + per)

    def _even(self, lo, hi):
        n = hi - lo
        return n % self.world == 0 and (n // self.world) % 64 == 0

    def _probe(self, what, t):
        """does the backend provide the tensor form of this collective?  (RCCL does; gloo lacks reduce_scatter)"""
        key = (what, t.device.type)
        if key not in self._native:
            try:
                buf = torch.zeros(self.world * 64, dtype=torch.float32, device=t.device)
                if what == "reduce_scatter":
                    self.dist.reduce_scatter_tensor(buf[self.rank * 64:(self.rank + 1) * 64], buf)
                else:
                    self.dist.all_gather_into_tensor(buf, buf[self.rank * 64:(self.rank + 1) * 64].clone())
                self._native[key] = True
            except Exception:
                self._native[key] = False
        return self._native[key]

    def reduce_scatter(self, flat, lo, hi):
        """sum flat[lo:hi] over the ranks; afterwards this rank's shard (self.shard) of it holds the sum, IN PLACE (the rest of the
        segment is scratch).  Returns a work handle (wait() makes the current stream wait).  One RCCL reduce-scatter when the
        segment splits evenly; otherwise (or on a backend without it: gloo in the tests) an all-reduce, whose result restricted to
        the shard is the same numbers"""
        seg = flat[lo:hi]
        if self._even(lo, hi) and self._probe("reduce_scatter", flat):
            a, b = self.shard(lo, hi)
            return self.dist.reduce_scatter_tensor(flat[a:b], seg, async_op=True)
        return self.dist.all_reduce(seg, async_op=True)

    def all_gather(self, flat, lo, hi):
        """every rank's shard of flat[lo:hi] -> the whole segment on every rank, in place"""
        if self._even(lo, hi) and self._probe("all_gather", flat):
            a, b = self.shard(lo, hi)
            return self.dist.all_gather_into_tensor(flat[lo:hi], flat[a:b], async_op=True)
        works = []
        for r in range(self.world):          # uneven tail or a backend without the tensor form: one broadcast per owner
            a, b = GradSync.shard(_As(self, r), lo, hi)
            if b > a:
                works.append(self.dist.broadcast(flat[a:b], r, async_op=True))
        return _Works(works)

    def all_gather_rows(self, row):
        """[n] on every rank -> [world][n] on every rank (the ranks' norm partials)"""
        out = torch.zeros(self.world, row.numel(), dtype=row.dtype, device=row.device)
        parts = list(out.unbind(0))
        self.dist.all_gather(parts, row.contiguous())
        return torch.stack(parts)

    def gather_moments(self):
        """sharded optimiser: every rank keeps Adam's moments for its shards only; before a checkpoint is written (or the sharding
        is switched off) every rank collects the full moment arenas"""
        if self.world == 1 or self.engine is None or not self.sharded:
            return
        e = self.engine
        for t in (e.flat_m, e.flat_v):
            for lo, hi in e.segments:
                self.all_gather(t, lo, hi).wait()

    def broadcast_replica(self, src=0):
        """re-synchronise the replicas (parameters + Adam moments) from rank `src`.  Replicas stay bit-identical by
        construction (identical reduced gradients, deterministic norm); the trainer calls this every few thousand updates as
        cheap insurance (240 MB x 3 over xGMI) and right after a checkpoint is loaded."""
        if self.world == 1 or self.engine is None:
            return
        e = self.engine
        for t in (e.flat_p, e.flat_m, e.flat_v):
            self.dist.broadcast(t, src)
        e.shadows_dirty = True

    def replicas_identical(self):
        """debug / test aid: True when every rank holds bit-identical parameters (one 8-byte all-reduce pair)"""
        if self.world == 1 or self.engine is None:
            return True
        p = self.engine.flat_p
        h = torch.stack([p.double().sum(), p.double().abs().sum()])
        lo, hi = h.clone(), h.clone()
        self.dist.all_reduce(lo, op=self.dist.ReduceOp.MIN)
        self.dist.all_reduce(hi, op=self.dist.ReduceOp.MAX)
        return bool(torch.equal(lo, hi))

    def reduce_stats(self, values):
        """sum a list of per-rank statistics (floats) over the ranks: every rank's VIStatistics holds its SHARE of the
        global figures (NLL / words / correct of its sentences, its share of the batch-mean KL and of the image term)"""
        if self.world == 1:
            return list(values)
        t = torch.tensor(list(values), device=self.flat.device, dtype=torch.float64)
        self.dist.all_reduce(t)
        return t.tolist()


class _As(object):
    """a view of a GradSync as another rank (shard arithmetic)"""

    def __init__(self, sync, rank):
        self.world, self.rank = sync.world, rank


class _Works(object):
    def __init__(self, works):
        self.works = works

    def wait(self):
        for w in self.works:
            w.wait()
