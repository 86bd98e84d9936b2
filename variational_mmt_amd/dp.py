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
import os

import torch


class _Rccl(object):
    """The step's collectives straight through RCCL's C API (ctypes on the librccl.so torch itself has loaded), on a communicator of this
    object's own.  Why: a torch.distributed call costs the host ~50 us (argument checks, work objects, stream bookkeeping) and a step
    issues ten of them -- the one-rank rehearsal on one MI355X was HOST-bound at 2.25 ms per step against 1.72 ms without data
    parallelism (bench.py: host_enqueue_ms); an ncclReduceScatter call is a few microseconds.  The unique id travels through the process
    group that is already up.  Everything here is checked against torch.distributed's own result when the communicator comes up
    (GradSync._direct_selfcheck); any failure leaves the run on torch.distributed, with a line on stderr."""
    F32, I32, SUM, MAX = 7, 2, 0, 2

    def __init__(self, dist, device, lib=None):
        """lib: a loaded librccl (tests/test_dp_gloo.py passes stand-ins that fail at chosen stages, to show that every rank takes
        the same way out); default: the librccl.so of the running torch"""
        import ctypes as C
        import os as _os
        cand = [_os.path.join(_os.path.dirname(torch.__file__), "lib", "librccl.so"), "librccl.so", "/opt/rocm/lib/librccl.so"]
        self.lib, lib_ok = (lib if lib is not None and lib != "missing" else None), 1
        for c in (cand if lib is None else []):
            try:
                self.lib = C.CDLL(c)
                break
            except OSError:
                continue

        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]
        L = self.lib
        try:
            if L is None:
                raise OSError("librccl.so not found")
            L.ncclGetUniqueId.argtypes, L.ncclGetUniqueId.restype = [C.POINTER(UniqueId)], C.c_int
            L.ncclCommInitRank.argtypes, L.ncclCommInitRank.restype = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int], C.c_int
            L.ncclCommDestroy.argtypes, L.ncclCommDestroy.restype = [C.c_void_p], C.c_int
            L.ncclGetErrorString.argtypes, L.ncclGetErrorString.restype = [C.c_int], C.c_char_p
            L.ncclReduceScatter.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
            L.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
            L.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
            for f in (L.ncclReduceScatter, L.ncclAllGather, L.ncclAllReduce):
                f.restype = C.c_int
        except (OSError, AttributeError):          # (no library / a symbol missing on THIS rank: the stages below still run, and say so)
            lib_ok = 0
        world, rank = dist.get_world_size(), dist.get_rank()
        self.comm = None
        self._dist, self._dev = dist, device
        # Bring-up in STAGES, each agreed on through the process group that is already up, so that no rank is ever alone in a blocking
        # call: (1) rank 0 draws the unique id; the broadcast is ALWAYS executed and carries a status byte in front of the id -- a rank 0
        # that failed says so instead of leaving its peers in the broadcast; (2) every rank reports whether it is ready (library loaded,
        # id received) and only if ALL are does anyone enter ncclCommInitRank; (3) the communicator is created on THIS engine's device
        # (ncclCommInitRank binds to the calling thread's current device, which torch.cuda.synchronize(device) does not select).
        uid, ready = UniqueId(), lib_ok
        if rank == 0 and lib_ok:
            try:
                self._ok(L.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
            except Exception:           # noqa: BLE001
                ready = 0
        t = torch.frombuffer(bytearray(bytes([ready]) + bytes(uid)), dtype=torch.uint8).to(device)
        dist.broadcast(t, 0)
        raw = bytes(t.cpu().numpy().tobytes())
        C.memmove(C.byref(uid), raw[1:], 128)
        self._agree(ready and raw[0], "library + unique id", mine=bool(ready))
        comm = C.c_void_p()
        import contextlib
        with (torch.cuda.device(device) if torch.device(device).type == "cuda" else contextlib.nullcontext()):
            if torch.device(device).type == "cuda":
                torch.cuda.synchronize(device)
            rc = L.ncclCommInitRank(C.byref(comm), world, uid, rank)
        if rc == 0:
            self.comm = comm
        self._agree(rc == 0, "ncclCommInitRank" + ("" if rc == 0 else ": " + L.ncclGetErrorString(rc).decode()))
        self.world, self.rank = world, rank
        import atexit
        atexit.register(self.close)         # (ncclCommDestroy on normal exit; GradSync.close() does it earlier)

    def _agree(self, ok, what, mine=None):
        """every rank learns whether ALL ranks passed the stage (one small all-reduce through torch.distributed); raises on every rank if not"""
        mine = bool(ok) if mine is None else mine
        v = torch.tensor([1 if ok else 0], device=self._dev, dtype=torch.int32)
        self._dist.all_reduce(v, op=self._dist.ReduceOp.MIN)
        if int(v.item()) != 1:
            self.close()
            raise RuntimeError("direct RCCL bring-up: stage '%s' failed on %s" % (what, "this rank" if not mine else "another rank"))

    def _ok(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %s" % (what, self.lib.ncclGetErrorString(rc).decode()))

    @staticmethod
    def _stream(t):
        return torch.cuda.current_stream(t.device).cuda_stream

    def reduce_scatter(self, seg, n_per_rank):
        """seg [world * n] f32 in place: this rank's piece seg[rank * n : (rank + 1) * n] receives the sum (RCCL's in-place form)"""
        p = seg.data_ptr()
        self._ok(self.lib.ncclReduceScatter(p, p + 4 * self.rank * n_per_rank, n_per_rank, self.F32, self.SUM, self.comm, self._stream(seg)), "ncclReduceScatter")

    def all_gather(self, seg, n_per_rank):
        """seg [world * n] f32 in place: every rank's piece -> the whole on every rank"""
        p = seg.data_ptr()
        self._ok(self.lib.ncclAllGather(p + 4 * self.rank * n_per_rank, p, n_per_rank, self.F32, self.comm, self._stream(seg)), "ncclAllGather")

    def all_gather_into(self, out, row):
        self._ok(self.lib.ncclAllGather(row.data_ptr(), out.data_ptr(), row.numel(), self.F32, self.comm, self._stream(row)), "ncclAllGather")

    def all_reduce(self, t, op="sum"):
        dt = {torch.float32: self.F32, torch.int32: self.I32}[t.dtype]
        self._ok(self.lib.ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), dt, self.SUM if op == "sum" else self.MAX, self.comm, self._stream(t)), "ncclAllReduce")

    def close(self):
        if getattr(self, "comm", None):
            try:
                self.lib.ncclCommDestroy(self.comm)
            except Exception:
                pass
            self.comm = None


class GradSync(object):
    def __init__(self, engine=None, flat=None, bucket_elems=16 * 1024 * 1024, sharded=None, direct=None):
        """sharded (default: on, VMMT_DP_SHARDED=0 switches it off): reduce-scatter the gradients, run clip + Adam on this rank's
        1 / world of every arena segment, all-gather the parameters (Engine._optim_step_sharded) -- the same bytes on the wire as the
        all-reduce it replaces (an all-reduce IS a reduce-scatter followed by an all-gather), but the optimiser's 28 B/param of HBM
        traffic shrinks world-fold and the all-gather half moves behind Adam, where the decoder-side half of it overlaps the next
        step's encoder.  sharded=False: all-reduce + replicated Adam."""
        self.engine = engine
        self._flat = flat
        self.bucket_elems = bucket_elems
        self.dist = None
        self.world = 1
        self.rank = 0
        self.backend = None
        self.sharded = (os.environ.get("VMMT_DP_SHARDED", "1") == "1") if sharded is None else bool(sharded)
        self._native = {}
        self._direct = None            # _Rccl: the step's collectives straight through RCCL's C API (backend nccl)
        self.timing, self.exposed, self.step_event, self._comm = None, [], None, None     # bench.py switches the timing on for a few steps
        self.branch_log = []           # (collective, "native" | "fallback", reason): which form of each collective this run uses
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("VMMT_DP_FORCE", "0") == "1"):
            # (VMMT_DP_FORCE=1: attach to a world of ONE as well -- every collective then runs through the backend with itself as the
            #  only peer: the one-GPU rehearsal of the RCCL code path, tests/test_gpu_nccl_world1.py)
            self.dist = dist
            self.world = dist.get_world_size()
            self.rank = dist.get_rank()
            self.backend = dist.get_backend()
            self._direct = None
            # OPT-IN (VMMT_DP_DIRECT=1, or direct=True): the direct path has run with a world of ONE only (tests/test_gpu_nccl_world1.py,
            # the one-GPU rehearsal) -- no node with two or more GPUs has been available to this project.  Until a multi-GPU parity run
            # has passed, a data-parallel job uses torch.distributed's own tensor collectives unless it asks for the direct calls.
            want = bool(direct) if direct is not None else os.environ.get("VMMT_DP_DIRECT", "0") == "1"
            if want and self.backend == "nccl" and os.environ.get("VMMT_DP_NATIVE", "1") != "0" and self.flat.is_cuda:
                self._direct_up()
            elif self.backend == "nccl":
                self._log("step collectives", "torch.distributed", "direct RCCL is opt-in: VMMT_DP_DIRECT=1" if not want else "VMMT_DP_NATIVE=0 / no device arena")
            if engine is not None and getattr(engine, "dense_optimizer", False):
                self.sharded = False               # torch's dense optimisers (-optim sgd|adagrad|adadelta) read the whole reduced gradient
            if engine is not None and engine.dp is None:
                # an error here must stop the run: `world` / `rank` set without `engine.dp` would train unsynchronised replicas
                # plans built for one process carry the lazy row-wise update of the embedding tables (engine._build_row_tables): every
                # row is brought up to date and the tables' gradients cleared before the dense, sharded optimiser takes over
                engine.flush_lazy_rows()
                engine.drop_workspaces()
                for t in engine.row_tables:
                    engine.flat_g[t["off"]:t["end"]].zero_()
                engine.dp = self
                # the seed is shared (identical initial parameters); the noise streams must not be: each replica draws
                # its own eps ~ N(0, I) and dropout masks (counter-based RNG: disjoint counter ranges per rank)
                engine.rng_counter += self.rank * (1 << 40)

    # ---- timing of the collectives (bench.py: the "dp" block of a multi-GPU line) ------------------------------------------------
    def time_begin(self, stream):
        """-> start event on `stream` in front of a collective, or None when timing is off (the default: no events on the step)"""
        if self.timing is None:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream)
        return ev

    def time_end(self, start, stream, what, lo, hi):
        """end event of a collective on the stream it ran on (the collectives are synchronous operations on explicit streams)"""
        if start is None:
            return
        end = torch.cuda.Event(enable_timing=True)
        end.record(stream)
        self.timing.append((what, lo, hi, start, end, self.step_event))

    def timed_wait(self, what, fn):
        """run fn() (a wait of the compute stream for collectives) between two events when timing is on: the EXPOSED part"""
        if self.timing is None:
            return fn()
        st = torch.cuda.current_stream(self.flat.device)
        a = torch.cuda.Event(enable_timing=True)
        a.record(st)
        out = fn()
        b = torch.cuda.Event(enable_timing=True)
        b.record(st)
        self.exposed.append((what, a, b))
        return out

    def timing_report(self):
        """[{collective, elems, bytes, ms, GB/s (bus: bytes (w-1)/w per rank), issued_at_us}] averaged per segment, exposed ms per step"""
        torch.cuda.synchronize()
        by = {}
        for what, lo, hi, s, e, t0 in self.timing:
            d = by.setdefault((what, lo, hi), dict(ms=[], at=[]))
            d["ms"].append(s.elapsed_time(e))
            if t0 is not None:
                d["at"].append(t0.elapsed_time(s) * 1e3)
        segs = []
        for (what, lo, hi), d in sorted(by.items(), key=lambda kv: (min(kv[1]["at"]) if kv[1]["at"] else 0.0)):
            ms = sum(d["ms"]) / len(d["ms"])
            nbytes = 4 * (hi - lo)
            wire = nbytes * (self.world - 1) / max(1, self.world) * (2 if what == "all_reduce" else 1)
            segs.append({"collective": what, "bytes": nbytes, "ms": round(ms, 4), "GB/s": round(wire / (ms * 1e-3) / 1e9, 1) if ms > 0 else None,
                         "issued_at_us": round(sum(d["at"]) / len(d["at"]), 1) if d["at"] else None, "samples": len(d["ms"])})
        steps = max(1, max((len(d["ms"]) for d in by.values()), default=1))
        ex = {}
        for what, a, b in self.exposed:
            ex[what] = ex.get(what, 0.0) + a.elapsed_time(b)
        return segs, {k: round(v / steps, 4) for k, v in ex.items()}

    def active(self):
        """True when collectives are issued at all (world > 1, or a forced world of one)"""
        return self.dist is not None

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
        self.serialize()
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
        if not self.active():
            return
        if self.engine is not None:
            self.timed_wait("gradient_wait", self.engine.finish_allreduce)
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
        """does the backend provide the tensor form of this collective?  (RCCL does; gloo lacks reduce_scatter.)  The answer is logged
        (`branch_log`, one line on stderr) and anything but the backend saying "not supported" is an error: a probe that fails for
        another reason on RCCL would otherwise degrade every step to `world` broadcasts per segment without a word"""
        key = (what, t.device.type)
        if key not in self._native:
            forced = os.environ.get("VMMT_DP_NATIVE")            # "1": the tensor collectives or an error; "0": the fallback
            if forced == "0":
                self._native[key] = False
                self._log(what, "fallback", "VMMT_DP_NATIVE=0")
                return False
            try:
                buf = torch.zeros(self.world * 64, dtype=torch.float32, device=t.device)
                if what == "reduce_scatter":
                    self.dist.reduce_scatter_tensor(buf[self.rank * 64:(self.rank + 1) * 64], buf)
                else:
                    self.dist.all_gather_into_tensor(buf, buf[self.rank * 64:(self.rank + 1) * 64].clone())
                self._native[key] = True
                self._log(what, "native", "%s on %s" % (self.backend, t.device.type))
            except (RuntimeError, NotImplementedError) as ex:
                msg = str(ex)
                unsupported = isinstance(ex, NotImplementedError) or any(m in msg.lower() for m in ("not support", "unsupported", "not implemented", "no backend type"))
                if not unsupported or forced == "1" or self.backend == "nccl":
                    raise RuntimeError("data-parallel %s probe failed on backend %s: %s" % (what, self.backend, msg)) from ex
                self._native[key] = False
                self._log(what, "fallback", "%s: %s" % (self.backend, msg.splitlines()[0][:120]))
        return self._native[key]

    def _direct_up(self):
        """bring up the communicator of the direct RCCL path and check it against torch.distributed on small tensors; on ANY failure the
        run stays on torch.distributed (every rank takes the same decision: the verdict is all-reduced through torch.distributed)"""
        dev = self.flat.device
        ok, why = 1, "ok"
        r = None
        try:
            r = _Rccl(self.dist, dev)         # (staged: raises on EVERY rank or on none)
            n = 64
            base = (torch.arange(self.world * n, device=dev, dtype=torch.float32) * 0.25 + self.rank).contiguous()
            mine = slice(self.rank * n, (self.rank + 1) * n)
            # torch.distributed's results FIRST, all of them: these calls are executed by every rank whatever happens to the direct ones
            # below (a rank whose direct call raises goes straight to the verdict; it never leaves a peer alone in a torch collective)
            b = base.clone()
            self.dist.reduce_scatter_tensor(b[mine], base.clone())
            b2 = torch.empty_like(base)
            self.dist.all_gather_into_tensor(b2, b[mine].clone())
            k2 = torch.tensor([1.5 + self.rank], device=dev)
            self.dist.all_reduce(k2)
            g2 = torch.tensor([self.rank + 3], device=dev, dtype=torch.int32)
            self.dist.all_reduce(g2, op=self.dist.ReduceOp.MAX)
            row = torch.arange(9, device=dev, dtype=torch.float32) + self.rank
            rows2 = torch.zeros(self.world, 9, device=dev)
            self.dist.all_gather_into_tensor(rows2.view(-1), row)
            torch.cuda.synchronize(dev)
            a = base.clone()
            r.reduce_scatter(a, n)
            a2 = a.clone()
            r.all_gather(a2, n)
            k1 = torch.tensor([1.5 + self.rank], device=dev)
            r.all_reduce(k1)
            g1 = torch.tensor([self.rank + 3], device=dev, dtype=torch.int32)
            r.all_reduce(g1, "max")
            rows = torch.zeros(self.world, 9, device=dev)
            r.all_gather_into(rows, row)
            torch.cuda.synchronize(dev)
            same = (torch.equal(a[mine], b[mine]) and torch.equal(a2, b2) and torch.equal(k1, k2) and torch.equal(g1, g2) and
                    torch.equal(rows, rows2))
            if not same:
                ok, why = 0, "results differ from torch.distributed"
        except Exception as ex:          # noqa: BLE001  (whatever went wrong: the run continues on torch.distributed)
            ok, why = 0, "%s: %s" % (type(ex).__name__, str(ex).splitlines()[0][:160] if str(ex) else "")
            if r is not None and r.comm is None:
                r = None
        v = torch.tensor([ok], device=dev, dtype=torch.int32)
        self.dist.all_reduce(v, op=self.dist.ReduceOp.MIN)
        if int(v.item()) == 1:
            self._direct = r
            self._log("step collectives", "direct RCCL (ncclReduceScatter / ncclAllGather / ncclAllReduce on a communicator of their own)", "checked against torch.distributed")
        else:
            if r is not None:
                r.close()
            self._log("step collectives", "torch.distributed", "direct RCCL not used on this rank: " + why)

    def all_reduce_tensor(self, t, op="sum"):
        """small tensors of the step (the KL float, the guard word): synchronous on the current stream"""
        if self._direct is not None and t.is_cuda and t.dtype in (torch.float32, torch.int32):
            self._direct.all_reduce(t, op)
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM if op == "sum" else self.dist.ReduceOp.MAX)

    def _log(self, what, branch, why):
        self.branch_log.append((what, branch, why))
        if self.rank == 0:
            import sys
            print("[vmmt dp] %s: %s (%s)" % (what, branch, why), file=sys.stderr, flush=True)

    def native_collectives(self):
        """{collective: bool} for the forms probed so far (bench.py prints it next to the multi-GPU number)"""
        out = {k[0]: v for k, v in self._native.items()}
        out["direct_rccl"] = self._direct is not None
        return out

    # Every collective is a SYNCHRONOUS operation of torch.distributed issued under an explicit stream: the backend then launches it
    # on that stream (observed with RCCL on torch 2.10: the copy kernel of a one-rank all-gather runs on the issuing stream), not on
    # an internal stream of its own behind two event hops.  With `async_op=True` the backend's one internal stream sat blocked on
    # whichever producer it had been told to wait for, and -- the GPU exposes 4 hardware queues to a process, streams share them --
    # held up the main stream's persistent recurrences behind it: 2.2-2.9 ms per step in the one-rank RCCL rehearsal against 1.75 ms
    # without data parallelism.  One communicator must not run two collectives at once, so they are kept in ONE order by stream
    # dependencies: gradient segments and the KL float on the COMM stream (each behind its producer), the norm row and the foreground
    # parameter all-gathers on the main stream behind the COMM stream, the background all-gathers on the side stream behind those, and
    # the next step's KL float behind the background all-gathers (Engine: KL_ALLREDUCE waits for `opt_side_done`).
    def comm_stream(self):
        if self._comm is None:
            try:
                lo_pri = max(torch.cuda.Stream.priority_range())
            except Exception:
                lo_pri = 0
            self._comm = torch.cuda.Stream(device=self.flat.device, priority=lo_pri)
        return self._comm

    def on_comm(self, after, fn, what=None, lo=0, hi=0):
        """run the collective(s) fn() on the COMM stream, behind everything `after` (a stream) has been given so far"""
        comm = self.comm_stream()
        comm.wait_stream(after)
        with torch.cuda.stream(comm):
            t0 = self.time_begin(comm) if what else None
            fn()
            self.time_end(t0, comm, what, lo, hi)
        return comm

    def reduce_scatter(self, flat, lo, hi):
        """sum flat[lo:hi] over the ranks; afterwards this rank's shard (self.shard) of it holds the sum, IN PLACE (the rest of the
        segment is scratch).  Synchronous on the current stream.  One RCCL reduce-scatter when the segment splits evenly; otherwise
        (or on a backend without the tensor form) an all-reduce, whose result restricted to the shard is the same numbers"""
        seg = flat[lo:hi]
        if self._direct is not None and self._even(lo, hi) and flat.is_cuda and flat.dtype == torch.float32:
            self._direct.reduce_scatter(seg, (hi - lo) // self.world)
        elif self._even(lo, hi) and self._probe("reduce_scatter", flat):
            a, b = self.shard(lo, hi)
            self.dist.reduce_scatter_tensor(flat[a:b], seg)
        else:
            self.dist.all_reduce(seg)
        return _Done()

    def reduce_segment(self, flat, lo, hi, producer):
        """the gradient collective of one arena segment from inside the backward plan: on the COMM stream, behind `producer` (the plan
        stream whose kernels completed the segment).  Nothing waits for it here: Engine.finish_allreduce / the shard norms do"""
        if self.sharded:
            return self.on_comm(producer, lambda: self.reduce_scatter(flat, lo, hi), "reduce_scatter", lo, hi)
        return self.on_comm(producer, lambda: self.all_reduce_tensor(flat[lo:hi]), "all_reduce", lo, hi)

    def all_gather(self, flat, lo, hi):
        """every rank's shard of flat[lo:hi] -> the whole segment on every rank, in place; synchronous on the current stream"""
        st = torch.cuda.current_stream(flat.device) if flat.is_cuda else None
        t0 = self.time_begin(st) if st is not None else None
        if self._direct is not None and self._even(lo, hi) and flat.is_cuda and flat.dtype == torch.float32:
            self._direct.all_gather(flat[lo:hi], (hi - lo) // self.world)
        elif self._even(lo, hi) and self._probe("all_gather", flat):
            a, b = self.shard(lo, hi)
            self.dist.all_gather_into_tensor(flat[lo:hi], flat[a:b])
        else:
            for r in range(self.world):          # uneven tail or a backend without the tensor form: one broadcast per owner
                a, b = GradSync.shard(_As(self, r), lo, hi)
                if b > a:
                    self.dist.broadcast(flat[a:b], r)
        self.time_end(t0, st, "all_gather", lo, hi)
        return _Done()

    def serialize(self):
        """the current stream waits for every collective issued so far (COMM stream, background half of the optimiser step): call before
        a collective that is issued outside the step's own ordering (statistics, replica checks, checkpoints)"""
        if self.engine is not None and self.flat.is_cuda:
            self.engine.wait_background()
            owner = self.engine.dp if getattr(self.engine, "dp", None) is not None else self      # (the GradSync whose COMM stream the plans use)
            for comm in {id(c): c for c in (self._comm, owner._comm) if c is not None}.values():
                torch.cuda.current_stream(self.flat.device).wait_stream(comm)

    def all_gather_rows(self, row):
        """[n] on every rank -> [world][n] on every rank (the ranks' norm partials)"""
        out = torch.zeros(self.world, row.numel(), dtype=row.dtype, device=row.device)
        parts = list(out.unbind(0))
        self.dist.all_gather(parts, row.contiguous())
        return torch.stack(parts)

    def all_gather_row_into(self, out, row):
        """row [n] on every rank -> out [world][n] on every rank, no temporaries: ONE tensor all-gather where the backend has it"""
        if self._direct is not None and row.is_cuda and row.dtype == torch.float32 and out.is_contiguous():
            self._direct.all_gather_into(out, row)
        elif self._probe("all_gather", row):
            self.dist.all_gather_into_tensor(out.view(-1), row)
        else:
            self.dist.all_gather(list(out.unbind(0)), row)
        return out

    def gather_moments(self):
        """sharded optimiser: every rank keeps Adam's moments for its shards only; before a checkpoint is written (or the sharding
        is switched off) every rank collects the full moment arenas"""
        if not self.active() or self.engine is None or not self.sharded:
            return
        e = self.engine
        self.serialize()
        for t in (e.flat_m, e.flat_v):
            for lo, hi in e.segments:
                self.all_gather(t, lo, hi)

    def broadcast_replica(self, src=0):
        """re-synchronise the replicas (parameters + Adam moments) from rank `src`.  Replicas stay bit-identical by
        construction (identical reduced gradients, deterministic norm); the trainer calls this every few thousand updates as
        cheap insurance (240 MB x 3 over xGMI) and right after a checkpoint is loaded.  Sharded optimiser: a rank maintains Adam's
        moments for ITS shards only, so rank `src` first collects the live moments from their owners (gather_moments) -- broadcasting
        its own stale copies would reset the other ranks' optimiser history (tests/test_dp_gloo.py)."""
        if not self.active() or self.engine is None:
            return
        e = self.engine
        self.serialize()
        self.gather_moments()
        for t in (e.flat_p, e.flat_m, e.flat_v):
            self.dist.broadcast(t, src)
        e.shadows_dirty = True

    def replicas_identical(self):
        """debug / test aid: True when every rank holds bit-identical parameters (one 8-byte all-reduce pair)"""
        if not self.active() or self.engine is None:
            return True
        self.serialize()
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
        self.serialize()
        self.dist.all_reduce(t)
        return t.tolist()


class _As(object):
    """a view of a GradSync as another rank (shard arithmetic)"""

    def __init__(self, sync, rank):
        self.world, self.rank = sync.world, rank


class _Done(object):
    """what a synchronous collective returns where callers used to get a work handle"""

    def wait(self):
        return None
