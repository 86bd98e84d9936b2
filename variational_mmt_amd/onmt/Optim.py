"""Mirror of onmt.Optim (reference: onmt/Optim.py:34-114): same constructor, `set_parameters`, `step`,
`update_learning_rate`, `.lr`, `.optimizer`; for method 'adam' (every reference recipe) the update is ONE fused HIP
kernel over the engine's flat arena (global-norm clip + Adam, eps = 1e-9).

Checkpoint format (TrainerMultimodal.py:580-587 pickles the whole object as checkpoint['optim']): this class pickles to
what the reference's does -- `params` = list of nn.Parameter, `optimizer` = a genuine torch.optim.Adam over them with its
per-parameter state (step / exp_avg / exp_avg_sq) -- so a checkpoint written here loads in the reference and vice versa
(tests/test_checkpoint_interop.py, with a checkpoint written by the real reference).

Resume semantics AS EXECUTED by the reference: `build_optim` (train_mm_vi_model1.py:433-454) takes checkpoint['optim'] and
then calls `set_parameters`, which constructs a NEW torch.optim.Adam (Optim.py:56-70): lr, _step and the decay flags
survive a resume, the Adam moments and its step counter do NOT.  `set_parameters` here does the same by default;
`Optim.resume_adam_state = True` (an extension) carries the moments over instead."""
import torch
import torch.nn as nn


class _ArenaAdam(object):
    """stands where torch.optim.Adam stands in the reference's Optim: param_groups / state_dict / load_state_dict"""

    def __init__(self, lr, betas, eps=1e-9):
        self.param_groups = [{"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": 0, "amsgrad": False}]
        self._saved = None        # a loaded state dict waiting for set_parameters()
        self._optim = None

    def bind(self, optim):
        self._optim = optim

    def state_dict(self):
        o = self._optim
        if o is None or o.engine is None:
            return self._saved or {"state": {}, "param_groups": self.param_groups}
        e = o.engine
        state = {}
        for i, p in enumerate(o.params):
            name = getattr(p, "_vmmt_name", None)
            if name is None or name not in e.grads or e.step_count == 0:
                continue                      # parameters that never received a gradient have no Adam state (H6)
            off, shp = e.offsets[name]
            n = p.numel()
            state[i] = {"step": torch.tensor(float(e.step_count), device="cpu"),
                        "exp_avg": e.flat_m[off:off + n].view(*shp).detach().cpu().clone(),
                        "exp_avg_sq": e.flat_v[off:off + n].view(*shp).detach().cpu().clone()}
        groups = [dict(self.param_groups[0], params=list(range(len(o.params))))]
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        self._saved = sd
        if self._optim is not None and self._optim.engine is not None:
            self._optim._push_state(sd)


class Optim(object):
    resume_adam_state = False      # see the module docstring: False = as executed by the reference

    def __init__(self, method, lr, max_grad_norm, lr_decay=1, start_decay_at=None, beta1=0.9, beta2=0.999,
                 adagrad_accum=0.0, decay_method=None, warmup_steps=4000, model_size=None):
        self.last_ppl = None
        self.lr = lr
        self.original_lr = lr
        self.max_grad_norm = max_grad_norm
        self.method = method
        self.lr_decay = lr_decay
        self.start_decay_at = start_decay_at
        self.start_decay = False
        self._step = 0
        self.betas = [beta1, beta2]
        self.adagrad_accum = adagrad_accum
        self.decay_method = decay_method
        self.warmup_steps = warmup_steps
        self.model_size = model_size
        self.engine = None
        self.params = []
        self.optimizer = None

    def set_parameters(self, params):
        old_params = list(self.params) if self.params else []       # of an unpickled checkpoint: CPU tensors in the WRITER's order
        self.params = [p for p in params if p.requires_grad]
        engines = {id(getattr(p, "_vmmt_engine", None)): getattr(p, "_vmmt_engine", None) for p in self.params}
        engines.pop(id(None), None)
        if len(engines) != 1:
            raise RuntimeError("Optim.set_parameters expects the parameters of one variational_mmt_amd model")
        self.engine = list(engines.values())[0]
        if self.method == "adam":
            saved = None
            if self.resume_adam_state and self.optimizer is not None:
                # an unpickled checkpoint['optim']: a torch.optim.Adam (reference / this class's pickle) or a loaded _ArenaAdam
                saved = self.optimizer._saved if isinstance(self.optimizer, _ArenaAdam) else self.optimizer.state_dict()
                if saved is not None and old_params:
                    saved = dict(saved, state=self._remap_state(saved.get("state", {}), old_params))
            self.optimizer = _ArenaAdam(self.lr, self.betas)
            self.optimizer.bind(self)
            e = self.engine
            e.flat_m.zero_()          # a NEW Adam, as in the reference (Optim.py:68-70)
            e.flat_v.zero_()
            e.step_count = 0
            if saved is not None:
                self._push_state(saved)
        elif self.method in ("sgd", "adagrad", "adadelta"):
            # not on the hot path (every reference recipe uses adam): thin delegation to torch on the arena views
            self.engine.row_adam = False           # torch's dense optimisers on the arena views read / clear the whole gradient
            # ... and the whole REDUCED gradient under data parallelism: all-reduce + replicated update instead of the sharded
            # optimiser's reduce-scatter (which leaves the sum in this rank's 1 / world of every segment only)
            self.engine.dense_optimizer = True
            if self.engine.dp is not None:
                self.engine.dp.sharded = False
            self.engine.drop_workspaces()
            cls = {"sgd": torch.optim.SGD, "adagrad": torch.optim.Adagrad, "adadelta": torch.optim.Adadelta}[self.method]
            self.optimizer = cls(self.params, lr=self.lr)
        else:
            raise RuntimeError("Invalid optim method: " + self.method)

    def _remap_state(self, state, old_params):
        """optimizer state of a checkpoint is keyed by the parameter's position in the WRITER's model.parameters() -- the
        reference's module order, not this model's arena order.  The model was just loaded from the same checkpoint, so every old
        parameter is found again by shape + value among the new ones."""
        new_cpu = [p.detach().cpu() for p in self.params]
        by_shape = {}
        for j, t in enumerate(new_cpu):
            by_shape.setdefault(tuple(t.shape), []).append(j)
        used, out = set(), {}
        for i, st in state.items():
            old = old_params[int(i)].detach().cpu()
            cands = [j for j in by_shape.get(tuple(old.shape), []) if j not in used]
            if not cands:
                raise RuntimeError("checkpoint optimizer state: no parameter of shape %s in the model" % (tuple(old.shape),))
            j = next((j for j in cands if torch.equal(new_cpu[j].float(), old.float())), cands[0])
            used.add(j)
            out[j] = st
        return out

    def _push_state(self, sd):
        """torch.optim.Adam state-dict layout: state keyed by the parameter's position in param_groups[0]['params'], which is
        the position in `self.params` (both follow model.parameters())"""
        e = self.engine
        step = 0
        for i, st in sd.get("state", {}).items():
            p = self.params[int(i)]
            off, shp = e.offsets[p._vmmt_name]
            n = p.numel()
            e.flat_m[off:off + n].copy_(st["exp_avg"].reshape(-1))
            e.flat_v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            step = max(step, int(float(st["step"])))
        e.step_count = step

    def _set_rate(self, lr):
        self.lr = lr
        self.optimizer.param_groups[0]["lr"] = self.lr

    def step(self):
        self._step += 1
        if self.decay_method == "noam":
            self._set_rate(self.original_lr * (self.model_size ** (-0.5) *
                                               min(self._step ** (-0.5), self._step * self.warmup_steps ** (-1.5))))
        if self.method == "adam":
            self.engine.optim_step(lr=self.lr, max_grad_norm=self.max_grad_norm or 0.0, beta1=self.betas[0], beta2=self.betas[1],
                                   eps=1e-9)
        else:
            # torch's dense optimisers know nothing of the engine's guard word (a persistent recurrence that timed out left garbage
            # gradients; the arena Adam kernels read the guard on the device and skip): look before every update -- a synchronisation,
            # off the hot path (every reference recipe uses adam) -- and on every data-parallel rank together
            e = self.engine
            if e.dp_on():
                e.finish_allreduce()
                e.dp.dist.all_reduce(e._guard[:1], op=e.dp.dist.ReduceOp.MAX)
            if int(e._guard[0].item()) != 0:
                e.check_async_errors()             # raises (Engine._seq_timeout_fallback with a dense optimiser)
            if self.max_grad_norm:
                torch.nn.utils.clip_grad_norm_(self.params, self.max_grad_norm)
            self.optimizer.step()
            self.engine.shadows_dirty = True

    def update_learning_rate(self, ppl, epoch):
        if self.start_decay_at is not None and epoch >= self.start_decay_at:
            self.start_decay = True
        if self.last_ppl is not None and ppl > self.last_ppl:
            self.start_decay = True
        if self.start_decay:
            self.lr = self.lr * self.lr_decay
            print("Decaying learning rate to %g" % self.lr)
        self.last_ppl = ppl
        self.optimizer.param_groups[0]["lr"] = self.lr

    # checkpoint['optim'] is this object pickled (TrainerMultimodal.py:586): same content as the reference's pickle
    def __getstate__(self):
        d = dict(self.__dict__)
        d["engine"] = None
        d.pop("_ckpt_cpu", None)
        e = self.engine
        if e is None or not isinstance(self.optimizer, _ArenaAdam):
            d["params"] = []
            return d
        shared = getattr(self, "_ckpt_cpu", None) or {}        # CPU copies the trainer already made for checkpoint['model']
        cpu_params = []
        for p in self.params:
            t = shared.get(p._vmmt_name)
            if t is None:
                t = p.detach().cpu().clone()
            cpu_params.append(nn.Parameter(t, requires_grad=True))
        adam = torch.optim.Adam(cpu_params, lr=self.lr, betas=tuple(self.betas), eps=1e-9)
        if e.step_count > 0:
            for q, p in zip(cpu_params, self.params):
                name = p._vmmt_name
                if name not in e.grads:
                    continue                  # never received a gradient: torch.optim.Adam holds no state for it (H6)
                off, shp = e.offsets[name]
                n = p.numel()
                adam.state[q] = {"step": torch.tensor(float(e.step_count), device="cpu"),
                                 "exp_avg": e.flat_m[off:off + n].view(*shp).detach().cpu().clone(),
                                 "exp_avg_sq": e.flat_v[off:off + n].view(*shp).detach().cpu().clone()}
        d["params"] = cpu_params
        d["optimizer"] = adam
        return d
