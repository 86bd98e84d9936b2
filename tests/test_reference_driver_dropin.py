"""Drop-in check of the `onmt` mirror against the REFERENCE's own driver (build container only: /root/reference is not on the GPU
box).  After `variational_mmt_amd.install_as_onmt()`:
  1. the reference's `train_mm_vi_model1.py` is executed as it lies (runpy; module level = its imports incl. the reference's own
     `opts.py` -> `onmt.modules.SRU.CheckSRU`, flag parsing, file checks) and its `main()` runs on the CPU up to the first call that
     needs the GPU -- `make_vi_model_mmt` -- through `tables.open_file(...).root.global_feats[:]`, the plain `torch.load` of the
     dataset / vocabulary pickles, `load_fields_from_vocab` and `collect_features`;
  2. every `onmt.*` / `opts.*` / `tables.*` attribute chain in the source of the two drivers (train_mm_vi_model1.py,
     translate_mm_vi.py) resolves against the installed modules, and the mirror's callables accept the keyword arguments the
     driver passes (signature binding), for the part of the flow a CPU cannot execute."""
import ast
import os
import subprocess
import sys

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "train_mm_vi_model1.py")), reason="reference not mounted")

_RUN = r'''
import runpy, sys
sys.path.insert(0, %(root)r)
import variational_mmt_amd
variational_mmt_amd.install_as_onmt()
sys.path.insert(0, %(ref)r)                # the driver's own directory (python puts it there when the script is run)
sys.argv = %(argv)r
ns = runpy.run_path(%(ref)r + "/train_mm_vi_model1.py", run_name="reference_driver")
import onmt, opts, tables
assert onmt.__name__ == "variational_mmt_amd.onmt" and opts.__file__.startswith(%(ref)r), (onmt.__name__, opts.__file__)
assert tables.__name__.endswith("h5tables")
print("MODULE-LEVEL-OK", ns["opt"].rnn_size, ns["opt"].brnn)
try:
    ns["main"]()
except RuntimeError as e:
    import traceback
    tb = traceback.extract_tb(e.__traceback__)
    print("STOPPED-IN", tb[-1].name, "|", e)
'''


def test_reference_driver_runs_to_the_first_gpu_call(tmp_path):
    argv = ["train_mm_vi_model1.py", "-data", G + "/textdata/demo", "-save_model", str(tmp_path / "m"),
            "-path_to_train_img_feats", G + "/h5/pt_feats2048.h5", "-path_to_valid_img_feats", G + "/h5/pt_feats2048.h5",
            "--multimodal_model_type", "vi-model1", "--z_latent_dim", "8", "--use_global_image_features", "-rnn_size", "32",
            "-word_vec_size", "16", "-layers", "1", "-encoder_type", "brnn", "-optim", "adam", "-learning_rate", "0.002",
            "-batch_size", "8", "-epochs", "1"]
    r = subprocess.run([sys.executable, "-c", _RUN % dict(root=ROOT, ref=REF, argv=argv)], capture_output=True, text=True, cwd=str(tmp_path),
                       timeout=600)
    out = r.stdout
    assert r.returncode == 0, r.stderr[-3000:]
    assert "MODULE-LEVEL-OK 32 True" in out
    assert "Using global image features..." in out                           # tables.open_file(...).root.global_feats[:]
    assert "number of examples: 57" in out                                   # torch.load of demo.train.1.pt
    assert " * vocabulary size. source = 14; target = 16" in out            # load_fields_from_vocab
    assert "Building model..." in out
    assert "STOPPED-IN make_vi_model_mmt | variational_mmt_amd needs -gpuid" in out, out[-2000:]


def _chains(tree):
    """dotted names rooted at onmt / opts / tables that the source uses as attribute chains"""
    found = set()
    for node in ast.walk(tree):
        if isinstance(node, ast.Attribute):
            parts, cur = [], node
            while isinstance(cur, ast.Attribute):
                parts.append(cur.attr)
                cur = cur.value
            if isinstance(cur, ast.Name) and cur.id in ("onmt", "opts", "tables"):
                found.add(".".join([cur.id] + parts[::-1]))
    return found


@pytest.mark.parametrize("script", ["train_mm_vi_model1.py", "translate_mm_vi.py"])
def test_every_attribute_the_drivers_use_resolves(script):
    src = open(os.path.join(REF, script)).read()
    tree = ast.parse(src)
    chains = sorted(_chains(tree))
    imports = sorted({a.name for n in ast.walk(tree) if isinstance(n, ast.Import) for a in n.names if a.name.split(".")[0] in ("onmt", "tables")} |
                     {n.module + "." + a.name for n in ast.walk(tree) if isinstance(n, ast.ImportFrom) and n.module and
                      n.module.split(".")[0] == "onmt" for a in n.names})
    assert any(c.startswith("onmt.ModelConstructor") for c in chains) and "tables.open_file" in chains
    prog = r'''
import importlib, sys
sys.path.insert(0, %r)
import variational_mmt_amd
variational_mmt_amd.install_as_onmt()
from variational_mmt_amd import opts
sys.modules["opts"] = opts              # (the build's own flag module: same surface as the reference's opts.py)
import onmt, tables
bad = []
for name in %r:
    try:
        importlib.import_module(name)
    except ImportError:
        mod, _, attr = name.rpartition(".")
        try:
            getattr(importlib.import_module(mod), attr)
        except Exception as e:
            bad.append(("import " + name, repr(e)))
roots = {"onmt": onmt, "opts": opts, "tables": tables}
for chain in %r:
    parts = chain.split(".")
    obj = roots[parts[0]]
    for i, p in enumerate(parts[1:], 1):
        if not hasattr(obj, p):
            # attributes of INSTANCES (trainer.early_stop...) never start at a module root; a module-rooted chain must resolve fully
            bad.append((chain, "no attribute %%s on %%s" %% (p, ".".join(parts[:i]))))
            break
        obj = getattr(obj, p)
print("BAD", bad)
''' % (ROOT, imports, chains)
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "BAD []" in r.stdout, r.stdout[-3000:]


def test_mirror_signatures_accept_the_drivers_calls():
    """the calls of train_mm_vi_model1.py that a CPU cannot execute (:217-242 loss, :263-272 trainer, :440-452 optimiser): the
    mirror's callables bind the driver's positional / keyword arguments"""
    import inspect

    from variational_mmt_amd.onmt import Optim, TrainerMultimodal, VIStatistics
    from variational_mmt_amd.onmt.VILoss import NMTVIModel1LossCompute
    inspect.signature(NMTVIModel1LossCompute.__init__).bind(
        None, "generator", "tgt_vocab", label_smoothing=0.0, use_kl_annealing=False, use_kl_freebits=False, kl_freebits_margin=0.0,
        kl_annealing_current=0.0, kl_annealing_increment=1e-4, kl_annealing_warmup_steps=500, image_loss_type="logprob",
        use_local_image_features=False, two_step_image_prediction=False)
    inspect.signature(TrainerMultimodal.__init__).bind(
        None, "model", "train_loss", "valid_loss", "optim", 0, 32, "text", "sents", 1, "train_feats", "valid_feats",
        multimodal_model_type="vi-model1", train_img_vecs=None, valid_img_vecs=None, model_opt="opt", fields={})
    inspect.signature(Optim.__init__).bind(None, "adam", 0.002, 5, lr_decay=0.5, start_decay_at=8, beta1=0.9, beta2=0.999, adagrad_accum=0,
                                           decay_method="", warmup_steps=4000, model_size=500)
    for m in ("train", "validate", "epoch_step", "drop_checkpoint", "drop_metric_scores"):
        assert callable(getattr(TrainerMultimodal, m))
    inspect.signature(TrainerMultimodal.drop_checkpoint).bind(None, "opt", 1, {}, "valid_stats", overwrite=True, checkpoint_type="last")
    inspect.signature(TrainerMultimodal.drop_metric_scores).bind(None, "opt", 1, {}, "valid_stats", overwrite=True, checkpoint_type="last")
    st = VIStatistics("vi-model1")
    for a in ("image_feats_loss", "image_feats_cos", "image_pixels_loss", "image_pixels_acc", "n_updates", "ppl", "accuracy", "output", "log"):
        assert hasattr(st, a), a
