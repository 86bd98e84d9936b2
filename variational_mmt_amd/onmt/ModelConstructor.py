"""Mirror of onmt.ModelConstructor.make_vi_model_mmt (reference: onmt/ModelConstructor.py:328-620)."""
import torch

from ..engine import Dims
from .Models import NMTVIModel
from .Utils import MODEL_TYPES, use_gpu


def make_vi_model_mmt(model_opt, fields, gpu, checkpoint=None):
    """Same signature / side effects as the reference: infers the image-feature size from the feature-file name
    (:350-354, mutating model_opt.global_image_features_dim), builds the model, loads `checkpoint['model']` /
    `['generator']` or initialises EVERY parameter uniform(-param_init, param_init) (:598-603, H7)."""
    assert model_opt.model_type == "text", "only text source modality is on the MI355X path"
    assert model_opt.multimodal_model_type in MODEL_TYPES
    if getattr(model_opt, "use_posterior_image_features", False):
        feat = 1000
    else:
        feat = 4096 if "vgg" in str(model_opt.path_to_train_img_feats).lower() else 2048
    model_opt.global_image_features_dim = feat
    if not gpu:
        raise RuntimeError("variational_mmt_amd needs -gpuid (MI355X); there is no CPU path")
    for flag, ok in (("rnn_type", "LSTM"), ("global_attention", "general")):
        if getattr(model_opt, flag, ok) != ok:
            raise NotImplementedError("%s=%s is outside the hot path (reference run scripts use %s)" % (flag, getattr(model_opt, flag), ok))
    for flag in ("copy_attn", "coverage_attn", "share_embeddings", "share_decoder_embeddings", "non_shared_inference_network",
                 "use_local_image_features", "two_step_image_prediction"):
        if getattr(model_opt, flag, False):
            raise NotImplementedError("-%s is outside the hot path" % flag)
    if getattr(model_opt, "context_gate", None) is not None:
        raise NotImplementedError("-context_gate is outside the hot path")
    assert getattr(model_opt, "word_dropout", 0.0) == 0.0, "word_dropout > 0 is broken in the reference (H4); only 0 is supported"
    assert model_opt.enc_layers == model_opt.dec_layers, "decoder init requires enc_layers == dec_layers (Models.py:1158-1174)"
    assert model_opt.src_word_vec_size == model_opt.tgt_word_vec_size
    brnn = getattr(model_opt, "brnn", False) or getattr(model_opt, "encoder_type", "rnn") == "brnn"
    dims = Dims(vs=len(fields["src"].vocab), vt=len(fields["tgt"].vocab), emb=model_opt.src_word_vec_size,
                hid=model_opt.rnn_size, z=model_opt.z_latent_dim, img=feat, layers=model_opt.enc_layers, brnn=brnn,
                dropout=model_opt.dropout, conditional=bool(getattr(model_opt, "conditional", False)))
    dtype = getattr(model_opt, "compute_dtype", "bf16")
    dev = "cuda:%d" % (model_opt.gpuid[0] if getattr(model_opt, "gpuid", None) else torch.cuda.current_device())
    model = NMTVIModel(dims, dtype=dtype, device=dev, param_init=(0.0 if checkpoint is not None else model_opt.param_init),
                       seed=getattr(model_opt, "seed", 0) if getattr(model_opt, "seed", 0) > 0 else 0,
                       conditional=getattr(model_opt, "conditional", False), image_loss_type=getattr(model_opt, "image_loss", "logprob"))
    if checkpoint is not None:
        print("Loading model parameters.")
        sd = dict(checkpoint["model"])
        sd.update({"generator." + k: v for k, v in checkpoint["generator"].items()})
        # strict: a missing or misnamed key is an error.  The one known alias: the conditional model's encoder_tgt shares the
        # decoder's embedding table (ModelConstructor.py:456-457), so the reference's state dict lists that tensor twice.
        alias = "encoder_tgt.embeddings.make_embedding.emb_luts.0.weight"
        if alias in sd:
            if not torch.equal(sd[alias], sd["decoder.embeddings.make_embedding.emb_luts.0.weight"]):
                raise RuntimeError("checkpoint: encoder_tgt and decoder embeddings differ (they are one shared table)")
            del sd[alias]
        model.load_state_dict(sd, strict=True)
    return model


def load_test_model(opt, dummy_opt):
    """onmt/ModelConstructor.py:148-174: checkpoint -> (fields, model in evaluation mode, model options); options the checkpoint's
    `opt` predates are filled from `dummy_opt` (a dict of defaults, translate_mm_vi.py:55-57)."""
    from . import io
    checkpoint = torch.load(opt.model, map_location="cpu", weights_only=False)
    fields = io.load_fields_from_vocab(checkpoint["vocab"], data_type=getattr(opt, "data_type", "text"))
    model_opt = checkpoint["opt"]
    for k, v in dict(dummy_opt).items():
        if k not in model_opt.__dict__:
            model_opt.__dict__[k] = v
    if getattr(opt, "multimodal_model_type", "vi-model1") not in MODEL_TYPES:
        raise NotImplementedError("only the variational multi-modal models (%s) are on the MI355X path" % ", ".join(MODEL_TYPES))
    print("Building variational multi-modal model...")
    model = make_vi_model_mmt(model_opt, fields, use_gpu(opt), checkpoint)
    model.eval()
    return fields, model, model_opt
