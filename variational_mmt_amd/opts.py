"""Command-line flags of the training / translation drivers, as one table per flag group.

Flag-compatible with the reference's `opts.py` (model_opts :6-117, train_opts :206-378, _train_mm_vi_opts :477-529,
translate_opts :381-465, translate_mm_vi_opts :536-544): same spellings (the reference mixes `-flag` and `--flag`; both are kept
exactly), destinations, defaults, types, choices and required-ness, so that the run scripts' command lines
(run_translated_m30k_only.sh:46-71, run_additional_data.sh) parse to the same namespace.  `tests/test_opts_flags.py` compares
every parser action with the reference's when it is mounted, and with the committed listing tests/golden/opts_flags.json
otherwise.  Flags that select code outside the VI_Model1 hot path still PARSE here; the model constructor refuses them
(ModelConstructor.make_vi_model_mmt).

A row is (flag, kind, default, extra):  kind = a type (int / float / str), None (plain store, string), "flag" (store_true),
"dead" (deprecated: using it is an error) or an argparse.Action subclass; extra = dict(choices=, nargs=, required=, help=).
"""
import argparse

from .onmt.modules.SRU import CheckSRU


class DeprecateAction(argparse.Action):
    """a flag that only exists to say that it is gone (reference opts.py:601-609)"""

    def __init__(self, option_strings, dest, help=None, **kw):
        super(DeprecateAction, self).__init__(option_strings, dest, nargs=0, help=help, **kw)

    def __call__(self, parser, namespace, values, flag_name):
        raise argparse.ArgumentTypeError("Flag '%s' is deprecated. %s" % (flag_name, self.help or ""))


class MarkdownHelpAction(argparse.Action):
    """`-md`: print the flag list as a markdown bullet list and exit (reference opts.py:547-598 prints its help the same way)"""

    def __init__(self, option_strings, dest=argparse.SUPPRESS, default=argparse.SUPPRESS, **kw):
        super(MarkdownHelpAction, self).__init__(option_strings=option_strings, dest=dest, default=default, nargs=0, **kw)

    def __call__(self, parser, namespace, values, option_string=None):
        print("# %s\n" % (parser.description or parser.prog))
        for grp in parser._action_groups:
            rows = [a for a in grp._group_actions if a.option_strings]
            if not rows:
                continue
            print("## %s\n" % grp.title)
            for a in rows:
                d = "" if a.default in (None, argparse.SUPPRESS) else " [%s]" % (a.default,)
                print("* **%s**%s: %s" % (" ".join(a.option_strings), d, a.help or ""))
            print("")
        parser.exit()


def _add(group, rows):
    for flag, kind, default, extra in rows:
        kw = dict(extra)
        if kind == "flag":
            group.add_argument(flag, action="store_true", **kw)
        elif kind == "dead":
            group.add_argument(flag, action=DeprecateAction, **kw)
        elif isinstance(kind, type) and issubclass(kind, argparse.Action):
            group.add_argument(flag, action=kind, default=default, **kw)
        elif kind is None:
            group.add_argument(flag, default=default, **kw)
        else:
            group.add_argument(flag, type=kind, default=default, **kw)


_MODEL = [
    ("Model-Embeddings", [
        ("-src_word_vec_size", int, 500, dict(help="source word-vector size")),
        ("-tgt_word_vec_size", int, 500, dict(help="target word-vector size")),
        ("-word_vec_size", int, -1, dict(help="one size for both sides (overrides the two above when not -1)")),
        ("-share_decoder_embeddings", "flag", None, dict(help="tie the generator to the target embeddings (not on the MI355X path)")),
        ("-share_embeddings", "flag", None, dict(help="one embedding table for both sides (not on the MI355X path)")),
        ("-position_encoding", "flag", None, dict(help="sinusoidal position encoding (transformer models; not on the MI355X path)")),
    ]),
    ("Model-Embedding Features", [
        ("-feat_merge", str, "concat", dict(choices=["concat", "sum", "mlp"], help="how word-feature embeddings are merged")),
        ("-feat_vec_size", int, -1, dict(help="feature embedding size (-1: from the exponent)")),
        ("-feat_vec_exponent", float, 0.7, dict(help="feature embedding size = N^exponent")),
    ]),
    ("Model- Encoder-Decoder", [
        ("-model_type", None, "text", dict(help="source modality; the VI_Model1 path is 'text'")),
        ("-encoder_type", str, "rnn", dict(choices=["rnn", "brnn", "mean", "transformer", "cnn"], help="encoder: rnn | brnn on the MI355X path")),
        ("-decoder_type", str, "rnn", dict(choices=["rnn", "transformer", "cnn", "doubly-attentive-rnn"], help="decoder: rnn on the MI355X path")),
        ("-layers", int, -1, dict(help="layers of encoder and decoder (overrides the two below when not -1)")),
        ("-enc_layers", int, 2, dict(help="encoder layers")),
        ("-dec_layers", int, 2, dict(help="decoder layers")),
        ("-rnn_size", int, 500, dict(help="LSTM hidden size (computed padded to a multiple of 32 on the MI355X path)")),
        ("-cnn_kernel_width", int, 3, dict(help="kernel width of the cnn encoder / decoder")),
        ("-input_feed", int, 1, dict(help="input feeding (the VI_Model1 decoder does not use it)")),
        ("-rnn_type", CheckSRU, "LSTM", dict(type=str, choices=["LSTM", "GRU", "SRU"], help="gate type; LSTM on the MI355X path")),
        ("-brnn", "dead", None, dict(help="Use `encoder_type`.")),
        ("-brnn_merge", None, "concat", dict(choices=["concat", "sum"], help="how a brnn's directions are merged")),
        ("-context_gate", str, None, dict(choices=["source", "target", "both"], help="context gate (not on the MI355X path)")),
    ]),
    ("Model- Attention", [
        ("-global_attention", str, "general", dict(choices=["dot", "general", "mlp"], help="attention score; 'general' (Luong) on the MI355X path")),
        ("-copy_attn", "flag", None, dict(help="copy attention (not on the MI355X path)")),
        ("-copy_attn_force", "flag", None, dict(help="force copying")),
        ("-reuse_copy_attn", "flag", None, dict(help="reuse the standard attention for copying")),
        ("-coverage_attn", "flag", None, dict(help="coverage attention (not on the MI355X path)")),
        ("-lambda_coverage", float, 1, dict(help="coverage loss weight")),
    ]),
]

_TRAIN = [
    ("General", [
        ("-data", None, None, dict(required=True, help="path prefix of the preprocessed .train.N.pt / .valid.N.pt / .vocab.pt files")),
        ("-save_model", None, "model", dict(help="checkpoint path prefix (<prefix>_acc_X_ppl_X_eN.pt, <prefix>_MostCurrentModel.pt, ...)")),
        ("-gpuid", int, [], dict(nargs="+", help="GPU to use (one id; data parallelism runs one process per GPU)")),
        ("-seed", int, -1, dict(help="random seed (> 0 for reproducible runs)")),
    ]),
    ("Early stopping", [
        ("-early_stopping_criteria", str, "perplexity", dict(choices=["perplexity", "bleu", "meteor"], help="model-selection metric")),
        ("-src", str, None, dict(help="validation source text (needed for bleu / meteor)")),
        ("-tgt", str, None, dict(help="validation reference text (needed for bleu / meteor)")),
        ("-evaluate_every_n_model_updates", int, 500, dict(help="translate + score the validation set every N updates")),
        ("-patience", int, 20, dict(help="evaluations without improvement before stopping")),
        ("-beam_size", int, 1, dict(help="beam size of the model-selection translations")),
        ("-start_early_stopping_at", int, 0, dict(help="first update at which model selection starts")),
        ("-overwrite_model_file", "flag", None, dict(help="keep one checkpoint file instead of one per epoch")),
    ]),
    ("Initialization", [
        ("-start_epoch", int, 1, dict(help="first epoch")),
        ("-param_init", float, 0.1, dict(help="every parameter ~ U(-param_init, param_init)")),
        ("-train_from", str, "", dict(help="checkpoint to continue from")),
        ("-finetune", "flag", None, dict(help="with -train_from: keep the weights, take model options and optimiser from THIS command line")),
        ("-pre_word_vecs_enc", None, None, dict(help="pretrained source embeddings (not on the MI355X path)")),
        ("-pre_word_vecs_dec", None, None, dict(help="pretrained target embeddings (not on the MI355X path)")),
        ("-fix_word_vecs_enc", "flag", None, dict(help="freeze the source embeddings")),
        ("-fix_word_vecs_dec", "flag", None, dict(help="freeze the target embeddings")),
    ]),
    ("Optimization- Type", [
        ("-batch_size", int, 64, dict(help="sentences (or tokens, see -batch_type) per minibatch")),
        ("-batch_type", None, "sents", dict(choices=["sents", "tokens"], help="unit of -batch_size")),
        ("-normalization", None, "sents", dict(choices=["sents", "tokens"], help="loss normalisation")),
        ("-accum_count", int, 1, dict(help="gradient accumulation (1 on the MI355X path)")),
        ("-valid_batch_size", int, 32, dict(help="validation minibatch")),
        ("-max_generator_batches", int, 32, dict(help="target rows per loss shard in the reference; the fused vocabulary sweep takes all rows at once")),
        ("-epochs", int, 13, dict(help="last epoch")),
        ("-optim", None, "sgd", dict(choices=["sgd", "adagrad", "adadelta", "adam"], help="optimiser; adam (and sgd) on the MI355X path")),
        ("-adagrad_accumulator_init", float, 0, dict(help="adagrad accumulator start value")),
        ("-max_grad_norm", float, 5, dict(help="global gradient-norm clip")),
        ("-dropout", float, 0.3, dict(help="dropout between LSTM layers and on the decoder output")),
        ("-word_dropout", float, 0.0, dict(help="must stay 0 (broken in the reference)")),
        ("-truncated_decoder", int, 0, dict(help="truncated BPTT (0 on the MI355X path)")),
        ("-adam_beta1", float, 0.9, dict(help="Adam beta1")),
        ("-adam_beta2", float, 0.999, dict(help="Adam beta2")),
        ("-label_smoothing", float, 0.0, dict(help="label smoothing (0 on the MI355X path)")),
    ]),
    ("Optimization- Rate", [
        ("-learning_rate", float, 1.0, dict(help="initial learning rate")),
        ("-learning_rate_decay", float, 0.5, dict(help="factor applied once decay starts")),
        ("-start_decay_at", int, 8, dict(help="decay from this epoch on (or when validation perplexity rises)")),
        ("-start_checkpoint_at", int, 0, dict(help="first epoch that writes a checkpoint")),
        ("-decay_method", str, "", dict(choices=["noam"], help="alternative schedule")),
        ("-warmup_steps", int, 4000, dict(help="noam warm-up")),
    ]),
    ("Logging", [
        ("-report_every", int, 50, dict(help="progress line every N updates")),
        ("-exp_host", str, "", dict(help="crayon server (unused here)")),
        ("-exp", str, "", dict(help="crayon experiment name")),
    ]),
    ("Speech", [
        ("-sample_rate", int, 16000, dict(help="audio source modality only")),
        ("-window_size", float, .02, dict(help="audio source modality only")),
    ]),
]

_TRAIN_MM_VI = [
    ("Variational multi-modal NMT", [
        ("-path_to_train_img_feats", None, None, dict(required=True, help="HDF5 file with the training image features (/global_feats)")),
        ("-path_to_valid_img_feats", None, None, dict(required=True, help="HDF5 file with the validation image features")),
        ("-dropout_imgs", float, 0.5, dict(help="dropout on image features (unused by vi-model1)")),
        ("--multimodal_model_type", str, "vi-model1", dict(required=True, choices=["vi-model1"], help="model family")),
        ("--z_latent_dim", int, None, dict(required=True, help="size of the latent variable z")),
        ("--use_standardised_image_features", "flag", None, dict(help="(x - mean) / std with the two files below")),
        ("-path_to_mean_train_img_feats", None, None, dict(help="HDF5 file with /global_feats_mean")),
        ("-path_to_std_train_img_feats", None, None, dict(help="HDF5 file with /global_feats_stds")),
        ("--conditional", "flag", None, dict(help="conditional prior p(z|x) and posterior q(z|x,y,v)")),
        ("--use_kl_annealing", "flag", None, dict(help="anneal the KL weight")),
        ("--use_kl_freebits", "flag", None, dict(help="free bits: max(KL, margin)")),
        ("--kl_freebits_margin", float, 0.0, dict(help="free-bits margin")),
        ("--kl_annealing_warmup_steps", int, 500, dict(help="updates before the KL weight starts to grow")),
        ("--kl_annealing_start", float, 0.0, dict(help="initial KL weight")),
        ("--kl_annealing_increment", float, 0.0001, dict(help="KL weight increment per update")),
        ("--image_loss", str, "logprob", dict(choices=["cosine", "logprob", "categorical", "none"], help="image term; logprob on the MI355X path")),
        ("-two_step_image_prediction", "flag", None, dict(help="local features only (not on the MI355X path)")),
        ("-use_rgb_images", "flag", None, dict(help="categorical image loss only")),
        ("-path_to_train_img_vecs", None, None, dict(help="categorical image loss only")),
        ("-path_to_valid_img_vecs", None, None, dict(help="categorical image loss only")),
        ("--use_posterior_image_features", "flag", None, dict(help="use /logits (1000-d)")),
        ("--use_global_image_features", "flag", None, dict(help="use /global_feats (pool5, 2048-d)")),
        ("--use_local_image_features", "flag", None, dict(help="use /local_feats (not on the MI355X path)")),
        ("-non_shared_inference_network", "flag", None, dict(help="separate encoder for the inference network (not on the MI355X path)")),
    ]),
]

_TRANSLATE = [
    ("Model", [
        ("-model", None, None, dict(required=True, help="checkpoint")),
    ]),
    ("Data", [
        ("-data_type", None, "text", dict(help="source modality")),
        ("-src", None, None, dict(required=True, help="source text, one sentence per line")),
        ("-src_dir", None, "", dict(help="image / audio source directory")),
        ("-tgt", None, None, dict(help="reference text")),
        ("-output", None, "pred.txt", dict(help="where the translations go")),
        ("-report_bleu", "flag", None, dict(help="score the output with multi-bleu")),
        ("-report_rouge", "flag", None, dict(help="score the output with rouge")),
        ("-dynamic_dict", "flag", None, dict(help="copy-attention models")),
        ("-share_vocab", "flag", None, dict(help="copy-attention models")),
    ]),
    ("Beam", [
        ("-beam_size", int, 5, dict(help="beam size")),
        ("-min_length", int, 0, dict(help="shortest output")),
        ("-max_length", int, 100, dict(help="longest output")),
        ("-max_sent_length", "dead", None, dict(help="Use `-max_length` instead")),
        ("-alpha", float, 0., dict(help="GNMT length penalty")),
        ("-beta", float, -0., dict(help="GNMT coverage penalty")),
        ("-replace_unk", "flag", None, dict(help="replace <unk> by the most attended source word")),
    ]),
    ("Logging", [
        ("-verbose", "flag", None, dict(help="print every translation")),
        ("-attn_debug", "flag", None, dict(help="print attention")),
        ("-dump_beam", str, "", dict(help="file for beam dumps")),
        ("-n_best", int, 1, dict(help="hypotheses per sentence")),
    ]),
    ("Efficiency", [
        ("-batch_size", int, 30, dict(help="sentences per batch")),
        ("-gpu", int, -1, dict(help="GPU id")),
    ]),
    ("Speech", [
        ("-sample_rate", int, 16000, dict(help="audio only")),
        ("-window_size", float, .02, dict(help="audio only")),
        ("-window_stride", float, .01, dict(help="audio only")),
        ("-window", None, "hamming", dict(help="audio only")),
    ]),
]

_TRANSLATE_MM_VI = [
    ("Variational multi-modal NMT", [
        ("-path_to_test_img_feats", None, None, dict(required=True, help="HDF5 file with the test-set image features")),
    ]),
]


def _groups(parser, table):
    for title, rows in table:
        _add(parser.add_argument_group(title), rows)


def model_opts(parser):
    _groups(parser, _MODEL)


def train_opts(parser):
    _groups(parser, _TRAIN)


def train_mm_vi_model1_opts(parser):
    _groups(parser, _TRAIN_MM_VI)


def translate_opts(parser):
    _groups(parser, _TRANSLATE)


def translate_mm_vi_opts(parser):
    _groups(parser, _TRANSLATE_MM_VI)


def add_md_help_argument(parser):
    parser.add_argument("-md", action=MarkdownHelpAction, help="print the flags as markdown and exit")


def finalise(opt):
    """what the driver does right after parsing (train_mm_vi_model1.py:40-48): the umbrella flags overwrite the per-side ones"""
    if opt.word_vec_size != -1:
        opt.src_word_vec_size = opt.tgt_word_vec_size = opt.word_vec_size
    if opt.layers != -1:
        opt.enc_layers = opt.dec_layers = opt.layers
    opt.brnn = opt.encoder_type == "brnn"
    return opt
