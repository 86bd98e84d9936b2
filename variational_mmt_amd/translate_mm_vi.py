"""The translation driver of the MI355X build: `python -m variational_mmt_amd.translate_mm_vi -model ckpt.pt -src test.en -output
pred.txt -path_to_test_img_feats test.hdf5 -gpu 0 [-beam_size 5 -n_best 1 -alpha .. -beta .. -min_length .. -max_length ..
-batch_size 30 -replace_unk -verbose -tgt ref.de -report_bleu]` -- the flags of the reference's translate_mm_vi.py (:20-27;
opts.translate_opts + translate_mm_vi_opts) on the mirrored surface: checkpoint -> load_test_model, source file -> build_dataset ->
OrderedIterator (sorted within a batch), TranslatorMultimodalVI.translate_batch (beam search on the device for every sentence of the
batch), TranslationBuilder -> one line per hypothesis in corpus order (reference: translate_mm_vi.py:54-169).

The reference forces one sentence per batch (:80-82); here -batch_size is honoured (identical beams either way,
tests/test_gpu_beam.py).  The test image features are opened (the flag is required, and the file must hold a row per sentence) but,
as in the reference, decoding itself never reads them: z is the mean of q(z|x) / p(z|x) (TranslatorMultimodalVI.py:128-131)."""
import argparse
import math
import sys

import torch

from . import install_as_onmt, opts


def main(argv=None):
    ap = argparse.ArgumentParser(description="translate_mm_vi.py", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    opts.add_md_help_argument(ap)
    opts.translate_opts(ap)
    opts.translate_mm_vi_opts(ap)
    opt = ap.parse_args(argv)
    if opt.gpu < 0:
        ap.error("the MI355X build has no CPU path: pass -gpu 0")
    onmt = install_as_onmt()
    defaults = argparse.ArgumentParser()
    opts.model_opts(defaults)
    torch.cuda.set_device(opt.gpu)
    print("Using GPU")
    opt.cuda = True
    opt.multimodal_model_type = "vi-model1"
    with onmt.h5tables.open_file(opt.path_to_test_img_feats, mode="r") as f:
        n_rows = max(int(f.root[n].shape[0]) for n in ("global_feats", "logits") if n in f.root)
    fields, model, model_opt = onmt.ModelConstructor.load_test_model(opt, defaults.parse_known_args([])[0].__dict__)
    print("Translating with multimodal_model_type: %s" % model_opt.multimodal_model_type)
    data = onmt.io.build_dataset(fields, opt.data_type, opt.src, opt.tgt, src_dir=opt.src_dir, use_filter_pred=False)
    if len(data) > n_rows:
        raise ValueError("%d sentences but %d rows of image features in %s" % (len(data), n_rows, opt.path_to_test_img_feats))
    it = onmt.io.OrderedIterator(dataset=data, device=opt.gpu, batch_size=opt.batch_size, train=False, sort=False,
                                 sort_within_batch=True, shuffle=False)
    tr = onmt.translate.TranslatorMultimodalVI(model, fields, beam_size=opt.beam_size, n_best=opt.n_best,
                                               global_scorer=onmt.translate.GNMTGlobalScorer(opt.alpha, opt.beta),
                                               max_length=opt.max_length, copy_attn=getattr(model_opt, "copy_attn", False), cuda=True,
                                               beam_trace=opt.dump_beam != "", min_length=opt.min_length,
                                               multimodal_model_type=model_opt.multimodal_model_type)
    builder = onmt.translate.TranslationBuilder(data, fields, opt.n_best, opt.replace_unk, opt.tgt)
    score = words = 0.0
    n_sent = 0
    with open(opt.output, "w", encoding="utf-8") as out:
        for batch in it:
            for t in builder.from_batch(tr.translate_batch(batch, data)):
                score += t.pred_scores[0]
                words += len(t.pred_sents[0])
                out.write("\n".join(" ".join(p) for p in t.pred_sents[:opt.n_best]) + "\n")
                n_sent += 1
                if opt.verbose:
                    sys.stdout.write(t.log(n_sent))
    print("PRED AVG SCORE: %.4f, PRED PPL: %.4f" % (score / max(words, 1), math.exp(-score / max(words, 1))))
    if opt.report_bleu and opt.tgt:
        from .onmt.bleu import multi_bleu       # restatement of tools/multi-bleu.perl (the reference shells out to it, :39-45)
        with open(opt.output, encoding="utf-8") as h, open(opt.tgt, encoding="utf-8") as r:
            print(">> " + (multi_bleu(h.readlines(), [r.readlines()])["line"] or ""))
    return n_sent


if __name__ == "__main__":
    main(sys.argv[1:])
