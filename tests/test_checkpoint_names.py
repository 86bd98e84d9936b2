"""CPU: the engine's parameter arena carries exactly the reference model's state-dict names and shapes (checkpoint
contract).  Uses the real reference when it is mounted (build container), else the oracle's table."""
import pytest

from oracle import vi1_oracle as O
from variational_mmt_amd.engine import Dims


@pytest.mark.parametrize("brnn,layers,cond", [(True, 1, False), (False, 2, False), (True, 2, False), (True, 1, True), (False, 2, True)])
def test_param_names_and_shapes(brnn, layers, cond):
    c = O.Cfg(vs=23, vt=29, emb=10, hid=12, z=6, img=2048, layers=layers, brnn=brnn, conditional=cond)
    d = Dims(c.vs, c.vt, c.emb, c.hid, c.z, c.img, c.layers, c.brnn, conditional=cond)
    wg, ng = d.param_shapes()
    mine = {n: tuple(s) for n, s in wg + ng}
    assert mine == {n: tuple(s) for n, s in O.param_shapes(c).items()}
    assert all("inf_net_image.scale" in n for n, _ in ng)          # H6: never receive gradients
    from oracle import ref_harness as RH
    if not RH.available():
        pytest.skip("reference not mounted")
    opt = RH.make_opt(src_word_vec_size=c.emb, tgt_word_vec_size=c.emb, rnn_size=c.hid, z_latent_dim=c.z, enc_layers=layers,
                      dec_layers=layers, encoder_type="brnn" if brnn else "rnn", dropout=0.0, conditional=cond)
    model, _ = RH.build_model(opt, c.vs, c.vt)
    ref = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    # encoder_tgt shares the decoder's embedding table: the reference lists it under a second name (Engine.state_dict adds it)
    alias = "encoder_tgt.embeddings.make_embedding.emb_luts.0.weight"
    if cond:
        assert ref.pop(alias) == mine["decoder.embeddings.make_embedding.emb_luts.0.weight"]
    assert mine == ref
