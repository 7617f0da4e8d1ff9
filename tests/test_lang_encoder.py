"""not gpu: the text side. The pretrained DistilBERT files are absent from the image, so the real
`transformers.DistilBertModel` architecture is run at a tiny size with a stub tokenizer (oracle/tiny_text.py); expected values
come from the REFERENCE's own LangEncoder.forward executed on the same stand-ins (tests/golden/lang_encoder_tiny.npz, G7)."""
import os

import numpy as np
import pytest
import torch

from util import rel_err


def _encoder(**kw):
    from oracle import tiny_text
    from r3m_amd.models_language import LangEncoder
    return LangEncoder("cpu", 0, 0, **kw).use_backend(tiny_text.WhitespaceTokenizer(), tiny_text.tiny_distilbert())


def test_features_match_the_reference_lang_encoder(golden_dir):
    from oracle import tiny_text
    g = np.load(os.path.join(golden_dir, "lang_encoder_tiny.npz"))
    enc = _encoder()
    f_all = enc(tiny_text.SENTENCES)
    assert f_all.shape == (8, 768) and not f_all.requires_grad
    assert rel_err(f_all.numpy(), g["feats_all"])[0] < 1e-5
    # the padding quirk (models_language.py:34): mean(1) runs over padded positions, so the SAME sentence gets a
    # different feature in a batch with a longer neighbour — reproduced, not "fixed"
    f_short = enc([tiny_text.SENTENCES[0], tiny_text.SENTENCES[3]])
    assert rel_err(f_short.numpy(), g["feats_short"])[0] < 1e-5
    assert float((f_short[0] - f_all[0]).abs().max()) > 0.1
    # numpy array of strings, as a DataLoader collates them (`langs.tolist()`, models_language.py:24-27)
    assert rel_err(enc(np.array(tiny_text.SENTENCES[3:6])).numpy(), g["feats_array_input"])[0] < 1e-5
    # tensors are frozen precomputed features and pass through untouched (BASELINE configs[2])
    t = torch.randn(3, 768)
    assert enc(t) is t


def test_mask_padding_flag_makes_features_batch_independent():
    from oracle import tiny_text
    enc = _encoder(mask_padding=True)
    f_all = enc(tiny_text.SENTENCES)
    f_short = enc([tiny_text.SENTENCES[0], tiny_text.SENTENCES[3]])
    assert rel_err(f_short[0].numpy(), f_all[0].numpy())[0] < 1e-5 and rel_err(f_short[1].numpy(), f_all[3].numpy())[0] < 1e-5
    ref = _encoder()                         # un-padded batches: both poolings agree
    assert rel_err(f_short.numpy(), ref([tiny_text.SENTENCES[0], tiny_text.SENTENCES[3]]).numpy())[0] < 1e-5


def test_once_per_step_equals_the_fifteen_call_form_and_cache():
    """The reference re-encodes the same sentences inside each of its 15 get_reward calls per step (trainer.py:72-92 ->
    models_r3m.py:78-81). The text model is frozen and deterministic here (kept in eval mode whatever model.train() does), so ONE
    pass per step gives the features of all 15 calls; a filled cache needs no pass at all."""
    from oracle import tiny_text
    enc = _encoder()
    enc.train()                               # what Trainer.update's model.train() does to every submodule (trainer.py:31)
    assert not enc._hf[1].training
    once = enc(tiny_text.SENTENCES)
    assert enc.encoder_calls == 1
    for _ in range(15):
        assert torch.equal(enc(tiny_text.SENTENCES), once)
    assert enc.encoder_calls == 16
    with pytest.raises(ValueError, match="mask_padding"):
        enc.precompute(tiny_text.SENTENCES)   # reference pooling is batch-dependent: a per-sentence cache would not be faithful
    cached = _encoder(mask_padding=True)
    n = cached.precompute(tiny_text.SENTENCES + tiny_text.SENTENCES[:3], batch_size=3)
    assert n == len(set(tiny_text.SENTENCES)) and cached.encoder_calls == 3
    cached._hf = None                         # cache hits must not touch the transformer
    got = cached([tiny_text.SENTENCES[5], tiny_text.SENTENCES[0]])
    assert cached.encoder_calls == 3
    direct = _encoder(mask_padding=True)([tiny_text.SENTENCES[5], tiny_text.SENTENCES[0]])
    assert rel_err(got.numpy(), direct.numpy())[0] < 1e-5


def test_missing_pretrained_files_fail_with_a_clear_message():
    from r3m_amd.models_language import LangEncoder
    enc = LangEncoder("cpu", 0, 0)
    with pytest.raises(RuntimeError, match="precomputed"):
        enc(["open the drawer"])


def _pretrained_distilbert():
    """(tokenizer, model) of distilbert-base-uncased when the local HuggingFace cache holds it, else None (no network here)."""
    try:
        from transformers import AutoModel, AutoTokenizer
        tok = AutoTokenizer.from_pretrained("distilbert-base-uncased", local_files_only=True)
        model = AutoModel.from_pretrained("distilbert-base-uncased", local_files_only=True)
        return tok, model.eval()
    except Exception:  # noqa: BLE001  (OSError / EnvironmentError / ValueError depending on the transformers version)
        return None


def test_real_distilbert_when_it_is_in_the_local_cache():
    """arms itself where `distilbert-base-uncased` exists in the local HF cache (skipped in this image, which
    has neither the files nor a network). LangEncoder, loading the model by itself, must reproduce the reference formula
    (/root/reference/r3m/models/models_language.py:29-34: tokenise with padding, one transformer pass under no_grad,
    last_hidden_state.mean(1) over ALL positions) on the real weights, in both pooling modes."""
    hf = _pretrained_distilbert()
    if hf is None:
        pytest.skip("distilbert-base-uncased is not in the local HuggingFace cache")
    from r3m_amd.models_language import LangEncoder
    tok, model = hf
    sentences = ["open the drawer", "pick up the red cup and put it on the shelf next to the window", "c", "turn off the tap"]
    with torch.no_grad():
        enc_in = tok(sentences, return_tensors="pt", padding=True)
        hidden = model(**enc_in).last_hidden_state
        ref_all = hidden.mean(1)                                        # the reference: padding included
        w = enc_in["attention_mask"].to(hidden.dtype).unsqueeze(-1)
        ref_masked = (hidden * w).sum(1) / w.sum(1)
    enc = LangEncoder("cpu", 0, 0)                                      # loads the SAME files itself (local_files_only)
    got = enc(sentences)
    assert got.shape == (4, 768) and not got.requires_grad and enc.encoder_calls == 1
    assert rel_err(got.numpy(), ref_all.numpy())[0] < 1e-5
    assert float((ref_all - ref_masked).abs().max()) > 1e-3              # the two poolings differ on a padded batch
    got_m = LangEncoder("cpu", 0, 0, mask_padding=True)(sentences)
    assert rel_err(got_m.numpy(), ref_masked.numpy())[0] < 1e-5
    # batch-independence of the masked form; batch-dependence of the reference form (App. C quirk) on real weights
    alone = LangEncoder("cpu", 0, 0, mask_padding=True)([sentences[0]])
    assert rel_err(alone.numpy(), got_m[:1].numpy())[0] < 1e-4
    alone_ref = LangEncoder("cpu", 0, 0)([sentences[0]])
    assert float((alone_ref - got[:1]).abs().max()) > 1e-3
    # frozen: stays in eval mode through model.train() (no dropout noise between the 15 get_reward calls of a step)
    enc.train()
    assert torch.equal(enc(sentences), got)
