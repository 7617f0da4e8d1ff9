"""TEST INFRASTRUCTURE: a tiny DistilBERT + a stub tokenizer, so that the text side of R3M (LangEncoder,
/root/reference/r3m/models/models_language.py:13-35) can be executed without the pretrained `distilbert-base-uncased` files
(absent from the image; no network). The model is the real `transformers.DistilBertModel` architecture at dim 768 (the
reward head's lang_dim) with ONE layer and a 64-word vocabulary; its weights come from oracle/detgen.py (hash generator), so
golden features regenerate bit-identically wherever the same transformers/torch build runs."""
import numpy as np
import torch

from . import detgen

VOCAB = ["[PAD]", "[CLS]", "[SEP]", "[UNK]"] + ("open close the a drawer door pick up put down cup bottle knife fork spoon plate pan pot lid turn on "
                                                 "off tap stove wash cut stir pour water onion take from to in into table shelf fridge hand left "
                                                 "right move push pull slide lift drop box bag phone book pen wipe cloth sponge").split()


class WhitespaceTokenizer:
    """The calling convention of a HuggingFace tokenizer as LangEncoder uses it (models_language.py:30):
    tok(list_of_str, return_tensors='pt', padding=True) -> {'input_ids', 'attention_mask'}; [CLS] words [SEP], pad id 0."""

    def __init__(self):
        self.index = {w: i for i, w in enumerate(VOCAB)}

    def __call__(self, langs, return_tensors="pt", padding=True):
        assert return_tensors == "pt" and padding is True
        rows = [[1] + [self.index.get(w, 3) for w in s.lower().split()] + [2] for s in langs]
        n = max(len(r) for r in rows)
        ids = torch.tensor([r + [0] * (n - len(r)) for r in rows], dtype=torch.long)
        am = torch.tensor([[1] * len(r) + [0] * (n - len(r)) for r in rows], dtype=torch.long)
        return {"input_ids": ids, "attention_mask": am}


def tiny_distilbert():
    from transformers import DistilBertConfig, DistilBertModel
    cfg = DistilBertConfig(vocab_size=len(VOCAB), dim=768, n_layers=1, n_heads=4, hidden_dim=256, max_position_embeddings=32)
    model = DistilBertModel(cfg)
    sd = {}
    for k, v in model.state_dict().items():
        if not v.is_floating_point():
            sd[k] = v
        elif k.endswith("LayerNorm.weight") or k.endswith("layer_norm.weight"):
            sd[k] = torch.from_numpy(detgen.uniform("tt" + k, tuple(v.shape), 0.8, 1.2))
        else:
            sd[k] = torch.from_numpy(detgen.uniform("tt" + k, tuple(v.shape), -0.08, 0.08))
    model.load_state_dict(sd)
    return model.eval()


SENTENCES = ["open the drawer", "pick up the cup from the table", "", "wash the pan", "put the knife down",
             "turn on the tap", "close the fridge door", "stir the pot"]
