"""PyTorch-CPU restatement of the R3M hot path (test infrastructure — see oracle/__init__.py). Every function cites the
reference lines it follows; pinned against the reference's own code by tests/golden (make_golden.py + test_oracle.py)."""
import torch
import torch.nn as nn

from . import resnet_ref

epsilon = 1e-8   # /root/reference/r3m/trainer.py:18, models_r3m.py:18


class R3MRef(nn.Module):
    """/root/reference/r3m/models/models_r3m.py:21-107 with langweight handled by LanguageRewardRef (no DistilBERT:
    sentence features are an input, BASELINE config 3 'frozen DistilBERT text feats')."""

    def __init__(self, size=34, hidden_dim=1024, l2weight=1.0, l1weight=1.0, langweight=1.0, tcnweight=0.0, l2dist=True,
                 lr=1e-4, lang_size=768):
        super().__init__()
        self.l2weight, self.l1weight, self.tcnweight, self.langweight = l2weight, l1weight, tcnweight, langweight
        self.l2dist, self.size, self.num_negatives = l2dist, size, 3          # models_r3m.py:26-34
        self.cs = nn.CosineSimilarity(1)                                       # models_r3m.py:37
        self.convnet = {18: resnet_ref.resnet18, 34: resnet_ref.resnet34, 50: resnet_ref.resnet50}[size]()  # :44-52
        self.outdim = 2048 if size == 50 else 512
        self.normlayer = resnet_ref.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])   # :61
        self.convnet.fc = nn.Identity()                                        # :62
        self.convnet.train()                                                   # :63
        params = list(self.convnet.parameters())
        if langweight > 0.0:
            self.lang_rew = LanguageRewardRef(self.outdim, hidden_dim, lang_size)              # :67-72
            params += list(self.lang_rew.parameters())
        self.encoder_opt = torch.optim.Adam(params, lr=lr)                     # :76

    def forward(self, obs):
        obs = obs.float() / 255.0                                              # :97
        return self.convnet(self.normlayer(obs))                               # :98-99

    def sim(self, t1, t2):                                                     # :102-107
        if self.l2dist:
            return -torch.linalg.norm(t1 - t2, dim=-1)
        return self.cs(t1, t2)


class LanguageRewardRef(nn.Module):
    """/root/reference/r3m/models/models_language.py:37-55."""

    def __init__(self, im_dim, hidden_dim, lang_dim):
        super().__init__()
        self.pred = nn.Sequential(nn.Linear(im_dim * 2 + lang_dim, hidden_dim), nn.ReLU(inplace=True),
                                  nn.Linear(hidden_dim, hidden_dim), nn.ReLU(inplace=True),
                                  nn.Linear(hidden_dim, hidden_dim), nn.ReLU(inplace=True),
                                  nn.Linear(hidden_dim, hidden_dim), nn.ReLU(inplace=True),
                                  nn.Linear(hidden_dim, 1))

    def forward(self, e0, eg, le):
        return self.pred(torch.cat([e0, eg, le], -1)).squeeze(-1)


def r3m_loss_ref(model, alle, tcn_perm=None, lang_feats=None, lang_mask=None, lang_perm=None):
    """/root/reference/r3m/trainer.py:42-152 on given embeddings alle [B,5,D], with the permutations as inputs:
    lang_perm [9,B] in draw order (a,b,c) x 3 (:86-92), tcn_perm [6,B] in draw order (es0, es2) x 3 (:136-137).
    Returns (full_loss, metrics dict of python floats, scores [15,B] or None)."""
    metrics = {}
    bs = alle.shape[0]
    alles = alle.reshape(bs * 5, -1)
    e0, eg, es0, es1, es2 = alle[:, 0], alle[:, 1], alle[:, 2], alle[:, 3], alle[:, 4]            # :43-47
    l2loss = torch.linalg.norm(alles, ord=2, dim=-1).mean()                                       # :52-54
    l1loss = torch.linalg.norm(alles, ord=1, dim=-1).mean()
    l0loss = torch.linalg.norm(alles, ord=0, dim=-1).mean()
    metrics["l2loss"], metrics["l1loss"], metrics["l0loss"] = l2loss.item(), l1loss.item(), l0loss.item()
    full_loss = 0
    full_loss = full_loss + model.l2weight * l2loss                                               # :58-59
    full_loss = full_loss + model.l1weight * l1loss
    scores = None
    if model.langweight > 0:                                                                      # :64-118
        G = lambda a, b: model.lang_rew(a, b, lang_feats)
        pos = [G(e0, eg), G(e0, es1), G(e0, es2)]                                                 # :72-74
        negs = [[G(e0, e0)], [G(e0, es0)], [G(e0, es1)]]                                          # :80-82
        for k in range(model.num_negatives):                                                      # :86-92
            for j, other in enumerate((eg, es1, es2)):
                p = lang_perm[3 * k + j]
                negs[j].append(G(e0[p], other[p]))
        scores = torch.stack(pos + [negs[j][0] for j in range(3)] +
                             [negs[j][1 + k] for k in range(model.num_negatives) for j in range(3)])
        rew = []
        for j in range(3):
            sn = torch.stack(negs[j], -1)
            rew.append(-torch.log(epsilon + (torch.exp(pos[j]) / (epsilon + torch.exp(pos[j]) + torch.exp(sn).sum(-1)))))  # :101-103
            metrics[f"rewacc{j+1}"] = (1.0 * (sn.max(-1)[0] < pos[j])).mean().item()              # :111-113
        rewloss = ((rew[0] + rew[1] + rew[2]) / 3 * lang_mask).mean()                             # :104-110
        metrics["rewloss"] = rewloss.item()
        full_loss = full_loss + model.langweight * rewloss
    if model.tcnweight > 0:                                                                       # :122-150
        s02, s12, s01 = model.sim(es2, es0), model.sim(es2, es1), model.sim(es1, es0)             # :127-129
        neg0, neg2 = [], []
        for k in range(model.num_negatives):                                                      # :135-139
            neg0.append(model.sim(es0, es0[tcn_perm[2 * k]]))
            neg2.append(model.sim(es2, es2[tcn_perm[2 * k + 1]]))
        neg0, neg2 = torch.stack(neg0, -1), torch.stack(neg2, -1)
        sl1 = -torch.log(epsilon + (torch.exp(s12) / (epsilon + torch.exp(s02) + torch.exp(s12) + torch.exp(neg2).sum(-1))))  # :144
        sl2 = -torch.log(epsilon + (torch.exp(s01) / (epsilon + torch.exp(s01) + torch.exp(s02) + torch.exp(neg0).sum(-1))))  # :145
        smooth = ((sl1 + sl2) / 2.0).mean()
        metrics["tcnloss"] = smooth.item()
        metrics["aligned"] = ((1.0 * (s02 < s12)) * (1.0 * (s01 > s02))).mean().item()            # :147
        full_loss = full_loss + model.tcnweight * smooth
    metrics["full_loss"] = float(full_loss.item())
    return full_loss, metrics, scores


def train_step_ref(model, frames, tcn_perm=None, lang_feats=None, lang_mask=None, lang_perm=None, eval=False):
    """One Trainer.update (/root/reference/r3m/trainer.py:25-162) on frames [B,5,3,224,224] (0..255)."""
    model.eval() if eval else model.train()
    bs = frames.shape[0]
    alles = model(frames.reshape(bs * 5, 3, 224, 224))                                            # :40-41
    full_loss, metrics, _ = r3m_loss_ref(model, alles.reshape(bs, 5, -1), tcn_perm, lang_feats, lang_mask, lang_perm)
    if not eval:
        model.encoder_opt.zero_grad()                                                             # :156-158
        full_loss.backward()
        model.encoder_opt.step()
    return metrics


def resize_center_crop_ref(obs, size=256, crop=224):
    """The non-224 branch of R3M.forward (/root/reference/r3m/models/models_r3m.py:85-90): transforms.Resize(256) then
    CenterCrop(224) applied to obs/255. torchvision (0.8.2, un-vendored) restated for tensors: resize = bilinear
    F.interpolate (align_corners=False, no antialias) with the smaller edge -> size and the other edge int(size * long / short);
    centre crop offsets int(round((dim - crop) / 2)). Input 0..255, output 0..255 (scaled back, to compose with R3MRef.forward)."""
    x = obs.float() / 255.0
    h, w = x.shape[-2:]
    if w <= h:
        nw, nh = size, int(size * h / w)
    else:
        nh, nw = size, int(size * w / h)
    x = torch.nn.functional.interpolate(x, size=(nh, nw), mode="bilinear", align_corners=False)
    top, left = int(round((nh - crop) / 2.0)), int(round((nw - crop) / 2.0))
    return x[..., top:top + crop, left:left + crop] * 255.0
