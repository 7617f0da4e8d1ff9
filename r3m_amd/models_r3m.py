"""R3M nn.Module — same constructor, attributes and methods as the reference class
(/root/reference/r3m/models/models_r3m.py:21-107), with every FLOP on the HIP path:

    R3M(device, lr, hidden_dim, size=34, l2weight=1.0, l1weight=1.0, langweight=1.0, tcnweight=0.0, l2dist=True, bs=16)
      .forward(obs, num_ims=1, obs_shape=[3,224,224]) -> [F, outdim]      obs in 0..255, NCHW        (models_r3m.py:84-100)
      .sim(t1, t2) -> [B]                              -||a-b||_2 or cosine                          (models_r3m.py:102-107)
      .get_reward(e0, es, sentences) -> (score[B], {})                                               (models_r3m.py:78-81)
      .encoder_opt                                     Adam over convnet (+ lang_rew) params          (models_r3m.py:76)
      attributes l2weight l1weight tcnweight langweight l2dist size num_negatives outdim           (models_r3m.py:26-34)

state_dict keys are the reference's (`convnet.<torchvision names>`, `lang_rew.pred.{0,2,4,6,8}.{weight,bias}`), so
snapshots and `~/.r3m/*/model.pt` interchange.
"""
import torch
import torch.nn as nn

from .encoder import HipResNet
from .optim import FusedAdam

epsilon = 1e-8

IMAGENET_MEAN = (0.485, 0.456, 0.406)   # models_r3m.py:61 — baked into the stem kernel (csrc/conv.hip)
IMAGENET_STD = (0.229, 0.224, 0.225)


class _NormalizeSpec(nn.Module):
    """Stateless stand-in for torchvision.transforms.Normalize (models_r3m.py:61): carries the constants; the arithmetic
    ((x/255 - mean)/std) is fused into the encoder's stem kernel."""

    def __init__(self, mean, std):
        super().__init__()
        self.mean, self.std = tuple(mean), tuple(std)


class R3M(nn.Module):
    def __init__(self, device, lr, hidden_dim, size=34, l2weight=1.0, l1weight=1.0, langweight=1.0, tcnweight=0.0,
                 l2dist=True, bs=16, precision="fp32", max_live_forwards=1):
        super().__init__()
        self.device = device
        self.use_tb = False
        self.l2weight = l2weight
        self.l1weight = l1weight
        self.tcnweight = tcnweight   # weight on the time-contrastive loss
        self.l2dist = l2dist         # -L2 distance (True) or cosine similarity (False)
        self.langweight = langweight
        self.size = size
        self.num_negatives = 3

        self.cs = nn.CosineSimilarity(1)
        self.bce = nn.BCELoss(reduction="none")
        self.sigm = nn.Sigmoid()

        if size not in (18, 34, 50):
            # size == 0 (ViT) is dead code in the reference (NameError: AutoModel never imported, models_r3m.py:53-59)
            raise ValueError(f"R3M: unsupported encoder size {size!r}; the HIP path implements ResNet-18/34/50")
        # "bf16": mixed-precision encoder (BASELINE configs[2], [4]); max_live_forwards: encoder.HipResNet (1 is all Trainer.update needs)
        self.convnet = HipResNet(size, precision=precision, max_live_forwards=max_live_forwards)
        self.outdim = self.convnet.outdim
        self.normlayer = _NormalizeSpec(IMAGENET_MEAN, IMAGENET_STD)
        self.convnet.train()         # models_r3m.py:63
        owners = [self.convnet]

        if self.langweight > 0.0:
            from .models_language import LangEncoder, LanguageReward
            self.lang_enc = LangEncoder(self.device, 0, 0)
            self.lang_rew = LanguageReward(None, self.outdim, hidden_dim, self.lang_enc.lang_size, simfunc=self.sim, precision=precision)
            owners.append(self.lang_rew)

        self.encoder_opt = FusedAdam(owners, lr=lr)

    def get_reward(self, e0, es, sentences):
        le = self.lang_enc(sentences)
        return self.lang_rew(e0, es, le)

    def forward(self, obs, num_ims=1, obs_shape=[3, 224, 224]):
        if list(obs_shape) != [3, 224, 224]:
            # models_r3m.py:85-90: Resize(256) + CenterCrop(224) before Normalize — off the pre-training hot path (frames arrive
            # as 224x224; example.py users with other sizes land here). One HIP gather pass that computes only the crop window.
            from .augment import resize_center_crop
            obs = resize_center_crop(obs.reshape(-1, *obs.shape[-3:]))
        # "Input must be [0, 255], [3,224,224]" (models_r3m.py:96): x.float()/255 -> Normalize -> convnet, all in the engine
        return self.convnet(obs)

    def sim(self, tensor1, tensor2):
        if self.l2dist:
            return -torch.linalg.norm(tensor1 - tensor2, dim=-1)
        return self.cs(tensor1, tensor2)
