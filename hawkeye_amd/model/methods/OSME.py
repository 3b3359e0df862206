"""OSME (one-squeeze multi-excitation) plugin for the MI355X heads.

Interface kept from the reference (model/methods/OSME.py:8-64) because trainers and checkpoints depend on it:
  * `OSMENet(config)` with `config.num_attention`, `config.num_classes`; `forward(x) -> (logits, parts[N,P,1024])`;
  * attributes `backbone` / `osme` / `classifier`, and inside `osme` the lists `blocks[p].block` (gate MLP) and
    `fcs[p]` (part head) - these names ARE the state_dict keys (`osme.blocks.0.block.0.weight`, ...).

What runs where: the ResNet-101 trunk, the two tiny gate MLPs and the 100352->1024 part heads are PyTorch-ROCm
(MIOpen / hipBLASLt); the squeeze (one global average pool shared by every gate - the reference pools once per gate)
and the channel re-scaling of all P gates (one pass over x instead of P broadcast multiplies) are the gfx950 kernels
`hk_osme_gap` and `hk_osme_scale_fwd/bwd`.
"""
import torch
import torch.nn as nn

from ... import functional as HF
from ..backbone import resnet101
from ..registry import MODEL
from ..utils import wide_linear

REDUCTION = 16          # squeeze ratio of the excitation MLP (reference: reduce_ratio = 16)
PART_DIM = 1024         # width of each part descriptor


def _excitation(channels, ratio):
    """channels -> channels/ratio -> channels, sigmoid gate."""
    hidden = channels // ratio
    return nn.Sequential(nn.Linear(channels, hidden), nn.ReLU(inplace=True), nn.Linear(hidden, channels), nn.Sigmoid())


class OSME_block(nn.Module):
    """One excitation gate.  `block` is the MLP; `avg_pool` exists only for attribute parity with the reference
    (the pooling itself is done once, outside, by hk_osme_gap)."""

    def __init__(self, channels, ratio):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.block = _excitation(channels, ratio)

    def gate(self, squeezed):
        """squeezed [N,C] -> per-channel gate in (0,1), [N,C]"""
        return self.block(squeezed)

    def forward(self, x):
        """Stand-alone use (the reference's OSME_block.forward): gate(x) * x."""
        gates = self.gate(HF.osme_gap(x)).unsqueeze(0)
        return HF.osme_scale(x, gates)[0]


class OSME(nn.Module):
    """P gates sharing one squeeze, each followed by its own fully connected part head."""

    def __init__(self, in_channels, out_channels=PART_DIM, feature_shape=(7, 7), num_attention=2):
        super().__init__()
        side = feature_shape if isinstance(feature_shape, tuple) else (feature_shape, feature_shape)
        flat = in_channels * side[0] * side[1]
        self.blocks = nn.ModuleList(OSME_block(in_channels, REDUCTION) for _ in range(num_attention))
        self.fcs = nn.ModuleList(nn.Linear(flat, out_channels) for _ in range(num_attention))

    def forward(self, x):
        batch = x.shape[0]
        squeezed = HF.osme_gap(x)                                          # [N,C], one pass over x
        gates = torch.stack([b.gate(squeezed) for b in self.blocks])       # [P,N,C]
        scaled = HF.osme_scale(x, gates)                                   # [P,N,C,H,W], one pass over x
        parts = [wide_linear(head, scaled[p].reshape(batch, -1)) for p, head in enumerate(self.fcs)]
        return sum(parts), torch.stack(parts, dim=1)                       # [N,1024], [N,P,1024]


@MODEL.register
class OSMENet(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.num_attention, self.num_classes = config.num_attention, config.num_classes
        trunk = resnet101(pretrained=True)
        self.backbone = nn.Sequential(*list(trunk.children())[:-2])        # up to layer4: [N,2048,7,7] at 224^2
        self.osme = OSME(2048, PART_DIM, feature_shape=7, num_attention=self.num_attention)
        self.classifier = nn.Linear(PART_DIM, self.num_classes)

    def forward(self, x):
        fused, parts = self.osme(self.backbone(x))
        return self.classifier(fused), parts
