"""Weight init / lenient checkpoint loading with the reference's behaviour
(model/utils.py:5-28): they decide the random-init statistics of synthetic runs."""
import torch.nn as nn


def initialize_weights(m):
    if isinstance(m, nn.Conv2d):
        nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.BatchNorm2d):
        nn.init.constant_(m.weight, 1)
        nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.Linear):
        nn.init.kaiming_normal_(m.weight.data)
        if m.bias is not None:
            nn.init.constant_(m.bias.data, val=0)


def load_state_dict(model, state_dict):
    """Copy only the entries whose name and shape match (model/utils.py:24-28)."""
    own = model.state_dict()
    own.update({k: v for k, v in state_dict.items() if k in own and v.shape == own[k].shape})
    model.load_state_dict(own)
